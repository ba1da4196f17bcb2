import sys, time
sys.path.insert(0, "/root/repo/pnmol-experiments_amd"); sys.path.insert(0, "/root/repo")
import bench
bench.MESH_N = int(sys.argv[1])
pde, solver = bench.build_problem(0.05, 105)
state = solver.initialize(pde)
flt, dev = solver._device_filter, state.y.device_state
solver._ensure_error_model(pde, bench.DT)
flt.steps(dev, 5, bench.DT)
flt.prepare_steps(dev, 100, bench.DT)
for rep in range(4):
    t0 = time.perf_counter(); flt.steps_begin(dev, 100, bench.DT); t1 = time.perf_counter()
    flt.steps_end(dev); t2 = time.perf_counter()
    print(f"N={bench.MESH_N} rep {rep}: begin {1e3*(t1-t0):.2f} ms, end {1e3*(t2-t1):.2f} ms, device {flt.last_steps_ms():.2f} ms")
