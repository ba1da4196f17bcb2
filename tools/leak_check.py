import sys, gc
sys.path.insert(0, "/root/repo/pnmol-experiments_amd"); sys.path.insert(0, "/root/repo")
import torch, bench
bench.MESH_N = 256
def free_mb():
    return torch.cuda.mem_get_info()[0] / 1e6
base = None
for it in range(12):
    pde, solver = bench.build_problem(0.05, 12)
    state = solver.initialize(pde)
    flt, dev = solver._device_filter, state.y.device_state
    solver._ensure_error_model(pde, bench.DT)
    flt.steps(dev, 12, bench.DT)
    dev.cov_sqrtm()
    del state, flt, dev, solver, pde
    gc.collect()
    f = free_mb()
    if it == 1: base = f
    print(it, f"{f:.0f} MB free")
print("drift after warm-up:", base - f, "MB")
