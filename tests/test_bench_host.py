"""Host-side pieces of bench.py that run without a GPU."""

import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_blas_thread_limit_survives_omp_num_threads_1():
    """`torch.distributed.run` starts every rank with OMP_NUM_THREADS=1.  bench.limit_blas_threads() must not raise the
    pool of an OpenBLAS that started with one thread (it segfaulted in the next LAPACK call when it did)."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import bench, numpy as np, scipy.linalg\n"
        "bench.limit_blas_threads()\n"
        "a = np.random.default_rng(0).standard_normal((600, 600)); a = a @ a.T + 600 * np.eye(600)\n"
        "c = scipy.linalg.cho_factor(a, lower=True)\n"
        "print('ok', np.isfinite(c[0]).all())\n" % (str(ROOT), str(ROOT / "pnmol-experiments_amd"))
    )
    env = dict(os.environ, OMP_NUM_THREADS="1", LOCAL_WORLD_SIZE="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ok True" in out.stdout


def test_f_alg_matches_survey_table():
    """F_alg(D, m, n) = m^3/3 + m^2 D + D^2 m + 4 n D^2 + 8 (D m + m^2); 1.70 GF at N=512, nu=2."""
    sys.path.insert(0, str(ROOT))
    import bench

    assert abs(bench.f_alg(1536, 514, 3) / 1e9 - 1.70) < 0.01
    assert abs(bench.f_alg(768, 258, 3) / 1e9 - 0.218) < 0.002
    assert abs(bench.f_alg(3072, 1026, 3) / 1e9 - 13.4) < 0.1
