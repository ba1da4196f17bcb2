"""world_size-2 test of the problem-per-rank sharding and the final gather (gloo on CPU).

The GPU step cannot run here; each rank produces its read-out with the CPU oracle instead, which is
enough to cover what the N>1 path adds: problem assignment, the all_gather and the max-over-ranks clock."""

import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from pnmol import batch  # noqa: E402


def _readout(kappa):
    import pnmol_oracle as o

    dt, K, N = 2.0**-6, 3, 12
    pde = o.heat_1d_discretized(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=kappa, kernel=o.SquareExponential())
    s = o.WhiteNoiseEK1(num_derivatives=1, steprule=o.Constant(dt), spatial_kernel=o.Matern52() + o.WhiteNoise())
    sol = s.solve(pde)
    means, stds = o.read_mean_and_std(sol, s.E0)
    return np.concatenate([means.ravel(), stds.ravel()])


def _worker(rank, world, port, out_dir):
    import sys
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    for p in (root / "pnmol-experiments_amd", root / "oracle"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = batch.shard(world, rank, world)
        assert mine == [rank]
        payload = _readout(batch.diffusion_sweep(mine[0], world))
        everything = batch.gather_readouts(payload, dist)
        slowest = batch.max_over_ranks(1.0 + rank, dist)
        if rank == 0:
            np.save(os.path.join(out_dir, "gathered.npy"), everything)
            np.save(os.path.join(out_dir, "clock.npy"), np.array([slowest]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape[0] == 2
    for g in range(2):
        np.testing.assert_allclose(got[g], _readout(batch.diffusion_sweep(g, 2)), rtol=1e-12)
    assert np.load(tmp_path / "clock.npy")[0] == 2.0
    assert not np.allclose(got[0], got[1])


def test_sweep_and_shards():
    ks = [batch.diffusion_sweep(g, 8) for g in range(8)]
    np.testing.assert_allclose(ks, 0.01 * 10.0 ** (np.arange(8) / 7.0))   # SURVEY section 8d batch definition
    assert batch.diffusion_sweep(0, 1) == 0.05
    assert sorted(sum((batch.shard(8, r, 4) for r in range(4)), [])) == list(range(8))
    np.testing.assert_array_equal(batch.gather_readouts(np.arange(3.0)), np.arange(3.0)[None])
