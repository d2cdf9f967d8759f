"""
Multi-process energy sharding on CPU (gloo, world_size 2): the N>1 path of
gaunegf_amd.distributed -- cyclic shard, per-rank partial sums, ONE sum all-reduce; per-energy scalars by ONE all-gather of the
shards -- must reproduce the single-process integral.  The per-rank partial integral is
computed with the oracle here (no GPU in this container); on the GPU box the same
``sharded_sum`` wraps the HIP engine (integrate.py).
"""
import os
import socket

import numpy as np
import pytest

import oracle
from helpers import random_system, const_sigma_pair


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from gaunegf_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D.enable()
        assert D.is_active() and D.rank_world() == (rank, world)
        N, M = 24, 37                     # M not divisible by world: ragged shards
        F, S = random_system(N, 5)
        inds, s1, s2 = const_sigma_pair(N, S, 4)
        g = oracle.ConstSigma(F, S, inds, -0.1j)
        E, w = oracle.real_axis_grid(-3.0, 0.4, M, 300.0)
        full = D.sharded_sum(lambda idx: oracle.GrInt(F, S, g, E[idx], w[idx]), M)
        # per-energy scalars
        st = g.sigmaTot(0.0)
        gam = [1j * (g.sigma(0, i) - g.sigma(0, i).conj().T) for i in (0, 1)]
        Tm = D.sharded_map(lambda idx: np.array([oracle.transmission_restricted(e, F, S, st, gam[0], gam[1])
                                                 for e in E[idx]]), M)
        # per-energy rows with a tail (T and four spin parts; DOS per site) and an empty shard (m < world)
        rows = D.sharded_map(lambda idx: np.stack([np.asarray(idx) * 1.5, -np.asarray(idx, dtype=float)], axis=1), M, (2,))
        one = D.sharded_map(lambda idx: np.array([7.25] * len(np.arange(1)[idx])), 1)
        # an empty shard (M < world) must still take part in the collective
        tiny = D.sharded_sum(lambda idx: oracle.GrInt(F, S, g, E[:1][idx], w[:1][idx]), 1)
        if rank == 0:
            q.put((full, Tm, tiny, rows, one))
    finally:
        D.disable()
        dist.destroy_process_group()


def test_sharded_integral_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, Tm, tiny, rows, one = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, M = 24, 37
    F, S = random_system(N, 5)
    inds, s1, s2 = const_sigma_pair(N, S, 4)
    g = oracle.ConstSigma(F, S, inds, -0.1j)
    E, w = oracle.real_axis_grid(-3.0, 0.4, M, 300.0)
    ref = oracle.GrInt(F, S, g, E, w)
    assert np.linalg.norm(full - ref) / np.linalg.norm(ref) < 1e-13
    st = g.sigmaTot(0.0)
    gam = [1j * (g.sigma(0, i) - g.sigma(0, i).conj().T) for i in (0, 1)]
    Tref = np.array([oracle.transmission_restricted(e, F, S, st, gam[0], gam[1]) for e in E])
    assert np.array_equal(Tm, Tref)            # an all-gather of the shards moves the values, exactly
    assert np.array_equal(rows, np.stack([np.arange(M) * 1.5, -np.arange(M, dtype=float)], axis=1))
    assert np.array_equal(one, np.array([7.25]))
    assert np.linalg.norm(tiny - oracle.GrInt(F, S, g, E[:1], w[:1])) < 1e-13


def test_shard_indices_partition():
    from gaunegf_amd.distributed import shard_indices
    for m in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(m, r, world) for r in range(world)]
            allidx = np.sort(np.concatenate(parts)) if parts else np.array([])
            assert np.array_equal(allidx, np.arange(m))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
