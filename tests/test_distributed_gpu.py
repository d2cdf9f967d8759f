"""
The N > 1 path driven through the HIP engine: two FRESH child processes (spawn; gloo backend, both on
device 0 -- RCCL refuses two ranks on one device, and the one-GPU box has one) run GrInt / GrLessInt /
calculate_transmission with gaunegf_amd.distributed enabled: cyclic shard of the grid, the engine's *_dev entry
points on each rank's shard, ONE sum all-reduce.  The result must equal the single-process integral.
"""
import os
import socket

import numpy as np
import pytest

from helpers import chain_lead, random_system

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _systems():
    from gaunegf_amd.surfGTester import surfGTest
    from gaunegf_amd.surfG1D import surfG
    N, nc = 48, 6
    F, S = random_system(N, 12)
    inds = [list(range(nc)), list(range(N - nc, N))]
    g_const = surfGTest(F, S, inds, -0.1j)
    aL = chain_lead(nc, 41); aR = chain_lead(nc, 42)
    g_chain = surfG(F, S, inds, taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()],
                    alphas=[aL[0], aR[0]], aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]],
                    bOverlaps=[aL[3], aR[3]], eta=1e-3)
    g_chain.force_iters = 30
    M = 37                                         # not divisible by the world size: ragged shards
    E = np.linspace(-1.5, 1.5, M) + 0.0j
    w = (np.cos(np.arange(M)) + 1.5) * (3.0 / M) + 0.0j
    return F, S, g_const, g_chain, E, w


def _run_all(F, S, g_const, g_chain, E, w):
    from gaunegf_amd.integrate import GrInt, GrIntSegments, GrLessInt, GrLessIntSegments
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    # several integrals as one pass (and, sharded, one all-reduce): ragged segments, one of a single point
    segs = [(E[:5], w[:5]), (E[5:6], w[5:6]), (E[6:], w[6:])]
    seg = {"gr_seg": np.stack(GrIntSegments(F, S, g_const, segs)), "gless_seg": np.stack(GrLessIntSegments(F, S, g_const, segs, 0)),
           "gr_seg_chain": np.stack(GrIntSegments(F, S, g_chain, segs))}
    out = {"gr_const": GrInt(F, S, g_const, E, w), "gless_const": GrLessInt(F, S, g_const, E, w, -1),
           "gr_chain": GrInt(F, S, g_chain, E, w), "gless_chain": GrLessInt(F, S, g_chain, E, w, 0),
           "T": calculate_transmission(F, S, SigmaCalculator(g_const.sig[0], g_const.sig[1]), np.real(E)),
           "gr_one": GrInt(F, S, g_const, E[:1], w[:1])}      # one point: the second rank's shard is empty
    out.update(seg)
    return out


def _worker(rank, world, port, q, backend="gloo"):
    import torch
    import torch.distributed as dist
    from gaunegf_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"                 # both ranks on the one visible GPU
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D.enable(single_rank_ok=(world == 1))
        res = _run_all(*_systems())
        maps = open("/proc/self/maps").read()
        if rank == 0:
            q.put((res, "libnegf_hip.so" in maps))
    finally:
        D.disable()
        dist.destroy_process_group()


def _spawn_and_compare(world, backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    res = None
    for _ in range(60):                                 # fail as soon as a rank has died, not after a long wait
        try:
            res, loaded = q.get(timeout=5)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    if res is None:
        for p in procs:
            p.kill()
        pytest.fail("a rank died (its traceback is on stderr)")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert loaded
    ref = _run_all(*_systems())
    for k in ("gr_const", "gless_const", "gr_chain", "gless_chain", "gr_one", "gr_seg", "gless_seg", "gr_seg_chain"):
        assert np.linalg.norm(res[k] - ref[k]) <= 1e-13 * np.linalg.norm(ref[k]), k
    assert np.array_equal(res["T"], ref["T"])          # the all-gather of per-energy scalars is exact


def test_two_ranks_through_the_engine_match_one_process(engine):
    _spawn_and_compare(2, "gloo")


def test_rccl_legs_on_one_gpu(engine):
    """The nccl (= RCCL) branches of gaunegf_amd.distributed -- the in-place all-reduce of the device-resident partial
    sum (sharded_device_sum) and the device all-gather of per-energy scalars (all_gather_shards) -- in a process
    group of ONE rank on the one GPU of the box (RCCL refuses two ranks on one device): the same code path a rank of
    an 8-GPU job runs, collectives included, must reproduce the local integrals."""
    _spawn_and_compare(1, "nccl")


def test_bench_gpus_2_rehearsal_carries_the_strong_scaling_blocks():
    """``bench.py --gpus N`` (the driver's multi-GPU command) keeps the weak C3 line and adds ``extra.c4_strong`` /
    ``extra.c5_strong``: BASELINE's multi-GPU configurations as FIXED steps through the product's own sharding
    (distributed.enable, one all-reduce / all-gather per pass of the engine).  Rehearsed here with two ranks on the one GPU over
    gloo (NEGF_BENCH_REHEARSAL=1): both blocks are present with their timing, communication and roofline fields, and the
    sharded results equal rank 0's un-sharded evaluation of the same step to 1e-13."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NEGF_BENCH_REHEARSAL="1", NEGF_BENCH_CHECK_LOCAL="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--energies", "24", "--no-cpu", "--no-warm"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "comm_ms" in line
    for key, pts in (("c4_strong", 742), ("c5_strong", 1024)):
        b = line["extra"][key]
        assert b["scaling"] == "strong" and b["n_gpus"] == 2 and b["ms_per_step"] > 0
        assert abs(b["value"] * b["ms_per_step"] * 1e-3 - pts) < 1e-6 * pts          # the FIXED grid, whatever N
        # (C4: contour + real axis as one pass with ONE all-reduce of both sums; C5: an all-reduce per spin block + an all-gather)
        assert b["comm_ms"] is not None and (b["collectives_per_step"] == 1 if key == "c4_strong" else b["collectives_per_step"] >= 2)
        assert b["sharded_vs_local_rel"] is not None and b["sharded_vs_local_rel"] <= 1e-13, b["sharded_vs_local_rel"]
        assert 0 < b["inverse"]["frac"] <= 1 and "family_ms_per_step" in b
