#!/usr/bin/env python3
"""
bench.py -- energy-points/sec of the NEGF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): synthetic random Hermitian F / S with
N_orb = 200, energy-independent contacts (Gamma = 0.2 eV on 20 + 20 orbitals plus
-i 1e-9 S everywhere, matTools.formSigma), 1000 Gauss-Legendre energy points on
[-3, 3] eV PER GPU (weak scaling: rank r owns the points r, r+W, ... of a W*1000
point grid).  One "step" = one GrInt pass over the local shard: for every energy
assemble E S - F - Sigma, invert (complex128, partial pivoting), accumulate w G --
followed, for N > 1, by ONE RCCL sum all-reduce of the N_orb x N_orb result.
F, S, Sigma, the energy grid and the weights are resident in HBM before the timed
region; the result stays in HBM.

The JSON line carries
  roofline     : the inverse kernel family vs the FP64 matrix-core peak.  achieved =
                 8 N^3 flops per energy point x points per launch
                 / average launch duration, timed with hipEvents recorded by the
                 library on the stream the kernels run on, during the timed steps.
  cpu_baseline : the numpy oracle (the reference's CPU restatement) timed on this
                 host on a bounded sample of the same energies.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense FP64 matrix peak (vector FP64 peak is the same)
N_ORB = 200
M_PER_GPU = 1000
NC = 20
SEED = 2


def make_system(N, seed, nc):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N))
    F = (A + A.T) * (1.0 / np.sqrt(2 * N)) * 2
    B = rng.standard_normal((N, N))
    S = np.eye(N) + 0.1 * (B + B.T) / np.sqrt(2 * N)
    inds = [list(range(nc)), list(range(N - nc, N))]
    return F, S, inds


def legendre_grid(M, lo=-3.0, hi=3.0):
    from scipy.special import roots_legendre
    x, w = roots_legendre(M)
    mid = (hi - lo) / 2
    return mid * (np.real(x) + 1) + lo, mid * w


def cpu_baseline(F, S, inds, E, w, budget_s=12.0):
    """Time the oracle's GrInt (plain numpy loop, solve(A, I)) on a bounded sample.

    The BLAS thread count that is fastest for this matrix size on this host is used
    (all cores is NOT the fastest for N ~ 200: oversubscription), so the baseline is the
    best the reference's CPU path can do here; `cores` reports that thread count."""
    import oracle
    g = oracle.ConstSigma(F, S, inds, -0.1j)
    ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    best_t, best_per = ncpu, None
    by_threads = {}
    cands = sorted({t for t in (1, 4, 8, 16, 32, 64, ncpu) if t <= ncpu}) if threadpool_limits else [ncpu]
    for t in cands:
        ctx = threadpool_limits(limits=t) if threadpool_limits else None
        try:
            oracle.GrInt(F, S, g, E[:2], w[:2])
            t0 = time.perf_counter()
            oracle.GrInt(F, S, g, E[:6], w[:6])
            per = (time.perf_counter() - t0) / 6
        finally:
            if ctx is not None:
                ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else ctx.unregister()
        by_threads[str(t)] = round(1.0 / per, 2)
        if best_per is None or per < best_per:
            best_t, best_per = t, per
    n = int(max(16, min(len(E), budget_s / max(best_per, 1e-6))))
    idx = np.linspace(0, len(E) - 1, n).astype(int)
    # a whole grid can take less than the ~10 s a stable CPU number needs: repeat the pass
    passes = int(max(1, min(50, np.ceil(budget_s / max(n * best_per, 1e-6)))))
    ctx = threadpool_limits(limits=best_t) if threadpool_limits else None
    try:
        t0 = time.perf_counter()
        for _ in range(passes):
            oracle.GrInt(F, S, g, E[idx], w[idx])
        dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else ctx.unregister()
    return {"value": n * passes / dt, "unit": "energy-points/s", "cores": int(best_t), "kind": "port",
            "points_per_s_by_blas_threads": by_threads,      # 6-point probes; "1" is the scalar figure
            "sample": f"{passes} x {n} of {len(E)} energies of the same N_orb={F.shape[0]} workload, numpy {np.__version__} "
                      f"solve(A,I) loop (oracle.GrInt), best of BLAS threads {cands} = {best_t} "
                      f"(host has {ncpu} logical CPUs), {dt:.2f} s"}


def pmc_traffic_bytes():
    """HBM bytes per inverse launch from the committed PMC profile of this same command
    (profiles/r01_pmc/..., separate --pmc passes): 2 x FETCH_SIZE (gfx950 reports half of a
    wide coalesced read stream, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, both in KiB."""
    path = os.path.join(ROOT, "profiles", "r01_pmc", "c2_bench_pmc_per_launch_avg.json")
    try:
        data = json.load(open(path))
        for name, ctr in data.items():
            if "gj_blocked_kernel" in name and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                return (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--norb", type=int, default=N_ORB)
    ap.add_argument("--energies", type=int, default=M_PER_GPU)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--extra", action="store_true", help="also time N_orb=500 (north-star target case)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # NEGF_BENCH_REHEARSAL=1: several ranks share the visible GPUs and talk over gloo -- a dry run
    # of the N > 1 control flow on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("NEGF_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gaunegf_amd.engine import Engine
    from gaunegf_amd.matTools import formSigma

    N, M = args.norb, args.energies
    F, S, inds = make_system(N, SEED, NC)
    sig = [formSigma(inds[0], -0.1j, N, S), formSigma(inds[1], -0.1j, N, S)]
    # global grid of world*M points, cyclic shard (distributed.shard_indices)
    Eg, wg = legendre_grid(M * world)
    E_loc = np.ascontiguousarray(Eg[rank::world], dtype=np.complex128)
    w_loc = np.ascontiguousarray(wg[rank::world], dtype=np.complex128)

    eng = Engine(local_rank)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    eng.set_system(F, S)
    h = eng.sigma_const(sig)
    dev = torch.device("cuda", local_rank)
    E_dev = torch.view_as_complex(torch.from_numpy(E_loc.view(np.float64).reshape(-1, 2).copy())).to(dev)
    w_dev = torch.view_as_complex(torch.from_numpy(w_loc.view(np.float64).reshape(-1, 2).copy())).to(dev)
    out = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    out_real = torch.view_as_real(out)

    def step():
        eng.gr_int_dev(h, M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
        if world > 1:
            dist.all_reduce(out_real, op=dist.ReduceOp.SUM)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile(True)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    inv_ms, inv_launches = eng.profile_read("inverse")
    asm_ms, _ = eng.profile_read("assemble")
    acc_ms, _ = eng.profile_read("accumulate")
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the all-reduced result equals the sum over the global grid on rank 0's view
    info = eng.last_info_dev(M)
    assert not np.any(info), "singular pivot reported"
    res = out.cpu().numpy()
    assert np.all(np.isfinite(res))

    if rank == 0:
        pts = args.steps * M * world
        flops_per_launch_pt = 8.0 * N ** 3
        launches = max(inv_launches, 1)
        pts_per_launch = args.steps * M / launches
        avg_launch_ms = inv_ms / launches
        achieved = flops_per_launch_pt * pts_per_launch / (avg_launch_ms * 1e-3) / 1e12 if inv_ms > 0 else 0.0
        line = {
            "metric": "energy-points/sec (complex128 G(E) solves)",
            "value": pts / dt,
            "unit": "energy-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 (complex128)",
            "data": "synthetic",
            "config": {"workload": f"C2: N_orb={N}, constant Sigma (Gamma=0.2 eV, n_c={NC}/side), "
                                   f"{M} Gauss-Legendre energies per GPU on [-3,3] eV, GrInt",
                       "n_orb": N, "energies_per_gpu": M, "sharding": f"energy-cyclic x{world}",
                       "density_matrix_wall_ms": dt / args.steps * 1e3},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": pmc_traffic_bytes()[0], "traffic_source": pmc_traffic_bytes()[1],
                         "algorithmic_bytes_per_launch": 2.0 * 16.0 * N * N * pts_per_launch,
                         "kernel": "inverse (blocked Gauss-Jordan)",
                         "avg_launch_ms": avg_launch_ms, "launches": launches,
                         "flops_per_point": flops_per_launch_pt,
                         "other_ms_per_step": {"assemble": asm_ms / args.steps, "accumulate": acc_ms / args.steps}},
        }
        if not args.no_cpu and world == 1:           # the CPU leg is an N=1 figure (rank 0 only)
            line["cpu_baseline"] = cpu_baseline(F, S, inds, np.real(E_loc), np.real(w_loc))
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        if args.extra:
            line["extra"] = extra_n500(eng, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extra_n500(eng, args):
    """North-star target case: N_orb=500, 1000 energies, density-matrix build vs CPU."""
    import torch
    from gaunegf_amd.matTools import formSigma
    N, M = 500, 1000
    F, S, inds = make_system(N, 3, 50)
    sig = [formSigma(inds[0], -0.1j, N, S), formSigma(inds[1], -0.1j, N, S)]
    E, w = legendre_grid(M)
    eng.set_system(F, S)
    h = eng.sigma_const(sig)
    eng.gr_int(h, E, w)                 # warm-up with the whole grid: the 12 GB workspace is allocated here
    torch.cuda.synchronize()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.gr_int(h, E, w)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    cpu = cpu_baseline(F, S, inds, E, w, budget_s=8.0)
    return {"n_orb": N, "energies": M, "gpu_wall_s": dt, "gpu_points_per_s": M / dt,
            "gpu_tflops": 8.0 * N ** 3 * M / dt / 1e12, "cpu_points_per_s": cpu["value"],
            "cpu_sample": cpu["sample"], "speedup": (M / dt) / cpu["value"]}


if __name__ == "__main__":
    main()
