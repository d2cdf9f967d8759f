#!/usr/bin/env python3
"""
bench.py -- energy-points/sec of the NEGF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: when launched by torch.distributed.run (RANK/WORLD_SIZE in the
environment) every rank runs the worker below; launched plainly, this process starts N workers
through torch.distributed.run as CHILD processes before it has touched the GPU, and passes their
output through.

Headline workload (BASELINE.json configs[2], "C3" -- the largest single-GPU configuration):
synthetic random Hermitian F / S with N_orb = 500 whose two leads are 1-D chains with a
50-orbital unit cell (surfG1D decimation self-energy, eta = 1e-4 as examples/SiNEGF.py:44), 2000
Gauss-Legendre energy points on [-2, 2] eV PER GPU (weak scaling: rank r owns the points r, r+W, ...
of a W*2000 point grid).  One "step" = one GrInt pass over the local shard: for every energy run
the two decimation fixed points to the reference's stopping rule (surfG1D.py:271-288: conv 1e-5,
relaxation 0.1, at most 2000 sweeps), form Sigma = t g t^H, assemble E S - F - Sigma, invert
(complex128, partial pivoting), accumulate w G -- followed, for N > 1, by ONE RCCL sum all-reduce of
the N_orb x N_orb result.  F, S, the lead matrices, the energy grid and the weights are resident in
HBM before the timed region; the result stays in HBM.

--config c4 / c5 run BASELINE's multi-GPU configurations through the drop-in Python API with the product's own
energy sharding (gaunegf_amd.distributed: cyclic shard, ONE sum all-reduce per integral, ONE all-gather of the
per-energy scalars), STRONG scaling (the grid is fixed, the ranks divide it):
  c4: N_orb = 800, Bethe-lattice Sigma (Au.bethe, 2 contacts x 3 atoms x 9 orbitals); one step = GrInt over the
      486-point ANT contour + GrInt over the 256-point real-axis grid (the density-matrix build of SURVEY 8d C4);
  c5: 2 x 1000 spin-block F / S, qV = 0.5 V window at 300 K, 512 Legendre points; one step = GrLessInt(ind=-1)
      + calculate_transmission(spin='u') on that grid.
The default (c3) is the headline and the only configuration with CPU / secondary legs.

The JSON line carries
  roofline     : the kernel that dominates the step (the chain fixed point, > 90 % of it) against the
                 FP64 matrix-core peak, TWO ways, everywhere with the same meaning:
                   achieved / frac_algorithmic -- ALGORITHMIC flops per launch / average launch duration; flops per
                     (energy, contact) = (8 + 24 sweeps + 16) n_c^3 (start inverse, per sweep an inverse and two
                     products, Sigma = t g t^H; 8 flop per complex multiply-add), with the sweep
                     counts the kernel reports; durations from hipEvents recorded by the library on the stream the
                     kernels run on, during the timed steps;
                   mfma_executed / frac -- the flops actually ISSUED to the matrix cores / the same duration / the
                     peak: the kernels form a complex product from three real ones (6 instead of 8 flop per complex
                     multiply-add) on 16-granular tiles (padding counts, the pivot steps' vector work does not).
                     frac <= 1 is asserted; only this one is a matrix-core utilisation.
                 mfma_busy_frac is the PMC counter SQ_VALU_MFMA_BUSY_CYCLES of the committed profile of the same
                 kernel source / (launch duration x 2.4 GHz x 1024 SIMDs).
  cpu_baseline : the numpy oracle (the reference's CPU restatement) timed on this host on a bounded
                 sample of the same energies.
  extra        : the north-star target case (N_orb = 500, constant Sigma, 1000 energies: GPU time and
                 CPU sample) and BASELINE config C2 (N_orb = 200, constant Sigma, 1000 energies) with the
                 roofline of the dense inverse kernels.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense FP64 matrix peak (vector FP64 peak is the same)
PEAK_CLOCK_HZ, N_SIMD = 2.4e9, 1024   # what the 78.6 TF are made of: 2048 flop / 64 cycles / SIMD


def chain_mfma_flops_per_sweep(nc):
    """Flops one sweep of chain1d_rs_kernel ISSUES to the matrix cores for an n_c-orbital lead (k_chain1d_rs.hip): two
    products and the rank-8 trailing updates of the blocked inverse, in the 3M form when the lead spans three tiles or
    more (else four real products), on 16 x 16 tiles; a last tile of <= 4 rows / columns runs as strips on the 4x4x4
    instruction (a quarter of a tile's flops).  A model of the instruction stream (it ignores the half-tile updates
    that compute a whole tile: a few per cent); the PMC counter behind mfma_busy_frac is the measurement."""
    T = -(-nc // 16)
    rem = T >= 2 and nc - 16 * (T - 1) <= 4
    FT = T - 1 if rem else T
    m = 3.0 if T >= 3 else 4.0
    full, strip = 2048.0, 512.0
    ks = -(-nc // 4)
    # one product: every wave but the strip wave runs FT full tiles (+ one strip), the strip wave T strips, per k-step
    waves_full = 3 if rem else 4
    product = ks * m * (waves_full * (FT * full + (strip if rem else 0.0)) + ((T * strip) if rem else 0.0))
    # inverse: panels of 8 columns, each applied to the T x T tiles with two k-steps (one when <= 4 columns remain)
    tiles = FT * FT * full + ((2 * FT + 1) * strip if rem else 0.0)
    inverse = sum((1 if min(8, nc - p0) <= 4 else 2) * m * tiles for p0 in range(0, nc, 8))
    return 2.0 * product + inverse


def roofline_pair(flops_alg, flops_mfma, seconds):
    """The two fractions every roofline object of this file carries (see the module docstring)."""
    tf_a = flops_alg / seconds / 1e12 if seconds > 0 else 0.0
    tf_m = flops_mfma / seconds / 1e12 if seconds > 0 else 0.0
    frac = tf_m / FP64_MFMA_PEAK_TFLOPS
    frac_alg = tf_a / FP64_MFMA_PEAK_TFLOPS
    assert frac <= 1.0, f"executed MFMA fraction {frac} above 1: the flop accounting is wrong"
    # No field named frac* exceeds 1.  A Hermitian product's algorithmic count is the symmetry-exploiting one (k_zgemm.hip);
    # what is left is a product in the 3-real-product form issued above 3/4 of the peak, whose 8-flop-per-multiply-add
    # equivalent lies above the peak by construction: it is then reported as a RATE only (reference_equivalent_tflops),
    # without a fraction.
    assert frac_alg <= 4.0 / 3.0 + 1e-9, f"algorithmic rate {tf_a} TF above 4/3 of the peak: the flop accounting is wrong"
    return {"achieved": tf_a, "reference_equivalent_tflops": tf_a, "frac_algorithmic": frac_alg if frac_alg <= 1.0 else None,
            "mfma_executed": tf_m, "frac": frac}


HBM_PEAK_BYTES_PER_S = 8.0e12    # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s


def hbm_fraction(traffic_bytes, seconds):
    """Achieved HBM-bandwidth fraction of a launch: bytes that crossed the L2 <-> fabric boundary (PMC, per launch) over
    the launch duration, against the 8 TB/s peak.  None without a matching PMC profile."""
    if not traffic_bytes or seconds <= 0:
        return None
    f = traffic_bytes / seconds / HBM_PEAK_BYTES_PER_S
    assert f <= 1.0, f"HBM fraction {f} above 1: the traffic counters or the duration are wrong"
    return f


# ------------------------------------------------------------------ synthetic systems
def random_system(N, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N))
    F = (A + A.T) * (1.0 / np.sqrt(2 * N)) * 2
    B = rng.standard_normal((N, N))
    S = np.eye(N) + 0.1 * (B + B.T) / np.sqrt(2 * N)
    return F, S


def chain_lead(nc, seed, scale_b=0.2):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((nc, nc)); alpha = (a + a.T) * 0.5 * 0.5
    beta = rng.standard_normal((nc, nc)) * scale_b
    s = rng.standard_normal((nc, nc)); Salpha = np.eye(nc) + 0.05 * (s + s.T) * 0.5 / np.sqrt(nc)
    Sbeta = 0.05 * rng.standard_normal((nc, nc)) / np.sqrt(nc)
    return alpha, Salpha, beta, Sbeta


def c3_system(N=500, nc=50, eta=1e-4):
    F, S = random_system(N, 3)
    left = list(range(nc)); right = list(range(N - nc, N))
    aL = chain_lead(nc, 31); aR = chain_lead(nc, 32)
    kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
              aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    return F, S, [left, right], kw


def legendre_grid(M, lo, hi):
    from scipy.special import roots_legendre
    x, w = roots_legendre(M)
    mid = (hi - lo) / 2
    return mid * (np.real(x) + 1) + lo, mid * w


# ------------------------------------------------------------------------------------ CPU legs
def _blas_limits():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits
    except Exception:
        return None


def cpu_baseline(make_call, n_total, label, budget_s, probe_pts=2):
    """Time `make_call(idx)` (an oracle integral over the energies idx) on a bounded sample, in THIS process.

    The BLAS thread count that is fastest for this workload on this host is used (all cores is not
    the fastest for these matrix sizes: oversubscription); `cores` reports that thread count."""
    limits = _blas_limits()
    ncpu = host_cpus()
    cands = sorted({t for t in (1, 4, 16, ncpu) if t <= ncpu}) if limits else [ncpu]
    probe_idx = np.linspace(0, n_total - 1, probe_pts).astype(int)
    best_t, best_per, by_threads = ncpu, None, {}
    for t in cands:
        ctx = limits(limits=t) if limits else None
        try:
            t0 = time.perf_counter(); make_call(probe_idx); per = (time.perf_counter() - t0) / probe_pts
        finally:
            if ctx is not None:
                ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else ctx.unregister()
        by_threads[str(t)] = round(1.0 / per, 3)
        if best_per is None or per < best_per:
            best_t, best_per = t, per
    n = int(max(4, min(n_total, budget_s / max(best_per, 1e-6))))
    idx = np.linspace(0, n_total - 1, n).astype(int)
    ctx = limits(limits=best_t) if limits else None
    try:
        t0 = time.perf_counter(); make_call(idx); dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else ctx.unregister()
    return {"value": n / dt, "unit": "energy-points/s", "cores": int(best_t), "kind": "port",
            "points_per_s_by_blas_threads": by_threads,
            "sample": f"{n} of {n_total} energies (evenly spaced, {100.0 * n / n_total:.1f} % of the grid) of {label}; "
                      f"numpy {np.__version__} oracle loop (reference CPU restatement), best of BLAS threads {cands} "
                      f"= {best_t} (host has {ncpu} usable CPUs), {dt:.1f} s"}


def host_cpus():
    """CPUs this process may use (affinity / container share), not the machine's total."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


# ---- whole-host baseline: a pool of oracle processes over the energies, one BLAS thread each -- the reference's own
# answer to "a Python loop over 50 x 50 matrices cannot use BLAS threads" (gauNEGF/density.py:121-210, chunked Pool).
_POOL = {}


def _pool_init(kind, payload):
    import oracle
    limits = _blas_limits()
    if limits:
        _POOL["limits"] = limits(limits=1)
    if kind == "c3":
        F, S, inds, kw, eta = payload
        _POOL["F"], _POOL["S"] = F, S
        _POOL["g"] = oracle.Chain1DSigma(F, S, inds, kw["taus"], kw["staus"], kw["alphas"], kw["aOverlaps"],
                                         kw["betas"], kw["bOverlaps"], eta=eta)
    else:
        F, S, inds, gamma = payload
        _POOL["F"], _POOL["S"] = F, S
        _POOL["g"] = oracle.ConstSigma(F, S, inds, gamma)


def _pool_chunk(args):
    import oracle
    E, w = args
    return oracle.GrInt(_POOL["F"], _POOL["S"], _POOL["g"], E, w)


def cpu_baseline_pool(kind, payload, E, w, n_sample, label, workers=None):
    """`n_sample` evenly spaced energies of the grid, dealt to `workers` processes (default: the usable CPUs, at
    most NEGF_BENCH_CPU_WORKERS = 16 -- the CPU share of a one-GPU box); wall time from the first task to the
    last result, the pool's start-up (process spawn, imports) excluded."""
    import multiprocessing as mp
    workers = workers or max(1, min(host_cpus(), int(os.environ.get("NEGF_BENCH_CPU_WORKERS", "16"))))
    idx = np.linspace(0, E.size - 1, min(n_sample, E.size)).astype(int)
    chunks = [(E[c], w[c]) for c in np.array_split(idx, min(len(idx), workers * 4)) if len(c)]
    ctx = mp.get_context("spawn")                        # never fork a process that has initialised the GPU
    # one BLAS thread per worker, from the worker's first import on (a pool of 256 processes must not start 256
    # BLAS thread pools of 256 threads each): the children inherit the environment of the moment they are spawned
    keys = ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in keys}
    os.environ.update({k: "1" for k in keys})
    try:
        pool = ctx.Pool(workers, initializer=_pool_init, initargs=(kind, payload))
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    with pool:
        pool.map(_pool_chunk, chunks[:workers])          # warm-up: imports, first BLAS calls
        t0 = time.perf_counter()
        parts = pool.map(_pool_chunk, chunks, chunksize=1)
        dt = time.perf_counter() - t0
    total = sum(parts)
    assert np.all(np.isfinite(total))
    return {"_sum": total, "_idx": idx,
            "value": len(idx) / dt, "unit": "energy-points/s", "cores": int(workers), "kind": "port",
            "sample": f"{len(idx)} of {E.size} energies (evenly spaced, {100.0 * len(idx) / E.size:.1f} % of the grid) of "
                      f"{label}; numpy {np.__version__} oracle loop (reference CPU restatement) in {workers} processes "
                      f"x 1 BLAS thread ({host_cpus()} usable CPUs of {os.cpu_count()} on the host), {dt:.1f} s"}


def kernel_source_id(fname="k_chain1d_rs.hip"):
    """sha256 (16 hex digits) of a kernel source file: what a profile must have been taken on to be quoted."""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "gaunegf_amd", "csrc", fname), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


# the chain kernel of the TIMED C3 steps as rocprofv3 names it: pitch class 51 (n_c = 50), three workgroups per CU, old iterate
# partly in global scratch, the plain instantiation (the round-robin one, "..., true>", serves first evaluations)
CHAIN_KERNEL_TIMED = "chain1d_rs_kernel<51, 3, true, false>"


def pmc_traffic_bytes(kernel_substr, source_file="k_chain1d_rs.hip"):
    """HBM bytes per launch from a committed PMC profile (separate --pmc passes): 2 x FETCH_SIZE (gfx950 reports
    half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, both in KiB.
    Only a profile whose kernel name matches AND that records the sha256 of the kernel source it was taken on
    (key "_kernel_source_sha16", written by scripts/pmc_summarize.py) equal to the tree's is quoted; otherwise
    (None, reason): a stale number is worse than none."""
    want = kernel_source_id(source_file)
    reason = "no PMC profile of this kernel under profiles/"
    for fname in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (fname.endswith(".json") and "pmc" in fname):
            continue
        try:
            data = json.load(open(os.path.join(ROOT, "profiles", fname)))
        except Exception:
            continue
        for name, ctr in data.items():
            if isinstance(ctr, dict) and kernel_substr in name and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                if data.get("_kernel_source_sha16", {}).get(source_file) == want and want is not None:
                    return (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0, os.path.join("profiles", fname)
                reason = f"profiles/{fname} was taken on another version of {source_file}"
    return None, reason


def pmc_counter(kernel_substr, counter, source_file="k_chain1d_rs.hip"):
    """A per-launch counter of the committed PMC profile taken on the tree's version of `source_file`, or None."""
    want = kernel_source_id(source_file)
    for fname in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (fname.endswith(".json") and "pmc" in fname):
            continue
        try:
            data = json.load(open(os.path.join(ROOT, "profiles", fname)))
        except Exception:
            continue
        if want is None or data.get("_kernel_source_sha16", {}).get(source_file) != want:
            continue
        for name, ctr in data.items():
            if isinstance(ctr, dict) and kernel_substr in name and counter in ctr:
                return ctr[counter]
    return None


# ---------------------------------------------------------------------------------- the worker
def _init_ranks():
    """(torch, dist, world, rank, local_rank, rehearsal) with the process group initialised for world > 1."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # NEGF_BENCH_REHEARSAL=1: several ranks share the visible GPUs and talk over gloo -- a dry run
    # of the N > 1 control flow on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("NEGF_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["LOCAL_RANK"] = str(local_rank)      # get_engine() picks the device from it
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return torch, dist, world, rank, local_rank, rehearsal


def worker_c3(args):
    torch, dist, world, rank, local_rank, rehearsal = _init_ranks()

    from gaunegf_amd.engine import Engine
    from gaunegf_amd.surfG1D import surfG

    N, NC, M, ETA = 500, 50, args.energies, 1e-4
    F, S, inds, kw = c3_system(N, NC, ETA)
    # global grid of world*M points, cyclic shard (distributed.shard_indices)
    Eg, wg = legendre_grid(M * world, -2.0, 2.0)
    E_loc = np.ascontiguousarray(Eg[rank::world], dtype=np.complex128)
    w_loc = np.ascontiguousarray(wg[rank::world], dtype=np.complex128)

    eng = Engine(local_rank)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    eng.set_system(F, S)
    # the headline is COLD: every step runs all fixed points.  (The context's g(E) cache -- negf_set_chain_cache --
    # would serve steps 2.. from the surface Green's functions of step 1; that figure is reported separately below.)
    eng.set_chain_cache(0)
    lead = surfG(F, S, inds, **kw)
    h = lead._negf_lower(eng)
    dev = torch.device("cuda", local_rank)
    to_dev = lambda a: torch.view_as_complex(torch.from_numpy(a.view(np.float64).reshape(-1, 2).copy())).to(dev)
    E_dev, w_dev = to_dev(E_loc), to_dev(w_loc)
    out = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    out_real = torch.view_as_real(out)

    comm_events = []

    def step(timed=False):
        eng.gr_int_dev(h, M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
        if world > 1:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
            if ev:
                ev[0].record(stream)
            if rehearsal:
                t = out_real.cpu(); dist.all_reduce(t, op=dist.ReduceOp.SUM); out_real.copy_(t)
            else:
                dist.all_reduce(out_real, op=dist.ReduceOp.SUM)
            if ev:
                ev[1].record(stream); comm_events.append(ev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_ms = None
    if args.warmup > 0:
        # (workspace and scratch of this grid size are allocated by a throw-away provider, so that the figure below is the
        #  evaluation, not hipMalloc)
        lead0 = surfG(F, S, inds, **kw)
        eng.gr_int_dev(lead0._negf_lower(eng), M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
    for i in range(args.warmup):
        if i == 0:
            # the provider's FIRST evaluation of this grid -- what a grid costs the first time it is seen: nothing is known
            # about the sweep counts of its fixed points yet (the launch runs them round robin)
            fence(); t_first = time.perf_counter(); step(); fence(); first_ms = (time.perf_counter() - t_first) * 1e3
        else:
            step()
    fence()
    eng.profile(True)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    prof = {k: eng.profile_read(k) for k in ("chain1d", "inverse", "assemble", "accumulate")}
    inv_flops = eng.profile_read_flops("inverse")
    eng.profile(False)
    comm_ms = sum(a.elapsed_time(b) for a, b in comm_events) / max(len(comm_events), 1) if comm_events else 0.0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res_cold = out.cpu().numpy()
    # warm steps: the same integral with the g(E) cache on -- one step fills it, the next ones only form
    # Sigma = t g t^H; bit-identical result (asserted); NOT the headline
    warm = None
    if world == 1 and not args.no_warm:
        eng.set_chain_cache(512)
        step(); fence()
        eng.profile(True); eng.profile_reset()
        t1 = time.perf_counter()
        for _ in range(max(args.steps, 3)):
            step()
        fence()
        warm_dt = (time.perf_counter() - t1) / max(args.steps, 3)
        hit_ms, hit_n = eng.profile_read("chain1d_hit")
        eng.profile(False)
        assert np.array_equal(out.cpu().numpy(), res_cold), "a cached step must reproduce the cold step bit for bit"
        st = eng.chain_cache_stats()
        warm = {"warm_ms_per_step": warm_dt * 1e3, "warm_points_per_s": M / warm_dt,
                "sigma_from_cached_g_ms": hit_ms / max(hit_n, 1), "cache_bytes": st["bytes"],
                "note": "g(E) cache on (negf_set_chain_cache, default in the product): steps after the first reuse the "
                        "surface Green's functions and only form Sigma = t g t^H; result bit-identical to the cold step"}
        eng.set_chain_cache(0)

    info = eng.last_info_dev(M)
    assert not np.any(info), "singular pivot reported"
    iters, conv = eng.last_iters_dev(h, M, 2)
    res = out.cpu().numpy()
    assert np.all(np.isfinite(res))

    if rank == 0:
        pts = args.steps * M * world
        ch_ms, ch_launches = prof["chain1d"]
        inv_ms, inv_launches = prof["inverse"]
        ch_launches = max(ch_launches, 1); inv_launches = max(inv_launches, 1)
        sweeps_per_step = float(iters.sum())                           # this rank's shard, one pass
        flops_chain_step = float(np.sum(8.0 + 24.0 * iters + 16.0)) * NC ** 3
        # issued to the matrix cores: per sweep the model of the instruction stream; the start inverse and the two
        # products of Sigma = t g t^H are one inverse / two products of the same code
        per_sweep = chain_mfma_flops_per_sweep(NC)
        mfma_chain_step = float(np.sum(iters + 1.0)) * per_sweep
        launches_per_step = ch_launches / args.steps
        avg_chain_ms = ch_ms / ch_launches
        rl = roofline_pair(flops_chain_step / launches_per_step, mfma_chain_step / launches_per_step, avg_chain_ms * 1e-3)
        traffic, traffic_src = pmc_traffic_bytes(CHAIN_KERNEL_TIMED)
        busy = pmc_counter(CHAIN_KERNEL_TIMED, "SQ_VALU_MFMA_BUSY_CYCLES")
        inv_rl = roofline_pair(inv_flops[0], inv_flops[1], inv_ms * 1e-3)
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
            "config": {"workload": f"C3: N_orb={N} + surfG1D decimation self-energy (two 1-D chain leads, n_c={NC}, "
                                   f"eta={ETA:g}, reference stopping rule), {M} Gauss-Legendre energies per GPU on "
                                   f"[-2,2] eV, GrInt",
                       "n_orb": N, "n_c": NC, "energies_per_gpu": M, "sharding": f"energy-cyclic x{world}",
                       "density_matrix_wall_ms": dt / args.steps * 1e3,
                       "first_evaluation_ms": first_ms,
                       "sweeps_per_energy_and_contact_mean": float(iters.mean()),
                       "fixed_points_converged_frac": float(conv.mean())},
            "roofline": {"bound": "mfma", **rl, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac_means": "mfma_executed / peak (matrix-core flops issued, 3M form, tile padding); "
                                       "frac_algorithmic = achieved / peak (8 flop per complex multiply-add, SURVEY 8d)",
                         "mfma_busy_frac": (busy / (avg_chain_ms * 1e-3 * PEAK_CLOCK_HZ * N_SIMD)) if busy else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         # north_star's "achieved HBM-bandwidth fraction": PMC bytes per launch / (launch duration x 8 TB/s)
                         "hbm_frac": hbm_fraction(traffic, avg_chain_ms * 1e-3), "hbm_peak_bytes_per_s": HBM_PEAK_BYTES_PER_S,
                         "kernel": "chain1d_rs_kernel (1-D chain decimation fixed points, energy x contact; timed steps: one workgroup per fixed point, longest first by the counts of the evaluation before; a first evaluation: persistent workgroups, the fixed points round robin in quanta of 100 sweeps)",
                         "avg_launch_ms": avg_chain_ms, "launches": ch_launches,
                         "sweeps_per_launch": sweeps_per_step / launches_per_step,
                         "flops_per_sweep": 24.0 * NC ** 3, "mfma_flops_per_sweep": per_sweep,
                         "algorithmic_flops_per_launch": flops_chain_step / launches_per_step,
                         # compulsory HBM bytes: the six lead matrices of both contacts once per launch (every
                         # workgroup re-reads them from L2) + one n_c x n_c Sigma block written per unit
                         "algorithmic_bytes_per_launch": 16.0 * NC * NC * 2 * (6 + M / launches_per_step),
                         "share_of_step": ch_ms / args.steps / (dt / args.steps * 1e3),
                         "other_ms_per_step": {"inverse": inv_ms / args.steps, "assemble": prof["assemble"][0] / args.steps,
                                               "accumulate": prof["accumulate"][0] / args.steps},
                         "inverse_kernels": inv_rl},
        }
        if warm:
            line["warm"] = warm
        if world > 1:
            line["comm_ms"] = comm_ms        # the sum all-reduce of the N_orb x N_orb result, per step (events on the stream)
        if not args.no_cpu and world == 1:           # the CPU leg is an N=1 figure (rank 0 only)
            import oracle
            ref = oracle.Chain1DSigma(F, S, inds, kw["taus"], kw["staus"], kw["alphas"], kw["aOverlaps"],
                                      kw["betas"], kw["bOverlaps"], eta=ETA)
            Er, wr = np.real(E_loc), np.real(w_loc)
            label = f"the same C3 workload (N_orb={N}, n_c={NC} decimation, GrInt)"
            one = cpu_baseline(lambda idx: oracle.GrInt(F, S, ref, Er[idx], wr[idx]), M, label,
                               budget_s=min(args.cpu_budget, 8.0))
            # the whole-host figure: >= 5 % of the grid over a pool of single-thread oracle processes
            pool = cpu_baseline_pool("c3", (F, S, inds, kw, ETA), Er, wr, max(M // 20, 64), label)
            # the same sample through the device path (not timed): the CPU leg doubles as a parity check of the
            # benchmarked configuration at full size, production stopping rule included
            cpu_sum, sidx = pool.pop("_sum"), pool.pop("_idx")
            dev_sum = eng.gr_int(h, E_loc[sidx], w_loc[sidx])
            pool["parity_rel_fro_device_vs_cpu_on_sample"] = float(np.linalg.norm(dev_sum - cpu_sum) / np.linalg.norm(cpu_sum))
            # (a converged fixed point may stop one sweep earlier or later than the oracle's: Sigma within 10 conv = 1e-4)
            assert pool["parity_rel_fro_device_vs_cpu_on_sample"] < 1e-4, pool["parity_rel_fro_device_vs_cpu_on_sample"]
            line["cpu_baseline"] = pool
            line["cpu_baseline"]["one_process"] = {k: one[k] for k in ("value", "cores", "points_per_s_by_blas_threads", "sample")}
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / pool["value"]
            line["cpu_baseline"]["gpu_over_one_process"] = line["value"] / one["value"]
            # the WHOLE host once: one single-thread oracle process per usable CPU (sched_getaffinity; the 16-process
            # figure above is the CPU share a one-GPU box is meant to use) -- the reference's chunked Pool, density.py:121-210
            ncpu = min(host_cpus(), int(os.environ.get("NEGF_BENCH_CPU_WORKERS_MAX", "256")))
            if ncpu > pool["cores"] and not args.no_whole_host:
                wh = cpu_baseline_pool("c3", (F, S, inds, kw, ETA), Er, wr, min(M, max(2 * ncpu, 128)), label, workers=ncpu)
                wh.pop("_sum"); wh.pop("_idx")
                wh["gpu_over_whole_host"] = line["value"] / wh["value"]
                line["cpu_baseline"]["whole_host"] = wh
        if not args.no_extra and world == 1:
            line["extra"] = {"north_star_N500_x_1000": extra_const(eng, 500, 50, 1000, 3, reps=3, cpu_budget=6.0,
                                                                   no_cpu=args.no_cpu, whole_host=not args.no_whole_host),
                             "C2_N200_x_1000": extra_const(eng, 200, 20, 1000, 2, reps=10, cpu_budget=4.0,
                                                           no_cpu=args.no_cpu),
                             # BASELINE configs[0]-sized: 60 basis functions (the ethane demo), 100 real-axis points
                             "C1_N60_x_100": extra_const(eng, 60, 6, 100, 1, reps=20, cpu_budget=2.0, no_cpu=args.no_cpu)}
    if world > 1 and not args.no_extra:
        # what BASELINE calls multi-GPU: the FIXED C4 / C5 steps through the product's own sharding over these ranks
        # (strong scaling), beside the weak C3 line -- the driver's command only passes --gpus N
        strong = {}
        for cfg, nsteps in (("c4", 5), ("c5", 2)):
            l = run_api(cfg, nsteps, 1, 1, torch, dist, world, rank, local_rank, rehearsal,
                        check_local=os.environ.get("NEGF_BENCH_CHECK_LOCAL", "1" if rehearsal else "0") == "1")
            if rank == 0:
                strong[cfg + "_strong"] = strong_summary(l)
        if rank == 0:
            line.setdefault("extra", {}).update(strong)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _bethe_contacts(N):
    """Geometry of C4's contacts: 2 contacts x 3 Au atoms x 9 orbitals at the two ends of an N-orbital
    device; the remaining orbitals belong to one device 'atom'."""
    coords = np.array([[0, 0, 0.0], [2.88, 0, 0], [1.44, 2.494, 0],
                       [0, 0, 20.0], [2.88, 0, 20.0], [1.44, 2.494, 20.0], [1.44, 0.8, 10.0]])
    orbMap = np.concatenate([np.full(9, a + 1) for a in range(6)] + [np.full(N - 54, 7)])
    typ_one = np.array([0, 1001, 1002, 1003, 2001, 2002, 2003, 2004, 2005])
    orbTyp = np.concatenate([typ_one] * 6 + [np.zeros(N - 54, dtype=int)])
    return coords, orbMap, orbTyp


ONE_GPU_PROFILES = {"c4": "r05_c4.json", "c5": "r05_c5_bench.json"}      # profiles/: the stored one-GPU lines of --config c4 / c5


def one_gpu_ms_per_step(config):
    """ms per step of the committed one-GPU run of --config c4 / c5 (profiles/), None when there is none."""
    try:
        with open(os.path.join(ROOT, "profiles", ONE_GPU_PROFILES[config])) as f:
            return float(json.load(f)["ms_per_step"])
    except (OSError, ValueError, KeyError):
        return None


def worker_api(args):
    """BASELINE configs C4 / C5 through the drop-in API with the product's own energy sharding (strong scaling)."""
    torch, dist, world, rank, local_rank, rehearsal = _init_ranks()
    line = run_api(args.config, args.steps, args.warmup, args.emulate_share, torch, dist, world, rank, local_rank, rehearsal)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_api(config, steps, warmup, emulate_share, torch, dist, world, rank, local_rank, rehearsal, check_local=False):
    """One strong-scaling measurement of C4 / C5 in an initialised process group: the bench line (rank 0; None elsewhere).
    ``check_local``: rank 0 also evaluates the step WITHOUT sharding and the line carries the largest relative deviation
    of the sharded result from it (``sharded_vs_local_rel``)."""
    import types
    args = types.SimpleNamespace(config=config, steps=steps, warmup=warmup, emulate_share=emulate_share)
    from gaunegf_amd import distributed as D
    from gaunegf_amd import density as DN
    from gaunegf_amd.engine import get_engine
    from gaunegf_amd.integrate import GrInt, GrIntSegments, GrLessInt
    if world > 1:
        D.enable()
    eng = get_engine()
    if args.config == "c4":
        from gaunegf_amd.surfGBethe import surfGB
        N = 800
        F, S = random_system(N, 4)
        coords, orbMap, orbTyp = _bethe_contacts(N)
        lat = os.path.join(ROOT, "gaunegf_amd", "data", "Au")
        g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
        Ec, wc = DN.contour_grid(-8.0, 0.0, 486, 0.0)
        Er, wr = DN.real_axis_grid(-1e6, -8.0, 256, 0.0)
        pts_per_step = 742
        workload = ("C4: N_orb=800, Bethe-lattice Sigma (Au.bethe, 2 contacts x 3 atoms x 9 orbitals, eta=1e-6); one step = "
                    "the two GrInt sums of a density step -- 486-point ANT contour and 256-point real-axis grid -- as ONE pass "
                    f"(GrIntSegments), the 742 energies dealt cyclically over {world} GPU(s), one all-reduce of both sums")

        def step():
            return tuple(GrIntSegments(F, S, g, [(Ec, wc), (Er, wr)]))
    else:
        from gaunegf_amd.matTools import formSigma
        from gaunegf_amd.surfGTester import surfGTest
        from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
        N = 1000
        Fa, Sa = random_system(N, 5); Fb, _ = random_system(N, 6)
        Z = np.zeros((N, N))
        F = np.block([[Fa, Z], [Z, Fb]]); S = np.kron(np.eye(2), Sa)
        nc = 30
        left = list(range(nc)); right = list(range(N - nc, N))
        s1 = formSigma(left, -0.1j, N, Sa); s2 = formSigma(right, -0.1j, N, Sa)
        g = surfGTest(F, S, [left + [N + i for i in left], right + [N + i for i in right]], -0.1j)
        sc = SigmaCalculator(s1, s2)
        M = 512
        Eg, wg = DN.bias_window_grid(-0.25, 0.25, M, 300.0)
        Et = np.real(np.asarray(Eg)).copy()
        # (two N = 1000 solves per energy and entry point: block-diagonal spin system)
        pts_per_step = 2 * M
        workload = ("C5: 2 x 1000 spin-block F/S (scf.py:177-180 layout), qV=0.5 V window at 300 K, 512 Legendre points; one "
                    "step = GrLessInt(ind=-1) + calculate_transmission(spin='u') on that grid (each as two N=1000 solves per "
                    f"energy), the grid sharded cyclically over {world} GPU(s), one sum all-reduce / one all-gather each")

        def step():
            return GrLessInt(F, S, g, Eg, wg, -1), calculate_transmission(F, S, sc, Et, spin='u')

    if args.emulate_share > 1:
        assert world == 1, "--emulate-share is a one-GPU figure"
        k = args.emulate_share
        if args.config == "c4":
            Ec, wc, Er, wr = Ec[0::k], wc[0::k], Er[0::k], wr[0::k]
            pts_per_step = Ec.size + Er.size
        else:
            Eg, wg, Et = Eg[0::k], wg[0::k], Et[0::k]
            pts_per_step = 2 * Eg.size

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):                # the first call uploads F / S and builds the providers
        res = step()
    fence()
    eng.profile(True); eng.profile_reset()
    D.comm_ms_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    fams = ("inverse", "zgemm", "bethe", "assemble", "accumulate", "gamma", "trace")
    prof = {k: eng.profile_read(k) for k in fams}
    flops = {k: eng.profile_read_flops(k) for k in ("inverse", "zgemm")}
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(np.all(np.isfinite(np.asarray(r if not isinstance(r, tuple) else r[0]))) for r in res)
    cm_calls = D.comm_ms_total() if world > 1 else (0.0, 0)
    D.comm_ms_stop()
    local_dev = None
    if check_local and world > 1:
        # the same step on rank 0 alone, not sharded: what the collectives have to reproduce
        D.disable()
        if rank == 0:
            loc = step()
            first = lambda r: np.asarray(r if not isinstance(r, tuple) else r[0])
            local_dev = max(float(np.linalg.norm(first(a) - first(b)) / max(np.linalg.norm(first(b)), 1e-300)) for a, b in zip(res, loc))
        D.enable()
        fence()
    line = None
    if rank == 0:
        inv_ms = prof["inverse"][0]; gm_ms = prof["zgemm"][0]
        # flops as the library counted them for rank 0's launches (negf_profile_read_flops): algorithmic, and issued
        # to the matrix cores (3M form, 16-granular tiles, upper block tiles of the Hermitian products)
        inv_rl = roofline_pair(flops["inverse"][0], flops["inverse"][1], inv_ms * 1e-3)
        gm_rl = roofline_pair(flops["zgemm"][0], flops["zgemm"][1], gm_ms * 1e-3)
        dom, dom_rl, dom_ms = ("zgemm_mfma_kernel (dense complex products G Gamma G^H, Gamma_L G Gamma_R G^H)", gm_rl, gm_ms) \
            if gm_ms > inv_ms else ("windowed Gauss-Jordan inverse (gj_window* + gj_colupdate + gj_gather kernels)", inv_rl, inv_ms)
        line = {
            "metric": "energy-points/sec (complex128 G(E) solves)",
            "value": args.steps * pts_per_step / dt, "unit": "energy-points/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 (complex128)",
            "data": "synthetic",
            "config": {"workload": workload, "n_orb": N if args.config == "c4" else 2 * N,
                       "energy_points_per_step": pts_per_step, "sharding": f"energy-cyclic x{world}",
                       "density_matrix_wall_ms": dt / args.steps * 1e3},
            "roofline": {"bound": "mfma", **dom_rl, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                         "frac_means": "mfma_executed / peak (matrix-core flops issued: 3M form, 16-granular tiles, upper "
                                       "block tiles of Hermitian products); frac_algorithmic = achieved / peak (8 n^3 per "
                                       "solve, 8 M N K per product)",
                         "kernel": dom + " on rank 0's shard",
                         "family_ms_per_step": {k: prof[k][0] / args.steps for k in fams},
                         "inverse": inv_rl, "zgemm": gm_rl,
                         "share_of_step": dom_ms / args.steps / (dt / args.steps * 1e3)},
        }
        if args.emulate_share > 1:
            line["config"]["emulated_share"] = (f"rank 0's share of an {args.emulate_share}-way energy-cyclic sharding run on "
                                                "one GPU (no collective): the per-GPU batch sizes of the multi-GPU configuration")
        if world > 1:
            cm, calls = cm_calls
            line["comm_ms"] = cm / args.steps          # the all-reduces / all-gathers of one step (events around the collectives)
            line["collectives_per_step"] = calls / args.steps
            one = one_gpu_ms_per_step(args.config)
            line["one_gpu_ms_per_step"] = one          # profiles/: the committed one-GPU run of the same step
            line["speedup_vs_one_gpu"] = one / line["ms_per_step"] if one else None
            if local_dev is not None:
                line["sharded_vs_local_rel"] = local_dev
    if world > 1:
        dist.barrier()
        D.disable()
    return line


def strong_summary(line):
    """What the headline's ``extra.c4_strong`` / ``extra.c5_strong`` keep of a run_api line."""
    rl = line["roofline"]
    return {"workload": line["config"]["workload"], "scaling": "strong", "n_gpus": line["n_gpus"], "steps": line["steps"],
            "ms_per_step": line["ms_per_step"], "value": line["value"], "unit": line["unit"],
            "comm_ms": line.get("comm_ms"), "collectives_per_step": line.get("collectives_per_step"),
            "one_gpu_ms_per_step": line.get("one_gpu_ms_per_step"), "speedup_vs_one_gpu": line.get("speedup_vs_one_gpu"),
            "sharded_vs_local_rel": line.get("sharded_vs_local_rel"),
            "inverse": rl["inverse"], "zgemm": rl["zgemm"], "family_ms_per_step": rl["family_ms_per_step"]}




# ------------------------------------------------------------------------- the SCF call pattern
SCF_FAMILIES = ("inverse", "assemble", "accumulate", "zgemm", "gamma", "trace", "chain1d", "chain1d_hit", "bethe", "small")


def _scf_system(name):
    """(label, F, S, g_dev, make_ref, ne, Eminf): the systems of --config scf.  make_ref() -> the oracle's provider of
    the same contacts (None where a CPU replay of the step would take minutes)."""
    import oracle  # noqa: F401  (only used by make_ref, after the timed region)
    if name in ("n60", "n200"):
        from gaunegf_amd.surfGTester import surfGTest
        N, nc = (60, 6) if name == "n60" else (200, 20)
        F, S = random_system(N, 60 if name == "n60" else 2)
        inds = [list(range(nc)), list(range(N - nc, N))]
        g = surfGTest(F, S, inds, -0.1j)
        label = (f"N_orb={N}, energy-independent Gamma=0.2 eV contacts on {nc} orbitals each "
                 f"({'C1-sized: the ethane demo has 60 basis functions' if N == 60 else 'C2-sized'})")
        return label, F, S, g, (lambda: __import__("oracle").ConstSigma(F, S, inds, -0.1j)), N, -60.0
    if name == "n800":
        from gaunegf_amd.surfGBethe import surfGB
        N = 800
        F, S = random_system(N, 4)
        coords, orbMap, orbTyp = _bethe_contacts(N)
        lat = os.path.join(ROOT, "gaunegf_amd", "data", "Au")
        g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
        return "N_orb=800 + Bethe-lattice Sigma (Au.bethe, 2 contacts x 3 atoms x 9 orbitals): C4-sized", F, S, g, None, N, -60.0
    if name == "chain":
        from gaunegf_amd.surfG1D import surfG
        N, nc = 120, 20
        F, S = random_system(N, 7)
        aL = chain_lead(nc, 71); aR = chain_lead(nc, 72)
        inds = [list(range(nc)), list(range(N - nc, N))]
        kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
                  aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=1e-3)
        g = surfG(F, S, inds, **kw)
        return ("N_orb=120 + two 1-D chain leads (n_c=20, eta=1e-3, reference stopping rule): the g(E) cache's case", F, S, g,
                None, N, -60.0)
    raise SystemExit(f"unknown scf system {name}")


def worker_scf(args):
    """One NEGFE.FockToP step (scfE.py:301-462) per system: adaptive real-axis integral below Emin, a Muller Fermi search
    whose every probe is an adaptive ANT contour integral (levels 2 ... 486 points, density.py:211-273, 750-816), and the
    adaptive bias-window integral -- the call pattern the hot path is used in: ~10^2 integrals of 2 ... 324 points.
    Reports wall time per step, integrals and energy points per step, and the split kernel time (the library's
    hipEvents around its kernel families) / everything else (host code, launch gaps, transfers)."""
    import contextlib
    import io
    import torch
    from gaunegf_amd import density as DN
    from gaunegf_amd.engine import get_engine
    from gaunegf_amd.scfE import NEGFE
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    eng = get_engine()
    names = [x for x in args.scf_systems.split(",") if x]
    out_sys = []
    # host numpy of the step (eigenvalues of S^-1 F, Lowdin occupations: the reference's own host work, scfE.py:460-468,
    # density.py:822): BLAS threads limited to this process's CPU share -- the box reports 256 CPUs to OpenBLAS but
    # schedules one GPU's share of them, and 256 BLAS threads on that turn a 200 x 200 solve into 90 ms
    limits = _blas_limits()
    blas_ctx = limits(limits=max(1, min(host_cpus(), int(os.environ.get("NEGF_BENCH_CPU_WORKERS", "16"))))) if limits else None
    # the interpreter holds ~10^6 objects once torch is imported; a full garbage collection over them costs tens of
    # milliseconds and lands in whatever small step happens to allocate at that moment: move what exists now into the
    # permanent generation (the collector stays on for everything allocated from here)
    import gc
    gc.collect(); gc.freeze()
    for name in names:
        label, F, S, g, make_ref, ne, Eminf = _scf_system(name)
        qV, T, tol = 0.1, 300.0, 1e-4

        fixed = {"mu": None}

        def new_step(gobj):
            n = NEGFE(F, S, gobj, ne=ne, spin='r', T=T, Eminf=Eminf)
            n.setIntegralLimits(tol=tol, Emin=None)            # adaptive integrals everywhere; Emin from the DOS
            if fixed["mu"] is None:
                n.setVoltage(qV, fermiMethod='muller')         # Fermi level searched (updFermi), bias window open
            else:
                n.setVoltage(qV, fermi=fixed["mu"])            # Fermi level given: the cycles of an SCF run at fixed mu
            return n
        sink = io.StringIO()
        variants = [("", {})]
        if name == "chain":
            # with a Fermi search every probe moves the contour (and, for leads defined by their own cell, the lead
            # itself: surfG1D.py:331-342), so the g(E) cache has nothing to reuse; at a FIXED Fermi level -- the cycles
            # of an SCF run that only update F -- every grid of a cycle repeats
            variants = [("fermi_search/cache_off", {"cache": 0}), ("fermi_search/cache_on", {"cache": 512}),
                        ("fixed_fermi/cache_off", {"cache": 0, "mu": 0.05}), ("fixed_fermi/cache_on", {"cache": 512, "mu": 0.05})]
        for vname, opt in variants:
            fixed["mu"] = opt.get("mu")
            if "cache" in opt:
                eng.set_chain_cache(0); eng.set_chain_cache(opt["cache"])
            with contextlib.redirect_stdout(sink):
                for _ in range(max(args.warmup, 1)):
                    new_step(g).FockToP()
                steps = [new_step(g) for _ in range(args.steps)]
                torch.cuda.synchronize()
                c0 = dict(eng.counters)
                eng.profile(True); eng.profile_reset()
                t0 = time.perf_counter()
                for n in steps:
                    n.FockToP()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
                fam = {k: eng.profile_read(k) for k in SCF_FAMILIES}
                eng.profile(False)
            calls = (eng.counters["calls"] - c0["calls"]) / args.steps
            pts = (eng.counters["points"] - c0["points"]) / args.steps
            kern_ms = sum(v[0] for v in fam.values()) / args.steps
            rec = {"system": name + ("/" + vname if vname else ""), "workload": label, "n_orb": len(F),
                   "wall_ms_per_step": dt * 1e3, "integrals_per_step": calls, "energy_points_per_step": pts,
                   "points_per_s": pts / dt, "kernel_ms_per_step": kern_ms,
                   "host_launch_gap_transfer_ms_per_step": dt * 1e3 - kern_ms,
                   "kernel_share": kern_ms / (dt * 1e3),
                   "profiled_kernel_groups_per_step": sum(v[1] for v in fam.values()) / args.steps,
                   "family_ms_per_step": {k: v[0] / args.steps for k, v in fam.items() if v[1]},
                   "fermi": float(steps[-1].fermi), "electrons": float(np.real(np.trace(steps[-1].P @ S)))}
            if "cache" in opt:
                rec["chain_cache"] = eng.chain_cache_stats()
            # parity: the same step replayed with the numpy oracle serving every integral (bounded: small systems)
            if make_ref is not None and not args.no_cpu:
                import oracle
                saved = (DN.GrInt, DN.GrLessInt, DN._compute_dos_at_energy)
                DN.GrInt, DN.GrLessInt, DN._compute_dos_at_energy = oracle.GrInt, oracle.GrLessInt, oracle.dos_at_energy
                try:
                    with contextlib.redirect_stdout(sink):
                        t1 = time.perf_counter(); ref = new_step(make_ref()); ref.FockToP(); cpu_dt = time.perf_counter() - t1
                finally:
                    DN.GrInt, DN.GrLessInt, DN._compute_dos_at_energy = saved
                rec["parity_rel_fro_P_vs_oracle_replay"] = float(np.linalg.norm(steps[-1].P - ref.P) / np.linalg.norm(ref.P))
                rec["parity_fermi_abs_diff"] = float(abs(steps[-1].fermi - ref.fermi))
                rec["cpu_oracle_replay_ms"] = cpu_dt * 1e3
                assert rec["parity_rel_fro_P_vs_oracle_replay"] < 1e-8, rec
            out_sys.append(rec)
    eng.set_chain_cache(0); eng.set_chain_cache(512)
    head = out_sys[0]
    line = {"metric": "density-matrix wall time, one NEGFE.FockToP step (adaptive contour + real axis + Muller Fermi search + bias window)",
            "value": head["wall_ms_per_step"], "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": head["wall_ms_per_step"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": "SCF call pattern (SURVEY 3.5(4), scfE.py:301-462): " + head["workload"], "systems": out_sys}}
    print(json.dumps(line), flush=True)



def extra_const(eng, N, nc, M, seed, reps, cpu_budget, no_cpu, whole_host=False):
    """Constant-Sigma GrInt (BASELINE C2 / the north-star sentence's N_orb=500 x 1000 case): device-resident
    timing like the headline, roofline of the dense inverse kernels, CPU sample."""
    import torch
    from gaunegf_amd.matTools import formSigma
    F, S = random_system(N, seed)
    inds = [list(range(nc)), list(range(N - nc, N))]
    sig = [formSigma(inds[0], -0.1j, N, S), formSigma(inds[1], -0.1j, N, S)]
    lo, hi = (-3.0, 3.0)
    E, w = legendre_grid(M, lo, hi)
    eng.set_system(F, S)
    h = eng.sigma_const(sig)
    dev = torch.device("cuda", eng.device)
    to_dev = lambda a: torch.view_as_complex(torch.from_numpy(
        np.ascontiguousarray(a, dtype=np.complex128).view(np.float64).reshape(-1, 2).copy())).to(dev)
    E_dev, w_dev = to_dev(E), to_dev(w)
    out = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    for _ in range(2):
        eng.gr_int_dev(h, M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.gr_int_dev(h, M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    inv_ms, inv_l = eng.profile_read("inverse")
    fl = eng.profile_read_flops("inverse")
    small_ms, small_l = eng.profile_read("small")
    eng.profile(False)
    res = {"n_orb": N, "energies": M, "gpu_ms_per_density_matrix": dt * 1e3, "gpu_points_per_s": M / dt}
    if inv_l:
        res.update({"inverse_ms_per_pass": inv_ms / reps, "inverse_roofline": roofline_pair(fl[0], fl[1], inv_ms * 1e-3)})
    if small_l:
        # n <= 96: assemble + inverse + weighted sum in one kernel on the vector units (no matrix cores): latency-bound
        res.update({"single_kernel_ms_per_pass": small_ms / reps,
                    "single_kernel_gflops_algorithmic": 8.0 * N ** 3 * M * reps / (small_ms * 1e-3) / 1e9})
    if not no_cpu:
        import oracle
        g = oracle.ConstSigma(F, S, inds, -0.1j)
        cpu = cpu_baseline(lambda idx: oracle.GrInt(F, S, g, E[idx], w[idx]), M,
                           f"the same N_orb={N} constant-Sigma workload", budget_s=cpu_budget, probe_pts=4)
        res.update({"cpu_points_per_s": cpu["value"], "cpu_cores": cpu["cores"], "cpu_sample": cpu["sample"],
                    "speedup": (M / dt) / cpu["value"]})
        ncpu = min(host_cpus(), int(os.environ.get("NEGF_BENCH_CPU_WORKERS_MAX", "256")))
        if whole_host and ncpu > 1:
            # north_star's target sentence (">= 10x the reference CPU density-matrix build"), against the whole host:
            # one single-thread oracle process per usable CPU over a sample of the grid
            wh = cpu_baseline_pool("const", (F, S, inds, -0.1j), E, w, min(M, max(2 * ncpu, 128)),
                                   f"the same N_orb={N} constant-Sigma workload", workers=ncpu)
            wh.pop("_sum"); wh.pop("_idx")
            res["cpu_whole_host"] = {**wh, "gpu_over_whole_host": (M / dt) / wh["value"]}
    eng.sigma_free(h)
    return res


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("c3", "c4", "c5", "scf"), default="c3",
                    help="BASELINE configuration: c3 = the headline (weak scaling); c4, c5 = the multi-GPU configurations "
                         "(strong scaling); scf = one NEGFE.FockToP step (the call pattern of an SCF run), one GPU")
    ap.add_argument("--scf-systems", default="n60,n200,n800,chain", help="systems of --config scf (comma separated)")
    ap.add_argument("--energies", type=int, default=2000, help="energy points per GPU (c3)")
    ap.add_argument("--emulate-share", type=int, default=1,
                    help="c4 / c5 on one GPU: run only rank 0's share of a K-way energy-cyclic sharding (the small per-GPU batches)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the N_orb=500 x 1000 and C2 secondary lines")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm (g(E) cache on) steps after the timed cold ones: "
                    "profiler runs, whose per-kernel averages should hold the cold launches only")
    ap.add_argument("--no-whole-host", action="store_true", help="skip the CPU pools over ALL usable CPUs (keep the 16-process ones)")
    ap.add_argument("--cpu-budget", type=float, default=18.0, help="seconds of CPU work for the headline baseline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start the ranks as fresh child processes (nothing in this process has
        # touched the GPU: torch is not even imported yet) and hand their exit code back
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.config == "scf":
        if args.gpus != 1:
            raise SystemExit("--config scf is a one-GPU figure")
        return worker_scf(args)
    (worker_c3 if args.config == "c3" else worker_api)(args)


if __name__ == "__main__":
    main()
