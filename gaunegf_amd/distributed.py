"""
Energy-grid sharding across the GPUs of one node.

The reference is single-process / single-device (no pmap, no collectives).  Here the
energy points of one integral are independent units: rank r of W takes the points
m = r, r+W, r+2W, ... (cyclic, which balances the data-dependent iteration counts
of the self-energy fixed points and the stiffer points near the real axis), every
rank accumulates its partial N x N sum on its own GPU, and ONE sum all-reduce of
2 N^2 doubles (RCCL over xGMI; ``backend="nccl"`` is RCCL on ROCm) finishes the
integral.  Per-energy scalars (T(E), DOS(E)) of the shards are put together by ONE
all-gather of ceil(M / W) rows per rank (``all_gather_shards``; on the device with the
nccl backend).  No other collective exists.

``comm_ms_reset()`` / ``comm_ms_total()`` time the collectives themselves (device events
around the RCCL calls on the stream they are issued on, the host clock for CPU backends),
so that a multi-GPU bench line can say how much of a step is communication.  Timing is
OPT-IN: it starts with ``comm_ms_reset()`` and ends with ``comm_ms_stop()``; without it a
collective records nothing (an SCF run issues 10^4 ... 10^5 of them), and events that
have completed are folded into the running total so that the list stays short.

Opt-in: call ``enable()`` after ``torch.distributed.init_process_group``; every rank
must then call GrInt / GrLessInt / calculate_transmission with the same arguments
(SPMD).  Without ``enable()`` the drop-in functions stay purely local.
"""
import numpy as np

_state = {"enabled": False, "group": None, "single_ok": False}
_comm = {"events": [], "host_ms": 0.0, "calls": 0, "on": False}


def comm_ms_reset():
    """Start (or restart) timing the collectives."""
    _comm["events"].clear(); _comm["host_ms"] = 0.0; _comm["calls"] = 0; _comm["on"] = True


def comm_ms_stop():
    """Stop timing; the totals stay readable."""
    _fold(wait=True)
    _comm["on"] = False


def _fold(wait=False):
    """Move completed event pairs into host_ms (all of them with ``wait``)."""
    pending = []
    for a, b in _comm["events"]:
        if wait:
            b.synchronize()
        if wait or b.query():
            _comm["host_ms"] += a.elapsed_time(b)
        else:
            pending.append((a, b))
    _comm["events"][:] = pending


def comm_ms_total():
    """(milliseconds spent in collectives since comm_ms_reset, number of collectives)."""
    _fold(wait=True)
    return _comm["host_ms"], _comm["calls"]


class _timed_collective:
    """Times one collective: device events on the current stream of ``tensor``'s device when it is a device tensor
    (the stream torch.distributed enqueues RCCL work behind), the host clock otherwise."""
    def __init__(self, tensor):
        self.cuda = bool(getattr(tensor, "is_cuda", False))
        self.dev = tensor.device if self.cuda else None

    def __enter__(self):
        import time
        self.on = _comm["on"]
        if not self.on:
            return self
        _comm["calls"] += 1
        if self.cuda:
            import torch
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.dev))
        else:
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        import time
        if not self.on:
            return False
        if self.cuda:
            import torch
            self.e1.record(torch.cuda.current_stream(self.dev))
            _comm["events"].append((self.e0, self.e1))
            if len(_comm["events"]) > 64:
                _fold()
        else:
            _comm["host_ms"] += (time.perf_counter() - self.t0) * 1e3
        return False


def _dist():
    import torch.distributed as dist
    return dist


def enable(group=None, single_rank_ok=False):
    """``single_rank_ok``: take the sharded path (device-resident partial sum, the collectives) in a group of ONE
    rank too -- the RCCL legs then run on a one-GPU box (tests/test_distributed_gpu.py); by default a group of
    one stays on the local path."""
    dist = _dist()
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    _state["enabled"] = True
    _state["group"] = group
    _state["single_ok"] = bool(single_rank_ok)


def disable():
    _state["enabled"] = False
    _state["group"] = None
    _state["single_ok"] = False


def is_active():
    if not _state["enabled"]:
        return False
    dist = _dist()
    return dist.is_initialized() and (dist.get_world_size(_state["group"]) > 1 or _state["single_ok"])


def rank_world():
    if not is_active():
        return 0, 1
    dist = _dist()
    return dist.get_rank(_state["group"]), dist.get_world_size(_state["group"])


def shard_indices(m, rank, world):
    """Cyclic partition: indices of the energies owned by ``rank``."""
    return np.arange(rank, m, world)


def allreduce_sum(arr):
    """Sum a numpy array over all ranks (float64 / complex128), result on every rank."""
    import torch
    dist = _dist()
    a = np.ascontiguousarray(arr)
    is_c = np.iscomplexobj(a)
    flat = a.view(np.float64) if is_c else a.astype(np.float64, copy=False)
    t = torch.from_numpy(flat.copy())
    backend = dist.get_backend(_state["group"])
    if backend == "nccl":
        t = t.cuda()
    with _timed_collective(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_state["group"])
    out = t.cpu().numpy()
    return out.view(np.complex128).reshape(a.shape) if is_c else out.reshape(a.shape)


def allreduce_sum_tensor(t):
    """In-place sum all-reduce of a (device) tensor; the device-resident path of
    bench.py and of large integrals: 2 N^2 doubles, one collective per integral."""
    dist = _dist()
    with _timed_collective(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_state["group"])
    return t


def barrier():
    if is_active():
        _dist().barrier(group=_state["group"])


def sharded_device_sum(engine, run_dev, E, w):
    """Device-resident integral over this rank's cyclic shard of the grid (E, w: complex128 numpy
    arrays, the same on every rank).  ``run_dev(m, E_ptr, w_ptr, out_ptr)`` launches the engine's
    ``*_dev`` entry point on the rank's shard; the partial n x n sum stays in HBM, is all-reduced in place
    -- ONE collective of 2 n^2 doubles (RCCL over xGMI with the nccl backend) -- and is downloaded once.
    With a CPU backend (gloo: tests, rehearsals on one GPU) the partial sum is staged through the host
    for the collective only."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    idx = shard_indices(E.size, rank, world)
    dev = torch.device("cuda", engine.device)

    def to_dev(a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        return torch.view_as_complex(torch.from_numpy(a.view(np.float64).reshape(-1, 2).copy())).to(dev)
    out = torch.zeros((engine.n, engine.n), dtype=torch.complex128, device=dev)
    if idx.size:
        E_t, w_t = to_dev(E[idx]), to_dev(w[idx])
        # the zero fill and the uploads ran on torch's stream, the engine accumulates on its own: order them
        torch.cuda.current_stream(dev).synchronize()
        run_dev(int(idx.size), E_t.data_ptr(), w_t.data_ptr(), out.data_ptr())
    # (an empty shard -- fewer points than ranks -- still takes part in the collective, with zeros)
    engine.sync()                                   # the engine's stream -> visible to the collective's stream
    if idx.size:
        engine.warn_if_singular_dev(int(idx.size), "sharded integral", grid_index=idx)
    flat = torch.view_as_real(out)
    if dist.get_backend(_state["group"]) == "nccl":
        with _timed_collective(flat):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_state["group"])
        return out.cpu().numpy()
    host = flat.cpu()
    with _timed_collective(host):
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=_state["group"])
    return torch.view_as_complex(host).numpy().copy()


def sharded_device_seg_sums(engine, run_dev, segs):
    """Several integrals of one system as ONE pass and ONE collective: ``segs`` = [(E, w), ...] (complex128, the same on
    every rank).  The points of all segments together are dealt out cyclically -- point t of the concatenated grid goes to
    rank t mod W, so a level of two nodes and a contour of 486 balance as one grid --, ``run_dev(m, E_ptr, w_ptr, ends,
    out_ptr)`` launches the engine's ``*_seg_dev`` entry point on the rank's share (``ends``: index one past each segment
    within the share), the [nseg, n, n] partial sums stay in HBM and are all-reduced in place, then downloaded once.
    Returns the list of full sums."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    dev = torch.device("cuda", engine.device)
    Es, ws, ends, index, off = [], [], [], [], 0
    for E, w in segs:
        E = np.asarray(E).ravel(); w = np.asarray(w).ravel()
        mine = np.arange((rank - off) % world, E.size, world)
        Es.append(E[mine]); ws.append(w[mine]); index.append(off + mine)
        ends.append((ends[-1] if ends else 0) + mine.size)
        off += E.size
    m = int(ends[-1]) if ends else 0

    def to_dev(a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        return torch.view_as_complex(torch.from_numpy(a.view(np.float64).reshape(-1, 2).copy())).to(dev)
    out = torch.zeros((len(segs), engine.n, engine.n), dtype=torch.complex128, device=dev)
    if m:
        E_t, w_t = to_dev(np.concatenate(Es)), to_dev(np.concatenate(ws))
        torch.cuda.current_stream(dev).synchronize()    # (zero fill and uploads: torch's stream; the engine has its own)
        run_dev(m, E_t.data_ptr(), w_t.data_ptr(), np.asarray(ends, dtype=np.int32), out.data_ptr())
    engine.sync()
    if m:
        engine.warn_if_singular_dev(m, "sharded integrals", grid_index=np.concatenate(index))
    flat = torch.view_as_real(out)
    if dist.get_backend(_state["group"]) == "nccl":
        with _timed_collective(flat):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_state["group"])
        host = out.cpu().numpy()
    else:
        h = flat.cpu()
        with _timed_collective(h):
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=_state["group"])
        host = torch.view_as_complex(h).numpy().copy()
    return [host[k] for k in range(len(segs))]


def sharded_sum(partial_fn, m):
    """``partial_fn(idx)`` returns the partial sum over energy indices ``idx``;
    returns the full sum on every rank."""
    if not is_active():
        return partial_fn(slice(None))
    rank, world = rank_world()
    part = partial_fn(shard_indices(m, rank, world))
    return allreduce_sum(part)


def all_gather_shards(part, m, device=None):
    """Per-energy values of this rank's cyclic shard, ``part`` [len(shard), *tail] (float64 numpy), from every
    rank -> the full [m, *tail] array on every rank.  ONE all-gather of ceil(m / W) rows per rank (SURVEY 8e:
    "all-gather of M/8 doubles") instead of a sum over zero-filled length-m vectors; with the nccl backend the
    shard is gathered on the device (RCCL over xGMI) and downloaded once, with a CPU backend (gloo: tests,
    rehearsals on one GPU) on the host."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    part = np.ascontiguousarray(part, dtype=np.float64)
    tail = part.shape[1:]
    cap = (m + world - 1) // world                      # rows of the largest shard
    mine = torch.zeros((cap,) + tuple(tail), dtype=torch.float64)
    if part.shape[0]:
        mine[:part.shape[0]] = torch.from_numpy(part)
    if dist.get_backend(_state["group"]) == "nccl":
        mine = mine.cuda(device) if device is not None else mine.cuda()
    if mine.is_cuda:
        gathered = torch.empty((world, cap) + tuple(tail), dtype=torch.float64, device=mine.device)
        with _timed_collective(mine):
            dist.all_gather_into_tensor(gathered, mine, group=_state["group"])
    else:
        chunks = [torch.empty_like(mine) for _ in range(world)]
        with _timed_collective(mine):
            dist.all_gather(chunks, mine, group=_state["group"])
        gathered = torch.stack(chunks)
    g = gathered.cpu().numpy()
    full = np.empty((m,) + tuple(tail), dtype=np.float64)
    for r in range(world):
        idx = shard_indices(m, r, world)
        full[idx] = g[r, :idx.size]
    return full


def sharded_map(partial_fn, m, tail_shape=()):
    """``partial_fn(idx)`` returns per-energy values [len(idx), *tail_shape]; returns
    the full [m, *tail_shape] array on every rank (one all-gather of the shards)."""
    if not is_active():
        return partial_fn(slice(None))
    rank, world = rank_world()
    idx = shard_indices(m, rank, world)
    part = np.asarray(partial_fn(idx), dtype=np.float64).reshape((idx.size,) + tuple(tail_shape)) if idx.size \
        else np.zeros((0,) + tuple(tail_shape))
    return all_gather_shards(part, m)
