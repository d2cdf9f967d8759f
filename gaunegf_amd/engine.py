"""
Engine: thin Python owner of one ``negf_ctx`` (one process, one GPU).

All O(N^3) M work happens in libnegf_hip.so; this class only converts numpy
arrays to the C ABI's layout (C-contiguous complex128) and keeps handles alive.
PyTorch is not needed here; the *_dev methods accept raw device pointers
(``tensor.data_ptr()``) so that a torch-managed buffer can be all-reduced with
RCCL afterwards (distributed.py).
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _lib
from ._lib import NEGF_IND_TOTAL, NEGF_SPIN_BLOCK, NEGF_SPIN_RESTRICTED, check

_engines = {}


def _c128(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.complex128)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


_FP_LIB = None


def fingerprint(a):
    """Checksum of an array's bytes, for "has the caller changed this array since the last call" tests of cached
    device-side providers and spin-block splits: the library's negf_hash_bytes (host code, eight threads on large
    buffers: a per-GPU share of BASELINE C5 checks 10 arrays of 16 ... 64 MB per step -- 9 of its 123 ms with a
    single-threaded xxh3).  (Values are only ever compared with values of this same function in this process.)"""
    global _FP_LIB
    b = np.ascontiguousarray(a)
    if _FP_LIB is None:
        _FP_LIB = _lib.load()
    return int(_FP_LIB.negf_hash_bytes(b.ctypes.data_as(C.c_void_p), C.c_ulonglong(b.nbytes)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ind(ind):
    return NEGF_IND_TOTAL if ind is None else int(ind)


class Engine:
    def __init__(self, device=None):
        self._lib = _lib.load()
        ndev = self._lib.negf_device_count()
        if ndev <= 0:
            raise RuntimeError(
                "gaunegf_amd: no HIP device visible.  The NEGF engine is GPU-only "
                "(hand-written gfx950 kernels); there is no CPU fallback.")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % ndev
        self.device = int(device)
        ctx = C.c_void_p()
        check(self._lib.negf_create(C.byref(ctx), self.device), "negf_create")
        self._ctx = ctx
        self.n = 0
        self._F = None
        self._S = None
        self.last_info = None
        self.last_iters = None
        self.last_converged = None
        # calls into the hot path and energy points they carried, since the object was made (bench.py --config scf)
        self.counters = {"calls": 0, "points": 0}

    # ------------------------------------------------------------- lifetime
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.negf_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(self._lib.negf_set_stream(self._ctx, C.c_void_p(stream_ptr or 0)), "negf_set_stream")

    def set_batch(self, batch):
        check(self._lib.negf_set_batch(self._ctx, int(batch)), "negf_set_batch")

    def get_batch(self):
        return self._lib.negf_get_batch(self._ctx)

    def set_inverse_algo(self, algo):
        check(self._lib.negf_set_inverse_algo(self._ctx, int(algo)), "negf_set_inverse_algo")

    def set_small_algo(self, algo):
        """0: systems of n <= 96 take the single-kernel path (default), 1: the kernel sequence of larger systems."""
        check(self._lib.negf_set_small_algo(self._ctx, int(algo)), "negf_set_small_algo")

    def set_chain_round_robin(self, quantum=-1, slots=0):
        """Round-robin execution of chain launches with more fixed points than resident slots: a fixed point runs
        `quantum` sweeps, then makes room for a waiting one (-1: default, 0: off); `slots` > 0 caps the slots (tests)."""
        check(self._lib.negf_set_chain_round_robin(self._ctx, int(quantum), int(slots)), "negf_set_chain_round_robin")

    def set_gamma_algo(self, algo):
        """0: compact Gamma products where the provider allows (default), 1: dense n x n products."""
        check(self._lib.negf_set_gamma_algo(self._ctx, int(algo)), "negf_set_gamma_algo")

    def sync(self):
        check(self._lib.negf_sync(self._ctx), "negf_sync")

    CHAIN_CACHE_DEFAULT = 512

    def set_chain_cache(self, max_grids=None, max_bytes=None):
        """g(E) cache of the 1-D chain providers: number of evaluated grids kept in HBM (default 512, 0 = off) and
        their total size in bytes (default 8 GB)."""
        if max_grids is not None:
            check(self._lib.negf_set_chain_cache(self._ctx, int(max_grids)), "negf_set_chain_cache")
        if max_bytes is not None:
            check(self._lib.negf_set_chain_cache_bytes(self._ctx, int(max_bytes)), "negf_set_chain_cache_bytes")

    def chain_cache_clear(self):
        check(self._lib.negf_chain_cache_clear(self._ctx), "negf_chain_cache_clear")

    def chain_cache_stats(self):
        """dict(hits, misses, entries, bytes)."""
        v = [C.c_longlong(0) for _ in range(4)]
        check(self._lib.negf_chain_cache_stats(self._ctx, *[C.byref(x) for x in v]), "negf_chain_cache_stats")
        return dict(zip(("hits", "misses", "entries", "bytes"), (int(x.value) for x in v)))

    # --------------------------------------------------------------- system
    def set_system(self, F, S):
        """Make F, S the resident system.  The library keeps the last two systems on the device and recognises them
        bitwise (negf_set_system): calls that repeat a system, or alternate between two -- the spin blocks of a
        blockdiag(alpha, beta) Fock matrix -- upload nothing.  Matrices that reach the library as this engine's own
        private complex copies (``_c128_keyed``) carry a key and are recognised without the comparison."""
        F, kF = self._c128_keyed(F)
        S, kS = self._c128_keyed(S)
        assert F.shape == S.shape, "F and S must have the same shape"
        assert F.ndim == 2 and F.shape[0] == F.shape[1], "F and S must be square matrices"
        n_changed = F.shape[0] != self.n
        key = (kF << 32) | kS if kF and kS else 0
        check(self._lib.negf_set_system_keyed(self._ctx, F.shape[0], _ptr(F), _ptr(S), C.c_ulonglong(key)), "negf_set_system")
        self.n = F.shape[0]
        if n_changed:
            self.generation = getattr(self, "generation", 0) + 1   # provider handles died
        return True

    def _c128_cached(self, a):
        return self._c128_keyed(a)[0]

    def _c128_keyed(self, a):
        """(complex128 C-contiguous form of a system matrix, its serial number or 0).  An SCF step hands the same REAL F and
        S to every one of its ~30 integrals; converting 2 x 5 MB to complex at N = 800 each time is 2 ms per call.  The
        conversions of the last four matrices are kept together with a snapshot of their source: the same array object
        with the same content (compared element by element -- a caller may have changed it in place; an owned read-only
        array cannot change and is not compared) gets its conversion back.  A kept conversion is private to the engine and
        never changes: it carries a serial number (never reused) by which the library recognises a resident system
        without comparing bytes (negf_set_system_keyed).  Arrays handed in as complex128 already are passed through
        (serial 0: the library compares) unless they are frozen."""
        a = np.asarray(a)
        frozen = (not a.flags.writeable) and a.base is None       # an owned read-only array (integrate._split_blocks): cannot change
        ready = a.dtype == np.complex128 and a.flags.c_contiguous
        if ready and not frozen:
            return a, 0                                           # (nothing to convert, nothing known about its future)
        cache = self.__dict__.setdefault("_sys_conv", [])
        for k, (src, snap, conv, serial) in enumerate(cache):
            if src is a and (snap is None or (snap.shape == a.shape and snap.dtype == a.dtype and np.array_equal(a, snap))):
                cache.append(cache.pop(k))
                return conv, serial
        conv = a if ready else _c128(a)
        if 128 * 128 <= a.size and a.nbytes <= (64 << 20):        # (small: the conversion costs less than the comparison; large: not worth the host memory)
            serial = self.__dict__["_sys_serial"] = self.__dict__.get("_sys_serial", 0) + 1
            if serial >= (1 << 32):
                return conv, 0
            cache.append((a, None if frozen else a.copy(), conv, serial))
            del cache[:-4]
            return conv, serial
        return conv, 0

    # ------------------------------------------------------------ providers
    def sigma_const(self, sigmas):
        sig = _c128(np.stack([np.asarray(s) for s in sigmas]))
        assert sig.shape[1:] == (self.n, self.n), "sigma shape must match F"
        h = C.c_int(-1)
        check(self._lib.negf_sigma_const(self._ctx, sig.shape[0], _ptr(sig), C.byref(h)), "negf_sigma_const")
        return h.value

    def sigma_chain1d(self, inds_list, alphas, Salphas, betas, Sbetas, taus, Staus,
                      eta, conv, relFactor, max_iter=2000, force_iters=-1):
        nc = np.array([len(i) for i in inds_list], dtype=np.int32)
        inds = np.ascontiguousarray(np.concatenate([np.asarray(i).ravel() for i in inds_list]), dtype=np.int32)

        def cat(mats):
            out = []
            for k, m in enumerate(mats):
                m = _c128(m)
                if m.shape != (nc[k], nc[k]):
                    raise ValueError(f"contact {k}: expected {nc[k]}x{nc[k]} matrix, got {m.shape}")
                out.append(m.ravel())
            return np.ascontiguousarray(np.concatenate(out))
        a, Sa, b, Sb, t, St = (cat(x) for x in (alphas, Salphas, betas, Sbetas, taus, Staus))
        h = C.c_int(-1)
        check(self._lib.negf_sigma_chain1d(self._ctx, len(nc), _ptr(nc), _ptr(inds), _ptr(a), _ptr(Sa),
                                           _ptr(b), _ptr(Sb), _ptr(t), _ptr(St), float(eta), float(conv),
                                           float(relFactor), int(max_iter), int(force_iters), C.byref(h)),
              "negf_sigma_chain1d")
        return h.value

    def sigma_bethe(self, atom_orbs, atom_nbs, H, Slist, Vlist, xi, eta, conv, mix=0.5,
                    max_iter=1000, force_iters=-1):
        """atom_orbs[c][a] = 9 orbital indices; atom_nbs[c][a] = attached directions;
        H[c] 9x9; Slist[c], Vlist[c] 12x9x9."""
        n_atoms = np.array([len(c) for c in atom_orbs], dtype=np.int32)
        orbs = np.ascontiguousarray(
            np.concatenate([np.asarray(a, dtype=np.int32).ravel() for c in atom_orbs for a in c]), dtype=np.int32)
        n_nb = np.array([len(a) for c in atom_nbs for a in c], dtype=np.int32)
        flat = [int(v) for c in atom_nbs for a in c for v in a]
        nb = np.array(flat if flat else [0], dtype=np.int32)
        Hc = np.ascontiguousarray(np.stack([np.asarray(h, dtype=np.float64) for h in H]))
        Sc = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.float64) for s in Slist]))
        Vc = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in Vlist]))
        assert Hc.shape[1:] == (9, 9) and Sc.shape[1:] == (12, 9, 9) and Vc.shape[1:] == (12, 9, 9)
        xi_c = None if xi is None else _c128(xi, (self.n, self.n))
        h = C.c_int(-1)
        check(self._lib.negf_sigma_bethe(self._ctx, len(n_atoms), _ptr(n_atoms), _ptr(orbs), _ptr(n_nb),
                                         _ptr(nb), _ptr(Hc), _ptr(Sc), _ptr(Vc), _ptr(xi_c), float(eta),
                                         float(conv), float(mix), int(max_iter), int(force_iters),
                                         C.byref(h)), "negf_sigma_bethe")
        return h.value

    def bethe_raw(self, H, Slist, Vlist, eta, conv, E, which, mix=0.5, max_iter=1000, force_iters=-1):
        """surfGBAt.sigmaK (which=1 -> [m,12,9,9]) / surfGBAt.sigma (which=2 -> [m,9,9,9])."""
        E, _ = self._grid(E)
        Hc = np.ascontiguousarray(H, dtype=np.float64)
        Sc = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.float64) for s in Slist]))
        Vc = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in Vlist]))
        assert Hc.shape == (9, 9) and Sc.shape == (12, 9, 9) and Vc.shape == (12, 9, 9)
        nd = 12 if which == 1 else 9
        out = np.zeros((E.size, nd, 9, 9), dtype=np.complex128)
        iters = np.zeros(max(E.size, 1), dtype=np.int32)
        conv_f = np.zeros(max(E.size, 1), dtype=np.int32)
        check(self._lib.negf_bethe_raw(self._ctx, _ptr(Hc), _ptr(Sc), _ptr(Vc), float(eta), float(conv),
                                       float(mix), int(max_iter), int(force_iters), int(which), E.size,
                                       _ptr(E), _ptr(out), _ptr(iters), _ptr(conv_f)), "negf_bethe_raw")
        self.last_iters = iters[:E.size]
        self.last_converged = conv_f[:E.size]
        return out

    def sigma_precomputed(self, sigma_tot, sigma_c=None, gammas=None):
        """sigma_tot [m,n,n]; sigma_c [m,n,n] or [m,k,n,n]; gammas [m,k,n,n] are used
        as coupling matrices directly (mutually exclusive with sigma_c)."""
        st = _c128(sigma_tot)
        m = st.shape[0]
        assert st.shape[1:] == (self.n, self.n)
        ncc, sc = 0, None
        if gammas is not None:
            sc = _c128(gammas)
            assert sc.ndim == 4 and sc.shape[0] == m
            ncc = -sc.shape[1]
        elif sigma_c is not None:
            sc = _c128(sigma_c)
            if sc.ndim == 3:
                sc = sc[:, None]
            assert sc.shape[0] == m
            ncc = sc.shape[1]
            sc = np.ascontiguousarray(sc)
        h = C.c_int(-1)
        check(self._lib.negf_sigma_precomputed(self._ctx, m, _ptr(st), ncc, _ptr(sc), C.byref(h)),
              "negf_sigma_precomputed")
        return h.value

    def sigma_free(self, handle):
        if getattr(self, "_ctx", None):
            self._lib.negf_sigma_free(self._ctx, int(handle))

    # -------------------------------------------------------------- hot path
    def _grid(self, E, w=None):
        E = np.ascontiguousarray(np.asarray(E).ravel(), dtype=np.complex128)
        self.counters["calls"] += 1; self.counters["points"] += E.size
        if w is None:
            return E, None
        w = np.ascontiguousarray(np.asarray(w).ravel(), dtype=np.complex128)
        assert E.size == w.size, "Elist and weights must have the same length"
        return E, w

    def _numerical(self, rc, info, where, grid_index=None):
        self.last_info = info
        if rc == _lib.NEGF_ESINGULAR:
            bad = np.nonzero(info)[0]
            if grid_index is not None:                  # positions in a shard -> indices of the caller's grid
                bad = np.asarray(grid_index)[bad]
            warnings.warn(f"{where}: exactly singular E*S-F-Sigma at energy indices {bad[:8].tolist()}"
                          f"{'...' if bad.size > 8 else ''}", RuntimeWarning)

    def gr_int(self, handle, E, w):
        E, w = self._grid(E, w)
        out = np.zeros((self.n, self.n), dtype=np.complex128)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gr_int(self._ctx, handle, E.size, _ptr(E), _ptr(w), _ptr(out), _ptr(info)),
                   "negf_gr_int")
        self._numerical(rc, info[:E.size], "gr_int")
        return out

    def gr_int_seg(self, handle, segments):
        """[sum_m w_m G(E_m) for (E, w) in segments] from ONE pass over all the energies (negf_gr_int_seg)."""
        Es = [np.asarray(E).ravel() for E, _ in segments]
        E, w = self._grid(np.concatenate(Es) if Es else np.zeros(0), np.concatenate([np.asarray(w).ravel() for _, w in segments]) if Es else np.zeros(0))
        ends = np.ascontiguousarray(np.cumsum([e.size for e in Es]), dtype=np.int32)
        out = np.zeros((len(segments), self.n, self.n), dtype=np.complex128)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gr_int_seg(self._ctx, handle, E.size, _ptr(E), _ptr(w), len(segments), _ptr(ends),
                                             _ptr(out), _ptr(info)), "negf_gr_int_seg")
        self._numerical(rc, info[:E.size], "gr_int_seg")
        return [out[k] for k in range(len(segments))]

    REFINE_MAX_N = 4096                                           # negf_gr_int_refine up to here (n <= 512: one workgroup walks an integral's levels; above: a launch per level); beyond, the running values alone are GBs
    REFINE_MAX_INTEGRALS, REFINE_MAX_LEVELS = 64, 2048

    def gr_int_refine(self, handle, requests, tol):
        """Nested adaptive refinement on the device (negf_gr_int_refine).  ``requests`` = [(E, w, counts, ratios, P_in)] per
        integral: ``E``, ``w`` the NEW nodes of consecutive levels of the nested rule, one after the other (``counts`` nodes
        per level), ``ratios`` the nested-weight ratio of each level (None for the first level of a fresh integration) and
        ``P_in`` the running value of an integration that continues (None otherwise).
        Returns [(P, converged, maxdps)]: the value at the converged level (``converged`` = its index among the levels
        handed in) or after the last level (``converged`` = -1), and the maxDP of every level consumed (NaN for a first level)."""
        Es, ws, ends, rats, nlev, off = [], [], [], [], [], 0
        for E, w, counts, ratios, P_in in requests:
            E = np.asarray(E).ravel(); w = np.asarray(w).ravel()
            assert E.size == w.size == sum(counts) and len(counts) == len(ratios), "levels do not add up to the grid"
            assert all((r is None) == (j == 0 and P_in is None) for j, r in enumerate(ratios)), \
                "only the first level of a fresh integration has no ratio"
            Es.append(E); ws.append(w); nlev.append(len(counts))
            for c, r in zip(counts, ratios):
                off += int(c)
                ends.append(off); rats.append(np.nan if r is None else float(r))
        E, w = self._grid(np.concatenate(Es), np.concatenate(ws))
        nint = len(requests)
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        nlev = np.ascontiguousarray(nlev, dtype=np.int32)
        rats = np.ascontiguousarray(rats, dtype=np.float64)
        P_in = None
        if any(r[4] is not None for r in requests):
            P_in = np.zeros((nint, self.n, self.n), dtype=np.complex128)
            for k, r in enumerate(requests):
                if r[4] is not None:
                    P_in[k] = r[4]
        out = np.empty((nint, self.n, self.n), dtype=np.complex128)
        level = np.zeros(nint, dtype=np.int32)
        maxdp = np.zeros(ends.size, dtype=np.float64)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gr_int_refine(self._ctx, handle, E.size, _ptr(E), _ptr(w), nint, _ptr(nlev), _ptr(ends),
                                                _ptr(rats), C.c_double(tol), _ptr(P_in), _ptr(out), _ptr(level), _ptr(maxdp),
                                                _ptr(info)), "negf_gr_int_refine")
        self._numerical(rc, info[:E.size], "gr_int_refine")
        res, s0 = [], 0
        for k in range(nint):
            res.append((out[k], int(level[k]), maxdp[s0:s0 + nlev[k]].copy()))
            s0 += int(nlev[k])
        return res

    def gless_int_seg(self, handle, ind, segments):
        """[sum_m w_m G Gamma G^H for (E, w) in segments] from ONE pass over all the energies (negf_gless_int_seg)."""
        Es = [np.asarray(E).ravel() for E, _ in segments]
        E, w = self._grid(np.concatenate(Es), np.concatenate([np.asarray(w).ravel() for _, w in segments]))
        ends = np.ascontiguousarray(np.cumsum([e.size for e in Es]), dtype=np.int32)
        out = np.zeros((len(segments), self.n, self.n), dtype=np.complex128)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gless_int_seg(self._ctx, handle, _ind(ind), E.size, _ptr(E), _ptr(w), len(segments),
                                                _ptr(ends), _ptr(out), _ptr(info)), "negf_gless_int_seg")
        self._numerical(rc, info[:E.size], "gless_int_seg")
        return [out[k] for k in range(len(segments))]

    def gless_int(self, handle, ind, E, w):
        E, w = self._grid(E, w)
        out = np.zeros((self.n, self.n), dtype=np.complex128)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gless_int(self._ctx, handle, _ind(ind), E.size, _ptr(E), _ptr(w),
                                            _ptr(out), _ptr(info)), "negf_gless_int")
        self._numerical(rc, info[:E.size], "gless_int")
        return out

    def gr_batch(self, handle, E):
        E, _ = self._grid(E)
        out = np.zeros((E.size, self.n, self.n), dtype=np.complex128)
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_gr_batch(self._ctx, handle, E.size, _ptr(E), _ptr(out), _ptr(info)),
                   "negf_gr_batch")
        self._numerical(rc, info[:E.size], "gr_batch")
        return out

    def transmission(self, handle, contact_L, contact_R, E, spin_block=False):
        E, _ = self._grid(E)
        T = np.zeros(E.size, dtype=np.float64)
        Ts = np.zeros((E.size, 4), dtype=np.float64) if spin_block else None
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        mode = NEGF_SPIN_BLOCK if spin_block else NEGF_SPIN_RESTRICTED
        rc = check(self._lib.negf_transmission(self._ctx, handle, int(contact_L), int(contact_R), mode,
                                               E.size, _ptr(E), _ptr(T), _ptr(Ts), _ptr(info)),
                   "negf_transmission")
        self._numerical(rc, info[:E.size], "transmission")
        return (T, Ts) if spin_block else T

    def dos(self, handle, E, per_site=True):
        E, _ = self._grid(E)
        tot = np.zeros(E.size, dtype=np.float64)
        site = np.zeros((E.size, self.n), dtype=np.float64) if per_site else None
        info = np.zeros(max(E.size, 1), dtype=np.int32)
        rc = check(self._lib.negf_dos(self._ctx, handle, E.size, _ptr(E), _ptr(tot), _ptr(site), _ptr(info)),
                   "negf_dos")
        self._numerical(rc, info[:E.size], "dos")
        return (tot, site) if per_site else tot

    def sigma_eval(self, handle, contact, E, n_contacts=1):
        E, _ = self._grid(E)
        out = np.zeros((E.size, self.n, self.n), dtype=np.complex128)
        iters = np.zeros((max(E.size, 1), max(n_contacts, 1)), dtype=np.int32)
        conv = np.zeros((max(E.size, 1), max(n_contacts, 1)), dtype=np.int32)
        check(self._lib.negf_sigma_eval(self._ctx, handle, _ind(contact), E.size, _ptr(E), _ptr(out),
                                        _ptr(iters), _ptr(conv)), "negf_sigma_eval")
        self.last_iters = iters[:E.size]
        self.last_converged = conv[:E.size]
        return out

    # -------------------------------------------------- device-resident calls
    def gr_int_dev(self, handle, m, E_ptr, w_ptr, out_ptr):
        self.counters["calls"] += 1; self.counters["points"] += int(m)
        check(self._lib.negf_gr_int_dev(self._ctx, handle, int(m), C.c_void_p(E_ptr), C.c_void_p(w_ptr),
                                        C.c_void_p(out_ptr)), "negf_gr_int_dev")

    def gless_int_dev(self, handle, ind, m, E_ptr, w_ptr, out_ptr):
        self.counters["calls"] += 1; self.counters["points"] += int(m)
        check(self._lib.negf_gless_int_dev(self._ctx, handle, _ind(ind), int(m), C.c_void_p(E_ptr),
                                           C.c_void_p(w_ptr), C.c_void_p(out_ptr)), "negf_gless_int_dev")

    def gr_int_seg_dev(self, handle, m, E_ptr, w_ptr, ends, out_ptr):
        """negf_gr_int_seg_dev: ``ends`` (host, int32) = index one past each segment; out [len(ends)][n][n] on the device."""
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        self.counters["calls"] += 1; self.counters["points"] += int(m)
        check(self._lib.negf_gr_int_seg_dev(self._ctx, handle, int(m), C.c_void_p(E_ptr), C.c_void_p(w_ptr), int(ends.size),
                                            _ptr(ends), C.c_void_p(out_ptr)), "negf_gr_int_seg_dev")

    def gless_int_seg_dev(self, handle, ind, m, E_ptr, w_ptr, ends, out_ptr):
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        self.counters["calls"] += 1; self.counters["points"] += int(m)
        check(self._lib.negf_gless_int_seg_dev(self._ctx, handle, _ind(ind), int(m), C.c_void_p(E_ptr), C.c_void_p(w_ptr),
                                               int(ends.size), _ptr(ends), C.c_void_p(out_ptr)), "negf_gless_int_seg_dev")

    def transmission_dev(self, handle, contact_L, contact_R, m, E_ptr, T_ptr, Tspin_ptr=0, spin_block=False):
        self.counters["calls"] += 1; self.counters["points"] += int(m)
        mode = NEGF_SPIN_BLOCK if spin_block else NEGF_SPIN_RESTRICTED
        check(self._lib.negf_transmission_dev(self._ctx, handle, int(contact_L), int(contact_R), mode, int(m),
                                              C.c_void_p(E_ptr), C.c_void_p(T_ptr),
                                              C.c_void_p(Tspin_ptr or 0)), "negf_transmission_dev")

    def last_info_dev(self, m):
        info = np.zeros(max(m, 1), dtype=np.int32)
        check(self._lib.negf_last_info(self._ctx, int(m), _ptr(info)), "negf_last_info")
        return info[:m]

    def warn_if_singular_dev(self, m, where, grid_index=None):
        """The *_dev entry points return no per-energy info: fetch it and warn like the host variants
        (``grid_index``: the grid indices of the m energies when they are a shard of a larger grid)."""
        info = self.last_info_dev(m)
        self._numerical(_lib.NEGF_ESINGULAR if np.any(info) else 0, info, where, grid_index)

    def last_iters_dev(self, handle, m, n_contacts):
        """(sweeps, converged) [m, n_contacts] of the fixed points run by the last call."""
        it = np.zeros((max(m, 1), max(n_contacts, 1)), dtype=np.int32)
        cv = np.zeros((max(m, 1), max(n_contacts, 1)), dtype=np.int32)
        check(self._lib.negf_last_iters(self._ctx, int(handle), int(m), _ptr(it), _ptr(cv)), "negf_last_iters")
        return it[:m], cv[:m]

    # ---------------------------------------------------------- diagnostics
    def profile(self, on=True):
        check(self._lib.negf_profile_enable(self._ctx, 1 if on else 0), "negf_profile_enable")

    def profile_reset(self):
        check(self._lib.negf_profile_reset(self._ctx), "negf_profile_reset")

    def profile_read(self, family):
        ms = C.c_double(0.0)
        n = C.c_int(0)
        check(self._lib.negf_profile_read(self._ctx, family.encode(), C.byref(ms), C.byref(n)),
              "negf_profile_read")
        return ms.value, n.value

    def profile_read_flops(self, family):
        """(algorithmic flops, flops issued to the matrix cores) of the family's launches since profile_reset."""
        a = C.c_double(0.0); m = C.c_double(0.0)
        check(self._lib.negf_profile_read_flops(self._ctx, family.encode(), C.byref(a), C.byref(m)),
              "negf_profile_read_flops")
        return a.value, m.value

    def selftest_mfma(self):
        err = C.c_double(-1.0)
        check(self._lib.negf_selftest_mfma(self._ctx, C.byref(err)), "negf_selftest_mfma")
        return err.value


def get_engine(device=None):
    """Process-wide engine for ``device`` (default: LOCAL_RANK, else 0)."""
    key = device
    if key not in _engines:
        _engines[key] = Engine(device)
    return _engines[key]


def reset_engines():
    for e in _engines.values():
        e.close()
    _engines.clear()
