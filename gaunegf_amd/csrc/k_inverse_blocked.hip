// Blocked Gauss-Jordan inversion with partial pivoting, FP64 MFMA trailing updates,
// one workgroup (512 threads, 8 waves) per matrix.   gfx950 / MI355X.
//
// Replaces G = solve(E S - F - Sigma, I)  (gauNEGF/integrate.py:71, utils.py:52-54,
// transport.py:154,163,186) for every energy point of the grid.
//
// The matrix is reduced IN PLACE and rows are NEVER moved ("implicit pivoting"):
// instead of swapping the pivot row into position c, the kernel records
//     pivrow[c] = physical row used as pivot for column c,   colof[r] = c.
// Row operations alone reduce A to a permutation matrix, T A = Pi, and the in-place
// trick stores column r_c of T in the storage of column c, so at the end
//     G[i][j] = W[pivrow[i]][colof[j]]                       (one gather pass).
// Keeping every row at its address makes each 16x16 tile update a pure
// read-modify-write by one wave and keeps the working set of the in-flight batch at
// one matrix per workgroup (256 x 640 KB = 164 MB at n = 200: resident in the 256 MB
// Infinity Cache instead of streaming through HBM every panel step).
//
// Per block column K = [k0, k0+kw), kw <= NB:
//   1. PANEL.  The n x kw panel lives in REGISTERS, one row strip per thread.  kw
//      unblocked Gauss-Jordan column steps with partial pivoting (|re|+|im| as LAPACK
//      izamax; among not-yet-used rows) run on the strips; only the pivot row, the
//      pivot column and the per-wave arg-max partials go through LDS: two barriers
//      per column step, the search for column j+1 is fused into the update of column
//      j, the wave arg-max uses DPP lane moves, and waves that do not hold the pivot
//      row run a select-free update.  Result: the block column of the elementary
//      transform M_K, i.e.  P = [ -A0K AKK^-1 ; AKK^-1 ; -A2K AKK^-1 ]  (physical rows).
//   2. The strips are stored straight into the panel columns of the matrix (P is not
//      kept in LDS: the kernel needs < 8 KB of LDS and <= 128 VGPRs, so TWO workgroups
//      share a CU and one matrix's latency-bound panel phase overlaps the other's
//      MFMA/memory-bound update); the kw pivot rows Q = W[pivrow[k0..], :] are
//      snapshotted to a scratch area.
//   3. TRAILING UPDATE on the matrix cores, in place:
//         W[i][J] = (i pivot row of this panel ? 0 : W[i][J]) + P[i][:] * Q[:][J]
//      Work item = (row tile I, chunk of 4 column tiles): the P fragments of I are
//      fetched once (L2-hot: just written by this CU), then per column tile the Q
//      fragments and the C tile; a 16x16 complex tile = 4 real v_mfma_f64_16x16x4_f64
//      chains per 4-deep k-step (Cr += Pr Qr; Cr += Pi (-Qi); Ci += Pr Qi; Ci += Pi Qr).
// Flops: 8 n^3 per matrix (complex MAC = 8) -- the LU + triangular-inversion optimum.
// Ties in the pivot search are broken by the lower physical row index (LAPACK: lower
// logical index); this only matters for exactly equal |.|_1 values.
//
// Data layout: row-major complex128 (interleaved), ld = n.  A lane fetches one
// complex element (16 B) per MFMA operand; 16 lanes cover 256 contiguous bytes of
// a matrix row, so tile loads/stores are 4 x 256-B row segments per wave instruction.
#include "negf_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_WAVES = 16;

template <int T, int NB, int CPR, int RPT>
struct GjCfg {
    static constexpr int THREADS = T;
    static constexpr int WAVES = T / 64;
    static constexpr int S = NB / CPR;             // complex values per strip
    static constexpr int TPR = T / CPR;            // threads along the row dimension
    static constexpr int ROWS = TPR * RPT;         // row capacity
    static constexpr int WPG = WAVES / CPR;        // waves per column part (owner group size)
    static constexpr int PITCH = NB + 1;           // LDS row pitch of P in complex (odd -> conflict free)
    static constexpr int RS = (WAVES >= 16) ? 4 : 2;   // row splits of a column tile in the update
};

struct RedSlot { double v; int key; int pad; };

// ---- wave-level arg-max of (v, key): larger v wins, ties -> smaller key -------------
// DPP lane moves inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror), then
// the four row results are combined through v_readlane.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ void dpp_step(double& v, int& key)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    const int okey = __builtin_amdgcn_update_dpp(key, key, CTRL, 0xF, 0xF, false);
    const double ov = __hiloint2double(ohi, olo);
    const bool take = (ov > v) | ((ov == v) & (okey < key));
    v = take ? ov : v; key = take ? okey : key;
}

__device__ __forceinline__ void wave_argmax(double& v, int& key)
{
    dpp_step<0xB1>(v, key);      // quad_perm [1,0,3,2]
    dpp_step<0x4E>(v, key);      // quad_perm [2,3,0,1]
    dpp_step<0x141>(v, key);     // row_half_mirror
    dpp_step<0x140>(v, key);     // row_mirror  -> every lane of a row holds the row result
    double bv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                 __builtin_amdgcn_readlane(__double2loint(v), 0));
    int bk = __builtin_amdgcn_readlane(key, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), r * 16);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), r * 16);
        const int k = __builtin_amdgcn_readlane(key, r * 16);
        const double ov = __hiloint2double(hi, lo);
        const bool take = (ov > bv) | ((ov == bv) & (k < bk));
        bv = take ? ov : bv; bk = take ? k : bk;
    }
    v = bv; key = bk;
}

constexpr int KEY_NONE = 0x7fffffff;

// ---- one Gauss-Jordan column step on the register strips, J known at compile time ----
template <int T, int NB, int CPR, int RPT>
struct PanelCtx {
    cplx (&a)[RPT][NB / CPR];
    bool (&avail)[RPT];              // row not used as a pivot yet
    cplx* rowbuf; cplx* colbuf; RedSlot* red; cplx* piv_ip; int* bad_sh;
    int* pivrow; int* colof;
    int n, k0, kw, tid, lane, wave, h, tr, wave_tr0;
};

template <int T, int NB, int CPR, int RPT, int J>
struct PanelSteps {
    static __device__ __forceinline__ void run(PanelCtx<T, NB, CPR, RPT>& x)
    {
        using C = GjCfg<T, NB, CPR, RPT>;
        constexpr int S = C::S, TPR = C::TPR, WPG = C::WPG;
        constexpr int hj = J / S, sj = J % S;
        if (J < x.kw) {                                     // uniform branch
            const int c = x.k0 + J;
            // (1) combine the partials published by the waves that own column J
            const RedSlot* red = x.red + (J & 1) * MAX_WAVES + hj * WPG;
            double wv = red[0].v; int pphys = red[0].key;
#pragma unroll
            for (int w = 1; w < WPG; ++w) {
                const double ov = red[w].v; const int ok = red[w].key;
                const bool take = (ov > wv) | ((ov == wv) & (ok < pphys));
                wv = take ? ov : wv; pphys = take ? ok : pphys;
            }
            if (!(wv > 0.0) && x.tid == 0 && *x.bad_sh == 0) *x.bad_sh = c + 1;   // exactly singular / NaN
            // a column of NaNs yields no candidate: fall back to any still-available row so the
            // bookkeeping stays a permutation (the result is NaN anyway and info is set)
            if (pphys == KEY_NONE) {
                int cand = KEY_NONE;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = x.tr + q * TPR;
                    if (x.h == 0 && r < x.n && x.avail[q]) cand = min(cand, r);
                }
                double dv = 0.0;
                int k2 = cand;
                // min over the workgroup via the same machinery (negated key trick not needed: use LDS)
                __syncthreads();
                if (x.tid == 0) x.red[0].pad = KEY_NONE;
                __syncthreads();
                if (k2 != KEY_NONE) atomicMin(&x.red[0].pad, k2);
                __syncthreads();
                pphys = x.red[0].pad;
                (void)dv;
            }
            // (2) publish the unscaled pivot row, 1/pivot and the pivot column
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = x.tr + q * TPR;
                if (r == pphys) {
#pragma unroll
                    for (int s = 0; s < S; ++s) x.rowbuf[x.h * S + s] = x.a[q][s];
                    if (x.h == hj) {
                        // 1/pivot = conj(pivot) / |pivot|^2 (one division; |pivot| is far from the
                        // overflow range for these matrices)
                        const cplx pv = x.a[q][sj];
                        const double sc = 1.0 / (pv.x * pv.x + pv.y * pv.y);
                        *x.piv_ip = cmake(pv.x * sc, -pv.y * sc);
                    }
                }
                if (x.h == hj) x.colbuf[r] = x.a[q][sj];
            }
            if (x.tid == 0) { x.pivrow[c] = pphys; x.colof[pphys] = c; }
            __syncthreads();
            // (3) rank-1 update of every strip in two half-strips (bounded register use): a batch of
            //     LDS reads of the pivot row part, then register arithmetic
            const cplx ip = *x.piv_ip;
            cplx nfm[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) nfm[q] = cneg(cmul(x.colbuf[x.tr + q * TPR], ip));   // -(f / pivot)
            bool wave_has_piv = false;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int d = pphys - q * TPR - x.wave_tr0;
                wave_has_piv |= (d >= 0 && d < 64);
            }
            constexpr int HS = (S >= 8) ? S / 2 : S;        // half-strip length
#pragma unroll
            for (int s0 = 0; s0 < S; s0 += HS) {
                cplx rb[HS];
#pragma unroll
                for (int s = 0; s < HS; ++s) rb[s] = x.rowbuf[x.h * S + s0 + s];
                if (!wave_has_piv) {
                    // select-free path: row <- row - (f/pivot) * pivot row
#pragma unroll
                    for (int q = 0; q < RPT; ++q)
#pragma unroll
                        for (int s = 0; s < HS; ++s) x.a[q][s0 + s] = cfma(x.a[q][s0 + s], nfm[q], rb[s]);
                } else {
                    // the wave holding the pivot row: that row becomes (pivot row) / pivot
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const bool is_piv = (x.tr + q * TPR) == pphys;
                        const cplx coef = is_piv ? ip : nfm[q];
#pragma unroll
                        for (int s = 0; s < HS; ++s) {
                            const cplx base = is_piv ? cmake(0.0, 0.0) : x.a[q][s0 + s];
                            x.a[q][s0 + s] = cfma(base, coef, rb[s]);
                        }
                    }
                }
            }
            // pivot-column entry: 1/pivot on the pivot row, -(f/pivot) elsewhere
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const bool is_piv = (x.tr + q * TPR) == pphys;
                if (x.h == hj) x.a[q][sj] = is_piv ? ip : nfm[q];
                x.avail[q] = x.avail[q] && !is_piv;
            }
            // (4) pivot search for column J+1 on the freshly updated strips (owner waves only)
            if constexpr (J + 1 < NB) {
                constexpr int hn = (J + 1) / S, sn = (J + 1) % S;
                if (x.h == hn && J + 1 < x.kw) {            // wave-uniform: a wave belongs to one column part
                    double bv = -1.0; int bkey = KEY_NONE;
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int r = x.tr + q * TPR;
                        if (r < x.n && x.avail[q]) {
                            const double v = cabs1(x.a[q][sn]);
                            const bool take = (v > bv) | ((v == bv) & (r < bkey));
                            bv = take ? v : bv; bkey = take ? r : bkey;
                        }
                    }
                    wave_argmax(bv, bkey);
                    RedSlot* rn = x.red + ((J + 1) & 1) * MAX_WAVES;
                    if (x.lane == 0) { rn[x.wave].v = bv; rn[x.wave].key = bkey; }
                }
            }
            __syncthreads();
            if constexpr (J + 1 < NB) PanelSteps<T, NB, CPR, RPT, J + 1>::run(x);
        }
    }
};

template <int T, int NB, int CPR, int RPT>
__global__ __launch_bounds__(T) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info,
    int dbg /* ablation switches, 0 in production: 1 = no pivot steps, 2 = no MFMA, 4 = no tile loads */)
{
    using C = GjCfg<T, NB, CPR, RPT>;
    constexpr int S = C::S, TPR = C::TPR, PITCH = C::PITCH, RS = C::RS;
    constexpr int GJB_THREADS = T, GJB_WAVES = C::WAVES;
    constexpr int KS = NB / 4;                     // MFMA k-steps per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows16 = (n + 15) & ~15;
    cplx* P = reinterpret_cast<cplx*>(smem_raw);                 // [rows16][PITCH]  physical rows
    cplx* rowbuf = P + (size_t)rows16 * PITCH;                   // [NB]   unscaled pivot row
    cplx* colbuf = rowbuf + NB;                                  // [ROWS] pivot column
    int* pivrow = reinterpret_cast<int*>(colbuf + C::ROWS);      // [rows16] physical pivot row of column c
    int* colof = pivrow + rows16;                                // [rows16] column a row was pivot for, or -1
    __shared__ RedSlot red[2][MAX_WAVES];
    __shared__ cplx piv_ip;
    __shared__ int bad_sh;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = tid / TPR;                 // column part of the panel held by this thread (wave-uniform)
    const int tr = tid - h * TPR;            // row slot
    const int wave_tr0 = (tid & ~63) - h * TPR;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;            // the matrix, updated in place
    cplx* X = bufB + (size_t)blockIdx.x * mat_stride;            // Q snapshots, then the result

    if (tid == 0) bad_sh = 0;
    for (int t = tid; t < rows16; t += GJB_THREADS) { colof[t] = -1; pivrow[t] = 0; }

    // rows >= n of P stay zero for the whole kernel (A operand of the edge tiles)
    for (int t = tid; t < (rows16 - n) * PITCH; t += GJB_THREADS) P[(size_t)n * PITCH + t] = cmake(0.0, 0.0);

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = rows16 >> 4;

    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kw = min(NB, n - k0);
        __syncthreads();                 // previous trailing update (stores to W, reads of P/colof) is complete
        // ---------------- panel: global -> register strips.  Every thread fetches its own strip
        // (S independent 16-byte loads issued back to back: one memory latency per panel)
        cplx a[RPT][S];
        bool avail[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            avail[q] = (r < n) && (colof[r < rows16 ? r : 0] < 0);
            const cplx* g = W + (size_t)(r < n ? r : 0) * n + k0 + h * S;
#pragma unroll
            for (int s = 0; s < S; ++s)
                a[q][s] = (r < n && h * S + s < kw) ? g[s] : cmake(0.0, 0.0);
        }
        // ---------------- pivot search for the first column of the panel (waves holding column part 0)
        if (!(dbg & 1) && h == 0) {
            double bv = -1.0; int bkey = KEY_NONE;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tr + q * TPR;
                if (r < n && avail[q]) {
                    const double v = cabs1(a[q][0]);
                    const bool take = (v > bv) | ((v == bv) & (r < bkey));
                    bv = take ? v : bv; bkey = take ? r : bkey;
                }
            }
            wave_argmax(bv, bkey);
            if (lane == 0) { red[0][wave].v = bv; red[0][wave].key = bkey; }
        }
        __syncthreads();
        // ---------------- kw Gauss-Jordan column steps on the register strips
        // (compile-time recursion over the panel column: every strip index is a constant)
        if (!(dbg & 1)) {
            PanelCtx<T, NB, CPR, RPT> ctx{a, avail, rowbuf, colbuf, &red[0][0], &piv_ip, &bad_sh, pivrow, colof,
                                       n, k0, kw, tid, lane, wave, h, tr, wave_tr0};
            PanelSteps<T, NB, CPR, RPT, 0>::run(ctx);
        } else if (tid == 0) {
            for (int j = 0; j < kw; ++j) { pivrow[k0 + j] = k0 + j; colof[k0 + j] = k0 + j; }
        }
        // ---------------- strips -> panel columns of the matrix (P, physical rows) ; pivot rows ->
        // Q snapshot X[k][:].  pivrow[] of this panel was written before the last barrier of the
        // pivot steps, so it is visible here.
        if (dbg & 1) __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            if (r < n) {
                cplx* g = W + (size_t)r * n + k0 + h * S;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (h * S + s < kw) g[s] = a[q][s];
                    P[(size_t)r * PITCH + h * S + s] = a[q][s];
                }
            }
        }
        for (int k = wave; k < kw; k += GJB_WAVES) {
            const cplx* srow = W + (size_t)pivrow[k0 + k] * n;
            cplx* drow = X + (size_t)k * n;
            for (int j = lane; j < n; j += 64) drow[j] = srow[j];
        }
        __syncthreads();                 // Q snapshot visible; nobody reads a pivot row of W after this point
        // ---------------- trailing update (in place)
        // column tiles fully inside the panel are skipped; a tile that only touches it
        // (NB = 8, or the ragged last panel) is computed and its panel columns masked
        const int pt_lo = (k0 + 15) >> 4;                 // first tile fully inside [k0, k0+kw) ...
        const int pt_hi = (k0 + kw) >> 4;                 // ... up to (excluding) this one
        const int n_skip = max(0, pt_hi - pt_lo);
        const int ct = tiles - n_skip;                    // column tiles to process
        // work item = (column tile, one of RS row ranges): Q fragments in registers, P from LDS
        const int rpart = (tiles + RS - 1) / RS;
        for (int item = wave; item < ct * RS; item += GJB_WAVES) {
            const int cx = item / RS, part = item - cx * RS;
            const int tj = (n_skip > 0 && cx >= pt_lo) ? cx + n_skip : cx;
            const int ti0 = part * rpart, ti1 = min(tiles, ti0 + rpart);
            const int col = tj * 16 + fi;
            const bool col_ok = col < n;
            const bool col_store = col_ok && !(col >= k0 && col < k0 + kw);
            cplx qf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 4 + fk;
                qf[ks] = cmake(0.0, 0.0);
                if (k < kw && col_ok && !(dbg & 4)) qf[ks] = X[(size_t)k * n + col];
            }
            // the next C tile is prefetched while the current one runs its MFMAs; rows used as
            // pivots in this panel start from zero
            auto load_c = [&](int ti, cplx (&dst)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + fk + 4 * r;
                    dst[r] = cmake(0.0, 0.0);
                    if (ti < ti1 && i < n && col_ok && !(dbg & 4)) {
                        const int cf = colof[i];
                        if (!(cf >= k0 && cf < k0 + kw)) dst[r] = W[(size_t)i * n + col];
                    }
                }
            };
            cplx c0[4];
            load_c(ti0, c0);
            for (int ti = ti0; ti < ti1; ++ti) {
                d4 accr, acci;
#pragma unroll
                for (int r = 0; r < 4; ++r) { accr[r] = c0[r].x; acci[r] = c0[r].y; }
                load_c(ti + 1, c0);
                const cplx* prow = P + (size_t)(ti * 16 + fi) * PITCH + fk;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks * 4 < kw) {
                        const cplx pa = prow[ks * 4];
                        if (dbg & 2) { accr[0] += pa.x * qf[ks].x; acci[0] += pa.y * qf[ks].y; continue; }
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qf[ks].x, accr, 0, 0, 0);
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, -qf[ks].y, accr, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qf[ks].y, acci, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, qf[ks].x, acci, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + fk + 4 * r;
                    if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r], acci[r]);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) info[blockIdx.x] = bad_sh;
    // ---------------- G[i][j] = W[pivrow[i]][colof[j]] : four rows per wave iteration,
    // up to 16 independent gathers in flight per lane
    for (int i0 = wave * 4; i0 < n; i0 += GJB_WAVES * 4) {
        for (int j0 = 0; j0 < n; j0 += 256) {
            cplx v[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = i0 + rr, j = j0 + jj * 64 + lane;
                    v[rr][jj] = (i < n && j < n) ? W[(size_t)pivrow[i] * n + colof[j]] : cmake(0.0, 0.0);
                }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = i0 + rr, j = j0 + jj * 64 + lane;
                    if (i < n && j < n) X[(size_t)i * n + j] = v[rr][jj];
                }
        }
    }
}

template <int T, int NB, int CPR, int RPT>
size_t gj_smem(int n)
{
    using C = GjCfg<T, NB, CPR, RPT>;
    const size_t rows16 = (size_t)((n + 15) & ~15);
    return rows16 * C::PITCH * sizeof(cplx) + NB * sizeof(cplx) + (size_t)C::ROWS * sizeof(cplx) +
           2 * rows16 * sizeof(int);
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 1024;      // static __shared__ of the kernel is < 1 KB

template <int T, int NB, int CPR, int RPT>
bool gj_fits(int n)
{
    return n <= GjCfg<T, NB, CPR, RPT>::ROWS && gj_smem<T, NB, CPR, RPT>(n) <= LDS_LIMIT;
}

template <int T, int NB, int CPR, int RPT>
void gj_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    auto kern = gj_blocked_kernel<T, NB, CPR, RPT>;
    const size_t smem = gj_smem<T, NB, CPR, RPT>(n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)LDS_LIMIT);
        attr_set = true;
    }
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("NEGF_GJ_DEBUG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(T), smem, st, n, A, B, stride, info, dbg);
}

int gj_variant()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("NEGF_GJ_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

// which configuration serves dimension n: 0 = none.  The Q snapshot needs NB*n <= n*n.
int gj_pick(int n)
{
    if (n < 32) return 0;                               // small matrices: the unblocked kernel
    if (gj_variant() == 1 && gj_fits<512, 32, 2, 1>(n)) return 4;      // 8-wave variant (A/B testing)
    if (gj_fits<1024, 32, 4, 1>(n)) return 1;           // n <= 256, panel 32, 16 waves
    if (gj_fits<1024, 16, 2, 1>(n)) return 2;           // n <= 512, panel 16
    if (gj_fits<1024, 8, 1, 1>(n)) return 3;            // n <= ~900, panel 8
    return 0;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0; }

// In-place reduction of A with B as scratch; the inverses are gathered into B.
// Returns true: the result is in B.
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    switch (gj_pick(n)) {
    case 1: gj_launch<1024, 32, 4, 1>(st, n, nb, A, B, stride, info); break;
    case 2: gj_launch<1024, 16, 2, 1>(st, n, nb, A, B, stride, info); break;
    case 3: gj_launch<1024, 8, 1, 1>(st, n, nb, A, B, stride, info); break;
    case 4: gj_launch<512, 32, 2, 1>(st, n, nb, A, B, stride, info); break;
    default: return false;
    }
    return true;
}
