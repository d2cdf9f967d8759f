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
// read-modify-write by one wave.
//
// Per block column K = [k0, k0+kw), kw <= NB:
//   1. PANEL (waves 0..PW-1).  The n x kw panel lives in REGISTERS, one full row strip
//      (NB complex128) per thread.  kw unblocked Gauss-Jordan column steps with partial
//      pivoting (|re|+|im| as LAPACK izamax; among not-yet-used rows) run on the strips;
//      only the pivot row, the pivot column and the per-wave arg-max partials go through
//      LDS: two barriers per column step, the search for column j+1 is fused into the
//      update of column j, the wave arg-max uses DPP lane moves, and waves that do not
//      hold the pivot row run a select-free update.  The phase is VALU-issue bound
//      (PMC: ~170 vector instructions per wave per step), so it runs on the FEWEST waves
//      that can hold the panel: the per-wave bookkeeping is not replicated 8 or 16 times.
//      Result: the block column of the elementary transform M_K, i.e.
//      P = [ -A0K AKK^-1 ; AKK^-1 ; -A2K AKK^-1 ]  (physical rows).
//   2. P -> LDS, k-major (conflict-free A-operand reads), and -> the panel columns of the
//      matrix; the kw pivot rows Q = W[pivrow[k0..], :] are snapshotted to a scratch area.
//   3. TRAILING UPDATE on the matrix cores (all 8 waves), in place:
//         W[i][J] = (i pivot row of this panel ? 0 : W[i][J]) + P[i][:] * Q[:][J]
//      Work item = (column tile J, half of the row tiles): Q fragments of J in registers,
//      row tiles swept with the next C tile prefetched; a 16x16 complex tile = 4 real
//      v_mfma_f64_16x16x4_f64 chains per 4-deep k-step
//      (Cr += Pr Qr; Cr += Pi (-Qi); Ci += Pr Qi; Ci += Pi Qr).
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

constexpr int GJB_THREADS = 512;
constexpr int GJB_WAVES = GJB_THREADS / 64;
constexpr int PW = 4;                              // panel waves
constexpr int PT = PW * 64;                        // panel threads = rows per pass

template <int NB, int RPT>
struct GjCfg {
    static constexpr int S = NB;                   // complex values per strip (a full panel row)
    static constexpr int ROWS = PT * RPT;          // row capacity
    static constexpr int RS = 2;                   // row splits of a column tile in the update
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
template <int NB, int RPT>
struct PanelCtx {
    cplx (&a)[RPT][NB];
    bool (&avail)[RPT];              // row not used as a pivot yet
    cplx* rowbuf; cplx* colbuf; RedSlot* red; cplx* piv_ip; int* bad_sh;
    int* pivrow; int* colof;
    int n, k0, kw, tid, lane, wave, wave_tr0;
    bool active;                     // this wave holds strips (waves >= PW only take the barriers)
};

template <int NB, int RPT, int J>
struct PanelSteps {
    static __device__ __forceinline__ void run(PanelCtx<NB, RPT>& x)
    {
        constexpr int S = NB;
        if (J < x.kw) {                                     // uniform branch
            const int c = x.k0 + J;
            int pphys = KEY_NONE;
            if (x.active) {
                // (1) combine the partials published by the panel waves
                const RedSlot* red = x.red + (J & 1) * PW;
                double wv = red[0].v; pphys = red[0].key;
#pragma unroll
                for (int w = 1; w < PW; ++w) {
                    const double ov = red[w].v; const int ok = red[w].key;
                    const bool take = (ov > wv) | ((ov == wv) & (ok < pphys));
                    wv = take ? ov : wv; pphys = take ? ok : pphys;
                }
                if (!(wv > 0.0) && x.tid == 0 && *x.bad_sh == 0) *x.bad_sh = c + 1;   // singular / NaN
            }
            // a column of NaNs yields no candidate: fall back to the lowest still-available row so
            // the bookkeeping stays a permutation (the result is NaN anyway and info is set)
            if (__syncthreads_or(x.active && pphys == KEY_NONE)) {
                if (x.tid == 0) x.red[0].pad = KEY_NONE;
                __syncthreads();
                if (x.active) {
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int r = x.tid + q * PT;
                        if (r < x.n && x.avail[q]) atomicMin(&x.red[0].pad, r);
                    }
                }
                __syncthreads();
                pphys = x.red[0].pad;
            }
            if (x.active) {
                // (2) publish the unscaled pivot row, 1/pivot and the pivot column
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = x.tid + q * PT;
                    if (r == pphys) {
#pragma unroll
                        for (int s = 0; s < S; ++s) x.rowbuf[s] = x.a[q][s];
                        // 1/pivot = conj(pivot) / |pivot|^2 (one division; |pivot| is far from the
                        // overflow range for these matrices)
                        const cplx pv = x.a[q][J];
                        const double sc = 1.0 / (pv.x * pv.x + pv.y * pv.y);
                        *x.piv_ip = cmake(pv.x * sc, -pv.y * sc);
                    }
                    x.colbuf[r] = x.a[q][J];
                }
                if (x.tid == 0) { x.pivrow[c] = pphys; x.colof[pphys] = c; }
            }
            __syncthreads();
            if (x.active) {
                // (3) rank-1 update of every strip in two half-strips (bounded register use): a batch
                //     of LDS reads of the pivot row part, then register arithmetic
                const cplx ip = *x.piv_ip;
                cplx nfm[RPT];
#pragma unroll
                for (int q = 0; q < RPT; ++q) nfm[q] = cneg(cmul(x.colbuf[x.tid + q * PT], ip));   // -(f / pivot)
                bool wave_has_piv = false;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int d = pphys - q * PT - x.wave_tr0;
                    wave_has_piv |= (d >= 0 && d < 64);
                }
                constexpr int HS = (S >= 8) ? 8 : S;            // sub-strip length (bounds register use)
#pragma unroll
                for (int s0 = 0; s0 < S; s0 += HS) {
                    cplx rb[HS];
#pragma unroll
                    for (int s = 0; s < HS; ++s) rb[s] = x.rowbuf[s0 + s];
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
                            const bool is_piv = (x.tid + q * PT) == pphys;
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
                    const bool is_piv = (x.tid + q * PT) == pphys;
                    x.a[q][J] = is_piv ? ip : nfm[q];
                    x.avail[q] = x.avail[q] && !is_piv;
                }
                // (4) pivot search for column J+1 on the freshly updated strips
                if constexpr (J + 1 < NB) {
                    if (J + 1 < x.kw) {
                        double bv = -1.0; int bkey = KEY_NONE;
#pragma unroll
                        for (int q = 0; q < RPT; ++q) {
                            const int r = x.tid + q * PT;
                            if (r < x.n && x.avail[q]) {
                                const double v = cabs1(x.a[q][J + 1]);
                                const bool take = (v > bv) | ((v == bv) & (r < bkey));
                                bv = take ? v : bv; bkey = take ? r : bkey;
                            }
                        }
                        wave_argmax(bv, bkey);
                        RedSlot* rn = x.red + ((J + 1) & 1) * PW;
                        if (x.lane == 0) { rn[x.wave].v = bv; rn[x.wave].key = bkey; }
                    }
                }
            }
            __syncthreads();
            if constexpr (J + 1 < NB) PanelSteps<NB, RPT, J + 1>::run(x);
        }
    }
};

template <int NB, int RPT>
__global__ __launch_bounds__(GJB_THREADS) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info,
    int dbg /* ablation switches, 0 in production: 1 = no pivot steps, 2 = no MFMA, 4 = no tile loads */)
{
    using C = GjCfg<NB, RPT>;
    constexpr int S = C::S, RS = C::RS;
    constexpr int KS = NB / 4;                     // MFMA k-steps per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows16 = (n + 15) & ~15;
    cplx* Pt = reinterpret_cast<cplx*>(smem_raw);                // [NB][rows16]  P, k-major, physical rows
    cplx* rowbuf = Pt + (size_t)NB * rows16;                     // [NB]   unscaled pivot row
    cplx* colbuf = rowbuf + NB;                                  // [ROWS] pivot column
    int* pivrow = reinterpret_cast<int*>(colbuf + C::ROWS);      // [rows16] physical pivot row of column c
    int* colof = pivrow + rows16;                                // [rows16] column a row was pivot for, or -1
    __shared__ RedSlot red[2][PW];
    __shared__ cplx piv_ip;
    __shared__ int bad_sh;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool pwave = wave < PW;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;            // the matrix, updated in place
    cplx* X = bufB + (size_t)blockIdx.x * mat_stride;            // Q snapshots, then the result

    if (tid == 0) bad_sh = 0;
    for (int t = tid; t < rows16; t += GJB_THREADS) { colof[t] = -1; pivrow[t] = 0; }
    // rows >= n of P stay zero for the whole kernel (A operand of the edge tiles)
    for (int t = tid; t < NB * (rows16 - n); t += GJB_THREADS) {
        const int k = t / (rows16 - n), r = n + t - k * (rows16 - n);
        Pt[(size_t)k * rows16 + r] = cmake(0.0, 0.0);
    }

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = rows16 >> 4;

    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kw = min(NB, n - k0);
        __syncthreads();                 // previous trailing update (stores to W, reads of Pt/colof) is complete
        // ---------------- panel: global -> register strips.  Every panel thread fetches its own
        // strip(s): NB independent 16-byte loads issued back to back, one memory latency per panel
        cplx a[RPT][S];
        bool avail[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + q * PT;
            const bool row_ok = pwave && r < n;
            avail[q] = row_ok && (colof[r < rows16 ? r : 0] < 0);
            const cplx* g = W + (size_t)(row_ok ? r : 0) * n + k0;
#pragma unroll
            for (int s = 0; s < S; ++s)
                a[q][s] = (row_ok && s < kw) ? g[s] : cmake(0.0, 0.0);
        }
        // ---------------- pivot search for the first column of the panel
        if (!(dbg & 1) && pwave) {
            double bv = -1.0; int bkey = KEY_NONE;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * PT;
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
            PanelCtx<NB, RPT> ctx{a, avail, rowbuf, colbuf, &red[0][0], &piv_ip, &bad_sh, pivrow, colof,
                                  n, k0, kw, tid, lane, wave, tid & ~63, pwave};
            PanelSteps<NB, RPT, 0>::run(ctx);
        } else {
            if (tid == 0) for (int j = 0; j < kw; ++j) { pivrow[k0 + j] = k0 + j; colof[k0 + j] = k0 + j; }
            __syncthreads();
        }
        // ---------------- strips -> panel columns of the matrix and -> Pt (LDS, k-major: lanes of a
        // wave write consecutive rows -> contiguous, conflict-free)
        if (pwave) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + q * PT;
                if (r < n) {
                    cplx* g = W + (size_t)r * n + k0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (s < kw) g[s] = a[q][s];
                        Pt[(size_t)s * rows16 + r] = a[q][s];
                    }
                }
            }
        }
        // ---------------- pivot rows -> Q snapshot X[k][:]  (pivrow[] of this panel was published
        // before the last barrier of the pivot steps)
        for (int k = wave; k < kw; k += GJB_WAVES) {
            const cplx* srow = W + (size_t)pivrow[k0 + k] * n;
            cplx* drow = X + (size_t)k * n;
            for (int j = lane; j < n; j += 64) drow[j] = srow[j];
        }
        __syncthreads();                 // Pt, Q snapshot visible; nobody reads a pivot row of W after this
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
                const cplx* pcol = Pt + (size_t)fk * rows16 + ti * 16 + fi;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks * 4 < kw) {
                        const cplx pa = pcol[(size_t)ks * 4 * rows16];
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

template <int NB, int RPT>
size_t gj_smem(int n)
{
    using C = GjCfg<NB, RPT>;
    const size_t rows16 = (size_t)((n + 15) & ~15);
    return (size_t)NB * rows16 * sizeof(cplx) + NB * sizeof(cplx) + (size_t)C::ROWS * sizeof(cplx) +
           2 * rows16 * sizeof(int);
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 1024;      // static __shared__ of the kernel is < 1 KB

template <int NB, int RPT>
bool gj_fits(int n)
{
    return n <= GjCfg<NB, RPT>::ROWS && gj_smem<NB, RPT>(n) <= LDS_LIMIT;
}

template <int NB, int RPT>
void gj_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    auto kern = gj_blocked_kernel<NB, RPT>;
    const size_t smem = gj_smem<NB, RPT>(n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)LDS_LIMIT);
        attr_set = true;
    }
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("NEGF_GJ_DEBUG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(GJB_THREADS), smem, st, n, A, B, stride, info, dbg);
}

// which configuration serves dimension n: 0 = none.  The Q snapshot needs NB*n <= n*n.
int gj_pick(int n)
{
    if (n < 32) return 0;                               // small matrices: the unblocked kernel
    if (gj_fits<32, 1>(n)) return 1;                    // n <= 256, panel 32
    if (gj_fits<16, 2>(n)) return 2;                    // n <= 512, panel 16
    if (gj_fits<8, 4>(n)) return 3;                     // n <= ~960, panel 8
    return 0;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0; }

// In-place reduction of A with B as scratch; the inverses are gathered into B.
// Returns true: the result is in B.
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    switch (gj_pick(n)) {
    case 1: gj_launch<32, 1>(st, n, nb, A, B, stride, info); break;
    case 2: gj_launch<16, 2>(st, n, nb, A, B, stride, info); break;
    case 3: gj_launch<8, 4>(st, n, nb, A, B, stride, info); break;
    default: return false;
    }
    return true;
}
