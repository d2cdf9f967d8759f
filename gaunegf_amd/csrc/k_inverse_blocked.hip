// Blocked Gauss-Jordan inversion with partial pivoting, FP64 MFMA trailing updates and
// panel LOOK-AHEAD, one workgroup (512 threads, 8 waves) per matrix.   gfx950 / MI355X.
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
// Block column K = [k0, k0+kw), kw <= NB.  768 threads = 12 waves (<= 168 VGPRs), two teams:
//   P-team (waves 0-7): PANEL.  The n x kw panel lives in REGISTERS, one row strip (16
//      complex128) per thread.  kw unblocked Gauss-Jordan column steps with partial
//      pivoting (|re|+|im| as LAPACK izamax; among not-yet-used rows) run on the strips;
//      only the pivot row, the pivot column and the per-wave arg-max partials go through
//      LDS; the search for column j+1 is fused into the update of column j, the wave
//      arg-max uses DPP lane moves, waves that do not hold the pivot row run a
//      select-free update.  The phase is a latency chain (two team barriers per column),
//      so it runs CONCURRENTLY with the update of the previous block column:
//   U-team (waves 8-11, joined by the P-team once its panel is done: the update items come
//      from a shared LDS work queue): TRAILING UPDATE of step k on the matrix cores, in place:
//         W[i][J] = (i pivot row of panel k ? 0 : W[i][J]) + P_k[i][:] * Q_k[:][J]
//      with P_k in LDS (k-major, conflict-free A-operand reads), the Q_k fragments of a
//      column tile in registers, C tiles prefetched; a 16x16 complex tile = 4 real
//      v_mfma_f64_16x16x4_f64 chains per 4-deep k-step
//      (Cr += Pr Qr; Cr += Pi (-Qi); Ci += Pr Qi; Ci += Pi Qr).
//   Look-ahead: at step k the P-team first applies step k to the columns of panel k+1
//      (a few tiles), then factors panel k+1 in registers while the U-team updates all
//      other columns.  The teams meet at a workgroup barrier; the strips of panel k+1
//      then become P_{k+1} in LDS and the pivot rows are snapshotted as Q_{k+1}.
//      Team barriers are LDS counters (gfx950 has one hardware barrier per workgroup).
// Flops: 8 n^3 per matrix (complex MAC = 8) -- the LU + triangular-inversion optimum.
// Ties in the pivot search are broken by the lower physical row index (LAPACK: lower
// logical index); this only matters for exactly equal |.|_1 values.
//
// Data layout: row-major complex128 (interleaved), ld = n.  A lane fetches one
// complex element (16 B) per MFMA operand; 16 lanes cover 256 contiguous bytes of
// a matrix row, so tile loads/stores are 4 x 256-B row segments per wave instruction.
#include "negf_common.h"
#include "wave_utils.h"

namespace {

constexpr int GJB_THREADS = 768;
constexpr int GJB_WAVES = GJB_THREADS / 64;
constexpr int PW = 8;                              // panel-team waves
constexpr int PT = PW * 64;                        // panel threads

template <int NB, int CPR, int RPT>
struct GjCfg {
    static constexpr int S = NB / CPR;             // complex values per strip (1/CPR of a panel row)
    static constexpr int TPR = PT / CPR;           // panel threads along the row dimension
    static constexpr int ROWS = TPR * RPT;         // row capacity
    static constexpr int WPG = PW / CPR;           // waves per column part (owner group of a column)
};

struct RedSlot { double v; int key; int pad; };

// ---- team barrier: an LDS counter (monotonic), release/acquire at workgroup scope ----
__device__ __forceinline__ void team_sync(int* ctr, int& expect, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    expect += PW;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expect)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int KEY_NONE = 0x7fffffff;

// ---- one Gauss-Jordan column step on the register strips (P-team only), J compile-time ----
template <int NB, int CPR, int RPT>
struct PanelCtx {
    cplx (&a)[RPT][NB / CPR];
    bool (&avail)[RPT];              // row not used as a pivot yet
    cplx* rowbuf; cplx* colbuf; RedSlot* red; cplx* piv_ip; int* bad_sh;
    int* pivrow; int* colof; int* team_ctr; int& team_expect;
    int n, k0, kw, tid, lane, wave, h, tr, wave_tr0;
};

template <int NB, int CPR, int RPT, int J>
struct PanelSteps {
    static __device__ __forceinline__ void run(PanelCtx<NB, CPR, RPT>& x)
    {
        using C = GjCfg<NB, CPR, RPT>;
        constexpr int S = C::S, TPR = C::TPR, WPG = C::WPG;
        constexpr int hj = J / S, sj = J % S;
        if (J < x.kw) {                                     // uniform branch
            const int c = x.k0 + J;
            // (1) combine the partials published by the waves that own column J
            const RedSlot* red = x.red + (J & 1) * PW + hj * WPG;
            double wv = red[0].v; int pphys = red[0].key;
#pragma unroll
            for (int w = 1; w < WPG; ++w) {
                const double ov = red[w].v; const int ok = red[w].key;
                const bool take = (ov > wv) | ((ov == wv) & (ok < pphys));
                wv = take ? ov : wv; pphys = take ? ok : pphys;
            }
            if (!(wv > 0.0) && x.tid == 0 && *x.bad_sh == 0) *x.bad_sh = c + 1;   // singular / NaN
            // a column of NaNs yields no candidate (the same for every wave of the team): fall back
            // to the lowest still-available row so the bookkeeping stays a permutation
            if (pphys == KEY_NONE) {
                if (x.tid == 0) x.red[0].pad = KEY_NONE;
                team_sync(x.team_ctr, x.team_expect, x.lane);
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = x.tr + q * TPR;
                    if (x.h == 0 && r < x.n && x.avail[q]) atomicMin(&x.red[0].pad, r);
                }
                team_sync(x.team_ctr, x.team_expect, x.lane);
                pphys = x.red[0].pad;
                team_sync(x.team_ctr, x.team_expect, x.lane);
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
            team_sync(x.team_ctr, x.team_expect, x.lane);
            // (3) rank-1 update of every strip in sub-strips of 8 (bounded register use): a batch
            //     of LDS reads of the pivot row part, then register arithmetic
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
            constexpr int HS = (S >= 8) ? 8 : S;
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
            // (4) pivot search for column J+1 on the freshly updated strips (its owner waves)
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
                    RedSlot* rn = x.red + ((J + 1) & 1) * PW;
                    if (x.lane == 0) { rn[x.wave].v = bv; rn[x.wave].key = bkey; }
                }
            }
            team_sync(x.team_ctr, x.team_expect, x.lane);
            if constexpr (J + 1 < NB) PanelSteps<NB, CPR, RPT, J + 1>::run(x);
        }
    }
};

template <int NB, int CPR, int RPT>
__global__ __launch_bounds__(GJB_THREADS) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info,
    int dbg /* ablation switches, 0 in production: 2 = no MFMA, 4 = no tile loads, 16 = U-team idle,
               32 = no pivot steps */,
    unsigned long long* __restrict__ stamps /* diagnostic build only (NEGF_GJ_STAMPS): wall-clock
               stamps of workgroup 0, [step+1][8]; nullptr in production */)
{
    using C = GjCfg<NB, CPR, RPT>;
    constexpr int S = C::S, TPR = C::TPR;
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
    __shared__ int team_ctr;
    __shared__ int next_item;        // work queue of the update items of the current step

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool pwave = wave < PW;
    const int h = (tid % PT) / TPR;          // column part of the panel held by this thread (wave-uniform)
    const int tr = (tid % PT) - h * TPR;     // row slot
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;            // the matrix, updated in place
    cplx* X = bufB + (size_t)blockIdx.x * mat_stride;            // Q snapshots, then the result
    int team_expect = 0;

    if (tid == 0) { bad_sh = 0; team_ctr = 0; }
    for (int t = tid; t < rows16; t += GJB_THREADS) { colof[t] = -1; pivrow[t] = 0; }
    // rows >= n of P stay zero for the whole kernel (A operand of the edge tiles)
    for (int t = tid; t < NB * (rows16 - n); t += GJB_THREADS) {
        const int k = t / (rows16 - n), r = n + t - k * (rows16 - n);
        Pt[(size_t)k * rows16 + r] = cmake(0.0, 0.0);
    }
    __syncthreads();

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = rows16 >> 4;
    auto stamp = [&](int step, int slot, int w) __attribute__((always_inline)) {
        if (stamps && blockIdx.x == 0 && wave == w && lane == 0)
            stamps[(size_t)(step + 1) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    };

    // ---- panel factorisation of block column [p0, p0+pw) by the P-team: strips in registers,
    // result written to the panel columns of the matrix (nobody else touches them meanwhile)
    auto factor_panel = [&](int p0, int pw) __attribute__((always_inline)) {
        cplx a[RPT][S];
        bool avail[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            const bool row_ok = r < n;
            avail[q] = row_ok && (colof[r < rows16 ? r : 0] < 0);
            const cplx* g = W + (size_t)(row_ok ? r : 0) * n + p0 + h * S;
#pragma unroll
            for (int s = 0; s < S; ++s)
                a[q][s] = (row_ok && h * S + s < pw) ? g[s] : cmake(0.0, 0.0);
        }
        if (h == 0) {                                       // waves owning column 0 of the panel
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
        team_sync(&team_ctr, team_expect, lane);
        if (!(dbg & 32)) {
            PanelCtx<NB, CPR, RPT> ctx{a, avail, rowbuf, colbuf, &red[0][0], &piv_ip, &bad_sh, pivrow, colof,
                                       &team_ctr, team_expect, n, p0, pw, tid, lane, wave, h, tr,
                                       (tid & ~63) - h * TPR};
            PanelSteps<NB, CPR, RPT, 0>::run(ctx);
        } else if (tid == 0) {
            for (int j = 0; j < pw; ++j) { pivrow[p0 + j] = p0 + j; colof[p0 + j] = p0 + j; }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            if (r < n) {
                cplx* g = W + (size_t)r * n + p0 + h * S;
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (h * S + s < pw) g[s] = a[q][s];
            }
        }
    };

    // ---- trailing update of one (column tile, row-tile range) item for the panel [k0, k0+kw):
    // stores only columns with  lo <= col < hi  XOR outside (mode): see callers
    auto update_item = [&](int k0, int kw, int tj, int ti0, int ti1, int st_lo, int st_hi,
                           bool inside) __attribute__((always_inline)) {
        const int col = tj * 16 + fi;
        const bool col_ok = col < n;
        const bool in_rng = col >= st_lo && col < st_hi;
        const bool col_store = col_ok && !(col >= k0 && col < k0 + kw) && (inside ? in_rng : !in_rng);
        cplx qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 4 + fk;
            qf[ks] = cmake(0.0, 0.0);
            if (k < kw && col_ok && !(dbg & 4)) qf[ks] = X[(size_t)k * n + col];
        }
        // the next C tile is prefetched while the current one runs its MFMAs; rows used as
        // pivots in this panel start from zero
        auto load_c = [&](int ti, cplx (&dst)[4]) __attribute__((always_inline)) {
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
    };

    // ---- main loop.  Step s >= 0 applies block column s to the matrix; during step s the P-team
    // factors block column s+1 (look-ahead).  Step -1 only factors block column 0.
    const int npanels = (n + NB - 1) / NB;
    for (int step = -1; step < npanels; ++step) {
        const bool has_cur = step >= 0;
        const int k0 = has_cur ? step * NB : 0;
        const int kw = has_cur ? min(NB, n - k0) : 0;
        const int n0 = (step + 1) * NB;                          // next block column
        const bool has_next = n0 < n;
        const int nw = has_next ? min(NB, n - n0) : 0;
        if (has_cur) {
            // ---------------- P_k (panel columns of W, written by the P-team) -> Pt in LDS, k-major.
            // A thread copies a run of consecutive k of one row: 16-byte global loads of one row
            // segment, LDS writes with consecutive lanes on consecutive rows (conflict-free).
            constexpr int KCH = (NB >= 16) ? 16 : NB;            // k values per thread
            constexpr int CHUNKS = NB / KCH;
            for (int t = tid; t < rows16 * CHUNKS; t += GJB_THREADS) {
                const int r = t % rows16, ch = t / rows16;
                if (r < n) {
                    const cplx* g = W + (size_t)r * n + k0 + ch * KCH;
#pragma unroll
                    for (int s = 0; s < KCH; ++s) {
                        const int k = ch * KCH + s;
                        Pt[(size_t)k * rows16 + r] = (k < kw) ? g[s] : cmake(0.0, 0.0);
                    }
                }
            }
            // ---------------- pivot rows -> Q snapshot X[k][:]  (a wave copies up to 4 rows at a time
            // with all their loads in flight)
            for (int kb = wave * 4; kb < kw; kb += GJB_WAVES * 4) {
                for (int j0 = 0; j0 < n; j0 += 128) {
                    cplx v[4][2];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int k = kb + rr;
                        const cplx* srow = W + (size_t)pivrow[k0 + (k < kw ? k : 0)] * n;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = j0 + jj * 64 + lane;
                            v[rr][jj] = (k < kw && j < n) ? srow[j] : cmake(0.0, 0.0);
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int k = kb + rr;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = j0 + jj * 64 + lane;
                            if (k < kw && j < n) X[(size_t)k * n + j] = v[rr][jj];
                        }
                    }
                }
            }
            if (tid == 0) next_item = 0;
            stamp(step, 0, 0);
            __syncthreads();             // [A] Pt, Q snapshot visible; pivot rows of W are not read again
            stamp(step, 1, 0);
        }
        if (has_cur && has_next) {
            // ---- every wave: apply step s to the columns of block column s+1 first (look-ahead)
            const int t_lo = n0 >> 4, t_hi = (n0 + nw + 15) >> 4;        // column tiles touching [n0, n0+nw)
            const int nq = 6;                                            // row ranges per column tile
            const int rq = (tiles + nq - 1) / nq;
            for (int item = wave; item < (t_hi - t_lo) * nq; item += GJB_WAVES) {
                const int tj = t_lo + item / nq, part = item % nq;
                if (part * rq < tiles)
                    update_item(k0, kw, tj, part * rq, min(tiles, part * rq + rq), n0, n0 + nw, true);
            }
            stamp(step, 2, 0);
            __syncthreads();             // [A2] block column s+1 is up to date
            stamp(step, 3, 0);
        }
        // ---- P-team: factor block column s+1; U-team: update the other columns.  The update items
        // come from a shared queue, so the P-team joins in as soon as its panel is done.
        if (pwave && has_next) factor_panel(n0, nw);
        stamp(step, 4, 0);
        if (has_cur && !((dbg & 16) && has_next)) {
            const int rhalf = (tiles + 1) >> 1;
            while (true) {
                int item = 0;
                if (lane == 0) item = atomicAdd(&next_item, 1);
                item = __builtin_amdgcn_readfirstlane(item);
                if (item >= tiles * 2) break;
                const int tj = item >> 1, part = item & 1;
                const int c_lo = tj * 16, c_hi = min(n, c_lo + 16);
                if (c_lo >= k0 && c_hi <= k0 + kw) continue;                         // inside block column s
                if (has_next && c_lo >= n0 && c_hi <= n0 + nw) continue;             // done in the look-ahead
                update_item(k0, kw, tj, part ? rhalf : 0, part ? tiles : rhalf,
                            has_next ? n0 : 0, has_next ? n0 + nw : 0, false);
            }
        }
        stamp(step, 5, 0);
        stamp(step, 6, PW);              // a U-team wave: end of its update work
        __syncthreads();                 // [B] step complete everywhere; block column s+1 factored
        stamp(step, 7, 0);
    }
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

template <int NB, int CPR, int RPT>
size_t gj_smem(int n)
{
    using C = GjCfg<NB, CPR, RPT>;
    const size_t rows16 = (size_t)((n + 15) & ~15);
    return (size_t)NB * rows16 * sizeof(cplx) + NB * sizeof(cplx) + (size_t)C::ROWS * sizeof(cplx) +
           2 * rows16 * sizeof(int);
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 1024;      // static __shared__ of the kernel is < 1 KB

template <int NB, int CPR, int RPT>
bool gj_fits(int n)
{
    return n <= GjCfg<NB, CPR, RPT>::ROWS && gj_smem<NB, CPR, RPT>(n) <= LDS_LIMIT;
}

template <int NB, int CPR, int RPT>
void gj_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    auto kern = gj_blocked_kernel<NB, CPR, RPT>;
    const size_t smem = gj_smem<NB, CPR, RPT>(n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)LDS_LIMIT);
        attr_set = true;
    }
    static int dbg = -1;
    static unsigned long long* d_stamps = nullptr;
    if (dbg < 0) {
        const char* e = getenv("NEGF_GJ_DEBUG"); dbg = e ? atoi(e) : 0;
        if (getenv("NEGF_GJ_STAMPS")) {
            (void)hipMalloc(&d_stamps, 64 * 8 * sizeof(unsigned long long));
            (void)hipMemset(d_stamps, 0, 64 * 8 * sizeof(unsigned long long));
        }
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(GJB_THREADS), smem, st, n, A, B, stride, info, dbg, d_stamps);
    if (d_stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[64 * 8];
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        const int np = (n + NB - 1) / NB;
        fprintf(stderr, "[gj stamps] 100 MHz ticks relative to step start; cols: A-in A-out A2-in A2-out panel-done upd-done(P) upd-done(U) B-out\n");
        unsigned long long t0 = h[7];
        for (int sidx = 0; sidx <= np; ++sidx) {
            fprintf(stderr, "[gj stamps] step %2d:", sidx - 1);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %8.2f", h[sidx * 8 + k] ? (double)(h[sidx * 8 + k] - t0) / 100.0 : -1.0);
            fprintf(stderr, "  (us since end of step -1)\n");
        }
    }
}

// which configuration serves dimension n: 0 = none.  The Q snapshot needs NB*n <= n*n.
int gj_pick(int n)
{
    if (n < 32) return 0;                               // small matrices: the unblocked kernel
    if (gj_fits<32, 2, 1>(n)) return 1;                 // n <= 256, panel 32, two threads per row
    if (gj_fits<16, 1, 1>(n)) return 2;                 // n <= 512, panel 16
    return 0;
}


// ======================================================================================
// Large matrices (n > 512): the same in-place, implicit-pivot Gauss-Jordan reduction,
// blocked at TWO levels across kernels.  For each window of WIN = 64 columns:
//   gj_window_kernel  (one workgroup per matrix): factors the n x 64 block column in
//       sub-panels of NBI columns with the register-strip pivot steps above and applies
//       every sub-panel transform to the 64 window columns only (MFMA, operands L2-hot);
//       afterwards the window holds the block column P' of the combined transform and the
//       64 pivot rows are snapshotted as Q (their other columns are still untouched).
//   gj_bigupdate_kernel (a 64 x 64 output block per workgroup, all other columns):
//       W[i][J] = (i pivot row of this window ? 0 : W[i][J]) + P'[i][:] * Q[:][J]
//       an LDS-tiled complex GEMM on the FP64 matrix cores with K = 64: 16 flop per byte of
//       matrix traffic, i.e. compute-bound for n >~ 500.
// pivrow / colof live in global memory between launches.  gj_gather_kernel forms
// G[i][j] = W[pivrow[i]][colof[j]].
// ======================================================================================
constexpr int WIN = 64;

template <int NBI, int RPT>
__global__ __launch_bounds__(PT) void gj_window_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride,
    int* __restrict__ piv_all /* [nb][2][n]: pivrow, colof */, int* __restrict__ info, int c0, int cw)
{
    using C = GjCfg<NBI, 1, RPT>;
    constexpr int S = NBI;
    constexpr int KS = (NBI + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx* rowbuf = reinterpret_cast<cplx*>(smem_raw);            // [NBI]
    cplx* colbuf = rowbuf + NBI;                                 // [ROWS]
    cplx* qwin = colbuf + C::ROWS;                               // [NBI][WIN] pivot rows, window columns
    __shared__ RedSlot red[2][PW];
    __shared__ cplx piv_ip;
    __shared__ int bad_sh;
    __shared__ int team_ctr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;
    cplx* X = bufB + (size_t)blockIdx.x * mat_stride;
    int* pivrow = piv_all + (size_t)blockIdx.x * 2 * n;
    int* colof = pivrow + n;
    int team_expect = 0;
    if (tid == 0) { bad_sh = 0; team_ctr = 0; }
    __syncthreads();

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = (n + 15) >> 4;
    const int wt0 = c0 >> 4, wt1 = (c0 + cw + 15) >> 4;          // column tiles of the window

    for (int k0 = c0; k0 < c0 + cw; k0 += NBI) {
        const int kw = min(NBI, c0 + cw - k0);
        __syncthreads();                 // previous window update (global stores) complete
        // ---- strips of the sub-panel, pivot steps (all 8 waves form the panel team)
        cplx a[RPT][S];
        bool avail[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + q * PT;
            const bool row_ok = r < n;
            avail[q] = row_ok && (colof[row_ok ? r : 0] < 0);
            const cplx* g = W + (size_t)(row_ok ? r : 0) * n + k0;
#pragma unroll
            for (int s = 0; s < S; ++s) a[q][s] = (row_ok && s < kw) ? g[s] : cmake(0.0, 0.0);
        }
        {
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
        team_sync(&team_ctr, team_expect, lane);
        {
            PanelCtx<NBI, 1, RPT> ctx{a, avail, rowbuf, colbuf, &red[0][0], &piv_ip, &bad_sh, pivrow, colof,
                                      &team_ctr, team_expect, n, k0, kw, tid, lane, wave, 0, tid, tid & ~63};
            PanelSteps<NBI, 1, RPT, 0>::run(ctx);
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + q * PT;
            if (r < n) {
                cplx* g = W + (size_t)r * n + k0;
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (s < kw) g[s] = a[q][s];
            }
        }
        __syncthreads();                 // sub-panel columns and pivrow/colof (global) visible
        // ---- pivot rows of this sub-panel, window columns -> LDS
        for (int t = tid; t < NBI * WIN; t += PT) {
            const int k = t / WIN, j = t - k * WIN;
            qwin[t] = (k < kw && c0 + j < n && j < cw) ? W[(size_t)pivrow[k0 + k] * n + c0 + j] : cmake(0.0, 0.0);
        }
        __syncthreads();
        // ---- apply the sub-panel transform to the other window columns (in place)
        for (int item = wave; item < tiles * (wt1 - wt0); item += PW) {
            const int ti = item / (wt1 - wt0), tj = wt0 + item % (wt1 - wt0);
            const int col = tj * 16 + fi;
            const bool col_ok = col < n && col >= c0 && col < c0 + cw;
            const bool col_store = col_ok && !(col >= k0 && col < k0 + kw);
            d4 accr = {0, 0, 0, 0}, acci = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                if (i < n && col_ok) {
                    const int cf = colof[i];
                    if (!(cf >= k0 && cf < k0 + kw)) { const cplx v = W[(size_t)i * n + col]; accr[r] = v.x; acci[r] = v.y; }
                }
            }
            const int prow = ti * 16 + fi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 4 + fk;
                cplx pa = cmake(0.0, 0.0), qb = cmake(0.0, 0.0);
                if (prow < n && k < kw) pa = W[(size_t)prow * n + k0 + k];
                if (k < kw && col_ok) qb = qwin[k * WIN + (col - c0)];
                accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.x, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, -qb.y, accr, 0, 0, 0);
                acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.y, acci, 0, 0, 0);
                acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, qb.x, acci, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r], acci[r]);
            }
        }
    }
    __syncthreads();
    // ---- Q snapshot for the big update: the cw pivot rows of this window, all columns
    for (int k = wave; k < cw; k += PW) {
        const cplx* srow = W + (size_t)pivrow[c0 + k] * n;
        cplx* drow = X + (size_t)k * n;
        for (int j = lane; j < n; j += 64) drow[j] = srow[j];
    }
    if (tid == 0 && bad_sh != 0 && info[blockIdx.x] == 0) info[blockIdx.x] = bad_sh;
}

// W[i][J] = (row i pivot of the window ? 0 : W[i][J]) + P'[i][0:cw) * Q[0:cw)[J]  for the column
// blocks outside the window.  Workgroup = 256 threads = 4 waves, 64 x 64 output block, wave tile
// 32 x 32 (2 x 2 MFMA tiles), K tile 16 through LDS (same tiling as zgemm_mfma_kernel).
__global__ __launch_bounds__(256) void gj_bigupdate_kernel(
    int n, cplx* __restrict__ bufA, const cplx* __restrict__ bufB, size_t mat_stride,
    const int* __restrict__ piv_all, int c0, int cw)
{
    constexpr int BM = 64, BN = 64, BK = 16, AP = BK + 1, BP = BN + 1;
    __shared__ cplx As[BM * AP];
    __shared__ cplx Bs[BK * BP];
    const int col0 = blockIdx.x * BN, row0 = blockIdx.y * BM;
    if (col0 >= c0 && col0 < c0 + cw) return;                    // window columns: already final (uniform)
    cplx* W = bufA + (size_t)blockIdx.z * mat_stride;
    const cplx* Q = bufB + (size_t)blockIdx.z * mat_stride;
    const int* colof = piv_all + (size_t)blockIdx.z * 2 * n + n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int fi = lane & 15, fk = lane >> 4;
    d4 accr[2][2], acci[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = row0 + wr + a * 16 + fk + 4 * r, gj = col0 + wc + c * 16 + fi;
                double vx = 0.0, vy = 0.0;
                if (gi < n && gj < n) {
                    const int cf = colof[gi];
                    if (!(cf >= c0 && cf < c0 + cw)) { const cplx v = W[(size_t)gi * n + gj]; vx = v.x; vy = v.y; }
                }
                accr[a][c][r] = vx; acci[a][c][r] = vy;
            }
    const int lr = tid >> 2, lc = (tid & 3) * 4;                 // A tile staging: row lr, k lc..lc+3
    const int br = tid >> 4, bc = (tid & 15) * 4;                // B tile staging: k br, cols bc..bc+3
    for (int k0 = 0; k0 < cw; k0 += BK) {
        {
            const int gi = row0 + lr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gk = k0 + lc + e;
                As[lr * AP + lc + e] = (gi < n && gk < cw) ? W[(size_t)gi * n + c0 + gk] : cmake(0.0, 0.0);
            }
            const int gk = k0 + br;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gj = col0 + bc + e;
                Bs[br * BP + bc + e] = (gk < cw && gj < n) ? Q[(size_t)gk * n + gj] : cmake(0.0, 0.0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            cplx af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[(wr + a * 16 + fi) * AP + ks + fk];
#pragma unroll
            for (int c = 0; c < 2; ++c) bf[c] = Bs[(ks + fk) * BP + wc + c * 16 + fi];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    accr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].x, bf[c].x, accr[a][c], 0, 0, 0);
                    accr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-af[a].y, bf[c].y, accr[a][c], 0, 0, 0);
                    acci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].x, bf[c].y, acci[a][c], 0, 0, 0);
                    acci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].y, bf[c].x, acci[a][c], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = row0 + wr + a * 16 + fk + 4 * r, gj = col0 + wc + c * 16 + fi;
                if (gi < n && gj < n) W[(size_t)gi * n + gj] = cmake(accr[a][c][r], acci[a][c][r]);
            }
}

__global__ __launch_bounds__(256) void gj_gather_kernel(int n, const cplx* __restrict__ bufA,
                                                         cplx* __restrict__ bufB, size_t mat_stride,
                                                         const int* __restrict__ piv_all)
{
    const cplx* W = bufA + (size_t)blockIdx.y * mat_stride;
    cplx* X = bufB + (size_t)blockIdx.y * mat_stride;
    const int* pivrow = piv_all + (size_t)blockIdx.y * 2 * n;
    const int* colof = pivrow + n;
    const int i = blockIdx.x;
    const cplx* srow = W + (size_t)pivrow[i] * n;
    for (int j = threadIdx.x; j < n; j += 256) X[(size_t)i * n + j] = srow[colof[j]];
}

__global__ void gj_state_init_kernel(int n, int* __restrict__ piv_all, int* __restrict__ info)
{
    int* p = piv_all + (size_t)blockIdx.x * 2 * n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) { p[t] = 0; p[n + t] = -1; }
    if (threadIdx.x == 0) info[blockIdx.x] = 0;
}

template <int NBI, int RPT>
void gj_large_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* piv, int* info)
{
    using C = GjCfg<NBI, 1, RPT>;
    const size_t smem = (size_t)(NBI + C::ROWS + NBI * WIN) * sizeof(cplx);
    auto kern = gj_window_kernel<NBI, RPT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(100 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(gj_state_init_kernel, dim3(nb), dim3(256), 0, st, n, piv, info);
    const int nblk = (n + 63) / 64;
    for (int c0 = 0; c0 < n; c0 += WIN) {
        const int cw = min(WIN, n - c0);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(PT), smem, st, n, A, B, stride, piv, info, c0, cw);
        hipLaunchKernelGGL(gj_bigupdate_kernel, dim3(nblk, nblk, nb), dim3(256), 0, st, n, A, B, stride,
                           (const int*)piv, c0, cw);
    }
    hipLaunchKernelGGL(gj_gather_kernel, dim3(n, nb), dim3(256), 0, st, n, (const cplx*)A, B, stride,
                       (const int*)piv);
}

int gj_large_pick(int n)
{
    if (n <= 512 || n < 64) return 0;
    if (n <= PT * 2) return 1;           // <= 1024: sub-panel 8, 2 rows per thread
    if (n <= PT * 4) return 2;           // <= 2048: sub-panel 8, 4 rows per thread
    if (n <= PT * 8) return 3;           // <= 4096: sub-panel 4, 8 rows per thread
    return 0;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0 || gj_large_pick(n) != 0; }

// In-place reduction of A with B as scratch; the inverses are gathered into B.
// piv: [nb][2][n] ints of pivot bookkeeping (used by the large-matrix path).
// Returns true: the result is in B.
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* piv, int* info)
{
    switch (gj_pick(n)) {
    case 1: gj_launch<32, 2, 1>(st, n, nb, A, B, stride, info); return true;
    case 2: gj_launch<16, 1, 1>(st, n, nb, A, B, stride, info); return true;
    default: break;
    }
    switch (gj_large_pick(n)) {
    case 1: gj_large_launch<8, 2>(st, n, nb, A, B, stride, piv, info); return true;
    case 2: gj_large_launch<8, 4>(st, n, nb, A, B, stride, piv, info); return true;
    case 3: gj_large_launch<4, 8>(st, n, nb, A, B, stride, piv, info); return true;
    default: return false;
    }
}
