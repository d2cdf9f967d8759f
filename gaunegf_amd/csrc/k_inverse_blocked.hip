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

typedef double d4 __attribute__((ext_vector_type(4)));

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
               32 = no pivot steps */)
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
            __syncthreads();             // [A] Pt, Q snapshot visible; pivot rows of W are not read again
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
            __syncthreads();             // [A2] block column s+1 is up to date
        }
        // ---- P-team: factor block column s+1; U-team: update the other columns.  The update items
        // come from a shared queue, so the P-team joins in as soon as its panel is done.
        if (pwave && has_next) factor_panel(n0, nw);
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
        __syncthreads();                 // [B] step complete everywhere; block column s+1 factored
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
    if (dbg < 0) { const char* e = getenv("NEGF_GJ_DEBUG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(GJB_THREADS), smem, st, n, A, B, stride, info, dbg);
}

// which configuration serves dimension n: 0 = none.  The Q snapshot needs NB*n <= n*n.
int gj_pick(int n)
{
    if (n < 32) return 0;                               // small matrices: the unblocked kernel
    if (gj_fits<32, 2, 1>(n)) return 1;                 // n <= 256, panel 32, two threads per row
    if (gj_fits<16, 1, 1>(n)) return 2;                 // n <= 512, panel 16
    return 0;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0; }

// In-place reduction of A with B as scratch; the inverses are gathered into B.
// Returns true: the result is in B.
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    switch (gj_pick(n)) {
    case 1: gj_launch<32, 2, 1>(st, n, nb, A, B, stride, info); break;
    case 2: gj_launch<16, 1, 1>(st, n, nb, A, B, stride, info); break;
    default: return false;
    }
    return true;
}
