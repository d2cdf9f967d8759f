// Blocked Gauss-Jordan inversion with partial pivoting, FP64 MFMA trailing updates and
// panel LOOK-AHEAD, one workgroup (768 threads, 12 waves) per matrix.   gfx950 / MI355X.
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
//      complex128) per lane; the two strips of a panel row sit in the same wave (lanes l and
//      l+32).  kw unblocked Gauss-Jordan column steps with partial pivoting (|re|+|im| as
//      LAPACK izamax; among not-yet-used rows) run on the strips with ONE team barrier per
//      column: after updating its strips for column j every wave finds its best candidate
//      for column j+1 (DPP max of a packed 64-bit key), publishes that candidate's whole,
//      already updated panel row to LDS and merges the key with an LDS atomic max; after the
//      barrier every lane reads the winning key and the winner's row.  The team barrier is an
//      LDS counter (gfx950 has one hardware barrier per workgroup) without fences: DS
//      instructions of a wave execute in order.  The phase is a latency chain, so it runs
//      CONCURRENTLY with the update of the previous block column:
//   U-team (waves 8-11, joined by the P-team once its panel is done: the update items come
//      from a shared LDS work queue): TRAILING UPDATE of step k on the matrix cores, in place:
//         W[i][J] = (i pivot row of panel k ? 0 : W[i][J]) + P_k[i][:] * Q_k[:][J]
//      with P_k in LDS (k-major, conflict-free A-operand reads), the Q_k fragments of a
//      column tile in registers, the next C tile in flight during the MFMAs, no divergent
//      branch in the loop (clamped loads + selects, pivot rows from a bit mask); a 16x16
//      complex tile = 4 real v_mfma_f64_16x16x4_f64 chains per 4-deep k-step
//      (Cr += Pr Qr; Cr += Pi (-Qi); Ci += Pr Qi; Ci += Pi Qr).
//   Look-ahead: at step k all waves first apply step k to the columns of panel k+1
//      (a few tiles), then the P-team factors panel k+1 in registers while the U-team updates
//      all other columns.  The teams meet at a workgroup barrier; the columns of panel k+1
//      then become P_{k+1} in LDS and the pivot rows are snapshotted as Q_{k+1}.
// Flops: 8 n^3 per matrix (complex MAC = 8) -- the LU + triangular-inversion optimum.
// Pivot candidates are compared on the upper 36 mantissa bits of |.|_1; ties go to the lower
// physical row index (LAPACK: lower logical index).  A singular or NaN matrix is reported through
// info (1-based column) and its result is NaN-filled.
//
// Data layout: row-major complex128 (interleaved), ld = n.  A lane fetches one
// complex element (16 B) per MFMA operand; 16 lanes cover 256 contiguous bytes of
// a matrix row, so tile loads/stores are 4 x 256-B row segments per wave instruction.
#include "negf_common.h"
#include "wave_utils.h"

namespace {

// NB panel width, CPR strips per panel row, RPT rows per panel thread, PW panel-team waves,
// NW waves of the workgroup (the NW - PW others only do trailing updates)
template <int NB_, int CPR_, int RPT_, int PW_, int NW_>
struct GjCfg {
    static constexpr int NB = NB_, CPR = CPR_, RPT = RPT_, PW = PW_, NW = NW_;
    static constexpr int PT = PW * 64;             // panel threads
    static constexpr int THREADS = NW * 64;
    static constexpr int S = NB / CPR;             // complex values per strip (1/CPR of a panel row)
    static constexpr int LPR = 64 / CPR;           // rows per wave and slab: the CPR strips of a row sit in
                                                   // ONE wave, at lanes  l, l + LPR, ...
    static constexpr int TPR = PW * LPR;           // rows per slab (= PT / CPR)
    static constexpr int ROWS = TPR * RPT;         // row capacity
};

// ---- team barrier on an LDS counter (monotonic).  All data the team exchanges lives in LDS and
// the DS instructions of a wave execute in issue order, so the counter increment is performed
// after the wave's earlier LDS writes without waiting for them to return (no release wait);
// the reads that follow are issued after the polling read has returned (no acquire action).
// The asm statements only stop the compiler from moving memory operations across the barrier.
template <int PW>
__device__ __forceinline__ void team_sync(int* ctr, int& expect, int lane)
{
    asm volatile("" ::: "memory");
    expect += PW;
    if (lane == 0) {                     // one lane polls: the LDS sees 4-byte reads, not 64 x 4
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expect) { }
    }
    asm volatile("" ::: "memory");
}

typedef unsigned long long u64;

// Pivot candidates are ordered by ONE 64-bit integer: the upper 48 bits of |value| (as a
// non-negative double, so integer order == numeric order) over 16 bits of (0xFFFF - row):
// larger magnitude wins, equal magnitudes (to 36 mantissa bits) go to the lower row.  0 = no
// candidate (no available row, or only NaNs).  Deterministic; a pivot within 2^-36 of the column
// maximum is as good as the maximum for the growth bound.
__device__ __forceinline__ u64 cand_key(double v, int row)
{
    return ((u64)__double_as_longlong(v) & ~0xFFFFull) | (u64)(0xFFFF - row);
}

template <int CTRL>
__device__ __forceinline__ u64 dpp_max_u64(u64 k)
{
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    const u64 o = ((u64)ohi << 32) | olo;
    return o > k ? o : k;
}

// maximum over the 64 lanes, returned wave-uniform.  All lanes must be active.
__device__ __forceinline__ u64 wave_max_u64(u64 k)
{
    k = dpp_max_u64<0xB1>(k);      // quad_perm [1,0,3,2]
    k = dpp_max_u64<0x4E>(k);      // quad_perm [2,3,0,1]
    k = dpp_max_u64<0x141>(k);     // row_half_mirror
    k = dpp_max_u64<0x140>(k);     // row_mirror -> every lane of a row of 16 holds the row maximum
    u64 best = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, r * 16);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), r * 16);
        const u64 o = ((u64)hi << 32) | lo;
        best = o > best ? o : best;
    }
    return best;
}

// ---- one Gauss-Jordan column step on the register strips (P-team only), J compile-time ----
// ONE team barrier per column: before the barrier every wave publishes its best pivot candidate
// for the next column (an LDS atomic max on the candidate key) TOGETHER with that candidate's whole
// (already updated) panel row; after the barrier every thread reads the winning key and the
// winner's row -- there is no second "owner publishes the pivot row" round trip.  Candidate rows
// are double buffered by column parity and the key slots triple buffered, so a fast wave
// publishing for column J+1 never disturbs what a slow wave still reads for column J.
template <class C>
struct PanelCtx {
    cplx (&a)[C::RPT][C::S];
    bool (&avail)[C::RPT];           // row not used as a pivot yet
    cplx* cand;                      // [2][PW][NB] candidate pivot rows
    u64* slot;                       // [3] winning candidate key of a column
    int* bad_sh;
    int* pivrow; int* colof; int* team_ctr; int& team_expect;
    int n, k0, kw, tid, lane, wave, h, tr;
    unsigned long long* st;          // diagnostic stamps (nullptr in production)
};

template <class C>
__device__ __forceinline__ void pstamp(PanelCtx<C>& x, int slot)
{
    if (x.st && x.tid == 0) x.st[slot] = __builtin_amdgcn_s_memrealtime();
}

// col: panel-relative column whose pivot is sought; its entries sit at strip position sn of part hn
template <class C>
__device__ __forceinline__ void publish_candidate(PanelCtx<C>& x, int col, int hn, int sn)
{
    constexpr int NB = C::NB, RPT = C::RPT, PW = C::PW, S = C::S, TPR = C::TPR;
    u64 key = 0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = x.tr + q * TPR;
        const double v = cabs1(x.a[q][sn]);
        const bool mine = (x.h == hn) & (r < x.n) & x.avail[q] & (v == v);
        const u64 k = mine ? cand_key(v, r) : 0ull;
        key = k > key ? k : key;
    }
    key = wave_max_u64(key);
    const int brow = 0xFFFF - (int)(key & 0xFFFFull);           // meaningless when key == 0
    cplx* cn = x.cand + ((size_t)((col & 1) * PW + x.wave)) * NB + x.h * S;
#pragma unroll
    for (int q = 0; q < RPT; ++q)
        if (key != 0 && x.tr + q * TPR == brow) {
#pragma unroll
            for (int s = 0; s < S; ++s) cn[s] = x.a[q][s];
        }
    if (x.lane == 0 && key != 0)
        __hip_atomic_fetch_max(x.slot + col % 3, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <class C, int J>
struct PanelSteps {
    static __device__ __forceinline__ void run(PanelCtx<C>& x)
    {
        constexpr int NB = C::NB, CPR = C::CPR, RPT = C::RPT, PW = C::PW;
        constexpr int S = C::S, TPR = C::TPR, LPR = C::LPR;
        constexpr int hj = J / S, sj = J % S;
        if (J < x.kw) {                                     // uniform branch
            const int c = x.k0 + J;
            pstamp(x, 300 + J);
            // (1) the winning candidate.  key == 0: no usable row (a column of NaNs) -- the step
            // then runs with no pivot row (nothing is indexed by it) and the matrix is reported
            // through info; (key >> 16) == 0: the column maximum is exactly zero (singular).
            const u64 key = x.slot[J % 3];
            const bool none = key == 0;
            const int pphys = none ? -1 : 0xFFFF - (int)(key & 0xFFFFull);
            const int ww = none ? 0 : (pphys % TPR) / LPR;
            if ((key >> 16) == 0 && x.tid == 0 && *x.bad_sh == 0) *x.bad_sh = c + 1;
            if (x.tid == 0) x.slot[(J + 2) % 3] = 0;        // slot of column J+2: last read in step J-1
            if constexpr (J == 4) pstamp(x, 340);
            // (2) the winner's row part for my columns, 1/pivot (computed by every thread)
            const cplx* prow = x.cand + (size_t)((J & 1) * PW + ww) * NB;
            const cplx pv = prow[J];
            // 1/pivot = conj(pivot) / |pivot|^2 (one division; |pivot| is far from the overflow
            // range for these matrices)
            const double sc = 1.0 / (pv.x * pv.x + pv.y * pv.y);
            const cplx ip = cmake(pv.x * sc, -pv.y * sc);
            // (3) -(f / pivot) with f = my row's entry of column J, held by the lane of part hj
            cplx nfm[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                cplx f = x.a[q][sj];
                if constexpr (CPR > 1) {
                    const int src = (x.lane % LPR) + hj * LPR;
                    f.x = __shfl(f.x, src, 64);
                    f.y = __shfl(f.y, src, 64);
                }
                nfm[q] = cneg(cmul(f, ip));
            }
            if constexpr (J == 4) pstamp(x, 341);
            if (x.tid == 0 && !none) { x.pivrow[c] = pphys; x.colof[pphys] = c; }
            bool wave_has_piv = false;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int d = pphys - q * TPR - x.wave * LPR;
                wave_has_piv |= (d >= 0 && d < LPR);
            }
            // (4) rank-1 update of every strip in sub-strips of 8 (bounded register use): a batch
            //     of LDS reads of the pivot row part, then register arithmetic
            // (a pure-panel configuration with extra update waves runs 12 waves: 168 VGPRs, pivot row in chunks of 4)
            constexpr int HS = (C::CPR == 1 && (C::NW > C::PW || C::RPT >= 2)) ? (S >= 4 ? 4 : S) : ((S >= 8) ? 8 : S);
#pragma unroll
            for (int s0 = 0; s0 < S; s0 += HS) {
                cplx rb[HS];
#pragma unroll
                for (int s = 0; s < HS; ++s) rb[s] = prow[x.h * S + s0 + s];
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
            if constexpr (J == 4) pstamp(x, 342);
            if constexpr (J == 4) { if (x.st && x.lane == 0) { x.st[360 + x.wave] = __builtin_amdgcn_s_memrealtime(); x.st[370 + x.wave] = wave_has_piv; } }
            // (5) candidates for column J+1 from the freshly updated strips
            if constexpr (J + 1 < NB) {
                if (J + 1 < x.kw) publish_candidate<C>(x, J + 1, (J + 1) / S, (J + 1) % S);
            }
            if constexpr (J == 4) pstamp(x, 343);
            if constexpr (J == 4) { if (x.st && x.lane == 0) x.st[350 + x.wave] = __builtin_amdgcn_s_memrealtime(); }
            team_sync<PW>(x.team_ctr, x.team_expect, x.lane);
            if constexpr (J == 4) pstamp(x, 344);
            // last column of the panel: every wave has read its slot by now; leave all three zero
            if (J + 1 >= x.kw && x.tid == 0) x.slot[J % 3] = 0;
            if constexpr (J + 1 < NB) PanelSteps<C, J + 1>::run(x);
        }
    }
};

// ---- trailing update of one (column tile tj, row tiles [ti0, ti1)) item with the panel in LDS:
//        W[i][col] = (i pivot row of this panel ? 0 : W[i][col]) + sum_k P[i][k] * Q[k][col]
// Q fragments stay in registers for the whole item, P comes from LDS (k-major), the C tile of the
// next row tile is in flight while the current one runs its MFMAs.  All loads are unconditional on
// clamped (always valid) addresses and masked by selects afterwards: no divergent branches in the
// loop, so the loads of the next tile really stay outstanding across the MFMAs.  Always KS k-steps:
// rows k >= kw of P and Q are zero.  Stores go to columns with lo <= col < hi (inside) or to the
// others (!inside), never to the panel's own columns.
template <int KS>
__device__ __forceinline__ void gj_update_item(const cplx* __restrict__ Pt, int rows16, cplx* W, const cplx* X,
                                               const unsigned* rowmask, int n, int k0, int kw, int tj, int ti0,
                                               int ti1, int st_lo, int st_hi, bool inside, int lane)
{
    const int fi = lane & 15, fk = lane >> 4;
    const int col = tj * 16 + fi;
    const bool col_ok = col < n;
    const int colc = col_ok ? col : n - 1;
    const bool in_rng = col >= st_lo && col < st_hi;
    const bool col_store = col_ok && !(col >= k0 && col < k0 + kw) && (inside ? in_rng : !in_rng);
    cplx qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 4 + fk;
        const cplx v = X[(size_t)(k < kw ? k : kw - 1) * n + colc];
        const bool ok = (k < kw) & col_ok;
        qf[ks] = cmake(ok ? v.x : 0.0, ok ? v.y : 0.0);
    }
    const cplx* wcol = W + colc;
    cplx c0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c0[r] = wcol[(size_t)min(ti0 * 16 + fk + 4 * r, n - 1) * n];
    for (int ti = ti0; ti < ti1; ++ti) {
        const unsigned m = rowmask[ti >> 1] >> ((ti & 1) * 16 + fk);    // bit 4r: row fk + 4r of this tile
        // complex product in 3M form: s1 = sum pr qr, s2 = sum pi qi, s3 = sum (pr + pi)(qr + qi); the old tile seeds
        // s1 (real part) and s3 (real + imaginary part): re = s1 - s2, im = s3 - s1 - s2.  Three matrix instructions per
        // k-step instead of four: the FP64 matrix instruction is what these kernels' time is made of.
        d4 accr, acci, accs = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool z = (m >> (4 * r)) & 1u;
            accr[r] = z ? 0.0 : c0[r].x; acci[r] = z ? 0.0 : c0[r].x + c0[r].y;
        }
        const int tn = min(ti + 1, ti1 - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[r] = wcol[(size_t)min(tn * 16 + fk + 4 * r, n - 1) * n];
        const cplx* pcol = Pt + (size_t)fk * rows16 + ti * 16 + fi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const cplx pa = pcol[(size_t)ks * 4 * rows16];
            accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qf[ks].x, accr, 0, 0, 0);
            accs = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, qf[ks].y, accs, 0, 0, 0);
            acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x + pa.y, qf[ks].x + qf[ks].y, acci, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + fk + 4 * r;
            if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r] - accs[r], acci[r] - accr[r] - accs[r]);
        }
    }
}

template <class C>
__global__ __launch_bounds__(C::THREADS) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info,
    int dbg /* ablation switches, 0 in production: 16 = U-team idle, 32 = no pivot steps */,
    unsigned long long* __restrict__ stamps /* diagnostic build only (NEGF_GJ_STAMPS): wall-clock
               stamps of workgroup 0, [step+1][8]; nullptr in production */)
{
    constexpr int NB = C::NB, RPT = C::RPT, PW = C::PW;
    constexpr int GJB_THREADS = C::THREADS, GJB_WAVES = C::NW;
    constexpr int S = C::S, TPR = C::TPR;
    constexpr int KS = NB / 4;                     // MFMA k-steps per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows16 = (n + 15) & ~15;
    cplx* Pt = reinterpret_cast<cplx*>(smem_raw);                // [NB][rows16]  P, k-major, physical rows
    cplx* cand = Pt + (size_t)NB * rows16;                       // [2][PW][NB] candidate pivot rows
    int* pivrow = reinterpret_cast<int*>(cand + 2 * PW * NB);    // [rows16] physical pivot row of column c
    int* colof = pivrow + rows16;                                // [rows16] column a row was pivot for, or -1
    __shared__ u64 slot[3];
    __shared__ int bad_sh;
    __shared__ int team_ctr;
    __shared__ int next_item;        // work queue of the update items of the current step
    __shared__ unsigned rowmask[16]; // bit r: row r was a pivot row of the current panel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool pwave = wave < PW;
    const int h = lane / C::LPR;                              // column part of the panel row held by this lane
    const int tr = (wave % PW) * C::LPR + lane % C::LPR;      // row slot
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;            // the matrix, updated in place
    cplx* X = bufB + (size_t)blockIdx.x * mat_stride;            // Q snapshots, then the result
    int team_expect = 0;

    if (tid == 0) { bad_sh = 0; team_ctr = 0; }
    if (tid < 3) slot[tid] = 0;
    if (stamps && blockIdx.x == 0 && tid == 0) {
        stamps[400] = __builtin_amdgcn_s_memrealtime();
        stamps[401] = __builtin_amdgcn_s_memtime();
    }
    for (int t = tid; t < rows16; t += GJB_THREADS) { colof[t] = -1; pivrow[t] = 0; }
    // rows >= n of P stay zero for the whole kernel (A operand of the edge tiles)
    for (int t = tid; t < NB * (rows16 - n); t += GJB_THREADS) {
        const int k = t / (rows16 - n), r = n + t - k * (rows16 - n);
        Pt[(size_t)k * rows16 + r] = cmake(0.0, 0.0);
    }
    __syncthreads();

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
        // the slots were left at zero by the previous panel (every column resets the slot two ahead;
        // the last two columns of a panel publish nothing into theirs)
        PanelCtx<C> ctx{a, avail, cand, slot, &bad_sh, pivrow, colof,
                                   &team_ctr, team_expect, n, p0, pw, tid, lane, wave, h, tr,
                                   (stamps && blockIdx.x == 0 && p0 == 2 * NB) ? stamps : nullptr};
        publish_candidate<C>(ctx, 0, 0, 0);
        team_sync<PW>(&team_ctr, team_expect, lane);
        if (!(dbg & 32)) {
            PanelSteps<C, 0>::run(ctx);
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
            if (tid >= GJB_THREADS - 16) {               // 16 lanes of the last wave: 32 rows each
                const int t = tid - (GJB_THREADS - 16);
                unsigned m = 0;
                for (int b = 0; b < 32; ++b) {
                    const int r = t * 32 + b;
                    if (r < n) { const int cf = colof[r]; m |= (cf >= k0 && cf < k0 + kw) ? (1u << b) : 0u; }
                }
                rowmask[t] = m;
            }
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
                    gj_update_item<KS>(Pt, rows16, W, X, rowmask, n, k0, kw, tj, part * rq, min(tiles, part * rq + rq), n0,
                                       n0 + nw, true, lane);
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
                const int r0 = part ? rhalf : 0, r1 = part ? tiles : rhalf;
                const int lo = has_next ? n0 : 0, hi = has_next ? n0 + nw : 0;
                // a narrow last panel runs fewer k-steps
                if (KS > 4 && kw <= 8) gj_update_item<2>(Pt, rows16, W, X, rowmask, n, k0, kw, tj, r0, r1, lo, hi, false, lane);
                else if (KS > 4 && kw <= 16) gj_update_item<4>(Pt, rows16, W, X, rowmask, n, k0, kw, tj, r0, r1, lo, hi, false, lane);
                else gj_update_item<KS>(Pt, rows16, W, X, rowmask, n, k0, kw, tj, r0, r1, lo, hi, false, lane);
            }
        }
        stamp(step, 5, 0);
        stamp(step, 6, PW);              // a U-team wave: end of its update work
        __syncthreads();                 // [B] step complete everywhere; block column s+1 factored
        stamp(step, 7, 0);
    }
    if (tid == 0) info[blockIdx.x] = bad_sh;
    // a singular (or NaN) matrix has no inverse: its result is NaN-filled and the pivot
    // bookkeeping, possibly incomplete, is not used as an index
    const bool dead = bad_sh != 0;
    const double fillv = __builtin_nan("");
    if (stamps && blockIdx.x == 0 && tid == 0) {
        stamps[402] = __builtin_amdgcn_s_memrealtime();
        stamps[403] = __builtin_amdgcn_s_memtime();
    }
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
                    v[rr][jj] = cmake(fillv, fillv);
                    if (i < n && j < n && !dead) v[rr][jj] = W[(size_t)pivrow[i] * n + colof[j]];
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

template <class C>
size_t gj_smem(int n)
{
    constexpr int NB = C::NB, PW = C::PW;
    const size_t rows16 = (size_t)((n + 15) & ~15);
    return (size_t)NB * rows16 * sizeof(cplx) + (size_t)2 * PW * NB * sizeof(cplx) + 2 * rows16 * sizeof(int);
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 1024;      // static __shared__ of the kernel is < 1 KB

template <class C>
bool gj_fits(int n)
{
    return n <= C::ROWS && gj_smem<C>(n) <= LDS_LIMIT;
}

template <class C>
void gj_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    auto kern = gj_blocked_kernel<C>;
    const size_t smem = gj_smem<C>(n);
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
    hipLaunchKernelGGL(kern, dim3(nb), dim3(C::THREADS), smem, st, n, A, B, stride, info, dbg, d_stamps);
    if (d_stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[64 * 8];
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        const int np = (n + C::NB - 1) / C::NB;
        fprintf(stderr, "[gj stamps] 100 MHz ticks relative to step start; cols: A-in A-out A2-in A2-out panel-done upd-done(P) upd-done(U) B-out\n");
        unsigned long long t0 = h[7];
        for (int sidx = 0; sidx <= np; ++sidx) {
            fprintf(stderr, "[gj stamps] step %2d:", sidx - 1);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %8.2f", h[sidx * 8 + k] ? (double)(h[sidx * 8 + k] - t0) / 100.0 : -1.0);
            fprintf(stderr, "  (us since end of step -1)\n");
        }
        fprintf(stderr, "[gj stamps] panel 2, step 4 per wave (us from step start; * = wave holding the pivot row): update done");
        for (int w = 0; w < C::PW; ++w) fprintf(stderr, " %.2f%s", (double)(h[360 + w] - h[304]) / 100.0, h[370 + w] ? "*" : "");
        fprintf(stderr, " | barrier arrival");
        for (int w = 0; w < C::PW; ++w) fprintf(stderr, " %.2f", (double)(h[350 + w] - h[304]) / 100.0);
        fprintf(stderr, "\n");
        fprintf(stderr, "[gj stamps] shader clock during the kernel: %.0f MHz (s_memtime / s_memrealtime)\n",
                (double)(h[403] - h[401]) / ((double)(h[402] - h[400]) / 100.0));
        fprintf(stderr, "[gj stamps] panel 2, pivot step starts (us):");
        for (int j = 0; j < 32; ++j) fprintf(stderr, " %.2f", h[300 + j] ? (double)(h[300 + j] - h[300]) / 100.0 : -1.0);
        fprintf(stderr, "\n[gj stamps] panel 2, step 4 phases (us from step start): winner %.2f  recip+f %.2f  update %.2f  publish %.2f  barrier %.2f\n",
                (double)(h[340] - h[304]) / 100.0, (double)(h[341] - h[304]) / 100.0, (double)(h[342] - h[304]) / 100.0,
                (double)(h[343] - h[304]) / 100.0, (double)(h[344] - h[304]) / 100.0);
    }
}

// which configuration serves dimension n: 0 = none.  The Q snapshot needs NB*n <= n*n.
using CfgSplit = GjCfg<32, 2, 1, 8, 12>;   // n <= 256: panel 32, a row split over two lanes, 8 + 4 waves

int gj_pick(int n)
{
    if (n < 32) return 0;                               // small matrices: the unblocked kernel
    if (gj_fits<CfgSplit>(n)) return 1;
    return 0;
}


// ======================================================================================
// Large matrices (257 <= n <= 8192): the same in-place, implicit-pivot Gauss-Jordan reduction,
// blocked at TWO levels across kernels.  For each window of WIN = 64 columns:
//   gj_window_kernel  (one workgroup per matrix): factors the n x 64 block column in
//       sub-panels of NBI columns with the register-strip pivot steps above and applies
//       every sub-panel transform to the 64 window columns only (MFMA, operands L2-hot);
//       afterwards the window holds the block column P' of the combined transform; the other
//       columns of its 64 pivot rows (Q) are still untouched.
//   gj_colupdate_kernel (one workgroup per block of 64 columns outside the window, all rows):
//       W[i][J] = (i pivot row of this window ? 0 : W[i][J]) + P'[i][:] * Q[:][J]
//       on the FP64 matrix cores with K = 64, Q in registers, P' streamed through LDS by LDS-DMA
//       (details at the kernel).
// pivrow / colof live in global memory between launches.  gj_gather_kernel forms
// G[i][j] = W[pivrow[i]][colof[j]].
// ======================================================================================
constexpr int WIN = 64;
constexpr int PW = 8;                              // window kernel: all 8 waves factor the sub-panels
constexpr int PT = PW * 64;

template <int NBI, int RPT>
__global__ __launch_bounds__(PT) void gj_window_kernel(
    int n, cplx* __restrict__ bufA, size_t mat_stride,
    int* __restrict__ piv_all /* [nb][2][n]: pivrow, colof */, int* __restrict__ info, int c0, int cw,
    unsigned long long* __restrict__ stamps /* diagnostic (NEGF_GJ_STAMPS): workgroup 0, window 1; nullptr in production */)
{
    using C = GjCfg<NBI, 1, RPT, PW, PW>;
    constexpr int S = NBI;
    constexpr int KS = (NBI + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx* cand = reinterpret_cast<cplx*>(smem_raw);              // [2][PW][NBI] candidate pivot rows
    cplx* qwin = cand + 2 * PW * NBI;                            // [NBI][WIN] pivot rows, window columns
    __shared__ u64 slot[3];
    __shared__ int bad_sh;
    __shared__ int team_ctr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;
    int* pivrow = piv_all + (size_t)blockIdx.x * 2 * n;
    int* colof = pivrow + n;
    int team_expect = 0;
    if (tid == 0) { bad_sh = 0; team_ctr = 0; }
    if (tid < 3) slot[tid] = 0;
    __syncthreads();

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = (n + 15) >> 4;
    cplx* qwin2 = qwin + NBI * WIN;                              // Q of the second sub-panel of a pair

    int sti = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (stamps && blockIdx.x == 0 && tid == 0 && sti < 60) stamps[sti] = __builtin_amdgcn_s_memrealtime();
        ++sti;
    };
    stamp();
    // ---- strips of sub-panel [k0, k0 + kw) into registers, pivot steps (all 8 waves form the panel team), strips back
    auto factor = [&](int k0, int kw) __attribute__((always_inline)) {
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
            // the slots were left at zero by the previous panel (every column resets the slot two ahead;
            // the last two columns of a panel publish nothing into theirs)
            PanelCtx<C> ctx{a, avail, cand, slot, &bad_sh, pivrow, colof,
                            &team_ctr, team_expect, n, k0, kw, tid, lane, wave, 0, tid, nullptr};
            stamp();                     // strips loaded
            publish_candidate<C>(ctx, 0, 0, 0);
            team_sync<PW>(&team_ctr, team_expect, lane);
            PanelSteps<C, 0>::run(ctx);
            stamp();                     // pivot steps done
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
    };
    // ---- pivot rows of sub-panel [k0, k0 + kw), window columns -> LDS
    auto stage_q = [&](cplx* qdst, int k0, int kw) __attribute__((always_inline)) {
        for (int t = tid; t < NBI * WIN; t += PT) {
            const int k = t / WIN, j = t - k * WIN;
            qdst[t] = (k < kw && c0 + j < n && j < cw) ? W[(size_t)pivrow[k0 + k] * n + c0 + j] : cmake(0.0, 0.0);
        }
    };
    // ---- apply sub-panel a = [ka, ka + kwa) (Q rows in qa) and, with kwb > 0, sub-panel b as well to the window
    // columns [clo, chi) minus the sub-panels' own columns, in place:
    //     W[i][col] = (i pivot row of a or b ? 0 : W[i][col]) + Pa[i][:] Qa[:][col] (+ Pb[i][:] Qb[:][col])
    // A wave takes whole row tiles: the P operands (the sub-panels' columns of the 16 rows) and the rows' pivot flags are
    // fetched once per row tile and serve all column tiles; every load is issued up front on clamped addresses
    // (selects afterwards, no branch between a load and its MFMA) -- the operands come from L2 / HBM, and a dependent
    // chain of such loads per tile was what this phase spent its time on.
    constexpr int WT = WIN / 16;                                     // column tiles of a full window
    auto update = [&](int clo, int chi, int ka, int kwa, const cplx* qa, int kb, int kwb, const cplx* qb) __attribute__((always_inline)) {
        const bool two = kwb > 0;                                    // (uniform)
        const int t_lo = (clo - c0) >> 4, t_hi = (chi - c0 + 15) >> 4;
        for (int ti = wave; ti < tiles; ti += PW) {
            const int prow = min(ti * 16 + fi, n - 1);
            cplx pa[KS], pb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pa[ks] = W[(size_t)prow * n + min(ka + ks * 4 + fk, n - 1)];   // k >= kw pairs with Q == 0
            if (two) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) pb[ks] = W[(size_t)prow * n + min(kb + ks * 4 + fk, n - 1)];
            }
            bool keep[4];
            const cplx* crow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = min(ti * 16 + fk + 4 * r, n - 1);
                const int cf = colof[i];
                keep[r] = !(cf >= ka && cf < ka + kwa) && !(two && cf >= kb && cf < kb + kwb);
                crow[r] = W + (size_t)i * n;
            }
            // the C tiles of all column tiles at once where the register strips leave room (<= 2 rows per
            // thread), tile by tile otherwise
            constexpr int CB = RPT <= 2 ? WT : 1;
            cplx cv[CB][4];
            if (CB == WT) {
#pragma unroll
                for (int t = 0; t < CB; ++t) {
                    if (t >= t_lo && t < t_hi) {                     // (uniform)
                        const int colc = min(c0 + t * 16 + fi, n - 1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) cv[t][r] = crow[r][colc];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if (t >= t_lo && t < t_hi) {                         // (uniform)
                    const int col = c0 + t * 16 + fi;
                    const bool col_store = col < n && col < c0 + cw && col >= clo && col < chi &&
                                           !(col >= ka && col < ka + kwa) && !(two && col >= kb && col < kb + kwb);
                    if (CB == 1) {
                        const int colc = min(col, n - 1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) cv[0][r] = crow[r][colc];
                    }
                    d4 accr, acci, accs = {0, 0, 0, 0};                  // 3M form, see gj_update_item
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx c = cv[CB == WT ? t : 0][r];
                        accr[r] = keep[r] ? c.x : 0.0; acci[r] = keep[r] ? c.x + c.y : 0.0;
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        // (zero outside the sub-panel / window; sub-panels narrower than a k-step have no such row)
                        const int kq = ks * 4 + fk;
                        cplx q = qa[(NBI % 4 == 0 ? kq : min(kq, NBI - 1)) * WIN + t * 16 + fi];
                        if (NBI % 4 != 0 && kq >= NBI) q = cmake(0.0, 0.0);
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x, q.x, accr, 0, 0, 0);
                        accs = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].y, q.y, accs, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x + pa[ks].y, q.x + q.y, acci, 0, 0, 0);
                    }
                    if (two) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const int kq = ks * 4 + fk;
                            cplx q = qb[(NBI % 4 == 0 ? kq : min(kq, NBI - 1)) * WIN + t * 16 + fi];
                            if (NBI % 4 != 0 && kq >= NBI) q = cmake(0.0, 0.0);
                            accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[ks].x, q.x, accr, 0, 0, 0);
                            accs = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[ks].y, q.y, accs, 0, 0, 0);
                            acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[ks].x + pb[ks].y, q.x + q.y, acci, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = ti * 16 + fk + 4 * r;
                        if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r] - accs[r], acci[r] - accr[r] - accs[r]);
                    }
                }
            }
        }
    };
    // ---- the same for ONE column tile (the two single-tile updates of a pair: A -> the columns of B, B -> the columns
    // of A).  With one 16 x 16 tile per row tile the loop above is a chain of memory round trips (28 us for 63 row
    // tiles at n = 1000 against 74 us for the fused pass over two tiles): here the operands of the wave's NEXT row tile
    // are requested before the current one runs its matrix instructions.
    auto update_one = [&](int clo, int chi, int ka, int kwa, const cplx* qa) __attribute__((always_inline)) {
        const int t = (clo - c0) >> 4;                               // the tile that holds [clo, chi)
        const int col = c0 + t * 16 + fi;
        const int colc = min(col, n - 1);
        const bool col_store = col < n && col < c0 + cw && col >= clo && col < chi && !(col >= ka && col < ka + kwa);
        cplx q[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = ks * 4 + fk;
            q[ks] = qa[(NBI % 4 == 0 ? kq : min(kq, NBI - 1)) * WIN + t * 16 + fi];
            if (NBI % 4 != 0 && kq >= NBI) q[ks] = cmake(0.0, 0.0);
        }
        cplx pa[2][KS], cv[2][4];
        int cf[2][4];
        auto fetch = [&](int ti, int b) __attribute__((always_inline)) {
            const int prow = min(ti * 16 + fi, n - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pa[b][ks] = W[(size_t)prow * n + min(ka + ks * 4 + fk, n - 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = min(ti * 16 + fk + 4 * r, n - 1);
                cf[b][r] = colof[i];
                cv[b][r] = W[(size_t)i * n + colc];
            }
        };
        auto compute = [&](int ti, auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            d4 accr, acci, accs = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = !(cf[b][r] >= ka && cf[b][r] < ka + kwa);
                accr[r] = keep ? cv[b][r].x : 0.0; acci[r] = keep ? cv[b][r].x + cv[b][r].y : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[b][ks].x, q[ks].x, accr, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[b][ks].y, q[ks].y, accs, 0, 0, 0);
                acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[b][ks].x + pa[b][ks].y, q[ks].x + q[ks].y, acci, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r] - accs[r], acci[r] - accr[r] - accs[r]);
            }
        };
        if (wave < tiles) fetch(wave, 0);
        for (int ti = wave; ti < tiles; ti += 2 * PW) {             // two row tiles per trip: the buffers are indexed statically
            if (ti + PW < tiles) fetch(ti + PW, 1);
            compute(ti, std::integral_constant<int, 0>());
            if (ti + PW < tiles) {
                if (ti + 2 * PW < tiles) fetch(ti + 2 * PW, 0);
                compute(ti + PW, std::integral_constant<int, 1>());
            }
        }
    };
    // Sub-panels in PAIRS (the same algebra as the window pairs of gj_colupdate2_kernel, one level down): sub-panel A is
    // applied to the columns of B only, B is factored and applied to the columns of A (-> P''A), and the other columns of
    // the window then take ONE rank-2*NBI update with [P''A | PB] and the RAW pivot rows of both -- the in-window update
    // streams the n x 64 window block through at Infinity-Cache bandwidth, and this way a third less of it.
    for (int k0 = c0; k0 < c0 + cw; ) {
        const int kw = min(NBI, c0 + cw - k0);
        const int k1 = k0 + NBI, kw1 = min(NBI, c0 + cw - k1);       // the second sub-panel of the pair (kw1 <= 0: none)
        __syncthreads();                 // previous window update (global stores) complete
        stamp();                         // 1 + 5 s: sub-panel s starts
        factor(k0, kw);
        __syncthreads();                 // sub-panel columns and pivrow/colof (global) visible
        stamp();                         // strips stored
        stage_q(qwin, k0, kw);
        __syncthreads();
        stamp();                         // Q of the sub-panel staged
        if (kw1 <= 0) {                  // a lone last sub-panel reaches all other window columns
            update(c0, c0 + cw, k0, kw, qwin, 0, 0, qwin);
            k0 += NBI;
            continue;
        }
        // (sub-panels of 16 -- the 256-VGPR instantiations with two row strips per lane -- spill into their pivot steps
        //  with the pipelined form: n = 800 83.9 -> 85.1 ms; the narrower ones gain: n = 2000 312 -> 296 ms)
        if constexpr (NBI < 16) update_one(k1, k1 + kw1, k0, kw, qwin);    // A -> the columns of B
        else update(k1, k1 + kw1, k0, kw, qwin, 0, 0, qwin);
        __syncthreads();
        stamp();
        factor(k1, kw1);
        __syncthreads();
        stamp();
        stage_q(qwin2, k1, kw1);
        __syncthreads();
        stamp();
        if constexpr (NBI < 16) update_one(k0, k0 + kw, k1, kw1, qwin2);   // B -> the columns of A: P''A
        else update(k0, k0 + kw, k1, kw1, qwin2, 0, 0, qwin2);
        __syncthreads();                 // P''A complete before any wave reads it as an operand
        update(c0, c0 + cw, k0, kw, qwin, k1, kw1, qwin2);           // everything else: one pass, both sub-panels
        k0 += 2 * NBI;
    }
    stamp();                             // thread 0's share of the last update done
    __syncthreads();
    stamp();                             // all updates done
    // (no Q snapshot: gj_colupdate_kernel reads the window's pivot rows in place)
    if (tid == 0 && bad_sh != 0 && info[blockIdx.x] == 0) info[blockIdx.x] = bad_sh;
}

// ---- Window kernel with LOOK-AHEAD inside the window (n <= 512: one row per panel-team lane).  The plain
// window kernel alternates two phases that leave most of the CU idle in turn -- the pivot steps of a
// sub-panel (a latency chain of LDS round trips and team barriers: 28 of the 75 us a sub-panel takes at
// n = 500) and the in-window update (all waves streaming the n x 64 window block through the matrix cores
// at Infinity-Cache bandwidth: 27-40 us).  Here twelve waves share them out:
//   after sub-panel s is factored, ALL waves apply it to the 16 columns of sub-panel s+1 only (look-ahead),
//   then waves 0-7 (the panel team) factor sub-panel s+1 WHILE waves 8-11 apply sub-panel s to the other
//   columns of the window; the last sub-panel's update is shared by all twelve.
// Transforms reach every column block in their order (each phase ends in a workgroup barrier); the panel
// team only touches the columns of the sub-panel it factors, the update team everything else.
// n <= 256 runs a lean form (4 panel waves + 2 update waves, TWO workgroups per CU at the same 168-VGPR budget): the
// panel team's barrier joins four waves instead of eight, of which four would hold no rows, and two workgroups
// cover each other's pivot chains.
template <int NBI, int LA_PW, int LA_NW>
__global__ __launch_bounds__(LA_NW * 64, LA_NW <= 6 ? 3 : 1) void gj_window_la_kernel(
    int n, cplx* __restrict__ bufA, size_t mat_stride,
    int* __restrict__ piv_all /* [nb][2][n]: pivrow, colof */, int* __restrict__ info, int c0, int cw,
    unsigned long long* __restrict__ stamps /* diagnostic (NEGF_GJ_STAMPS): workgroup 0, window 1; nullptr in production */)
{
    using C = GjCfg<NBI, 1, 1, LA_PW, LA_NW>;
    constexpr int S = NBI, KS = NBI / 4, WT = WIN / 16, LA_THREADS = LA_NW * 64;
    static_assert(NBI == 16, "sub-panel = one column tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx* cand = reinterpret_cast<cplx*>(smem_raw);              // [2][PW][NBI] candidate pivot rows
    cplx* qwin = cand + 2 * LA_PW * NBI;                         // [NBI][WIN] pivot rows, window columns
    __shared__ u64 slot[3];
    __shared__ int bad_sh;
    __shared__ int team_ctr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;
    int* pivrow = piv_all + (size_t)blockIdx.x * 2 * n;
    int* colof = pivrow + n;
    int team_expect = 0;
    if (tid == 0) { bad_sh = 0; team_ctr = 0; }
    if (tid < 3) slot[tid] = 0;
    __syncthreads();

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = (n + 15) >> 4;
    const int nsub = (cw + NBI - 1) / NBI;                       // sub-panels = column tiles of the window

    // apply sub-panel [k0, k0+kw) (its columns: the P operand, global; its pivot rows in the window columns:
    // qwin, LDS) to the column tiles of the window selected by tmask; row tiles widx, widx + nw, ...
    auto update_tiles = [&](int k0, int kw, unsigned tmask, int widx, int nw) __attribute__((always_inline)) {
        for (int ti = widx; ti < tiles; ti += nw) {
            const int prow = min(ti * 16 + fi, n - 1);
            cplx pa[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pa[ks] = W[(size_t)prow * n + min(k0 + ks * 4 + fk, n - 1)];   // k >= kw pairs with Q == 0
            bool keep[4];
            const cplx* crow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = min(ti * 16 + fk + 4 * r, n - 1);
                const int cf = colof[i];
                keep[r] = !(cf >= k0 && cf < k0 + kw);
                crow[r] = W + (size_t)i * n;
            }
            // the C tiles of every selected column tile are requested before the first MFMA: one L2 / Infinity-
            // Cache round trip per row tile, not one per 16 x 16 tile
            cplx cv[WT][4];
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if ((tmask >> t) & 1u) {                             // (uniform)
                    const int colc = min(c0 + t * 16 + fi, n - 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[t][r] = crow[r][colc];
                }
            }
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if ((tmask >> t) & 1u) {                             // (uniform)
                    const int col = c0 + t * 16 + fi;
                    const bool col_store = col < n && col < c0 + cw;
                    d4 accr, acci, accs = {0, 0, 0, 0};                  // 3M form, see gj_update_item
#pragma unroll
                    for (int r = 0; r < 4; ++r) { accr[r] = keep[r] ? cv[t][r].x : 0.0; acci[r] = keep[r] ? cv[t][r].x + cv[t][r].y : 0.0; }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const cplx qb = qwin[(ks * 4 + fk) * WIN + t * 16 + fi];
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x, qb.x, accr, 0, 0, 0);
                        accs = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].y, qb.y, accs, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x + pa[ks].y, qb.x + qb.y, acci, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = ti * 16 + fk + 4 * r;
                        if (i < n && col_store) W[(size_t)i * n + col] = cmake(accr[r] - accs[r], acci[r] - accr[r] - accs[r]);
                    }
                }
            }
        }
    };
    const unsigned all_tiles = (1u << nsub) - 1u;
    int sti = 0;
    auto stamp = [&](int w) __attribute__((always_inline)) {      // slot sti: wave w's clock at this point
        if (stamps && blockIdx.x == 0 && wave == w && lane == 0 && sti < 60) stamps[sti] = __builtin_amdgcn_s_memrealtime();
        ++sti;
    };
    stamp(0);

    for (int sp = 0; sp < nsub; ++sp) {
        const int k0 = c0 + sp * NBI, kw = min(NBI, c0 + cw - k0);
        __syncthreads();                 // [A] look-ahead columns (global stores) complete
        stamp(0);                        // 1 + 6 sp: [A] passed
        if (wave < LA_PW) {
            // ---- panel team: strips of the sub-panel in registers, pivot steps, strips back
            cplx a[1][S];
            bool avail[1];
            const int r = tid;
            const bool row_ok = r < n;
            avail[0] = row_ok && (colof[row_ok ? r : 0] < 0);
            {
                const cplx* g = W + (size_t)(row_ok ? r : 0) * n + k0;
#pragma unroll
                for (int s = 0; s < S; ++s) a[0][s] = (row_ok && s < kw) ? g[s] : cmake(0.0, 0.0);
            }
            PanelCtx<C> ctx{a, avail, cand, slot, &bad_sh, pivrow, colof,
                            &team_ctr, team_expect, n, k0, kw, tid, lane, wave, 0, tid, nullptr};
            publish_candidate<C>(ctx, 0, 0, 0);
            team_sync<LA_PW>(&team_ctr, team_expect, lane);
            PanelSteps<C, 0>::run(ctx);
            if (row_ok) {
                cplx* g = W + (size_t)r * n + k0;
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (s < kw) g[s] = a[0][s];
            }
        } else if (sp > 0) {
            // ---- update team: the previous sub-panel reaches the columns outside itself and this sub-panel
            update_tiles(k0 - NBI, NBI, all_tiles & ~(3u << (sp - 1)), wave - LA_PW, LA_NW - LA_PW);
        }
        stamp(0);                        // panel team done (wave 0)
        stamp(LA_PW);                    // update team done (wave 8)
        __syncthreads();                 // [B] sub-panel columns, pivrow / colof (global) visible; Q of sp-1 free
        stamp(0);                        // [B] passed
        for (int t = tid; t < NBI * WIN; t += LA_THREADS) {
            const int k = t / WIN, j = t - k * WIN;
            qwin[t] = (k < kw && c0 + j < n && j < cw) ? W[(size_t)pivrow[k0 + k] * n + c0 + j] : cmake(0.0, 0.0);
        }
        __syncthreads();                 // [C]
        stamp(0);                        // [C] passed: Q staged
        if (sp + 1 < nsub) update_tiles(k0, kw, 1u << (sp + 1), wave, LA_NW);       // look-ahead, all waves
        stamp(0);                        // look-ahead done (wave 0)
    }
    {
        // the last sub-panel reaches all other columns of the window (all twelve waves)
        const int sp = nsub - 1, k0 = c0 + sp * NBI, kw = min(NBI, c0 + cw - k0);
        update_tiles(k0, kw, all_tiles & ~(1u << sp), wave, LA_NW);
        stamp(0);                        // last update done (wave 0)
    }
    if (tid == 0 && bad_sh != 0 && info[blockIdx.x] == 0) info[blockIdx.x] = bad_sh;
}

#include "gj_strip.h"

// ---- Column-block big update.  A workgroup (256 threads, 4 waves; TWO per CU, which run their barriers and
// operand waits independently of each other) OWNS a block of 64 columns outside the window and applies the window
// to all of its rows:
//     W[i][J] = (row i pivot of the window ? 0 : W[i][J]) + P'[i][0:cw) * Q[0:cw)[J]
// * Q[:, J] -- the window's pivot rows in the block's columns -- is read IN PLACE into registers before the
//   first store (16 x n_k fragments of the wave's column tile, 64 VGPRs): only this workgroup ever touches
//   column block J, so the window kernel needs no Q snapshot (0.5 MB written and read back per matrix and
//   window at n = 500) and the matrix-core loop has no B-operand traffic at all;
// * the row blocks of 32 stream through: P'[I] (32 x cw) -> LDS (A operand), the old block C[I][J] -> the
//   accumulators; the operands of row block I+1 are requested before row block I runs its 96 MFMAs per
//   wave, so the L2 / HBM latency of the operands and of the old block is covered by matrix work;
// * launch order is XCD-aware: workgroup L runs on XCD L % 8, and the column blocks of one matrix are
//   consecutive slots of ONE XCD, so that they share P' in that XCD's L2.
// (Round 2 ran this as one 8-wave workgroup per CU with row blocks of 64: measured alone, MFMA work 2.06 ms and
// loads + stores 1.56 ms gave 2.61 ms per window of 1000 x n = 500.)
constexpr int CU_THREADS = 256;
// Pitch of the A tile in LDS (elements).  A ds_read_b128 serves 16 lanes per cycle in the groups {0-3, 12-15, 20-27}, ...
// and an A-operand fragment (lane = row fi + 16 k-index fk) then touches the 16-byte bank quads (fi * pitch + fk) mod 16:
// all distinct for pitch = 2 (mod 4) -- 66 -- while 65 lets (fi 12, fk 0) and (fi 11, fk 1) collide: one conflict
// cycle per fragment read (round 2's PMC: SQ_LDS_BANK_CONFLICT = SQ_INSTS_MFMA / 3 x ... exactly one per read).
#ifndef NEGF_CU_AP
#define NEGF_CU_AP 66
#endif
constexpr int CU_AP = NEGF_CU_AP;

// NW = 4: the lean form above.  NW = 8: one workgroup of eight waves per CU with row blocks of 64 (round 2's shape) for
// launches that do not fill the chip (fewer workgroups than CUs -- the small per-GPU batches of a sharded grid): the
// column block is then the unit of parallelism, and eight waves take it through in half the time of four.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void gj_colupdate_kernel(
    int n, int nb, cplx* __restrict__ bufA, size_t mat_stride, const int* __restrict__ piv_all, int c0, int cw,
    int only_blk /* >= 0: this column block of every matrix and no other (window pairs, see gj_colupdate2_kernel) */)
{
    constexpr int RB = 8 * NW, THREADS = NW * 64;   // rows of a row block (a wave fills eight rows and owns 32 x 16 of them)
    __shared__ cplx As[2][RB * CU_AP];            // P'[I] of the current and of the next row block
    __shared__ unsigned char pflag[8192];         // row is a pivot row of this window (its old content counts as zero)
    const int nblk = (n + 63) >> 6, jwin = c0 >> 6, per_mat = only_blk >= 0 ? 1 : nblk - 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int m = (slot / per_mat) * 8 + xcd;
    if (m >= nb) return;                                         // (uniform)
    int jb = slot % per_mat;
    if (only_blk >= 0) jb = only_blk;
    else if (jb >= jwin) ++jb;                                   // skip the window's own block
    cplx* W = bufA + (size_t)m * mat_stride;
    const int* pivrow = piv_all + (size_t)m * 2 * n;
    const int* colof = pivrow + n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 2) * 32, wc = (wave & 3) * 16;       // wave tile 32 x 16 of the RB x 64 block
    const int fi = lane & 15, fk = lane >> 4;
    const int col = jb * 64 + wc + fi;
    const bool col_ok = col < n;
    const int colc = col_ok ? col : n - 1;
    // the matrix base as an explicitly wave-uniform value: scalar base + 32-bit byte offsets in every access of the
    // pipeline, recomputed where they are used instead of carried as 64-bit pointers (n <= 8192: 16 n^2 < 2^32)
    const unsigned long long wbits = (unsigned long long)(size_t)W;
    const char* Wb = reinterpret_cast<const char*>((size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wbits >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)(wbits & 0xffffffffu))));
    char* Wbw = const_cast<char*>(Wb);
    const unsigned un = (unsigned)n;

    for (int i = tid; i < n; i += THREADS) { const int c = colof[i]; pflag[i] = (c >= c0 && c < c0 + cw) ? 1 : 0; }
    // Q fragments of this wave's column tile: B operand element (k = ks*4 + fk, col)
    cplx qf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int k = ks * 4 + fk;
        const cplx v = W[(size_t)pivrow[c0 + min(k, cw - 1)] * n + colc];
        const bool ok = (k < cw) & col_ok;
        qf[ks] = cmake(ok ? v.x : 0.0, ok ? v.y : 0.0);
    }
    // P'[I] (RB rows x 64 k) goes global -> LDS directly (global_load_lds_dwordx4: no register staging): a wave
    // fills eight rows, one instruction per row -- lane = k, 1 KB contiguous in LDS (pitch 65 stays legal: no
    // instruction crosses a row) and in global memory.  Lanes k >= cw and rows >= n re-read the last valid
    // column / row: finite values that meet zero Q rows / feed discarded output rows.
    const unsigned a_lane = (unsigned)(c0 + min(lane, cw - 1));
    auto fetch_a = [&](int ib, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wave * 8 + j;
            const unsigned byte_off = ((unsigned)min(ib * RB + row, n - 1) * un + a_lane) * 16u;
            // (asm: hipcc would drain a builtin LDS-DMA -- vmcnt(0) -- in front of the next ds_read of the OTHER
            //  buffer; the completion of these loads is awaited by the vmcnt(0) of the barrier at the loop top)
            const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(unsigned)(size_t)(__attribute__((address_space(3))) void*)&As[buf][row * CU_AP]);   // wave-uniform -> SGPR
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(byte_off), "s"(lds_dst), "s"(Wb) : "memory");
        }
    };
    cplx cv[2][4];
    auto fetch_c = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned off = ((unsigned)min(ib * RB + wr + a * 16 + fk + 4 * r, n - 1) * un + (unsigned)colc) * 16u;
                cv[a][r] = *reinterpret_cast<const cplx*>(Wb + off);
            }
    };
    fetch_a(0, 0);
    fetch_c(0);
    // Software pipeline over the row blocks (one barrier each).  At the top of iteration ib everything this
    // wave has in flight is awaited: P'[ib] and C[ib] (requested an iteration ago, in front of the matrix
    // instructions) and the stores of block ib-2 (issued an iteration ago as well) -- the results of block ib-1 are
    // still in registers and are stored only AFTER the requests for block ib+1, so that no wait ever sees a fresh store.
    d4 sr[2], si[2];                      // results of the previous row block, not stored yet
    auto store_block = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = ib * RB + wr + a * 16 + fk + 4 * r;
                const unsigned off = ((unsigned)gi * un + (unsigned)col) * 16u;
                if (gi < n && col_ok) *reinterpret_cast<cplx*>(Wbw + off) = cmake(sr[a][r], si[a][r]);
            }
    };
    const int nrb = (n + RB - 1) / RB;
#pragma unroll 1
    for (int ib = 0; ib < nrb; ++ib) {
        const int buf = ib & 1;
        d4 cr[2], ci[2], cs[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // 3M form (see gj_update_item): cr = s1, cs = s2, ci = s3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the LDS-DMA of P'[ib] (not in hipcc's bookkeeping)
        __syncthreads();                  // (vmcnt(0): the Q fragments too, before the first store to the block)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool z = pflag[min(ib * RB + wr + a * 16 + fk + 4 * r, n - 1)] != 0;
                cr[a][r] = z ? 0.0 : cv[a][r].x; ci[a][r] = z ? 0.0 : cv[a][r].x + cv[a][r].y;
            }
        const cplx* ab = &As[buf][(wr + fi) * CU_AP + fk];
        // (A fragments two k-steps ahead of their matrix instructions, held there by scheduling barriers: see gj_colupdate2_kernel)
        constexpr int CU_LA = 1;
        cplx af[CU_LA + 1][2];
#pragma unroll
        for (int l = 0; l < CU_LA; ++l)
#pragma unroll
            for (int a = 0; a < 2; ++a) af[l][a] = ab[a * 16 * CU_AP + l * 4];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cur = ks % (CU_LA + 1);
            if (ks == 2) {
                // the requests for row block ib+1 and the stores of row block ib-1 are issued HERE, behind the first
                // matrix instructions: a wave issues them in the shadow of its matrix instructions, not in front of them
                __builtin_amdgcn_sched_barrier(0);
                if (ib + 1 < nrb) { fetch_a(ib + 1, buf ^ 1); fetch_c(ib + 1); }
                if (ib > 0) store_block(ib - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + CU_LA < 16) {
#pragma unroll
                for (int a = 0; a < 2; ++a) af[(ks + CU_LA) % (CU_LA + 1)][a] = ab[a * 16 * CU_AP + (ks + CU_LA) * 4];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks * 4 < cw) {                                // (uniform: a ragged last window has fewer k-steps; its Q rows beyond are zero)
                const double qs = qf[ks].x + qf[ks].y;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    cr[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a].x, qf[ks].x, cr[a], 0, 0, 0);
                    cs[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a].y, qf[ks].y, cs[a], 0, 0, 0);
                    ci[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a].x + af[cur][a].y, qs, ci[a], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sr[a][r] = cr[a][r] - cs[a][r]; si[a][r] = ci[a][r] - cr[a][r] - cs[a][r]; }
    }
    store_block(nrb - 1);
}

// ---- Column-block update with a PAIR of windows, A = [c0, c0+64) and B = [c0+64, c0+64+cwB), in ONE pass over the
// column block: the single-window update above reads and writes every block outside the window once per window
// (1000 matrices of n = 500: 7 GB per window, as much time in loads and stores as in matrix instructions); with
// the pair fused the old block is read once, takes a rank-128 update in the accumulators and is written once.
// Window by window,
//     W' = keepA(W) + P'A QA               QA = rows pivA of W                (keepX: the pivot rows of X count as zero)
//     W''= keepB(W') + P'B QB'             QB' = rows pivB of W' = W[pivB] + P'A[pivB] QA
// and, multiplied out,
//     W''= keepAB(W) + P''A QA + P'B QB    QB = rows pivB of W (raw),   P''A = keepB(P'A) + P'B P'A[pivB]
// where P''A is nothing but block A after ITS update with window B: the pair acts as one 128-column window whose
// transform is [P''A | P'B], and both Q sets are read in place from the untouched block.  Launch sequence of a pair
// (gj_large_launch): window kernel A; single-window update of block B alone; window kernel B (its columns are up to
// date with A); single-window update with B of block A alone (-> P''A); THIS kernel on all blocks but A and B.
// (A first version kept P'A and corrected QB on the matrix cores -- the accumulator layout of the FP64 16x16x4
// instruction is the B-operand fragment of the next product -- : one eighth more matrix work at n = 500.)
// * TWO lean workgroups per CU (4 waves each, one column tile per wave): the two run their barriers and operand
//   waits independently of each other (one 8-wave workgroup per CU with row blocks of 32: N = 500 x 1000 26.8 ms
//   against 25.7, N = 1000 167 against 154);
// * both Q sets sit in registers for the whole pass (128 VGPRs); row blocks of 16 stream through (P''A[I] and P'B[I]
//   by LDS-DMA, double buffered: 4 x 16.9 KB), one 16 x 16 tile per wave, 32 k-steps, sums in 3M form;
// * addresses are 32-bit byte offsets from the wave-uniform matrix base, recomputed where they are used: with
//   both Q sets in registers there is no room for loop-carried 64-bit pointers (n <= 8192: 16 n^2 < 2^32);
// * software pipeline and XCD-aware launch order as in gj_colupdate_kernel.
__global__ __launch_bounds__(CU_THREADS, 2) void gj_colupdate2_kernel(
    int n, int nb, cplx* __restrict__ bufA, size_t mat_stride, const int* __restrict__ piv_all, int c0, int cwB)
{
    __shared__ cplx As2[2][2][16 * CU_AP];        // [buffer][window A / B] P[I], 16 rows
    __shared__ unsigned char pflag[8192];         // row is a pivot row of window A or B (its old content counts as zero)
    const int nblk = (n + 63) >> 6, jwin = c0 >> 6, per_mat = nblk - 2, cB = c0 + 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int m = (slot / per_mat) * 8 + xcd;
    if (m >= nb) return;                                         // (uniform)
    int jb = slot % per_mat;
    if (jb >= jwin) jb += 2;                                     // skip the blocks of the two windows
    cplx* W = bufA + (size_t)m * mat_stride;
    const int* pivrow = piv_all + (size_t)m * 2 * n;
    const int* colof = pivrow + n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave * 16;                                    // column tile of this wave
    const int fi = lane & 15, fk = lane >> 4;
    const int col = jb * 64 + wc + fi;
    const bool col_ok = col < n;
    const int colc = col_ok ? col : n - 1;
    // (the matrix base as an explicitly wave-uniform value: scalar base + 32-bit lane offsets in every access)
    const unsigned long long wbits = (unsigned long long)(size_t)W;
    const char* Wb = reinterpret_cast<const char*>((size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wbits >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)(wbits & 0xffffffffu))));
    char* Wbw = const_cast<char*>(Wb);
    const unsigned un = (unsigned)n;
    const unsigned a_lane = (unsigned)(c0 + lane), b_lane = (unsigned)(cB + min(lane, cwB - 1));

    for (int i = tid; i < n; i += CU_THREADS) { const int c = colof[i]; pflag[i] = (c >= c0 && c < cB + cwB) ? 1 : 0; }
    auto lds_dma_off = [&](unsigned byte_off, cplx* dst) __attribute__((always_inline)) {
        const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)(unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);     // wave-uniform -> SGPR
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(byte_off), "s"(lds_dst), "s"(Wb) : "memory");
    };
    // ---- QA, QB: the pivot rows of both windows in this wave's column tile, in place (read before the first store)
    cplx qfA[16], qfB[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int k = ks * 4 + fk;
        const cplx va = W[(size_t)pivrow[c0 + k] * n + colc];
        const cplx vb = W[(size_t)pivrow[cB + min(k, cwB - 1)] * n + colc];
        const bool okb = (k < cwB) & col_ok;
        qfA[ks] = cmake(col_ok ? va.x : 0.0, col_ok ? va.y : 0.0);
        qfB[ks] = cmake(okb ? vb.x : 0.0, okb ? vb.y : 0.0);
    }

    const int nrb = (n + 15) >> 4;
    auto fetch_a = [&](int ib, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 4 + j;
            const unsigned rbase = (unsigned)min(ib * 16 + row, n - 1) * un;
            lds_dma_off((rbase + a_lane) * 16u, &As2[buf][0][row * CU_AP]);
            lds_dma_off((rbase + b_lane) * 16u, &As2[buf][1][row * CU_AP]);
        }
    };
    cplx cv[4];
    auto fetch_c = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = ((unsigned)min(ib * 16 + fk + 4 * r, n - 1) * un + (unsigned)colc) * 16u;
            cv[r] = *reinterpret_cast<const cplx*>(Wb + off);
        }
    };
    fetch_a(0, 0);
    fetch_c(0);
    d4 sr, si;                             // results of the previous row block, not stored yet
    auto store_block = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = ib * 16 + fk + 4 * r;
            const unsigned off = ((unsigned)gi * un + (unsigned)col) * 16u;
            if (gi < n && col_ok) *reinterpret_cast<cplx*>(Wbw + off) = cmake(sr[r], si[r]);
        }
    };
#pragma unroll 1
    for (int ib = 0; ib < nrb; ++ib) {
        const int buf = ib & 1;
        d4 cr, ci, cs = {0, 0, 0, 0};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the LDS-DMA of P[ib] (not in hipcc's bookkeeping); the Q fragments
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool z = pflag[min(ib * 16 + fk + 4 * r, n - 1)] != 0;
            cr[r] = z ? 0.0 : cv[r].x; ci[r] = z ? 0.0 : cv[r].x + cv[r].y;
        }
        const cplx* abA = &As2[buf][0][fi * CU_AP + fk];
        const cplx* abB = &As2[buf][1][fi * CU_AP + fk];
        // The A fragments come from LDS CU2_LA k-steps ahead of the matrix instructions that use them, and the scheduling
        // barriers keep them there: left to itself the compiler sinks every ds_read to just in front of its first use (the
        // registers are full: both Q sets), and a wave then waits an LDS round trip per two k-steps with its matrix pipe idle
        // (round-5 ISA: "ds_read x 2; s_waitcnt lgkmcnt(1); mfma" -- 64 % of the 3M matrix peak).
        constexpr int CU2_LA = 2;
        const int ksB = (cwB + 3) >> 2;
        auto opA = [&](int ks) __attribute__((always_inline)) { return ks < 16 ? abA[ks * 4] : abB[(ks - 16) * 4]; };
        cplx af[CU2_LA + 1];
#pragma unroll
        for (int l = 0; l < CU2_LA; ++l) af[l] = opA(l);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            if (ks == 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (ib + 1 < nrb) { fetch_a(ib + 1, buf ^ 1); fetch_c(ib + 1); }
                if (ib > 0) store_block(ib - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + CU2_LA < 32) af[(ks + CU2_LA) % (CU2_LA + 1)] = opA(ks + CU2_LA);
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 16 || ks - 16 < ksB) {                  // (uniform: a ragged last window B has fewer k-steps; its Q rows beyond are zero)
                const cplx a = af[ks % (CU2_LA + 1)];
                const cplx q = ks < 16 ? qfA[ks & 15] : qfB[ks & 15];
                const double qs = q.x + q.y;
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, q.x, cr, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, q.y, cs, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, qs, ci, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { sr[r] = cr[r] - cs[r]; si[r] = ci[r] - cr[r] - cs[r]; }
    }
    store_block(nrb - 1);
}

__global__ __launch_bounds__(256) void gj_gather_kernel(int n, const cplx* __restrict__ bufA,
                                                         cplx* __restrict__ bufB, size_t mat_stride,
                                                         const int* __restrict__ piv_all,
                                                         const int* __restrict__ info)
{
    const cplx* W = bufA + (size_t)blockIdx.y * mat_stride;
    cplx* X = bufB + (size_t)blockIdx.y * mat_stride;
    const int* pivrow = piv_all + (size_t)blockIdx.y * 2 * n;
    const int* colof = pivrow + n;
    const int i = blockIdx.x;
    if (info[blockIdx.y] != 0) {         // singular / NaN matrix: NaN-filled, bookkeeping not used as an index
        const double fillv = __builtin_nan("");
        for (int j = threadIdx.x; j < n; j += 256) X[(size_t)i * n + j] = cmake(fillv, fillv);
        return;
    }
    const cplx* srow = W + (size_t)pivrow[i] * n;
    for (int j = threadIdx.x; j < n; j += 256) X[(size_t)i * n + j] = srow[colof[j]];
}

__global__ void gj_state_init_kernel(int n, int* __restrict__ piv_all, int* __restrict__ info)
{
    int* p = piv_all + (size_t)blockIdx.x * 2 * n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) { p[t] = 0; p[n + t] = -1; }
    if (threadIdx.x == 0) info[blockIdx.x] = 0;
}

double g_gj_vector_flops = 0;            // 3M-equivalent flops of the last launch that ran as VECTOR work (strip window kernels)

template <int NBI, int RPT>
void gj_large_launch(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* piv, int* info, GjSideStreams* sdp, int win_mode, bool skip_gather)
{
    const size_t smem = (size_t)(2 * PW * NBI + 2 * NBI * WIN) * sizeof(cplx);      // candidate rows + Q of two sub-panels
    auto kern = gj_window_kernel<NBI, RPT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(100 * 1024));
        attr_set = true;
    }
    static int want_stamps = -1;
    static unsigned long long* d_stamps = nullptr;
    if (want_stamps < 0) {
        want_stamps = getenv("NEGF_GJ_STAMPS") ? 1 : 0;
        if (want_stamps) { (void)hipMalloc(&d_stamps, 64 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 64 * sizeof(unsigned long long)); }
    }
    static int winla = -1;
    if (winla < 0) { const char* e = getenv("NEGF_GJ_WINLA"); winla = e ? atoi(e) : 1; }
    // Which window kernel (win_mode 0 = auto, 1 = strip wherever an instantiation serves n, 2 = team kernels only;
    // NEGF_GJ_STRIP = 0 / 1 sets the auto rule's answer for A/B runs).  Measured on MI355X, inverse ms, strip / team:
    //   1000 matrices: n = 256 4.01 / 5.26 (single-workgroup kernel), 300 6.62 / 7.53, 400 13.0 / 14.6, 500 21.5 / 24.5 (the
    //   strip kernel moves 3.3 MB per window and matrix instead of 4.9, and both run at the ~4.3 TB/s this access pattern
    //   gets out of the memory system), 650 55.5 / 53.5, 800 84.8 / 84.4, 1000 147.9 / 150.4 (the 8-wave, one-per-CU form);
    //   small batches (stream groups of m / 4): 32 x n = 500 2.29 / 2.20, 128 x 500 4.10 / 3.97, 250 x 500 6.20 / 6.12 (a lean
    //   4-wave workgroup has a longer chain per matrix than the 12-wave look-ahead kernel), 12 x 800 5.51 / 5.77,
    //   61 x 800 7.54 / 8.10, 64 x 1000 13.1 / 14.1, 128 x 1000 23.0 / 23.9, 486 x 800 42.5 / 42.4.
    static int strip_env = -2;
    if (strip_env == -2) { const char* e = getenv("NEGF_GJ_STRIP"); strip_env = e ? atoi(e) : -1; }
    static int strip_min = -1;           // matrices per stream group from which the auto rule takes the strip kernel (257 <= n <= 512)
    if (strip_min < 0) { const char* e = getenv("NEGF_GJ_STRIP_MIN"); strip_min = e ? atoi(e) : 64; }
    static int strip_cfg = -1;           // 0: the measured choice per size; 1: n <= 512 fat (8 waves x 1 row, sub-windows of 32); 3: n <= 256 with sub-windows of 16; 4: n <= 512 lean with two chunks in flight
    if (strip_cfg < 0) { const char* e = getenv("NEGF_GJ_STRIP_CFG"); strip_cfg = e ? atoi(e) : 0; }
    static int strip_dbg = -1;           // timing ablations of the strip kernel (wrong results)
    if (strip_dbg < 0) { const char* e = getenv("NEGF_GJ_STRIP_DBG"); strip_dbg = e ? atoi(e) : 0; }
    static int pair = -1;                // windows in pairs (one pass over the other column blocks per pair); 0: one by one
    if (pair < 0) { const char* e = getenv("NEGF_GJ_PAIR"); pair = e ? atoi(e) : 1; }
    static long pair_min = -1, fat_max = -1, fat_total_max = -1;
    // (round 5, with the strip window kernels, inverse ms without / with pairs: 122 x N = 800 (330 workgroups per group) 13.46 /
    //  12.09, 128 x N = 500 (192) 3.93 / 3.94, 61 x N = 800 (165) 7.39 / 8.95: the threshold moved from 352 to 256)
    if (pair_min < 0) { const char* e = getenv("NEGF_GJ_PAIR_MIN"); pair_min = e ? atol(e) : 256; }
    if (fat_max < 0) { const char* e = getenv("NEGF_GJ_FAT_MAX"); fat_max = e ? atol(e) : 256; }
    if (fat_total_max < 0) { const char* e = getenv("NEGF_GJ_FAT_TOTAL_MAX"); fat_total_max = e ? atol(e) : 2000; }
    const int nblk = (n + 63) / 64;
    // the chain of one group of matrices: per window the panel kernel (one workgroup per matrix: a latency chain
    // that covers at most `count` CUs) and the column-block update (throughput-bound), then the gather
    auto chain = [&](hipStream_t s, int first, int count) {
        cplx* Ag = A + (size_t)first * stride; cplx* Bg = B + (size_t)first * stride;
        int* pg = piv + (size_t)first * 2 * n; int* ig = info + first;
        hipLaunchKernelGGL(gj_state_init_kernel, dim3(count), dim3(256), 0, s, n, pg, ig);
        auto window = [&](int c0, int cw) {
            unsigned long long* stp = (c0 == WIN && first == 0) ? d_stamps : (unsigned long long*)nullptr;
            // (matrix-core flop accounting: with the strip kernel the window's own 6 n cw^2 are vector work)
            const bool strip = NBI == 16 && (win_mode == 1 || (win_mode == 0 && (strip_env >= 0 ? strip_env != 0 : (n <= 256 || (n <= 512 ? count >= strip_min : (count <= 160 || n >= 900))))));
            if (strip) {
                g_gj_vector_flops += 6.0 * (double)((n + 15) & ~15) * cw * (double)cw * count;
                if (n <= 256 && strip_cfg != 3)
                    hipLaunchKernelGGL((gj_window_strip_kernel<1, 32, 4, 2, 4>), dim3(count), dim3(256), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                else if (n <= 256)
                    hipLaunchKernelGGL((gj_window_strip_kernel<1, 16, 4, 3, 4>), dim3(count), dim3(256), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                else if (n <= 512 && strip_cfg == 1)
                    hipLaunchKernelGGL((gj_window_strip_kernel<1, 32, 8, 1, 4>), dim3(count), dim3(512), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                else if (n <= 512 && strip_cfg == 4)     // two chunks in flight: 72 spilled registers, 23.3 ms against 22.6 (1000 x n = 500)
                    hipLaunchKernelGGL((gj_window_strip_kernel<2, 16, 4, 2, 2>), dim3(count), dim3(256), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                else if (n <= 512)
                    hipLaunchKernelGGL((gj_window_strip_kernel<2, 16, 4, 2, 1>), dim3(count), dim3(256), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                else
                    hipLaunchKernelGGL((gj_window_strip_kernel<2, 16, 8, 1, 2>), dim3(count), dim3(512), 0, s, n, Ag, Bg, stride, pg, ig, c0, cw, stp, strip_dbg);
                return;
            }
            if (NBI == 16 && RPT == 1 && winla && n <= 256)
                hipLaunchKernelGGL((gj_window_la_kernel<16, 4, 6>), dim3(count), dim3(6 * 64), smem, s, n, Ag, stride, pg, ig, c0, cw, stp);
            else if (NBI == 16 && RPT == 1 && winla)
                hipLaunchKernelGGL((gj_window_la_kernel<16, 8, 12>), dim3(count), dim3(12 * 64), smem, s, n, Ag, stride, pg, ig, c0, cw, stp);
            else
                hipLaunchKernelGGL(kern, dim3(count), dim3(PT), smem, s, n, Ag, stride, pg, ig, c0, cw, stp);
        };
        auto colupdate = [&](int c0, int cw, int only_blk) {
            const int blocks = only_blk >= 0 ? 1 : nblk - 1;
            const dim3 grid(8 * ((count + 7) / 8) * blocks);
            // eight waves per column block where the launch has fewer workgroups than the chip has CUs AND the whole
            // batch (all stream groups together) leaves the chip room: there a column block's latency counts, not the
            // CU's throughput (MI355X, inverse ms, lean / fat: 61 x N = 800 8.87 / 7.57, 64 x N = 1000 14.40 / 13.83,
            // 250 x N = 500 6.81 / 6.49 (single-block launches of the pairs); but 1000 x N = 500 25.4 / 29.8: a fat
            // workgroup takes a CU's whole LDS and shuts the other groups' kernels out)
            if ((long)count * blocks > fat_max || (long)nb * nblk > fat_total_max)
                hipLaunchKernelGGL(gj_colupdate_kernel<4>, grid, dim3(256), 0, s, n, count, Ag, stride, (const int*)pg, c0, cw, only_blk);
            else
                hipLaunchKernelGGL(gj_colupdate_kernel<8>, grid, dim3(512), 0, s, n, count, Ag, stride, (const int*)pg, c0, cw, only_blk);
        };
        // Pairs pay where the update is throughput-bound (they halve its traffic); a small group of matrices is a
        // latency chain of kernels, to which a pair adds two single-block launches: measured cross-over (MI355X,
        // workgroups of the fused kernel per group) 61 x N = 800 in four groups (165: 8.87 ms one by one, 9.25 in
        // pairs), 250 x N = 500 (372: 6.91 / 6.81), 128 x N = 1000 (448: 24.0 / 22.95)
        const bool use_pairs = pair && nblk >= 3 && (long)count * (nblk - 2) >= pair_min;
        for (int c0 = 0; c0 < n; c0 += WIN) {
            const int cw = min(WIN, n - c0);
            if (use_pairs && c0 + WIN < n) {
                // windows A = [c0, c0 + 64) and B = the next one as a pair: every block outside the two is read and
                // written once for both (gj_colupdate2_kernel)
                const int cB = c0 + WIN, cwB = min(WIN, n - cB), jA = c0 / WIN;
                window(c0, cw);
                colupdate(c0, cw, jA + 1);                        // A -> block B
                window(cB, cwB);
                colupdate(cB, cwB, jA);                           // B -> block A: the pair's transform is now [P''A | P'B]
                hipLaunchKernelGGL(gj_colupdate2_kernel, dim3(8 * ((count + 7) / 8) * (nblk - 2)), dim3(CU_THREADS), 0, s,
                                   n, count, Ag, stride, (const int*)pg, c0, cwB);
                c0 += WIN;
                continue;
            }
            window(c0, cw);
            if (nblk > 1) colupdate(c0, cw, -1);
        }
        // (skip_gather: the caller resolves G[i][j] = W[pivrow[i]][colof[j]] itself -- the weighted sum of GrInt reads the
        //  reduced matrices through the permutation, launch_accumulate_perm, and G is never written)
        if (!skip_gather)
            hipLaunchKernelGGL(gj_gather_kernel, dim3(n, count), dim3(256), 0, s, n, (const cplx*)Ag, Bg, stride,
                               (const int*)pg, (const int*)ig);
    };
    // Small batches -- what the energy grid of BASELINE's multi-GPU configurations leaves one GPU: C4 sharded 8
    // ways is 61 matrices of N = 800, C5 128 of N = 1000 -- cannot cover the chip with one panel workgroup per
    // matrix, and window k+1 waits for the column update of window k.  The batch is then cut into up to four
    // groups on streams of their own: while one group factors a panel (a few CUs), the column updates of the
    // others use the rest of the chip: 61 x N = 800 24.5 -> 30.5 TF, 128 x N = 1000 34.9 -> 40.0 TF.  At full batch both
    // kernels fill the chip by themselves and the groups mostly run in lockstep: + 2 % (1000 x N = 500 33.1 -> 33.8 TF).
    static int split_max = -1;
    if (split_max < 0) { const char* e = getenv("NEGF_GJ_SPLIT_MAX"); split_max = e ? atoi(e) : 1 << 30; }
    static int grp_max = -1, grp_min = -1;          // NEGF_GJ_GROUP_MAX (<= GjSideStreams::MAXG), NEGF_GJ_GROUP_MIN: matrices per group
    if (grp_max < 0) { const char* e = getenv("NEGF_GJ_GROUP_MAX"); grp_max = e ? std::min(std::max(atoi(e), 1), (int)GjSideStreams::MAXG) : 4; }
    if (grp_min < 0) { const char* e = getenv("NEGF_GJ_GROUP_MIN"); grp_min = e ? std::max(atoi(e), 1) : 12; }
    const int groups = (nb <= split_max && !d_stamps) ? std::max(1, std::min(grp_max, nb / grp_min)) : 1;
    if (groups == 1) {
        chain(st, 0, nb);
    } else {
        // side streams and fork / join events belong to the CONTEXT (created on its device at first use, destroyed with
        // it): two contexts never record the same events
        if (!sdp) { chain(st, 0, nb); return; }
        GjSideStreams& sd = *sdp;
        if (!sd.ok) {
            bool good = hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess;
            for (int g = 0; g < GjSideStreams::MAXG - 1; ++g) {
                good = good && hipStreamCreateWithFlags(&sd.s[g], hipStreamNonBlocking) == hipSuccess;
                good = good && hipEventCreateWithFlags(&sd.join[g], hipEventDisableTiming) == hipSuccess;
            }
            if (!good) { (void)hipGetLastError(); chain(st, 0, nb); return; }
            sd.ok = true;
        }
        hipStream_t* side = sd.s;
        hipEvent_t ev_fork = sd.fork;
        hipEvent_t* ev_join = sd.join;
        (void)hipEventRecord(ev_fork, st);
        const int per = (nb + groups - 1) / groups;
        for (int g = 1; g < groups; ++g) {
            const int first = g * per, count = std::min(per, nb - first);
            if (count <= 0) continue;
            (void)hipStreamWaitEvent(side[g - 1], ev_fork, 0);
            chain(side[g - 1], first, count);
            (void)hipEventRecord(ev_join[g - 1], side[g - 1]);
        }
        chain(st, 0, std::min(per, nb));
        for (int g = 1; g < groups; ++g)
            if (g * per < nb) (void)hipStreamWaitEvent(st, ev_join[g - 1], 0);
    }
    if (d_stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[64];
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        auto us = [&](int i) { return (double)(h[i] - h[0]) / 100.0; };
        if (NBI == 16 && (win_mode == 1 || (win_mode == 0 && strip_env != 0))) {
            fprintf(stderr, "[gj strip stamps] n=%d window 1, workgroup 0 (us since the window's start):", n);
            for (int sp = 0; sp < 4; ++sp)
                fprintf(stderr, " | sub %d: top %.1f loaded %.1f forward %.1f factored %.1f stored %.1f staged %.1f backward %.1f", sp,
                        us(8 * sp), us(8 * sp + 1), us(8 * sp + 2), us(8 * sp + 3), us(8 * sp + 4), sp ? us(8 * sp + 5) : 0.0, us(8 * sp + 6));
            fprintf(stderr, "\n[gj strip stamps] sub-window 1, column 4 (shader cycles from the arrival at the barrier): barrier passed %lld, winner known %lld, reciprocal %lld, strip updated %lld, published %lld; next arrival %lld\n",
                    (long long)(h[41] - h[40]), (long long)(h[42] - h[40]), (long long)(h[43] - h[40]), (long long)(h[44] - h[40]), (long long)(h[45] - h[40]), (long long)(h[45] - h[40]));
            return;
        }
        if (NBI == 16 && RPT == 1 && winla) {
            fprintf(stderr, "[gj window-la stamps] n=%d window 1, workgroup 0 (us):", n);
            for (int sp = 0; sp < 4; ++sp)
                fprintf(stderr, " | sub %d: A %.1f panel %.1f upd-team %.1f B %.1f C %.1f look-ahead %.1f", sp, us(1 + 6 * sp), us(2 + 6 * sp),
                        us(3 + 6 * sp), us(4 + 6 * sp), us(5 + 6 * sp), us(6 + 6 * sp));
            fprintf(stderr, " | last update %.1f\n", us(25));
            return;
        }
        fprintf(stderr, "[gj window stamps] n=%d window 1, workgroup 0 (us):", n);
        const int nsub = (WIN + NBI - 1) / NBI;
        for (int sp = 0; sp < nsub && 1 + 5 * sp + 4 < 60; ++sp)
            fprintf(stderr, " | sub %d: start %.1f loaded %.1f pivots %.1f stored %.1f q %.1f", sp, us(1 + 5 * sp), us(2 + 5 * sp),
                    us(3 + 5 * sp), us(4 + 5 * sp), us(5 + 5 * sp));
        if (1 + 5 * nsub + 2 < 60)
            fprintf(stderr, " | upd(t0) %.1f upd(all) %.1f\n", us(1 + 5 * nsub), us(2 + 5 * nsub));
    }
}

int gj_large_pick(int n)
{
    if (n < 64) return 0;
    if (n <= PT) return 4;               // <= 512: sub-panel 16, 1 row per thread (only above NEGF_GJ_LARGE_MIN)
    if (n <= PT * 2) return 1;           // <= 1024: sub-panel 16, 2 rows per thread
    if (n <= PT * 4) return 2;           // <= 2048: sub-panel 8, 4 rows per thread
    if (n <= PT * 8) return 3;           // <= 4096: sub-panel 4, 8 rows per thread
    if (n <= PT * 16) return 5;          // <= 8192: sub-panel 2, 16 rows per thread
    return 0;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0 || gj_large_pick(n) != 0; }

// In-place reduction of A with B as scratch; the inverses are gathered into B.
// piv: [nb][2][n] ints of pivot bookkeeping (used by the large-matrix path).
// Returns true: the result is in B.
double inverse_blocked_vector_flops() { return g_gj_vector_flops; }

bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* piv, int* info, GjSideStreams* side, int win_mode,
                            bool* skip_gather)
{
    g_gj_vector_flops = 0;
    // single-workgroup kernel up to 256 rows (its 32-column panel configuration); above, the windowed
    // path with 16-column sub-panels is faster (measured on MI355X, ms per 1000 matrices: n = 260 9.3 vs
    // 9.1, 300 13.3 vs 11.1, 340 20.5 vs 17.3, 370 28.5 vs 20.1); NEGF_GJ_LARGE_MIN moves the switch-over
    // (round 5: with the strip window kernel the windowed path wins from n = 209 on -- ms per 1000 matrices, windowed /
    // single workgroup: n = 192 2.45 / 2.58 but 200 3.01 / 3.01 and 208 3.08 / 3.08 (a ragged fourth window), 224 3.33 / 3.56,
    // 240 3.70 / 4.49, 256 4.01 / 5.26)
    static int large_min = -1;
    if (large_min < 0) { const char* e = getenv("NEGF_GJ_LARGE_MIN"); large_min = e ? atoi(e) : 209; }
    // win_mode 1 (tests, A/B) takes the windowed path with the strip kernel wherever it exists (64 <= n), win_mode 2 the
    // pre-strip configuration (single workgroup up to 256, team window kernels above)
    // (below large_min the single-workgroup kernel keeps the batches that do not fill the chip -- one matrix per CU costs what a
    // single matrix does: 108 x n = 200 0.65 ms against 0.94 windowed -- and the windowed path takes the rest, ms per 1000
    // matrices windowed / single workgroup: n = 100 0.87 / 0.93, 128 1.09 / 1.29, 160 1.60 / 1.83, 192 2.21 / 2.60,
    // 200 2.67 / 2.97 (C2), 208 2.66 / 3.09; 324 x n = 200 1.26 / 1.36)
    static int small_win_min = -1;       // matrices from which 100 <= n < large_min goes windowed
    if (small_win_min < 0) { const char* e = getenv("NEGF_GJ_SMALL_WIN_MIN"); small_win_min = e ? atoi(e) : 257; }
    const bool single_wg = win_mode == 1 ? n < 64 : win_mode == 2 ? n <= 256 : (n < large_min && !(n >= 100 && nb >= small_win_min));
    const bool defer = skip_gather && *skip_gather;               // only the windowed path can leave the gather to the caller
    if (skip_gather) *skip_gather = false;
    if (single_wg && gj_pick(n) == 1) { gj_launch<CfgSplit>(st, n, nb, A, B, stride, info); return true; }
    if (defer && gj_large_pick(n) != 0) *skip_gather = true;
    // sub-panels of 16 columns up to n = 1024 (measured on MI355X, 1000 matrices: n = 500 46.2 -> 41.1 ms,
    // n = 1000 296 -> 257 ms against sub-panels of 8: half as many passes over the 64-column window)
    switch (gj_large_pick(n)) {
    case 1: gj_large_launch<16, 2>(st, n, nb, A, B, stride, piv, info, side, win_mode, defer); return true;
    case 2: gj_large_launch<8, 4>(st, n, nb, A, B, stride, piv, info, side, win_mode, defer); return true;
    case 3: gj_large_launch<4, 8>(st, n, nb, A, B, stride, piv, info, side, win_mode, defer); return true;
    case 4: gj_large_launch<16, 1>(st, n, nb, A, B, stride, piv, info, side, win_mode, defer); return true;
    case 5: gj_large_launch<2, 16>(st, n, nb, A, B, stride, piv, info, side, win_mode, defer); return true;
    default: return false;
    }
}
