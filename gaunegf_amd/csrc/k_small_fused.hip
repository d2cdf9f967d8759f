// Small systems (n <= 96): assemble, invert and accumulate in ONE kernel, the matrix never leaves the CU.
//   A = E S - H - Sigma(E)                  _gr_matrix_ops, gauNEGF/integrate.py:67-71
//   G = inv(A)   (partial pivoting, izamax)  utils.inv, gauNEGF/utils.py:52-54
//   out = sum_m w_m G(E_m)                   _GInt, gauNEGF/integrate.py:84-142
//
// Why a kernel of its own.  One SCF density step is ~10^2 integrals of 2 ... 324 energy points (adaptive ANT levels of
// the contour, density.py:211-273, inside a Fermi search, scfE.py:301-462) and the reference's own demo system has 60
// basis functions: at those sizes the three-kernel sequence through HBM (assemble_kernel -> gj_blocked_kernel ->
// accumulate_*) is launch- and latency-bound (measured: 115 us of kernel time per call at n = 60, 14.7 k points per
// step).  A 96 x 96 complex128 matrix is 147 KB: it fits the 512 KB register file of a CU five times over and, for the
// un-permuted inverse, its 160 KB LDS once.
//
// One workgroup (256 threads = 4 waves, one per SIMD) per energy point, grid-stride over the points of the integral:
//   * the matrix lives in REGISTERS: thread (ty, tx) of a 16 x 16 thread grid owns the T x T elements
//     (ty + 16 a, tx + 16 b), T = ceil(n / 16) (a template parameter: every register index is static);
//   * Gauss-Jordan with implicit partial pivoting, one column per step, two workgroup barriers per step:
//       1. the 16 threads owning column k publish it to LDS                                   -- barrier
//       2. every wave finds the pivot row p by itself (|re|+|im| of the unused rows, exact comparison on the 64-bit
//          pattern, the lowest row among equals: LAPACK's izamax rule) -- no broadcast needed --, the 16 threads
//          owning row p publish it                                                            -- barrier
//       3. every thread: rs_j = row_p[j] / pivot, f_i = column_k[i];  a_ij -= f_i rs_j  (T^2 complex FMAs), with
//          column k replaced by the unit vector first (a_ik -> -f_i / pivot, a_pk -> 1 / pivot) and row p by rs;
//     rows are never moved: inv[i][j] = W[pivrow[i]][colof[j]], resolved when the result is written to LDS;
//   * mode ACCUMULATE: acc += w_m G over the workgroup's energies (same cfma as accumulate_*_kernel) in the workgroup's
//     partial-sum record, the records summed in workgroup order by small_reduce_kernel: bitwise reproducible;
//     mode STORE: G(E_m) to HBM for the entry points that need it (G Gamma G^H, transmission, DOS, negf_gr_batch).
// Exactly singular / NaN columns: info[m] = 1-based column, the point's G is NaN-filled (as the blocked kernels do).
#include "negf_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int SF_THREADS = 256;
constexpr int SF_MAXN = 96;

// wave maximum of a 32-bit key, wave-uniform result: DPP steps inside the rows of 16 lanes, row_bcast across them
// (zero is the identity: bound_ctrl reads zero for lanes a step does not reach), the result from lane 63
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned sf_dpp_max_u32(unsigned k)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, CTRL, ROW_MASK, 0xF, true);
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned sf_wave_max_u32(unsigned k)
{
    k = sf_dpp_max_u32<0xB1>(k);            // quad_perm [1,0,3,2]
    k = sf_dpp_max_u32<0x4E>(k);            // quad_perm [2,3,0,1]
    k = sf_dpp_max_u32<0x141>(k);           // row_half_mirror
    k = sf_dpp_max_u32<0x140>(k);           // row_mirror
    k = sf_dpp_max_u32<0x142, 0xA>(k);      // row_bcast:15
    k = sf_dpp_max_u32<0x143, 0xC>(k);      // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

// Hide a loop-invariant value from the optimiser: LLVM otherwise hoists the T^2 element addresses of every phase
// (S, H, the partial-sum record, the LDS tile) out of the energy loop and keeps them alive across the pivot loop --
// twice the registers of the matrix tile itself (the chain kernel's rs_opaque, for the same reason).
template <class V>
__device__ __forceinline__ V sf_opaque(V v)
{
    asm volatile("" : "+v"(v));
    return v;
}

template <int T>
__global__ __launch_bounds__(SF_THREADS) void small_fused_kernel(SmallFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    cplx* Gs = reinterpret_cast<cplx*>(sf_smem);                  // [n][gp]: the un-permuted inverse
    __shared__ cplx colb[2][SF_MAXN], rowb[2][SF_MAXN];           // pivot column / pivot row, double buffered by step parity
    __shared__ int pivrow_s[SF_MAXN], colof_s[SF_MAXN];

    const int n = a.n, gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tx = tid & 15, ty = tid >> 4;
    const unsigned long long m0 = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long m1 = n <= 64 ? 0ull : ((1ull << (n - 64)) - 1ull);

    // the workgroup's partial sum lives in its record of a.partial (HBM, L2-resident: every thread re-reads only what it
    // wrote itself), not in registers: a T = 6 accumulator tile would be 144 more VGPRs, and most workgroups of an
    // SCF-sized integral see one energy
    cplx* part = a.Gout ? nullptr : a.partial + (size_t)blockIdx.x * n * n;
    bool first = true;

    for (int e = blockIdx.x; e < a.m; e += gridDim.x) {
        // ---- assemble: (E S - H) - Sigma in the reference's order (integrate.py:70); contact blocks one contact
        // after the other, as scatter_sub_kernel subtracts them
        const cplx z = a.E[e];
        cplx W[T][T];
        const int tya = sf_opaque(ty), txa = sf_opaque(tx);
#pragma unroll
        for (int r = 0; r < T; ++r)
#pragma unroll
            for (int c = 0; c < T; ++c) {
                const int i = tya + 16 * r, j = txa + 16 * c;
                cplx v = cmake(0.0, 0.0);
                if (i < n && j < n) {
                    const int o = i * n + j;
                    const cplx s = a.S[o], h = a.H[o];
                    v = cmake(z.x * s.x - z.y * s.y - h.x, z.x * s.y + z.y * s.x - h.y);
                    if (a.sig_dense) v = csub(v, a.sig_dense[(size_t)e * a.sig_stride + o]);
                    for (int ct = 0; ct < a.n_contacts; ++ct) {
                        const int pi = a.pos[ct * n + i], pj = a.pos[ct * n + j];
                        if (pi >= 0 && pj >= 0)
                            v = csub(v, a.blk[(size_t)e * a.blk_stride + a.blk_off[ct] + pi * a.nc[ct] + pj]);
                    }
                }
                W[r][c] = v;
                if (c == T - 1) __builtin_amdgcn_sched_barrier(0);     // one tile row's loads in flight, not all T^2
            }

        // ---- Gauss-Jordan, implicit partial pivoting.  Everything that touches W is either a wave-uniform branch (on
        // the tile index of column k / row p) or a per-lane SELECT: no divergent region modifies the register tile
        unsigned long long used0 = 0ull, used1 = 0ull;            // rows that have been pivots (wave-uniform)
        int bad = 0;
        for (int k = 0; k < n; ++k) {
            const int kb = k >> 4, kx = k & 15, buf = k & 1;
            {
                cplx cv[T];
#pragma unroll
                for (int c = 0; c < T; ++c)
                    if (c == kb) {                                  // (uniform)
#pragma unroll
                        for (int r = 0; r < T; ++r) cv[r] = W[r][c];
                    }
                if (tx == kx) {
#pragma unroll
                    for (int r = 0; r < T; ++r) colb[buf][ty + 16 * r] = cv[r];
                }
            }
            __syncthreads();
            // pivot: every wave by itself.  |re| + |im| >= 0 orders like its bit pattern: wave maximum of the high words,
            // then of the low words among the lanes that hold it; the lowest ROW with that value wins (izamax)
            unsigned h0 = 0u, l0 = 0u, h1 = 0u, l1 = 0u;
            {
                const int i1 = lane + 64;
                if (lane < n && !((used0 >> lane) & 1ull)) {
                    const double v = cabs1(colb[buf][lane]);
                    if (v > 0.0) { h0 = (unsigned)__double2hiint(v); l0 = (unsigned)__double2loint(v); }
                }
                if (i1 < n && !((used1 >> lane) & 1ull)) {
                    const double v = cabs1(colb[buf][i1]);
                    if (v > 0.0) { h1 = (unsigned)__double2hiint(v); l1 = (unsigned)__double2loint(v); }
                }
            }
            const unsigned hm = sf_wave_max_u32(h0 > h1 ? h0 : h1);
            const unsigned lm = sf_wave_max_u32((h0 == hm ? l0 : 0u) > (h1 == hm ? l1 : 0u) ? (h0 == hm ? l0 : 0u) : (h1 == hm ? l1 : 0u));
            int p;
            if (hm != 0u || lm != 0u) {
                const unsigned long long b0 = __ballot(h0 == hm && l0 == lm), b1 = __ballot(h1 == hm && l1 == lm);
                p = b0 ? (int)__ffsll((long long)b0) - 1 : 64 + (int)__ffsll((long long)b1) - 1;
            } else {                                              // exactly singular or NaN column: lowest unused row
                if (!bad) bad = k + 1;
                const unsigned long long f0 = ~used0 & m0, f1 = ~used1 & m1;
                p = f0 ? (int)__ffsll((long long)f0) - 1 : 64 + (int)__ffsll((long long)f1) - 1;
            }
            p = __builtin_amdgcn_readfirstlane(p);
            if (p < 64) used0 |= 1ull << p; else used1 |= 1ull << (p - 64);
            const int pa = p >> 4, py = p & 15;
            {
                cplx rv[T];
#pragma unroll
                for (int r = 0; r < T; ++r)
                    if (r == pa) {                                  // (uniform)
#pragma unroll
                        for (int c = 0; c < T; ++c) rv[c] = W[r][c];
                    }
                if (ty == py) {
#pragma unroll
                    for (int c = 0; c < T; ++c) rowb[buf][tx + 16 * c] = rv[c];
                }
            }
            if (tid == 0) { pivrow_s[k] = p; colof_s[p] = k; }
            __syncthreads();
            // update
            // 1 / pivot = conj(pivot) / |pivot|^2, the reciprocal by v_rcp_f64 and two Newton steps (the IEEE division
            // sequence is ~60 dependent instructions on every thread's critical path; the blocked kernels divide by
            // |pivot|^2 as well)
            const cplx pv = rowb[buf][k];
            const double pd = pv.x * pv.x + pv.y * pv.y;
            double sc = __builtin_amdgcn_rcp(pd);
            sc = fma(sc, fma(-pd, sc, 1.0), sc);
            sc = fma(sc, fma(-pd, sc, 1.0), sc);
            const cplx ip = cmake(pv.x * sc, -pv.y * sc);
            cplx f[T], rs[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                const int i = ty + 16 * r;
                const cplx cv = colb[buf][i];
                f[r] = cmake(i == p ? 0.0 : cv.x, i == p ? 0.0 : cv.y);
            }
#pragma unroll
            for (int c = 0; c < T; ++c) {
                const int j = tx + 16 * c;
                const cplx rv = cmul(rowb[buf][j], ip);
                rs[c] = cmake(j == k ? ip.x : rv.x, j == k ? ip.y : rv.y);
            }
            const bool in_col = tx == kx, in_row = ty == py;
#pragma unroll
            for (int c = 0; c < T; ++c)
                if (c == kb) {                                      // column k becomes the unit vector e_p first
#pragma unroll
                    for (int r = 0; r < T; ++r) W[r][c] = cmake(in_col ? 0.0 : W[r][c].x, in_col ? 0.0 : W[r][c].y);
                }
#pragma unroll
            for (int r = 0; r < T; ++r)
#pragma unroll
                for (int c = 0; c < T; ++c) W[r][c] = cfnma(W[r][c], f[r], rs[c]);
#pragma unroll
            for (int r = 0; r < T; ++r)
                if (r == pa) {                                      // row p becomes the scaled pivot row
#pragma unroll
                    for (int c = 0; c < T; ++c) W[r][c] = cmake(in_row ? rs[c].x : W[r][c].x, in_row ? rs[c].y : W[r][c].y);
                }
        }
        // ---- un-permute through LDS: W[r][c] is G[colof[r]][pivrow[c]]
        const double qnan = __builtin_nan("");
        const int tyu = sf_opaque(ty), txu = sf_opaque(tx);
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = tyu + 16 * r;
            if (i < n) {
                const int gi = colof_s[i];
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    const int j = txu + 16 * c;
                    if (j < n) Gs[gi * gp + pivrow_s[j]] = bad ? cmake(qnan, qnan) : W[r][c];
                }
            }
        }
        if (tid == 0 && a.info) a.info[e] = bad;
        __syncthreads();
        const int tyw = sf_opaque(ty), txw = sf_opaque(tx);
        if (a.Gout) {
            cplx* out = a.Gout + (size_t)e * a.g_stride;
#pragma unroll
            for (int r = 0; r < T; ++r)
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    const int i = tyw + 16 * r, j = txw + 16 * c;
                    if (i < n && j < n) out[i * n + j] = Gs[i * gp + j];
                    if (c == T - 1) __builtin_amdgcn_sched_barrier(0);
                }
        } else {
            const cplx w = a.w[e];
#pragma unroll
            for (int r = 0; r < T; ++r)
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    const int i = tyw + 16 * r, j = txw + 16 * c;
                    if (i < n && j < n) {
                        const cplx old = first ? cmake(0.0, 0.0) : part[i * n + j];
                        part[i * n + j] = cfma(old, w, Gs[i * gp + j]);      // acc += w G, as accumulate_partial_kernel
                    }
                    if (c == T - 1) __builtin_amdgcn_sched_barrier(0);
                }
            first = false;
        }
        __syncthreads();                                          // Gs, the pivot tables and the buffers are reused
    }
}

// pivot search of the columns-per-wave layouts, by the wave that holds column k (lane = row, RPL rows per lane).
// |re| + |im| >= 0 orders like its bit pattern: wave maximum of the high words, then of the low words among the lanes that
// hold it; the lowest row wins (LAPACK's izamax rule).  An exactly singular or NaN column takes the lowest unused row and
// reports k + 1.  The result is wave-uniform.
template <int RPL>
__device__ __forceinline__ int cw_pivot_search(const cplx (&col)[RPL], const bool (&used)[RPL], int lane, int n, int k, int* bad_s)
{
    unsigned h_[RPL], l_[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const double v = cabs1(col[r]);
        const bool ok = lane + 64 * r < n && !used[r] && v > 0.0;
        h_[r] = ok ? (unsigned)__double2hiint(v) : 0u;
        l_[r] = ok ? (unsigned)__double2loint(v) : 0u;
    }
    unsigned hmax = h_[0];
    if (RPL == 2) hmax = h_[RPL - 1] > hmax ? h_[RPL - 1] : hmax;
    const unsigned hm = sf_wave_max_u32(hmax);
    unsigned lcand = h_[0] == hm ? l_[0] : 0u;
    if (RPL == 2) { const unsigned l1 = h_[RPL - 1] == hm ? l_[RPL - 1] : 0u; lcand = l1 > lcand ? l1 : lcand; }
    const unsigned lm = sf_wave_max_u32(lcand);
    int p;
    if (hm != 0u || lm != 0u) {
        const unsigned long long b0 = __ballot(h_[0] == hm && l_[0] == lm);
        const unsigned long long b1 = RPL == 2 ? __ballot(h_[RPL - 1] == hm && l_[RPL - 1] == lm) : 0ull;
        p = b0 ? (int)__ffsll((long long)b0) - 1 : 64 + (int)__ffsll((long long)b1) - 1;
    } else {
        const unsigned long long f0 = __ballot(lane < n && !used[0]);
        const unsigned long long f1 = RPL == 2 ? __ballot(lane + 64 < n && !used[RPL - 1]) : 0ull;
        p = f0 ? (int)__ffsll((long long)f0) - 1 : 64 + (int)__ffsll((long long)f1) - 1;
        if (lane == 0) atomicCAS(bad_s, 0, k + 1);
    }
    return p;
}

__device__ __forceinline__ double sf_readlane_f64(double v, int lane)       // (lane: wave-uniform)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// ---- second layout (round 4, later; the default): COLUMNS PER WAVE, ROWS PER LANE.  Wave w owns the columns
// [w CPW, (w+1) CPW), lane l the rows l and l + 64 (RPL = 1 for n <= 64): the tile W[RPL][CPW] is again in registers
// with static indices, because the pivot steps run as  for (owner wave) for (c = 0 .. CPW-1, unrolled)  -- step
// k = owner * CPW + c.
//   * the pivot column k lies in ONE wave: its owner finds the pivot alone (lane = row: DPP maximum, no exchange at
//     all), publishes the column and p to LDS -- ONE workgroup barrier per step (the 16 x 16 thread-grid layout above
//     needs two: there column and row are both spread over all waves);
//   * the pivot row's entries in a wave's own columns are in that wave's lane p: a wave-local LDS line (written by one
//     lane, read by all, no barrier -- a wave's LDS operations execute in order), as in the chain kernel's rs_factor;
//   * the update is select-free: multipliers f_i = c_i / pivot for i != p and f_p = 1 - 1 / pivot (row p then becomes
//     row_p / pivot by the same formula), column k reset to the unit vector e_p beforehand, so every lane runs
//     RPL x CPW complex FMAs on the RAW pivot row: ~150 vector instructions per step and wave against ~350.
// The matrix is assembled into LDS with coalesced reads first (same operation order as the tile layout), the
// un-permuted inverse goes back to LDS, and the accumulate / store phases are those of the tile layout.
template <int RPL, int CPW, bool PANEL>
__global__ __launch_bounds__(SF_THREADS, RPL == 1 ? 4 : 2) void small_cw_kernel(SmallFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];   // PANEL: the panels' multipliers [NBUF][CPW][64 RPL]
    __shared__ cplx colb[PANEL ? 1 : 2][PANEL ? 1 : 128];         // (!PANEL) pivot column, double buffered by step parity
    __shared__ cplx rowl[4][CPW];                                 // per wave: the pivot row in the wave's columns
    __shared__ int piv_s[2];                                      // (!PANEL) pivot row of the step, by parity
    __shared__ int bad_s;
    __shared__ int pivrow_s[SF_MAXN + 8], colof_s[SF_MAXN + 8];

    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = wave * CPW;
    cplx* part = a.Gout ? nullptr : a.partial + (size_t)blockIdx.x * n * n;
    bool first = true;

    for (int e = blockIdx.x; e < a.m; e += gridDim.x) {
        // ---- assemble straight into the register tile: (E S - H) - Sigma in the reference's order (integrate.py:70),
        // contact blocks one contact after the other, as scatter_sub_kernel subtracts them.  Lane = row reads its
        // CPW consecutive columns (256 contiguous bytes per lane at CPW = 16; S and H are the same for every energy
        // and stay in L2) -- no LDS copy of the matrix: the kernel's LDS is 6 KB and the register file, not a
        // 59-KB buffer per workgroup, decides how many matrices a CU works on (n = 60: 2 -> 4)
        const cplx z = a.E[e];
        if (tid == 0) bad_s = 0;
        cplx W[RPL][CPW];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = lane + 64 * r;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int j = col0 + c;
                cplx v = cmake(0.0, 0.0);
                if (i < n && j < n) {
                    const int o = i * n + j;
                    const cplx s_ = a.S[o], h = a.H[o];
                    v = cmake(z.x * s_.x - z.y * s_.y - h.x, z.x * s_.y + z.y * s_.x - h.y);
                    if (a.sig_dense) v = csub(v, a.sig_dense[(size_t)e * a.sig_stride + o]);
                    for (int ct = 0; ct < a.n_contacts; ++ct) {
                        const int pi = a.pos[ct * n + i], pj = a.pos[ct * n + j];
                        if (pi >= 0 && pj >= 0)
                            v = csub(v, a.blk[(size_t)e * a.blk_stride + a.blk_off[ct] + pi * a.nc[ct] + pj]);
                    }
                }
                W[r][c] = v;
                if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four columns' loads in flight, not all CPW: registers)
            }
        }
        bool used[RPL];                                           // this lane's rows that have been pivots
#pragma unroll
        for (int r = 0; r < RPL; ++r) used[r] = false;
        __syncthreads();                                          // bad_s is reset; the tables of the last matrix are free

        // ---- Gauss-Jordan, implicit partial pivoting.  PANEL (NEGF_SMALL_KERNEL=panel): a wave factors its CPW columns ALONE --
        // the pivot column lies in its registers (lane = row: DPP maximum), the pivot comes from lane p by v_readlane,
        // the pivot row in its own columns through the wave-local line -- and publishes the multipliers f (not the raw
        // column) and the pivot rows of the whole panel; ONE workgroup barrier per panel; then the other three waves run
        // the panel's CPW steps on their own columns back to back (multipliers and pivot indices prefetched a step ahead,
        // no barrier, no 1 / pivot).  The next owner applies and goes straight on factoring: that is the look-ahead.
        // Every element sees the same operations in the same order as in the per-step form below: bitwise equal.
        if constexpr (PANEL) {
            constexpr int ROWS = 64 * RPL;
            constexpr int NBUF = RPL == 1 ? 2 : 1;                // (two rows per lane: one buffer and a second barrier -- LDS)
            cplx* Fp = reinterpret_cast<cplx*>(sf_smem);
            for (int q = 0; q < 4; ++q) {
                if (q * CPW >= n) break;
                cplx* Fq = Fp + (size_t)(NBUF == 2 ? (q & 1) : 0) * CPW * ROWS;
                if (wave == q) {
#pragma unroll
                    for (int c = 0; c < CPW; ++c) {
                        const int k = q * CPW + c;
                        if (k < n) {
                            cplx colv[RPL];
#pragma unroll
                            for (int r = 0; r < RPL; ++r) colv[r] = W[r][c];
                            const int p = __builtin_amdgcn_readfirstlane(cw_pivot_search<RPL>(colv, used, lane, n, k, &bad_s));
                            const int pl = p & 63, pr = p >> 6;
                            cplx pv = cmake(sf_readlane_f64(W[0][c].x, pl), sf_readlane_f64(W[0][c].y, pl));
                            if (RPL == 2 && pr) pv = cmake(sf_readlane_f64(W[RPL - 1][c].x, pl), sf_readlane_f64(W[RPL - 1][c].y, pl));
                            const double pd = pv.x * pv.x + pv.y * pv.y;
                            double sc = __builtin_amdgcn_rcp(pd);
                            sc = fma(sc, fma(-pd, sc, 1.0), sc);
                            sc = fma(sc, fma(-pd, sc, 1.0), sc);
                            const cplx ip = cmake(pv.x * sc, -pv.y * sc);
                            cplx f[RPL];
#pragma unroll
                            for (int r = 0; r < RPL; ++r) {
                                const int i = lane + 64 * r;
                                const cplx m_ = cmul(W[r][c], ip);
                                const bool isp = i == p;
                                f[r] = cmake(isp ? 1.0 - ip.x : m_.x, isp ? -ip.y : m_.y);
                                used[r] = used[r] || isp;
                                Fq[c * ROWS + i] = f[r];
                            }
                            if (lane == 0) { pivrow_s[k] = p; colof_s[p] = k; }
                            __builtin_amdgcn_wave_barrier();
                            if (lane == pl) {                     // (column k becomes e_p: its entry in row p is 1)
                                if (RPL == 2 && pr) {             // (a uniform branch, not a select between the register rows)
#pragma unroll
                                    for (int cc = 0; cc < CPW; ++cc) rowl[wave][cc] = cc == c ? cmake(1.0, 0.0) : W[RPL - 1][cc];
                                } else {
#pragma unroll
                                    for (int cc = 0; cc < CPW; ++cc) rowl[wave][cc] = cc == c ? cmake(1.0, 0.0) : W[0][cc];
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                            for (int r = 0; r < RPL; ++r) W[r][c] = cmake(lane + 64 * r == p ? 1.0 : 0.0, 0.0);
#pragma unroll
                            for (int cc = 0; cc < CPW; ++cc) {
                                const cplx rb = rowl[wave][cc];
#pragma unroll
                                for (int r = 0; r < RPL; ++r) W[r][cc] = cfnma(W[r][cc], f[r], rb);
                            }
                        }
                    }
                }
                __syncthreads();
                if (wave != q) {
                    cplx fn[RPL];
#pragma unroll
                    for (int r = 0; r < RPL; ++r) fn[r] = Fq[lane + 64 * r];
                    int pn = pivrow_s[q * CPW];
#pragma unroll
                    for (int c = 0; c < CPW; ++c) {
                        const int k = q * CPW + c;
                        if (k < n) {
                            const int p = __builtin_amdgcn_readfirstlane(pn);
                            const int pl = p & 63, pr = p >> 6;
                            cplx f[RPL];
#pragma unroll
                            for (int r = 0; r < RPL; ++r) { f[r] = fn[r]; used[r] = used[r] || lane + 64 * r == p; }
                            if (c + 1 < CPW) {                    // (the next step's multipliers and pivot row: in flight over this step)
#pragma unroll
                                for (int r = 0; r < RPL; ++r) fn[r] = Fq[(c + 1) * ROWS + lane + 64 * r];
                                pn = pivrow_s[k + 1];
                            }
                            __builtin_amdgcn_wave_barrier();
                            if (lane == pl) {
                                if (RPL == 2 && pr) {
#pragma unroll
                                    for (int cc = 0; cc < CPW; ++cc) rowl[wave][cc] = W[RPL - 1][cc];
                                } else {
#pragma unroll
                                    for (int cc = 0; cc < CPW; ++cc) rowl[wave][cc] = W[0][cc];
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                            for (int cc = 0; cc < CPW; ++cc) {
                                const cplx rb = rowl[wave][cc];
#pragma unroll
                                for (int r = 0; r < RPL; ++r) W[r][cc] = cfnma(W[r][cc], f[r], rb);
                            }
                        }
                    }
                }
                if (NBUF == 1) __syncthreads();                   // the single buffer is rewritten by the next owner
            }
        } else
        // ---- the per-step form (the default): the owner publishes column and pivot row, one barrier per step
        for (int ow = 0; ow < 4; ++ow) {
            if (ow * CPW >= n) break;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int k = ow * CPW + c;
                if (k < n) {                                      // (uniform; the steps beyond n fall away at the tail)
                    const int buf = k & 1;
                    if (wave == ow) {
                        cplx colv[RPL];
#pragma unroll
                        for (int r = 0; r < RPL; ++r) colv[r] = W[r][c];
                        const int p = cw_pivot_search<RPL>(colv, used, lane, n, k, &bad_s);
#pragma unroll
                        for (int r = 0; r < RPL; ++r) colb[buf][lane + 64 * r] = W[r][c];
                        if (lane == 0) { piv_s[buf] = p; pivrow_s[k] = p; colof_s[p] = k; }
                    }
                    __syncthreads();
                    const int p = piv_s[buf];
                    const int pl = p & 63, pr = p >> 6;
                    // 1 / pivot = conj(pivot) / |pivot|^2 by v_rcp_f64 and two Newton steps
                    const cplx pv = colb[buf][p];
                    const double pd = pv.x * pv.x + pv.y * pv.y;
                    double sc = __builtin_amdgcn_rcp(pd);
                    sc = fma(sc, fma(-pd, sc, 1.0), sc);
                    sc = fma(sc, fma(-pd, sc, 1.0), sc);
                    const cplx ip = cmake(pv.x * sc, -pv.y * sc);
                    // the pivot row in this wave's columns: lane pl writes, all read (wave-local line)
                    __builtin_amdgcn_wave_barrier();
                    if (lane == pl) {
#pragma unroll
                        for (int cc = 0; cc < CPW; ++cc) {
                            const cplx v = (RPL == 2 && pr) ? W[RPL - 1][cc] : W[0][cc];
                            // (in the owner wave column k is about to become the unit vector e_p: its entry in row p is 1)
                            rowl[wave][cc] = (wave == ow && cc == c) ? cmake(1.0, 0.0) : v;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    cplx f[RPL];
#pragma unroll
                    for (int r = 0; r < RPL; ++r) {
                        const int i = lane + 64 * r;
                        const cplx cv = colb[buf][i];             // (rows >= n hold zeros: published by the owner)
                        const cplx m_ = cmul(cv, ip);
                        const bool isp = i == p;
                        f[r] = cmake(isp ? 1.0 - ip.x : m_.x, isp ? -ip.y : m_.y);
                        used[r] = used[r] || isp;
                    }
                    if (wave == ow) {
#pragma unroll
                        for (int r = 0; r < RPL; ++r) W[r][c] = cmake(lane + 64 * r == p ? 1.0 : 0.0, 0.0);
                    }
#pragma unroll
                    for (int cc = 0; cc < CPW; ++cc) {
                        const cplx rb = rowl[wave][cc];
#pragma unroll
                        for (int r = 0; r < RPL; ++r) W[r][cc] = cfnma(W[r][cc], f[r], rb);
                    }
                }
            }
        }
        // ---- W[r][c] is G[colof[r]][pivrow[c]] (pivrow / colof / bad_s are complete: every step's owner wrote them in
        // front of that step's barrier): the weighted sum (or G itself) goes straight to its place in memory -- lane =
        // one row of G, its CPW entries at permuted columns; the four waves fill the row between them
        const int bad = bad_s;
        const double qnan = __builtin_nan("");
        if (tid == 0 && a.info) a.info[e] = bad;
        cplx* out = a.Gout ? a.Gout + (size_t)e * a.g_stride : part;
        const cplx w = a.Gout ? cmake(0.0, 0.0) : a.w[e];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = lane + 64 * r;
            if (i < n) {
                cplx* orow = out + (size_t)colof_s[i] * n;
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    const int j = col0 + c;
                    if (j < n) {
                        const cplx g = bad ? cmake(qnan, qnan) : W[r][c];
                        const int gj = pivrow_s[j];
                        if (a.Gout) orow[gj] = g;
                        else orow[gj] = cfma(first ? cmake(0.0, 0.0) : orow[gj], w, g);   // acc += w G, as accumulate_partial_kernel
                    }
                    if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        first = false;
        __syncthreads();                                          // the pivot tables and the buffers are reused
    }
}

// out[i] = sum over the workgroups' partial sums, in workgroup order (fixed -> reproducible).  256 threads = 32
// elements x 8 segments of the workgroup range; the segment sums are combined in segment order.
__global__ __launch_bounds__(256) void small_reduce_kernel(int n2, int parts, const cplx* __restrict__ partial,
                                                           cplx* __restrict__ out)
{
    __shared__ cplx seg[8][32];
    const int el = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + el;
    const int per = (parts + 7) / 8;
    const int g0 = sg * per, g1 = min(parts, g0 + per);
    cplx s = cmake(0.0, 0.0);
    if (i < n2)
        for (int g = g0; g < g1; ++g) s = cadd(s, partial[(size_t)g * n2 + i]);
    seg[sg][el] = s;
    __syncthreads();
    if (sg == 0 && i < n2) {
        cplx t = seg[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) t = cadd(t, seg[q][el]);
        out[i] = t;
    }
}

// the same for up to SEG_MAX segments in ONE launch (blockIdx.y = segment; an empty segment gets zeros): the twelve
// level sums of a joint arc + tail refinement were twelve dependent launches behind a 90-us kernel
constexpr int SEG_MAX = 32;
struct SegEnds { int end[SEG_MAX]; };
__global__ __launch_bounds__(256) void small_reduce_seg_kernel(int n2, SegEnds se, const cplx* __restrict__ partial,
                                                               cplx* __restrict__ out)
{
    __shared__ cplx seg[8][32];
    const int k = blockIdx.y;
    const int start = k ? se.end[k - 1] : 0, parts = se.end[k] - start;
    partial += (size_t)start * n2;
    out += (size_t)k * n2;
    const int el = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + el;
    const int per = (parts + 7) / 8;
    const int g0 = sg * per, g1 = min(parts, g0 + per);
    cplx s = cmake(0.0, 0.0);
    if (i < n2)
        for (int g = g0; g < g1; ++g) s = cadd(s, partial[(size_t)g * n2 + i]);
    seg[sg][el] = s;
    __syncthreads();
    if (sg == 0 && i < n2) {
        cplx t = seg[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) t = cadd(t, seg[q][el]);
        out[i] = t;
    }
}

template <int RPL, int CPW>
void cw_launch(hipStream_t st, const SmallFusedArgs& a, int grid, bool panel)
{
    constexpr size_t panel_lds = (size_t)(RPL == 1 ? 2 : 1) * CPW * 64 * RPL * sizeof(cplx);     // <= 48 KB
    if (panel) hipLaunchKernelGGL((small_cw_kernel<RPL, CPW, true>), dim3(grid), dim3(SF_THREADS), panel_lds, st, a);
    else hipLaunchKernelGGL((small_cw_kernel<RPL, CPW, false>), dim3(grid), dim3(SF_THREADS), 0, st, a);
}

template <int T>
void sf_launch(hipStream_t st, const SmallFusedArgs& a, int grid, size_t smem)
{
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_fused_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) (void)hipGetLastError();
        attr_set = true;
    }
    hipLaunchKernelGGL(small_fused_kernel<T>, dim3(grid), dim3(SF_THREADS), smem, st, a);
}

}  // namespace

bool small_fused_supported(int n) { return n >= 1 && n <= SF_MAXN; }

// NEGF_SMALL_KERNEL (A/B, cross-checks): "tile" = the 16 x 16 thread-grid layout (1), "panel" = columns per wave with one
// barrier per PANEL of CPW pivot steps (0); default "cw": columns per wave, one barrier per pivot step (2).  The panel form
// is bitwise equal and measured SLOWER (MI355X, kernel time per call: one n = 60 matrix 100 against 85 us, 972 points
// 210 against 201 us; n = 96: 283 / 222 and 747 / 631 us): the other three waves of a workgroup idle while one factors,
// and the per-step chain search -> 1 / pivot -> pivot row -> update is as long inside one wave as across the barrier.
static int sf_layout()
{
    static int layout = -1;
    if (layout < 0) { const char* e = getenv("NEGF_SMALL_KERNEL"); layout = !e ? 2 : strcmp(e, "tile") == 0 ? 1 : strcmp(e, "panel") == 0 ? 0 : 2; }
    return layout;
}

int small_fused_grid(int n, int m)
{
    // resident workgroups per CU.  Columns-per-wave layouts: no matrix in LDS (the panel form: 8 ... 48 KB of multipliers),
    // the register tile decides (n <= 64: 128 VGPRs, four waves per SIMD; above: 256, two).  Tile layout: the LDS holds
    // one un-permuted inverse per workgroup.
    int per_cu;
    if (sf_layout() != 1) per_cu = n <= 64 ? 4 : 2;
    else {
        const size_t smem = (size_t)n * (n | 1) * sizeof(cplx) + 8 * 1024;
        per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / smem));
    }
    return std::max(1, std::min(m, 256 * per_cu));
}

void launch_small_fused(hipStream_t st, SmallFusedArgs a)
{
    if (a.m <= 0) return;
    a.gp = a.n | 1;                                               // odd pitch (tile layout)
    const size_t smem = (size_t)a.n * a.gp * sizeof(cplx);
    // segments (a.nseg > 0): one workgroup and one partial record per energy, summed per segment below
    const int grid = a.nseg > 0 ? a.m : small_fused_grid(a.n, a.m);
    const int T = (a.n + 15) / 16;
    const int layout = sf_layout();
    if (layout != 1) switch (T) {
    case 1: cw_launch<1, 4>(st, a, grid, layout == 0); break;
    case 2: cw_launch<1, 8>(st, a, grid, layout == 0); break;
    case 3: cw_launch<1, 12>(st, a, grid, layout == 0); break;
    case 4: cw_launch<1, 16>(st, a, grid, layout == 0); break;
    case 5: cw_launch<2, 20>(st, a, grid, layout == 0); break;
    default: cw_launch<2, 24>(st, a, grid, layout == 0); break;
    }
    else switch (T) {
    case 1: sf_launch<1>(st, a, grid, smem); break;
    case 2: sf_launch<2>(st, a, grid, smem); break;
    case 3: sf_launch<3>(st, a, grid, smem); break;
    case 4: sf_launch<4>(st, a, grid, smem); break;
    case 5: sf_launch<5>(st, a, grid, smem); break;
    default: sf_launch<6>(st, a, grid, smem); break;
    }
    if (!a.Gout) {
        const int n2 = a.n * a.n;
        if (a.nseg > 1 && a.nseg <= SEG_MAX) {
            SegEnds se;
            for (int sg = 0; sg < SEG_MAX; ++sg) se.end[sg] = a.seg_end[std::min(sg, a.nseg - 1)];
            hipLaunchKernelGGL(small_reduce_seg_kernel, dim3((n2 + 31) / 32, a.nseg), dim3(256), 0, st, n2, se, a.partial, a.out);
        } else if (a.nseg > 0) {
            int start = 0;
            for (int sg = 0; sg < a.nseg; ++sg) {
                const int end = a.seg_end[sg];
                if (end > start)
                    hipLaunchKernelGGL(small_reduce_kernel, dim3((n2 + 31) / 32), dim3(256), 0, st, n2, end - start,
                                       a.partial + (size_t)start * n2, a.out + (size_t)sg * n2);
                else (void)hipMemsetAsync(a.out + (size_t)sg * n2, 0, (size_t)n2 * sizeof(cplx), st);
                start = end;
            }
        } else {
            hipLaunchKernelGGL(small_reduce_kernel, dim3((n2 + 31) / 32), dim3(256), 0, st, n2, grid, a.partial, a.out);
        }
    }
}
