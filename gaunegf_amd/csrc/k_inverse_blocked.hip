// Blocked Gauss-Jordan inversion with partial pivoting, FP64 MFMA trailing updates,
// one workgroup (512 threads, 8 waves) per matrix.   gfx950 / MI355X.
//
// Replaces G = solve(E S - F - Sigma, I)  (gauNEGF/integrate.py:71, utils.py:52-54,
// transport.py:154,163,186) for every energy point of the grid.
//
// Algorithm (block column K = columns [k0, k0+kw), kw <= NB):
//   1. PANEL.  The n x kw panel is held in REGISTERS, one row strip per thread
//      (S complex128 per strip).  kw unblocked Gauss-Jordan column steps with
//      partial pivoting (LAPACK izamax rule |re|+|im|, first maximum) run on the
//      strips; only the pivot row (NB values), the pivot column (n values) and the
//      per-wave arg-max partials go through LDS (two barriers per column step; the
//      pivot search of step j+1 is fused into the update of step j; the wave-level
//      arg-max uses DPP lane moves, not LDS permutes).  Rows are never moved: each
//      strip carries its logical position `pos`; the interchange sequence ipiv[] is
//      recorded.  After the kw steps the panel equals the block column of the
//      elementary transform  M_K = [ -A01 A11^-1 ; A11^-1 ; -A21 A11^-1 ].
//   2. The strips are written to LDS as P (logical row order) together with the row
//      map src[] (new logical row i <- old row src[i]).
//   3. TRAILING UPDATE on the matrix cores, OUT OF PLACE (ping-pong buffers), which
//      folds the row interchanges into the tile loads and removes every in-place
//      hazard:     new[i][J] = (i in K ? 0 : old[src[i]][J]) + P[i][:] * Q[:][J],
//      Q[k][J] = old[src[k0+k]][J];   new[:, K] = P.
//      Work item = (column tile J, half of the row tiles): the wave loads the Q
//      fragments of J once into registers and sweeps its row tiles, prefetching the
//      next C tile while the current 16x16 tile runs its 4*NB/4 MFMAs
//      (Cr += Pr Qr; Cr += Pi (-Qi); Ci += Pr Qi; Ci += Pi Qr).
//   4. After the last panel the column interchanges are undone (reverse order) while
//      copying to the other buffer.
// Flops: 8 n^3 per matrix (complex MAC = 8), the LU + triangular-inversion optimum;
// every step updates the full n x n matrix, so the MFMA work per step is uniform.
//
// Data layout: row-major complex128 (interleaved), ld = n.  A lane fetches one
// complex element (16 B) per MFMA operand; 16 lanes cover 256 contiguous bytes of
// a matrix row, so tile loads/stores are 4 x 256-B row segments per wave instruction.
#include "negf_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GJB_THREADS = 512;
constexpr int GJB_WAVES = GJB_THREADS / 64;

template <int NB, int CPR, int RPT>
struct GjCfg {
    static constexpr int S = NB / CPR;             // complex values per strip
    static constexpr int TPR = GJB_THREADS / CPR;  // threads along the row dimension
    static constexpr int ROWS = TPR * RPT;         // row capacity
    static constexpr int PITCH = NB + 1;           // LDS row pitch of P in complex (odd -> conflict free)
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
__device__ __forceinline__ int make_key(int pos, int phys) { return (pos << 12) | phys; }

// ---- one Gauss-Jordan column step on the register strips, J known at compile time ----
template <int NB, int CPR, int RPT>
struct PanelCtx {
    cplx (&a)[RPT][NB / CPR];
    int (&pos)[RPT];
    cplx* rowbuf; cplx* colbuf; RedSlot* red; cplx* piv_ip; int* bad_sh; int* ipiv;
    int n, k0, kw, tid, lane, wave, h, tr;
};

template <int NB, int CPR, int RPT, int J>
struct PanelSteps {
    static __device__ __forceinline__ void run(PanelCtx<NB, CPR, RPT>& x)
    {
        using C = GjCfg<NB, CPR, RPT>;
        constexpr int S = C::S, TPR = C::TPR;
        constexpr int hj = J / S, sj = J % S;
        if (J < x.kw) {                                     // uniform branch
            const int c = x.k0 + J;
            RedSlot* red = x.red + (J & 1) * GJB_WAVES;
            // (1) combine the per-wave partials published by the previous step
            double wv = red[0].v; int wkey = red[0].key;
#pragma unroll
            for (int w = 1; w < GJB_WAVES; ++w) {
                const double ov = red[w].v; const int ok = red[w].key;
                const bool take = (ov > wv) | ((ov == wv) & (ok < wkey));
                wv = take ? ov : wv; wkey = take ? ok : wkey;
            }
            int p, pphys;
            if (wkey != KEY_NONE) { p = wkey >> 12; pphys = wkey & 0xFFF; }
            else { p = c; pphys = -1; }                     // NaN column: keep the diagonal row
            if (!(wv > 0.0) && x.tid == 0 && *x.bad_sh == 0) *x.bad_sh = c + 1;   // exactly singular / NaN
            // (2) publish the unscaled pivot row, 1/pivot and the pivot column
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = x.tr + q * TPR;
                const bool is_piv = (pphys >= 0) ? (r == pphys) : (x.pos[q] == c && r < x.n);
                if (is_piv) {
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
            if (x.tid == 0) x.ipiv[c] = p;
            __syncthreads();
            // (3) rank-1 update of every strip.  One batch of LDS reads (pivot row part, 1/pivot,
            //     own pivot-column entries), then pure register arithmetic.  The pivot row and
            //     the other rows share one form  a <- base + coef * rowbuf:
            //        pivot row : base = 0, coef = 1/pivot        (row / pivot)
            //        other rows: base = a, coef = -(f / pivot)   (row - f/pivot * pivot row)
            //     and the pivot-column entry becomes coef in both cases.
            const cplx ip = *x.piv_ip;
            cplx rb[S];
#pragma unroll
            for (int s = 0; s < S; ++s) rb[s] = x.rowbuf[x.h * S + s];
            cplx fcol[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) fcol[q] = x.colbuf[x.tr + q * TPR];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = x.tr + q * TPR;
                const bool is_piv = (pphys >= 0) ? (r == pphys) : (x.pos[q] == c && r < x.n);
                const cplx nfm = cneg(cmul(fcol[q], ip));
                const cplx coef = is_piv ? ip : nfm;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const cplx base = is_piv ? cmake(0.0, 0.0) : x.a[q][s];
                    x.a[q][s] = cfma(base, coef, rb[s]);
                }
                if (x.h == hj) x.a[q][sj] = coef;
                x.pos[q] = is_piv ? c : (x.pos[q] == c ? p : x.pos[q]);
            }
            // (4) pivot search for column J+1 on the freshly updated strips
            if constexpr (J + 1 < NB) {
                constexpr int hn = (J + 1) / S, sn = (J + 1) % S;
                double bv = -1.0; int bkey = KEY_NONE;
                if (x.h == hn && J + 1 < x.kw) {
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int r = x.tr + q * TPR;
                        if (r < x.n && x.pos[q] >= c + 1) {
                            const double v = cabs1(x.a[q][sn]);
                            const int key = make_key(x.pos[q], r);
                            if (v > bv || (v == bv && key < bkey)) { bv = v; bkey = key; }
                        }
                    }
                }
                wave_argmax(bv, bkey);
                RedSlot* rn = x.red + ((J + 1) & 1) * GJB_WAVES;
                if (x.lane == 0) { rn[x.wave].v = bv; rn[x.wave].key = bkey; }
            }
            __syncthreads();
            if constexpr (J + 1 < NB) PanelSteps<NB, CPR, RPT, J + 1>::run(x);
        }
    }
};

template <int NB, int CPR, int RPT>
__global__ __launch_bounds__(GJB_THREADS) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info,
    int dbg /* ablation switches, 0 in production: 1 = no pivot steps, 2 = no MFMA, 4 = no tile loads */)
{
    using C = GjCfg<NB, CPR, RPT>;
    constexpr int S = C::S, TPR = C::TPR, PITCH = C::PITCH;
    constexpr int KS = NB / 4;                     // MFMA k-steps per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows16 = (n + 15) & ~15;
    cplx* P = reinterpret_cast<cplx*>(smem_raw);                 // [rows16][PITCH]
    cplx* rowbuf = P + (size_t)rows16 * PITCH;                   // [NB]  unscaled pivot row
    cplx* colbuf = rowbuf + NB;                                  // [ROWS] pivot column
    int* src = reinterpret_cast<int*>(colbuf + C::ROWS);         // [rows16] new row i <- old row src[i]
    int* ipiv = src + rows16;                                    // [n]
    int* colsrc = ipiv + rows16;                                 // [rows16]
    __shared__ RedSlot red[2][GJB_WAVES];
    __shared__ cplx piv_ip;
    __shared__ int bad_sh;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = tid / TPR;                 // which column part of the panel this thread holds
    const int tr = tid - h * TPR;            // row slot
    cplx* cur = bufA + (size_t)blockIdx.x * mat_stride;
    cplx* nxt = bufB + (size_t)blockIdx.x * mat_stride;

    if (tid == 0) bad_sh = 0;
    // rows >= n of P stay zero for the whole kernel (A operand of the edge tiles)
    for (int t = tid; t < (rows16 - n) * PITCH; t += GJB_THREADS) P[(size_t)n * PITCH + t] = cmake(0.0, 0.0);

    const int fi = lane & 15, fk = lane >> 4;
    const int tiles = rows16 >> 4;

    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kw = min(NB, n - k0);
        __syncthreads();
        // ---------------- panel: global -> register strips.  Every thread fetches its own strip
        // (S independent 16-byte loads issued back to back: one memory latency per panel)
        cplx a[RPT][S];
        int pos[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            pos[q] = r;
            const cplx* g = cur + (size_t)(r < n ? r : 0) * n + k0 + h * S;
#pragma unroll
            for (int s = 0; s < S; ++s)
                a[q][s] = (r < n && h * S + s < kw) ? g[s] : cmake(0.0, 0.0);
        }
        // ---------------- pivot search for the first column of the panel
        if (!(dbg & 1)) {
            double bv = -1.0; int bkey = KEY_NONE;
            if (h == 0) {
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = tr + q * TPR;
                    if (r < n && pos[q] >= k0) {
                        const double v = cabs1(a[q][0]);
                        const int key = make_key(pos[q], r);
                        if (v > bv || (v == bv && key < bkey)) { bv = v; bkey = key; }
                    }
                }
            }
            wave_argmax(bv, bkey);
            if (lane == 0) { red[0][wave].v = bv; red[0][wave].key = bkey; }
        }
        __syncthreads();
        // ---------------- kw Gauss-Jordan column steps on the register strips
        // (compile-time recursion over the panel column: every strip index is a constant)
        if (!(dbg & 1)) {
            PanelCtx<NB, CPR, RPT> ctx{a, pos, rowbuf, colbuf, &red[0][0], &piv_ip, &bad_sh, ipiv,
                                       n, k0, kw, tid, lane, wave, h, tr};
            PanelSteps<NB, CPR, RPT, 0>::run(ctx);
        }
        // ---------------- strips -> P (logical rows) and the row map
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            if (r < n) {
#pragma unroll
                for (int s = 0; s < S; ++s) P[(size_t)pos[q] * PITCH + h * S + s] = a[q][s];
                if (h == 0) src[pos[q]] = r;
            }
        }
        __syncthreads();
        // ---------------- panel columns of the new buffer
        for (int t = tid; t < n * kw; t += GJB_THREADS) {
            const int r = t / kw, j = t - r * kw;
            nxt[(size_t)r * n + k0 + j] = P[(size_t)r * PITCH + j];
        }
        // ---------------- trailing update
        // column tiles fully inside the panel are skipped; a tile that only touches it
        // (NB = 8, or the ragged last panel) is computed and its panel columns masked
        const int pt_lo = (k0 + 15) >> 4;                 // first tile fully inside [k0, k0+kw) ...
        const int pt_hi = (k0 + kw) >> 4;                 // ... up to (excluding) this one
        const int n_skip = max(0, pt_hi - pt_lo);
        const int ct = tiles - n_skip;                    // column tiles to process
        const int rhalf = (tiles + 1) >> 1;               // row tiles in the first half
        for (int item = wave; item < ct * 2; item += GJB_WAVES) {
            const int cx = item >> 1, part = item & 1;
            const int tj = (n_skip > 0 && cx >= pt_lo) ? cx + n_skip : cx;
            const int ti0 = part ? rhalf : 0, ti1 = part ? tiles : rhalf;
            const int col = tj * 16 + fi;
            const bool col_ok = col < n;
            const bool col_store = col_ok && !(col >= k0 && col < k0 + kw);
            // Q fragments of this column tile: Q[k][col] = old[src[k0+k]][col]
            cplx qf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 4 + fk;
                qf[ks] = cmake(0.0, 0.0);
                if (k < kw && col_ok && !(dbg & 4)) qf[ks] = cur[(size_t)src[k0 + k] * n + col];
            }
            // C tiles are prefetched two row tiles ahead (three tiles of loads in flight per wave
            // with the one being consumed): keeps enough bytes in flight to cover HBM latency
            auto load_c = [&](int ti, cplx (&dst)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + fk + 4 * r;
                    dst[r] = cmake(0.0, 0.0);
                    if (ti < ti1 && i < n && col_ok && !(i >= k0 && i < k0 + kw) && !(dbg & 4))
                        dst[r] = cur[(size_t)src[i] * n + col];
                }
            };
            cplx c0[4], c1[4];
            load_c(ti0, c0);
            load_c(ti0 + 1, c1);
            for (int ti = ti0; ti < ti1; ++ti) {
                d4 accr, acci;
#pragma unroll
                for (int r = 0; r < 4; ++r) { accr[r] = c0[r].x; acci[r] = c0[r].y; }
#pragma unroll
                for (int r = 0; r < 4; ++r) c0[r] = c1[r];
                load_c(ti + 2, c1);
                const cplx* prow = P + (size_t)(ti * 16 + fi) * PITCH + fk;
                cplx pa[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) pa[ks] = prow[ks * 4];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks * 4 < kw) {
                        if (dbg & 2) { accr[0] += pa[ks].x * qf[ks].x; acci[0] += pa[ks].y * qf[ks].y; continue; }
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x, qf[ks].x, accr, 0, 0, 0);
                        accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].y, -qf[ks].y, accr, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].x, qf[ks].y, acci, 0, 0, 0);
                        acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks].y, qf[ks].x, acci, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + fk + 4 * r;
                    if (i < n && col_store) nxt[(size_t)i * n + col] = cmake(accr[r], acci[r]);
                }
            }
        }
        // swap buffers
        cplx* tmp = cur; cur = nxt; nxt = tmp;
    }
    __syncthreads();
    // ---------------- undo the column interchanges (reverse order) while copying
    for (int t = tid; t < n; t += GJB_THREADS) colsrc[t] = t;
    __syncthreads();
    if (tid == 0) {
        for (int c = n - 1; c >= 0; --c) {
            const int p = ipiv[c];
            if (p != c) { const int x = colsrc[c]; colsrc[c] = colsrc[p]; colsrc[p] = x; }
        }
        info[blockIdx.x] = bad_sh;
    }
    __syncthreads();
    // four rows per wave iteration: up to 16 independent gathers in flight per lane
    for (int i0 = wave * 4; i0 < n; i0 += GJB_WAVES * 4) {
        for (int j0 = 0; j0 < n; j0 += 256) {
            cplx v[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = i0 + rr, j = j0 + jj * 64 + lane;
                    v[rr][jj] = (i < n && j < n) ? cur[(size_t)i * n + colsrc[j]] : cmake(0.0, 0.0);
                }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = i0 + rr, j = j0 + jj * 64 + lane;
                    if (i < n && j < n) nxt[(size_t)i * n + j] = v[rr][jj];
                }
        }
    }
}

template <int NB, int CPR, int RPT>
size_t gj_smem(int n)
{
    using C = GjCfg<NB, CPR, RPT>;
    const size_t rows16 = (size_t)((n + 15) & ~15);
    return rows16 * C::PITCH * sizeof(cplx) + NB * sizeof(cplx) + (size_t)C::ROWS * sizeof(cplx) +
           3 * rows16 * sizeof(int);
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 512;       // static __shared__ of the kernel is < 512 B

template <int NB, int CPR, int RPT>
bool gj_fits(int n)
{
    return n <= GjCfg<NB, CPR, RPT>::ROWS && n < 4096 && gj_smem<NB, CPR, RPT>(n) <= LDS_LIMIT;
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

// which configuration serves dimension n: 0 = none
int gj_pick(int n)
{
    if (n < 16) return 0;                               // tiny matrices: the unblocked kernel
    if (gj_fits<32, 2, 1>(n)) return 1;                 // n <= 256, panel 32
    if (gj_fits<16, 1, 1>(n)) return 2;                 // n <= 512, panel 16
    if (gj_fits<8, 1, 2>(n)) return 3;                  // n <= ~960, panel 8
    return 0;
}

int gj_panels(int n, int cfg)
{
    const int NB = cfg == 1 ? 32 : (cfg == 2 ? 16 : 8);
    return (n + NB - 1) / NB;
}

}  // namespace

bool inverse_blocked_supported(int n) { return gj_pick(n) != 0; }

// Returns true when the result ends up in B (the ping-pong parity), false when in A.
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* info)
{
    const int cfg = gj_pick(n);
    switch (cfg) {
    case 1: gj_launch<32, 2, 1>(st, n, nb, A, B, stride, info); break;
    case 2: gj_launch<16, 1, 1>(st, n, nb, A, B, stride, info); break;
    case 3: gj_launch<8, 1, 2>(st, n, nb, A, B, stride, info); break;
    default: return false;
    }
    // np panel passes + 1 unscramble pass, each flipping the buffer
    return ((gj_panels(n, cfg) + 1) & 1) != 0;
}
