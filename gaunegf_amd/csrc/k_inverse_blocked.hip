// Blocked in-place-style Gauss-Jordan inversion with partial pivoting, FP64 MFMA
// trailing updates, one workgroup (512 threads, 8 waves) per matrix.   gfx950.
//
// Replaces G = solve(E S - F - Sigma, I)  (gauNEGF/integrate.py:71, utils.py:52-54,
// transport.py:154,163,186) for every energy point of the grid.
//
// Algorithm (block column K = columns [k0, k0+kw), kw <= NB):
//   1. PANEL.  The n x kw panel is held in REGISTERS, one row strip per thread
//      (S complex128 per strip).  kw unblocked Gauss-Jordan column steps with
//      partial pivoting (LAPACK izamax rule |re|+|im|, first maximum) run on the
//      strips; only the pivot row (NB values), the pivot column (n values) and the
//      arg-max partials go through LDS.  Rows are never moved: each strip carries
//      its logical position `pos`; the interchange sequence ipiv[] is recorded.
//      After the kw steps the panel equals the block column of the elementary
//      transform  M_K = [ -A01 A11^-1 ; A11^-1 ; -A21 A11^-1 ]  (rows in permuted order).
//   2. The strips are written to LDS as P (logical row order) together with the row
//      map src[] (new logical row i <- old row src[i]).
//   3. TRAILING UPDATE on the matrix cores, OUT OF PLACE (ping-pong buffers), which
//      folds the row interchanges into the tile loads and removes every in-place
//      hazard:     new[i][J] = (i in K ? 0 : old[src[i]][J]) + P[i][:] * Q[:][J],
//      Q[k][J] = old[src[k0+k]][J];   new[:, K] = P.
//      A 16x16 complex tile is 4 real v_mfma_f64_16x16x4_f64 chains per 4-deep k-step
//      (Cr += Pr Qr; Cr += (-Pi) Qi; Ci += Pr Qi; Ci += Pi Qr).
//   4. After the last panel the column interchanges are undone (reverse order) while
//      copying to the other buffer.
// Flops: 8 n^3 per matrix (complex MAC = 8), the same as LU + triangular inversion;
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

struct RedSlot { double v; int pos; int phys; };

template <int NB, int CPR, int RPT>
__global__ __launch_bounds__(GJB_THREADS) void gj_blocked_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB, size_t mat_stride, int* __restrict__ info)
{
    using C = GjCfg<NB, CPR, RPT>;
    constexpr int S = C::S, TPR = C::TPR, PITCH = C::PITCH;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows16 = (n + 15) & ~15;
    cplx* P = reinterpret_cast<cplx*>(smem_raw);                 // [rows16][PITCH]
    cplx* rowbuf = P + (size_t)rows16 * PITCH;                   // [NB]  unscaled pivot row
    cplx* colbuf = rowbuf + NB;                                  // [ROWS] pivot column
    int* src = reinterpret_cast<int*>(colbuf + C::ROWS);         // [rows16] new row i <- old row src[i]
    int* ipiv = src + rows16;                                    // [n]
    int* colsrc = ipiv + rows16;                                 // [rows16]
    __shared__ RedSlot red[GJB_WAVES];
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
        // ---------------- panel: global (coalesced) -> LDS -> register strips
        for (int t = tid; t < n * NB; t += GJB_THREADS) {
            const int r = t / NB, j = t - r * NB;
            P[(size_t)r * PITCH + j] = (j < kw) ? cur[(size_t)r * n + k0 + j] : cmake(0.0, 0.0);
        }
        __syncthreads();
        cplx a[RPT][S];
        int pos[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tr + q * TPR;
            pos[q] = r;
#pragma unroll
            for (int s = 0; s < S; ++s) a[q][s] = (r < n) ? P[(size_t)r * PITCH + h * S + s] : cmake(0.0, 0.0);
        }
        // ---------------- kw Gauss-Jordan column steps on the register strips
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < kw) {                                   // uniform branch
                const int c = k0 + j;
                const int hj = j / S, sj = j % S;           // compile-time: j is unrolled
                // (1) arg-max of |.|_1 over logical rows >= c
                double bv = -1.0; int bpos = 0x7fffffff, bphys = -1;
                if (h == hj) {
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int r = tr + q * TPR;
                        if (r < n && pos[q] >= c) {
                            const double v = cabs1(a[q][sj]);
                            if (v > bv || (v == bv && pos[q] < bpos)) { bv = v; bpos = pos[q]; bphys = r; }
                        }
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ov = __shfl_down(bv, off, 64);
                    const int op = __shfl_down(bpos, off, 64);
                    const int oph = __shfl_down(bphys, off, 64);
                    if (ov > bv || (ov == bv && op < bpos)) { bv = ov; bpos = op; bphys = oph; }
                }
                __syncthreads();                            // previous step's LDS reads are done
                if (lane == 0) { red[wave].v = bv; red[wave].pos = bpos; red[wave].phys = bphys; }
                __syncthreads();
                double wv = red[0].v; int p = red[0].pos, pphys = red[0].phys;
#pragma unroll
                for (int w = 1; w < GJB_WAVES; ++w) {
                    const double ov = red[w].v; const int op = red[w].pos;
                    if (ov > wv || (ov == wv && op < p)) { wv = ov; p = op; pphys = red[w].phys; }
                }
                if (!(wv > 0.0)) {                          // exactly singular / NaN column
                    if (tid == 0 && bad_sh == 0) bad_sh = c + 1;
                    if (pphys < 0) p = c;                   // NaN everywhere: keep the diagonal row
                }
                // (2) publish the unscaled pivot row, 1/pivot and the pivot column
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = tr + q * TPR;
                    const bool is_piv = (pphys >= 0) ? (r == pphys) : (pos[q] == c && r < n);
                    if (is_piv) {
#pragma unroll
                        for (int s = 0; s < S; ++s) rowbuf[h * S + s] = a[q][s];
                        if (h == hj) piv_ip = crecip(a[q][sj]);
                    }
                    if (h == hj) colbuf[r] = a[q][sj];
                }
                if (tid == 0) ipiv[c] = p;
                __syncthreads();
                // (3) rank-1 update of every strip
                const cplx ip = piv_ip;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = tr + q * TPR;
                    const bool is_piv = (pphys >= 0) ? (r == pphys) : (pos[q] == c && r < n);
                    if (is_piv) {
#pragma unroll
                        for (int s = 0; s < S; ++s) a[q][s] = cmul(rowbuf[h * S + s], ip);
                        if (h == hj) a[q][sj] = ip;
                        pos[q] = c;
                    } else {
                        const cplx fm = cmul(colbuf[r], ip);            // multiplier f / pivot
#pragma unroll
                        for (int s = 0; s < S; ++s) a[q][s] = cfnma(a[q][s], fm, rowbuf[h * S + s]);
                        if (h == hj) a[q][sj] = cneg(fm);
                        if (pos[q] == c) pos[q] = p;
                    }
                }
            }
        }
        __syncthreads();
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
        // ---------------- trailing update: one 16 x 16 tile per wave iteration
        const int first_pt = k0 >> 4, last_pt = (k0 + kw - 1) >> 4;     // column tiles touching the panel
        const bool panel_full_tiles = ((k0 & 15) == 0) && (((k0 + kw) & 15) == 0 || k0 + kw == n);
        for (int item = wave; item < tiles * tiles; item += GJB_WAVES) {
            const int tj = item / tiles, ti = item - tj * tiles;          // column-major: consecutive items share Q
            if (panel_full_tiles && tj >= first_pt && tj <= last_pt) continue;
            const int col = tj * 16 + fi;
            const bool col_ok = col < n;
            d4 accr = {0, 0, 0, 0}, acci = {0, 0, 0, 0};
            // C init: old[src[i]][col] unless i is a pivot-block row
            int row_i[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                row_i[r] = i;
                if (i < n && col_ok && !(i >= k0 && i < k0 + kw)) {
                    const cplx v = cur[(size_t)src[i] * n + col];
                    accr[r] = v.x; acci[r] = v.y;
                }
            }
#pragma unroll
            for (int ks = 0; ks < NB; ks += 4) {
                if (ks < kw) {
                    const cplx pa = P[(size_t)(ti * 16 + fi) * PITCH + ks + fk];
                    cplx qb = cmake(0.0, 0.0);
                    const int k = ks + fk;
                    if (k < kw && col_ok) qb = cur[(size_t)src[k0 + k] * n + col];
                    accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.x, accr, 0, 0, 0);
                    accr = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.y, qb.y, accr, 0, 0, 0);
                    acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.y, acci, 0, 0, 0);
                    acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, qb.x, acci, 0, 0, 0);
                }
            }
            const bool col_store = col_ok && !(col >= k0 && col < k0 + kw);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row_i[r] < n && col_store) nxt[(size_t)row_i[r] * n + col] = cmake(accr[r], acci[r]);
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
    for (int i = wave; i < n; i += GJB_WAVES) {
        const cplx* srow = cur + (size_t)i * n;
        cplx* drow = nxt + (size_t)i * n;
        for (int j = lane; j < n; j += 64) drow[j] = srow[colsrc[j]];
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

constexpr size_t LDS_LIMIT = 160 * 1024 - 256;       // static __shared__ of the kernel is < 256 B

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
    hipLaunchKernelGGL(kern, dim3(nb), dim3(GJB_THREADS), smem, st, n, A, B, stride, info);
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
