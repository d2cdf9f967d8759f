// 1-D chain contact self-energy ("decimation") -- gauNEGF/surfG1D.py:223-295 (g),
// :344-373 (sigma).  One workgroup per (energy, contact): the fixed point is a
// strictly sequential chain of small dense operations, so parallelism comes from
// the energy grid (M x n_contacts independent workgroups) and from the nc x nc
// elements inside one step.
//
//   A = (E + i eta) Sa - a ;  B = (E + i eta) Sb - b ;  B^H = conj(B)^T   (:260-262)
//   g0 = inv(A)                                                          (:287)
//   repeat: g_new = inv(A - B g B^H)                                     (:275)
//           diff  = max |g_new - g| / max(|g_new|, 1e-12)                (:278-279)
//           g     = r g_new + (1-r) g ; count++                          (:282-283)
//   while diff > conv and count < max_iter                               (:269)
//   Sigma_c = t g t^H,  t = E Stau - tau   (no eta)                      (:369-371)
//
// Every workgroup stops on ITS OWN convergence (the reference's vmap runs all
// energies until the slowest lane converges; results are identical because a
// converged lane is frozen there).
#include "negf_common.h"

static constexpr int CH_THREADS = 256;

// Z (n x n) = X * op(Y), op = identity or conjugate transpose; all row-major ld = n
__device__ void wg_zgemm(int n, const cplx* __restrict__ X, const cplx* __restrict__ Y, int opY,
                         cplx* __restrict__ Z)
{
    for (int t = threadIdx.x; t < n * n; t += CH_THREADS) {
        const int i = t / n, j = t - i * n;
        cplx acc = cmake(0.0, 0.0);
        if (opY == 0) {
            for (int k = 0; k < n; ++k) acc = cfma(acc, X[i * n + k], Y[k * n + j]);
        } else {
            for (int k = 0; k < n; ++k) acc = cfma(acc, X[i * n + k], cconj(Y[j * n + k]));
        }
        Z[t] = acc;
    }
}

// in-place Gauss-Jordan inverse with partial pivoting (same rule as
// k_inverse_unblocked.hip) on an n x n matrix owned by this workgroup
__device__ void wg_gj_inverse(int n, cplx* __restrict__ A, cplx* rowk, cplx* colk, int* ipiv,
                              double* red_v, int* red_i, int* piv_row)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < n; ++k) {
        double best = -1.0;
        int bi = n;
        for (int r = k + tid; r < n; r += CH_THREADS) {
            const double v = cabs1(A[r * n + k]);
            if (v > best) { best = v; bi = r; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double bv = red_v[0]; int bb = red_i[0];
            for (int w = 1; w < CH_THREADS / 64; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bb)) { bv = red_v[w]; bb = red_i[w]; }
            if (bb >= n) bb = k;
            *piv_row = bb;
            ipiv[k] = bb;
        }
        __syncthreads();
        const int p = *piv_row;
        if (p != k) {
            for (int j = tid; j < n; j += CH_THREADS) {
                const cplx a = A[k * n + j], b = A[p * n + j];
                A[k * n + j] = b; A[p * n + j] = a;
            }
        }
        __syncthreads();
        const cplx ip = crecip(A[k * n + k]);
        for (int j = tid; j < n; j += CH_THREADS) {
            rowk[j] = (j == k) ? ip : cmul(A[k * n + j], ip);
            colk[j] = A[j * n + k];
        }
        __syncthreads();
        for (int t = tid; t < n * n; t += CH_THREADS) {
            const int i = t / n, j = t - i * n;
            if (i == k)      A[t] = rowk[j];
            else if (j == k) A[t] = cneg(cmul(colk[i], ip));
            else             A[t] = cfnma(A[t], colk[i], rowk[j]);
        }
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; --k) {
        const int p = ipiv[k];
        if (p != k) {
            for (int i = tid; i < n; i += CH_THREADS) {
                const cplx a = A[i * n + k], b = A[i * n + p];
                A[i * n + k] = b; A[i * n + p] = a;
            }
            __syncthreads();
        }
    }
}

struct Chain1DArgs {
    const cplx *alpha, *Salpha, *beta, *Sbeta, *tau, *Stau;   // concatenated per contact
    const int* nc;          // [n_contacts]
    const int* blk_off;     // [n_contacts] offset of the contact inside one energy record
    int n_contacts;
    int blk_stride;
    double eta, conv, relFactor;
    int max_iter, force_iters;
};

__global__ __launch_bounds__(CH_THREADS) void chain1d_kernel(
    Chain1DArgs a, const cplx* __restrict__ E, cplx* __restrict__ blk, int* __restrict__ iters,
    int* __restrict__ converged, cplx* __restrict__ scratch, size_t scratch_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red_v[CH_THREADS / 64];
    __shared__ int red_i[CH_THREADS / 64];
    __shared__ int piv_row;
    __shared__ double diff_sh;

    const int c = blockIdx.x, b = blockIdx.y;
    const int n = a.nc[c];
    const int off = a.blk_off[c];
    const int n2 = n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    cplx* rowk = reinterpret_cast<cplx*>(smem_raw);
    cplx* colk = rowk + n;
    int* ipiv = reinterpret_cast<int*>(colk + n);

    cplx* ws = scratch + ((size_t)b * a.n_contacts + c) * scratch_per_wg;
    cplx* A = ws;             // A
    cplx* B = A + n2;         // B
    cplx* g = B + n2;         // current (mixed) g
    cplx* T = g + n2;         // B g
    cplx* Mx = T + n2;        // A - T B^H  -> g_new in place

    const cplx e = E[b];
    const cplx z = cmake(e.x, e.y + a.eta);
    for (int t = tid; t < n2; t += CH_THREADS) {
        A[t] = csub(cmul(z, a.Salpha[off + t]), a.alpha[off + t]);
        B[t] = csub(cmul(z, a.Sbeta[off + t]), a.beta[off + t]);
        g[t] = A[t];
    }
    __syncthreads();
    wg_gj_inverse(n, g, rowk, colk, ipiv, red_v, red_i, &piv_row);   // g = inv(A)

    int count = 0;
    double diff = INFINITY;
    while (true) {
        if (a.force_iters >= 0) { if (count >= a.force_iters) break; }
        else if (!(diff > a.conv && count < a.max_iter)) break;
        wg_zgemm(n, B, g, 0, T);
        __syncthreads();
        // Mx = A - T B^H
        for (int t = tid; t < n2; t += CH_THREADS) {
            const int i = t / n, j = t - i * n;
            cplx acc = A[t];
            for (int k = 0; k < n; ++k) acc = cfnma(acc, T[i * n + k], cconj(B[j * n + k]));
            Mx[t] = acc;
        }
        __syncthreads();
        wg_gj_inverse(n, Mx, rowk, colk, ipiv, red_v, red_i, &piv_row);
        // diff and mixing
        double d = 0.0;
        for (int t = tid; t < n2; t += CH_THREADS) {
            const cplx gn = Mx[t], go = g[t];
            const double num = hypot(gn.x - go.x, gn.y - go.y);
            const double den = fmax(hypot(gn.x, gn.y), 1e-12);
            d = fmax(d, num / den);
            g[t] = cmake(gn.x * a.relFactor + go.x * (1.0 - a.relFactor),
                         gn.y * a.relFactor + go.y * (1.0 - a.relFactor));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_down(d, o, 64));
        if (lane == 0) red_v[wave] = d;
        __syncthreads();
        if (tid == 0) {
            double m = red_v[0];
            for (int w = 1; w < CH_THREADS / 64; ++w) m = fmax(m, red_v[w]);
            diff_sh = m;
        }
        __syncthreads();
        diff = diff_sh;
        ++count;
    }
    // Sigma_c = t g t^H, t = E Stau - tau
    for (int t = tid; t < n2; t += CH_THREADS) A[t] = csub(cmul(e, a.Stau[off + t]), a.tau[off + t]);
    __syncthreads();
    wg_zgemm(n, A, g, 0, T);
    __syncthreads();
    wg_zgemm(n, T, A, 1, blk + (size_t)b * a.blk_stride + off);
    if (tid == 0) {
        if (iters) iters[(size_t)b * a.n_contacts + c] = count;
        if (converged) converged[(size_t)b * a.n_contacts + c] = (diff <= a.conv) ? 1 : 0;
    }
}

size_t chain1d_scratch_per_wg(int nc_max) { return (size_t)5 * nc_max * nc_max; }

void launch_chain1d(hipStream_t st, const SigmaProvider& p, const int* d_nc, const int* d_blk_off,
                    int nb, const cplx* E, cplx* blk, int* iters, int* conv, cplx* scratch,
                    size_t scratch_per_wg)
{
    Chain1DArgs a;
    a.alpha = p.d_alpha; a.Salpha = p.d_Salpha; a.beta = p.d_beta; a.Sbeta = p.d_Sbeta;
    a.tau = p.d_tau; a.Stau = p.d_Stau;
    a.nc = d_nc; a.blk_off = d_blk_off;
    a.n_contacts = p.n_contacts; a.blk_stride = p.blk_stride;
    a.eta = p.eta; a.conv = p.conv; a.relFactor = p.relFactor;
    a.max_iter = p.max_iter; a.force_iters = p.force_iters;
    const size_t smem = (size_t)p.nc_max * (2 * sizeof(cplx) + sizeof(int));
    hipLaunchKernelGGL(chain1d_kernel, dim3(p.n_contacts, nb), dim3(CH_THREADS), smem, st, a, E, blk,
                       iters, conv, scratch, scratch_per_wg);
}
