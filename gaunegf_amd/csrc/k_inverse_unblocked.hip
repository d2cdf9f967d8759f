// Unblocked in-place Gauss-Jordan inversion with partial (row) pivoting, one
// workgroup per matrix.  Works for any n; used for small matrices (n < 32), as the
// fall-back algorithm (negf_set_inverse_algo(ctx, 1)) and as the independent
// cross-check of the blocked MFMA kernel in the parity tests.
//
// Replaces utils.inv = solve(A, I) (gauNEGF/utils.py:52-54) as used by
// _gr_matrix_ops (integrate.py:71) and the transport kernels (transport.py:154,
// 163,186).  Pivot rule = LAPACK izamax (|re|+|im|, first maximum), i.e. the rule
// zgetrf applies inside solve().
//
// Step k:  p = argmax_{r>=k} |A[r][k]|_1 ; swap rows k,p ; piv = A[k][k]
//          row k  <- row k / piv, A[k][k] <- 1/piv
//          row i  <- row i - A[i][k] * row k, A[i][k] <- -A[i][k]/piv   (i != k)
// Afterwards the column swaps are undone in reverse order.
#include "negf_common.h"

static constexpr int GJ_THREADS = 256;

__global__ __launch_bounds__(GJ_THREADS) void gj_unblocked_kernel(int n, cplx* __restrict__ Aall,
                                                                    int* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx* rowk = reinterpret_cast<cplx*>(smem_raw);          // [n] scaled pivot row
    cplx* colk = rowk + n;                                   // [n] pivot column (pre-update)
    int* ipiv = reinterpret_cast<int*>(colk + n);            // [n]
    __shared__ double red_v[GJ_THREADS / 64];
    __shared__ int red_i[GJ_THREADS / 64];
    __shared__ int piv_row;
    __shared__ int bad;

    cplx* A = Aall + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) bad = 0;

    for (int k = 0; k < n; ++k) {
        // ---- pivot search in column k, rows k..n-1
        double best = -1.0;
        int bi = n;
        for (int r = k + tid; r < n; r += GJ_THREADS) {
            const double v = cabs1(A[(size_t)r * n + k]);
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
            for (int w = 1; w < GJ_THREADS / 64; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bb)) { bv = red_v[w]; bb = red_i[w]; }
            if (!(bv > 0.0)) {            // exactly singular (or NaN) column
                if (bad == 0) bad = k + 1;
                if (bb >= n) bb = k;
            }
            piv_row = bb;
            ipiv[k] = bb;
        }
        __syncthreads();
        const int p = piv_row;
        // ---- row interchange (whole rows, already-processed columns included)
        if (p != k) {
            for (int j = tid; j < n; j += GJ_THREADS) {
                const cplx a = A[(size_t)k * n + j];
                const cplx b = A[(size_t)p * n + j];
                A[(size_t)k * n + j] = b;
                A[(size_t)p * n + j] = a;
            }
        }
        __syncthreads();
        // ---- stage scaled pivot row and the pivot column
        const cplx ip = crecip(A[(size_t)k * n + k]);
        for (int j = tid; j < n; j += GJ_THREADS) {
            rowk[j] = (j == k) ? ip : cmul(A[(size_t)k * n + j], ip);
            colk[j] = A[(size_t)j * n + k];
        }
        __syncthreads();
        // ---- rank-1 update of the whole matrix; a wave walks a row segment so the
        //      16-byte accesses of its 64 lanes are contiguous
        for (int i = wave; i < n; i += GJ_THREADS / 64) {
            cplx* Ai = A + (size_t)i * n;
            if (i == k) {
                for (int j = lane; j < n; j += 64) Ai[j] = rowk[j];
            } else {
                const cplx f = colk[i];
                for (int j = lane; j < n; j += 64) {
                    if (j == k) Ai[j] = cneg(cmul(f, ip));
                    else        Ai[j] = cfnma(Ai[j], f, rowk[j]);
                }
            }
        }
        __syncthreads();
    }
    // ---- undo the interchanges on the columns, last first
    for (int k = n - 1; k >= 0; --k) {
        const int p = ipiv[k];
        if (p != k) {
            for (int i = tid; i < n; i += GJ_THREADS) {
                const cplx a = A[(size_t)i * n + k];
                const cplx b = A[(size_t)i * n + p];
                A[(size_t)i * n + k] = b;
                A[(size_t)i * n + p] = a;
            }
            __syncthreads();
        }
    }
    if (tid == 0) info[blockIdx.x] = bad;
}

bool launch_inverse_unblocked(hipStream_t st, int n, int nb, cplx* A, int* info)
{
    const size_t smem = (size_t)n * (2 * sizeof(cplx) + sizeof(int));       // pivot row, pivot column, ipiv
    const size_t limit = 158 * 1024;
    if (smem > limit) return false;                                          // n > 4494
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gj_unblocked_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)limit) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(gj_unblocked_kernel, dim3(nb), dim3(GJ_THREADS), smem, st, n, A, info);
    return true;
}
