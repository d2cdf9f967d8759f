// Job order of the 1-D chain fixed-point kernel (k_chain1d_rs.hip): the jobs of a launch differ by up to 20x in length
// (100 ... 2000 sweeps) and are started LONGEST FIRST, by sweep counts known or predicted from the previous evaluation of
// the provider -- which spreads the long jobs of a launch smaller than the chip over the compute units, and is the initial
// queue of a larger one (that one runs round robin and does not depend on it) (gauNEGF/surfG1D.py:271-288 runs every energy until ITS OWN stopping rule fires; the order never changes a
// result).  Kept apart from the kernel itself so that the predictor can change without touching the profiled source.
#include "negf_common.h"
#include <algorithm>

namespace {

constexpr int RS_ORDER_MAX = 16384;

// order[0..count) = job indices by decreasing PREDICTED sweep count (ties: increasing index): one workgroup, bitonic sort
// of packed keys in LDS, count <= RS_ORDER_MAX.  Also for a grid that was NOT evaluated before (a Fermi search moves its
// contour, an adaptive grid doubles): the sweep count of a job is predicted from the previous evaluation -- the counts vary smoothly with the energy (they
// follow the lead's bands).  Round 4: where the previous grid brackets the new energy on a line of constant Im E (the
// real-axis grids: the expensive ones), the counts of the two bracketing energies are interpolated linearly in Re E,
// per contact; elsewhere (contour points, energies outside the previous range) the count of the nearest previous energy
// is taken, as before.  For the grid of the previous evaluation itself both rules return its own counts: the learned order.
__global__ __launch_bounds__(1024) void chain1d_predict_order_kernel(
    const cplx* __restrict__ prevE, const int* __restrict__ prev_iters, int prev_n, int n_contacts,
    const cplx* __restrict__ E, int nb, int* __restrict__ order)
{
    extern __shared__ int keys[];
    const int count = nb * n_contacts;
    int np2 = 1;
    while (np2 < count) np2 <<= 1;
    for (int t = threadIdx.x; t < np2; t += blockDim.x) keys[t] = -1;
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        const cplx e = E[b];
        const double tol_im = 1e-12 + 1e-9 * fabs(e.y);
        int best = 0, lo = -1, hi = -1;
        double bd = 1e300, relo = -1e300, rehi = 1e300;
        for (int k = 0; k < prev_n; ++k) {
            const cplx q = prevE[k];
            const double dx = q.x - e.x, dy = q.y - e.y, d = dx * dx + dy * dy;
            if (d < bd) { bd = d; best = k; }           // (a NaN distance never wins: best stays a valid index)
            if (fabs(dy) <= tol_im) {
                if (q.x <= e.x && q.x > relo) { relo = q.x; lo = k; }
                if (q.x >= e.x && q.x < rehi) { rehi = q.x; hi = k; }
            }
        }
        const bool bracket = lo >= 0 && hi >= 0 && rehi > relo;
        const double tt = bracket ? (e.x - relo) / (rehi - relo) : 0.0;
        for (int cc = 0; cc < n_contacts; ++cc) {
            const int t = b * n_contacts + cc;
            int pred = prev_iters[best * n_contacts + cc];
            if (bracket) {
                const double c0 = (double)prev_iters[lo * n_contacts + cc], c1 = (double)prev_iters[hi * n_contacts + cc];
                pred = (int)(c0 + tt * (c1 - c0) + 0.5);
            }
            keys[t] = min(max(pred, 0), 131071) * RS_ORDER_MAX + (RS_ORDER_MAX - 1 - t);
        }
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np2; t += blockDim.x) {
                const int u = t ^ j;
                if (u > t) {
                    const bool desc = (t & k) == 0;
                    const int a = keys[t], b2 = keys[u];
                    if (desc ? a < b2 : a > b2) { keys[t] = b2; keys[u] = a; }
                }
            }
            __syncthreads();
        }
    for (int t = threadIdx.x; t < count; t += blockDim.x) order[t] = RS_ORDER_MAX - 1 - (keys[t] % RS_ORDER_MAX);
}

}  // namespace

bool chain1d_order_supported(int count) { return count > 0 && count <= RS_ORDER_MAX; }

void launch_chain1d_predict_order(hipStream_t st, const cplx* prevE, const int* prev_iters, int prev_n, int n_contacts,
                                  const cplx* E, int nb, int* order)
{
    const int count = nb * n_contacts;
    int np2 = 1;
    while (np2 < count) np2 <<= 1;
    hipLaunchKernelGGL(chain1d_predict_order_kernel, dim3(1), dim3(1024), (size_t)np2 * sizeof(int), st, prevE, prev_iters,
                       prev_n, n_contacts, E, nb, order);
}
