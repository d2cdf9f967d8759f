// HBM-bound elementwise / reduction kernels of the energy-grid engine (gfx950).
//   assemble   : A_b = E_b S - H - Sigma_b            (integrate.py:70, transport.py:153)
//   accumulate : acc += sum_b w_b X_b                 (integrate.py:105,119)
//   trace_dot  : Re sum_ij X_ij conj(G_ij)            (transport.py:157 -- Tr[X G^H])
//   dos        : -Im diag(G)/pi and its sum           (transport.py:188-189, density.py:54)
// All accesses are 16-byte (one complex128) per lane, consecutive lanes on
// consecutive elements -> fully coalesced 1 KiB wave transactions.
#include "negf_common.h"
#include <algorithm>

static constexpr int EW_THREADS = 256;

// ---------------------------------------------------------------- assemble
__global__ __launch_bounds__(EW_THREADS) void assemble_kernel(
    int n2, const cplx* __restrict__ E, const cplx* __restrict__ S, const cplx* __restrict__ H,
    const cplx* __restrict__ sig_dense, cplx* __restrict__ A)
{
    const int b = blockIdx.y;
    const cplx e = E[b];
    const size_t base = (size_t)b * n2;
    for (int i = blockIdx.x * EW_THREADS + threadIdx.x; i < n2; i += gridDim.x * EW_THREADS) {
        const cplx s = S[i];
        const cplx h = H[i];
        // (E*S - H) - Sigma, the reference's evaluation order (integrate.py:70)
        cplx a = cmake(e.x * s.x - e.y * s.y - h.x, e.x * s.y + e.y * s.x - h.y);
        if (sig_dense) a = csub(a, sig_dense[base + i]);
        A[base + i] = a;
    }
}

// A_b[inds_c[i]][inds_c[j]] -= blk_b[c][i][j]   (one block per (contact, energy))
__global__ __launch_bounds__(EW_THREADS) void scatter_sub_kernel(
    int n, const cplx* __restrict__ blk, int blk_stride, const int* __restrict__ d_nc,
    const int* __restrict__ d_blk_off, const int* __restrict__ d_inds_off,
    const int* __restrict__ d_inds, cplx* __restrict__ A)
{
    const int c = blockIdx.x, b = blockIdx.y;
    const int nc = d_nc[c];
    const int* inds = d_inds + d_inds_off[c];
    const cplx* src = blk + (size_t)b * blk_stride + d_blk_off[c];
    cplx* dst = A + (size_t)b * n * n;
    for (int t = threadIdx.x; t < nc * nc; t += EW_THREADS) {
        const int i = t / nc, j = t - i * nc;
        const size_t o = (size_t)inds[i] * n + inds[j];
        dst[o] = csub(dst[o], src[t]);
    }
}

void launch_assemble(hipStream_t st, int n, int nb, const cplx* E, const cplx* S, const cplx* H,
                     const cplx* sig_dense, const cplx* blk, int blk_stride, int n_contacts,
                     const int* d_nc, const int* d_blk_off, const int* d_inds_off,
                     const int* d_inds, cplx* A)
{
    const int n2 = n * n;
    int gx = (n2 + EW_THREADS - 1) / EW_THREADS;
    if (gx > 64) gx = 64;                          // grid-stride the rest
    hipLaunchKernelGGL(assemble_kernel, dim3(gx, nb), dim3(EW_THREADS), 0, st, n2, E, S, H,
                       sig_dense, A);
    if (blk && n_contacts > 0) {
        // contacts may share orbitals, so blocks are subtracted one contact at a
        // time (stream order) instead of racing on the same element
        for (int c = 0; c < n_contacts; ++c)
            hipLaunchKernelGGL(scatter_sub_kernel, dim3(1, nb), dim3(EW_THREADS), 0, st, n, blk,
                               blk_stride, d_nc + c, d_blk_off + c, d_inds_off + c, d_inds, A);
    }
}

// out_b = scatter-add of contact blocks into a zeroed n x n matrix
__global__ __launch_bounds__(EW_THREADS) void zero_kernel(size_t count, cplx* __restrict__ p)
{
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < count;
         i += (size_t)gridDim.x * EW_THREADS)
        p[i] = cmake(0.0, 0.0);
}

__global__ __launch_bounds__(EW_THREADS) void scatter_add_kernel(
    int n, const cplx* __restrict__ blk, int blk_stride, const int* __restrict__ d_nc,
    const int* __restrict__ d_blk_off, const int* __restrict__ d_inds_off,
    const int* __restrict__ d_inds, cplx* __restrict__ out)
{
    const int c = blockIdx.x, b = blockIdx.y;
    const int nc = d_nc[c];
    const int* inds = d_inds + d_inds_off[c];
    const cplx* src = blk + (size_t)b * blk_stride + d_blk_off[c];
    cplx* dst = out + (size_t)b * n * n;
    for (int t = threadIdx.x; t < nc * nc; t += EW_THREADS) {
        const int i = t / nc, j = t - i * nc;
        const size_t o = (size_t)inds[i] * n + inds[j];
        dst[o] = cadd(dst[o], src[t]);
    }
}

void launch_scatter_blocks(hipStream_t st, int n, int nb, const cplx* blk, int blk_stride,
                           int n_contacts, const int* d_nc, const int* d_blk_off,
                           const int* d_inds_off, const int* d_inds, int contact, cplx* out)
{
    const size_t count = (size_t)nb * n * n;
    int g = (int)((count + EW_THREADS - 1) / EW_THREADS);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(zero_kernel, dim3(g), dim3(EW_THREADS), 0, st, count, out);
    const int c0 = contact < 0 ? 0 : contact;
    const int c1 = contact < 0 ? n_contacts : contact + 1;
    for (int c = c0; c < c1; ++c)
        hipLaunchKernelGGL(scatter_add_kernel, dim3(1, nb), dim3(EW_THREADS), 0, st, n, blk,
                           blk_stride, d_nc + c, d_blk_off + c, d_inds_off + c, d_inds, out);
}

// ------------------------------------------------------------------- gamma
// Gamma = i (Sigma - Sigma^H)   (integrate.py:80, transport.py:146)
__global__ __launch_bounds__(EW_THREADS) void gamma_kernel(
    int n, const cplx* __restrict__ sig, size_t stride_sig, cplx* __restrict__ gam)
{
    const int b = blockIdx.y;
    const cplx* s = sig + (size_t)b * stride_sig;
    cplx* g = gam + (size_t)b * n * n;
    const int n2 = n * n;
    for (int t = blockIdx.x * EW_THREADS + threadIdx.x; t < n2; t += gridDim.x * EW_THREADS) {
        const int i = t / n, j = t - i * n;
        const cplx a = s[t];
        const cplx bt = s[(size_t)j * n + i];
        // d = a - conj(bt); i*d = (-d.y, d.x)
        const double dx = a.x - bt.x, dy = a.y + bt.y;
        g[t] = cmake(-dy, dx);
    }
}

void launch_gamma_dense(hipStream_t st, int n, int nb, const cplx* sig, size_t stride_sig, cplx* gam)
{
    int gx = (n * n + EW_THREADS - 1) / EW_THREADS;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gamma_kernel, dim3(gx, nb), dim3(EW_THREADS), 0, st, n, sig, stride_sig, gam);
}

// -------------------------------------------------------------- compact Gamma
// The coupling matrices of block-structured self-energies vanish outside the contact orbitals:
//     Gamma = scatter of a small K x K matrix over an index list idx[0..K).
// G Gamma G^H and Tr[Gamma_L G Gamma_R G^H] then only need the columns / the block of G on those
// indices (integrate.py:74-82 and transport.py:150-163 multiply the dense n x n matrices).
//
// small Gamma of the contacts [c0, c1): block diagonal, block c at its position in the
// concatenated index list, i (B - B^H) of the contact block B (stride 0: constant blocks).
__global__ __launch_bounds__(EW_THREADS) void gamma_small_kernel(
    int K, int c0, int c1, const int* __restrict__ d_nc, const int* __restrict__ d_blk_off,
    const int* __restrict__ d_inds_off, const cplx* __restrict__ blk, size_t blk_stride,
    cplx* __restrict__ out, size_t out_stride)
{
    const int b = blockIdx.y;
    const cplx* B = blk + (size_t)b * blk_stride;
    cplx* O = out + (size_t)b * out_stride;
    const int base = d_inds_off[c0];
    for (int t = blockIdx.x * EW_THREADS + threadIdx.x; t < K * K; t += gridDim.x * EW_THREADS) {
        const int a = t / K, e = t - a * K;
        cplx v = cmake(0.0, 0.0);
        for (int c = c0; c < c1; ++c) {
            const int o = d_inds_off[c] - base, k = d_nc[c];
            if (a >= o && a < o + k && e >= o && e < o + k) {
                const cplx x = B[d_blk_off[c] + (a - o) * k + (e - o)];
                const cplx y = B[d_blk_off[c] + (e - o) * k + (a - o)];
                // i (x - conj(y))
                v = cmake(-(x.y + y.y), x.x - y.x);
            }
        }
        O[t] = v;
    }
}

void launch_gamma_small(hipStream_t st, int K, int c0, int c1, int nb, const int* d_nc, const int* d_blk_off,
                        const int* d_inds_off, const cplx* blk, size_t blk_stride, cplx* out, size_t out_stride)
{
    int gx = (K * K + EW_THREADS - 1) / EW_THREADS;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(gamma_small_kernel, dim3(gx, nb), dim3(EW_THREADS), 0, st, K, c0, c1, d_nc, d_blk_off,
                       d_inds_off, blk, blk_stride, out, out_stride);
}

// out[b][a][e] = G[b][ridx ? ridx[a] : a][cidx[e]]     (nr x nc, leading dimension nc)
__global__ __launch_bounds__(EW_THREADS) void gather_block_kernel(
    int n, int nr, int nc, const cplx* __restrict__ G, size_t strideG, const int* __restrict__ ridx,
    const int* __restrict__ cidx, cplx* __restrict__ out, size_t out_stride)
{
    const int b = blockIdx.y;
    const cplx* Gb = G + (size_t)b * strideG;
    cplx* O = out + (size_t)b * out_stride;
    for (int t = blockIdx.x * EW_THREADS + threadIdx.x; t < nr * nc; t += gridDim.x * EW_THREADS) {
        const int a = t / nc, e = t - a * nc;
        O[t] = Gb[(size_t)(ridx ? ridx[a] : a) * n + cidx[e]];
    }
}

void launch_gather_block(hipStream_t st, int n, int nr, int nc, int nb, const cplx* G, size_t strideG,
                         const int* ridx, const int* cidx, cplx* out, size_t out_stride)
{
    int gx = (nr * nc + EW_THREADS - 1) / EW_THREADS;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gather_block_kernel, dim3(gx, nb), dim3(EW_THREADS), 0, st, n, nr, nc, G, strideG, ridx,
                       cidx, out, out_stride);
}

// out = a + b (elementwise): F + Sigma_tot of the constant providers when the resident system changes
__global__ __launch_bounds__(EW_THREADS) void cadd_kernel(size_t count, const cplx* __restrict__ a, const cplx* __restrict__ b, cplx* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i < count) out[i] = cadd(a[i], b[i]);
}

void launch_cadd(hipStream_t st, size_t count, const cplx* a, const cplx* b, cplx* out)
{
    if (count == 0) return;
    hipLaunchKernelGGL(cadd_kernel, dim3((unsigned)((count + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, st, count, a, b, out);
}

// -------------------------------------------------------------- accumulate
// acc[i] += sum_b w[b] * X[b][i] in a FIXED order (bitwise reproducible from run to run):
// the batch is cut into chunks of ACC_CHUNK energies; pass 1 reduces each chunk into
// part[chunk][i] (b ascending inside the chunk, 4 independent loads in flight per thread),
// pass 2 adds the chunk sums in ascending chunk order.  Enough workgroups to stream the
// batch at HBM rate (a single pass over i alone launches only n^2/256 workgroups).
static constexpr int ACC_CHUNK = 32;

// chunk table of a launch that serves SEVERAL sums at once (the segments of negf_gr_int_seg inside one workspace batch): chunk c
// covers the matrices [b0[c], b1[c]) and writes partial record c; segment s owns the chunks [first[s], first[s + 1]) and adds
// them, in order, to out + slot[s] * n2 -- the chunks of each segment are what launch_accumulate[_perm] would cut for it alone,
// so every sum keeps its bits; nchunks == 0: the regular cut of one sum (blockIdx.y * ACC_CHUNK).
static constexpr int ACC_TAB_CHUNKS = 96, ACC_TAB_SEGS = 32;
struct AccChunks { int nchunks, nseg; int b0[ACC_TAB_CHUNKS], b1[ACC_TAB_CHUNKS]; int first[ACC_TAB_SEGS + 1], slot[ACC_TAB_SEGS]; };

__global__ __launch_bounds__(EW_THREADS) void accumulate_partial_kernel(
    int n2, int nb, const cplx* __restrict__ w, const cplx* __restrict__ X, cplx* __restrict__ part, const AccChunks tab)
{
    const int i = blockIdx.x * EW_THREADS + threadIdx.x;
    const int b0 = tab.nchunks ? tab.b0[blockIdx.y] : blockIdx.y * ACC_CHUNK;
    const int b1 = tab.nchunks ? tab.b1[blockIdx.y] : min(nb, b0 + ACC_CHUNK);
    if (i >= n2) return;
    cplx a = cmake(0.0, 0.0);
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        const cplx x0 = X[(size_t)(b + 0) * n2 + i], x1 = X[(size_t)(b + 1) * n2 + i];
        const cplx x2 = X[(size_t)(b + 2) * n2 + i], x3 = X[(size_t)(b + 3) * n2 + i];
        a = cfma(a, w[b + 0], x0); a = cfma(a, w[b + 1], x1);
        a = cfma(a, w[b + 2], x2); a = cfma(a, w[b + 3], x3);
    }
    for (; b < b1; ++b) a = cfma(a, w[b], X[(size_t)b * n2 + i]);
    part[(size_t)blockIdx.y * n2 + i] = a;
}

__global__ __launch_bounds__(EW_THREADS) void accumulate_final_kernel(
    int n2, int nchunks, const cplx* __restrict__ part, cplx* __restrict__ acc)
{
    const int i = blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n2) return;
    cplx a = acc[i];
    for (int c = 0; c < nchunks; ++c) a = cadd(a, part[(size_t)c * n2 + i]);
    acc[i] = a;
}

__global__ __launch_bounds__(EW_THREADS) void accumulate_final_seg_kernel(
    int n2, const AccChunks tab, const cplx* __restrict__ part, cplx* __restrict__ out)
{
    const int i = blockIdx.x * EW_THREADS + threadIdx.x, sg = blockIdx.y;
    if (i >= n2) return;
    cplx* acc = out + (size_t)tab.slot[sg] * n2;
    cplx a = acc[i];
    for (int c = tab.first[sg]; c < tab.first[sg + 1]; ++c) a = cadd(a, part[(size_t)c * n2 + i]);
    acc[i] = a;
}

// The same reduction over matrices that the windowed inverse left in their reduced form (rows never moved: G[i][j] =
// W[pivrow[i]][colof[j]]): the weighted sum of GrInt needs no G, so the gather pass (read + write of every matrix) and the
// re-read by accumulate_partial_kernel become ONE read through the permutation.  Same chunks, same order of additions.
__global__ __launch_bounds__(EW_THREADS) void accumulate_perm_partial_kernel(
    int n, int nb, const cplx* __restrict__ w, const cplx* __restrict__ W, const int* __restrict__ piv,
    const int* __restrict__ info, cplx* __restrict__ part)
{
    const int n2 = n * n;
    const int i = blockIdx.x * EW_THREADS + threadIdx.x;
    const int b0 = blockIdx.y * ACC_CHUNK, b1 = min(nb, b0 + ACC_CHUNK);
    if (i >= n2) return;
    const int r = i / n, cj = i - r * n;
    const double qnan = __builtin_nan("");
    cplx a = cmake(0.0, 0.0);
    auto elem = [&](int b) {
        if (info[b] != 0) return cmake(qnan, qnan);                       // (uniform; a dead matrix's bookkeeping is not an index)
        const int* pb = piv + (size_t)b * 2 * n;
        return W[(size_t)b * n2 + (size_t)pb[r] * n + pb[n + cj]];
    };
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        const cplx x0 = elem(b), x1 = elem(b + 1), x2 = elem(b + 2), x3 = elem(b + 3);
        a = cfma(a, w[b + 0], x0); a = cfma(a, w[b + 1], x1);
        a = cfma(a, w[b + 2], x2); a = cfma(a, w[b + 3], x3);
    }
    for (; b < b1; ++b) a = cfma(a, w[b], elem(b));
    part[(size_t)blockIdx.y * n2 + i] = a;
}

// Row form (n <= 2048): a workgroup owns ONE output row i of one chunk of energies.  Row pivrow_b[i] of matrix b is read as what it
// is -- n contiguous elements -- into LDS (double buffered: the next matrix's row is in flight while this one is used) and the
// column permutation is resolved there: out[i][j] += w_b row[colof_b[j]].  The element form above reads 2.2 x the bytes (PMC: 8.9
// GB per 1000 x n = 500 against 4.0) because its gathers use a quarter or half of every cache line they touch per wave.
static constexpr int APR_MAXN = 2048;
__global__ __launch_bounds__(EW_THREADS) void accumulate_perm_rows_kernel(
    int n, int nb, const cplx* __restrict__ w, const cplx* __restrict__ W, const int* __restrict__ piv,
    const int* __restrict__ info, cplx* __restrict__ part, const AccChunks tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char apr_raw[];
    cplx* rowbuf = reinterpret_cast<cplx*>(apr_raw);                      // [2][n]
    const int i = blockIdx.x, n2 = n * n;
    const int b0 = tab.nchunks ? tab.b0[blockIdx.y] : blockIdx.y * ACC_CHUNK;
    const int b1 = tab.nchunks ? tab.b1[blockIdx.y] : min(nb, b0 + ACC_CHUNK);
    const int tid = threadIdx.x;
    constexpr int CPT = APR_MAXN / EW_THREADS;                            // columns per thread
    const double qnan = __builtin_nan("");
    cplx acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = cmake(0.0, 0.0);
    auto fetch = [&](int b, int buf) {
        if (info[b] != 0) return;                                         // (uniform; a dead matrix's bookkeeping is not an index)
        const cplx* src = W + (size_t)b * n2 + (size_t)piv[(size_t)b * 2 * n + i] * n;
        for (int j = tid; j < n; j += EW_THREADS) rowbuf[buf * n + j] = src[j];
    };
    fetch(b0, 0);
    for (int b = b0; b < b1; ++b) {
        const int buf = (b - b0) & 1;
        __syncthreads();                                                  // row b is in LDS; everybody is done with the other buffer
        if (b + 1 < b1) fetch(b + 1, buf ^ 1);
        const bool dead = info[b] != 0;
        const int* cf = piv + (size_t)b * 2 * n + n;
        const cplx wb = w[b];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int j = tid + c * EW_THREADS;
            if (j < n) {
                const cplx x = dead ? cmake(qnan, qnan) : rowbuf[buf * n + cf[j]];
                acc[c] = cfma(acc[c], wb, x);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int j = tid + c * EW_THREADS;
        if (j < n) part[(size_t)blockIdx.y * n2 + (size_t)i * n + j] = acc[c];
    }
}

void launch_accumulate_perm(hipStream_t st, int n, int nb, const cplx* w, const cplx* W, const int* piv, const int* info, cplx* acc, cplx* part)
{
    const int n2 = n * n;
    const int g = (n2 + EW_THREADS - 1) / EW_THREADS;
    const int nchunks = (nb + ACC_CHUNK - 1) / ACC_CHUNK;
    static int rows = -1;
    if (rows < 0) { const char* e = getenv("NEGF_ACC_PERM_ROWS"); rows = e ? atoi(e) : 1; }
    if (rows && n <= APR_MAXN)
        hipLaunchKernelGGL(accumulate_perm_rows_kernel, dim3(n, nchunks), dim3(EW_THREADS), (size_t)2 * n * sizeof(cplx), st, n, nb, w, W, piv, info, part, AccChunks{});
    else
        hipLaunchKernelGGL(accumulate_perm_partial_kernel, dim3(g, nchunks), dim3(EW_THREADS), 0, st, n, nb, w, W, piv, info, part);
    hipLaunchKernelGGL(accumulate_final_kernel, dim3(g), dim3(EW_THREADS), 0, st, n2, nchunks, part, acc);
}

// Several sums over sub-ranges of ONE workspace batch in one launch pair: range r = matrices [lo[r], hi[r]) of the batch (w, info,
// X / W, piv all indexed from the batch's first matrix) is added to out + slot[r] * n2.  `perm`: the matrices are the windowed
// inverse's reduced form (W, piv) instead of G (X).  Returns false when the table does not hold the request (the caller then
// launches range by range).  `part` must hold one record per chunk (at most one per matrix of the batch).
bool launch_accumulate_ranges(hipStream_t st, int n, bool perm, const cplx* w, const cplx* XW, const int* piv, const int* info,
                              int nranges, const int* lo, const int* hi, const int* slot, cplx* out, cplx* part)
{
    static int rows = -1;
    if (rows < 0) { const char* e = getenv("NEGF_ACC_PERM_ROWS"); rows = e ? atoi(e) : 1; }
    if (nranges > ACC_TAB_SEGS || (perm && !(rows && n <= APR_MAXN))) return false;
    AccChunks tab{};
    for (int r = 0; r < nranges; ++r) {
        tab.first[r] = tab.nchunks; tab.slot[r] = slot[r];
        for (int b = lo[r]; b < hi[r]; b += ACC_CHUNK) {
            if (tab.nchunks == ACC_TAB_CHUNKS) return false;
            tab.b0[tab.nchunks] = b; tab.b1[tab.nchunks] = std::min(hi[r], b + ACC_CHUNK); ++tab.nchunks;
        }
    }
    tab.first[nranges] = tab.nchunks; tab.nseg = nranges;
    if (tab.nchunks == 0) return true;
    const int n2 = n * n;
    const int g = (n2 + EW_THREADS - 1) / EW_THREADS;
    if (perm)
        hipLaunchKernelGGL(accumulate_perm_rows_kernel, dim3(n, tab.nchunks), dim3(EW_THREADS), (size_t)2 * n * sizeof(cplx), st, n, 0, w, XW, piv, info, part, tab);
    else
        hipLaunchKernelGGL(accumulate_partial_kernel, dim3(g, tab.nchunks), dim3(EW_THREADS), 0, st, n2, 0, w, XW, part, tab);
    hipLaunchKernelGGL(accumulate_final_seg_kernel, dim3(g, nranges), dim3(EW_THREADS), 0, st, n2, tab, part, out);
    return true;
}

// `part` must hold ceil(nb / ACC_CHUNK) * n2 elements
size_t accumulate_scratch_elems(int n2, int nb) { return (size_t)((nb + ACC_CHUNK - 1) / ACC_CHUNK) * n2; }

void launch_accumulate(hipStream_t st, int n2, int nb, const cplx* w, const cplx* X, cplx* acc, cplx* part)
{
    const int g = (n2 + EW_THREADS - 1) / EW_THREADS;
    const int nchunks = (nb + ACC_CHUNK - 1) / ACC_CHUNK;
    hipLaunchKernelGGL(accumulate_partial_kernel, dim3(g, nchunks), dim3(EW_THREADS), 0, st, n2, nb, w, X, part, AccChunks{});
    hipLaunchKernelGGL(accumulate_final_kernel, dim3(g), dim3(EW_THREADS), 0, st, n2, nchunks, part, acc);
}

// ------------------------------------------------------------------ nested refinement (density.py:239-268)
// One workgroup per integral.  sums [levels][n2]: the weighted sums over the NEW nodes of consecutive levels of the nested rule;
// the reference's update and stopping test, level by level and in its operation order:
//     first level of an integration (ratio = NaN):  P = sum                                        (no test)
//     every other level:  new_P = P * ratio;  new_P += sum;  maxDP = max |new_P - P|;  P = new_P;  stop when maxDP < tol
// (P * ratio is numpy's complex128 x (ratio + 0j) product; |.| = hypot).  P [n2] holds the running value on entry (levels with a
// ratio) and the value of the last level consumed on exit; level_out = index of the level that converged, -1 if none did;
// maxdp_out [levels] the maxDP of every level consumed (NaN where there is none).  A NaN anywhere makes maxDP NaN (numpy's max),
// which never converges -- as in the reference.
__global__ __launch_bounds__(1024) void refine_levels_kernel(int n2, const cplx* __restrict__ sums, const int* __restrict__ first,
                                                             const double* __restrict__ ratio, double tol, cplx* __restrict__ P,
                                                             int* __restrict__ level_out, double* __restrict__ maxdp_out)
{
    __shared__ double red[16];
    __shared__ int red_nan[16];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int s0 = first[k], s1 = first[k + 1];
    cplx* Pk = P + (size_t)k * n2;
    const double qnan = __builtin_nan("");
    int conv = -1;
    for (int s = s0; s < s1; ++s) {
        const cplx* inc = sums + (size_t)s * n2;
        const double r = ratio[s];
        if (r != r) {                                             // first level of the integration
            for (int i = tid; i < n2; i += 1024) Pk[i] = inc[i];
            if (tid == 0) maxdp_out[s] = qnan;
            __syncthreads();
            continue;
        }
        double mx = 0.0;
        int any_nan = 0;
        for (int i = tid; i < n2; i += 1024) {
            const cplx p = Pk[i];
            // (p.x + i p.y)(r + i 0), every term as numpy forms it
            cplx q = cmake(p.x * r - p.y * 0.0, p.x * 0.0 + p.y * r);
            const cplx v = inc[i];
            q = cmake(q.x + v.x, q.y + v.y);
            const double a = hypot(q.x - p.x, q.y - p.y);
            if (a != a) any_nan = 1; else mx = a > mx ? a : mx;
            Pk[i] = q;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(mx, off, 64);
            mx = o > mx ? o : mx;
            any_nan |= __shfl_down(any_nan, off, 64);
        }
        if ((tid & 63) == 0) { red[tid >> 6] = mx; red_nan[tid >> 6] = any_nan; }
        __syncthreads();
        double m_all = red[0];
        int n_all = red_nan[0];
        for (int q = 1; q < 16; ++q) { m_all = red[q] > m_all ? red[q] : m_all; n_all |= red_nan[q]; }
        const double maxdp = n_all ? qnan : m_all;
        if (tid == 0) maxdp_out[s] = maxdp;
        __syncthreads();                                          // red is reused by the next level
        if (maxdp < tol) { conv = s - s0; break; }                // (uniform)
    }
    if (tid == 0) {
        level_out[k] = conv;
        for (int s = s0 + (conv < 0 ? s1 - s0 : conv + 1); s < s1; ++s) maxdp_out[s] = qnan;   // levels not consumed
    }
}

// The same for matrices too large for one workgroup per integration (n > REF_MAX_N): one level = one launch of many
// workgroups (update, block maxima folded into a 64-bit atomic maximum of the bit pattern -- |.| >= 0 orders like its bits) and
// a one-thread decision kernel; a level launched after the integration has converged returns at once (conv >= 0), so the
// whole walk is queued on the stream without a host round trip.
__global__ __launch_bounds__(256) void refine_wide_kernel(int n2, const cplx* __restrict__ inc, double r, cplx* __restrict__ Pk,
                                                          const int* __restrict__ conv, unsigned long long* __restrict__ maxbits,
                                                          int* __restrict__ nanflag)
{
    if (*conv >= 0) return;
    __shared__ double red[4];
    __shared__ int red_nan[4];
    double mx = 0.0;
    int any_nan = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
        const cplx p = Pk[i];
        cplx q = cmake(p.x * r - p.y * 0.0, p.x * 0.0 + p.y * r);
        const cplx v = inc[i];
        q = cmake(q.x + v.x, q.y + v.y);
        const double a = hypot(q.x - p.x, q.y - p.y);
        if (a != a) any_nan = 1; else mx = a > mx ? a : mx;
        Pk[i] = q;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off, 64);
        mx = o > mx ? o : mx;
        any_nan |= __shfl_down(any_nan, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; red_nan[threadIdx.x >> 6] = any_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) { mx = red[q] > mx ? red[q] : mx; any_nan |= red_nan[q]; }
        atomicMax(maxbits, (unsigned long long)__double_as_longlong(mx));
        if (any_nan) atomicOr(nanflag, 1);
    }
}

__global__ void refine_decide_kernel(int level_idx, double tol, const unsigned long long* __restrict__ maxbits,
                                     const int* __restrict__ nanflag, int* __restrict__ conv, double* __restrict__ maxdp_slot)
{
    const double qnan = __builtin_nan("");
    if (*conv >= 0) { *maxdp_slot = qnan; return; }               // level not consumed
    const double maxdp = *nanflag ? qnan : __longlong_as_double((long long)*maxbits);
    *maxdp_slot = maxdp;
    if (maxdp < tol) *conv = level_idx;
}

void launch_refine_level_wide(hipStream_t st, int n2, const cplx* inc, double ratio, cplx* Pk, int level_idx, double tol,
                              int* conv, unsigned long long* maxbits, int* nanflag, double* maxdp_slot)
{
    const int grid = std::min(2048, (n2 + 255) / 256);
    hipLaunchKernelGGL(refine_wide_kernel, dim3(grid), dim3(256), 0, st, n2, inc, ratio, Pk, conv, maxbits, nanflag);
    hipLaunchKernelGGL(refine_decide_kernel, dim3(1), dim3(1), 0, st, level_idx, tol, maxbits, nanflag, conv, maxdp_slot);
}

void launch_refine_levels(hipStream_t st, int n2, int nint, const cplx* sums, const int* first, const double* ratio, double tol,
                          cplx* P, int* level_out, double* maxdp_out)
{
    hipLaunchKernelGGL(refine_levels_kernel, dim3(nint), dim3(1024), 0, st, n2, sums, first, ratio, tol, P, level_out, maxdp_out);
}

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// one workgroup per energy: Re sum X_ij conj(G_ij) = sum (Xr Gr + Xi Gi)
__global__ __launch_bounds__(EW_THREADS) void trace_dot_kernel(
    int nr, int ncol, const cplx* __restrict__ X, int ldx, size_t strideX,
    const cplx* __restrict__ G, int ldg, size_t strideG, double* __restrict__ out, int out_stride)
{
    __shared__ double part[EW_THREADS / 64];
    const int b = blockIdx.x;
    const cplx* x = X + (size_t)b * strideX;
    const cplx* g = G + (size_t)b * strideG;
    double s = 0.0;
    const int total = nr * ncol;
    for (int t = threadIdx.x; t < total; t += EW_THREADS) {
        const int i = t / ncol, j = t - i * ncol;
        const cplx a = x[(size_t)i * ldx + j];
        const cplx c = g[(size_t)i * ldg + j];
        s += a.x * c.x + a.y * c.y;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < EW_THREADS / 64; ++k) t += part[k];
        out[(size_t)b * out_stride] = t;
    }
}

void launch_trace_dot(hipStream_t st, int nr, int ncol, int nb, const cplx* X, int ldx,
                      size_t strideX, const cplx* G, int ldg, size_t strideG, double* out,
                      int out_stride)
{
    hipLaunchKernelGGL(trace_dot_kernel, dim3(nb), dim3(EW_THREADS), 0, st, nr, ncol, X, ldx,
                       strideX, G, ldg, strideG, out, out_stride);
}

__global__ __launch_bounds__(EW_THREADS) void dos_kernel(
    int n, const cplx* __restrict__ G, double* __restrict__ dos_tot, double* __restrict__ dos_site)
{
    __shared__ double part[EW_THREADS / 64];
    const int b = blockIdx.x;
    const cplx* g = G + (size_t)b * n * n;
    const double pi = 3.14159265358979323846;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += EW_THREADS) {
        const double d = -g[(size_t)i * n + i].y / pi;      // same rounding as -imag/np.pi
        if (dos_site) dos_site[(size_t)b * n + i] = d;
        s += d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < EW_THREADS / 64; ++k) t += part[k];
        dos_tot[b] = t;
    }
}

void launch_dos(hipStream_t st, int n, int nb, const cplx* G, double* dos_tot, double* dos_site)
{
    hipLaunchKernelGGL(dos_kernel, dim3(nb), dim3(EW_THREADS), 0, st, n, G, dos_tot, dos_site);
}
