// Batched complex128 GEMM on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
//   C_b (M x N) = A_b (M x K) * op(B_b),   op(B) = B  or  B^H
//
// used for the dense products of the lesser Green's function G Gamma G^H
// (integrate.py:81) and of the transmission Gamma_L G Gamma_R (transport.py:156,
// 176).  A complex product is THREE real MFMA chains per 16x16 tile ("3M"):
//   S1 += Ar*Br ; S2 += Ai*Bi ; S3 += (Ar+Ai)*(Br+Bi)   =>   Cr = S1 - S2 ,  Ci = S3 - S1 - S2
// (one addition per operand fragment and k-step instead of a fourth matrix instruction; the FP64 matrix
// instruction holds its SIMD's vector issue for most of its 64 cycles, so matrix-pipe time is the kernel's time;
// the result differs from the four-product form by a few ulp of |A||B|, five orders inside the 1e-8 bar).
// Fragment layout of v_mfma_f64_16x16x4_f64 (guide section 3):
//   A operand: lane l holds A[i = l&15][k = l>>4]       (one f64 per lane)
//   B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D      : lane l, register r holds C[row = (l>>4) + 4r][col = l&15]
// Interleaved (re,im) storage means ONE 16-byte LDS read gives a lane both the
// real and the imaginary operand of its element.
//
// opB = 3: op(B) = B^H AND the caller states that the product is Hermitian (G Gamma G^H with Gamma = i (Sigma - Sigma^H),
// integrate.py:79-81: X G^H with X = G Gamma): only the block tiles on and above the diagonal are computed, every block
// above it also stores its conjugate transpose -- the same sums, each taken once (n = 1000: 136 of 256 block tiles).
//
// opB bit 4: the result is stored CONJUGATE-TRANSPOSED, C = (A op(B))^H (N x M, leading dimension ldc), every 16 x 16
// sub-tile transposed through a per-wave LDS patch so that the stores stay 256-byte runs.  The products X G^H of
// G Gamma G^H and of the transmission are taken as G X^H with X^H = (G Gamma)^H written this way by the first product
// (Gamma Hermitian): the K-tile of a PLAIN second operand is 16 rows x 1 KB and mostly served by the compute unit's L1,
// the K-tile of an operand conjugate-transposed on the fly is 64 rows x 256 B and costs the L2 twice the requests
// (n = 1000, 256 products: 25.8 against 31.3 ms).
//
// Workgroup = 256 threads = 4 waves; block tile 64 x 64, wave tile 32 x 32
// (2 x 2 MFMA tiles, 64 accumulator VGPRs), K tile 16 staged through LDS.
#include "negf_common.h"
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));

static constexpr int ZG_BM = 64, ZG_BN = 64, ZG_BK = 16;
static constexpr int ZG_APITCH = ZG_BK + 1;     // odd pitch: 16 rows -> 16 distinct 16-B slots
static constexpr int ZG_BPITCH = ZG_BN + 1;

// Hermitian form: the 1-D grid enumerates the upper triangle of the T x T block tiles of nb matrices so that the XCDs
// (workgroup index mod 8) get equal shares AND whole block columns: columns c and T-1-c form a pair of c+1 and T-c =
// T+1 blocks, pair gp = z * P + p (P = ceil(T/2) pairs per matrix) goes to XCD gp mod 8, and an XCD walks down a pair's
// blocks in turn -- the B operand of a block column stays in that XCD's L2.  the batch size comes as a kernel argument (blockIdx.z is 0);
// (x, y) of a full T x T grid with the lower blocks returning at once loaded the XCDs unevenly: slower than the full product.
__device__ __forceinline__ bool zg_herm_decode(int T, int nb, int L, int* by, int* bx, int* b)
{
    if (T & 1) {
        // an odd number of block columns leaves the middle one without a partner (half a pair's work: XCDs 2 : 1 at
        // T = 3); there the triangle is simply enumerated row by row, matrix after matrix
        const int ntri = T * (T + 1) / 2;
        const int z = L / ntri;
        int t = L - z * ntri, y = 0;
        if (z >= nb) return false;
        while (t >= T - y) { t -= T - y; ++y; }
        *by = y; *bx = y + t; *b = z;
        return true;
    }
    const int P = (T + 1) >> 1;
    const int xcd = L & 7, slot = L >> 3;
    const int q = slot / (T + 1), r = slot - q * (T + 1);
    const int gp = xcd + 8 * q;
    const int z = gp / P, p = gp - z * P;
    if (z >= nb) return false;
    const int c1 = p, c2 = T - 1 - p;
    if (r <= c1) { *by = r; *bx = c1; }
    else if (c1 != c2) { *by = r - (c1 + 1); *bx = c2; }
    else return false;
    *b = z;
    return true;
}
static inline unsigned zg_herm_grid(long T, long nb)
{
    if (T & 1) return (unsigned)(T * (T + 1) / 2 * nb);
    const long P = (T + 1) / 2;
    return (unsigned)(8 * ((P * nb + 7) / 8) * (T + 1));
}

__global__ __launch_bounds__(256, 3) void zgemm_mfma_kernel(
    int M, int N, int K,
    const cplx* __restrict__ Aall, int lda, size_t strideA,
    const cplx* __restrict__ Ball, int ldb, size_t strideB, int opB_arg,
    cplx* __restrict__ Call, int ldc, size_t strideC, int nbatch)
{
    __shared__ cplx As[ZG_BM * ZG_APITCH];      // As[i][k]
    __shared__ cplx Bs[ZG_BK * ZG_BPITCH];      // Bs[k][j]  (already op()'ed)

    const bool herm = (opB_arg & 2) != 0;       // Hermitian product: blocks below the diagonal are mirrored, not computed
    const bool store_t = (opB_arg & 4) != 0;    // the result is stored conjugate-transposed (not with herm)
    const int opB = opB_arg & 1;
    // Hermitian form: the grid is the upper triangle itself, row by row (gridDim.x = T (T + 1) / 2 block tiles) -- a
    // full T x T grid with the lower blocks returning at once loads the XCDs (block index mod 8) unevenly and ran
    // SLOWER than the full product (n = 1000: 67 ms against 52)
    int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
    if (herm && !zg_herm_decode((N + ZG_BN - 1) / ZG_BN, nbatch, (int)blockIdx.x, &by, &bx, &b)) return;
    const bool mirror = herm && by < bx;
    const cplx* A = Aall + (size_t)b * strideA;
    const cplx* B = Ball + (size_t)b * strideB;
    cplx* C = Call + (size_t)b * strideC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = by * ZG_BM, col0 = bx * ZG_BN;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;   // wave tile origin in the block tile
    const int fi = lane & 15, fk = lane >> 4;

    d4 s1[2][2], s2[2][2], s3[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) { s1[a][c] = (d4){0, 0, 0, 0}; s2[a][c] = (d4){0, 0, 0, 0}; s3[a][c] = (d4){0, 0, 0, 0}; }

    // edge blocks: sub-tiles that start beyond M or N do no work (n = 200 pads to 208, not 256)
    bool va[2], vc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) { va[a] = row0 + wr + a * 16 < M; vc[a] = col0 + wc + a * 16 < N; }

    // staging assignment: 64 rows x 16 cols, 4 consecutive elements per thread
    const int lr = tid >> 2, lc = (tid & 3) * 4;         // A tile (and B^H tile): row lr, cols lc..lc+3
    const int br = tid >> 4, bc = (tid & 15) * 4;        // B tile (opB=0): row br, cols bc..bc+3

    // global -> registers for the k-tile starting at k0 (zero fill outside the matrix); the tile
    // after the current one is fetched while the current one runs its MFMAs
    cplx ra[4], rb[4];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        const int gi = row0 + lr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gk = k0 + lc + e;
            ra[e] = (gi < M && gk < K) ? A[(size_t)gi * lda + gk] : cmake(0.0, 0.0);
        }
        if (opB == 0) {
            const int gk = k0 + br;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gj = col0 + bc + e;
                rb[e] = (gk < K && gj < N) ? B[(size_t)gk * ldb + gj] : cmake(0.0, 0.0);
            }
        } else {
            // op(B)[k][j] = conj(B[j][k]); B stored N x K
            const int gj = col0 + lr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gk = k0 + lc + e;
                rb[e] = (gj < N && gk < K) ? cconj(B[(size_t)gj * ldb + gk]) : cmake(0.0, 0.0);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += ZG_BK) {
        // ---- registers -> LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) As[lr * ZG_APITCH + lc + e] = ra[e];
        if (opB == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[br * ZG_BPITCH + bc + e] = rb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[(lc + e) * ZG_BPITCH + lr] = rb[e];
        }
        __syncthreads();
        if (k0 + ZG_BK < K) fetch(k0 + ZG_BK);
        // ---- 4 k-steps of 4
#pragma unroll
        for (int ks = 0; ks < ZG_BK; ks += 4) {
            cplx af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[(wr + a * 16 + fi) * ZG_APITCH + ks + fk];
#pragma unroll
            for (int c = 0; c < 2; ++c) bf[c] = Bs[(ks + fk) * ZG_BPITCH + wc + c * 16 + fi];
            const double as_[2] = {af[0].x + af[0].y, af[1].x + af[1].y};
            const double bs_[2] = {bf[0].x + bf[0].y, bf[1].x + bf[1].y};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (!(va[a] && vc[c])) continue;       // 16x16 sub-tile entirely outside the matrix (wave-uniform)
                    s1[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].x, bf[c].x, s1[a][c], 0, 0, 0);
                    s2[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].y, bf[c].y, s2[a][c], 0, 0, 0);
                    s3[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(as_[a], bs_[c], s3[a][c], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // ---- store: lane l, reg r -> row (l>>4) + 4r, col l&15 ; 16 lanes write 256 contiguous bytes
    if (!store_t) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = row0 + wr + a * 16 + fk + 4 * r;
                    const int gj = col0 + wc + c * 16 + fi;
                    if (gi < M && gj < N)
                        C[(size_t)gi * ldc + gj] = cmake(s1[a][c][r] - s2[a][c][r], s3[a][c][r] - s1[a][c][r] - s2[a][c][r]);
                }
    }
    if (mirror || store_t) {
        // the conjugate transpose of the block, transposed through LDS (the A staging area is free after the K loop:
        // one 16 x 17 patch per wave) so that 16 lanes again write 256 contiguous bytes -- written straight from the
        // accumulator layout the mirror image is 64-byte pieces in 16 different rows, and the two Hermitian products
        // of C5 took 67 ms instead of 52 for half the matrix work
        cplx* T = As + wave * (16 * ZG_APITCH);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    T[fi * ZG_APITCH + fk + 4 * r] = cmake(s1[a][c][r] - s2[a][c][r], -(s3[a][c][r] - s1[a][c][r] - s2[a][c][r]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // T[j][i] = conj(tile[i][j]); this lane reads row j = fk + 4r, column i = fi
                    const int gr = col0 + wc + c * 16 + fk + 4 * r;      // row of the mirror image
                    const int gc = row0 + wr + a * 16 + fi;              // its column
                    const cplx v = T[(fk + 4 * r) * ZG_APITCH + fi];
                    if (gr < N && gc < M) C[(size_t)gr * ldc + gc] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
    }
}

// ---- flexible-block version for shapes the 64 x 64 block tile pads badly (n = 200: 13 tiles of 16 per
// dimension fill 4 x 4 blocks of 64 only to 66 %, and an edge block costs almost what a full one does:
// measured 34 TF at n = 200 against 53 TF at n = 192).  A block covers tm x tn sub-tiles of 16 x 16 with
// tm, tn <= 5, the sub-tile counts of the blocks of a dimension differ by at most one (13 -> 5 + 4 + 4), so no
// block is mostly padding; the block's sub-tiles are dealt to 8 waves, up to 4 each (3M: 96 accumulator VGPRs;
// blocks of 7 x 7 sub-tiles -- 7 per wave, 168 accumulator VGPRs -- spilled and ran 40.3 TF at n = 200 against
// 43.5 TF), every sub-tile reads its own A and B fragment from LDS (one read each per 3 MFMAs).
static constexpr int ZF_MAXT = 5, ZF_THREADS = 512, ZF_WAVES = 8;
static constexpr int ZF_ROWS = ZF_MAXT * 16;                    // 80
static constexpr int ZF_APITCH = ZG_BK + 1, ZF_BPITCH = ZF_ROWS + 1;

__host__ __device__ inline void zf_block_range(int tiles, int nblk, int b, int* t0, int* tn)
{
    const int base = tiles / nblk, rem = tiles % nblk;
    *t0 = b * base + (b < rem ? b : rem);
    *tn = base + (b < rem ? 1 : 0);
}

__global__ __launch_bounds__(ZF_THREADS) void zgemm_flex_kernel(
    int M, int N, int K, int nbm, int nbn,
    const cplx* __restrict__ Aall, int lda, size_t strideA,
    const cplx* __restrict__ Ball, int ldb, size_t strideB, int opB_arg,
    cplx* __restrict__ Call, int ldc, size_t strideC, int nbatch)
{
    __shared__ cplx As[ZF_ROWS * ZF_APITCH];     // As[i][k]
    __shared__ cplx Bs[ZG_BK * ZF_BPITCH];       // Bs[k][j]  (already op()'ed)
    __shared__ cplx Ts[ZF_WAVES * 16 * 17];      // per-wave transpose patch of the Hermitian mirror image
    const bool herm = (opB_arg & 2) != 0;        // see zgemm_mfma_kernel (M == N: the row and column blocks coincide)
    const bool store_t = (opB_arg & 4) != 0;
    const int opB = opB_arg & 1;
    int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
    if (herm && !zg_herm_decode(nbn, nbatch, (int)blockIdx.x, &by, &bx, &b)) return;
    const bool mirror = herm && by < bx;
    const cplx* A = Aall + (size_t)b * strideA;
    const cplx* B = Ball + (size_t)b * strideB;
    cplx* C = Call + (size_t)b * strideC;
    int tr0, tm, tc0, tn;
    zf_block_range((M + 15) >> 4, nbm, by, &tr0, &tm);
    zf_block_range((N + 15) >> 4, nbn, bx, &tc0, &tn);
    const int row0 = tr0 * 16, col0 = tc0 * 16, rows = tm * 16, cols = tn * 16, ntiles = tm * tn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;

    d4 s1[ZF_MAXT], s2[ZF_MAXT], s3[ZF_MAXT];   // 3M accumulators (see the head of the file)
    int toffA[ZF_MAXT], toffB[ZF_MAXT];          // LDS offsets of the sub-tile's A rows / B columns
#pragma unroll
    for (int s = 0; s < ZF_MAXT; ++s) {
        s1[s] = (d4){0, 0, 0, 0}; s2[s] = (d4){0, 0, 0, 0}; s3[s] = (d4){0, 0, 0, 0};
        const int t = wave + s * ZF_WAVES;
        const int ti = t < ntiles ? t / tn : 0, tj = t < ntiles ? t - (t / tn) * tn : 0;
        toffA[s] = (ti * 16 + fi) * ZF_APITCH + fk;
        toffB[s] = fk * ZF_BPITCH + tj * 16 + fi;
    }
    // staging: A block rows x 16, B block 16 x cols; element e of a thread's share: index tid + e * 512
    constexpr int NE = (ZF_ROWS * ZG_BK + ZF_THREADS - 1) / ZF_THREADS;          // 4 (3.5 rounded up)
    cplx ra[NE], rb[NE];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * ZF_THREADS;
            {   // A: row idx / 16, k idx % 16
                const int r = idx >> 4, k = idx & 15, gi = row0 + r, gk = k0 + k;
                ra[e] = (r < rows && gi < M && gk < K) ? A[(size_t)gi * lda + gk] : cmake(0.0, 0.0);
            }
            if (opB == 0) {   // B[k][j]: k = idx / cols16, j = idx % cols16 with cols16 = 112
                const int k = idx / ZF_ROWS, j = idx - k * ZF_ROWS, gk = k0 + k, gj = col0 + j;
                rb[e] = (k < ZG_BK && j < cols && gk < K && gj < N) ? B[(size_t)gk * ldb + gj] : cmake(0.0, 0.0);
            } else {          // op(B)[k][j] = conj(B[j][k]); B stored N x K: j = idx / 16, k = idx % 16
                const int j = idx >> 4, k = idx & 15, gj = col0 + j, gk = k0 + k;
                rb[e] = (j < cols && gj < N && gk < K) ? cconj(B[(size_t)gj * ldb + gk]) : cmake(0.0, 0.0);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += ZG_BK) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * ZF_THREADS;
            if (idx < ZF_ROWS * ZG_BK) {
                As[(idx >> 4) * ZF_APITCH + (idx & 15)] = ra[e];
                if (opB == 0) { const int k = idx / ZF_ROWS, j = idx - k * ZF_ROWS; Bs[k * ZF_BPITCH + j] = rb[e]; }
                else Bs[(idx & 15) * ZF_BPITCH + (idx >> 4)] = rb[e];
            }
        }
        __syncthreads();
        if (k0 + ZG_BK < K) fetch(k0 + ZG_BK);
#pragma unroll
        for (int ks = 0; ks < ZG_BK; ks += 4) {
#pragma unroll
            for (int s = 0; s < ZF_MAXT; ++s) {
                if (wave + s * ZF_WAVES < ntiles) {              // wave-uniform
                    const cplx af = As[toffA[s] + ks];
                    const cplx bf = Bs[toffB[s] + ks * ZF_BPITCH];
                    s1[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.x, bf.x, s1[s], 0, 0, 0);
                    s2[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.y, bf.y, s2[s], 0, 0, 0);
                    s3[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.x + af.y, bf.x + bf.y, s3[s], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < ZF_MAXT; ++s) {
        const int t = wave + s * ZF_WAVES;
        if (t < ntiles) {
            const int ti = t / tn, tj = t - ti * tn;
            if (!store_t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = row0 + ti * 16 + fk + 4 * r, gj = col0 + tj * 16 + fi;
                    if (gi < M && gj < N) C[(size_t)gi * ldc + gj] = cmake(s1[s][r] - s2[s][r], s3[s][r] - s1[s][r] - s2[s][r]);
                }
            }
            if (mirror || store_t) {                                    // (wave-uniform) the conjugate transpose, through LDS: see zgemm_mfma_kernel
                cplx* T = Ts + wave * (16 * 17);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    T[fi * 17 + fk + 4 * r] = cmake(s1[s][r] - s2[s][r], -(s3[s][r] - s1[s][r] - s2[s][r]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = col0 + tj * 16 + fk + 4 * r, gc = row0 + ti * 16 + fi;
                    const cplx v = T[(fk + 4 * r) * 17 + fi];
                    if (gr < N && gc < M) C[(size_t)gr * ldc + gc] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

// Plain VALU version (no matrix cores): kept as the independent cross-check of
// the MFMA fragment maps (tests compare the two) and selectable with
// NEGF_ZGEMM_ALGO=valu for debugging.
__global__ __launch_bounds__(256) void zgemm_valu_kernel(
    int M, int N, int K,
    const cplx* __restrict__ Aall, int lda, size_t strideA,
    const cplx* __restrict__ Ball, int ldb, size_t strideB, int opB_arg,
    cplx* __restrict__ Call, int ldc, size_t strideC)
{
    const int opB = opB_arg & 1;                 // (the Hermitian hint is not used: every element is computed)
    const bool store_t = (opB_arg & 4) != 0;
    __shared__ cplx As[32][17];
    __shared__ cplx Bs[16][33];
    const int b = blockIdx.z;
    const cplx* A = Aall + (size_t)b * strideA;
    const cplx* B = Ball + (size_t)b * strideB;
    cplx* C = Call + (size_t)b * strideC;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;              // 16 x 16 threads, 2 x 2 outputs each
    const int row0 = blockIdx.y * 32, col0 = blockIdx.x * 32;
    cplx acc[2][2];
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) acc[a][c] = cmake(0.0, 0.0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int t = tid; t < 32 * 16; t += 256) {
            const int i = t >> 4, k = t & 15;
            cplx v = cmake(0.0, 0.0);
            if (row0 + i < M && k0 + k < K) v = A[(size_t)(row0 + i) * lda + k0 + k];
            As[i][k] = v;
        }
        for (int t = tid; t < 16 * 32; t += 256) {
            cplx v = cmake(0.0, 0.0);
            if (opB == 0) {
                const int k = t >> 5, j = t & 31;
                if (k0 + k < K && col0 + j < N) v = B[(size_t)(k0 + k) * ldb + col0 + j];
                Bs[k][j] = v;
            } else {
                const int j = t >> 4, k = t & 15;
                if (col0 + j < N && k0 + k < K) v = cconj(B[(size_t)(col0 + j) * ldb + k0 + k]);
                Bs[k][j] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = cfma(acc[a][c], As[ty + 16 * a][k], Bs[k][tx + 16 * c]);
        __syncthreads();
    }
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int gi = row0 + ty + 16 * a, gj = col0 + tx + 16 * c;
            if (gi < M && gj < N) {
                if (store_t) C[(size_t)gj * ldc + gi] = cconj(acc[a][c]);
                else C[(size_t)gi * ldc + gj] = acc[a][c];
            }
        }
}

static int zgemm_algo()
{
    static int algo = -1;
    if (algo < 0) {
        const char* e = getenv("NEGF_ZGEMM_ALGO");
        algo = (e && strcmp(e, "valu") == 0) ? 1 : 0;
    }
    return algo;
}

void launch_zgemm(hipStream_t st, int M, int N, int K, int nb,
                  const cplx* A, int lda, size_t strideA,
                  const cplx* B, int ldb, size_t strideB, int opB,
                  cplx* C, int ldc, size_t strideC)
{
    if (M <= 0 || N <= 0 || nb <= 0) return;
    if ((opB & 2) && (M != N || (opB & 4))) opB &= ~2;   // the Hermitian form needs a square result, stored as it is
    static int herm_env = -1;                    // NEGF_ZGEMM_HERM=0: compute Hermitian products in full (A/B, tests)
    if (herm_env < 0) { const char* e = getenv("NEGF_ZGEMM_HERM"); herm_env = e ? atoi(e) : 1; }
    if (!herm_env) opB &= ~2;
    {   // flop accounting (negf_common.h, FlopCount): 16 x 16 sub-tiles that hold part of the result, K padded to the
        // staged K-tile, three real products per sub-tile and k-step; a Hermitian product runs the sub-tiles of the
        // block tiles on and above the diagonal only
        const double tm = (M + 15) >> 4, tn = (N + 15) >> 4, kp = (double)((K + ZG_BK - 1) / ZG_BK * ZG_BK);
        double tiles = tm * tn;
        if (opB & 2) {
            const int bt = 4;                               // sub-tiles per block-tile edge (64 x 64 blocks; the flexible
            tiles = 0;                                      //  blocks are of similar size)
            for (int tj = 0; tj < (int)tn; ++tj) tiles += std::min<double>(tm, bt * (tj / bt + 1));
        }
        // ALGORITHMIC flops of a Hermitian product: the symmetry-exploiting count -- the elements on and above the diagonal,
        // 8 K flops each (N (N + 1) / 2 of the N^2: the mirrored half is a copy, not arithmetic) -- so that no fraction
        // derived from it can exceed the peak (round 4 charged 8 M N K and reported 1.30 "of peak" for C5's products)
        const double alg = (opB & 2) ? 8.0 * K * 0.5 * N * ((double)N + 1.0) * nb : 8.0 * M * (double)N * K * nb;
        negf_count_flops(alg, zgemm_algo() == 1 ? 0.0 : 3.0 * 2.0 * 256.0 * kp * tiles * nb);
    }
    if (zgemm_algo() == 1) {
        dim3 grid((N + 31) / 32, (M + 31) / 32, nb);
        hipLaunchKernelGGL(zgemm_valu_kernel, grid, dim3(256), 0, st, M, N, K, A, lda, strideA, B, ldb,
                           strideB, opB, C, ldc, strideC);
    } else {
        // how much of the 64 x 64 blocks' area is padding beyond the 16-granular tiles?  Above 20 % the
        // flexible-block kernel (balanced blocks of <= 7 x 7 sub-tiles) is used
        const long tm16 = (M + 15) >> 4, tn16 = (N + 15) >> 4;
        const long bm64 = (M + ZG_BM - 1) / ZG_BM, bn64 = (N + ZG_BN - 1) / ZG_BN;
        static int flex_env = -1;
        if (flex_env < 0) { const char* e = getenv("NEGF_ZGEMM_FLEX"); flex_env = e ? atoi(e) : 1; }
        const bool padded = 10 * (bm64 * 4 * bn64 * 4) > 12 * (tm16 * tn16);
        if (flex_env == 2 || (flex_env == 1 && padded)) {
            const int nbm = (int)((tm16 + ZF_MAXT - 1) / ZF_MAXT), nbn = (int)((tn16 + ZF_MAXT - 1) / ZF_MAXT);
            dim3 grid(nbn, nbm, nb);
            if (opB & 2) grid = dim3(zg_herm_grid(nbn, nb), 1, 1);
            hipLaunchKernelGGL(zgemm_flex_kernel, grid, dim3(ZF_THREADS), 0, st, M, N, K, nbm, nbn, A, lda, strideA,
                               B, ldb, strideB, opB, C, ldc, strideC, nb);
        } else {
            dim3 grid((unsigned)bn64, (unsigned)bm64, nb);
            if (opB & 2) grid = dim3(zg_herm_grid(bn64, nb), 1, 1);
            hipLaunchKernelGGL(zgemm_mfma_kernel, grid, dim3(256), 0, st, M, N, K, A, lda, strideA, B, ldb,
                               strideB, opB, C, ldc, strideC, nb);
        }
    }
}

// ------------------------------------------------------------------ self test
// One wave multiplies exact small-integer matrices A (16x4) and an ASYMMETRIC B
// (4x16) with a single MFMA and compares against the exact integer product under
// the documented fragment maps.  max_err must be exactly 0.
__global__ void mfma_probe_kernel(double* __restrict__ err_out)
{
    const int lane = threadIdx.x;
    const int i = lane & 15, k = lane >> 4;
    // A[i][k] = i + 3k + 1 ; B[k][j] = 2j - 5k + (j*j)%7  (no symmetry)
    const double a = (double)(i + 3 * k + 1);
    const int j = lane & 15;
    const double bval = (double)(2 * j - 5 * k + (j * j) % 7);
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bval, acc, 0, 0, 0);
    double e = 0.0;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) + 4 * r, col = lane & 15;
        double ref = 0.0;
        for (int kk = 0; kk < 4; ++kk)
            ref += (double)(row + 3 * kk + 1) * (double)(2 * col - 5 * kk + (col * col) % 7);
        e = fmax(e, fabs(acc[r] - ref));
    }
    for (int off = 32; off > 0; off >>= 1) e = fmax(e, __shfl_down(e, off, 64));
    if (lane == 0) err_out[0] = e;
}

int run_mfma_selftest(hipStream_t st, double* max_err)
{
    double* d = nullptr;
    NEGF_HIP_CHECK(hipMalloc(&d, sizeof(double)));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, st, d);
    NEGF_HIP_CHECK(hipGetLastError());
    NEGF_HIP_CHECK(hipMemcpyAsync(max_err, d, sizeof(double), hipMemcpyDeviceToHost, st));
    NEGF_HIP_CHECK(hipStreamSynchronize(st));
    NEGF_HIP_CHECK(hipFree(d));
    return NEGF_OK;
}
