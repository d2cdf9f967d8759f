// 1-D chain contact self-energy ("decimation"), LDS-resident version for n_c <= 64.
// gauNEGF/surfG1D.py:223-295 (g), :344-373 (sigma).  gfx950.
//
// One workgroup (256 threads, 4 waves) per (energy, contact).  The iterate g and the work
// matrix live in LDS for the whole fixed point; A, B, tau, Stau are read-only operands
// fetched from L2.  Per sweep
//     T     = B g                       complex GEMM on the FP64 matrix cores
//     M     = A - T B^H                 second GEMM, conj-transposed B fetched on the fly
//     g_new = inv(M)                    blocked Gauss-Jordan, implicit pivoting (below)
//     diff  = max |g_new - g| / max(|g_new|, 1e-12) ;  g = r g_new + (1-r) g
// The inverse is the small-matrix form of k_inverse_blocked.hip: panels of 16 columns;
// because n_c <= 64 ONE wave holds every row (lane = row, 16 complex per lane), so the 16
// pivot steps of a panel need no barrier at all (DPP arg-max, pivot row through a tiny
// LDS buffer, wave-synchronous); the trailing update of the other columns is MFMA work
// shared by the four waves.  Rows are never swapped; pivrow[]/colof[] are resolved when
// g_new is read:  g_new[i][j] = W[pivrow[i]][colof[j]].
//
// Every workgroup stops on ITS OWN convergence (the reference's vmap runs all energies
// until the slowest lane converges; results are identical because a converged lane is
// frozen there).
#include "negf_common.h"
#include "wave_utils.h"

namespace {

constexpr int CL_THREADS = 256;
constexpr int CL_WAVES = 4;
constexpr int CL_NB = 16;                // panel width of the small inverse

struct ChainArgs {
    const cplx *alpha, *Salpha, *beta, *Sbeta, *tau, *Stau;   // concatenated per contact
    const int* nc;
    const int* blk_off;
    int n_contacts, blk_stride;
    double eta, conv, relFactor;
    int max_iter, force_iters;
    int b_in_lds;
};

// C tile loop of a small complex GEMM shared by the four waves.  Tiles (ti, tj) of the
// T16 x T16 grid are dealt to the waves; fa(i,k) / fb(k,j) fetch the operands (zero outside
// the matrix), c0(i,j) the initial value, out(i,j,v) consumes the result.
template <class FA, class FB, class FC, class FO>
__device__ __forceinline__ void small_gemm(int n, int T16, int wave, int lane, FA fa, FB fb, FC c0, FO out)
{
    const int fi = lane & 15, fk = lane >> 4;
    const int ksteps = (n + 3) >> 2;
    for (int t = wave; t < T16 * T16; t += CL_WAVES) {
        const int ti = t / T16, tj = t - ti * T16;
        d4 accr, acci;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cplx v = c0(ti * 16 + fk + 4 * r, tj * 16 + fi);
            accr[r] = v.x; acci[r] = v.y;
        }
        for (int ks = 0; ks < ksteps; ++ks) {
            const cplx pa = fa(ti * 16 + fi, ks * 4 + fk);
            const cplx qb = fb(ks * 4 + fk, tj * 16 + fi);
            zmfma(accr, acci, pa, qb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) out(ti * 16 + fk + 4 * r, tj * 16 + fi, cmake(accr[r], acci[r]));
    }
}

// In-place blocked Gauss-Jordan reduction of the n x n matrix W (LDS, pitch LP) with implicit
// pivoting.  On return  inv[i][j] = W[pivrow[i]][colof[j]].  All 256 threads call it.
__device__ void small_inverse(int n, int T16, cplx* W, int LP, cplx* Qs /*[16][LP]*/, cplx* rowbuf /*[16]*/,
                              int* pivrow, int* colof, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    for (int t = tid; t < 64; t += CL_THREADS) { colof[t] = -1; pivrow[t] = 0; }
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += CL_NB) {
        const int pw = min(CL_NB, n - p0);
        if (wave == 0) {
            // ---- panel: lane = row, 16 complex per lane, no workgroup barrier inside
            const int r = lane;
            cplx a[CL_NB];
            bool avail = r < n && colof[r] < 0;
#pragma unroll
            for (int s = 0; s < CL_NB; ++s) a[s] = (r < n && s < pw) ? W[r * LP + p0 + s] : cmake(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < CL_NB; ++j) {
                if (j < pw) {
                    double bv = avail ? cabs1(a[j]) : -1.0;
                    int bkey = avail ? r : 0x7fffffff;
                    wave_argmax(bv, bkey);
                    int pphys = bkey;
                    if (pphys == 0x7fffffff) {                  // NaN column: lowest available row
                        int cand = avail ? r : 0x7fffffff;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
                        pphys = cand;
                    }
                    if (r == pphys) {
#pragma unroll
                        for (int s = 0; s < CL_NB; ++s) rowbuf[s] = a[s];
                        pivrow[p0 + j] = pphys; colof[pphys] = p0 + j;
                    }
                    __builtin_amdgcn_wave_barrier();
                    cplx rb[CL_NB];
#pragma unroll
                    for (int s = 0; s < CL_NB; ++s) rb[s] = rowbuf[s];
                    const cplx pv = rb[j];
                    const double sc = 1.0 / (pv.x * pv.x + pv.y * pv.y);
                    const cplx ip = cmake(pv.x * sc, -pv.y * sc);
                    const bool is_piv = r == pphys;
                    const cplx coef = is_piv ? ip : cneg(cmul(a[j], ip));
#pragma unroll
                    for (int s = 0; s < CL_NB; ++s) {
                        const cplx base = is_piv ? cmake(0.0, 0.0) : a[s];
                        a[s] = cfma(base, coef, rb[s]);
                    }
                    a[j] = coef;
                    avail = avail && !is_piv;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (r < n) {
#pragma unroll
                for (int s = 0; s < CL_NB; ++s)
                    if (s < pw) W[r * LP + p0 + s] = a[s];
            }
        }
        __syncthreads();                 // P (panel columns of W), pivrow/colof visible
        // ---- pivot rows -> Q snapshot
        for (int t = tid; t < pw * n; t += CL_THREADS) {
            const int k = t / n, j = t - k * n;
            Qs[k * LP + j] = W[pivrow[p0 + k] * LP + j];
        }
        __syncthreads();
        // ---- trailing update of the other columns, in place
        const int pt = p0 >> 4;                                   // the panel's column tile
        for (int t = wave; t < T16 * T16; t += CL_WAVES) {
            const int ti = t / T16, tj = t - ti * T16;
            if (tj == pt) continue;
            const int col = tj * 16 + fi;
            d4 accr, acci;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                cplx v = cmake(0.0, 0.0);
                if (i < n && col < n) {
                    const int cf = colof[i];
                    if (!(cf >= p0 && cf < p0 + pw)) v = W[i * LP + col];
                }
                accr[r] = v.x; acci[r] = v.y;
            }
#pragma unroll
            for (int ks = 0; ks < CL_NB / 4; ++ks) {
                const int k = ks * 4 + fk, pr = ti * 16 + fi;
                const cplx pa = (pr < n && k < pw) ? W[pr * LP + p0 + k] : cmake(0.0, 0.0);
                const cplx qb = (k < pw && col < n) ? Qs[k * LP + col] : cmake(0.0, 0.0);
                zmfma(accr, acci, pa, qb);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + fk + 4 * r;
                if (i < n && col < n) W[i * LP + col] = cmake(accr[r], acci[r]);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(CL_THREADS) void chain1d_lds_kernel(
    ChainArgs a, const cplx* __restrict__ E, cplx* __restrict__ blk, int* __restrict__ iters,
    int* __restrict__ converged)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red_v[CL_WAVES];
    __shared__ int pivrow[64], colof[64];
    __shared__ cplx rowbuf[CL_NB];

    const int c = blockIdx.x, b = blockIdx.y;
    const int n = a.nc[c];
    const int off = a.blk_off[c];
    const int T16 = (n + 15) >> 4;
    const int LP = n | 1;                               // odd pitch; no padded rows: tile accesses are guarded
    cplx* Gs = reinterpret_cast<cplx*>(smem_raw);       // [n][LP] current (mixed) g
    cplx* Ws = Gs + n * LP;                             // [n][LP] work matrix
    cplx* Qs = Ws + n * LP;                             // [16][LP] pivot rows of a panel
    cplx* Bs = a.b_in_lds ? Qs + CL_NB * LP : nullptr;  // [n][LP] B = (E + i eta) Sb - b, when it fits
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const cplx* alpha = a.alpha + off; const cplx* Salpha = a.Salpha + off;
    const cplx* beta = a.beta + off;   const cplx* Sbeta = a.Sbeta + off;
    const cplx* tau = a.tau + off;     const cplx* Stau = a.Stau + off;
    const cplx e = E[b];
    const cplx z = cmake(e.x, e.y + a.eta);
    auto Aat = [&](int i, int j) { return (i < n && j < n) ? csub(cmul(z, Salpha[i * n + j]), alpha[i * n + j]) : cmake(0.0, 0.0); };
    auto Bglob = [&](int i, int j) { return csub(cmul(z, Sbeta[i * n + j]), beta[i * n + j]); };
    auto Bat = [&](int i, int j) {
        if (!(i < n && j < n)) return cmake(0.0, 0.0);
        return Bs ? Bs[i * LP + j] : Bglob(i, j);
    };
    auto tat = [&](int i, int j) { return (i < n && j < n) ? csub(cmul(e, Stau[i * n + j]), tau[i * n + j]) : cmake(0.0, 0.0); };
    auto Gat = [&](int i, int j) { return (i < n && j < n) ? Gs[i * LP + j] : cmake(0.0, 0.0); };
    auto Wat = [&](int i, int j) { return (i < n && j < n) ? Ws[i * LP + j] : cmake(0.0, 0.0); };

    // ---- g0 = inv(A)
    for (int t = tid; t < n * n; t += CL_THREADS) {
        const int i = t / n, j = t - i * n;
        Ws[i * LP + j] = Aat(i, j);
        if (Bs) Bs[i * LP + j] = Bglob(i, j);
    }
    __syncthreads();
    small_inverse(n, T16, Ws, LP, Qs, rowbuf, pivrow, colof, tid);
    for (int t = tid; t < n * n; t += CL_THREADS) {
        const int i = t / n, j = t - i * n;
        Gs[i * LP + j] = Ws[pivrow[i] * LP + colof[j]];
    }
    __syncthreads();

    int count = 0;
    double diff = INFINITY;
    while (true) {
        if (a.force_iters >= 0) { if (count >= a.force_iters) break; }
        else if (!(diff > a.conv && count < a.max_iter)) break;
        // T = B g -> Ws
        small_gemm(n, T16, wave, lane,
                   [&](int i, int k) { return Bat(i, k); },
                   [&](int k, int j) { return Gat(k, j); },
                   [&](int, int) { return cmake(0.0, 0.0); },
                   [&](int i, int j, cplx v) { if (i < n && j < n) Ws[i * LP + j] = v; });
        __syncthreads();
        // M = A - T B^H : accumulate with the negated left operand; results stay in registers
        // until every wave has finished reading T, then overwrite Ws
        {
            const int fi = lane & 15, fk = lane >> 4;
            const int ksteps = (n + 3) >> 2;
            d4 mr[4], mi[4];                         // up to 4 tiles per wave (T16 <= 4)
            int nt = 0;
            for (int t = wave; t < T16 * T16; t += CL_WAVES, ++nt) {
                const int ti = t / T16, tj = t - ti * T16;
                d4 accr, acci;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx v = Aat(ti * 16 + fk + 4 * r, tj * 16 + fi);
                    accr[r] = v.x; acci[r] = v.y;
                }
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int k = ks * 4 + fk;
                    const cplx pa = cneg(Wat(ti * 16 + fi, k));                     // -T[i][k]
                    const cplx qb = cconj(Bat(tj * 16 + fi, k));                    // (B^H)[k][j] = conj(B[j][k])
                    zmfma(accr, acci, pa, qb);
                }
                if (nt == 0) { mr[0] = accr; mi[0] = acci; }
                else if (nt == 1) { mr[1] = accr; mi[1] = acci; }
                else if (nt == 2) { mr[2] = accr; mi[2] = acci; }
                else { mr[3] = accr; mi[3] = acci; }
            }
            __syncthreads();
            nt = 0;
            for (int t = wave; t < T16 * T16; t += CL_WAVES, ++nt) {
                const int ti = t / T16, tj = t - ti * T16;
                const d4 accr = nt == 0 ? mr[0] : (nt == 1 ? mr[1] : (nt == 2 ? mr[2] : mr[3]));
                const d4 acci = nt == 0 ? mi[0] : (nt == 1 ? mi[1] : (nt == 2 ? mi[2] : mi[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + fk + 4 * r, j = tj * 16 + fi;
                    if (i < n && j < n) Ws[i * LP + j] = cmake(accr[r], acci[r]);
                }
            }
        }
        __syncthreads();
        small_inverse(n, T16, Ws, LP, Qs, rowbuf, pivrow, colof, tid);
        // diff and mixing
        double d = 0.0;
        for (int t = tid; t < n * n; t += CL_THREADS) {
            const int i = t / n, j = t - i * n;
            const cplx gn = Ws[pivrow[i] * LP + colof[j]], go = Gs[i * LP + j];
            const double num = hypot(gn.x - go.x, gn.y - go.y);
            const double den = fmax(hypot(gn.x, gn.y), 1e-12);
            d = fmax(d, num / den);
            Gs[i * LP + j] = cmake(gn.x * a.relFactor + go.x * (1.0 - a.relFactor),
                                   gn.y * a.relFactor + go.y * (1.0 - a.relFactor));
        }
        d = wave_max(d);
        if (lane == 0) red_v[wave] = d;
        __syncthreads();
        diff = fmax(fmax(red_v[0], red_v[1]), fmax(red_v[2], red_v[3]));
        __syncthreads();
        ++count;
    }
    // ---- Sigma_c = t g t^H, t = E Stau - tau (no eta):  X = t g -> Ws ; Sigma = X t^H -> global
    small_gemm(n, T16, wave, lane,
               [&](int i, int k) { return tat(i, k); },
               [&](int k, int j) { return Gat(k, j); },
               [&](int, int) { return cmake(0.0, 0.0); },
               [&](int i, int j, cplx v) { if (i < n && j < n) Ws[i * LP + j] = v; });
    __syncthreads();
    cplx* out = blk + (size_t)b * a.blk_stride + off;
    small_gemm(n, T16, wave, lane,
               [&](int i, int k) { return Wat(i, k); },
               [&](int k, int j) { return cconj(tat(j, k)); },
               [&](int, int) { return cmake(0.0, 0.0); },
               [&](int i, int j, cplx v) { if (i < n && j < n) out[i * n + j] = v; });
    if (tid == 0) {
        if (iters) iters[(size_t)b * a.n_contacts + c] = count;
        if (converged) converged[(size_t)b * a.n_contacts + c] = (diff <= a.conv) ? 1 : 0;
    }
}

}  // namespace

bool chain1d_lds_supported(int nc_max) { return nc_max <= 64; }

void launch_chain1d_lds(hipStream_t st, const SigmaProvider& p, const int* d_nc, const int* d_blk_off, int nb,
                        const cplx* E, cplx* blk, int* iters, int* conv)
{
    ChainArgs a;
    a.alpha = p.d_alpha; a.Salpha = p.d_Salpha; a.beta = p.d_beta; a.Sbeta = p.d_Sbeta;
    a.tau = p.d_tau; a.Stau = p.d_Stau;
    a.nc = d_nc; a.blk_off = d_blk_off;
    a.n_contacts = p.n_contacts; a.blk_stride = p.blk_stride;
    a.eta = p.eta; a.conv = p.conv; a.relFactor = p.relFactor;
    a.max_iter = p.max_iter; a.force_iters = p.force_iters;
    const int n = p.nc_max, LP = n | 1;
    const size_t limit = 158 * 1024;
    size_t smem = (size_t)(2 * n + CL_NB) * LP * sizeof(cplx);
    a.b_in_lds = (smem + (size_t)n * LP * sizeof(cplx) <= limit) ? 1 : 0;
    if (a.b_in_lds) smem += (size_t)n * LP * sizeof(cplx);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain1d_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(chain1d_lds_kernel, dim3(p.n_contacts, nb), dim3(CL_THREADS), smem, st, a, E, blk, iters, conv);
}
