// Bethe-lattice contact self-energy -- gauNEGF/surfGBethe.py:958-1030 (sigmaK,
// bulk), :1032-1108 (sigma, surface), :512-527 (per-atom assembly).
//
// One workgroup (6 waves) per (energy, contact).  All state is 9x9 complex
// blocks held in LDS.  The bulk sweep is Gauss-Seidel over the 12 FCC directions
// with Sigma_tot frozen per sweep (:1007-1014): direction k reads
// sigma[(k+6)%12], which for k<6 is still the previous sweep's value and for
// k>=6 is this sweep's value -> two phases of six mutually independent
// directions, one wave per direction.
//
//   z = E - i eta                      (note the MINUS, :995)
//   A = z I - H ; B_k = z S_k - V_k ; B_k^H = conj(z) S_k^T - V_k^T
//   bulk   : g_k = inv(A - Sigma_tot + sigma[(k+6)%12]); s_k <- mix B_k g_k B_k^H + (1-mix) s_k
//   surface: g = inv(A - sum_{k<9} s_k); k in {0,1,2,6,7,8}: s_k <- mix B_k g B_k^H + (1-mix) s_k
//   stop   : diff = max|s - s_old| / max|s_old| <= conv, or count == max_iter
//   atom   : sigma_atom = sum_{k<9} s_k - sum_{nb attached} s[min(nb,8)]   (:523-527)
#include "negf_common.h"

static constexpr int BE_WAVES = 6;
static constexpr int BE_THREADS = BE_WAVES * 64;
static constexpr int D = 9, D2 = 81;

struct BetheArgs {
    const double* H;        // [n_contacts][81]
    const double* Slist;    // [n_contacts][12][81]
    const double* Vlist;
    const int* n_atoms;     // [n_contacts]
    const int* atom_off;    // [n_contacts] first atom of the contact
    const int* nb_off;      // [total_atoms + 1]
    const int* nb_dirs;
    const int* blk_off;     // [n_contacts]
    int n_contacts, blk_stride;
    double eta, conv, mix;
    int max_iter, force_iters;
    int mode;               // 0: contact assembly, 1: dump bulk sigmaK[12], 2: dump surface sigma[9]
};

// wave-level 9x9 complex inverse (Gauss-Jordan, partial pivoting, izamax rule) on a
// matrix in LDS; lanes 0..8 own one row each for the pivot search, element updates
// are spread over (lane, lane+64).
__device__ void wave_inv9(cplx* M, int lane)
{
    unsigned long long ipiv = 0ull;          // 9 x 4-bit pivot rows (no private-memory array)
#pragma unroll 1
    for (int k = 0; k < D; ++k) {
        // pivot search among rows k..8 of column k
        double v = (lane >= k && lane < D) ? cabs1(M[lane * D + k]) : -1.0;
        int idx = lane;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_down(v, off, 64);
            const int oi = __shfl_down(idx, off, 64);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        int p = __shfl(idx, 0, 64);
        if (p < k || p >= D) p = k;
        ipiv |= (unsigned long long)p << (4 * k);
        if (p != k && lane < D) {
            const cplx a = M[k * D + lane], b = M[p * D + lane];
            M[k * D + lane] = b; M[p * D + lane] = a;
        }
        __builtin_amdgcn_wave_barrier();
        const cplx ip = crecip(M[k * D + k]);
        // each lane takes elements lane and lane+64 (< 81); read everything it needs first
        cplx nv[2];
        bool act[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = lane + 64 * h;
            act[h] = t < D2;
            nv[h] = cmake(0.0, 0.0);
            if (act[h]) {
                const int i = t / D, j = t - i * D;
                const cplx rk = (j == k) ? ip : cmul(M[k * D + j], ip);   // scaled pivot row entry
                if (i == k) nv[h] = rk;
                else {
                    const cplx f = M[i * D + k];
                    nv[h] = (j == k) ? cneg(cmul(f, ip)) : cfnma(M[t], f, rk);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (act[h]) M[lane + 64 * h] = nv[h];
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll 1
    for (int k = D - 1; k >= 0; --k) {
        const int p = (int)((ipiv >> (4 * k)) & 15ull);
        if (p != k && lane < D) {
            const cplx a = M[lane * D + k], b = M[lane * D + p];
            M[lane * D + k] = b; M[lane * D + p] = a;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// s_new = mix * (B g B^H) + (1-mix) * s_old for direction k; B = z S - V (S,V real).
// X (scratch) = B g, then out = X B^H.  B^H[l][j] = conj(z) S[j][l] - V[j][l].
__device__ void wave_bgb(const double* __restrict__ S, const double* __restrict__ V, cplx z,
                         const cplx* g, cplx* X, const cplx* s_old, cplx* s_new, double mix, int lane)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < D2) {
            const int i = t / D, j = t - i * D;
            cplx acc = cmake(0.0, 0.0);
            for (int l = 0; l < D; ++l) {
                const cplx bil = cmake(z.x * S[i * D + l] - V[i * D + l], z.y * S[i * D + l]);
                acc = cfma(acc, bil, g[l * D + j]);
            }
            X[t] = acc;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < D2) {
            const int i = t / D, j = t - i * D;
            cplx acc = cmake(0.0, 0.0);
            for (int l = 0; l < D; ++l) {
                const cplx bh = cmake(z.x * S[j * D + l] - V[j * D + l], -z.y * S[j * D + l]);
                acc = cfma(acc, X[i * D + l], bh);
            }
            const cplx so = s_old[t];
            s_new[t] = cmake(mix * acc.x + (1.0 - mix) * so.x, mix * acc.y + (1.0 - mix) * so.y);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ double block_max(double v, double* red, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double m = red[0];
    for (int w = 1; w < BE_WAVES; ++w) m = fmax(m, red[w]);
    return m;
}

__global__ __launch_bounds__(BE_THREADS) void bethe_kernel(
    BetheArgs a, const cplx* __restrict__ E, cplx* __restrict__ blk, int* __restrict__ iters,
    int* __restrict__ converged)
{
    __shared__ cplx sig[12 * D2];        // current sigma_k
    __shared__ cplx sold[12 * D2];       // previous sweep
    __shared__ cplx tot[D2];
    __shared__ cplx Mw[BE_WAVES * D2];   // per-wave matrix to invert
    __shared__ cplx Xw[BE_WAVES * D2];   // per-wave scratch
    __shared__ double red[BE_WAVES];
    __shared__ double red2[BE_WAVES];

    const int c = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* H = a.H + (size_t)c * D2;
    const double* Sl = a.Slist + (size_t)c * 12 * D2;
    const double* Vl = a.Vlist + (size_t)c * 12 * D2;
    const cplx e = E[b];
    const cplx z = cmake(e.x, e.y - a.eta);
    cplx* M = Mw + wave * D2;
    cplx* X = Xw + wave * D2;

    for (int t = tid; t < 12 * D2; t += BE_THREADS) {
        const int r = t % D2;
        sig[t] = ((r / D) == (r % D)) ? cmake(0.0, -1.0) : cmake(0.0, 0.0);
    }
    __syncthreads();

    // ------------------------------------------------------------ bulk sweeps
    int count = 0;
    double diff = INFINITY;
    while (true) {
        if (a.force_iters >= 0) { if (count >= a.force_iters) break; }
        else if (!(diff > a.conv && count < a.max_iter)) break;
        for (int t = tid; t < 12 * D2; t += BE_THREADS) sold[t] = sig[t];
        for (int t = tid; t < D2; t += BE_THREADS) {
            cplx s = cmake(0.0, 0.0);
            for (int k = 0; k < 12; ++k) s = cadd(s, sig[k * D2 + t]);
            tot[t] = s;
        }
        __syncthreads();
        for (int phase = 0; phase < 2; ++phase) {
            const int k = phase * 6 + wave;
            const int pk = (k + 6) % 12;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = lane + 64 * h;
                if (t < D2) {
                    const int i = t / D, j = t - i * D;
                    cplx m = cmake(-H[t], 0.0);
                    if (i == j) m = cadd(m, z);
                    m = csub(m, tot[t]);
                    m = cadd(m, sig[pk * D2 + t]);   // phase 1 sees this sweep's sigma[0..5]
                    M[t] = m;
                }
            }
            __builtin_amdgcn_wave_barrier();
            wave_inv9(M, lane);
            wave_bgb(Sl + k * D2, Vl + k * D2, z, M, X, sold + k * D2, sig + k * D2, a.mix, lane);
            __syncthreads();
        }
        double num = 0.0, den = 0.0;
        for (int t = tid; t < 12 * D2; t += BE_THREADS) {
            const cplx s = sig[t], o = sold[t];
            num = fmax(num, hypot(s.x - o.x, s.y - o.y));
            den = fmax(den, hypot(o.x, o.y));
        }
        num = block_max(num, red, tid);
        den = block_max(den, red2, tid);
        diff = num / den;
        ++count;
    }
    const int countK = count;
    const double diffK = diff;
    if (a.mode == 1) {
        // raw bulk self-energies (surfGBAt.sigmaK): record = [12][81]
        cplx* out = blk + ((size_t)b * a.n_contacts + c) * (12 * D2);
        for (int t = tid; t < 12 * D2; t += BE_THREADS) out[t] = sig[t];
        if (tid == 0) {
            if (iters) iters[(size_t)b * a.n_contacts + c] = countK;
            if (converged) converged[(size_t)b * a.n_contacts + c] = (diffK <= a.conv) ? 1 : 0;
        }
        return;
    }

    // --------------------------------------------------------- surface sweeps
    // s = sigmaK[:9] lives in sig[0..8]; planeVec = {0,1,2,6,7,8}
    count = 0;
    diff = INFINITY;
    const int plane_k = (wave < 3) ? wave : wave + 3;
    while (true) {
        if (a.force_iters >= 0) { if (count >= a.force_iters) break; }
        else if (!(diff > a.conv && count < a.max_iter)) break;
        for (int t = tid; t < 9 * D2; t += BE_THREADS) sold[t] = sig[t];
        __syncthreads();
        // every wave forms the same g = inv(A - sum_{k<9} s_k) in its own LDS slot
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = lane + 64 * h;
            if (t < D2) {
                const int i = t / D, j = t - i * D;
                cplx m = cmake(-H[t], 0.0);
                if (i == j) m = cadd(m, z);
                cplx s = cmake(0.0, 0.0);
                for (int k = 0; k < 9; ++k) s = cadd(s, sold[k * D2 + t]);
                M[t] = csub(m, s);
            }
        }
        __builtin_amdgcn_wave_barrier();
        wave_inv9(M, lane);
        wave_bgb(Sl + plane_k * D2, Vl + plane_k * D2, z, M, X, sold + plane_k * D2,
                 sig + plane_k * D2, a.mix, lane);
        __syncthreads();
        double num = 0.0, den = 0.0;
        for (int t = tid; t < 9 * D2; t += BE_THREADS) {
            const cplx s = sig[t], o = sold[t];
            num = fmax(num, hypot(s.x - o.x, s.y - o.y));
            den = fmax(den, hypot(o.x, o.y));
        }
        num = block_max(num, red, tid);
        den = block_max(den, red2, tid);
        diff = num / den;
        ++count;
    }

    if (a.mode == 2) {
        // raw surface self-energies (surfGBAt.sigma): record = [9][81]
        cplx* out = blk + ((size_t)b * a.n_contacts + c) * (9 * D2);
        for (int t = tid; t < 9 * D2; t += BE_THREADS) out[t] = sig[t];
        if (tid == 0) {
            if (iters) iters[(size_t)b * a.n_contacts + c] = countK + (count << 16);
            if (converged) converged[(size_t)b * a.n_contacts + c] =
                ((diffK <= a.conv) ? 1 : 0) | ((diff <= a.conv) ? 2 : 0);
        }
        return;
    }
    // ---------------------------------------------------------- atom assembly
    // contact block = block-diagonal (9 n_atoms) x (9 n_atoms), one 9x9 per atom
    for (int t = tid; t < D2; t += BE_THREADS) {
        cplx s = cmake(0.0, 0.0);
        for (int k = 0; k < 9; ++k) s = cadd(s, sig[k * D2 + t]);
        tot[t] = s;
    }
    __syncthreads();
    const int na = a.n_atoms[c];
    const int nc = na * D;
    cplx* out = blk + (size_t)b * a.blk_stride + a.blk_off[c];
    for (int t = tid; t < nc * nc; t += BE_THREADS) {
        const int i = t / nc, j = t - i * nc;
        const int ai = i / D, aj = j / D;
        cplx v = cmake(0.0, 0.0);
        if (ai == aj) {
            const int r = (i - ai * D) * D + (j - aj * D);
            v = tot[r];
            const int atom = a.atom_off[c] + ai;
            for (int q = a.nb_off[atom]; q < a.nb_off[atom + 1]; ++q) {
                int nb = a.nb_dirs[q];
                // jax indexing of a length-9 array: a negative index wraps (numpy rule), what is still out of
                // range clamps (jax retrieval rule); pinned against the reference's numpy twin inside 0..8 only
                if (nb < 0) nb += 9;
                nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
                v = csub(v, sig[nb * D2 + r]);
            }
        }
        out[t] = v;
    }
    if (tid == 0) {
        // iteration record: bulk count in iters, surface count encoded in the upper half
        if (iters) iters[(size_t)b * a.n_contacts + c] = countK + (count << 16);
        if (converged) converged[(size_t)b * a.n_contacts + c] =
            ((diffK <= a.conv) ? 1 : 0) | ((diff <= a.conv) ? 2 : 0);
    }
}

void launch_bethe(hipStream_t st, const SigmaProvider& p, int nb, const cplx* E, cplx* blk,
                  int* iters, int* conv)
{
    BetheArgs a;
    a.H = p.d_H; a.Slist = p.d_Slist; a.Vlist = p.d_Vlist;
    a.n_atoms = p.d_n_atoms; a.atom_off = p.d_atom_off;
    a.nb_off = p.d_nb_off; a.nb_dirs = p.d_nb_dirs; a.blk_off = p.d_blk_off;
    a.n_contacts = p.n_contacts; a.blk_stride = p.blk_stride;
    a.eta = p.eta; a.conv = p.conv; a.mix = p.mix;
    a.max_iter = p.max_iter; a.force_iters = p.force_iters;
    a.mode = 0;
    hipLaunchKernelGGL(bethe_kernel, dim3(p.n_contacts, nb), dim3(BE_THREADS), 0, st, a, E, blk, iters,
                       conv);
}

void launch_bethe_raw(hipStream_t st, const double* d_H, const double* d_S, const double* d_V, double eta,
                      double conv, double mix, int max_iter, int force_iters, int which, int nb,
                      const cplx* E, cplx* out, int* iters, int* converged)
{
    BetheArgs a;
    a.H = d_H; a.Slist = d_S; a.Vlist = d_V;
    a.n_atoms = nullptr; a.atom_off = nullptr; a.nb_off = nullptr; a.nb_dirs = nullptr; a.blk_off = nullptr;
    a.n_contacts = 1; a.blk_stride = 0;
    a.eta = eta; a.conv = conv; a.mix = mix; a.max_iter = max_iter; a.force_iters = force_iters;
    a.mode = which;
    hipLaunchKernelGGL(bethe_kernel, dim3(1, nb), dim3(BE_THREADS), 0, st, a, E, out, iters, converged);
}
