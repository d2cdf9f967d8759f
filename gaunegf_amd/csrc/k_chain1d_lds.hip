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
    unsigned long long* stamps;      // diagnostic (NEGF_CHAIN_STAMPS): wall-clock stamps of workgroup (0,0), sweep 10
};

// Small complex GEMM shared by the four waves (T16 <= 4 tiles per dimension): wave w owns column
// tile w and accumulates ALL its row tiles at once -- one B-operand read serves up to four MFMA
// groups and the four independent accumulator sets keep the matrix pipe busy across the LDS
// latency.  fa(i,k) / fb(k,j) fetch the operands (zero outside the matrix), c0(i,j) the initial
// value; the result stays in registers (accr/acci[ti]) for the caller to consume with gemm_store.
template <int T16, class FA, class FB, class FC>
__device__ __forceinline__ void small_gemm(int n, int wave, int lane, FA fa, FB fb, FC c0,
                                           d4 (&accr)[T16], d4 (&acci)[T16])
{
    const int fi = lane & 15, fk = lane >> 4;
    const int ksteps = (n + 3) >> 2;
    const int tj = wave;
    if (tj >= T16) return;
#pragma unroll
    for (int ti = 0; ti < T16; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cplx v = c0(ti * 16 + fk + 4 * r, tj * 16 + fi);
            accr[ti][r] = v.x; acci[ti][r] = v.y;
        }
    for (int ks = 0; ks < ksteps; ++ks) {
        const cplx qb = fb(ks * 4 + fk, tj * 16 + fi);
        cplx pa[T16];
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) pa[ti] = fa(ti * 16 + fi, ks * 4 + fk);
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) zmfma(accr[ti], acci[ti], pa[ti], qb);
    }
}

template <int T16, class FO>
__device__ __forceinline__ void gemm_store(int wave, int lane, const d4 (&accr)[T16], const d4 (&acci)[T16], FO out)
{
    const int fi = lane & 15, fk = lane >> 4;
    const int tj = wave;
    if (tj >= T16) return;
#pragma unroll
    for (int ti = 0; ti < T16; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) out(ti * 16 + fk + 4 * r, tj * 16 + fi, cmake(accr[ti][r], acci[ti][r]));
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_key(unsigned long long k)
{
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > k ? o : k;
}

// maximum of a 64-bit key over the wave (all lanes active), wave-uniform result
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k)
{
    k = dpp_max_key<0xB1>(k);      // quad_perm [1,0,3,2]
    k = dpp_max_key<0x4E>(k);      // quad_perm [2,3,0,1]
    k = dpp_max_key<0x141>(k);     // row_half_mirror
    k = dpp_max_key<0x140>(k);     // row_mirror
    unsigned long long best = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, r * 16);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), r * 16);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        best = o > best ? o : best;
    }
    return best;
}

// In-place blocked Gauss-Jordan reduction of the n x n matrix W (LDS, pitch LP) with implicit
// pivoting.  On return  inv[i][j] = W[pivrow[i]][colof[j]].  All 256 threads call it.
// Panels of 16 columns with look-ahead: at stage s all waves first apply panel s to the column
// tile of panel s+1, then wave 0 factors panel s+1 while waves 1-3 apply panel s to the remaining
// column tiles.
__device__ __forceinline__ void small_inverse(int n, int T16, cplx* W, int LP, cplx* Qs /*[16][LP]*/, cplx* rowbuf /*[16]*/,
                              int* pivrow, int* colof, int tid, unsigned long long* st = nullptr)
{
    (void)rowbuf;
    int sti = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if (st && tid == 0) st[sti] = __builtin_amdgcn_s_memrealtime(); ++sti; };
    const int lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    for (int t = tid; t < 64; t += CL_THREADS) { colof[t] = -1; pivrow[t] = 0; }
    __syncthreads();

    // ---- panel [p0, p0+pw): lane = row, 16 complex per lane; no barrier and no LDS traffic inside:
    // the pivot search is a DPP max of a packed 64-bit key, the pivot row is spread to all lanes
    // through v_readlane (the row index is wave-uniform).  A pivot row is not scaled at its column
    // step (multiplier 0, a one in the pivot column) but once at the end of the panel: the later
    // steps act linearly on it, and every lane runs the same select-free update.
    auto factor = [&](int p0, int pw) __attribute__((always_inline)) {
        const int r = lane;
        cplx a[CL_NB];
        bool avail = r < n && colof[r] < 0;
        cplx myip = cmake(1.0, 0.0);
#pragma unroll
        for (int s = 0; s < CL_NB; ++s) a[s] = (r < n && s < pw) ? W[r * LP + p0 + s] : cmake(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < CL_NB; ++j) {
            if (j < pw) {
                // key: upper 48 bits of |a|_1 over (0xFFFF - row): larger value, then lower row; 0 = none
                const double v = cabs1(a[j]);
                unsigned long long key = 0;
                if (avail && v == v)
                    key = ((unsigned long long)__double_as_longlong(v) & ~0xFFFFull) | (unsigned long long)(0xFFFF - r);
                key = wave_max_key(key);
                int pphys;
                if (key != 0) {
                    pphys = 0xFFFF - (int)(key & 0xFFFFull);
                } else {                                    // NaN column: lowest available row
                    int cand = avail ? r : 0x7fffffff;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
                    pphys = cand;
                }
                pphys = __builtin_amdgcn_readfirstlane(pphys);
                if (r == pphys) { pivrow[p0 + j] = pphys; colof[pphys] = p0 + j; }
                cplx rb[CL_NB];
#pragma unroll
                for (int s = 0; s < CL_NB; ++s) {
                    rb[s].x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[s].x), pphys),
                                               __builtin_amdgcn_readlane(__double2loint(a[s].x), pphys));
                    rb[s].y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[s].y), pphys),
                                               __builtin_amdgcn_readlane(__double2loint(a[s].y), pphys));
                }
                const cplx pv = rb[j];
                const double sc = 1.0 / (pv.x * pv.x + pv.y * pv.y);
                const cplx ip = cmake(pv.x * sc, -pv.y * sc);
                const bool is_piv = r == pphys;
                const cplx mf = cneg(cmul(a[j], ip));
                const cplx coef = cmake(is_piv ? 0.0 : mf.x, is_piv ? 0.0 : mf.y);
#pragma unroll
                for (int s = 0; s < CL_NB; ++s) a[s] = cfma(a[s], coef, rb[s]);
                a[j] = is_piv ? cmake(1.0, 0.0) : coef;
                myip = cmake(is_piv ? ip.x : myip.x, is_piv ? ip.y : myip.y);
                avail = avail && !is_piv;
            }
        }
        if (r < n) {
#pragma unroll
            for (int s = 0; s < CL_NB; ++s)
                if (s < pw) W[r * LP + p0 + s] = cmul(a[s], myip);       // the deferred pivot-row scaling
        }
    };

    // ---- trailing update of tile (ti, tj) with panel [p0, p0+pw), in place:
    //      W[i][col] = (i pivot row of the panel ? 0 : W[i][col]) + P[i][:] Q[:][col]
    auto update_tile = [&](int ti, int tj, int p0, int pw) __attribute__((always_inline)) {
        const int col = tj * 16 + fi;
        d4 accr, acci;
        const int colc = min(col, n - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // unconditional loads at clamped indices + selects: no branches in front of the MFMAs
            const int i = ti * 16 + fk + 4 * r, ic = min(i, n - 1);
            const int cf = colof[ic];
            const cplx v = W[ic * LP + colc];
            const bool keep = (i < n) & (col < n) & !(cf >= p0 && cf < p0 + pw);
            accr[r] = keep ? v.x : 0.0; acci[r] = keep ? v.y : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < CL_NB / 4; ++ks) {
            const int k = ks * 4 + fk, pr = ti * 16 + fi;
            const int kc = min(k, pw - 1);
            const cplx pav = W[min(pr, n - 1) * LP + p0 + kc];
            const cplx qbv = Qs[kc * LP + colc];
            const bool oka = (pr < n) & (k < pw), okb = (k < pw) & (col < n);
            const cplx pa = cmake(oka ? pav.x : 0.0, oka ? pav.y : 0.0);
            const cplx qb = cmake(okb ? qbv.x : 0.0, okb ? qbv.y : 0.0);
            zmfma(accr, acci, pa, qb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + fk + 4 * r;
            if (i < n && col < n) W[i * LP + col] = cmake(accr[r], acci[r]);
        }
    };

    const int npanels = (n + CL_NB - 1) / CL_NB;
    for (int sgi = -1; sgi < npanels; ++sgi) {
        const bool has_cur = sgi >= 0, has_next = sgi + 1 < npanels;
        const int p0 = has_cur ? sgi * CL_NB : 0, pw = has_cur ? min(CL_NB, n - p0) : 0;
        const int n0 = (sgi + 1) * CL_NB, nw = has_next ? min(CL_NB, n - n0) : 0;
        if (has_cur) {
            // pivot rows of panel sgi -> Q snapshot
            for (int t = tid; t < pw * n; t += CL_THREADS) {
                const int k = t / n, j = t - k * n;
                Qs[k * LP + j] = W[pivrow[p0 + k] * LP + j];
            }
            __syncthreads();
            stamp();
            if (has_next) {
                // look-ahead: the column tile of panel sgi+1, one row tile per wave
                for (int ti = wave; ti < T16; ti += CL_WAVES) update_tile(ti, sgi + 1, p0, pw);
                __syncthreads();
            }
        }
        if (wave == 0 && has_next) {
            factor(n0, nw);
        } else if (has_cur) {
            // the other column tiles: waves 1-3 while wave 0 factors, all four after the last panel
            const int team = has_next ? CL_WAVES - 1 : CL_WAVES, me = has_next ? wave - 1 : wave;
            int cnt = 0;
            for (int t = 0; t < T16 * T16; ++t) {
                const int ti = t / T16, tj = t - ti * T16;
                if (tj == sgi || (has_next && tj == sgi + 1)) continue;
                if (cnt++ % team == me) update_tile(ti, tj, p0, pw);
            }
        }
        stamp();
        __syncthreads();                 // panel sgi+1 (columns of W, pivrow/colof) and the update complete
        stamp();
    }
}

template <bool B_IN_LDS, int T16>
__global__ __launch_bounds__(CL_THREADS, 2) void chain1d_lds_kernel(
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
    const int LP = n | 1;                               // odd pitch; no padded rows: tile accesses are guarded
    cplx* Gs = reinterpret_cast<cplx*>(smem_raw);       // [n][LP] current (mixed) g
    cplx* Ws = Gs + n * LP;                             // [n][LP] work matrix
    cplx* Qs = Ws + n * LP;                             // [16][LP] pivot rows of a panel
    cplx* Bs = Qs + CL_NB * LP;                         // [n][LP] B = (E + i eta) Sb - b, when it fits (B_IN_LDS)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const cplx* alpha = a.alpha + off; const cplx* Salpha = a.Salpha + off;
    const cplx* beta = a.beta + off;   const cplx* Sbeta = a.Sbeta + off;
    const cplx* tau = a.tau + off;     const cplx* Stau = a.Stau + off;
    const cplx e = E[b];
    const cplx z = cmake(e.x, e.y + a.eta);
    // operand fetches: unconditional loads at clamped (always valid) indices, zeroed by a select
    // outside the matrix -- no branches in the MFMA loops
    auto sel = [](bool ok, cplx v) { return cmake(ok ? v.x : 0.0, ok ? v.y : 0.0); };
    auto Aat = [&](int i, int j) {
        const int o = min(i, n - 1) * n + min(j, n - 1);
        return sel(i < n && j < n, csub(cmul(z, Salpha[o]), alpha[o]));
    };
    auto Bglob = [&](int i, int j) { return csub(cmul(z, Sbeta[i * n + j]), beta[i * n + j]); };
    auto Bat = [&](int i, int j) {
        const int ic = min(i, n - 1), jc = min(j, n - 1);
        cplx v;
        if constexpr (B_IN_LDS) v = Bs[ic * LP + jc]; else v = Bglob(ic, jc);
        return sel(i < n && j < n, v);
    };
    auto tat = [&](int i, int j) {
        const int o = min(i, n - 1) * n + min(j, n - 1);
        return sel(i < n && j < n, csub(cmul(e, Stau[o]), tau[o]));
    };
    auto Gat = [&](int i, int j) { return sel(i < n && j < n, Gs[min(i, n - 1) * LP + min(j, n - 1)]); };
    auto Wat = [&](int i, int j) { return sel(i < n && j < n, Ws[min(i, n - 1) * LP + min(j, n - 1)]); };

    // ---- g0 = inv(A)
    for (int t = tid; t < n * n; t += CL_THREADS) {
        const int i = t / n, j = t - i * n;
        Ws[i * LP + j] = Aat(i, j);
        if constexpr (B_IN_LDS) Bs[i * LP + j] = Bglob(i, j);
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
        unsigned long long* st = (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && count == 10) ? a.stamps : nullptr;
        if (st && tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        // T = B g -> Ws
        d4 mr[T16], mi[T16];
        small_gemm<T16>(n, wave, lane,
                   [&](int i, int k) { return Bat(i, k); },
                   [&](int k, int j) { return Gat(k, j); },
                   [&](int, int) { return cmake(0.0, 0.0); }, mr, mi);
        gemm_store<T16>(wave, lane, mr, mi, [&](int i, int j, cplx v) { if (i < n && j < n) Ws[i * LP + j] = v; });
        __syncthreads();
        if (st && tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
        // M = A - T B^H : accumulate with the negated left operand; the results stay in registers
        // until every wave has finished reading T, then overwrite Ws
        small_gemm<T16>(n, wave, lane,
                   [&](int i, int k) { return cneg(Wat(i, k)); },                   // -T[i][k]
                   [&](int k, int j) { return cconj(Bat(j, k)); },                  // (B^H)[k][j] = conj(B[j][k])
                   [&](int i, int j) { return Aat(i, j); }, mr, mi);
        __syncthreads();
        gemm_store<T16>(wave, lane, mr, mi, [&](int i, int j, cplx v) { if (i < n && j < n) Ws[i * LP + j] = v; });
        __syncthreads();
        if (st && tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        small_inverse(n, T16, Ws, LP, Qs, rowbuf, pivrow, colof, tid, st ? st + 8 : nullptr);
        if (st && tid == 0) st[3] = __builtin_amdgcn_s_memrealtime();
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
        if (st && tid == 0) st[4] = __builtin_amdgcn_s_memrealtime();
        ++count;
    }
    // ---- Sigma_c = t g t^H, t = E Stau - tau (no eta):  X = t g -> Ws ; Sigma = X t^H -> global
    {
        d4 xr[T16], xi[T16];
        small_gemm<T16>(n, wave, lane,
                   [&](int i, int k) { return tat(i, k); },
                   [&](int k, int j) { return Gat(k, j); },
                   [&](int, int) { return cmake(0.0, 0.0); }, xr, xi);
        gemm_store<T16>(wave, lane, xr, xi, [&](int i, int j, cplx v) { if (i < n && j < n) Ws[i * LP + j] = v; });
        __syncthreads();
        cplx* out = blk + (size_t)b * a.blk_stride + off;
        small_gemm<T16>(n, wave, lane,
                   [&](int i, int k) { return Wat(i, k); },
                   [&](int k, int j) { return cconj(tat(j, k)); },
                   [&](int, int) { return cmake(0.0, 0.0); }, xr, xi);
        gemm_store<T16>(wave, lane, xr, xi, [&](int i, int j, cplx v) { if (i < n && j < n) out[i * n + j] = v; });
    }
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
    static unsigned long long* d_stamps = nullptr;
    static int want_stamps = -1;
    if (want_stamps < 0) {
        want_stamps = getenv("NEGF_CHAIN_STAMPS") ? 1 : 0;
        if (want_stamps) { (void)hipMalloc(&d_stamps, 64 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 64 * sizeof(unsigned long long)); }
    }
    a.stamps = d_stamps;
    // one instantiation per (B in LDS, number of 16-row tiles): compile-time tile loops
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        hipLaunchKernelGGL(kern, dim3(p.n_contacts, nb), dim3(CL_THREADS), smem, st, a, E, blk, iters, conv);
    };
    const int T16 = (n + 15) >> 4;                    // by the largest contact (smaller ones run padded tiles)
    if (a.b_in_lds) {
        if (T16 == 1) launch(chain1d_lds_kernel<true, 1>); else if (T16 == 2) launch(chain1d_lds_kernel<true, 2>);
        else if (T16 == 3) launch(chain1d_lds_kernel<true, 3>); else launch(chain1d_lds_kernel<true, 4>);
    } else {
        if (T16 == 1) launch(chain1d_lds_kernel<false, 1>); else if (T16 == 2) launch(chain1d_lds_kernel<false, 2>);
        else if (T16 == 3) launch(chain1d_lds_kernel<false, 3>); else launch(chain1d_lds_kernel<false, 4>);
    }
    if (d_stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[64];
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        if (h[0]) {
            auto us = [&](int i) { return h[i] ? (double)(h[i] - h[0]) / 100.0 : -1.0; };
            fprintf(stderr, "[chain stamps] sweep 10 (us): gemm1 %.2f  gemm2 %.2f  inverse %.2f  diff+mix %.2f | inverse panels:", us(1), us(2), us(3), us(4));
            for (int i = 8; i < 8 + 24 && h[i]; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[2]) / 100.0);
            fprintf(stderr, "\n");
        }
    }
}
