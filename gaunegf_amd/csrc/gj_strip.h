// Window kernel of the two-level Gauss-Jordan inverse, "strip" form (round 5): the whole workgroup is the panel
// team, lane = row, and everything a window does happens on register strips.   gfx950 / MI355X.
//
// Included by k_inverse_blocked.hip (inside its anonymous namespace, after cand_key / wave_max_u64).
// Same contract as gj_window_kernel / gj_window_la_kernel: after the launch the window's WIN columns hold
// the block column P'' of the window's COMBINED transform, pivrow / colof are extended by the window's
// pivots, and the other columns of its pivot rows (Q) are untouched -- gj_colupdate*_kernel take it from there.
// Replaces the window phase of  G = solve(E S - F - Sigma, I)  (gauNEGF/integrate.py:67-71, utils.py:52-54).
//
// What is different.  The older window kernels are RIGHT-looking inside the window: every 16-column
// sub-panel, once factored, is applied to the other 48 columns of the window by a matrix-core update that
// streams the n x 64 window block through the CU (4.9 MB moved per window and matrix at n = 500, at
// Infinity-Cache bandwidth), and their pivot step is a team barrier on an LDS counter beside a running update
// team (2.5-3 us per pivot column).  Here a window is NSUB = WIN / SW sub-windows of SW columns and a lane
// owns RPL whole rows:
//   for sub-window s:
//     1. the lane's rows of the SW columns are loaded ONCE into its register strip a[RPL][SW];
//     2. FORWARD: the strip takes the composite transform of sub-windows 0 .. s-1 in one go,
//            a[r][:] = keep(a[r][:]) + sum_k P''[r][k] Qraw[k][:]          (k over the s SW earlier columns)
//        P''[r][k] are the lane's OWN rows of the earlier columns (global, written by this lane), Qraw the RAW
//        rows piv[k] of sub-window s (staged to LDS, read back as broadcasts) -- the pair algebra of
//        gj_colupdate2_kernel, which needs no sequential Q exchange because the earlier columns already hold
//        the composite P'' (step 4);
//     3. FACTOR: SW pivot steps on the strip.  One hardware barrier per pivot column (every wave of the
//        workgroup is in the team): before the barrier each wave publishes its best candidate's key and whole
//        updated strip row, after it every lane reads the NW keys and the winner's row (broadcast reads);
//     4. the strip is stored once, and BACKWARD: the earlier columns of the window take this sub-window,
//            x[r][:] = keep_s(x[r][:]) + sum_k a[r][k] x[piv_s[k]][:]        (x = the lane's rows of columns c0 .. cs)
//        with a[r][k] still in registers and the SW pivot rows staged to LDS before anyone stores.
// All arithmetic is FP64 vector FMAs with lane = row (on this chip the FP64 vector and matrix peaks are the
// same 78.6 TF, and an FP64 MFMA blocks its SIMD's vector issue anyway); there is no operand staging, no
// update team and no C-tile round trip: 3.3 MB per window and matrix at n = 500 instead of 4.9.  The
// workgroups are LEAN (4 waves up to n = 512: two or three per CU cover each other's pivot chains -- the
// lesson of the chain kernel).
#pragma once

// maximum of a 32-bit key over the wave (all lanes active), wave-uniform: four DPP steps inside the rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the row maxima into lane 63 (the chain kernel's rs_wave_max_u32)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned strip_dpp_max_u32(unsigned k)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, CTRL, ROW_MASK, 0xF, true);
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned strip_wave_max_u32(unsigned k)
{
    k = strip_dpp_max_u32<0xB1>(k);            // quad_perm [1,0,3,2]
    k = strip_dpp_max_u32<0x4E>(k);            // quad_perm [2,3,0,1]
    k = strip_dpp_max_u32<0x141>(k);           // row_half_mirror
    k = strip_dpp_max_u32<0x140>(k);           // row_mirror: every lane of a row holds the row maximum
    k = strip_dpp_max_u32<0x142, 0xA>(k);      // row_bcast:15 into rows 1 and 3
    k = strip_dpp_max_u32<0x143, 0xC>(k);      // row_bcast:31 into rows 2 and 3: lane 63 holds the maximum
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

template <int RPL, int SW, int NW, int OCC, int PF /* chunks of 4 columns in flight in the forward / backward loops */>
__global__ __launch_bounds__(NW * 64, OCC) void gj_window_strip_kernel(
    int n, cplx* __restrict__ bufA, cplx* __restrict__ bufB /* unused */, size_t mat_stride, int* __restrict__ piv_all /* [nb][2][n]: pivrow, colof */, int* __restrict__ info, int c0, int cw,
    unsigned long long* __restrict__ stamps /* diagnostic (NEGF_GJ_STAMPS): workgroup 0, [sub-window][8]; nullptr in production */,
    int dbg /* timing ablations (NEGF_GJ_STRIP_DBG; wrong results): 1 no forward FMAs, 2 no backward FMAs, 4 no pivot steps, 8 no
               backward.  A RUN-TIME value on purpose: compiled out (constexpr 0) the uniform branches around the FMA blocks go away and
               the register allocator hoists across them -- 1314 spilled registers instead of 3 (measured) */)
{
    constexpr int T = NW * 64;
    constexpr int KMAX = WIN - SW;                   // columns of the window in front of its last sub-window
    static_assert(WIN % SW == 0 && SW % 4 == 0, "sub-windows tile the window; chunks of 4");
    __shared__ cplx cand[2][NW][SW];                 // candidate pivot rows, double buffered by column parity
    __shared__ u64 keys[2][NW];                      // candidate keys, the same
    __shared__ cplx qs[KMAX * SW];                   // staged Q rows: forward [K][SW], backward [SW][KMAX]
    __shared__ int piv_lds[WIN];                     // physical pivot row of every window column so far
    __shared__ int bad_sh;
    constexpr int TP = 5;                            // pitch of the transposition tile: 4 columns + 1 (conflict-free both ways)
    __shared__ cplx tiles[NW][64 * TP];              // one tile per wave, used by that wave only (no barriers)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* W = bufA + (size_t)blockIdx.x * mat_stride;
    int* pivrow = piv_all + (size_t)blockIdx.x * 2 * n;
    int* colof = pivrow + n;

    // Global memory is only touched in COALESCED chunks of 64 rows x 4 columns per wave: lane l moves the 16-byte piece
    // l % 4 of the rows 16 i + l / 4 (i = 0 .. 3; 16 rows x 64 contiguous bytes per instruction), and the chunk is
    // transposed through the wave's own LDS tile into "lane = row" (and back for stores).  With lane = row accesses
    // straight to global memory every instruction touches 64 different cache lines for 16 bytes each, and the
    // texture-address unit -- not the arithmetic -- is what the kernel waits for (first version: 1.1 ms per window launch
    // of 1000 x n = 500, of which the forward and backward FMAs were 35 us each).
    int row[RPL];
    unsigned goff[RPL][4];                           // byte offset of (row 16 i + l / 4 of slab q, column c0 + l % 4)
    bool gok[RPL][4];
    bool avail[RPL], winpiv[RPL], pivnow[RPL];
    const unsigned long long wbits = (unsigned long long)(size_t)W;
    char* Wb = reinterpret_cast<char*>((size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wbits >> 32)) << 32) |
                                                (unsigned)__builtin_amdgcn_readfirstlane((int)(wbits & 0xffffffffu))));
    cplx* tl = &tiles[wave][0];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        row[q] = q * T + tid;
        const bool ok = row[q] < n;
        avail[q] = ok && colof[ok ? row[q] : 0] < 0;
        winpiv[q] = false; pivnow[q] = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = q * T + wave * 64 + 16 * i + (lane >> 2);
            gok[q][i] = r < n;
            goff[q][i] = ((unsigned)min(r, n - 1) * (unsigned)n + (unsigned)(c0 + (lane & 3))) * 16u;     // rows outside re-read the last row
        }
    }
    // chunk [col, col + 4) of the window (col relative to c0) of slab q: global -> t (coalesced), t -> v (lane = row)
    // (staging values are 2-vectors, not structs: arrays of structs handed to these helpers end up on the stack)
    typedef double d2v __attribute__((ext_vector_type(2)));
    auto gload4 = [&](int q, int col, d2v (&t)[4]) __attribute__((always_inline)) {
        const bool cok = (lane & 3) < cw - col;                           // columns outside the window read (row, c0) and count as zero
        const unsigned adj = cok ? (unsigned)col * 16u : 0u - (unsigned)(lane & 3) * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const d2v v = *reinterpret_cast<const d2v*>(Wb + (goff[q][i] + adj));
            t[i].x = cok ? v.x : 0.0; t[i].y = cok ? v.y : 0.0;
        }
    };
    auto to_rows = [&](const d2v (&t)[4], cplx (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<d2v*>(&tl[(16 * i + (lane >> 2)) * TP + (lane & 3)]) = t[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) { const d2v w = *reinterpret_cast<const d2v*>(&tl[lane * TP + u]); v[u] = cmake(w.x, w.y); }
        __builtin_amdgcn_wave_barrier();
    };
    auto gstore4 = [&](int q, int col, const cplx (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) tl[lane * TP + u] = v[u];
        __builtin_amdgcn_wave_barrier();
        d2v t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const d2v*>(&tl[(16 * i + (lane >> 2)) * TP + (lane & 3)]);
        __builtin_amdgcn_wave_barrier();
        const bool cok = (lane & 3) < cw - col;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (gok[q][i] && cok) *reinterpret_cast<d2v*>(Wb + (goff[q][i] + (unsigned)col * 16u)) = t[i];
    };
    if (tid == 0) bad_sh = 0;
    if (tid < WIN) piv_lds[tid] = 0;

    const int nsub = (cw + SW - 1) / SW;
    for (int s = 0; s < nsub; ++s) {
        const int cs = c0 + s * SW, ws = min(SW, c0 + cw - cs), K = s * SW;
        auto stamp = [&](int slot) __attribute__((always_inline)) {
            if (stamps && blockIdx.x == 0 && tid == 0) stamps[s * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        };
        __syncthreads();                 // qs free (backward reads of sub-window s-1 done); piv_lds / bad_sh visible
        stamp(0);
        // ---- 1. the strips
        cplx a[RPL][SW];
        {
            d2v t[RPL][SW / 4][4];                                        // every load of the strip in flight before the first use
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int c = 0; c < SW / 4; ++c) gload4(q, K + 4 * c, t[q][c]);
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int c = 0; c < SW / 4; ++c) {
                    cplx v[4];
                    to_rows(t[q][c], v);
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[q][4 * c + u] = v[u];
                }
        }
        stamp(1);
        // ---- 2. forward: the composite transform of the earlier sub-windows
        if (s > 0) {
            for (int t = tid; t < K * SW; t += T) {
                const int k = t / SW, j = t % SW;
                qs[t] = j < ws ? W[(size_t)piv_lds[k] * n + cs + j] : cmake(0.0, 0.0);
            }
            d2v pt[PF][RPL][4];                                           // PF chunks of P'' in flight (K / 4 is a multiple of 4)
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int q = 0; q < RPL; ++q) gload4(q, 4 * p, pt[p][q]);
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                if (winpiv[q]) {
#pragma unroll
                    for (int j = 0; j < SW; ++j) a[q][j] = cmake(0.0, 0.0);
                }
            __syncthreads();
#pragma unroll 1
            for (int k0 = 0; k0 < K; k0 += 4 * PF) {
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    const int kc = k0 + 4 * p;
                    cplx pch[RPL][4];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) to_rows(pt[p][q], pch[q]);
                    if (kc + 4 * PF < K) {
#pragma unroll
                        for (int q = 0; q < RPL; ++q) gload4(q, kc + 4 * PF, pt[p][q]);
                    }
                    if (!(dbg & 1))
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int j0 = 0; j0 < SW; j0 += 4) {
                            cplx qv[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) qv[j] = qs[(kc + kk) * SW + j0 + j];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int q = 0; q < RPL; ++q) a[q][j0 + j] = cfma(a[q][j0 + j], pch[q][kk], qv[j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
        }
        stamp(2);
        // ---- 3. factor the strip: SW pivot steps, one barrier each.  The step is a latency chain (2.4 us per column in the
        // first version, at 12 matrices on an idle chip), so it is written for its dependent path:
        //   * a pivot row is NOT scaled at its step (multiplier 0, a one in the pivot column; myip remembers 1 / pivot) but
        //     once at the end of the sub-window -- the later steps act linearly on it -- so every wave runs the same
        //     select-free update and no wave is the slow one at the barrier (the chain kernel's rs_factor idiom);
        //   * the next column is updated FIRST and its candidate search started on it (|re| + |im| compared on the high
        //     word of the double, one v_max_u32 per DPP step, lowest lane among equals: ties within 2^-20 go to the lane, across
        //     waves to the lower row), so that the reduction's latency overlaps the update of the other columns;
        //   * 1 / |pivot|^2 by v_rcp_f64 and two Newton steps.
        cplx myip[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) myip[q] = cmake(1.0, 0.0);
        // candidate of this wave for column J: (high word of the largest |.|_1 among its available rows) -> key, row -> LDS
        auto search = [&](const cplx (&col)[RPL], unsigned& hi, int& qb) __attribute__((always_inline)) {
            hi = 0u; qb = 0;
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const double v = cabs1(col[q]);
                const unsigned h = (avail[q] && v == v) ? (unsigned)__double2hiint(v) : 0u;
                if (h > hi) { hi = h; qb = q; }
            }
        };
        auto publish = [&](int J, unsigned hi, int qb) __attribute__((always_inline)) {
            const unsigned m = strip_wave_max_u32(hi);
            const unsigned long long bal = __ballot(hi == m);
            const int wl = (int)__ffsll((unsigned long long)bal) - 1;     // lowest lane that holds the maximum
            cplx* cn = &cand[J & 1][wave][0];
            u64 key = 0;
            if (m != 0) {
                const int brow = __builtin_amdgcn_readlane(qb, wl) * T + wave * 64 + wl;
                key = ((u64)m << 16) | (u64)(0xFFFF - brow);
#pragma unroll
                for (int q = 0; q < RPL; ++q)
                    if (lane == wl && qb == q) {
#pragma unroll
                        for (int j = 0; j < SW; ++j) cn[j] = a[q][j];
                    }
            }
            if (lane == 0) keys[J & 1][wave] = key;
        };
        {
            cplx col[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) col[q] = a[q][0];
            unsigned hi; int qb;
            search(col, hi, qb);
            publish(0, hi, qb);
        }
        // (template recursion, not a loop: "#pragma unroll" gives up on 32 steps of this size and the strip lands on the stack)
        auto steps = [&](auto self, auto jc) __attribute__((always_inline)) -> void {
            constexpr int J = decltype(jc)::value;
            constexpr int JN = J + 1 < SW ? J + 1 : J;                   // the next column (updated first)
            if (J < ws && !(dbg & 4)) {                                   // (uniform)
                auto sst = [&](int slot) __attribute__((always_inline)) {    // in-step stamps of column 4, sub-window 1
#if defined(GJ_STRIP_STEP_STAMPS)       // diagnostic build only (NEGF_EXTRA_HIPCC_FLAGS=-DGJ_STRIP_STEP_STAMPS): costs registers
                    if constexpr (J == 4) { if (stamps && blockIdx.x == 0 && tid == 0 && s == 1) stamps[40 + slot] = __builtin_amdgcn_s_memtime(); }
#endif
                };
                sst(0);
                __syncthreads();
                sst(1);
                // the winning candidate.  key == 0: no usable row (a column of zeros / NaNs) -- the step then runs with no
                // pivot row and the matrix is reported through info
                u64 key = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) { const u64 k = keys[J & 1][w]; key = k > key ? k : key; }
                const bool none = key == 0;
                const int pphys = none ? -1 : 0xFFFF - (int)(key & 0xFFFFull);
                const int ww = none ? 0 : (pphys % T) >> 6;
                if (tid == 0) {
                    if (none && bad_sh == 0) bad_sh = cs + J + 1;
                    piv_lds[K + J] = none ? 0 : pphys;
                }
                sst(2);
                const cplx* prow = &cand[J & 1][ww][0];
                const cplx pv = prow[J];
                const double d = pv.x * pv.x + pv.y * pv.y;
                double sc = __builtin_amdgcn_rcp(d);
                sc = fma(sc, fma(-d, sc, 1.0), sc);
                sc = fma(sc, fma(-d, sc, 1.0), sc);
                const cplx ip = cmake(pv.x * sc, -pv.y * sc);
                sst(3);
                cplx coef[RPL];
                bool isp[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    isp[q] = row[q] == pphys;
                    const cplx mf = cneg(cmul(a[q][J], ip));
                    coef[q] = cmake(isp[q] ? 0.0 : mf.x, isp[q] ? 0.0 : mf.y);
                    avail[q] = avail[q] && !isp[q];
                    pivnow[q] = pivnow[q] || isp[q];
                    myip[q] = cmake(isp[q] ? ip.x : myip[q].x, isp[q] ? ip.y : myip[q].y);
                }
                // the next column first, and its candidate search started
                unsigned hi = 0u; int qb = 0;
                if constexpr (J + 1 < SW) {
                    const cplx rn = prow[JN];
                    cplx col[RPL];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) { a[q][JN] = cfma(a[q][JN], coef[q], rn); col[q] = a[q][JN]; }
                    search(col, hi, qb);
                }
#pragma unroll
                for (int j0 = 0; j0 < SW; j0 += 4) {
                    cplx rb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rb[j] = prow[j0 + j];
#pragma unroll
                    for (int q = 0; q < RPL; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j0 + j != J && !(J + 1 < SW && j0 + j == JN)) a[q][j0 + j] = cfma(a[q][j0 + j], coef[q], rb[j]);
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) a[q][J] = cmake(isp[q] ? 1.0 : coef[q].x, isp[q] ? 0.0 : coef[q].y);
                sst(4);
                if constexpr (J + 1 < SW) {
                    if (J + 1 < ws) publish(J + 1, hi, qb);
                }
                sst(5);
                if constexpr (J + 1 < SW) self(self, std::integral_constant<int, J + 1>());
            }
        };
        steps(steps, std::integral_constant<int, 0>());
        // the deferred scaling of the pivot rows
#pragma unroll
        for (int q = 0; q < RPL; ++q)
            if (pivnow[q]) {
#pragma unroll
                for (int j = 0; j < SW; ++j) a[q][j] = cmul(a[q][j], myip[q]);
            }
        stamp(3);
        // ---- 4. the strip back; this sub-window's pivots into the global bookkeeping
#pragma unroll
        for (int q = 0; q < RPL; ++q)
#pragma unroll
            for (int c = 0; c < SW; c += 4) {
                cplx v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = a[q][c + u];
                gstore4(q, K + c, v);
            }
        __syncthreads();                 // piv_lds of the last column; every wave is done with cand / keys
        stamp(4);
        if (tid < ws && bad_sh == 0) { const int p = piv_lds[K + tid]; pivrow[cs + tid] = p; colof[p] = cs + tid; }
        // ---- 5. backward: the earlier columns of the window take this sub-window
        if (s > 0 && !(dbg & 8)) {
            for (int t = tid; t < SW * KMAX; t += T) {
                const int k = t / KMAX, jj = t % KMAX;
                if (jj < K) qs[t] = k < ws ? W[(size_t)piv_lds[K + k] * n + c0 + jj] : cmake(0.0, 0.0);
            }
            d2v xt[PF][RPL][4];
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int q = 0; q < RPL; ++q) gload4(q, 4 * p, xt[p][q]);
            __syncthreads();             // every pivot row is staged before anybody stores
            stamp(5);
#pragma unroll 1
            for (int j0 = 0; j0 < K; j0 += 4 * PF) {
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    const int jc = j0 + 4 * p;
                    cplx x[RPL][4];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        to_rows(xt[p][q], x[q]);
                        if (pivnow[q]) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) x[q][u] = cmake(0.0, 0.0);
                        }
                    }
                    if (jc + 4 * PF < K) {
#pragma unroll
                        for (int q = 0; q < RPL; ++q) gload4(q, jc + 4 * PF, xt[p][q]);
                    }
                    if (!(dbg & 2))
#pragma unroll
                    for (int k = 0; k < SW; ++k) {
                        cplx qv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) qv[u] = qs[k * KMAX + jc + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int q = 0; q < RPL; ++q) x[q][u] = cfma(x[q][u], a[q][k], qv[u]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < RPL; ++q) gstore4(q, jc, x[q]);
                }
            }
        }
        stamp(6);
#pragma unroll
        for (int q = 0; q < RPL; ++q) { winpiv[q] = winpiv[q] || pivnow[q]; pivnow[q] = false; }
    }
    __syncthreads();
    if (tid == 0 && bad_sh != 0 && info[blockIdx.x] == 0) info[blockIdx.x] = bad_sh;
}
