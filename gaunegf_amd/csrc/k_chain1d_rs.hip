// 1-D chain contact self-energy ("decimation") for n_c <= 64, register-stationary version.
// gauNEGF/surfG1D.py:223-295 (g), :344-373 (sigma).  gfx950.
//
// A JOB is one fixed point (energy, contact); a workgroup (256 threads, 4 waves) runs one job at a time, SEVERAL
// workgroups per CU.  The fixed point is a chain of up to 2000 dependent sweeps, each a chain of 50 dependent pivot
// steps, so one workgroup can never fill a CU: the kernel is built to be small enough that three (n_c <= 57) or two
// workgroups share a CU and cover each other's latency chains.  That means ONE n x n work matrix in LDS per workgroup
// (40.8 KB at n_c = 50) and <= 168 VGPRs:
//   * B = (E + i eta) Sb - b is not stored per workgroup at all: wave w needs only row tile w of B, as
//     the A operand of T = B g (wave w owns row tile w of T) and, conjugated, as the B operand of
//     M = A - T B^H (wave w owns column tile w of M); it streams those 16 x n elements from the lead
//     matrices Sb, b (shared by all workgroups of the contact, L2-resident) a few k-steps ahead.
//   * the iterate g is kept twice: in the work matrix at the start of a sweep (B operand of T = B g;
//     overwritten by T, then M) and as the "old" g of the mixing step, which each lane writes and
//     reads back for its own 13 elements only: in a second LDS matrix when two workgroups share
//     the CU; when three do, the first slots in the unused rows of the work matrix and the rest in a
//     lane-private global scratch record (L2-resident).
//   * the in-place Gauss-Jordan inverse of M works on panels of 8 columns (RS_PANEL).  The wave on SIMD 0 -- the
//     CHAIN wave (rs_wave_role: the roles follow the SIMD a wave runs on, because an FP64 matrix instruction holds
//     its SIMD's vector issue) -- factors every panel: lane = row, DPP arg-max on the high word of |re| + |im|, the
//     pivot row through a 128-byte LDS line, no workgroup barrier inside.  The other three waves apply the previous
//     panel to the column tiles they OWN (the owner reads the pivot rows it needs, the Q fragment of its tile, into
//     registers before it writes: no snapshot buffer); all four bring the next panel's columns up to date first
//     (look-ahead).  Products and updates are three real matrix instructions per complex tile and k-step (3M).
//   * launches with more jobs than resident slots run them ROUND ROBIN through a device-side queue (ChainRsArgs,
//     rs_rr_pop / rs_rr_push): a persistent workgroup per slot, quanta of 100 sweeps, the iterate of a job that is
//     set aside waits in the job's output block.
// Per sweep
//     T     = B g ;  M = A - T B^H          2 complex GEMMs on the FP64 matrix cores
//     g_new = inv(M)                         blocked Gauss-Jordan, partial pivoting (izamax rule)
//     diff  = max |g_new - g| / max(|g_new|, 1e-12) ;  g = r g_new + (1-r) g
// Rows are never swapped: g_new[i][j] = W[pivrow[i]][colof[j]] is resolved when g_new is read.
// The stopping rule only needs "diff > conv" / "diff <= conv"; both are evaluated on the squares
// (|d|^2 > conv^2 max(|g_new|^2, 1e-24)), which is the same predicate without sqrt and divide.
//
// Every job stops on ITS OWN convergence (the reference's vmap runs all energies until the
// slowest lane converges; results are identical because a converged lane is frozen there).
#include "negf_common.h"
#include "wave_utils.h"
#include <algorithm>

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = 4;
#ifndef RS_PANEL
#define RS_PANEL 8
#endif
constexpr int RS_NB = RS_PANEL;           // panel width of the small inverse: half a column tile (8) or a whole one (16)
#ifndef RS_PRIO
#define RS_PRIO 1
#endif
#ifndef RS_REMAINDER
#define RS_REMAINDER 1
#endif
#ifndef RS_ROW_MODE
#define RS_ROW_MODE 0                 // pivot row of the factoring wave: 0 LDS line (round 2), 1 v_readlane, 2 ds_bpermute
#endif
#ifndef RS_LA_FLAGS
#define RS_LA_FLAGS 0                 // look-ahead of the small inverse synchronised by two workgroup barriers (0) or by LDS counters
                                      // (1: built and measured in round 4 -- parity green, the C3 launch 666 ms against 668: with
                                      // three workgroups per CU a barrier's wait is another workgroup's issue slot, not idle time)
#endif
#ifndef RS_RR
#define RS_RR 1
#endif
#ifndef RS_PHAT
#define RS_PHAT 1                     // 1: the factoring wave stores P - E (E: a one at every (pivot row, its column) of the panel), so that the
#endif                                //    updates are PURE accumulations W += (P - E) Q -- the pivot rows' old content cancels in the
                                      //    product instead of being masked per tile (8 compares + 16 selects of the 42 vector
                                      //    instructions a rank-8 tile update carried beside its 6 matrix instructions); the ones ride
                                      //    along as constants (every later update is additive, and neither a Q row nor a P operand of a
                                      //    later panel contains such an element) and are added back where the inverse is read (gather_mix)
#ifndef RS_UPD_3M
#define RS_UPD_3M 1                   // trailing / look-ahead updates (K = 8): 1 = three real products per tile and k-step (3M: operand
#endif                                //    sums and a 12-instruction recombination per tile), 0 = four (no vector work per tile at all)
#ifndef RS_ABLATE
#define RS_ABLATE 0                   // diagnostic builds only (wrong results; timing with force_iters): 1 = no panel factoring (the chain
#endif                                //    wave's pivot steps), 2 = no trailing / look-ahead updates, 4 = no products, 8 = no mixing phase
#ifndef RS_STAMPS
#define RS_STAMPS 0                   // 1: diagnostic build -- the phase / cycle stamps of NEGF_CHAIN_STAMPS=1 are compiled in
#endif                                //    (NEGF_EXTRA_HIPCC_FLAGS=-DRS_STAMPS=1 python -m gaunegf_amd.build --force); the production
                                      //    kernel carries none of their branches

struct ChainRsArgs {
    const cplx *alpha, *Salpha, *beta, *Sbeta, *tau, *Stau;   // concatenated per contact
    const int* nc;
    const int* blk_off;
    int n_contacts, blk_stride;
    double eta, conv, relFactor;
    int max_iter, force_iters;
    cplx* gold;                      // [workgroups][KS][256] lane-private copies of the iterate (GOLD_GLOBAL kernels)
    int gold_lds_off, gold_lds_slots; // the first slots of a lane's copy live in LDS at this element offset
    const int* order;                // launch slot -> job (energy * n_contacts + contact), longest jobs first; or null
    // surface Green's function cache (negf_set_chain_cache): gc_mode 1 = this launch is a miss, every job stores its
    // final iterate g into gcache [energy][blk_stride] (the layout of blk); 2 = a hit, every job loads g from there and
    // only runs Sigma = t g t^H -- the same instructions on the same operands as the last pass of a miss
    cplx* gcache;
    int gc_mode;
    unsigned long long* stamps;      // diagnostic (RS_STAMPS build + NEGF_CHAIN_STAMPS): wall-clock stamps of workgroup (0,0), 10th sweep
    int stamp_sweep;                 // diagnostic: the sweep of job 0 whose phases are stamped (NEGF_CHAIN_STAMP_SWEEP, default 10)
    int simd_roles;                  // 1: wave roles follow the SIMD a wave runs on (see rs_wave_role), 0: the wave number
    // round-robin execution (rr_quantum > 0): the launch is one PERSISTENT workgroup per resident slot; the jobs wait
    // in a FIFO queue (entries 0 .. rr_jobs-1: the launch order, then rr_ring), a workgroup runs a job for rr_quantum
    // sweeps and -- if other jobs are waiting -- leaves the iterate in the job's (still unused) output block, puts the
    // job at the back of the queue and takes the one at the front.  Jobs of unknown length then finish in order of
    // their length and the chip stays full until fewer jobs than slots are left, whatever the launch order was.
    int rr_quantum;
    struct RsQueue* rr_q;            // (in device memory: its fields are only needed when a job starts or is set aside)
};
struct RsQueue {
    unsigned head, tail;             // next entry to pop / to push
    unsigned jobs, cap;              // jobs of the launch (entries 0 .. jobs-1 of the queue), entries of the ring
    unsigned long long ring[];       // (sweep count << 32) | job; ~0 = not written yet
};

// front of the job queue (lane 0 of a workgroup): (count << 32) | job, or -1 when the queue is empty.  An empty queue
// stays empty: a job is only ever pushed by a workgroup that saw other jobs waiting, so a workgroup that finds it
// empty is done (every job is then held by a running workgroup, which finishes it itself).
__device__ __forceinline__ long long rs_rr_pop(RsQueue* q, const int* order)
{
    const unsigned jobs = q->jobs;
    for (;;) {
        const unsigned h = __hip_atomic_load(&q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_load(&q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h >= t) return -1;
        unsigned expect = h;
        if (!__hip_atomic_compare_exchange_strong(&q->head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            continue;
        if (h < jobs) return order ? order[h] : (int)h;
        // (the pusher reserved this entry before writing it: it is a running workgroup between two instructions)
        unsigned long long v;
        do { v = __hip_atomic_load(&q->ring[h - jobs], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (v == ~0ull);
        return (long long)v;
    }
}
// are other jobs waiting (and is there room to queue this one)?
__device__ __forceinline__ bool rs_rr_waiting(RsQueue* q)
{
    const unsigned h = __hip_atomic_load(&q->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t = __hip_atomic_load(&q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return h < t && t - q->jobs + 1024u < q->cap;
}
__device__ __forceinline__ void rs_rr_push(RsQueue* q, int job, int count)
{
    const unsigned t = __hip_atomic_fetch_add(&q->tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&q->ring[t - q->jobs], ((unsigned long long)(unsigned)count << 32) | (unsigned)job,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void rs_rr_init_kernel(RsQueue* q, unsigned cap, unsigned jobs)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) q->ring[i] = ~0ull;
    if (i == 0) { q->head = 0u; q->tail = jobs; q->jobs = jobs; q->cap = cap; }
}

// maximum of a 32-bit key over the wave (all lanes active), wave-uniform result.  Four DPP steps inside the rows
// of 16 lanes leave the row maximum in every lane of a row; row_bcast:15 / row_bcast:31 (the GFX9 wave-reduction
// steps) then carry it across the rows into lane 63: seven vector instructions and one v_readlane.  With
// bound_ctrl and a zero "old" value the compiler folds each lane move into its v_max_u32 (v_max_u32_dpp);
// zero is the identity of the maximum, and the row steps read no invalid lane.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned rs_dpp_max_u32(unsigned k)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, CTRL, ROW_MASK, 0xF, true);
    return o > k ? o : k;
}

__device__ __forceinline__ unsigned rs_wave_max_u32(unsigned k)
{
    k = rs_dpp_max_u32<0xB1>(k);            // quad_perm [1,0,3,2]
    k = rs_dpp_max_u32<0x4E>(k);            // quad_perm [2,3,0,1]
    k = rs_dpp_max_u32<0x141>(k);           // row_half_mirror
    k = rs_dpp_max_u32<0x140>(k);           // row_mirror: every lane of a row holds the row maximum
    k = rs_dpp_max_u32<0x142, 0xA>(k);      // row_bcast:15 into rows 1 and 3
    k = rs_dpp_max_u32<0x143, 0xC>(k);      // row_bcast:31 into rows 2 and 3: lane 63 holds the maximum
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

// ---- remainder tiles.  n_c = 50 is three tiles of 16 and two more rows / columns; a fourth 16 x 16 tile for
// them costs as much as a full one.  v_mfma_f64_4x4x4_4b_f64 has the operand maps of the 16x16x4 instruction
// (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15]) and computes the four DIAGONAL 4 x 4 blocks of the
// 16 x 16 product, D_b[i][j] at lane 16 i + 4 b + j, in a quarter of the time (probed on MI355X,
// scripts/probe/mfma4x4_probe.hip).  Used in two ways when the last tile holds <= 4 rows / columns:
//   row strip  (last ROW tile):    every 4-row block of the A operand is loaded with the SAME rows
//              (row R0 + (l&3)); the B operand is the normal 16-column fragment.  D lane l holds element
//              (R0 + (l>>4), C0 + (l&15)) -- exactly component r = 0 of the 16 x 16 C layout.
//   column strip (last COLUMN tile): every 4-column block of the B operand holds the SAME columns
//              (column C0 + (l&3)); the A operand is the normal 16-row fragment.  D lane l holds element
//              (R0 + 4 ((l>>2)&3) + (l>>4), C0 + (l&3)).
// Both leave their result in component 0 of the tile's accumulator.
__device__ __forceinline__ void zmfma4(double& ar, double& ai, cplx pa, cplx qb)
{
    ar = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.x, qb.x, ar, 0, 0, 0);
    ar = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.y, -qb.y, ar, 0, 0, 0);
    ai = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.x, qb.y, ai, 0, 0, 0);
    ai = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.y, qb.x, ai, 0, 0, 0);
}
// pa * conj(b)
__device__ __forceinline__ void zmfma4_conjb(double& ar, double& ai, cplx pa, cplx b)
{
    ar = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.x, b.x, ar, 0, 0, 0);
    ar = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.y, b.y, ar, 0, 0, 0);
    ai = __builtin_amdgcn_mfma_f64_4x4x4f64(pa.y, b.x, ai, 0, 0, 0);
    ai = __builtin_amdgcn_mfma_f64_4x4x4f64(-pa.x, b.y, ai, 0, 0, 0);
}
#define RS_M3S(A, B, C, PR, PI, PS, QR, QI, QS) do { double a_ = (A)[0], b_ = (B)[0], c_ = (C)[0]; mfma3s(a_, b_, c_, PR, PI, PS, QR, QI, QS); (A)[0] = a_; (B)[0] = b_; (C)[0] = c_; } while (0)
#define RS_ZMFMA4(ACCR, ACCI, PA, QB) do { double r_ = (ACCR)[0], i_ = (ACCI)[0]; zmfma4(r_, i_, PA, QB); (ACCR)[0] = r_; (ACCI)[0] = i_; } while (0)
#define RS_ZMFMA4C(ACCR, ACCI, PA, QB) do { double r_ = (ACCR)[0], i_ = (ACCI)[0]; zmfma4_conjb(r_, i_, PA, QB); (ACCR)[0] = r_; (ACCI)[0] = i_; } while (0)

// ---- complex products by three real ones ("3M"): with p = pr + i pi, q = qr + i qi
//        a = sum pr qr,   b = sum pi qi,   c = sum (pr + pi)(qr + qi)     =>  p q       = (a - b) + i (c - a - b)
//        a, b as above,                    c = sum (pr + pi)(qr - qi)     =>  p conj(q) = (a + b) + i (c - a + b)
// i.e. 3 matrix instructions per tile and k-step instead of 4: a quarter of the matrix-pipe time of every product
// and update, for one or two additions per operand fragment.  The FP64 matrix instruction holds its SIMD's vector
// issue for most of its 64 cycles (rs_wave_role below), so matrix-pipe time is what the sweep is made of.
__device__ __forceinline__ void mfma3(d4& a, d4& b, d4& c, double pr, double pi, double ps, double qr, double qi, double qs)
{
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(pr, qr, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f64_16x16x4f64(pi, qi, b, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(ps, qs, c, 0, 0, 0);
}
#ifndef RS_3M_GEMM
#define RS_3M_GEMM 1
#endif
// the same on the 4x4x4 instruction (remainder strips, one value per lane)
__device__ __forceinline__ void mfma3s(double& a, double& b, double& c, double pr, double pi, double ps, double qr, double qi, double qs)
{
    a = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, qr, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f64_4x4x4f64(pi, qi, b, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(ps, qs, c, 0, 0, 0);
}

// Hide a loop-invariant value from the optimiser: without this LLVM hoists every (tile, k-step)
// LDS address of the sweep out of the fixed-point loop -- hundreds of live address registers that
// are then spilled to scratch and reloaded inside the MFMA loops.
template <class T>
__device__ __forceinline__ T rs_opaque(T v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ double rs_readlane_f64(double v, int srclane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane),
                            __builtin_amdgcn_readlane(__double2loint(v), srclane));
}

// ---- panel [p0, p0+pw) factored by ONE wave: lane = row, 16 complex per lane, no barrier inside.  A wave
// alone issues one instruction per ~4 cycles whatever its kind, and the vector ALU of its SIMD is what
// the co-resident workgroups compete for, so the column step is written for instruction count:
//   * pivot search: |re|+|im| (izamax metric) compared on the HIGH WORD of the double -- sign 0, exponent,
//     20 mantissa bits -- with one v_max_u32 per DPP step; the pivot is the lowest row whose high word
//     equals the maximum (ballot + find-first): the LAPACK choice up to ties within 2^-20 relative, which
//     go to the lower row as LAPACK's exact ties do;
//   * the pivot row reaches the other lanes through a 256-byte LDS line (the pivot lane writes its 16
//     values, every lane reads them back: 32 LDS instructions instead of 68 v_readlane, and off the VALU);
//   * 1/|pivot|^2 by v_rcp_f64 and two Newton steps (the pivots of these matrices are far from the
//     overflow / denormal range the IEEE division sequence guards).
// A pivot row is not scaled at its column step (multiplier 0, a one in the pivot column) but once at
// the end of the panel: the later steps act linearly on it, and every lane runs the same select-free update.
template <int P>
__device__ __forceinline__ void rs_factor(int n, cplx* W, int* pivrow, int* colof, cplx* rowline /*[16] LDS*/,
                                          int p0, int pw, int lane, unsigned long long* fst = nullptr /* diagnostic: cycle stamps of column step 4 */)
{
    const int r = rs_opaque(lane);                      // (see rs_opaque: nothing derived from the lane index is
    cplx a[RS_NB];                                      //  hoisted out of the fixed-point loop and kept alive)
    bool avail = r < n && colof[r] < 0;
    cplx myip = cmake(1.0, 0.0);
    int mycol = -1;                                     // the panel column this lane's row is the pivot row of
    cplx* wrow = W + r * P + p0;                        // rows >= n are zero padding
#pragma unroll
    for (int s = 0; s < RS_NB; ++s) {
        const cplx v = wrow[s];
        const bool ok = (r < n) & (s < pw);
        a[s] = cmake(ok ? v.x : 0.0, ok ? v.y : 0.0);
    }
#pragma unroll
    for (int j = 0; j < RS_NB; ++j) {
        if (j < pw) {
            const bool stamp_here = fst && j == 4;
            unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0;
            if (stamp_here) tq0 = __builtin_amdgcn_s_memtime();
            const double v = cabs1(a[j]);
            const unsigned hi = (avail && v == v) ? (unsigned)__double2hiint(v) : 0u;
            const unsigned m = rs_wave_max_u32(hi);
            int pphys;
            if (m != 0) {
                pphys = (int)__ffsll((unsigned long long)__ballot(hi == m)) - 1;
            } else {                                    // no usable candidate (zero / NaN column): lowest available row
                const unsigned long long av = __ballot(avail);
                pphys = av ? (int)__ffsll(av) - 1 : 0x7fffffff;
            }
            pphys = __builtin_amdgcn_readfirstlane(pphys);
            if (stamp_here) tq1 = __builtin_amdgcn_s_memtime();
            const bool is_piv = r == pphys;
            cplx rb[RS_NB];
#if RS_ROW_MODE == 0
            // through a 256-byte LDS line: one lane writes its 16 values, all lanes read them back.  The wave-level
            // barriers keep the compiler from ordering the two sides of the divergent branch the other way round
            // (it does, without them), the LDS then executes the wave's instructions in order
            __builtin_amdgcn_wave_barrier();            // the reads of the previous column step are issued
            if (is_piv) {
                pivrow[p0 + j] = pphys; colof[pphys] = p0 + j;
#pragma unroll
                for (int s = 0; s < RS_NB; ++s) rowline[s] = a[s];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < RS_NB; ++s) rb[s] = rowline[s];
#elif RS_ROW_MODE == 1
            // through v_readlane (wave-uniform results, scalar operands of the FMAs): measured 10 cycles each
            if (is_piv) { pivrow[p0 + j] = pphys; colof[pphys] = p0 + j; }
#pragma unroll
            for (int s = 0; s < RS_NB; ++s)
                rb[s] = cmake(rs_readlane_f64(a[s].x, pphys), rs_readlane_f64(a[s].y, pphys));
#else
            // through ds_bpermute_b32: every lane reads the pivot lane's register over the LDS crossbar -- no memory,
            // no write -> wait -> read round trip, no wave barriers: one pass of 64 pipelined LDS instructions
            if (is_piv) { pivrow[p0 + j] = pphys; colof[pphys] = p0 + j; }
            const int baddr = pphys << 2;
#pragma unroll
            for (int s = 0; s < RS_NB; ++s) {
                rb[s].x = __hiloint2double(__builtin_amdgcn_ds_bpermute(baddr, __double2hiint(a[s].x)),
                                           __builtin_amdgcn_ds_bpermute(baddr, __double2loint(a[s].x)));
                rb[s].y = __hiloint2double(__builtin_amdgcn_ds_bpermute(baddr, __double2hiint(a[s].y)),
                                           __builtin_amdgcn_ds_bpermute(baddr, __double2loint(a[s].y)));
            }
#endif
            if (stamp_here) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq2 = __builtin_amdgcn_s_memtime(); }
            const cplx pv = rb[j];
            const double d = pv.x * pv.x + pv.y * pv.y;
            double sc = __builtin_amdgcn_rcp(d);
            sc = fma(sc, fma(-d, sc, 1.0), sc);
            sc = fma(sc, fma(-d, sc, 1.0), sc);
            const cplx ip = cmake(pv.x * sc, -pv.y * sc);
            const cplx mf = cneg(cmul(a[j], ip));
            const cplx coef = cmake(is_piv ? 0.0 : mf.x, is_piv ? 0.0 : mf.y);
            if (stamp_here) { asm volatile("" :: "v"(coef.x), "v"(coef.y)); tq3 = __builtin_amdgcn_s_memtime(); }
#pragma unroll
            for (int s = 0; s < RS_NB; ++s) a[s] = cfma(a[s], coef, rb[s]);
            if (stamp_here) {
#pragma unroll
                for (int s = 0; s < RS_NB; ++s) asm volatile("" :: "v"(a[s].x), "v"(a[s].y));
                tq4 = __builtin_amdgcn_s_memtime();
                if (lane == 0) { fst[0] = tq0; fst[1] = tq1; fst[2] = tq2; fst[3] = tq3; fst[4] = tq4; }
            }
            a[j] = is_piv ? cmake(1.0, 0.0) : coef;
            myip = cmake(is_piv ? ip.x : myip.x, is_piv ? ip.y : myip.y);
            mycol = is_piv ? j : mycol;
            avail = avail && !is_piv;
        }
    }
    if (r < n) {
#pragma unroll
        for (int s = 0; s < RS_NB; ++s)
            if (s < pw) {
                cplx v = cmul(a[s], myip);                  // the deferred pivot-row scaling
                if (RS_PHAT && s == mycol) v.x -= 1.0;      // P - E (see RS_PHAT)
                wrow[s] = v;
            }
    }
}

// ---- trailing update with panel [p0, p0+pw), in place:
//        W[i][col] = (i pivot row of the panel ? 0 : W[i][col]) + P[i][:] Q[:][col]
// A Q fragment holds the panel's pivot rows in the columns of one column tile (B operand); for the
// column-strip tile TR every 4-column block holds the same columns TR*16 + (l&3) (see mfma3s).
template <int P, int NKS, int TR /* last tile when it is a remainder strip, else -1 */>
__device__ __forceinline__ void rs_load_qf(const cplx* W, const int* pivrow, int tj, int p0, int pw, int fi, int fk,
                                           cplx (&qf)[NKS])
{
    const int col = (TR >= 0 && tj == TR) ? TR * 16 + (fi & 3) : tj * 16 + fi;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int k = ks * 4 + fk;
        const cplx v = W[pivrow[p0 + k] * P + col];          // k >= pw: some valid row, zeroed below
        const bool ok = k < pw;
        qf[ks] = cmake(ok ? v.x : 0.0, ok ? v.y : 0.0);
    }
}

// one tile (ti, tj): a full 16 x 16 tile, or a row / column strip (one value per lane, corner: one block)
__device__ __forceinline__ void rs_wait_count(const unsigned* cnt, unsigned target)
{
    // (LDS counter of the look-ahead, see rs_inverse: spins only while another wave of the workgroup is behind)
    while (*reinterpret_cast<const volatile unsigned*>(cnt) < target) __builtin_amdgcn_s_sleep(1);
}

template <int P, int NKS, int TR>
__device__ __forceinline__ void rs_update_tile(int n, cplx* W, const int* colof, int ti, int tj, int p0, int pw,
                                               int fi, int fk, const cplx (&qf)[NKS], int clo, int chi /* columns [clo, chi) are stored */,
                                               const unsigned* wait_cnt = nullptr, unsigned wait_target = 0 /* stores wait for *wait_cnt >= wait_target */)
{
    const bool rowstrip = TR >= 0 && ti == TR, colstrip = TR >= 0 && tj == TR;
    if (!rowstrip && !colstrip) {
        cplx* cbase = W + (ti * 16 + fk) * P + tj * 16 + fi;
        const cplx* pbase = W + (ti * 16 + fi) * P + p0 + fk;
        cplx cv[4], pa[NKS];
        int cf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (!RS_PHAT) cf[r] = colof[ti * 16 + fk + 4 * r]; cv[r] = cbase[4 * r * P]; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) pa[ks] = pbase[ks * 4];
        constexpr bool M3 = RS_UPD_3M && P > 35;            // 3M (mfma3) in the 168-VGPR kernels
        d4 ua, ub = {0, 0, 0, 0}, uc;                       // accumulators, seeded with the old tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool keep = RS_PHAT || !(cf[r] >= p0 && cf[r] < p0 + pw);
            ua[r] = keep ? cv[r].x : 0.0; uc[r] = keep ? (M3 ? cv[r].x + cv[r].y : cv[r].y) : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (M3) mfma3(ua, ub, uc, pa[ks].x, pa[ks].y, pa[ks].x + pa[ks].y, qf[ks].x, qf[ks].y, qf[ks].x + qf[ks].y);
            else zmfma(ua, uc, pa[ks], qf[ks]);
        }
        if (wait_cnt) rs_wait_count(wait_cnt, wait_target);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + fk + 4 * r;
            const int cj = tj * 16 + fi;
            if (i < n && cj < n && cj >= clo && cj < chi) cbase[4 * r * P] = M3 ? cmake(ua[r] - ub[r], uc[r] - ua[r] - ub[r]) : cmake(ua[r], uc[r]);
        }
    } else {
        const int row = rowstrip ? TR * 16 + fk : ti * 16 + 4 * (fi >> 2) + fk;
        const int col = colstrip ? TR * 16 + (fi & 3) : tj * 16 + fi;
        const bool mine = !(rowstrip && colstrip) || (fi >> 2) == 0;      // the corner block exists four times
        const cplx* prow = W + (rowstrip ? TR * 16 + (fi & 3) : ti * 16 + fi) * P + p0 + fk;
        cplx* cptr = W + row * P + col;
        const int cf = RS_PHAT ? -1 : colof[row];
        const cplx cv = *cptr;
        cplx pa[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) pa[ks] = prow[ks * 4];
        const bool keep = RS_PHAT || !(cf >= p0 && cf < p0 + pw);
        double ua = keep ? cv.x : 0.0, ub = 0.0, uc = keep ? cv.x + cv.y : 0.0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            mfma3s(ua, ub, uc, pa[ks].x, pa[ks].y, pa[ks].x + pa[ks].y, qf[ks].x, qf[ks].y, qf[ks].x + qf[ks].y);
        if (wait_cnt) rs_wait_count(wait_cnt, wait_target);
        if (mine && row < n && col < n && col >= clo && col < chi) *cptr = cmake(ua - ub, uc - ua - ub);
    }
}

// column tile tj, all row tiles, by the wave that owns the column tile in this stage.  The Q fragment is read
// into registers before the first store, so the owner needs no snapshot of the pivot rows.
template <int T16, int P, int NKS, int TR>
__device__ __forceinline__ void rs_update_col(int n, cplx* W, const int* pivrow, const int* colof,
                                              int tj, int p0, int pw, int lane, int clo, int chi)
{
    constexpr int FT = TR >= 0 ? TR : T16;                   // full row tiles
    const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
    cplx qf[NKS];
    rs_load_qf<P, NKS, TR>(W, pivrow, tj, p0, pw, fi, fk, qf);
    if (TR >= 0 && tj == TR) {                               // the column strip: every tile on the 4x4x4 instruction
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) {
            rs_update_tile<P, NKS, TR>(n, W, colof, ti, tj, p0, pw, fi, fk, qf, clo, chi);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    const int col = tj * 16 + fi;
    const bool colin = col >= clo && col < chi;             // this lane's column is one of those to be updated
    constexpr bool M3 = RS_UPD_3M && P > 35;                 // 3M (mfma3) in the 168-VGPR kernels
    double qs[NKS];                                          // 3M: re + im of the Q fragment, once per column tile
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qs[ks] = qf[ks].x + qf[ks].y;
    cplx* cbase = W + fk * P + col;                          // C tile element (ti*16 + fk + 4r, col)
    const cplx* pbase = W + fi * P + p0 + fk;                // P operand element (ti*16 + fi, p0 + ks*4 + fk)
    const int* cfb = colof + fk;
    // the operands of row tile ti+1 are requested before tile ti is stored (LDS operations of a wave
    // execute in order, and tile ti+1 shares no element with tile ti), so that the loads overlap the MFMAs
    cplx cv[2][4], pa[2][NKS];
    int cf[2][4];
    auto fetch = [&](int ti, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (!RS_PHAT) cf[s][r] = cfb[ti * 16 + 4 * r]; cv[s][r] = cbase[(ti * 16 + 4 * r) * P]; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) pa[s][ks] = pbase[ti * 16 * P + ks * 4];   // k >= pw pairs with qf == 0 (finite element)
    };
    fetch(0, 0);
#pragma unroll
    for (int ti = 0; ti < FT; ++ti) {
        const int s = ti & 1;
        if (ti + 1 < FT) fetch(ti + 1, s ^ 1);
        d4 ua, ub = {0, 0, 0, 0}, uc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool keep = RS_PHAT || !(cf[s][r] >= p0 && cf[s][r] < p0 + pw);
            ua[r] = keep ? cv[s][r].x : 0.0; uc[r] = keep ? (M3 ? cv[s][r].x + cv[s][r].y : cv[s][r].y) : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (M3) mfma3(ua, ub, uc, pa[s][ks].x, pa[s][ks].y, pa[s][ks].x + pa[s][ks].y, qf[ks].x, qf[ks].y, qs[ks]);
            else zmfma(ua, uc, pa[s][ks], qf[ks]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + fk + 4 * r;
            // with a remainder strip (TR >= 0) the full tiles lie inside the matrix: rows, columns < 16 TR < n
            if (colin && (TR >= 0 || (i < n && col < n))) cbase[(ti * 16 + 4 * r) * P] = M3 ? cmake(ua[r] - ub[r], uc[r] - ua[r] - ub[r]) : cmake(ua[r], uc[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (TR >= 0) rs_update_tile<P, NKS, TR>(n, W, colof, TR, tj, p0, pw, fi, fk, qf, clo, chi);    // the row strip
}

// In-place blocked Gauss-Jordan reduction of the n x n matrix W (LDS, pitch P) with implicit
// pivoting.  On return  inv[i][j] = W[pivrow[i]][colof[j]].  All 256 threads call it; colof[] must be
// -1 and visible (a barrier since it was reset).  Stage s: all waves apply panel s to the column tile of
// panel s+1 (one row tile each, two barriers), then one wave factors panel s+1 while the other waves apply
// panel s to the remaining column tiles (one owner per column tile, no barrier), one barrier at the end.
template <int T16, int P, int TR>
__device__ __forceinline__ void rs_inverse(int n, cplx* W, int* pivrow, int* colof, cplx* rowline, int tid,
                                           int wave /* role number of this wave, see rs_wave_role */, bool fixed_fw,
                                           unsigned* la_cnt /* [2] LDS */, unsigned& la_epoch,
                                           unsigned long long* st = nullptr)
{
    const int lane = tid & 63;
    int sti = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if (st && tid == 0) st[sti] = __builtin_amdgcn_s_memrealtime(); ++sti; };
    const int npanels = (n + RS_NB - 1) / RS_NB;
    const int ntiles = (n + 15) >> 4;
    for (int sgi = -1; sgi < npanels; ++sgi) {
        const bool has_cur = sgi >= 0, has_next = sgi + 1 < npanels;
        const int p0 = has_cur ? sgi * RS_NB : 0, pw = has_cur ? min(RS_NB, n - p0) : 0;
        const int n0 = (sgi + 1) * RS_NB, nw = has_next ? min(RS_NB, n - n0) : 0;
        const int tp = p0 >> 4, tl = n0 >> 4;                   // column tiles of the panel and of the next one
        // the wave that factors panel sgi+1: the chain wave (role RS_WAVES-1), or -- roles by wave number -- each in turn
        const int fw = fixed_fw ? RS_WAVES - 1 : (sgi + 1) & (RS_WAVES - 1);
        if (has_cur && has_next) {
            // look-ahead: the columns of panel sgi+1 (a whole column tile, or one half of one), one row tile per wave
            // (T16 <= 4 = number of waves)
            const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
            cplx qf[RS_NB / 4];
            rs_load_qf<P, RS_NB / 4, TR>(W, pivrow, tl, p0, pw, fi, fk, qf);
#if RS_LA_FLAGS
            // No workgroup barrier in the look-ahead (there were two, seven times a sweep): two LDS counters instead.
            // la_cnt[0] counts the waves whose Q fragment -- the panel's pivot rows in the next panel's columns, rows that
            // other waves are about to overwrite -- is on its way (a wave's LDS operations execute in order, so its
            // counter increment is behind its reads); a wave stores its look-ahead tile only when all four are.
            // la_cnt[1] counts the waves that have stored; only the FACTORING wave waits for it -- the others go straight
            // on to the trailing update, which touches neither the next panel's columns nor anybody else's pivot rows.
            la_epoch += RS_WAVES;
            if (lane == 0) atomicAdd(&la_cnt[0], 1u);
            if (wave < T16 && wave * 16 < n) rs_update_tile<P, RS_NB / 4, TR>(n, W, colof, wave, tl, p0, pw, fi, fk, qf, n0, n0 + RS_NB, &la_cnt[0], la_epoch);
            if (lane == 0) atomicAdd(&la_cnt[1], 1u);
            if (wave == fw) rs_wait_count(&la_cnt[1], la_epoch);
#else
            __syncthreads();
            if (!(RS_ABLATE & 2) && wave < T16 && wave * 16 < n) rs_update_tile<P, RS_NB / 4, TR>(n, W, colof, wave, tl, p0, pw, fi, fk, qf, n0, n0 + RS_NB);
            __syncthreads();
#endif
        }
        if (has_cur) {
            // the other columns, tile by tile, one owner per tile (the owner reads its Q fragment before it writes):
            // dealt to the three waves that do not factor, to all four after the last panel.  A tile's columns minus
            // the panel's own and minus the look-ahead columns (panels of 8: one half of the panel's tile remains when
            // the panel is its upper half, one half of the next tile when the look-ahead took its lower half).
            const int team = has_next ? RS_WAVES - 1 : RS_WAVES;
            const int me = has_next ? ((wave - fw - 1) & (RS_WAVES - 1)) : wave;
            int cnt = 0;
#pragma unroll 1
            for (int tj = 0; tj < ntiles; ++tj) {
                int clo = tj * 16, chi = tj * 16 + 16;
                if (tj == tp) {
                    if (RS_NB == 16) continue;
                    if (p0 & 8) chi = p0; else clo = p0 + RS_NB;
                }
                if (has_next && tj == tl) {
                    if (RS_NB == 16) continue;
                    if (n0 & 8) chi = min(chi, n0); else clo = max(clo, n0 + RS_NB);
                }
                if (clo >= chi || clo >= n) continue;
                const bool mine = (!has_next || wave != fw) && (cnt % team == me);
                ++cnt;
                if (mine && !(RS_ABLATE & 2)) {
                    // a narrow last panel (<= 4 columns) runs one k-step
                    if (pw <= 4) rs_update_col<T16, P, 1, TR>(n, W, pivrow, colof, tj, p0, pw, lane, clo, chi);
                    else rs_update_col<T16, P, RS_NB / 4, TR>(n, W, pivrow, colof, tj, p0, pw, lane, clo, chi);
                }
            }
        }
        if (has_next && wave == fw) {
            if (st && lane == 0) st[16 + 2 * (sgi + 1)] = __builtin_amdgcn_s_memrealtime();
            // the factoring wave is its workgroup's critical path (the others wait for it at the barrier):
            // it goes first when it shares its SIMD's issue slots with waves of the other workgroups
            if (RS_PRIO) __builtin_amdgcn_s_setprio(3);
            if (!(RS_ABLATE & 1)) rs_factor<P>(n, W, pivrow, colof, rowline, n0, nw, lane, (st && sgi + 1 == 1) ? st + 40 : nullptr);
            else if (lane < nw) { pivrow[n0 + lane] = n0 + lane; colof[n0 + lane] = n0 + lane; }
            if (RS_PRIO) __builtin_amdgcn_s_setprio(0);
            if (st && lane == 0) st[17 + 2 * (sgi + 1)] = __builtin_amdgcn_s_memrealtime();
        }
        stamp();
        __syncthreads();                 // panel sgi+1 (columns of W, pivrow/colof) and the update complete
    }
}

// P: compile-time pitch of the work matrix (odd, >= n): every tile / k-step offset is an immediate of the
// DS instruction, a lane needs ONE address register per operand stream.  The matrix has 16*T16 rows
// of which rows >= n (and the columns >= n of a row) stay zero: operands of the padded tiles are read
// without clamps or selects and are finite.
// RR: the round-robin instantiation (a persistent workgroup that takes its jobs from the queue); the plain one -- a launch
// whose jobs all fit the resident slots -- carries none of the job loop (its per-job values are loop invariants again:
// 4.7 % fewer scalar instructions per sweep, 1.7 % faster).
template <int P, int OCC, bool GOLD_GLOBAL, bool RR>
__global__ __launch_bounds__(RS_THREADS, OCC) void chain1d_rs_kernel(
    ChainRsArgs a, const cplx* __restrict__ E, cplx* __restrict__ blk, int* __restrict__ iters,
    int* __restrict__ converged)
{
    constexpr int T16 = (P - 1 + 15) / 16;              // 16-row tiles per dimension
    constexpr int KS = (P + 3) / 4 < 4 * T16 ? (P + 3) / 4 : 4 * T16;   // k-steps of a full-width product (n <= P)
    constexpr int WELEMS = 16 * T16 * P + 16;           // 16*T16 rows and a few elements of slack behind the last one
    constexpr int TR = T16 - 1;                         // the last tile
    constexpr bool REM = RS_REMAINDER && T16 >= 2 && P - 16 * TR <= 4;   // ... holds <= 4 rows / columns: strips
    constexpr int FT = REM ? TR : T16;                  // full 16 x 16 tiles per dimension
    constexpr bool M3G = T16 >= 3;                      // products in 3M form (mfma3): the 168-VGPR kernels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int flags[2 * RS_WAVES];                 // per wave: any(diff > conv), all(diff <= conv)
    __shared__ int pivrow[64], colof[64];
    __shared__ cplx rowline[RS_NB];                     // pivot row of the column step being factored
    __shared__ unsigned la_cnt[2];                      // look-ahead counters of rs_inverse
    unsigned la_epoch = 0;
    if (threadIdx.x == 0) { la_cnt[0] = 0u; la_cnt[1] = 0u; }

    // job of this launch slot: in launch order, or -- when the provider has seen this grid before -- in
    // the order of decreasing sweep counts of the previous evaluation (the jobs differ by up to 20x in
    // length; started longest first, the last workgroups of the grid do not leave the chip idle).
    // Round robin (a.rr_quantum > 0): the slot is a persistent workgroup and takes its jobs from the queue.
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr bool rr = RS_RR && RR;
    __shared__ long long rr_msg;
    cplx* Ws = reinterpret_cast<cplx*>(smem_raw);       // [16*T16][P] (+ slack): g (start of a sweep), T, M, the reduced M
    const int tid = threadIdx.x, lane = tid & 63;
    // ---- wave roles.  A v_mfma_f64_16x16x4 holds its SIMD's vector issue for ~45 of its 64 cycles whatever the
    // priorities (scripts/probe/fp64_coexec_probe.hip: beside a wave that streams them, another wave's v_fma_f64 take
    // 22 cycles instead of 5.6 and an LDS round trip 230 instead of 105), so the wave that factors the panels -- a
    // chain of dependent vector and LDS instructions, the workgroup's critical path -- must not share its SIMD with
    // waves that stream matrix instructions.  The four waves of a workgroup always land on four different SIMDs
    // (scripts/probe/wave_placement_probe.hip), so the roles follow the SIMD: the wave on SIMD 0 is the CHAIN wave
    // (role 3: factors every panel; its share of the products and updates is the last tile, for n_c = 50 the
    // 2-row / 2-column remainder strip), the waves on SIMDs 1-3 are the MATRIX waves (roles 0-2).  With several
    // workgroups per CU all chain waves then sit on SIMD 0 and all matrix-instruction streams on SIMDs 1-3.
    // Only speed depends on the placement: were two waves ever to report the same SIMD, the roles fall back to
    // the wave numbers.
    int wave = tid >> 6;
    bool chain_roles = false;
    if (a.simd_roles) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int simd = (hwid >> 4) & 3;
        if (lane == 0) flags[wave] = simd;
        __syncthreads();
        const int seen = (1 << flags[0]) | (1 << flags[1]) | (1 << flags[2]) | (1 << flags[3]);
        chain_roles = __builtin_amdgcn_readfirstlane(seen) == 15;
        if (chain_roles) wave = simd == 0 ? RS_WAVES - 1 : simd - 1;
        wave = __builtin_amdgcn_readfirstlane(wave);
        __syncthreads();
    }
    // ---- the job: set by next_job() below -- once, or (round robin) whenever this workgroup takes another one
    int job = 0, count = 0, c = 0, b = 0, n = 1, off = 0, ksteps = 1;
    bool resume = false;                                // a job that was set aside after `count` sweeps: its iterate waits in blk
    const int fi = lane & 15, fk = lane >> 4;
    // the old iterate, element (ks*4 + fk, wave*16 + fi) of g at slot ks of this lane
    // (global: [slot][thread], coalesced; LDS: a compact n x n matrix behind the work matrix)
    // Slot ks < gold_lds_slots: LDS, a compact matrix (pitch n) at element gold_lds_off of the dynamic LDS --
    // behind the work matrix when two workgroups share the CU; when three do, in the rows n_max+2 .. 16*T16-1
    // of the work matrix itself, which only feed discarded output rows of the padded tiles and may hold any
    // finite values.  The other slots: global scratch [slot][thread] (coalesced); with the LDS part taken
    // off, the scratch of the workgroups of an XCD fits its 4 MB L2 and is rewritten there sweep after sweep.
    cplx* gold_g = GOLD_GLOBAL ? a.gold + ((size_t)slot * KS) * RS_THREADS + tid : nullptr;
    cplx* gold_l = Ws;
    // LDS slots of the old iterate: LS_CT (what fits when n = P, a compile-time number) or one more
    constexpr int LS_FIT = (16 * T16 * P + 16 - (P + 2) * P) / (4 * P);
    constexpr int LS_CT = GOLD_GLOBAL ? (LS_FIT > 0 ? LS_FIT : 0) : KS;
    const int lds_slots = GOLD_GLOBAL ? a.gold_lds_slots : KS;

    const cplx *alpha = a.alpha, *Salpha = a.Salpha, *beta = a.beta, *Sbeta = a.Sbeta, *tau = a.tau, *Stau = a.Stau;
    cplx e = cmake(0.0, 0.0), z = e;
    const double conv2 = a.conv * a.conv, rf = a.relFactor, rf1 = 1.0 - a.relFactor;
    auto sel = [](bool ok, cplx v) { return cmake(ok ? v.x : 0.0, ok ? v.y : 0.0); };
    // A = (E + i eta) Sa - a at (i, j); global loads at clamped (always valid) indices
    auto Aat = [&](int i, int j) {
        const int o = min(i, n - 1) * n + min(j, n - 1);
        return csub(cmul(z, Salpha[o]), alpha[o]);
    };

    // ---- the stationary operand of the two products: row tile `wave` of  zz * Smat - mat  in the MFMA
    // A-operand layout, element (wave*16 + fi, ks*4 + fk).  It is NOT kept in registers across the sweep
    // (52-64 VGPRs that the three-workgroups-per-CU budget does not have): each product streams it from the
    // lead matrices, which every workgroup of the contact shares (L2 / L1 resident), PF k-steps ahead.
    // Rows >= n give garbage in output rows / columns >= n only (never stored); k >= n is zeroed.
#ifndef RS_PF
#define RS_PF 1
#endif
    constexpr int PF = RS_PF;                       // k-steps the streamed operand is requested ahead
    const cplx* opS = Sbeta; const cplx* opM = beta; cplx opz = z;     // (set per job and per pass)
    struct Stream { cplx s[PF], m[PF]; };
    auto stream_fetch = [&](Stream& q, const cplx* sS, const cplx* sM, int ks, int fk) __attribute__((always_inline)) {
        const int kc = min(ks * 4 + fk, n - 1);
        q.s[ks % PF] = sS[kc]; q.m[ks % PF] = sM[kc];
    };
    auto stream_elem = [&](const Stream& q, int ks, int fk) __attribute__((always_inline)) {
        return sel(ks * 4 + fk < n, csub(cmul(opz, q.s[ks % PF]), q.m[ks % PF]));
    };

    // wave w: acc[tj] = sum_k Op[w*16 + fi][k] * Ws[k][tj*16 + fi]  (row tile w of  Op Ws).  The padding of
    // Ws is zero / finite, output columns >= n are never stored.
    auto gemm_rowtile = [&](d4 (&accr)[T16], d4 (&acci)[T16], d4 (&accc)[T16]) __attribute__((always_inline)) {
        const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
        const bool strip_wave = REM && wave == TR;      // this wave's row tile is the row strip
        const cplx* bb = Ws + fk * P + fi;
        const cplx* bb4 = Ws + fk * P + (fi & 3);       // column-strip B operand: column TR*16 + (l&3) in every block
        const int irow = min(strip_wave ? TR * 16 + (fi & 3) : wave * 16 + fi, n - 1) * n;
        const cplx* sS = opS + irow; const cplx* sM = opM + irow;
        Stream q;
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) stream_fetch(q, sS, sM, ks, fk);
#pragma unroll
        for (int tj = 0; tj < T16; ++tj) { accr[tj] = (d4){0, 0, 0, 0}; acci[tj] = (d4){0, 0, 0, 0}; accc[tj] = (d4){0, 0, 0, 0}; }
        // the LDS operands (B fragments of the work matrix) of k-step ks+1 are requested in front of the matrix
        // instructions of k-step ks: an LDS round trip is ~300 cycles on a CU whose LDS three workgroups share
        cplx qbuf[2][T16];
        auto load_qb = [&](int ks, cplx (&qb)[T16]) __attribute__((always_inline)) {
#pragma unroll
            for (int tj = 0; tj < FT; ++tj) qb[tj] = bb[ks * 4 * P + tj * 16];
            if (REM) qb[TR] = strip_wave ? bb[ks * 4 * P + TR * 16] : bb4[ks * 4 * P + TR * 16];
        };
        load_qb(0, qbuf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < ksteps) {
                const cplx pa = stream_elem(q, ks, fk);
                const double ps = pa.x + pa.y;
                if (ks + PF < KS) stream_fetch(q, sS, sM, ks + PF, fk);
                cplx (&qb)[T16] = qbuf[ks & 1];
                if (ks + 1 < KS && ks + 1 < ksteps) load_qb(ks + 1, qbuf[(ks + 1) & 1]);
                if (strip_wave) {
#pragma unroll
                    for (int tj = 0; tj < T16; ++tj) { if (M3G) RS_M3S(accr[tj], acci[tj], accc[tj], pa.x, pa.y, ps, qb[tj].x, qb[tj].y, qb[tj].x + qb[tj].y); else RS_ZMFMA4(accr[tj], acci[tj], pa, qb[tj]); }
                } else {
#pragma unroll
                    for (int tj = 0; tj < FT; ++tj) { if (M3G) mfma3(accr[tj], acci[tj], accc[tj], pa.x, pa.y, ps, qb[tj].x, qb[tj].y, qb[tj].x + qb[tj].y); else zmfma(accr[tj], acci[tj], pa, qb[tj]); }
                    if (REM) { if (M3G) RS_M3S(accr[TR], acci[TR], accc[TR], pa.x, pa.y, ps, qb[TR].x, qb[TR].y, qb[TR].x + qb[TR].y); else RS_ZMFMA4(accr[TR], acci[TR], pa, qb[TR]); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // wave w: acc[ti] = sum_k Ws[ti*16 + fi][k] * conj(Op[w*16 + fi][k])  (column tile w of  Ws Op^H)
    auto gemm_coltile = [&](d4 (&accr)[T16], d4 (&acci)[T16], d4 (&accc)[T16]) __attribute__((always_inline)) {
        const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
        const bool strip_wave = REM && wave == TR;      // this wave's column tile is the column strip
        const cplx* ab = Ws + fi * P + fk;
        const cplx* ab4 = Ws + (fi & 3) * P + fk;       // row-strip A operand: row TR*16 + (l&3) in every block
        const int irow = min(strip_wave ? TR * 16 + (fi & 3) : wave * 16 + fi, n - 1) * n;
        const cplx* sS = opS + irow; const cplx* sM = opM + irow;
        Stream q;
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) stream_fetch(q, sS, sM, ks, fk);
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) { accr[ti] = (d4){0, 0, 0, 0}; acci[ti] = (d4){0, 0, 0, 0}; accc[ti] = (d4){0, 0, 0, 0}; }
        cplx pbuf[2][T16];                               // A fragments of the work matrix, double buffered (see gemm_rowtile)
        auto load_pa = [&](int ks, cplx (&pa)[T16]) __attribute__((always_inline)) {
#pragma unroll
            for (int ti = 0; ti < FT; ++ti) pa[ti] = ab[ti * 16 * P + ks * 4];
            if (REM) pa[TR] = ab4[TR * 16 * P + ks * 4];
        };
        load_pa(0, pbuf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < ksteps) {
                const cplx br = stream_elem(q, ks, fk);
                const double bd = br.x - br.y;
                if (ks + PF < KS) stream_fetch(q, sS, sM, ks + PF, fk);
                cplx (&pa)[T16] = pbuf[ks & 1];
                if (ks + 1 < KS && ks + 1 < ksteps) load_pa(ks + 1, pbuf[(ks + 1) & 1]);
                // pa * conj(b)
                if (strip_wave) {
#pragma unroll
                    for (int ti = 0; ti < T16; ++ti) { if (M3G) RS_M3S(accr[ti], acci[ti], accc[ti], pa[ti].x, pa[ti].y, pa[ti].x + pa[ti].y, br.x, br.y, bd); else RS_ZMFMA4C(accr[ti], acci[ti], pa[ti], br); }
                } else {
#pragma unroll
                    for (int ti = 0; ti < FT; ++ti) {
                        if (M3G) mfma3(accr[ti], acci[ti], accc[ti], pa[ti].x, pa[ti].y, pa[ti].x + pa[ti].y, br.x, br.y, bd);
                        else {
                            accr[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ti].x, br.x, accr[ti], 0, 0, 0);
                            accr[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ti].y, br.y, accr[ti], 0, 0, 0);
                            acci[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ti].y, br.x, acci[ti], 0, 0, 0);
                            acci[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[ti].x, br.y, acci[ti], 0, 0, 0);
                        }
                    }
                    if (REM) { if (M3G) RS_M3S(accr[TR], acci[TR], accc[TR], pa[TR].x, pa[TR].y, pa[TR].x + pa[TR].y, br.x, br.y, bd); else RS_ZMFMA4C(accr[TR], acci[TR], pa[TR], br); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Where a lane's accumulator values of tile (ti, tj) live: FULL and ROW-STRIP tiles use the 16 x 16 C layout
    // (rows ti*16 + fk + 4r, column tj*16 + fi; a row strip only fills r = 0), a COLUMN-STRIP tile (tj == TR of a
    // full row tile) holds one value per lane at (ti*16 + 4 (fi>>2) + fk, TR*16 + (fi&3)).
    // f(i, j, re, im) is called for every element this lane holds.
    auto for_tile = [&](int ti, int tj, const d4& va, const d4& vb, const d4& vc, bool cj, int fi, int fk, auto f) __attribute__((always_inline)) {
        d4 vr, vi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vr[r] = !M3G ? va[r] : cj ? va[r] + vb[r] : va[r] - vb[r];
            vi[r] = !M3G ? vb[r] : cj ? vc[r] - va[r] + vb[r] : vc[r] - va[r] - vb[r];
        }
        if (REM && tj == TR && ti != TR) {
            f(ti * 16 + 4 * (fi >> 2) + fk, TR * 16 + (fi & 3), vr[0], vi[0]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!(REM && ti == TR) || r == 0) f(ti * 16 + fk + 4 * r, tj * 16 + fi, vr[r], vi[r]);
        }
    };
    // store row tile `wave` held as accumulators (C layout: rows fk + 4r, column fi of tile tj)
    auto store_rowtile = [&](const d4 (&accr)[T16], const d4 (&acci)[T16], const d4 (&accc)[T16]) __attribute__((always_inline)) {
        const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
#pragma unroll
        for (int tj = 0; tj < T16; ++tj)
            for_tile(wave, tj, accr[tj], acci[tj], accc[tj], false, fi, fk,
                     [&](int i, int j, double re, double im) { if (i < n && j < n) Ws[i * P + j] = cmake(re, im); });
    };
    // g_new[k][col] = W[pivrow[k]][colof[col]] for this lane's elements; first: g = g_new, else the
    // reference's mixing and stopping test (surfG1D.py:276-284).  Leaves g in Ws and in gold.
    int over = 1, allok = 0;
    auto gather_mix = [&](bool first, unsigned long long* st) __attribute__((always_inline)) {
        const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
        const int col = wave * 16 + fi;
        const bool colok = col < n;
        const int colc = min(col, n - 1);               // every load below is unconditional, at a clamped (valid) address
        const int* pvr = pivrow + fk;
        cplx* gdst = Ws + fk * P + col;
        cplx* gl = rs_opaque(gold_l) - col + colc;
        cplx* gg = GOLD_GLOBAL ? rs_opaque(gold_g) : nullptr;
        // slot ks of the old iterate lives in LDS?  compile-time below / above LS_CT, wave-uniform at LS_CT
        auto in_lds = [&](int ks) { return !GOLD_GLOBAL || ks < LS_CT || (ks == LS_CT && lds_slots > LS_CT); };
        // element (ks*4 + fk, col) exists?  with a remainder strip all k-steps below 4 TR hold rows < 16 TR < n
        auto valid = [&](int ks) { return colok && ((REM && ks * 4 + 3 < 16 * TR) || ks * 4 + fk < n); };
        // All requests of the phase go out in three batches -- pivot indices, old iterate (global scratch: the
        // longest latency, requested before the gathers), gathered new iterate -- instead of slot by slot: the
        // phase is three dependent round trips long, not thirteen.
        int pr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pr[ks] = ks < ksteps ? pvr[ks * 4] : 0;
        const int cfc = colof[colc];
        const cplx* gsrc = Ws + cfc;
        cplx go[KS], gm[KS];
        if (!first) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (ks < ksteps) go[ks] = in_lds(ks) ? gl[ks * 4 * n] : gg[ks * RS_THREADS];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gm[ks] = ks < ksteps ? gsrc[pr[ks] * P] : cmake(0.0, 0.0);
        if (RS_PHAT) {
            // the work matrix holds  (reduced M) - E : element (pivrow[c], c) is one short for every column c.  g_new[k][col]
            // = W[pivrow[k]][colof[col]] is such an element exactly when colof[col] == k
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) gm[ks].x += (cfc == ks * 4 + fk) ? 1.0 : 0.0;
        }
        if (st && tid == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st[5] = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[6] = __builtin_amdgcn_s_memrealtime(); }
        bool lane_over = false, lane_ok = true;
        if (!first) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < ksteps) {
                    const cplx gn = gm[ks];
                    const double dx = gn.x - go[ks].x, dy = gn.y - go[ks].y;
                    const double num2 = dx * dx + dy * dy;
                    const double den2 = fmax(gn.x * gn.x + gn.y * gn.y, 1e-24);
                    const bool v = valid(ks);
                    lane_over |= v && num2 > conv2 * den2;
                    lane_ok &= !v || num2 <= conv2 * den2;
                    gm[ks] = cmake(gn.x * rf + go[ks].x * rf1, gn.y * rf + go[ks].y * rf1);
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (ks < ksteps && valid(ks)) { if (in_lds(ks)) gl[ks * 4 * n] = gm[ks]; else gg[ks * RS_THREADS] = gm[ks]; }
        if (!first) {
            const bool w_over = __ballot(lane_over) != 0ull;
            const bool w_ok = __ballot(!lane_ok) == 0ull;
            if (lane == 0) { flags[wave] = w_over ? 1 : 0; flags[RS_WAVES + wave] = w_ok ? 1 : 0; }
        }
        if (st && tid == 0) st[7] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();                                // all gathers done: Ws may be overwritten
        if (st && tid == 0) st[40] = __builtin_amdgcn_s_memrealtime();
        if (tid < 64) colof[tid] = -1;                  // for the next inverse (every lane has read its colof)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (ks < ksteps && valid(ks)) gdst[ks * 4 * P] = gm[ks];
        if (!first) {
            over = flags[0] | flags[1] | flags[2] | flags[3];
            allok = flags[4] & flags[5] & flags[6] & flags[7];
        }
        __syncthreads();
    };

    // ---- a job starts: g0 = inv(A) is the first pass through the inverse of a fresh job, so the work matrix starts as A
    // (its padding zero; it is never written afterwards).  A cache hit starts from the stored iterate instead and goes
    // straight to Sigma = t g t^H; a job that was set aside continues from its iterate with the products of its next sweep.
    const bool gc_hit = a.gc_mode == 2;
    bool first = true, final_pass = false, skip = false;
    int q_end = 0;
    auto next_job = [&]() __attribute__((always_inline)) -> bool {
        if (rr) {
            if (tid == 0) rr_msg = rs_rr_pop(a.rr_q, a.order);
            __syncthreads();                            // (also: every wave is done with the LDS of the previous job)
            const long long ent = rr_msg;
            __syncthreads();
            if (ent < 0) return false;
            job = (int)(ent & 0xffffffffll); count = (int)(ent >> 32);
        } else {
            job = a.order ? a.order[slot] : slot; count = 0;
        }
        resume = count > 0;
        c = job % a.n_contacts; b = job / a.n_contacts;
        n = a.nc[c]; off = a.blk_off[c]; ksteps = (n + 3) >> 2;
        gold_l = Ws + a.gold_lds_off + fk * n + wave * 16 + fi;
        alpha = a.alpha + off; Salpha = a.Salpha + off; beta = a.beta + off; Sbeta = a.Sbeta + off;
        tau = a.tau + off; Stau = a.Stau + off;
        e = E[b]; z = cmake(e.x, e.y + a.eta);
        opS = Sbeta; opM = beta; opz = z;
        over = 1; allok = 0;
        first = !resume; skip = resume; final_pass = false;
        q_end = count + a.rr_quantum;
        if (gc_hit || resume) {
            if (resume) __threadfence();                // (acquire side of the queue entry lane 0 popped: the iterate below was
                                                        //  written by another workgroup of this launch)
            const cplx* gcj = (gc_hit ? a.gcache : blk) + (size_t)b * a.blk_stride + off;
            for (int t = tid; t < WELEMS; t += RS_THREADS) {
                const int i = t / P, j = t - i * P;
                Ws[t] = sel(i < n && j < n, gcj[min(i, n - 1) * n + min(j, n - 1)]);
            }
        } else {
            for (int t = tid; t < WELEMS; t += RS_THREADS) {
                const int i = t / P, j = t - i * P;
                Ws[t] = sel(i < n && j < n, Aat(i, j));
            }
        }
        if (resume) {
            // the old iterate of the mixing step is the iterate itself (gather_mix leaves the mixed g in both places)
            __syncthreads();
            const int fi = rs_opaque(lane & 15), fk = rs_opaque(lane >> 4);
            const int col = wave * 16 + fi;
            cplx* gl = rs_opaque(gold_l);
            cplx* gg = GOLD_GLOBAL ? rs_opaque(gold_g) : nullptr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < ksteps && col < n && ks * 4 + fk < n) {
                    const cplx v = Ws[(ks * 4 + fk) * P + col];
                    if (!GOLD_GLOBAL || ks < LS_CT || (ks == LS_CT && lds_slots > LS_CT)) gl[ks * 4 * n] = v; else gg[ks * RS_THREADS] = v;
                }
            }
        }
        if (tid < 64) { colof[tid] = -1; pivrow[tid] = 0; }
        __syncthreads();
        return true;
    };

    // One copy of every phase in the instruction stream (the loop body has to stay inside the 64 KB
    // instruction cache two CUs share): the start g0 = inv(A) is the first pass through the inverse, and
    // Sigma = t g t^H (t = E Stau - tau, no eta) runs as a last pass through the two products with t in
    // place of B:  X = t g (row tiles) -> Ws,  Sigma = X t^H (column tiles) -> global.
    bool need_job = true;
    while (true) {
        if (__builtin_expect(need_job, 0)) {
            if (!next_job()) break;                     // (round robin: the queue is empty -- every job is finished or held by a running workgroup)
            need_job = false;
        }
        unsigned long long* st = (RS_STAMPS && a.stamps && job == 0 && count == a.stamp_sweep) ? a.stamps : nullptr;
        if (st && tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        if (!gc_hit && !skip) {
            rs_inverse<T16, P, REM ? TR : -1>(n, Ws, pivrow, colof, rowline, tid, wave, chain_roles, la_cnt, la_epoch, st ? st + 8 : nullptr);   // st + 8: stage stamps, st + 24: factor
            if (st && tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
            if (!(RS_ABLATE & 8) || first) gather_mix(first, st);
            else { if (tid < 64) colof[tid] = -1; __syncthreads(); }
            if (st && tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
            if (!first) ++count;
            first = false;
        }
        if (__builtin_expect(skip, 0)) {
            skip = false;
        } else if (gc_hit || (a.force_iters >= 0 ? count >= a.force_iters : !(over && count < a.max_iter))) {
            final_pass = true;
            opS = Stau; opM = tau; opz = e;
            if (a.gc_mode == 1) {                       // a miss leaves its final iterate in the cache (g sits in Ws, complete
                cplx* gcj = a.gcache + (size_t)b * a.blk_stride + off;                              //  since the barrier that ends gather_mix)
                for (int t = tid; t < n * n; t += RS_THREADS) { const int i = t / n; gcj[t] = Ws[i * P + (t - i * n)]; }
            }
        } else if (__builtin_expect(rr && count >= q_end, 0)) {
            // the quantum is over.  Nobody waiting: carry on.  Otherwise the iterate goes to the job's output block (g sits in
            // Ws, complete since the barrier that ends gather_mix), the job to the back of the queue, and this workgroup
            // takes the job at the front
            if (tid == 0) rr_msg = rs_rr_waiting(a.rr_q) ? 1 : 0;
            __syncthreads();
            const bool sw = rr_msg != 0;
            __syncthreads();
            if (sw) {
                cplx* gsj = blk + (size_t)b * a.blk_stride + off;
                for (int t = tid; t < n * n; t += RS_THREADS) { const int i = t / n; gsj[t] = Ws[i * P + (t - i * n)]; }
                __threadfence();                        // release side: the iterate is visible before the queue entry is
                __syncthreads();
                if (tid == 0) rs_rr_push(a.rr_q, job, count);
                need_job = true;
                continue;
            }
            q_end = count + a.rr_quantum;
        }
        d4 mr[T16], mi[T16], mc[T16];
        if (RS_ABLATE & 4) {
#pragma unroll
            for (int t = 0; t < T16; ++t) { mr[t] = (d4){1e-3, 0, 0, 0}; mi[t] = mr[t]; mc[t] = mr[t]; }
        }
        // T = B g : row tile `wave`; g is read from Ws by every wave, so T waits in registers
        if (!(RS_ABLATE & 4)) gemm_rowtile(mr, mi, mc);
        __syncthreads();
        store_rowtile(mr, mi, mc);
        __syncthreads();
        if (st && tid == 0) st[3] = __builtin_amdgcn_s_memrealtime();
        // T B^H : column tile `wave`
        if (!(RS_ABLATE & 4)) gemm_coltile(mr, mi, mc);
        const int fis = rs_opaque(lane & 15), fks = rs_opaque(lane >> 4);
        if (final_pass) {
            cplx* out = blk + (size_t)b * a.blk_stride + off;
#pragma unroll
            for (int ti = 0; ti < T16; ++ti)
                for_tile(ti, wave, mr[ti], mi[ti], mc[ti], true, fis, fks,
                         [&](int i, int j, double re, double im) { if (i < n && j < n) out[i * n + j] = cmake(re, im); });
            if (tid == 0 && !gc_hit) {                  // (a hit's counts and flags are copied from the cache by the launcher's caller)
                if (iters) iters[(size_t)b * a.n_contacts + c] = count;
                if (converged) converged[(size_t)b * a.n_contacts + c] = (count > 0 && allok) ? 1 : 0;
            }
            if (!rr) break;
            need_job = true;
            continue;
        }
        __syncthreads();
        // M = A - T B^H
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) {
            for_tile(ti, wave, mr[ti], mi[ti], mc[ti], true, fis, fks, [&](int i, int j, double re, double im) {
                const cplx av = Aat(i, j);
                if (i < n && j < n) Ws[i * P + j] = cmake(av.x - re, av.y - im);
            });
            __builtin_amdgcn_sched_barrier(0);          // the A elements of one tile in flight, not of all
        }
        __syncthreads();
        if (st && tid == 0) st[4] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

bool chain1d_lds_supported(int nc_max) { return nc_max <= 64; }

namespace {
// sweeps a job runs before it makes room for a waiting one (NEGF_CHAIN_RR, 0 = every job runs to its end in one go)
int rs_rr_quantum(int user)
{
    if (user >= 0) return user;                      // negf_set_chain_round_robin
    static int q = -1;
    if (q < 0) { const char* e = getenv("NEGF_CHAIN_RR"); q = e ? std::max(atoi(e), 0) : 100; }
    return q;
}
size_t rs_gold_elems(int nc_max, int n_contacts, int nb)
{
    return (size_t)4 * ((nc_max + 15) >> 4) * RS_THREADS * n_contacts * nb;     // >= KS slots per lane
}
// entries of the job queue's ring: a job is queued at most once per quantum (0: no round robin for this launch)
size_t rs_ring_entries(int n_contacts, int nb, int max_sweeps, int rr_user)
{
    const int q = rs_rr_quantum(rr_user);
    if (q <= 0 || max_sweeps <= q) return 0;
    const size_t jobs = (size_t)n_contacts * nb;
    const size_t cap = jobs * ((size_t)max_sweeps / q + 1) + 2048;
    return (cap >> 27) ? 0 : cap;
}
}  // namespace

// lane-private copies of the iterate (one set per launch slot) + the job queue of the round-robin launch
size_t chain1d_lds_scratch_elems(int nc_max, int n_contacts, int nb, int max_sweeps, int rr_quantum)
{
    return rs_gold_elems(nc_max, n_contacts, nb) + 1 + (rs_ring_entries(n_contacts, nb, max_sweeps, rr_quantum) + 1) / 2;
}

namespace {

template <int P>
void chain1d_rs_launch(hipStream_t st, ChainRsArgs a, int n_max, int n_contacts, int nb, const cplx* E, cplx* blk,
                       int* iters, int* conv, cplx* gold_scratch, int occ_env, int rr_slots, unsigned rr_cap, bool rr_forced)
{
    constexpr int T16 = (P - 1 + 15) / 16;
    const size_t wmat = (size_t)(16 * T16 * P + 16) * sizeof(cplx);
    const size_t gold_lds = (size_t)n_max * n_max * sizeof(cplx);
    // occupancy from the LDS share of a workgroup (+ < 1 KB of static LDS): the work matrix alone when the
    // old iterate goes to global scratch, both matrices otherwise
    // (allocated in granules of 1280 bytes: 53.8 KB per workgroup is the most that still fits three times)
    auto fits = [](size_t smem, int per_cu) { return ((smem + 832 + 1279) / 1280) * 1280 * per_cu <= 160 * 1024; };
    // old iterate behind the work matrix (all slots in LDS) ...
    ChainRsArgs al = a; al.gold_lds_off = 16 * T16 * P + 16; al.gold_lds_slots = 1 << 20;
    // ... or its first slots in the unused rows n_max+2 .. of the work matrix, the rest in global scratch
    ChainRsArgs ag = a; ag.gold_lds_off = (n_max + 2) * P;
    constexpr int LS_FIT = (16 * T16 * P + 16 - (P + 2) * P) / (4 * P);      // the kernel's LS_CT
    ag.gold_lds_slots = std::min(std::max(0, (16 * T16 * P + 16 - ag.gold_lds_off) / (4 * n_max)), std::max(LS_FIT, 0) + 1);
    if (ag.gold_lds_slots < std::max(LS_FIT, 0)) ag.gold_lds_slots = 0;       // (cannot happen: n_max <= P)
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0; hipDeviceProp_t prop;
        n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    auto launch = [&](auto kern, auto kern_rr, size_t smem, int occ) {
        ChainRsArgs aa = smem > wmat ? al : ag;
        dim3 grid(n_contacts, nb);
        // Round robin pays when jobs have to wait for a slot AND nothing is known about their lengths (the first
        // evaluation of a grid by this provider: 725 against 800 ms in launch order on the C3 grid): one persistent
        // workgroup per resident slot then.  With a predicted order the plain launch, longest job first, is the faster
        // one (718 against 726 ms: its kernel carries no job loop), so the round robin is left to the launches without
        // an order or with one predicted from too few points (order_trusted) -- unless the caller set the quantum
        // himself (negf_set_chain_round_robin: tests, A/B).
        const int jobs = n_contacts * nb, slots = rr_slots > 0 ? std::min(rr_slots, occ * n_cus) : occ * n_cus;
        const bool rr = RS_RR && aa.rr_quantum > 0 && jobs > slots && (aa.order == nullptr || rr_forced);
        if (rr) {
            hipLaunchKernelGGL(rs_rr_init_kernel, dim3((rr_cap + 255) / 256), dim3(256), 0, st, aa.rr_q, rr_cap, (unsigned)jobs);
            grid = dim3(slots, 1);
        } else aa.rr_quantum = 0;
        static int log_env = -1;
        if (log_env < 0) log_env = getenv("NEGF_CHAIN_LOG") ? 1 : 0;
        if (log_env) fprintf(stderr, "[chain launch] jobs %d slots %d quantum %d order %d gc_mode %d\n", jobs, slots, aa.rr_quantum, aa.order ? 1 : 0, aa.gc_mode);
        auto go = [&](auto k) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
                (void)hipGetLastError();
            hipLaunchKernelGGL(k, grid, dim3(RS_THREADS), smem, st, aa, E, blk, iters, conv);
        };
        if (rr) go(kern_rr); else go(kern);
    };
    constexpr int OCC_MAX = T16 <= 2 ? 4 : 3;          // register budget: 128 VGPRs (T16 <= 2), 168 above
    if (occ_env != 2 && fits(wmat + gold_lds, OCC_MAX)) launch(chain1d_rs_kernel<P, OCC_MAX, false, false>, chain1d_rs_kernel<P, OCC_MAX, false, true>, wmat + gold_lds, OCC_MAX);
    else if (occ_env != 2 && gold_scratch && fits(wmat, OCC_MAX)) launch(chain1d_rs_kernel<P, OCC_MAX, true, false>, chain1d_rs_kernel<P, OCC_MAX, true, true>, wmat, OCC_MAX);
    else if (fits(wmat + gold_lds, 2) || !gold_scratch) launch(chain1d_rs_kernel<P, 2, false, false>, chain1d_rs_kernel<P, 2, false, true>, wmat + gold_lds, 2);
    else launch(chain1d_rs_kernel<P, 2, true, false>, chain1d_rs_kernel<P, 2, true, true>, wmat, 2);
}

}  // namespace

void launch_chain1d_lds(hipStream_t st, const SigmaProvider& p, const int* d_nc, const int* d_blk_off, int nb,
                        const cplx* E, cplx* blk, int* iters, int* conv, cplx* gold_scratch, const int* order,
                        cplx* gcache, int gc_mode, int rr_quantum, int rr_slots, bool order_trusted)
{
    ChainRsArgs a;
    a.order = order;
    a.gcache = gcache; a.gc_mode = gcache ? gc_mode : 0;
    a.alpha = p.d_alpha; a.Salpha = p.d_Salpha; a.beta = p.d_beta; a.Sbeta = p.d_Sbeta;
    a.tau = p.d_tau; a.Stau = p.d_Stau;
    a.nc = d_nc; a.blk_off = d_blk_off;
    a.n_contacts = p.n_contacts; a.blk_stride = p.blk_stride;
    a.eta = p.eta; a.conv = p.conv; a.relFactor = p.relFactor;
    a.max_iter = p.max_iter; a.force_iters = p.force_iters;
    a.gold = gold_scratch;
    // the job queue sits behind the iterate copies (chain1d_lds_scratch_elems); a cache hit runs no sweeps
    a.rr_quantum = 0; a.rr_q = nullptr;
    unsigned rr_cap = 0;
    {
        const size_t cap = rs_ring_entries(p.n_contacts, nb, std::max(p.max_iter, p.force_iters), rr_quantum);
        if (gold_scratch && cap && a.gc_mode != 2) {
            a.rr_q = reinterpret_cast<RsQueue*>(gold_scratch + rs_gold_elems(p.nc_max, p.n_contacts, nb));
            rr_cap = (unsigned)cap; a.rr_quantum = rs_rr_quantum(rr_quantum);
        }
    }
    static unsigned long long* d_stamps = nullptr;
    static int want_stamps = -1;
    if (want_stamps < 0) {
        want_stamps = getenv("NEGF_CHAIN_STAMPS") ? 1 : 0;
        if (want_stamps) { (void)hipMalloc(&d_stamps, 64 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 64 * sizeof(unsigned long long)); }
    }
    a.stamps = d_stamps;
    { const char* e = getenv("NEGF_CHAIN_STAMP_SWEEP"); a.stamp_sweep = e ? atoi(e) : 10; }
    static int occ_env = -1;
    if (occ_env < 0) { const char* e = getenv("NEGF_CHAIN1D_OCC"); occ_env = e ? atoi(e) : 0; }
    static int roles_env = -1;
    if (roles_env < 0) { const char* e = getenv("NEGF_CHAIN1D_ROLES"); roles_env = e ? atoi(e) : 1; }
    a.simd_roles = roles_env;
    // one instantiation per pitch class of the largest contact (smaller contacts of the same launch run
    // in the same padded matrix): the smallest odd pitch of the list that holds n_max columns.  The remainder-strip
    // classes (19, 35, 51: the last tile holds <= 4 rows / columns and runs on the 4x4x4 instruction) store their
    // full tiles unguarded and count every k-step below the strip as inside the matrix, i.e. they assume that EVERY
    // contact of the launch reaches into the strip, n > 16 TR.  A launch whose smallest contact does not (contacts of
    // unequal size, e.g. n_c = (50, 40)) takes the next class without strips, whose stores and k-steps are guarded
    // by the contact's own n.
    const int n_max = p.nc_max;
    int n = n_max;                                  // selects the class
    {
        int n_min = n;
        for (int k : p.nc) n_min = std::min(n_min, k);
        const int strip_base = n <= 16 ? -1 : n <= 19 ? 16 : (n > 32 && n <= 35) ? 32 : (n > 48 && n <= 51) ? 48 : -1;
        if (strip_base > 0 && n_min <= strip_base) n = strip_base == 16 ? 25 : strip_base == 32 ? 41 : 57;
    }
#define RS_CASE(PP) chain1d_rs_launch<PP>(st, a, n_max, p.n_contacts, nb, E, blk, iters, conv, gold_scratch, occ_env, rr_slots, rr_cap, rr_quantum > 0 || !order_trusted)
#ifdef RS_FAST_BUILD
    if (n <= 16) RS_CASE(17); else RS_CASE(51);
#else
    if (n <= 16) RS_CASE(17);
    else if (n <= 19) RS_CASE(19);                // 16 + a remainder strip
    else if (n <= 25) RS_CASE(25);
    else if (n <= 32) RS_CASE(33);
    else if (n <= 35) RS_CASE(35);                // 32 + a remainder strip
    else if (n <= 41) RS_CASE(41);
    else if (n <= 48) RS_CASE(49);
    else if (n <= 51) RS_CASE(51);
    else if (n <= 57) RS_CASE(57);
    else RS_CASE(65);
#endif
#undef RS_CASE
    if (d_stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[64];
        (void)hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        if (h[0]) {
            auto us = [&](int i) { return h[i] ? (double)(h[i] - h[0]) / 100.0 : -1.0; };
            fprintf(stderr, "[chain stamps] sweep %d (us):", a.stamp_sweep); fprintf(stderr, " inverse %.2f  diff+mix %.2f  T=Bg %.2f  M=A-TB^H %.2f  (total %.2f) | inverse stages:",
                    us(1), us(2) - us(1), us(3) - us(2), us(4) - us(3), us(4));
            for (int i = 8; i < 16 && h[i]; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[0]) / 100.0);
            fprintf(stderr, " | mix: lds %.2f vmem %.2f computed+stored %.2f barrier %.2f", us(5) - us(1), us(6) - us(1), us(7) - us(1), us(40) - us(1));
            fprintf(stderr, " | column step 4 of panel 1 (shader cycles): search %llu  pivot-row exchange %llu  reciprocal+coef %llu  64 FMAs %llu",
                    h[8 + 41] - h[8 + 40], h[8 + 42] - h[8 + 41], h[8 + 43] - h[8 + 42], h[8 + 44] - h[8 + 43]);
            fprintf(stderr, " | factor begin-end:");
            for (int i = 24; i < 38 && h[i]; i += 2) fprintf(stderr, " %.2f-%.2f", (double)(h[i] - h[0]) / 100.0, (double)(h[i + 1] - h[0]) / 100.0);
            fprintf(stderr, "\n");
        }
    }
}
