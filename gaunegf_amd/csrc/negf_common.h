// Internal declarations shared by the translation units of libnegf_hip.so.
// gfx950 (MI355X) only; wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include "../../include/negf.h"

// ------------------------------------------------------------------ complex
// complex128 as an aligned pair: one 16-byte global/LDS access per element.
struct __attribute__((aligned(16))) cplx {
    double x, y;
};
static_assert(sizeof(cplx) == 16, "cplx must be 16 bytes");

__host__ __device__ __forceinline__ cplx cmake(double a, double b) { cplx r; r.x = a; r.y = b; return r; }
__host__ __device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__host__ __device__ __forceinline__ cplx cneg(cplx a) { return cmake(-a.x, -a.y); }
__host__ __device__ __forceinline__ cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__host__ __device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a - b*c
__host__ __device__ __forceinline__ cplx cfnma(cplx a, cplx b, cplx c) {
    return cmake(a.x - b.x * c.x + b.y * c.y, a.y - b.x * c.y - b.y * c.x);
}
// a + b*c
__host__ __device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) {
    return cmake(a.x + b.x * c.x - b.y * c.y, a.y + b.x * c.y + b.y * c.x);
}
__host__ __device__ __forceinline__ cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
// LAPACK izamax metric |re|+|im| (what zgetrf pivots on)
__host__ __device__ __forceinline__ double cabs1(cplx a) { return fabs(a.x) + fabs(a.y); }
__host__ __device__ __forceinline__ double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }
// 1/a with scaling against overflow/underflow (Smith)
__host__ __device__ __forceinline__ cplx crecip(cplx a) {
    if (fabs(a.x) >= fabs(a.y)) {
        double r = a.y / a.x;
        double d = a.x + a.y * r;
        return cmake(1.0 / d, -r / d);
    } else {
        double r = a.x / a.y;
        double d = a.x * r + a.y;
        return cmake(r / d, -1.0 / d);
    }
}

// -------------------------------------------------------------- error macros
#define NEGF_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            fprintf(stderr, "[negf] HIP error %s at %s:%d: %s\n", #expr, __FILE__,    \
                    __LINE__, hipGetErrorString(_e));                                 \
            return NEGF_EHIP;                                                         \
        }                                                                             \
    } while (0)

// ------------------------------------------------------------------- context
enum SigmaKind { SK_CONST = 0, SK_CHAIN1D = 2, SK_BETHE = 3, SK_PRECOMPUTED = 4 };

struct SigmaProvider {
    int kind = -1;
    int n_contacts = 0;
    // CONST: dense per-contact Sigma [n_contacts][n][n] and their sum, on device
    cplx* d_const_c = nullptr;     // [n_contacts][n*n]
    cplx* d_const_tot = nullptr;   // [n*n]
    cplx* d_hbase = nullptr;       // F + Sigma_tot (CONST only): A = E S - hbase
    // block providers (CHAIN1D / BETHE): contact blocks scattered at inds x inds
    std::vector<int> nc;           // block size per contact
    std::vector<int> blk_off;      // offset (in cplx) of contact c inside one energy's block record
    int blk_stride = 0;            // cplx per energy = sum nc^2
    int nc_max = 0;
    int* d_inds = nullptr;         // concatenated orbital indices
    int *d_nc = nullptr, *d_blk_off = nullptr, *d_inds_off = nullptr;   // device copies
    int *d_n_atoms = nullptr, *d_atom_off = nullptr;
    int* d_pos = nullptr;          // [n_contacts][n]: position of orbital i in contact c's index list, -1 outside (small fused path)
    std::vector<int> inds_off;     // offset of contact c in d_inds
    std::vector<int> h_inds;
    // CHAIN1D matrices, concatenated [sum nc^2] each
    cplx *d_alpha = nullptr, *d_Salpha = nullptr, *d_beta = nullptr, *d_Sbeta = nullptr,
         *d_tau = nullptr, *d_Stau = nullptr;
    // host copy of alpha | Salpha | beta | Sbeta and its 64-bit hash: what the surface Green's function depends on
    // besides eta / conv / relFactor / max_iter -- the key of the context's g(E) cache (ChainGEntry)
    std::shared_ptr<std::vector<cplx>> h_lead;     // shared with the cache entries this provider fills
    unsigned long long lead_hash = 0;
    double eta = 0, conv = 0, relFactor = 0, mix = 0;
    int max_iter = 0, force_iters = -1;
    // job order of the next chain launch (predicted on the device)
    int* d_order = nullptr;
    int order_cap = 0;
    // energies and sweep counts of the previous evaluation: the order of a NEW grid is predicted from them.
    // An evaluation may arrive in several batch chunks (m0 = 0, nb, 2 nb ...): the chunks are appended to
    // cur*, and the chunk with m0 = 0 of the NEXT evaluation promotes cur* to prev*.
    cplx *d_prevE = nullptr, *d_curE = nullptr;
    int *d_prev_iters = nullptr, *d_cur_iters = nullptr;
    int prev_n = 0, prev_cap = 0, cur_n = 0, cur_cap = 0;
    // BETHE
    std::vector<int> n_atoms;      // atoms per contact
    int* d_atom_orbs = nullptr;    // [total_atoms][9]
    int* d_nb_off = nullptr;       // [total_atoms+1]
    int* d_nb_dirs = nullptr;
    std::vector<int> atom_off;     // first atom of contact c
    double* d_H = nullptr;         // [n_contacts][81]
    double* d_Slist = nullptr;     // [n_contacts][12][81]
    double* d_Vlist = nullptr;
    cplx* d_xi = nullptr;          // [n*n] or null
    // compact coupling matrices: Gamma_c lives on the index list inds[inds_off[c] .. +nc[c]) only
    // (block providers without Xi; CONST providers whose matrices vanish outside a small support)
    bool compact_ok = false;
    cplx* d_const_blk = nullptr;   // CONST: Sigma_c restricted to its support, [blk_stride]
    // PRECOMPUTED
    int m_pre = 0;
    int pre_nc = 0;
    bool pre_is_gamma = false;     // d_pre_c holds the Gamma matrices themselves
    cplx* d_pre_tot = nullptr;     // [m][n*n]
    cplx* d_pre_c = nullptr;       // [m][pre_nc][n*n] or null
};

// One cached evaluation of the 1-D chain fixed point: the final iterates g(E_m) of every (energy, contact) of one
// launch, with the sweep counts and convergence flags the launch reported.  g depends on the lead cell (alpha, Salpha,
// beta, Sbeta), eta, conv, relFactor, max_iter (and force_iters) and E -- NOT on F or on the coupling blocks tau
// (surfG1D.py:256-262; setF refreshes tau only, :319-329) -- so an entry is keyed on exactly those, bitwise, and
// outlives the provider that filled it: a provider re-created after setF, the t = I variant behind surfG.g(), the two
// Sigma evaluations of GrLessInt, calculate_transmission + calculate_dos on one grid and SCF cycles at a fixed Fermi
// level all find it.  A hit runs only Sigma = t g t^H (the last pass of the chain kernel) and is bit-identical to a miss.
struct ChainGEntry {
    std::vector<int> nc;
    std::shared_ptr<std::vector<cplx>> lead;
    unsigned long long lead_hash = 0, E_hash = 0;
    double eta = 0, conv = 0, relFactor = 0;
    int max_iter = 0, force_iters = -1;
    std::vector<cplx> E;
    cplx* d_g = nullptr;    size_t g_cap = 0;      // [energies][blk_stride]
    int* d_it = nullptr;    int* d_cv = nullptr;   size_t it_cap = 0;   // [energies][n_contacts]
    unsigned long long used = 0;
    bool valid = false;
};

// side streams of the windowed inverse (small batches are cut into up to four stream groups, k_inverse_blocked.hip):
// owned by the context
struct GjSideStreams {
    static constexpr int MAXG = 4;
    hipStream_t s[MAXG - 1] = {};
    hipEvent_t fork = nullptr, join[MAXG - 1] = {};
    bool ok = false;
};

struct ProfEntry { double ms = 0; int launches = 0; double flops_alg = 0, flops_mfma = 0; };

// Flop accounting of the dense kernels, per kernel family (negf_profile_read_flops): the launchers add, for every
// launch, the ALGORITHMIC flops (8 per complex multiply-add: 8 M N K per product -- also for a Hermitian product,
// whose mirrored half the reference computes --, 8 n^3 per inverse) and the flops actually ISSUED to the
// matrix cores (3 real products per tile and k-step in the 3M form, 16-granular tiles, K padded to the staged
// K-tile, only the block tiles on and above the diagonal of a Hermitian product).  Per process, like the context
// not thread-safe; ProfScope attributes the difference across its lifetime to its family.
struct FlopCount { double alg = 0, mfma = 0; };
extern FlopCount g_negf_flops;
inline void negf_count_flops(double alg, double mfma) { g_negf_flops.alg += alg; g_negf_flops.mfma += mfma; }

struct negf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n = 0;
    cplx* d_F = nullptr;           // the resident system: aliases of sys[sys_cur]
    cplx* d_S = nullptr;
    // the last NEGF_SYS_SLOTS systems handed to negf_set_system stay on the device (with a host copy to recognise
    // them by): front-ends that alternate between two systems -- the two spin blocks of a blockdiag(alpha, beta)
    // Fock matrix, scf.py:177-180 -- re-select instead of re-uploading
    struct SysSlot { cplx* dF = nullptr; cplx* dS = nullptr; std::vector<cplx> hF, hS; unsigned long long used = 0, key = 0; bool valid = false; };
    static constexpr int NEGF_SYS_SLOTS = 2;
    SysSlot sys[NEGF_SYS_SLOTS];
    int sys_cur = -1;
    unsigned long long sys_clock = 0;
    int batch_user = 0;            // 0 = auto
    int batch = 0;                 // allocated workspace batch
    bool transmission_seen = false; // negf_transmission[_dev] has run on this context: size the workspace for it
    // workspace, sized for `batch` energies
    cplx* d_A = nullptr;           // [batch][n*n]   assembled matrix -> inverse in place
    cplx* d_T1 = nullptr;          // [batch][n*n]   temp products
    cplx* d_T2 = nullptr;          // [batch][n*n]
    cplx* G = nullptr;             // where the last inverse left its result (d_A or d_T1)
    cplx* W1 = nullptr;            // the other of (d_A, d_T1): free work area after the inverse
    cplx* W2 = nullptr;            // = d_T2
    cplx* d_blk = nullptr;         // [batch][blk_stride] contact blocks of Sigma(E)
    int blk_cap = 0;
    cplx* d_scratch = nullptr;     // per-workgroup scratch of the Sigma kernels
    size_t scratch_cap = 0;
    int* d_ipiv = nullptr;         // [batch][2][n] pivot bookkeeping of the large-matrix inverse
    int* d_info = nullptr;         // [m_cap]
    int* d_iters = nullptr;        // [m_cap][contacts]
    int* d_conv = nullptr;
    int m_cap = 0;
    int contacts_cap = 0;
    cplx* d_E = nullptr;           // staging for host-pointer API
    cplx* d_w = nullptr;
    std::vector<cplx> h_E;         // host copy of what stage_grid put into d_E (valid while h_E_valid): the g(E) cache
    bool h_E_valid = false;        //   keys on the energies without a device round trip
    // g(E) cache of the 1-D chain providers (negf_set_chain_cache)
    std::vector<ChainGEntry> gcache;
    int gcache_max = 512;          // entries (evaluated grids) kept, 0 = off
    size_t gcache_entry_bytes_max = (size_t)4 << 30;
    size_t gcache_bytes_max = (size_t)8 << 30;     // all entries together
    unsigned long long gcache_clock = 0, gcache_hits = 0, gcache_misses = 0;
    cplx* d_acc = nullptr;         // [n*n] result staging
    double* d_scal = nullptr;      // [m_cap][8] scalar outputs
    double* d_site = nullptr;      // [batch][n] per-site DOS staging
    bool gcache_pinned = false;    // a launch holds a pointer to one of the entries: allocation failures elsewhere must not drop the cache
    bool defer_gather = false;     // negf_gr_int: the weighted sum reads the reduced matrices through the permutation (set around run_assemble_inverse)
    bool G_deferred = false;       // ... and the last run_inverse did leave its result un-gathered in W1 (with d_ipiv)
    int inverse_algo = 0;
    int gamma_algo = 0;            // 0: compact Gamma products when the provider allows, 1: always dense
    cplx* d_gsmall = nullptr;      // small Gamma matrices of a batch (compact path)
    size_t gsmall_cap = 0;
    cplx* d_seg_out = nullptr;     // [segments][n*n] results of negf_gr_int_seg
    size_t seg_out_cap = 0;
    cplx* d_ref_P = nullptr;       // [integrals][n*n] running values of negf_gr_int_refine
    size_t ref_P_cap = 0;
    unsigned char* d_ref_meta = nullptr;   // its level table: ratio | maxdp | maxbits [REF_MAX_LEVELS each] | first[REF_MAX_INTS + 1] | level[REF_MAX_INTS] | nanflag[REF_MAX_LEVELS]
    cplx* d_small_part = nullptr;  // per-workgroup partial sums of the small fused kernel
    size_t small_part_cap = 0;
    GjSideStreams gj_side;
    int chain_rr_quantum = -1, chain_rr_slots = 0;   // negf_set_chain_round_robin
    int small_algo = 0;            // 0: n <= 96 takes the fused single-kernel path, 1: never (negf_set_small_algo)
    // pinned host staging of the host-pointer entry points: [E | w] up, [result | info] down, ONE synchronisation
    unsigned char* h_pin = nullptr;
    size_t h_pin_cap = 0;
    int last_m = 0;
    bool profiling = false;
    std::map<std::string, ProfEntry> prof;
    struct Pending { std::string name; hipEvent_t e0, e1; };
    std::vector<Pending> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    std::vector<SigmaProvider*> providers;
};

// Profiling bracket used by the orchestration code: two hipEvents recorded on the
// context's stream around a kernel family, WITHOUT any host synchronisation (the
// timed region of bench.py runs with this enabled); elapsed times are resolved when
// negf_profile_read is called.
struct ProfScope {
    negf_ctx* c; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr;
    FlopCount f0;
    ProfScope(negf_ctx* ctx, const char* nm);
    ~ProfScope();
};

// ------------------------------------------------------------ kernel launchers
// (implemented in k_*.hip; all asynchronous on `st`)

// A[b] = E[b]*S - H - (dense Sigma_b) - scatter(blocks_b)
void launch_assemble(hipStream_t st, int n, int nb, const cplx* E, const cplx* S, const cplx* H,
                     const cplx* sig_dense /*[nb][n*n] or null*/,
                     const cplx* blk /*[nb][blk_stride] or null*/, int blk_stride,
                     int n_contacts, const int* d_nc /*device [n_contacts]*/,
                     const int* d_blk_off, const int* d_inds_off, const int* d_inds,
                     cplx* A);

// in-place inverse of nb matrices; info[b] = 0 or 1-based column of a zero pivot.  false: n exceeds what the
// kernel's LDS staging holds (nothing launched)
bool launch_inverse_unblocked(hipStream_t st, int n, int nb, cplx* A, int* info);
// out-of-place ping-pong between A and B (both [nb][stride]); returns true when the
// inverses end up in B, false when in A
// win_mode: 0 = the measured choice of window kernel per size and batch, 1 = the strip window kernel (gj_strip.h)
// wherever it exists, 2 = the pre-strip kernels (negf_set_inverse_algo 3 / 4)
// skip_gather (in / out): in = the caller can read the reduced matrices through the pivot bookkeeping itself
// (G[i][j] = A[pivrow[i]][colof[j]], piv = [nb][2][n]); out = the gather was left out (windowed path only) -- the return
// value then still says "B", but B was not written
bool launch_inverse_blocked(hipStream_t st, int n, int nb, cplx* A, cplx* B, size_t stride, int* piv, int* info,
                            GjSideStreams* side = nullptr, int win_mode = 0, bool* skip_gather = nullptr);
bool inverse_blocked_supported(int n);
// of the last launch_inverse_blocked: the part of its 6 n np^2 three-real-product flops that ran on the VECTOR pipe
// (the register-strip window kernels), i.e. was not issued to the matrix cores
double inverse_blocked_vector_flops();

// acc += sum_b w[b] * X[b]   (fixed summation order, deterministic); `part` is scratch of
// accumulate_scratch_elems(n2, nb) elements (<= nb * n2 / 32)
void launch_accumulate(hipStream_t st, int n2, int nb, const cplx* w, const cplx* X, cplx* acc, cplx* part);
// the same sum over matrices still in their reduced, un-gathered form: X[b][i][j] = W[b][pivrow_b[i]][colof_b[j]] (piv = [nb][2][n];
// a matrix with info[b] != 0 counts as NaN, as its gathered form would)
void launch_accumulate_perm(hipStream_t st, int n, int nb, const cplx* w, const cplx* W, const int* piv, const int* info, cplx* acc, cplx* part);
bool launch_accumulate_ranges(hipStream_t st, int n, bool perm, const cplx* w, const cplx* XW, const int* piv, const int* info,
                              int nranges, const int* lo, const int* hi, const int* slot, cplx* out, cplx* part);
// nested refinement on the device (density.py:239-268): see refine_levels_kernel
constexpr int REF_MAX_LEVELS = 2048, REF_MAX_INTS = 64, REF_MAX_N = 512;
void launch_refine_levels(hipStream_t st, int n2, int nint, const cplx* sums, const int* first, const double* ratio, double tol,
                          cplx* P, int* level_out, double* maxdp_out);
void launch_refine_level_wide(hipStream_t st, int n2, const cplx* inc, double ratio, cplx* Pk, int level_idx, double tol,
                              int* conv, unsigned long long* maxbits, int* nanflag, double* maxdp_slot);
void launch_cadd(hipStream_t st, size_t count, const cplx* a, const cplx* b, cplx* out);
size_t accumulate_scratch_elems(int n2, int nb);

// C[b] (M x N) = A[b] (M x K) * op(B[b]);  a batch stride of 0 broadcasts one matrix.
// opB bit 1 (value 1): op(B) = B^H with B stored N x K (ldb >= K), else B is K x N (ldb >= N); bit 2 (value 2): the
// caller states that the (square) product is Hermitian -- block tiles above the diagonal are computed and mirrored, those
// below skipped; bit 4 (value 4): the result is stored conjugate-transposed, C = (A op(B))^H (N x M; not with bit 2).
void launch_zgemm(hipStream_t st, int M, int N, int K, int nb,
                  const cplx* A, int lda, size_t strideA,
                  const cplx* B, int ldb, size_t strideB, int opB,
                  cplx* C, int ldc, size_t strideC);

// out[b*out_stride] = Re sum_{i<nr, j<ncol} X[b][i*ldx+j] * conj(G[b][i*ldg+j])
void launch_trace_dot(hipStream_t st, int nr, int ncol, int nb, const cplx* X, int ldx,
                      size_t strideX, const cplx* G, int ldg, size_t strideG,
                      double* out, int out_stride);

// dos_site[b][i] = -Im G[b]_ii / pi ; dos_tot[b] = sum_i
void launch_dos(hipStream_t st, int n, int nb, const cplx* G, double* dos_tot, double* dos_site);

// Gamma = i (Sigma - Sigma^H) for nb dense matrices (stride 0 allowed on input)
void launch_gamma_dense(hipStream_t st, int n, int nb, const cplx* sig, size_t stride_sig, cplx* gam);
void launch_gamma_small(hipStream_t st, int K, int c0, int c1, int nb, const int* d_nc, const int* d_blk_off,
                        const int* d_inds_off, const cplx* blk, size_t blk_stride, cplx* out, size_t out_stride);
void launch_gather_block(hipStream_t st, int n, int nr, int nc, int nb, const cplx* G, size_t strideG,
                         const int* ridx, const int* cidx, cplx* out, size_t out_stride);

// dense Sigma from contact blocks: out[b] = scatter-add of selected contacts (contact<0: all)
void launch_scatter_blocks(hipStream_t st, int n, int nb, const cplx* blk, int blk_stride,
                           int n_contacts, const int* d_nc, const int* d_blk_off,
                           const int* d_inds_off, const int* d_inds, int contact, cplx* out);

// 1-D chain decimation: one workgroup per (energy, contact)
void launch_chain1d(hipStream_t st, const SigmaProvider& p, const int* d_nc, const int* d_blk_off,
                    int nb, const cplx* E, cplx* blk, int* iters, int* conv, cplx* scratch,
                    size_t scratch_per_wg);
size_t chain1d_scratch_per_wg(int nc_max);
// register-stationary MFMA version for n_c <= 64 (k_chain1d_rs.hip)
bool chain1d_lds_supported(int nc_max);
// gold_scratch: chain1d_lds_scratch_elems() complex values of lane-private scratch (may be null:
// the kernel then keeps the old iterate in LDS at a lower occupancy)
// (+ the job queue of a round-robin launch: rr_quantum < 0 = the default, NEGF_CHAIN_RR)
size_t chain1d_lds_scratch_elems(int nc_max, int n_contacts, int nb, int max_sweeps, int rr_quantum);
// order: launch slot -> job (energy * n_contacts + contact) or null for launch order
// gcache / gc_mode: see ChainGEntry -- 0 no cache, 1 store the final iterates, 2 load them and only form Sigma
// rr_quantum / rr_slots: round-robin execution of a launch with more jobs than resident slots (ChainRsArgs; < 0 / 0:
// the defaults -- NEGF_CHAIN_RR or 100 sweeps, every slot of the device); order_trusted = false: the order is a
// guess (predicted from few points), a launch with more jobs than slots runs round robin all the same
void launch_chain1d_lds(hipStream_t st, const SigmaProvider& p, const int* d_nc, const int* d_blk_off, int nb,
                        const cplx* E, cplx* blk, int* iters, int* conv, cplx* gold_scratch, const int* order,
                        cplx* gcache = nullptr, int gc_mode = 0, int rr_quantum = -1, int rr_slots = 0,
                        bool order_trusted = true);
// order[0..count) = jobs by decreasing sweep count predicted from the previous evaluation (k_chain1d_order.hip)
bool chain1d_order_supported(int count);
void launch_chain1d_predict_order(hipStream_t st, const cplx* prevE, const int* prev_iters, int prev_n, int n_contacts,
                                  const cplx* E, int nb, int* order);

// Bethe lattice: one workgroup per (energy, contact); writes per-atom 9x9 blocks
void launch_bethe(hipStream_t st, const SigmaProvider& p, int nb, const cplx* E, cplx* blk,
                  int* iters, int* conv);

void launch_bethe_raw(hipStream_t st, const double* d_H, const double* d_S, const double* d_V, double eta,
                      double conv, double mix, int max_iter, int force_iters, int which, int nb,
                      const cplx* E, cplx* out, int* iters, int* converged);

int run_mfma_selftest(hipStream_t st, double* max_err);

// Small systems (n <= 96): assemble + invert (+ accumulate) in one kernel, k_small_fused.hip
struct SmallFusedArgs {
    int n = 0, m = 0, gp = 0;
    const cplx* E = nullptr;          // [m]
    const cplx* w = nullptr;          // [m]   (accumulate mode)
    const cplx* S = nullptr;          // [n*n]
    const cplx* H = nullptr;          // [n*n]  F, or F + Sigma_tot of a constant provider
    const cplx* sig_dense = nullptr;  // [m][sig_stride] dense Sigma per energy, or null
    size_t sig_stride = 0;
    const cplx* blk = nullptr;        // [m][blk_stride] contact blocks of Sigma(E), or null (then n_contacts = 0)
    int blk_stride = 0, n_contacts = 0;
    const int* pos = nullptr;         // [n_contacts][n] position of an orbital in the contact's index list, -1 outside
    const int* nc = nullptr;          // [n_contacts]
    const int* blk_off = nullptr;     // [n_contacts]
    cplx* partial = nullptr;          // [small_fused_grid(n, m)][n*n] scratch (accumulate mode)
    cplx* out = nullptr;              // [n*n] sum_m w_m G(E_m)   (accumulate mode)
    cplx* Gout = nullptr;             // non-null: STORE mode, G(E_m) -> Gout + m * g_stride
    size_t g_stride = 0;
    int* info = nullptr;              // [m]
    int nseg = 0;                     // > 0: the energies are nseg consecutive segments ending at seg_end[s] (HOST array);
    const int* seg_end = nullptr;     //      out holds one n x n sum per segment
};
bool small_fused_supported(int n);
int small_fused_grid(int n, int m);
void launch_small_fused(hipStream_t st, SmallFusedArgs a);
