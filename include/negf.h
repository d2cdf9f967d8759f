/*
 * negf.h -- C ABI of libnegf_hip.so, the MI355X (gfx950) NEGF energy-grid engine.
 *
 * The reference (wliverno/GauNEGF) is pure Python and has no FFI of its own; the
 * seams this library sits behind are its Python call sites.  Each entry point
 * below names the reference function it replaces (file:line in the reference
 * checkout).  gaunegf_amd/_lib.py binds exactly these symbols with ctypes and
 * gaunegf_amd/{integrate,transport,density,surfG1D,surfGBethe}.py re-expose them
 * under the reference's own names (GrInt, GrLessInt, calculate_transmission ...).
 *
 * Conventions
 *   - complex128 arrays are interleaved (re,im) doubles, row-major (C order), i.e.
 *     the memory of a C-contiguous numpy complex128 array.
 *   - "host" pointers are ordinary CPU memory; "*_dev" pointers are HIP device
 *     memory on the context's GPU (e.g. torch.Tensor.data_ptr()).
 *   - the caller owns every buffer; nothing is retained after a call returns,
 *     except data copied by negf_set_system / negf_sigma_* into the context.
 *   - return 0 = ok, <0 = argument / runtime error, >0 = numerical condition
 *     (NEGF_ESINGULAR: at least one energy hit an exactly zero pivot or a NaN
 *     column; info[k] holds the 1-based pivot column for energy k, LAPACK style,
 *     and G(E_k) is NaN-filled by the blocked kernels -- numpy/jax solve would
 *     return inf/NaN there as well; the other energies are unaffected).  A self-energy
 *     fixed point that stops at its iteration cap is NOT an error (the reference
 *     stops silently too, surfG1D.py:290-293); it is reported via converged[].
 *   - one negf_ctx per (process, GPU); calls are blocking unless noted and the
 *     context is not thread-safe (mirrors FORCE_SYNCHRONOUS, integrate.py:56).
 *   - there is NO CPU fallback: negf_create fails with NEGF_ENODEV without a GPU.
 */
#ifndef NEGF_H
#define NEGF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct negf_ctx negf_ctx;

#define NEGF_OK          0
#define NEGF_EINVAL     (-1)
#define NEGF_ENOMEM     (-2)
#define NEGF_EHIP       (-3)
#define NEGF_ENODEV     (-4)
#define NEGF_ESTATE     (-5)
#define NEGF_ESINGULAR    1

/* contact selector for "use the total self-energy" (Python ind=None,
 * integrate.py:201-204).  Other negative values index from the end like Python
 * (ind=-1 is the last contact, scfE.py:441,444). */
#define NEGF_IND_TOTAL  (-1000)

/* spin layouts of transport.py:193-271 */
#define NEGF_SPIN_RESTRICTED 0   /* 'r'                         */
#define NEGF_SPIN_BLOCK      1   /* 'u' / 'ro' (and 'g' after the host permutes spinor -> block form) */

/* ---------------------------------------------------------------- lifetime */
int         negf_device_count(void);                 /* 0 when no GPU / no driver */
int         negf_create(negf_ctx** out, int device);
void        negf_destroy(negf_ctx* ctx);
const char* negf_strerror(int code);
const char* negf_version(void);
/* launch everything on this hipStream_t (NULL = the default stream) */
int         negf_set_stream(negf_ctx* ctx, void* hip_stream);
/* energies processed per sweep of the workspace (0 = choose from n and free HBM);
 * replaces MAX_VMAP_MEMORY_GB batching, integrate.py:55,100-142 */
int         negf_set_batch(negf_ctx* ctx, int batch);
int         negf_get_batch(negf_ctx* ctx);

/* F,S of the device region: the (F, S) arguments of GrInt/GrLessInt
 * (integrate.py:146,177) and of the transport kernels (transport.py:150-190).
 * Both n*n complex128 (a real F is passed with zero imaginary parts). */
int negf_set_system(negf_ctx* ctx, int n, const double* F_c128, const double* S_c128);

/* 64-bit checksum of a host buffer (no context, no GPU; large buffers on up to 8 threads): what the front end's caches use
 * to notice that a caller changed a matrix in place between two entry points (gaunegf_amd/engine.py fingerprint). */
unsigned long long negf_hash_bytes(const void* data, unsigned long long bytes);

/* The same with a caller's key (0: none).  Equal nonzero keys VOUCH that (F, S) are bitwise the matrices handed over with
 * that key before: a resident system with the key is selected without comparing 2 x 16 n^2 bytes on the host (a front end
 * that keeps private, immutable complex copies of the caller's matrices numbers them -- gaunegf_amd/engine.py; an entry
 * point of a per-GPU share of BASELINE C5 spent 4 ms of its 30 comparing).  An unknown key falls back to the comparison. */
int negf_set_system_keyed(negf_ctx* ctx, int n, const double* F_c128, const double* S_c128, unsigned long long key);

/* ------------------------------------------------- self-energy providers
 * A provider is the device-side lowering of the reference's duck-typed ``g``
 * object (.sigma(E,i) / .sigmaTot(E)).  Handles are small
 * non-negative ints owned by the context. */

/* energy-independent Sigma: surfGTester.py:94-132 / SigmaCalculator static
 * (sig1,sig2), transport.py:77-117.  sigma_c128 = [n_contacts][n][n]; the
 * total is their sum in contact order. */
int negf_sigma_const(negf_ctx* ctx, int n_contacts, const double* sigma_c128, int* handle);

/* 1-D chain decimation provider: surfG1D.py:223-399.  Per contact c the block
 * size nc[c] and orbital indices inds (concatenated), and the nc x nc complex128
 * matrices alpha,Salpha,beta,Sbeta,tau,Stau (concatenated in contact order).
 * eta/conv/relFactor/max_iter default in the reference to ETA=1e-6 (config.py:9),
 * 1e-5 (config.py:15), 0.1 (config.py:16), 2000 (surfG1D.py:265).
 * Pivoting inside the fixed point's inverses (n_c <= 64 kernel): partial pivoting by LAPACK's izamax metric
 * |re| + |im|, compared on the HIGH 32-bit word of the double (sign, exponent, 20 mantissa bits); candidates whose
 * metrics agree to 2^-20 relative count as tied and the lowest row wins, as LAPACK's exact ties do.  (The dense
 * inverses of the hot path compare 36 mantissa bits.)  A different pivot among near-equal candidates changes
 * rounding only: parity with the reference is 1e-10 at a fixed trip count (observed 1e-13).
 * force_iters >= 0 runs exactly that many sweeps (parity at fixed trip count);
 * pass -1 for the reference's data-dependent stopping rule. */
int negf_sigma_chain1d(negf_ctx* ctx, int n_contacts, const int* nc, const int* inds,
                       const double* alpha, const double* Salpha,
                       const double* beta, const double* Sbeta,
                       const double* tau, const double* Stau,
                       double eta, double conv, double relFactor, int max_iter,
                       int force_iters, int* handle);

/* Bethe-lattice provider: surfGBethe.py:479-575, 958-1108.  Per contact: onsite
 * H [9][9] and the 12 direction matrices S,V [12][9][9] (all float64), the list
 * of contact atoms (orbital indices [n_atoms][9], concatenated over contacts) and
 * for every atom its attached-direction list (n_nb[atom] entries of nb_dirs,
 * concatenated).  xi_c128 = S^{1/2} [n][n] or NULL (applied as Xi sig Xi when the
 * .bethe file has Ssss == 0, surfGBethe.py:530-533).  mix=0.5, max_iter=1000 in
 * the reference (:958,:998). */
int negf_sigma_bethe(negf_ctx* ctx, int n_contacts, const int* n_atoms,
                     const int* atom_orbs, const int* n_nb, const int* nb_dirs,
                     const double* H, const double* Slist, const double* Vlist,
                     const double* xi_c128,
                     double eta, double conv, double mix, int max_iter,
                     int force_iters, int* handle);

/* The single-atom Bethe lattice itself: surfGBAt.sigmaK (which = 1, out [m][12][9][9],
 * surfGBethe.py:958-1030) or surfGBAt.sigma (which = 2, out [m][9][9][9], :1032-1108).
 * H [9][9], Slist/Vlist [12][9][9] float64.  iters[m]: bulk sweeps (which = 1) or
 * bulk + (surface << 16) (which = 2); converged[m] likewise (bit 0 bulk, bit 1 surface). */
int negf_bethe_raw(negf_ctx* ctx, const double* H, const double* Slist, const double* Vlist,
                   double eta, double conv, double mix, int max_iter, int force_iters,
                   int which, int m, const double* E_c128, double* out_c128,
                   int* iters, int* converged);

/* Sigma evaluated by the caller for exactly the energies of the NEXT integral
 * (arbitrary user ``g`` objects, integrate.py:169,203-204): sigma_tot [m][n][n]
 * and, optionally, the contact Sigma_c used for Gamma [m][n][n] (NULL = use
 * sigma_tot).  n_contacts_c > 1 means sigma_c holds [m][n_contacts_c][n][n].
 * n_contacts_c < 0: sigma_c holds [m][-n_contacts_c][n][n] matrices that ARE the
 * couplings Gamma (used as given instead of i(Sigma_c - Sigma_c^H)); this is how
 * _transmission_kernel_restricted(E,F,S,sigma_total,gamma1,gamma2)
 * (transport.py:150-157), whose gammas are arguments, is served. */
int negf_sigma_precomputed(negf_ctx* ctx, int m, const double* sigma_tot_c128,
                           int n_contacts_c, const double* sigma_c_c128, int* handle);

int negf_sigma_free(negf_ctx* ctx, int handle);

/* Sigma(E) itself: g.sigma(E,i) / g.sigmaTot(E) (surfG1D.py:344-399,
 * surfGBethe.py:479-575).  contact = NEGF_IND_TOTAL or a contact index.
 * sigma_out [m][n][n]; iters / converged are [m][n_contacts] (may be NULL). */
int negf_sigma_eval(negf_ctx* ctx, int handle, int contact, int m, const double* E_c128,
                    double* sigma_out_c128, int* iters, int* converged);

/* ------------------------------------------------------------- the hot path */

/* sum_m w_m G^r(E_m) -- GrInt, integrate.py:146-173 (+ _gr_matrix_ops :67-71,
 * _GInt :84-142).  out [n][n]; info [m] or NULL. */
int negf_gr_int(negf_ctx* ctx, int handle, int m, const double* E_c128,
                const double* w_c128, double* out_c128, int* info);

/* Several GrInt integrals of one system in ONE pass: the m energies are nseg consecutive segments, seg_end[s] = index
 * one past segment s (seg_end[nseg-1] = m), out [nseg][n][n] receives one sum per segment.  Serves the adaptive
 * integrations -- integratePointsAdaptiveANT, density.py:211-273 (nested levels of 2, 6, 18, 54 ... nodes, only the
 * new nodes of a level are evaluated) and the doubling grids of densityReal, :438-484 -- whose first levels are a
 * handful of points each: evaluated level by level they are launch latency, evaluated together they are one launch.
 * Every segment's sum equals negf_gr_int on that segment alone up to summation order. */
int negf_gr_int_seg(negf_ctx* ctx, int handle, int m, const double* E_c128, const double* w_c128,
                    int nseg, const int* seg_end, double* out_c128, int* info);

/* integratePointsAdaptiveANT (density.py:211-273) with the refinement on the device.  The m energies are the NEW nodes of
 * consecutive levels of nint adaptive integrations of one system: nlev[k] levels for integration k, seg_end[s] as above over all
 * sum(nlev) levels (integration after integration), ratio[s] = the nested-weight ratio of level s (density.py:248-252) -- NaN
 * for the first level of an integration (P = that level's sum, no test); an integration whose first ratio is a number continues
 * from P_in[k].  Per level, in the reference's order: new_P = P * ratio; new_P += sum; maxDP = max|new_P - P|; stop when
 * maxDP < tol (density.py:253-268).  P_out [nint][n][n]: the value at the converged level or after the last one; level_out [nint]:
 * index of the converged level within the call or -1 (continue with P_out as P_in); maxdp_out [sum(nlev)] (NaN: level not
 * consumed / first level).  nint <= 64, at most 2048 levels in all. */
int negf_gr_int_refine(negf_ctx* ctx, int handle, int m, const double* E_c128, const double* w_c128, int nint,
                       const int* nlev, const int* seg_end, const double* ratio, double tol, const double* P_in_c128,
                       double* P_out_c128, int* level_out, double* maxdp_out, int* info);

/* ... and the same for GrLessInt: the levels of the adaptive bias-window integral (densityGrid, density.py:605-658). */
int negf_gless_int_seg(negf_ctx* ctx, int handle, int ind, int m, const double* E_c128, const double* w_c128,
                       int nseg, const int* seg_end, double* out_c128, int* info);

/* sum_m w_m G Gamma_c G^H -- GrLessInt, integrate.py:177-208 (+ :74-82). */
int negf_gless_int(negf_ctx* ctx, int handle, int ind, int m, const double* E_c128,
                   const double* w_c128, double* out_c128, int* info);

/* every G^r(E_m) [m][n][n]; used by parity tests and by callers that need G(E). */
int negf_gr_batch(negf_ctx* ctx, int handle, int m, const double* E_c128,
                  double* G_out_c128, int* info);

/* Re Tr[Gamma_L G Gamma_R G^H] per energy -- _transmission_kernel_restricted /
 * _transmission_kernel_spin_block, transport.py:150-181.  T [m]; Tspin [m][4]
 * (uu,ud,du,dd) or NULL, required for NEGF_SPIN_BLOCK. */
int negf_transmission(negf_ctx* ctx, int handle, int contact_L, int contact_R,
                      int spin_mode, int m, const double* E_c128,
                      double* T, double* Tspin, int* info);

/* -Im diag G / pi and its sum -- _dos_kernel transport.py:183-190,
 * _compute_dos_at_energy density.py:49-54.  dos_site [m][n] or NULL. */
int negf_dos(negf_ctx* ctx, int handle, int m, const double* E_c128,
             double* dos_total, double* dos_site, int* info);

/* ----------------------------------------------- device-resident variants
 * Same operations with the energy grid, weights and result already in HBM on
 * the context's GPU (used by bench.py and by the multi-GPU driver, which
 * all-reduces out_dev with RCCL through torch.distributed).  Asynchronous on
 * the context's stream; negf_sync waits.  negf_last_info copies the per-energy
 * info of the last *_dev call. */
int negf_gr_int_dev(negf_ctx* ctx, int handle, int m, const double* E_dev,
                    const double* w_dev, double* out_dev);
int negf_gless_int_dev(negf_ctx* ctx, int handle, int ind, int m, const double* E_dev,
                       const double* w_dev, double* out_dev);
/* negf_gr_int_seg / negf_gless_int_seg with grid, weights and the nseg results [nseg][n][n] in HBM (seg_end stays a host
 * array): several integrals of one system -- contour + real axis of a density step (scfE.py:316-328), the levels of an
 * adaptive integration -- as ONE pass over this rank's shard of all their energies and ONE all-reduce of out_dev. */
int negf_gr_int_seg_dev(negf_ctx* ctx, int handle, int m, const double* E_dev, const double* w_dev,
                        int nseg, const int* seg_end, double* out_dev);
int negf_gless_int_seg_dev(negf_ctx* ctx, int handle, int ind, int m, const double* E_dev, const double* w_dev,
                           int nseg, const int* seg_end, double* out_dev);
int negf_transmission_dev(negf_ctx* ctx, int handle, int contact_L, int contact_R,
                          int spin_mode, int m, const double* E_dev,
                          double* T_dev, double* Tspin_dev);
int negf_sync(negf_ctx* ctx);
int negf_last_info(negf_ctx* ctx, int m, int* info);
/* sweeps / converged flags of the self-energy fixed points run by the last call
 * ([m][n_contacts] each; the counts the reference's lax.while_loop state carries,
 * surfG1D.py:271-288, surfGBethe.py:1004-1022); zeros / ones for providers without a loop */
int negf_last_iters(negf_ctx* ctx, int handle, int m, int* iters, int* converged);

/* ------------------------------------------ surface Green's function cache
 * The decimation fixed point g(E) of a 1-D chain lead (surfG1D.py:223-295) depends on the lead cell (alpha, Salpha,
 * beta, Sbeta), eta, conv, relFactor, max_iter and E -- not on F and not on the coupling blocks tau (:256-262; setF
 * refreshes tau only, :319-329).  The reference recomputes it inside every vmapped closure (integrate.py:168-171) and
 * twice per energy in GrLessInt (:201-204).  The context keeps the final iterates of the last `max_grids` launches of
 * the n_c <= 64 chain kernel in HBM (n_contacts * n_c^2 * 16 bytes per energy), keyed BITWISE on those inputs and on
 * the energy list of the launch; a launch that finds its key only forms Sigma = t g t^H with the provider's CURRENT
 * tau -- the last pass of the same kernel, so a hit equals a miss bit for bit, sweep counts and flags included.  The
 * key outlives providers: one re-created after setF, or the t = I variant behind surfG.g(), hits entries of its
 * predecessor.  Defaults: 512 grids and 8 GB in all, least recently used entries evicted first (one SCF cycle at a
 * fixed Fermi level is ~10^2 adaptive grids of 2 ... 324 points; BASELINE C3's 2000-point grid is 160 MB); 0 grids
 * switches the cache off and frees it (bench.py's headline runs cold that way).  Launches whose g would exceed 4 GB
 * are not cached.  The *_dev entry points download their energy list (16 bytes per point, one stream
 * synchronisation) to form the key while the cache is on.  Environment: NEGF_CHAIN_CACHE=<grids> presets max_grids. */
int negf_set_chain_cache(negf_ctx* ctx, int max_grids);
int negf_set_chain_cache_bytes(negf_ctx* ctx, long long max_bytes);
int negf_chain_cache_clear(negf_ctx* ctx);
/* counters since negf_create; entries / bytes currently held (any pointer may be NULL) */
int negf_chain_cache_stats(negf_ctx* ctx, long long* hits, long long* misses, long long* entries, long long* bytes);

/* ------------------------------------------------------------- diagnostics */
/* hipEvent timing of the library's own kernels, per kernel family
 * ("inverse", "assemble", "accumulate", "zgemm", "trace", "chain1d", "bethe"). */
int negf_profile_enable(negf_ctx* ctx, int on);
int negf_profile_reset(negf_ctx* ctx);
int negf_profile_read(negf_ctx* ctx, const char* family, double* total_ms, int* launches);
/* flops of the family's launches since negf_profile_reset, two ways: ALGORITHMIC (8 per complex multiply-add: 8 M N K
 * per dense product, 8 n^3 per inverse -- what the reference's solve / matmul would be charged) and ISSUED
 * to the matrix cores (the kernels use the 3-real-product form of a complex product, compute 16-granular tiles, skip
 * the lower block tiles of Hermitian products; pivot steps and other vector work are not matrix-core flops).  Only
 * the second may be divided by the FP64 MFMA peak and called utilisation.  Families: "inverse", "zgemm". */
int negf_profile_read_flops(negf_ctx* ctx, const char* family, double* flops_algorithmic, double* flops_mfma_issued);
/* choose the inverse kernel: 0 = auto, 1 = unblocked Gauss-Jordan (any n),
 * 2 = blocked Gauss-Jordan with FP64 MFMA trailing updates (window kernel by size and batch),
 * 3 = the same with the register-strip window kernel wherever it exists (64 <= n <= 1024; tests, A/B),
 * 4 = the same with the team window kernels / the single-workgroup kernel only (the round-4 configuration) */
int negf_set_inverse_algo(negf_ctx* ctx, int algo);
/* systems of n <= 96 orbitals: 0 = auto -- with negf_set_inverse_algo(0), E S - F - Sigma is assembled, inverted and
 * (GrInt) accumulated in ONE kernel with the matrix held in the registers of a compute unit (no n x n work area in
 * HBM; the SCF call pattern is ~10^2 integrals of 2 ... 324 points per density step, scfE.py:301-462); 1 = the
 * assemble / inverse / accumulate kernel sequence through HBM that larger systems use (cross-check, A/B) */
int negf_set_small_algo(negf_ctx* ctx, int algo);
/* CHAIN1D launches with more fixed points (energy x contact) than the device holds at once: the fixed points differ
 * up to 20x in their sweep counts (surfG1D.py:271-288 stops each at its own residual), and the counts are unknown
 * before the first evaluation.  The kernel runs them ROUND ROBIN: one persistent workgroup per resident slot, a fixed
 * point runs `quantum` sweeps and, when others wait, goes to the back of a device-side queue with its iterate.  The
 * chip then stays full until fewer fixed points than slots are left, whatever order they were started in; results
 * do not depend on it (a fixed point is a sequence of sweeps on its own data).  quantum: < 0 = default -- quanta of
 * NEGF_CHAIN_RR (else 100) sweeps for launches whose sweep counts CANNOT be predicted (the first evaluation of a grid by
 * a provider), and one workgroup per fixed point, started longest first by the counts predicted from the previous
 * evaluation, for those that can (that launch is ~1 % faster when the order is good); 0 = never round robin;
 * > 0 = always, with this quantum.  slots: 0 = every resident slot of the device; > 0 caps them (tests). */
int negf_set_chain_round_robin(negf_ctx* ctx, int quantum, int slots);
/* G Gamma G^H (integrate.py:81) and Tr[Gamma_L G Gamma_R G^H] (transport.py:156-157):
 * 0 = auto -- when the coupling matrices only touch the contact orbitals (CONST providers
 * with a small support, CHAIN1D / BETHE blocks without an orthogonalisation matrix) the
 * products run on the columns / the block of G on those orbitals only (same sums, the
 * terms that are exactly zero are skipped); 1 = always the dense n x n products */
int negf_set_gamma_algo(negf_ctx* ctx, int algo);
/* run the FP64 MFMA fragment-layout probe; max abs error vs an exact integer
 * product (0.0 expected) */
int negf_selftest_mfma(negf_ctx* ctx, double* max_err);

#ifdef __cplusplus
}
#endif
#endif /* NEGF_H */
