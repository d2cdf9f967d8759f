// C ABI of libnegf_hip.so (include/negf.h): context, self-energy providers and the
// orchestration of the per-batch kernel sequence.  No compute happens on the
// host here; there is no CPU fallback (negf_create fails without a GPU).
#include "negf_common.h"
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdlib>
#include <new>

static hipEvent_t prof_get_event(negf_ctx* c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

FlopCount g_negf_flops;

ProfScope::ProfScope(negf_ctx* ctx, const char* nm) : c(ctx), name(nm)
{
    if (!c->profiling) return;
    f0 = g_negf_flops;
    e0 = prof_get_event(c); e1 = prof_get_event(c);
    if (e0) (void)hipEventRecord(e0, c->stream);
}

ProfScope::~ProfScope()
{
    if (!c->profiling) return;
    { auto& e = c->prof[name]; e.flops_alg += g_negf_flops.alg - f0.alg; e.flops_mfma += g_negf_flops.mfma - f0.mfma; }
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, c->stream);
    c->prof_pending.push_back({name, e0, e1});
}

static void prof_resolve(negf_ctx* c)
{
    for (auto& p : c->prof_pending) {
        (void)hipEventSynchronize(p.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            auto& e = c->prof[p.name]; e.ms += ms; e.launches += 1;
        }
        c->ev_pool.push_back(p.e0); c->ev_pool.push_back(p.e1);
    }
    c->prof_pending.clear();
}

namespace {

int wait_stream(negf_ctx* c);

// every live context of the process: when an allocation fails, the g(E) caches (up to 8 GB each, reusable results, not
// state) are dropped before the allocation is given up
std::vector<negf_ctx*>& live_contexts() { static std::vector<negf_ctx*> v; return v; }
bool drop_gcaches();

template <typename T>
int dev_alloc(T** p, size_t count, bool may_drop_caches = true /* false: the allocation IS a cache entry */)
{
    *p = nullptr;
    if (count == 0) return NEGF_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (may_drop_caches && drop_gcaches()) e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));      // once more without the caches
        if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return NEGF_ENOMEM; }
    }
    return NEGF_OK;
}
template <typename T>
void dev_free(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

template <typename T>
int upload(negf_ctx* c, T* dst, const T* src, size_t count)
{
    if (count == 0) return NEGF_OK;
    NEGF_HIP_CHECK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));   // src may be freed by the caller on return
    return NEGF_OK;
}
template <typename T>
int download(negf_ctx* c, T* dst, const T* src, size_t count)
{
    if (count == 0) return NEGF_OK;
    NEGF_HIP_CHECK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return wait_stream(c);
}

void free_provider(SigmaProvider* p)
{
    if (!p) return;
    dev_free(p->d_const_c); dev_free(p->d_const_tot); dev_free(p->d_hbase); dev_free(p->d_const_blk);
    dev_free(p->d_inds); dev_free(p->d_nc); dev_free(p->d_blk_off); dev_free(p->d_inds_off);
    dev_free(p->d_n_atoms); dev_free(p->d_atom_off); dev_free(p->d_pos);
    dev_free(p->d_alpha); dev_free(p->d_Salpha); dev_free(p->d_beta); dev_free(p->d_Sbeta);
    dev_free(p->d_tau); dev_free(p->d_Stau);
    dev_free(p->d_atom_orbs); dev_free(p->d_nb_off); dev_free(p->d_nb_dirs);
    dev_free(p->d_H); dev_free(p->d_Slist); dev_free(p->d_Vlist); dev_free(p->d_xi);
    dev_free(p->d_pre_tot); dev_free(p->d_pre_c); dev_free(p->d_order); dev_free(p->d_prevE); dev_free(p->d_prev_iters); dev_free(p->d_curE); dev_free(p->d_cur_iters);
    delete p;
}

void free_workspace(negf_ctx* c)
{
    dev_free(c->d_A); dev_free(c->d_T1); dev_free(c->d_T2); dev_free(c->d_blk);
    dev_free(c->d_ipiv); dev_free(c->d_site); dev_free(c->d_scratch); dev_free(c->d_gsmall);
    dev_free(c->d_small_part);
    c->batch = 0; c->blk_cap = 0; c->scratch_cap = 0; c->gsmall_cap = 0; c->small_part_cap = 0;
}

void free_mbuffers(negf_ctx* c)
{
    dev_free(c->d_info); dev_free(c->d_iters); dev_free(c->d_conv);
    dev_free(c->d_E); dev_free(c->d_w); dev_free(c->d_scal);
    c->m_cap = 0; c->contacts_cap = 0; c->h_E_valid = false;
}

void free_gentry(ChainGEntry& e)
{
    dev_free(e.d_g); dev_free(e.d_it); dev_free(e.d_cv);
    e.g_cap = 0; e.it_cap = 0; e.valid = false;
}

void free_gcache(negf_ctx* c)
{
    for (auto& e : c->gcache) free_gentry(e);
    c->gcache.clear();
}

// true: something was freed.  (Entries in use by a launch in flight are freed by hipFree after that launch: hipFree
// synchronises the device.)
bool drop_gcaches()
{
    bool any = false;
    for (negf_ctx* c : live_contexts())
        if (!c->gcache.empty() && !c->gcache_pinned) { free_gcache(c); any = true; }
    return any;
}

// 64-bit mixing hash over the 8-byte words of a buffer (a filter in front of the bitwise comparisons of the g(E) cache)
unsigned long long hash_words(const void* data, size_t bytes, unsigned long long h = 0x9E3779B97F4A7C15ull)
{
    const unsigned long long* w = static_cast<const unsigned long long*>(data);
    for (size_t i = 0; i < bytes / 8; ++i) { h = (h ^ w[i]) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    return h;
}

// the energies E_dev[0..nb) on the host: the copy stage_grid kept when they are the staged grid, a download otherwise
const cplx* host_energies(negf_ctx* c, const cplx* E_dev, int nb, std::vector<cplx>& tmp)
{
    if (c->h_E_valid && c->d_E && E_dev >= c->d_E && E_dev + nb <= c->d_E + c->h_E.size())
        return c->h_E.data() + (E_dev - c->d_E);
    tmp.resize((size_t)nb);
    if (hipMemcpyAsync(tmp.data(), E_dev, (size_t)nb * sizeof(cplx), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return tmp.data();
}

// Look the launch (provider p, energies Eh[0..nb)) up in the context's g(E) cache.  *hit: the entry holds it.  Otherwise
// the returned entry (least recently used, or a new one) has room for it and is to be filled by the launch; nullptr:
// not cached (cache off, entry too large, out of memory).
ChainGEntry* gcache_lookup(negf_ctx* c, const SigmaProvider* p, const cplx* Eh, int nb, bool* hit)
{
    *hit = false;
    if (c->gcache_max <= 0 || !Eh || nb <= 0 || !p->h_lead) return nullptr;
    const size_t g_elems = (size_t)nb * p->blk_stride, it_elems = (size_t)nb * p->n_contacts;
    if (g_elems * sizeof(cplx) > c->gcache_entry_bytes_max) return nullptr;
    const unsigned long long eh = hash_words(Eh, (size_t)nb * sizeof(cplx));
    for (auto& e : c->gcache) {
        if (!e.valid || e.E_hash != eh || e.lead_hash != p->lead_hash || (int)e.E.size() != nb) continue;
        if (e.eta != p->eta || e.conv != p->conv || e.relFactor != p->relFactor || e.max_iter != p->max_iter ||
            e.force_iters != p->force_iters || e.nc != p->nc) continue;
        if (std::memcmp(e.E.data(), Eh, (size_t)nb * sizeof(cplx)) != 0) continue;
        if (e.lead != p->h_lead && (e.lead->size() != p->h_lead->size() ||
                                    std::memcmp(e.lead->data(), p->h_lead->data(), e.lead->size() * sizeof(cplx)) != 0)) continue;
        e.used = ++c->gcache_clock;
        ++c->gcache_hits;
        *hit = true;
        return &e;
    }
    ++c->gcache_misses;
    // room for the new entry: a free slot while the cache is below its entry and byte limits, else the least
    // recently used entries go -- one whose buffers are large enough is reused as it is
    auto bytes_of = [](const ChainGEntry& e) { return e.g_cap * sizeof(cplx) + 2 * e.it_cap * sizeof(int); };
    const size_t need_b = g_elems * sizeof(cplx) + 2 * it_elems * sizeof(int);
    size_t held = 0;
    for (const auto& e : c->gcache) held += bytes_of(e);
    ChainGEntry* v = nullptr;
    for (auto& e : c->gcache) if (!e.valid && e.g_cap >= g_elems && e.it_cap >= it_elems) { v = &e; break; }   // an invalidated entry that fits
    bool synced = false;
    while (!v) {
        int with_buffers = 0;
        ChainGEntry *lru = nullptr, *empty = nullptr;
        for (auto& e : c->gcache) {
            if (!e.d_g) { if (!empty) empty = &e; continue; }
            ++with_buffers;
            if (!lru || (!e.valid && lru->valid) || (e.valid == lru->valid && e.used < lru->used)) lru = &e;
        }
        if (with_buffers < c->gcache_max && held + need_b <= c->gcache_bytes_max) {
            if (!empty) { c->gcache.emplace_back(); empty = &c->gcache.back(); }     // (capacity reserved: no reallocation)
            v = empty;
            break;
        }
        if (!lru) return nullptr;                                                    // larger than the whole budget
        lru->valid = false;
        if (lru->g_cap >= g_elems && lru->it_cap >= it_elems) { v = lru; break; }    // taken over as it is
        if (!synced) { (void)hipStreamSynchronize(c->stream); synced = true; }       // earlier kernels may still read it
        held -= bytes_of(*lru);
        free_gentry(*lru);
    }
    if (g_elems > v->g_cap || it_elems > v->it_cap) {
        free_gentry(*v);
        if (dev_alloc(&v->d_g, g_elems, false) || dev_alloc(&v->d_it, it_elems, false) || dev_alloc(&v->d_cv, it_elems, false)) { free_gentry(*v); return nullptr; }
        v->g_cap = g_elems; v->it_cap = it_elems;
    }
    v->nc = p->nc; v->lead = p->h_lead; v->lead_hash = p->lead_hash; v->E_hash = eh;
    v->eta = p->eta; v->conv = p->conv; v->relFactor = p->relFactor; v->max_iter = p->max_iter; v->force_iters = p->force_iters;
    v->E.assign(Eh, Eh + nb);
    v->used = ++c->gcache_clock;
    return v;
}

// Energies in flight for a grid of m points.  Once the transmission entry point has been used on this context
// the workspace is sized for TWICE the grid (it carves two work areas per energy out of it), so that the calls of
// one workflow -- GrInt, calculate_transmission, GrLessInt on the same grid -- find it in place: it only ever
// grows, and only when a larger grid arrives.  Contexts that only integrate (SCF loops, the headline bench) get
// the single-grid size: half the HBM, which matters when ranks or processes share a device.
int auto_batch(negf_ctx* c, int m)
{
    if (c->batch_user > 0) return std::min(c->batch_user, std::max(m, 1));
    m = (c->transmission_seen ? 2 : 1) * std::max(m, 1);
    // three n x n complex128 work matrices per in-flight energy.  The working set may take a
    // quarter of the free HBM (288 GB per MI355X), at most 64 GB: large matrices need hundreds of
    // energies in flight so that the one-workgroup-per-matrix panel kernels cover the 256 CUs
    // (n = 2000: 192 MB per energy -> 333 in flight); never above the grid length.
    const double per = 3.0 * 16.0 * (double)c->n * (double)c->n;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)24e9;
    const double budget = std::min(64.0e9, std::max(2.0e9, 0.25 * (double)free_b));
    long b = (long)(budget / std::max(per, 1.0));
    b = std::max(1L, std::min(b, 4096L));
    return (int)std::min<long>(b, std::max(m, 1));
}

int ensure_workspace(negf_ctx* c, int m, int blk_stride, int min_batch = 1)
{
    // (the workspace in place already covers the grid: no hipMemGetInfo, which costs more than a small integral)
    const int covers = (c->transmission_seen ? 2 : 1) * std::max(m, 1);
    const int want = (c->batch_user == 0 && c->batch >= std::min(covers, 4096) && c->batch >= min_batch)
                         ? c->batch : std::max(auto_batch(c, m), min_batch);
    if (want > c->batch) {
        free_workspace(c);
        const size_t n2 = (size_t)c->n * c->n;
        int rc;
        // (+ 64 elements: a guard behind the last matrix; no kernel relies on it any more -- the column-block
        //  update of the windowed inverse clamps its lane offset to the window width)
        if ((rc = dev_alloc(&c->d_A, n2 * want + 64))) return rc;
        if ((rc = dev_alloc(&c->d_T1, n2 * want + 64))) return rc;
        if ((rc = dev_alloc(&c->d_T2, n2 * want + 64))) return rc;
        if ((rc = dev_alloc(&c->d_ipiv, (size_t)2 * c->n * want))) return rc;
        if ((rc = dev_alloc(&c->d_site, (size_t)c->n * want))) return rc;
        c->batch = want;
    }
    const int need_blk = blk_stride * c->batch;
    if (need_blk > c->blk_cap) {
        dev_free(c->d_blk);
        int rc = dev_alloc(&c->d_blk, (size_t)need_blk);
        if (rc) return rc;
        c->blk_cap = need_blk;
    }
    return NEGF_OK;
}

// Wait for the context's stream at the end of a host-pointer call.  hipStreamSynchronize parks the thread and is woken
// by an interrupt -- tens of microseconds after the GPU is done, as much as a small integral itself takes -- so the
// stream is POLLED for the first 2 ms (an SCF-sized call is over long before) and only then handed to the blocking wait.
// NEGF_SYNC_SPIN=0: always block.
int wait_stream(negf_ctx* c)
{
    static int spin = -1;
    if (spin < 0) { const char* e = getenv("NEGF_SYNC_SPIN"); spin = e ? atoi(e) : 1; }
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) return NEGF_OK;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return NEGF_OK;
}

int ensure_blk(negf_ctx* c, size_t elems)
{
    if (elems <= (size_t)c->blk_cap) return NEGF_OK;
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    dev_free(c->d_blk); c->blk_cap = 0;
    int rc = dev_alloc(&c->d_blk, elems);
    if (rc) return rc;
    c->blk_cap = (int)elems;
    return NEGF_OK;
}

int ensure_pinned(negf_ctx* c, size_t bytes)
{
    if (bytes <= c->h_pin_cap) return NEGF_OK;
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));           // copies from / into the old buffer
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_cap = 0; }
    if (c->gj_side.ok) {
        (void)hipEventDestroy(c->gj_side.fork);
        for (int g = 0; g < GjSideStreams::MAXG - 1; ++g) { (void)hipStreamDestroy(c->gj_side.s[g]); (void)hipEventDestroy(c->gj_side.join[g]); }
        c->gj_side.ok = false;
    }
    const size_t cap = std::max(bytes + bytes / 2, (size_t)1 << 20);
    void* q = nullptr;
    if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return NEGF_ENOMEM; }
    c->h_pin = static_cast<unsigned char*>(q); c->h_pin_cap = cap;
    return NEGF_OK;
}

// ---- small systems: the single-kernel path (k_small_fused.hip)
bool small_path(const negf_ctx* c, const SigmaProvider* p)
{
    if (c->small_algo != 0 || c->inverse_algo != 0 || !small_fused_supported(c->n)) return false;
    if (p->kind == SK_CONST || p->kind == SK_PRECOMPUTED) return true;
    return (p->kind == SK_CHAIN1D || p->kind == SK_BETHE) && !p->d_xi && p->d_pos;
}

// arguments of the fused kernel for the energies [m0, m0 + nb) of provider p (Sigma blocks of a block provider are in
// c->d_blk: run_sigma_blocks has run for exactly this range)
SmallFusedArgs small_args(negf_ctx* c, SigmaProvider* p, int m0, int nb, const cplx* E)
{
    SmallFusedArgs a;
    a.n = c->n; a.m = nb; a.E = E + m0; a.S = c->d_S; a.H = c->d_F;
    a.info = c->d_info + m0;
    const size_t n2 = (size_t)c->n * c->n;
    if (p->kind == SK_CONST) a.H = p->d_hbase;
    else if (p->kind == SK_PRECOMPUTED) { a.sig_dense = p->d_pre_tot + n2 * m0; a.sig_stride = n2; }
    else { a.blk = c->d_blk; a.blk_stride = p->blk_stride; a.n_contacts = p->n_contacts; a.pos = p->d_pos; a.nc = p->d_nc; a.blk_off = p->d_blk_off; }
    return a;
}

int ensure_mbuffers(negf_ctx* c, int m, int contacts)
{
    contacts = std::max(contacts, 1);
    if (m <= c->m_cap && contacts <= c->contacts_cap) return NEGF_OK;
    const int mm = std::max(m, c->m_cap), cc = std::max(contacts, c->contacts_cap);
    free_mbuffers(c);
    int rc;
    if ((rc = dev_alloc(&c->d_info, (size_t)mm))) return rc;
    if ((rc = dev_alloc(&c->d_iters, (size_t)mm * cc))) return rc;
    if ((rc = dev_alloc(&c->d_conv, (size_t)mm * cc))) return rc;
    if ((rc = dev_alloc(&c->d_E, (size_t)mm))) return rc;
    if ((rc = dev_alloc(&c->d_w, (size_t)mm))) return rc;
    if ((rc = dev_alloc(&c->d_scal, (size_t)mm * 8))) return rc;
    c->m_cap = mm; c->contacts_cap = cc;
    return NEGF_OK;
}

SigmaProvider* get_provider(negf_ctx* c, int handle)
{
    if (!c || handle < 0 || handle >= (int)c->providers.size()) return nullptr;
    return c->providers[handle];
}

// A handle is the provider's index in a table that only grows: a freed slot stays empty and is never
// handed out again, so a stale handle (freed, or dropped by a change of the matrix dimension) can only
// ever name an empty slot -- negf_* calls then return NEGF_EINVAL instead of running on another
// object's self-energy.
int add_provider(negf_ctx* c, SigmaProvider* p)
{
    c->providers.push_back(p);
    return (int)c->providers.size() - 1;
}

// Python-style contact index normalisation; returns -1 for "total", -2 for invalid
int norm_contact(const SigmaProvider* p, int ind)
{
    if (ind == NEGF_IND_TOTAL) return -1;
    int nc = p->n_contacts;
    if (p->kind == SK_PRECOMPUTED) nc = std::max(p->pre_nc, 1);
    if (ind < 0) ind += nc;
    if (ind < 0 || ind >= nc) return -2;
    return ind;
}

int run_inverse(negf_ctx* c, int nb, int* info)
{
    ProfScope ps(c, "inverse");
    int algo = c->inverse_algo;
    c->G_deferred = false;
    if (algo == 0) algo = inverse_blocked_supported(c->n) ? 2 : 1;
    const int win_mode = algo == 3 ? 1 : algo == 4 ? 2 : 0;      // 3 / 4: the blocked path with the window kernel chosen
    if (algo > 2) algo = 2;
    bool in_b = false;
    if (algo == 2) {
        // false: no blocked kernel serves this n (nothing was launched) -> the unblocked kernel
        bool skip = c->defer_gather;
        in_b = launch_inverse_blocked(c->stream, c->n, nb, c->d_A, c->d_T1, (size_t)c->n * c->n, c->d_ipiv, info, &c->gj_side, win_mode, &skip);
        c->G_deferred = skip;
        if (!in_b) algo = 1;
    }
    if (algo == 1 && !launch_inverse_unblocked(c->stream, c->n, nb, c->d_A, info)) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipGetLastError());
    {   // 8 n^3 algorithmic; the blocked kernels run every rank-NB update on the matrix cores in 3M form over
        // 16-granular tiles with K in steps of four (the zero k-steps of a ragged last window are skipped); the pivot steps
        // themselves are vector work, and so is everything the strip window kernels do; the unblocked kernel issues none
        const double n = c->n, np = (double)((c->n + 15) & ~15), k4 = (double)((c->n + 3) & ~3);
        negf_count_flops(8.0 * n * n * n * nb, algo == 2 ? 6.0 * np * np * k4 * nb - inverse_blocked_vector_flops() : 0.0);
    }
    c->G = in_b ? c->d_T1 : c->d_A;
    c->W1 = in_b ? c->d_A : c->d_T1;
    c->W2 = c->d_T2;
    return NEGF_OK;
}

// Sigma blocks of a block provider for energies E[0..nb) -> c->d_blk
// (m0 = position of this chunk in the grid being evaluated: the sweep-count prediction keeps whole evaluations)
int run_sigma_blocks(negf_ctx* c, SigmaProvider* p, int nb, const cplx* E, int* iters, int* conv, int m0)
{
    if (p->kind == SK_CHAIN1D) {
        static int force_v1 = -1;
        if (force_v1 < 0) { const char* e = getenv("NEGF_CHAIN1D_ALGO"); force_v1 = (e && strcmp(e, "global") == 0) ? 1 : 0; }
        const bool lds_path = chain1d_lds_supported(p->nc_max) && !force_v1;
        // the g(E) cache (ChainGEntry): a launch whose lead and energies were evaluated before only forms Sigma = t g t^H
        ChainGEntry* ent = nullptr;
        bool hit = false;
        struct Pin { negf_ctx* c; ~Pin() { c->gcache_pinned = false; } } pin{c};      // `ent` points into the cache until this block ends
        c->gcache_pinned = true;
        if (lds_path && c->gcache_max > 0) {
            std::vector<cplx> tmp;
            ent = gcache_lookup(c, p, host_energies(c, E, nb, tmp), nb, &hit);
        }
        ProfScope ps(c, hit ? "chain1d_hit" : "chain1d");
        if (lds_path) {
            const size_t need = chain1d_lds_scratch_elems(p->nc_max, p->n_contacts, nb, std::max(p->max_iter, p->force_iters), c->chain_rr_quantum);
            if (need > c->scratch_cap) {
                NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
                dev_free(c->d_scratch); c->scratch_cap = 0;
                int rc = dev_alloc(&c->d_scratch, need);
                if (rc) return rc;
                c->scratch_cap = need;
            }
            const int jobs = nb * p->n_contacts;
            if (hit) {
                launch_chain1d_lds(c->stream, *p, p->d_nc, p->d_blk_off, nb, E, c->d_blk, iters, conv, c->d_scratch, nullptr, ent->d_g, 2);
                if (iters) NEGF_HIP_CHECK(hipMemcpyAsync(iters, ent->d_it, (size_t)jobs * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                if (conv) NEGF_HIP_CHECK(hipMemcpyAsync(conv, ent->d_cv, (size_t)jobs * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                return NEGF_OK;
            }
            if (ent && (!iters || !conv)) ent = nullptr;    // (an entry carries the counts and flags of its launch)
            // jobs in the order of decreasing sweep counts: the counts are predicted from the previous evaluation of
            // this provider -- for each energy the count of the nearest energy evaluated then (Fermi searches and SCF
            // cycles evaluate the same or slightly moved grids over and over; the same grid gets exactly its learned
            // order) -- and sorted on the device, no host sync; the first evaluation runs in launch order.  What the
            // order is for: a launch with FEWER jobs than resident slots (the SCF-sized grids) is dispatched to the
            // compute units in this order, so the long jobs are spread over the chip instead of sharing a unit with
            // their neighbours in energy (648 jobs: 42 against 48 ms); a launch with MORE jobs runs them round robin
            // (k_chain1d_rs.hip) and only takes the order as the initial content of its queue.
            const bool can_order = p->force_iters < 0 && iters && chain1d_order_supported(jobs);
            const int* order = nullptr;
            bool order_trusted = true;                      // a prediction from fewer than half as many energies as this
            if (can_order) {                                //  launch evaluates is a guess: such a launch runs round robin
                if (jobs > p->order_cap) {
                    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
                    dev_free(p->d_order); p->order_cap = 0;
                    int rc = dev_alloc(&p->d_order, (size_t)jobs);
                    if (rc) return rc;
                    p->order_cap = jobs;
                }
                if (m0 == 0 && p->cur_n > 0) {
                    // a new evaluation begins: the one recorded so far becomes the reference
                    std::swap(p->d_prevE, p->d_curE); std::swap(p->d_prev_iters, p->d_cur_iters);
                    std::swap(p->prev_cap, p->cur_cap);
                    p->prev_n = p->cur_n; p->cur_n = 0;
                }
                order_trusted = 2 * (p->prev_n > 0 ? p->prev_n : p->cur_n) >= nb;
                if (p->prev_n > 0) {
                    launch_chain1d_predict_order(c->stream, p->d_prevE, p->d_prev_iters, p->prev_n, p->n_contacts, E, nb, p->d_order);
                    order = p->d_order;
                } else if (p->cur_n > 0) {
                    // the first evaluation ever, arriving in chunks: the chunks so far are all there is to learn from
                    launch_chain1d_predict_order(c->stream, p->d_curE, p->d_cur_iters, p->cur_n, p->n_contacts, E, nb, p->d_order);
                    order = p->d_order;
                }
            }
            launch_chain1d_lds(c->stream, *p, p->d_nc, p->d_blk_off, nb, E, c->d_blk, iters, conv, c->d_scratch, order,
                               ent ? ent->d_g : nullptr, 1, c->chain_rr_quantum, c->chain_rr_slots, order_trusted);
            if (ent) {
                NEGF_HIP_CHECK(hipMemcpyAsync(ent->d_it, iters, (size_t)jobs * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                NEGF_HIP_CHECK(hipMemcpyAsync(ent->d_cv, conv, (size_t)jobs * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                NEGF_HIP_CHECK(hipGetLastError());                 // an entry whose launch failed never becomes a hit
                ent->valid = true;
            }
            if (can_order && (m0 == 0 || m0 == p->cur_n)) {
                // append this chunk's energies and counts to the record of the evaluation in progress
                if (m0 == 0) p->cur_n = 0;
                if (m0 + nb > p->cur_cap) {
                    const int cap = std::max(m0 + nb, 2 * p->cur_cap);
                    cplx* nE = nullptr; int* nI = nullptr;
                    int rc;
                    if ((rc = dev_alloc(&nE, (size_t)cap)) || (rc = dev_alloc(&nI, (size_t)cap * p->n_contacts))) { dev_free(nE); return rc; }
                    if (p->cur_n > 0) {
                        NEGF_HIP_CHECK(hipMemcpyAsync(nE, p->d_curE, (size_t)p->cur_n * sizeof(cplx), hipMemcpyDeviceToDevice, c->stream));
                        NEGF_HIP_CHECK(hipMemcpyAsync(nI, p->d_cur_iters, (size_t)p->cur_n * p->n_contacts * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                    }
                    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
                    dev_free(p->d_curE); dev_free(p->d_cur_iters);
                    p->d_curE = nE; p->d_cur_iters = nI; p->cur_cap = cap;
                }
                NEGF_HIP_CHECK(hipMemcpyAsync(p->d_curE + m0, E, (size_t)nb * sizeof(cplx), hipMemcpyDeviceToDevice, c->stream));
                NEGF_HIP_CHECK(hipMemcpyAsync(p->d_cur_iters + (size_t)m0 * p->n_contacts, iters, (size_t)jobs * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                p->cur_n = m0 + nb;
            }
            return NEGF_OK;
        }
        const size_t per = chain1d_scratch_per_wg(p->nc_max);
        const size_t need = per * p->n_contacts * nb;
        if (need > c->scratch_cap) {
            NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
            dev_free(c->d_scratch); c->scratch_cap = 0;
            int rc = dev_alloc(&c->d_scratch, need);
            if (rc) return rc;
            c->scratch_cap = need;
        }
        launch_chain1d(c->stream, *p, p->d_nc, p->d_blk_off, nb, E, c->d_blk, iters, conv, c->d_scratch, per);
    } else if (p->kind == SK_BETHE) {
        ProfScope ps(c, "bethe");
        launch_bethe(c->stream, *p, nb, E, c->d_blk, iters, conv);
    }
    return NEGF_OK;
}

// assemble A_b = E_b S - F - Sigma_b for batch [m0, m0+nb) into c->d_A
int run_assemble(negf_ctx* c, SigmaProvider* p, int m0, int nb, const cplx* E)
{
    int rc = NEGF_OK;
    if (p->kind == SK_CHAIN1D || p->kind == SK_BETHE) {
        rc = run_sigma_blocks(c, p, nb, E + m0, c->d_iters + (size_t)m0 * p->n_contacts,
                              c->d_conv + (size_t)m0 * p->n_contacts, m0);
        if (rc) return rc;
    }
    ProfScope ps(c, "assemble");
    const size_t n2 = (size_t)c->n * c->n;
    switch (p->kind) {
    case SK_CONST:
        launch_assemble(c->stream, c->n, nb, E + m0, c->d_S, p->d_hbase, nullptr, nullptr, 0, 0,
                        nullptr, nullptr, nullptr, nullptr, c->d_A);
        break;
    case SK_PRECOMPUTED:
        launch_assemble(c->stream, c->n, nb, E + m0, c->d_S, c->d_F, p->d_pre_tot + n2 * m0, nullptr, 0,
                        0, nullptr, nullptr, nullptr, nullptr, c->d_A);
        break;
    case SK_CHAIN1D:
    case SK_BETHE:
        if (p->d_xi) {
            // Sigma = Xi blockdiag Xi (surfGBethe.py:530-533): dense via two products
            launch_scatter_blocks(c->stream, c->n, nb, c->d_blk, p->blk_stride, p->n_contacts, p->d_nc,
                                  p->d_blk_off, p->d_inds_off, p->d_inds, -1, c->d_T1);
            launch_zgemm(c->stream, c->n, c->n, c->n, nb, p->d_xi, c->n, 0, c->d_T1, c->n, n2, 0,
                         c->d_T2, c->n, n2);
            launch_zgemm(c->stream, c->n, c->n, c->n, nb, c->d_T2, c->n, n2, p->d_xi, c->n, 0, 0,
                         c->d_T1, c->n, n2);
            launch_assemble(c->stream, c->n, nb, E + m0, c->d_S, c->d_F, c->d_T1, nullptr, 0, 0, nullptr,
                            nullptr, nullptr, nullptr, c->d_A);
        } else {
            launch_assemble(c->stream, c->n, nb, E + m0, c->d_S, c->d_F, nullptr, c->d_blk, p->blk_stride,
                            p->n_contacts, p->d_nc, p->d_blk_off, p->d_inds_off, p->d_inds, c->d_A);
        }
        break;
    default:
        return NEGF_EINVAL;
    }
    return NEGF_OK;
}

// G(E) for the batch [m0, m0 + nb): assemble + inverse, leaving c->G / c->W1 / c->W2 set.  Small systems take the
// fused kernel in STORE mode (one launch, the matrix stays on the CU; k_small_fused.hip).
int run_assemble_inverse(negf_ctx* c, SigmaProvider* p, int m0, int nb, const cplx* E)
{
    int rc;
    if (small_path(c, p)) {
        if (p->kind == SK_CHAIN1D || p->kind == SK_BETHE) {
            if ((rc = run_sigma_blocks(c, p, nb, E + m0, c->d_iters + (size_t)m0 * p->n_contacts,
                                       c->d_conv + (size_t)m0 * p->n_contacts, m0))) return rc;
        }
        ProfScope ps(c, "small");
        SmallFusedArgs a = small_args(c, p, m0, nb, E);
        a.Gout = c->d_A; a.g_stride = (size_t)c->n * c->n;
        launch_small_fused(c->stream, a);
        negf_count_flops(8.0 * c->n * (double)c->n * c->n * nb, 0.0);
        NEGF_HIP_CHECK(hipGetLastError());
        c->G = c->d_A; c->W1 = c->d_T1; c->W2 = c->d_T2;
        return NEGF_OK;
    }
    if ((rc = run_assemble(c, p, m0, nb, E))) return rc;
    return run_inverse(c, nb, c->d_info + m0);
}

// dense Gamma_b = i (Sigma_c - Sigma_c^H) for batch [m0, m0+nb) into `out`
// (stride n*n); returns the batch stride to use (0 when one matrix serves all).
int run_gamma(negf_ctx* c, SigmaProvider* p, int contact /* -1 total */, int m0, int nb, cplx* out,
              cplx* scratch, const cplx** gptr, size_t* stride_out)
{
    const size_t n2 = (size_t)c->n * c->n;
    *gptr = out;
    if (p->kind == SK_PRECOMPUTED && p->pre_is_gamma) {
        // the caller supplied the coupling matrices themselves (transport.py:150-181 take
        // gamma1/gamma2 as arguments): use them in place
        if (contact < 0 || !p->d_pre_c) return NEGF_EINVAL;
        *gptr = p->d_pre_c + n2 * ((size_t)m0 * p->pre_nc + contact);
        *stride_out = n2 * p->pre_nc;
        return NEGF_OK;
    }
    switch (p->kind) {
    case SK_CONST: {
        const cplx* src = contact < 0 ? p->d_const_tot : p->d_const_c + n2 * contact;
        launch_gamma_dense(c->stream, c->n, 1, src, 0, out);
        *stride_out = 0;
        return NEGF_OK;
    }
    case SK_PRECOMPUTED: {
        if (contact < 0 || !p->d_pre_c) {
            launch_gamma_dense(c->stream, c->n, nb, p->d_pre_tot + n2 * m0, n2, out);
        } else {
            const int pn = std::max(p->pre_nc, 1);
            launch_gamma_dense(c->stream, c->n, nb, p->d_pre_c + n2 * ((size_t)m0 * pn + contact),
                               n2 * pn, out);
        }
        *stride_out = n2;
        return NEGF_OK;
    }
    case SK_CHAIN1D:
    case SK_BETHE: {
        // c->d_blk holds the blocks of this batch (run_assemble was just called)
        launch_scatter_blocks(c->stream, c->n, nb, c->d_blk, p->blk_stride, p->n_contacts, p->d_nc,
                              p->d_blk_off, p->d_inds_off, p->d_inds, contact, out);
        if (p->d_xi) {
            // Xi sig Xi through the scratch area
            launch_zgemm(c->stream, c->n, c->n, c->n, nb, p->d_xi, c->n, 0, out, c->n, n2, 0, scratch,
                         c->n, n2);
            launch_zgemm(c->stream, c->n, c->n, c->n, nb, scratch, c->n, n2, p->d_xi, c->n, 0, 0, out,
                         c->n, n2);
        }
        // the gamma kernel reads s[t] and s[transpose]: not in place -> via scratch
        launch_gamma_dense(c->stream, c->n, nb, out, n2, scratch);
        (void)hipMemcpyAsync(out, scratch, n2 * nb * sizeof(cplx), hipMemcpyDeviceToDevice, c->stream);
        *stride_out = n2;
        return NEGF_OK;
    }
    }
    return NEGF_EINVAL;
}

// ---- compact coupling matrices (see k_elementwise.hip): usable when Gamma_c is confined to the
// contact's index list and all lists together cover at most half of the orbitals
bool compact_available(const negf_ctx* c, const SigmaProvider* p)
{
    if (c->gamma_algo == 1) return false;
    if (p->kind == SK_CONST) return p->compact_ok;
    if ((p->kind == SK_CHAIN1D || p->kind == SK_BETHE) && !p->d_xi) {
        int tot = 0;
        for (int k : p->nc) tot += k;
        return 2 * tot <= c->n;
    }
    return false;
}

struct GammaSmall { const cplx* mat; size_t stride; const int* idx; int K; };

// small Gamma of `contact` (or of all contacts, block diagonal, for contact < 0) for the batch
// [m0, m0+nb) into slot `slot` (0/1) of c->d_gsmall
int run_gamma_small(negf_ctx* c, SigmaProvider* p, int contact, int nb, int slot, GammaSmall* g)
{
    const int c0 = contact < 0 ? 0 : contact, c1 = contact < 0 ? p->n_contacts : contact + 1;
    int K = 0;
    for (int k = c0; k < c1; ++k) K += p->nc[k];
    int Kmax = 0;
    for (int k : p->nc) Kmax += k;
    const size_t need = (size_t)2 * nb * Kmax * Kmax;
    if (need > c->gsmall_cap) {
        NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
        dev_free(c->d_gsmall); c->gsmall_cap = 0;
        int rc = dev_alloc(&c->d_gsmall, need);
        if (rc) return rc;
        c->gsmall_cap = need;
    }
    cplx* out = c->d_gsmall + (size_t)slot * nb * Kmax * Kmax;
    const bool constant = p->kind == SK_CONST;
    launch_gamma_small(c->stream, K, c0, c1, constant ? 1 : nb, p->d_nc, p->d_blk_off, p->d_inds_off,
                       constant ? p->d_const_blk : c->d_blk, constant ? 0 : (size_t)p->blk_stride, out,
                       (size_t)K * K);
    g->mat = out; g->stride = constant ? 0 : (size_t)K * K; g->idx = p->d_inds + p->inds_off[c0]; g->K = K;
    return NEGF_OK;
}

int check_ready(negf_ctx* c, SigmaProvider* p, int m)
{
    if (!c || c->n <= 0) return NEGF_ESTATE;
    if (!p) return NEGF_EINVAL;
    if (m < 0) return NEGF_EINVAL;
    if (p->kind == SK_PRECOMPUTED && m > p->m_pre) return NEGF_EINVAL;
    return NEGF_OK;
}

int reduce_info(negf_ctx* c, int m, int* info_host)
{
    if (m == 0) return NEGF_OK;
    std::vector<int> tmp;
    int* dst = info_host;
    if (!dst) { tmp.resize(m); dst = tmp.data(); }
    int rc = download(c, dst, c->d_info, (size_t)m);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) if (dst[i] != 0) return NEGF_ESINGULAR;
    return NEGF_OK;
}

}  // namespace

// =========================================================================== //
extern "C" {

int negf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

const char* negf_version(void) { return "gaunegf_amd-0.1 (gfx950)"; }

const char* negf_strerror(int code)
{
    switch (code) {
    case NEGF_OK: return "ok";
    case NEGF_EINVAL: return "invalid argument";
    case NEGF_ENOMEM: return "out of device memory";
    case NEGF_EHIP: return "HIP runtime error";
    case NEGF_ENODEV: return "no HIP device available (this library has no CPU fallback)";
    case NEGF_ESTATE: return "call negf_set_system first";
    case NEGF_ESINGULAR: return "exactly singular matrix at one or more energies (see info[])";
    default: return "unknown error";
    }
}

int negf_create(negf_ctx** out, int device)
{
    if (!out) return NEGF_EINVAL;
    *out = nullptr;
    const int nd = negf_device_count();
    if (nd <= 0) return NEGF_ENODEV;
    if (device < 0 || device >= nd) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(device));
    negf_ctx* c = new (std::nothrow) negf_ctx();
    if (!c) return NEGF_ENOMEM;
    c->device = device;
    if (const char* e = getenv("NEGF_CHAIN_CACHE")) c->gcache_max = std::min(std::max(atoi(e), 0), 4096);   // default of negf_set_chain_cache
    c->gcache.reserve((size_t)c->gcache_max);
    live_contexts().push_back(c);
    *out = c;
    return NEGF_OK;
}

void negf_destroy(negf_ctx* c)
{
    if (!c) return;
    { auto& v = live_contexts(); v.erase(std::remove(v.begin(), v.end(), c), v.end()); }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto* p : c->providers) free_provider(p);
    free_workspace(c); free_mbuffers(c); free_gcache(c);
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_cap = 0; }
    if (c->gj_side.ok) {
        (void)hipEventDestroy(c->gj_side.fork);
        for (int g = 0; g < GjSideStreams::MAXG - 1; ++g) { (void)hipStreamDestroy(c->gj_side.s[g]); (void)hipEventDestroy(c->gj_side.join[g]); }
        c->gj_side.ok = false;
    }
    for (auto& sl : c->sys) { dev_free(sl.dF); dev_free(sl.dS); }
    c->d_F = c->d_S = nullptr;
    dev_free(c->d_acc); dev_free(c->d_seg_out); dev_free(c->d_ref_P); dev_free(c->d_ref_meta);
    prof_resolve(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

int negf_set_stream(negf_ctx* c, void* s)
{
    if (!c) return NEGF_EINVAL;
    c->stream = reinterpret_cast<hipStream_t>(s);
    return NEGF_OK;
}

int negf_set_batch(negf_ctx* c, int batch)
{
    if (!c || batch < 0) return NEGF_EINVAL;
    c->batch_user = batch;
    return NEGF_OK;
}
int negf_get_batch(negf_ctx* c) { return c ? c->batch : 0; }

int negf_set_inverse_algo(negf_ctx* c, int algo)
{
    if (!c || algo < 0 || algo > 4) return NEGF_EINVAL;
    c->inverse_algo = algo;
    return NEGF_OK;
}

int negf_set_small_algo(negf_ctx* c, int algo)
{
    if (!c || algo < 0 || algo > 1) return NEGF_EINVAL;
    c->small_algo = algo;
    return NEGF_OK;
}

int negf_set_chain_round_robin(negf_ctx* c, int quantum, int slots)
{
    if (!c || slots < 0) return NEGF_EINVAL;
    c->chain_rr_quantum = quantum < 0 ? -1 : quantum;
    c->chain_rr_slots = slots;
    return NEGF_OK;
}

int negf_set_gamma_algo(negf_ctx* c, int algo)
{
    if (!c || algo < 0 || algo > 1) return NEGF_EINVAL;
    c->gamma_algo = algo;
    return NEGF_OK;
}

// bitwise comparison of two host buffers; large ones in parallel chunks (a 2 x 10 MB comparison per integral is 2 ms of the
// 11 ms a per-GPU share of BASELINE C4 leaves an entry point; a different system differs within the first bytes: the
// first chunk is compared alone before the threads are started)
static bool same_bytes(const void* a, const void* b, size_t bytes)
{
    const size_t head = std::min<size_t>(bytes, (size_t)1 << 16);
    if (std::memcmp(a, b, head) != 0) return false;
    if (bytes <= ((size_t)1 << 21)) return std::memcmp(a, b, bytes) == 0;
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const size_t chunk = (bytes + hw - 1) / hw;
    std::atomic<bool> same{true};
    std::vector<std::thread> th;
    try {
        for (unsigned t = 1; t < hw; ++t) {
            const size_t lo = std::min(bytes, t * chunk), hi = std::min(bytes, lo + chunk);
            if (hi <= lo) break;
            th.emplace_back([&same, a, b, lo, hi] { if (std::memcmp((const char*)a + lo, (const char*)b + lo, hi - lo) != 0) same = false; });
        }
    } catch (...) {                                              // (no more threads: the rest is compared here)
    }
    const size_t mine_hi = std::min(bytes, chunk);
    if (std::memcmp(a, b, mine_hi) != 0) same = false;
    if (th.size() + 1 < hw) {                                    // chunks nobody took
        const size_t lo = std::min(bytes, (th.size() + 1) * chunk);
        if (lo < bytes && std::memcmp((const char*)a + lo, (const char*)b + lo, bytes - lo) != 0) same = false;
    }
    for (auto& x : th) x.join();
    return same;
}

// host copy of a result out of the pinned staging buffer; results of several MB (an 800 x 800 matrix is 10 MB: 1.3 ms of a
// 14-ms entry point on one core) in parallel chunks
static void copy_bytes(void* dst, const void* src, size_t bytes)
{
    if (bytes < ((size_t)1 << 22)) { std::memcpy(dst, src, bytes); return; }
    constexpr unsigned NT = 8;
    const size_t chunk = (((bytes + NT - 1) / NT) + 63) / 64 * 64;
    std::vector<std::thread> th;
    size_t done_to = std::min(bytes, chunk);                     // [0, chunk) is this thread's
    try {
        for (unsigned t = 1; t < NT; ++t) {
            const size_t lo = std::min(bytes, t * chunk), hi = std::min(bytes, lo + chunk);
            if (hi <= lo) break;
            th.emplace_back([dst, src, lo, hi] { std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
            done_to = hi;
        }
    } catch (...) {                                              // (no more threads: the rest is copied here)
    }
    std::memcpy(dst, src, std::min(bytes, chunk));
    if (done_to < bytes) std::memcpy((char*)dst + done_to, (const char*)src + done_to, bytes - done_to);
    for (auto& x : th) x.join();
}

// 64-bit checksum of a host buffer (change detection for the front end's caches, not cryptography): four independent
// multiply-xorshift lanes per chunk, large buffers in parallel chunks, the chunk values folded in order
static unsigned long long hash_chunk(const unsigned char* p, size_t bytes)
{
    const unsigned long long K = 0x9E3779B97F4A7C15ull;
    unsigned long long h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        unsigned long long x[4];
        std::memcpy(x, p + i, 32);
        for (int l = 0; l < 4; ++l) { h[l] = (h[l] ^ x[l]) * K; h[l] ^= h[l] >> 32; }
    }
    unsigned long long tail[4] = {0, 0, 0, 0};
    if (i < bytes) {
        std::memcpy(tail, p + i, bytes - i);
        for (int l = 0; l < 4; ++l) { h[l] = (h[l] ^ tail[l]) * K; h[l] ^= h[l] >> 32; }
    }
    unsigned long long r = bytes * K;
    for (int l = 0; l < 4; ++l) { r = (r ^ h[l]) * K; r ^= r >> 29; }
    return r;
}

unsigned long long negf_hash_bytes(const void* data, unsigned long long bytes)
{
    if (!data || bytes == 0) return 0x9E3779B97F4A7C15ull;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    if (bytes < ((size_t)1 << 21)) return hash_chunk(p, (size_t)bytes);
    constexpr unsigned NT = 8;
    const size_t chunk = ((((size_t)bytes + NT - 1) / NT) + 31) / 32 * 32;
    unsigned long long part[NT] = {0};
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        for (unsigned t = 1; t < NT; ++t) {
            const size_t lo = std::min<size_t>(bytes, t * chunk), hi = std::min<size_t>(bytes, lo + chunk);
            if (hi <= lo) break;
            th.emplace_back([&part, p, lo, hi, t] { part[t] = hash_chunk(p + lo, hi - lo); });
            started = t + 1;
        }
    } catch (...) {                                              // (no more threads: the rest is hashed here)
    }
    part[0] = hash_chunk(p, std::min<size_t>(bytes, chunk));
    for (unsigned t = started; t < NT; ++t) {                    // chunks nobody took
        const size_t lo = std::min<size_t>(bytes, t * chunk), hi = std::min<size_t>(bytes, lo + chunk);
        if (hi > lo) part[t] = hash_chunk(p + lo, hi - lo);
    }
    for (auto& x : th) x.join();
    return hash_chunk(reinterpret_cast<const unsigned char*>(part), sizeof(part)) ^ bytes;
}

int negf_set_system(negf_ctx* c, int n, const double* F, const double* S)
{
    return negf_set_system_keyed(c, n, F, S, 0ull);
}

int negf_set_system_keyed(negf_ctx* c, int n, const double* F, const double* S, unsigned long long key)
{
    if (!c || n <= 0 || !F || !S) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    const size_t n2 = (size_t)n * n;
    int rc;
    if (n != c->n) {
        // providers are tied to the matrix dimension
        for (auto*& p : c->providers) { free_provider(p); p = nullptr; }
        free_workspace(c);
        for (auto& sl : c->sys) { dev_free(sl.dF); dev_free(sl.dS); sl.valid = false; sl.hF.clear(); sl.hS.clear(); }
        c->d_F = c->d_S = nullptr; c->sys_cur = -1;
        dev_free(c->d_acc);
        c->n = n;
        if ((rc = dev_alloc(&c->d_acc, n2))) return rc;
    }
    const cplx* Fh = reinterpret_cast<const cplx*>(F);
    const cplx* Sh = reinterpret_cast<const cplx*>(S);
    // one of the systems already on the device?  (bitwise comparison with the host copies; matrices above 256 MB
    // -- n > 4096 -- keep one system only: a slot costs 2 x 16 n^2 bytes of HBM and of host memory)
    const int nslots = n2 * sizeof(cplx) <= ((size_t)256 << 20) ? negf_ctx::NEGF_SYS_SLOTS : 1;
    int slot = -1;
    // (a caller's key: equal keys vouch for bitwise-equal contents -- the front end's private, immutable copies -- and
    //  select a resident system without the 2 x 16 n^2 bytes of comparison)
    if (key)
        for (int k = 0; k < nslots && slot < 0; ++k)
            if (c->sys[k].valid && c->sys[k].hF.size() == n2 && c->sys[k].key == key) slot = k;
    for (int k = 0; k < nslots && slot < 0; ++k) {
        auto& sl = c->sys[k];
        if (sl.valid && sl.hF.size() == n2 && same_bytes(sl.hF.data(), Fh, n2 * sizeof(cplx)) &&
            same_bytes(sl.hS.data(), Sh, n2 * sizeof(cplx))) { slot = k; sl.key = key; }
    }
    if (slot >= 0 && slot == c->sys_cur) return NEGF_OK;             // resident: nothing to do
    if (slot < 0) {
        // least recently used slot (an empty one first)
        slot = 0;
        for (int k = 1; k < nslots; ++k)
            if (!c->sys[k].valid ? c->sys[slot].valid : (c->sys[slot].valid && c->sys[k].used < c->sys[slot].used)) slot = k;
        auto& sl = c->sys[slot];
        sl.valid = false;
        if (!sl.dF && (rc = dev_alloc(&sl.dF, n2))) return rc;
        if (!sl.dS && (rc = dev_alloc(&sl.dS, n2))) return rc;
        if ((rc = upload(c, sl.dF, Fh, n2))) return rc;
        if ((rc = upload(c, sl.dS, Sh, n2))) return rc;
        sl.hF.assign(Fh, Fh + n2);
        sl.hS.assign(Sh, Sh + n2);
        sl.key = key;
        sl.valid = true;
    }
    c->sys[slot].used = ++c->sys_clock;
    c->sys_cur = slot;
    c->d_F = c->sys[slot].dF;
    c->d_S = c->sys[slot].dS;
    // constant providers cache F + Sigma_tot: refresh them for the resident F (on the device, same rounding as
    // the host sum at creation: one IEEE addition per component)
    for (auto* p : c->providers)
        if (p && p->kind == SK_CONST) launch_cadd(c->stream, n2, c->d_F, p->d_const_tot, p->d_hbase);
    NEGF_HIP_CHECK(hipGetLastError());
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return NEGF_OK;
}

// ------------------------------------------------------------------ providers
static int setup_blocks(negf_ctx* c, SigmaProvider* p, int n_contacts, const int* nc, const int* inds);

int negf_sigma_const(negf_ctx* c, int n_contacts, const double* sigma, int* handle)
{
    if (!c || c->n <= 0) return NEGF_ESTATE;
    if (n_contacts <= 0 || !sigma || !handle) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    SigmaProvider* p = new SigmaProvider();
    p->kind = SK_CONST; p->n_contacts = n_contacts;
    int rc;
    if ((rc = dev_alloc(&p->d_const_c, n2 * n_contacts)) || (rc = dev_alloc(&p->d_const_tot, n2)) ||
        (rc = dev_alloc(&p->d_hbase, n2))) { free_provider(p); return rc; }
    const cplx* s = reinterpret_cast<const cplx*>(sigma);
    // total = sum over contacts in contact order (surfGTester.py:128-131); F + total
    // is a host-side O(n^2) setup step, the per-energy work stays on the GPU
    std::vector<cplx> tot(n2, cmake(0.0, 0.0)), hb(n2), Fh(n2);
    for (int k = 0; k < n_contacts; ++k)
        for (size_t i = 0; i < n2; ++i) tot[i] = cadd(tot[i], s[k * n2 + i]);
    if ((rc = download(c, Fh.data(), c->d_F, n2))) { free_provider(p); return rc; }
    for (size_t i = 0; i < n2; ++i) hb[i] = cadd(Fh[i], tot[i]);
    if ((rc = upload(c, p->d_const_c, s, n2 * n_contacts)) || (rc = upload(c, p->d_const_tot, tot.data(), n2)) ||
        (rc = upload(c, p->d_hbase, hb.data(), n2))) { free_provider(p); return rc; }
    // support of each contact matrix (rows or columns holding a nonzero): formSigma-style contacts
    // (matTools.py:90-120) only touch their own orbitals, so Gamma_c is a small block
    {
        const int n = c->n;
        std::vector<int> ks(n_contacts), inds;
        for (int k = 0; k < n_contacts; ++k) {
            const cplx* m = s + (size_t)k * n2;
            std::vector<char> used(n, 0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (m[(size_t)i * n + j].x != 0.0 || m[(size_t)i * n + j].y != 0.0) { used[i] = 1; used[j] = 1; }
            int cnt = 0;
            for (int i = 0; i < n; ++i) if (used[i]) { inds.push_back(i); ++cnt; }
            ks[k] = cnt;
        }
        bool ok = 2 * (int)inds.size() <= n;
        for (int k : ks) ok = ok && k > 0;
        if (ok) {
            if ((rc = setup_blocks(c, p, n_contacts, ks.data(), inds.data()))) { free_provider(p); return rc; }
            std::vector<cplx> blk((size_t)p->blk_stride);
            for (int k = 0; k < n_contacts; ++k) {
                const cplx* m = s + (size_t)k * n2;
                const int* id = p->h_inds.data() + p->inds_off[k];
                for (int a = 0; a < ks[k]; ++a)
                    for (int e = 0; e < ks[k]; ++e)
                        blk[(size_t)p->blk_off[k] + (size_t)a * ks[k] + e] = m[(size_t)id[a] * n + id[e]];
            }
            if ((rc = dev_alloc(&p->d_const_blk, blk.size())) || (rc = upload(c, p->d_const_blk, blk.data(), blk.size()))) {
                free_provider(p); return rc;
            }
            p->compact_ok = true;
        }
    }
    *handle = add_provider(c, p);
    return NEGF_OK;
}

static int setup_blocks(negf_ctx* c, SigmaProvider* p, int n_contacts, const int* nc, const int* inds)
{
    p->n_contacts = n_contacts;
    p->nc.assign(nc, nc + n_contacts);
    p->blk_off.resize(n_contacts); p->inds_off.resize(n_contacts);
    int off = 0, ioff = 0;
    p->nc_max = 0;
    for (int k = 0; k < n_contacts; ++k) {
        if (nc[k] <= 0 || nc[k] > c->n) return NEGF_EINVAL;
        p->blk_off[k] = off; p->inds_off[k] = ioff;
        off += nc[k] * nc[k]; ioff += nc[k];
        p->nc_max = std::max(p->nc_max, nc[k]);
    }
    p->blk_stride = off;
    p->h_inds.assign(inds, inds + ioff);
    for (int v : p->h_inds) if (v < 0 || v >= c->n) return NEGF_EINVAL;
    int rc;
    if ((rc = dev_alloc(&p->d_inds, (size_t)ioff)) || (rc = dev_alloc(&p->d_nc, (size_t)n_contacts)) ||
        (rc = dev_alloc(&p->d_blk_off, (size_t)n_contacts)) || (rc = dev_alloc(&p->d_inds_off, (size_t)n_contacts)))
        return rc;
    if ((rc = upload(c, p->d_inds, p->h_inds.data(), (size_t)ioff)) ||
        (rc = upload(c, p->d_nc, p->nc.data(), (size_t)n_contacts)) ||
        (rc = upload(c, p->d_blk_off, p->blk_off.data(), (size_t)n_contacts)) ||
        (rc = upload(c, p->d_inds_off, p->inds_off.data(), (size_t)n_contacts)))
        return rc;
    // position of every orbital in each contact's index list (the small fused kernel subtracts the contact blocks while
    // it assembles); a list that names an orbital twice has no such map and keeps the scatter kernels
    std::vector<int> pos((size_t)n_contacts * c->n, -1);
    bool unique = true;
    for (int k = 0; k < n_contacts; ++k)
        for (int a = 0; a < nc[k]; ++a) {
            int& slot = pos[(size_t)k * c->n + p->h_inds[p->inds_off[k] + a]];
            if (slot >= 0) unique = false;
            slot = a;
        }
    if (unique) {
        if ((rc = dev_alloc(&p->d_pos, pos.size())) || (rc = upload(c, p->d_pos, pos.data(), pos.size()))) return rc;
    }
    return NEGF_OK;
}

int negf_sigma_chain1d(negf_ctx* c, int n_contacts, const int* nc, const int* inds,
                       const double* alpha, const double* Salpha, const double* beta,
                       const double* Sbeta, const double* tau, const double* Stau,
                       double eta, double conv, double relFactor, int max_iter, int force_iters,
                       int* handle)
{
    if (!c || c->n <= 0) return NEGF_ESTATE;
    if (n_contacts <= 0 || !nc || !inds || !alpha || !Salpha || !beta || !Sbeta || !tau || !Stau || !handle)
        return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    SigmaProvider* p = new SigmaProvider();
    p->kind = SK_CHAIN1D;
    int rc = setup_blocks(c, p, n_contacts, nc, inds);
    if (rc) { free_provider(p); return rc; }
    const size_t tot = (size_t)p->blk_stride;
    cplx** dsts[6] = {&p->d_alpha, &p->d_Salpha, &p->d_beta, &p->d_Sbeta, &p->d_tau, &p->d_Stau};
    const double* srcs[6] = {alpha, Salpha, beta, Sbeta, tau, Stau};
    for (int k = 0; k < 6; ++k) {
        if ((rc = dev_alloc(dsts[k], tot)) ||
            (rc = upload(c, *dsts[k], reinterpret_cast<const cplx*>(srcs[k]), tot))) { free_provider(p); return rc; }
    }
    // what g(E) depends on, for the context's g(E) cache
    p->h_lead = std::make_shared<std::vector<cplx>>();
    p->h_lead->reserve(4 * tot);
    for (int k = 0; k < 4; ++k) {
        const cplx* src = reinterpret_cast<const cplx*>(srcs[k]);
        p->h_lead->insert(p->h_lead->end(), src, src + tot);
    }
    p->lead_hash = hash_words(p->h_lead->data(), p->h_lead->size() * sizeof(cplx));
    p->eta = eta; p->conv = conv; p->relFactor = relFactor; p->max_iter = max_iter;
    p->force_iters = force_iters;
    *handle = add_provider(c, p);
    return NEGF_OK;
}

int negf_sigma_bethe(negf_ctx* c, int n_contacts, const int* n_atoms, const int* atom_orbs,
                     const int* n_nb, const int* nb_dirs, const double* H, const double* Slist,
                     const double* Vlist, const double* xi, double eta, double conv, double mix,
                     int max_iter, int force_iters, int* handle)
{
    if (!c || c->n <= 0) return NEGF_ESTATE;
    if (n_contacts <= 0 || !n_atoms || !atom_orbs || !n_nb || !H || !Slist || !Vlist || !handle)
        return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    SigmaProvider* p = new SigmaProvider();
    p->kind = SK_BETHE;
    std::vector<int> nc(n_contacts);
    int total_atoms = 0;
    p->n_atoms.assign(n_atoms, n_atoms + n_contacts);
    p->atom_off.resize(n_contacts);
    for (int k = 0; k < n_contacts; ++k) {
        if (n_atoms[k] <= 0) { free_provider(p); return NEGF_EINVAL; }
        p->atom_off[k] = total_atoms;
        nc[k] = 9 * n_atoms[k];
        total_atoms += n_atoms[k];
    }
    int rc = setup_blocks(c, p, n_contacts, nc.data(), atom_orbs);
    if (rc) { free_provider(p); return rc; }
    std::vector<int> nb_off(total_atoms + 1, 0);
    for (int a = 0; a < total_atoms; ++a) {
        if (n_nb[a] < 0) { free_provider(p); return NEGF_EINVAL; }
        nb_off[a + 1] = nb_off[a] + n_nb[a];
    }
    const int total_nb = nb_off[total_atoms];
    if (total_nb > 0 && !nb_dirs) { free_provider(p); return NEGF_EINVAL; }
    if ((rc = dev_alloc(&p->d_n_atoms, (size_t)n_contacts)) || (rc = dev_alloc(&p->d_atom_off, (size_t)n_contacts)) ||
        (rc = dev_alloc(&p->d_nb_off, (size_t)total_atoms + 1)) || (rc = dev_alloc(&p->d_nb_dirs, (size_t)std::max(total_nb, 1))) ||
        (rc = dev_alloc(&p->d_H, (size_t)n_contacts * 81)) || (rc = dev_alloc(&p->d_Slist, (size_t)n_contacts * 12 * 81)) ||
        (rc = dev_alloc(&p->d_Vlist, (size_t)n_contacts * 12 * 81))) { free_provider(p); return rc; }
    if ((rc = upload(c, p->d_n_atoms, p->n_atoms.data(), (size_t)n_contacts)) ||
        (rc = upload(c, p->d_atom_off, p->atom_off.data(), (size_t)n_contacts)) ||
        (rc = upload(c, p->d_nb_off, nb_off.data(), (size_t)total_atoms + 1)) ||
        (rc = upload(c, p->d_nb_dirs, nb_dirs, (size_t)total_nb)) ||
        (rc = upload(c, p->d_H, H, (size_t)n_contacts * 81)) ||
        (rc = upload(c, p->d_Slist, Slist, (size_t)n_contacts * 12 * 81)) ||
        (rc = upload(c, p->d_Vlist, Vlist, (size_t)n_contacts * 12 * 81))) { free_provider(p); return rc; }
    if (xi) {
        const size_t n2 = (size_t)c->n * c->n;
        if ((rc = dev_alloc(&p->d_xi, n2)) ||
            (rc = upload(c, p->d_xi, reinterpret_cast<const cplx*>(xi), n2))) { free_provider(p); return rc; }
    }
    p->eta = eta; p->conv = conv; p->mix = mix; p->max_iter = max_iter; p->force_iters = force_iters;
    *handle = add_provider(c, p);
    return NEGF_OK;
}

int negf_sigma_precomputed(negf_ctx* c, int m, const double* sigma_tot, int n_contacts_c,
                           const double* sigma_c, int* handle)
{
    if (!c || c->n <= 0) return NEGF_ESTATE;
    if (m <= 0 || !sigma_tot || !handle) return NEGF_EINVAL;
    const bool is_gamma = n_contacts_c < 0;
    if (is_gamma) { n_contacts_c = -n_contacts_c; if (!sigma_c) return NEGF_EINVAL; }
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    SigmaProvider* p = new SigmaProvider();
    p->kind = SK_PRECOMPUTED; p->m_pre = m; p->n_contacts = std::max(n_contacts_c, 1);
    p->pre_nc = sigma_c ? std::max(n_contacts_c, 1) : 0;
    p->pre_is_gamma = is_gamma;
    int rc;
    if ((rc = dev_alloc(&p->d_pre_tot, n2 * m)) ||
        (rc = upload(c, p->d_pre_tot, reinterpret_cast<const cplx*>(sigma_tot), n2 * m))) { free_provider(p); return rc; }
    if (sigma_c) {
        const size_t cnt = n2 * m * p->pre_nc;
        if ((rc = dev_alloc(&p->d_pre_c, cnt)) ||
            (rc = upload(c, p->d_pre_c, reinterpret_cast<const cplx*>(sigma_c), cnt))) { free_provider(p); return rc; }
    }
    *handle = add_provider(c, p);
    return NEGF_OK;
}

int negf_sigma_free(negf_ctx* c, int handle)
{
    SigmaProvider* p = get_provider(c, handle);
    if (!p) return NEGF_EINVAL;
    (void)hipStreamSynchronize(c->stream);
    free_provider(p);
    c->providers[handle] = nullptr;
    return NEGF_OK;
}

// ------------------------------------------------------------- device variants
int negf_gr_int_dev(negf_ctx* c, int handle, int m, const double* E_dev, const double* w_dev,
                    double* out_dev)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out_dev || (m > 0 && (!E_dev || !w_dev))) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    if ((rc = ensure_mbuffers(c, m, p->n_contacts))) return rc;
    if (small_path(c, p) && m > 0) {
        // n <= 96: assemble + inverse + weighted sum in one kernel (plus the reduction of the workgroups' partial sums);
        // no n x n work area in HBM at all.  Grids above SMALL_CHUNK points go through in chunks (the Sigma blocks of a
        // block provider are staged per chunk); chunk sums are added in order.
        const cplx* E = reinterpret_cast<const cplx*>(E_dev);
        const cplx* w = reinterpret_cast<const cplx*>(w_dev);
        cplx* out = reinterpret_cast<cplx*>(out_dev);
        constexpr int SMALL_CHUNK = 16384;
        const bool blocks = p->kind == SK_CHAIN1D || p->kind == SK_BETHE;
        const int chunk = std::min(m, SMALL_CHUNK);
        if (blocks && (rc = ensure_blk(c, (size_t)chunk * p->blk_stride))) return rc;
        const size_t part_need = (size_t)small_fused_grid(c->n, chunk) * n2 + (m > chunk ? n2 : 0);
        if (part_need > c->small_part_cap) {
            NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
            dev_free(c->d_small_part); c->small_part_cap = 0;
            if ((rc = dev_alloc(&c->d_small_part, part_need))) return rc;
            c->small_part_cap = part_need;
        }
        cplx* chunk_sum = c->d_small_part + (size_t)small_fused_grid(c->n, chunk) * n2;   // (only with several chunks)
        for (int m0 = 0; m0 < m; m0 += chunk) {
            const int nb = std::min(chunk, m - m0);
            if (blocks && (rc = run_sigma_blocks(c, p, nb, E + m0, c->d_iters + (size_t)m0 * p->n_contacts,
                                                 c->d_conv + (size_t)m0 * p->n_contacts, m0))) return rc;
            ProfScope ps(c, "small");
            SmallFusedArgs a = small_args(c, p, m0, nb, E);
            a.w = w + m0; a.partial = c->d_small_part; a.out = m0 == 0 ? out : chunk_sum;
            launch_small_fused(c->stream, a);
            if (m0 > 0) launch_cadd(c->stream, n2, out, chunk_sum, out);
            negf_count_flops(8.0 * c->n * (double)c->n * c->n * nb, 0.0);
        }
        c->last_m = m;
        NEGF_HIP_CHECK(hipGetLastError());
        return NEGF_OK;
    }
    if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
    const cplx* E = reinterpret_cast<const cplx*>(E_dev);
    const cplx* w = reinterpret_cast<const cplx*>(w_dev);
    cplx* out = reinterpret_cast<cplx*>(out_dev);
    NEGF_HIP_CHECK(hipMemsetAsync(out, 0, n2 * sizeof(cplx), c->stream));
    for (int m0 = 0; m0 < m; m0 += c->batch) {
        const int nb = std::min(c->batch, m - m0);
        // the weighted sum needs no G: the windowed inverse leaves its gather out and the sum reads the reduced matrices
        // through the pivot bookkeeping (NEGF_GATHER_FUSED=0: the gather + accumulate sequence)
        static int fused = -1;
        if (fused < 0) { const char* e = getenv("NEGF_GATHER_FUSED"); fused = e ? atoi(e) : 1; }
        c->defer_gather = fused != 0;
        rc = run_assemble_inverse(c, p, m0, nb, E);
        c->defer_gather = false;
        if (rc) return rc;
        ProfScope ps(c, "accumulate");
        if (c->G_deferred) launch_accumulate_perm(c->stream, c->n, nb, w + m0, c->W1, c->d_ipiv, c->d_info + m0, out, c->W2);
        else launch_accumulate(c->stream, (int)n2, nb, w + m0, c->G, out, c->W2);
        c->G_deferred = false;
    }
    c->last_m = m;
    NEGF_HIP_CHECK(hipGetLastError());
    return NEGF_OK;
}

// sum_m w_m G Gamma_c G^H over the energies E[0..m) (device pointers).  nseg == 0: one sum into out [n*n]; nseg > 0: the
// energies are nseg consecutive segments ending at seg_end[s] (host array) and out [nseg][n*n] receives one sum each.
static int gless_core(negf_ctx* c, SigmaProvider* p, int contact, int m, const cplx* E, const cplx* w, cplx* out,
                      int nseg, const int* seg_end)
{
    int rc;
    const size_t n2 = (size_t)c->n * c->n;
    const int n = c->n;
    if ((rc = ensure_mbuffers(c, m, p->n_contacts))) return rc;
    if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
    NEGF_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)std::max(nseg, 1) * n2 * sizeof(cplx), c->stream));
    auto accumulate = [&](const cplx* X, int m0, int nb) {
        ProfScope ps(c, "accumulate");
        if (nseg == 0) { launch_accumulate(c->stream, (int)n2, nb, w + m0, X, out, c->W2); return; }
        for (int sg = 0, start = 0; sg < nseg; start = seg_end[sg], ++sg) {
            const int lo = std::max(start, m0), hi = std::min(seg_end[sg], m0 + nb);
            if (hi > lo) launch_accumulate(c->stream, (int)n2, hi - lo, w + lo, X + (size_t)(lo - m0) * n2, out + (size_t)sg * n2, c->W2);
        }
    };
    for (int m0 = 0; m0 < m; m0 += c->batch) {
        const int nb = std::min(c->batch, m - m0);
        if ((rc = run_assemble_inverse(c, p, m0, nb, E))) return rc;
        if (compact_available(c, p)) {
            // G Gamma G^H = (G[:, I] Gamma_II) G[:, I]^H : Gc and X live in the free buffer W2
            GammaSmall g;
            { ProfScope ps(c, "gamma"); if ((rc = run_gamma_small(c, p, contact, nb, 0, &g))) return rc; }
            cplx* Gc = c->W2;                              // [nb][n x K]
            cplx* X = c->W2 + (size_t)n * g.K;             // [nb][n x K]   (2 n K <= n^2)
            {
                ProfScope ps(c, "zgemm");
                launch_gather_block(c->stream, n, n, g.K, nb, c->G, n2, nullptr, g.idx, Gc, n2);
                // X = (Gc Gamma)^H, K x n, stored conjugate-transposed by the first product; W1 = Gc X: the second operand
                // in the plain form (see the dense products below)
                launch_zgemm(c->stream, n, g.K, g.K, nb, Gc, g.K, n2, g.mat, g.K, g.stride, 4, X, n, n2);
                launch_zgemm(c->stream, n, n, g.K, nb, Gc, g.K, n2, X, n, n2, 2, c->W1, n, n2);     // Hermitian result
            }
            accumulate(c->W1, m0, nb);
            continue;
        }
        size_t gs = 0;
        const cplx* gam = nullptr;
        { ProfScope ps(c, "gamma"); if ((rc = run_gamma(c, p, contact, m0, nb, c->W1, c->W2, &gam, &gs))) return rc; }
        {
            ProfScope ps(c, "zgemm");
            // G Gamma G^H (integrate.py:81).  Gamma = i (Sigma - Sigma^H) is Hermitian, element by element, and so is the
            // result: W2 = (G Gamma)^H -- the first product stores its result conjugate-transposed -- and W1 = G W2
            // ( = G Gamma^H G^H), upper block tiles computed, lower ones mirrored.  Both products read their second
            // operand in the plain form (k_zgemm.hip: the operand conjugate-transposed on the fly costs the L2 twice
            // the requests).
            // (coupling matrices handed in by the caller -- pre_is_gamma -- need not be Hermitian: X = G Gamma, then
            //  the full product X G^H)
            const bool full = p->kind == SK_PRECOMPUTED && p->pre_is_gamma;
            if (!full) {
                launch_zgemm(c->stream, n, n, n, nb, c->G, n, n2, gam, n, gs, 4, c->W2, n, n2);
                launch_zgemm(c->stream, n, n, n, nb, c->G, n, n2, c->W2, n, n2, 2, c->W1, n, n2);
            } else {
                launch_zgemm(c->stream, n, n, n, nb, c->G, n, n2, gam, n, gs, 0, c->W2, n, n2);
                launch_zgemm(c->stream, n, n, n, nb, c->W2, n, n2, c->G, n, n2, 1, c->W1, n, n2);
            }
        }
        accumulate(c->W1, m0, nb);
    }
    c->last_m = m;
    NEGF_HIP_CHECK(hipGetLastError());
    return NEGF_OK;
}

int negf_gless_int_dev(negf_ctx* c, int handle, int ind, int m, const double* E_dev,
                       const double* w_dev, double* out_dev)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out_dev || (m > 0 && (!E_dev || !w_dev))) return NEGF_EINVAL;
    const int contact = norm_contact(p, ind);
    if (contact == -2) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    return gless_core(c, p, contact, m, reinterpret_cast<const cplx*>(E_dev), reinterpret_cast<const cplx*>(w_dev),
                      reinterpret_cast<cplx*>(out_dev), 0, nullptr);
}

int negf_transmission_dev(negf_ctx* c, int handle, int contact_L, int contact_R, int spin_mode,
                          int m, const double* E_dev, double* T_dev, double* Tspin_dev)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!T_dev || (m > 0 && !E_dev)) return NEGF_EINVAL;
    if (spin_mode != NEGF_SPIN_RESTRICTED && spin_mode != NEGF_SPIN_BLOCK) return NEGF_EINVAL;
    if (spin_mode == NEGF_SPIN_BLOCK && (!Tspin_dev || (c->n & 1))) return NEGF_EINVAL;
    const int cL = norm_contact(p, contact_L), cR = norm_contact(p, contact_R);
    if (cL == -2 || cR == -2) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const int n = c->n;
    const size_t n2 = (size_t)n * n;
    if ((rc = ensure_mbuffers(c, m, p->n_contacts))) return rc;
    // two extra n x n work areas per energy for Gamma_L / Gamma_R: reuse T1/T2 for
    // the Gammas and carve the product temporaries out of a second workspace half
    c->transmission_seen = true;
    if ((rc = ensure_workspace(c, m, p->blk_stride, 2))) return rc;
    const cplx* E = reinterpret_cast<const cplx*>(E_dev);
    // process with half the allocated batch so that [0,half) holds this sweep's
    // matrices and [half, 2*half) is free scratch in each of A/T1/T2
    const int half = std::max(1, c->batch / 2);
    for (int m0 = 0; m0 < m; m0 += half) {
        const int nb = std::min(half, m - m0);
        if ((rc = run_assemble_inverse(c, p, m0, nb, E))) return rc;
        if (spin_mode == NEGF_SPIN_RESTRICTED && compact_available(c, p)) {
            // T = Re sum_ij Y_ij conj(G_ij), Y = Gamma_L G Gamma_R: only G[I_L, I_R] enters
            GammaSmall gL, gR;
            {
                ProfScope ps(c, "gamma");
                if ((rc = run_gamma_small(c, p, cL, nb, 0, &gL))) return rc;
                if ((rc = run_gamma_small(c, p, cR, nb, 1, &gR))) return rc;
            }
            const size_t kk = (size_t)gL.K * gR.K;            // 3 kk <= n^2
            cplx* Glr = c->W2; cplx* Xs = c->W2 + kk; cplx* Ys = c->W2 + 2 * kk;
            {
                ProfScope ps(c, "zgemm");
                launch_gather_block(c->stream, n, gL.K, gR.K, nb, c->G, n2, gL.idx, gR.idx, Glr, n2);
                launch_zgemm(c->stream, gL.K, gR.K, gL.K, nb, gL.mat, gL.K, gL.stride, Glr, gR.K, n2, 0, Xs, gR.K, n2);
                launch_zgemm(c->stream, gL.K, gR.K, gR.K, nb, Xs, gR.K, n2, gR.mat, gR.K, gR.stride, 0, Ys, gR.K, n2);
            }
            ProfScope ps(c, "trace");
            launch_trace_dot(c->stream, gL.K, gR.K, nb, Ys, gR.K, n2, Glr, gR.K, n2, T_dev + m0, 1);
            continue;
        }
        cplx* G = c->G;
        const cplx* gamL = nullptr;           // [nb] (or one shared matrix)
        const cplx* gamR = nullptr;
        cplx* X = c->W2;                      // [nb]
        cplx* Y = c->G + n2 * half;           // [nb] upper half of the buffer holding G (unused by this sweep)
        size_t gsL = 0, gsR = 0;
        {
            ProfScope ps(c, "gamma");
            // W2 doubles as run_gamma's scratch -> build both Gammas before X is live
            if ((rc = run_gamma(c, p, cL, m0, nb, c->W1, c->W2, &gamL, &gsL))) return rc;
            if ((rc = run_gamma(c, p, cR, m0, nb, c->W1 + n2 * half, c->W2, &gamR, &gsR))) return rc;
        }
        if (spin_mode == NEGF_SPIN_RESTRICTED) {
            {
                ProfScope ps(c, "zgemm");
                // T = Re Tr[Gamma_L G Gamma_R G^H]  (transport.py:156-157) as  X = (G Gamma_R)^H ;  M = G X -- Hermitian,
                // as Gamma_R is: upper block tiles only (launch_zgemm opB bit 2) -- ;  T = Re sum Gamma_L,ij conj(M_ij)
                // ( = Re Tr[Gamma_L M], M_ji = conj(M_ij)): one and a half dense products instead of two
                // (explicit Gamma matrices handed in by the caller -- negf_sigma_precomputed with gammas, the
                //  reference's _transmission_kernel_restricted(E, F, S, sigma, gamma1, gamma2) -- need not be
                //  Hermitian: M is then computed in full, opB = 1, and T = Re Tr[Gamma_L M] = Re sum Gamma_L,ij M_ji
                //  is taken with M^H: sum Re(Gamma_L,ij conj(M^H_ij)))
                const bool herm_ok = !(p->kind == SK_PRECOMPUTED && p->pre_is_gamma);
                if (herm_ok) {
                    // X = (G Gamma_R)^H stored by the first product, M = G X: both second operands in the plain form
                    launch_zgemm(c->stream, n, n, n, nb, G, n, n2, gamR, n, gsR, 4, X, n, n2);
                    launch_zgemm(c->stream, n, n, n, nb, G, n, n2, X, n, n2, 2, Y, n, n2);           // Y = M = G Gamma_R G^H
                } else {
                    launch_zgemm(c->stream, n, n, n, nb, G, n, n2, gamR, n, gsR, 0, X, n, n2);
                    launch_zgemm(c->stream, n, n, n, nb, G, n, n2, X, n, n2, 1, Y, n, n2);           // Y = G X^H = M^H
                }
            }
            ProfScope ps(c, "trace");
            launch_trace_dot(c->stream, n, n, nb, gamL, n, gsL, Y, n, n2, T_dev + m0, 1);
        } else {
            const int h = n / 2;
            // blocks [uu, ud, du, dd]: G rows/cols offsets; Gamma_L blocks [uu,uu,dd,dd];
            // Gamma_R blocks [uu,dd,uu,dd]; Ga_k = (G^H)[R,C] = conj(G[C,R])^T  (transport.py:166-177)
            const int gr[4] = {0, 0, h, h}, gc[4] = {0, h, 0, h};
            const int l_off[4] = {0, 0, h, h}, r_off[4] = {0, h, 0, h};
            for (int k = 0; k < 4; ++k) {
                const cplx* Gk = G + (size_t)gr[k] * n + gc[k];
                const cplx* GLk = gamL + (size_t)l_off[k] * n + l_off[k];
                const cplx* GRk = gamR + (size_t)r_off[k] * n + r_off[k];
                // trace pairs X_k[i][j] with conj(G[C0+i][R0+j])
                const cplx* Gpair = G + (size_t)gc[k] * n + gr[k];
                {
                    ProfScope ps(c, "zgemm");
                    launch_zgemm(c->stream, h, h, h, nb, GLk, n, gsL, Gk, n, n2, 0, X, h, n2);
                    launch_zgemm(c->stream, h, h, h, nb, X, h, n2, GRk, n, gsR, 0, Y, h, n2);
                }
                ProfScope ps(c, "trace");
                launch_trace_dot(c->stream, h, h, nb, Y, h, n2, Gpair, n, n2, Tspin_dev + (size_t)m0 * 4 + k, 4);
            }
        }
    }
    c->last_m = m;
    NEGF_HIP_CHECK(hipGetLastError());
    return NEGF_OK;
}

int negf_sync(negf_ctx* c)
{
    if (!c) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return NEGF_OK;
}

int negf_last_info(negf_ctx* c, int m, int* info)
{
    if (!c || !info || m < 0 || m > c->m_cap) return NEGF_EINVAL;
    return download(c, info, c->d_info, (size_t)m);
}

int negf_last_iters(negf_ctx* c, int handle, int m, int* iters, int* converged)
{
    SigmaProvider* p = get_provider(c, handle);
    if (!c || !p || m < 0 || m > c->m_cap) return NEGF_EINVAL;
    const size_t cnt = (size_t)m * p->n_contacts;
    if (p->kind == SK_CHAIN1D || p->kind == SK_BETHE) {
        if (p->n_contacts > c->contacts_cap) return NEGF_EINVAL;
        int rc;
        if (iters && (rc = download(c, iters, c->d_iters, cnt))) return rc;
        if (converged && (rc = download(c, converged, c->d_conv, cnt))) return rc;
    } else {
        for (size_t i = 0; i < cnt; ++i) { if (iters) iters[i] = 0; if (converged) converged[i] = 1; }
    }
    return NEGF_OK;
}

// --------------------------------------------------------------- host variants
// The host-pointer entry points stage through ONE pinned buffer of the context: [E | w] go up with two asynchronous
// copies and no synchronisation (the buffer is ours; the call's final synchronisation covers them), the result and the
// per-energy info come down into [out | info] and are handed over after ONE synchronisation.  (Pageable copies with a
// synchronisation each made a 2-point integral of a 60-orbital system cost 0.5 ms of host time: bench.py --config scf.)
static int stage_grid(negf_ctx* c, int m, int contacts, const double* E, const double* w)
{
    int rc = ensure_mbuffers(c, m, contacts);
    if (rc) return rc;
    const size_t gb = (size_t)m * sizeof(cplx);
    const size_t n2b = (size_t)c->n * c->n * sizeof(cplx);
    if ((rc = ensure_pinned(c, 2 * gb + n2b + (size_t)m * sizeof(int) + 64))) return rc;
    if (E && m > 0) {
        c->h_E_valid = false;
        std::memcpy(c->h_pin, E, gb);
        NEGF_HIP_CHECK(hipMemcpyAsync(c->d_E, c->h_pin, gb, hipMemcpyHostToDevice, c->stream));
        const cplx* Eh = reinterpret_cast<const cplx*>(E);
        c->h_E.assign(Eh, Eh + m);
        c->h_E_valid = true;
    }
    if (w && m > 0) {
        std::memcpy(c->h_pin + gb, w, gb);
        NEGF_HIP_CHECK(hipMemcpyAsync(c->d_w, c->h_pin + gb, gb, hipMemcpyHostToDevice, c->stream));
    }
    return NEGF_OK;
}

// result (n x n, from d_src) and per-energy info back to the caller: two asynchronous copies into the pinned buffer,
// one synchronisation; returns NEGF_ESINGULAR when an energy reported a zero pivot
static int fetch_matrix_and_info(negf_ctx* c, int m, const cplx* d_src, double* out_host, int* info_host)
{
    const size_t gb = (size_t)m * sizeof(cplx);
    const size_t n2b = (size_t)c->n * c->n * sizeof(cplx);
    unsigned char* pout = c->h_pin + 2 * gb;
    int* pinfo = reinterpret_cast<int*>(pout + n2b);
    NEGF_HIP_CHECK(hipMemcpyAsync(pout, d_src, n2b, hipMemcpyDeviceToHost, c->stream));
    if (m > 0) NEGF_HIP_CHECK(hipMemcpyAsync(pinfo, c->d_info, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    { const int wrc = wait_stream(c); if (wrc) return wrc; }
    copy_bytes(out_host, pout, n2b);
    int rc = NEGF_OK;
    for (int i = 0; i < m; ++i) { if (info_host) info_host[i] = pinfo[i]; if (pinfo[i] != 0) rc = NEGF_ESINGULAR; }
    return rc;
}

int negf_gr_int(negf_ctx* c, int handle, int m, const double* E, const double* w, double* out,
                int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out || (m > 0 && (!E || !w))) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = stage_grid(c, m, p->n_contacts, E, w))) return rc;
    if ((rc = negf_gr_int_dev(c, handle, m, reinterpret_cast<double*>(c->d_E),
                              reinterpret_cast<double*>(c->d_w), reinterpret_cast<double*>(c->d_acc)))) return rc;
    return fetch_matrix_and_info(c, m, c->d_acc, out, info);
}

// Several integrals of the same system in ONE pass: the energies are nseg consecutive segments (seg_end[s] = index one
// past the last point of segment s) and out receives one n x n sum per segment.  This is how the adaptive integrations
// (nested ANT levels 2, 6, 18, 54 ... of density.py:211-273; the doubling real-axis grids of :438-484) are served: the
// levels a refinement is going to visit are evaluated together -- a level of 2 ... 36 points alone in a launch is pure
// latency on this chip -- and handed back level by level.
static int check_segments(int m, int nseg, const int* seg_end)
{
    if (nseg <= 0 || !seg_end) return NEGF_EINVAL;
    for (int sg = 0, prev = 0; sg < nseg; ++sg) { if (seg_end[sg] < prev || seg_end[sg] > m) return NEGF_EINVAL; prev = seg_end[sg]; }
    return seg_end[nseg - 1] == m ? NEGF_OK : NEGF_EINVAL;
}

// sum_m w_m G(E_m) per segment, everything on the device: Ed, wd [m], out [nseg][n*n]; seg_end is a host array
static int gr_seg_core(negf_ctx* c, SigmaProvider* p, int m, const cplx* Ed, const cplx* wd, int nseg, const int* seg_end,
                       cplx* out)
{
    int rc;
    const size_t n2 = (size_t)c->n * c->n;
    if (m > 0 && small_path(c, p) && m <= 4096) {
        const bool blocks = p->kind == SK_CHAIN1D || p->kind == SK_BETHE;
        if (blocks && (rc = ensure_blk(c, (size_t)m * p->blk_stride))) return rc;
        if ((size_t)m * n2 > c->small_part_cap) {
            NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
            dev_free(c->d_small_part); c->small_part_cap = 0;
            if ((rc = dev_alloc(&c->d_small_part, (size_t)m * n2))) return rc;
            c->small_part_cap = (size_t)m * n2;
        }
        if (blocks && (rc = run_sigma_blocks(c, p, m, Ed, c->d_iters, c->d_conv, 0))) return rc;
        ProfScope ps(c, "small");
        SmallFusedArgs a = small_args(c, p, 0, m, Ed);
        a.w = wd; a.partial = c->d_small_part; a.out = out; a.nseg = nseg; a.seg_end = seg_end;
        launch_small_fused(c->stream, a);
        negf_count_flops(8.0 * c->n * (double)c->n * c->n * m, 0.0);
    } else {
        if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
        NEGF_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)nseg * n2 * sizeof(cplx), c->stream));
        static int fused = -1;
        if (fused < 0) { const char* e = getenv("NEGF_GATHER_FUSED"); fused = e ? atoi(e) : 1; }
        for (int m0 = 0; m0 < m; m0 += c->batch) {
            const int nb = std::min(c->batch, m - m0);
            // (as in negf_gr_int_dev: the sums read the windowed inverse through its pivot bookkeeping, no gather)
            c->defer_gather = fused != 0;
            rc = run_assemble_inverse(c, p, m0, nb, Ed);
            c->defer_gather = false;
            if (rc) return rc;
            const bool perm = c->G_deferred;
            c->G_deferred = false;
            ProfScope ps(c, "accumulate");
            {
                // every segment's share of this batch in ONE launch pair (a joint arc + tail probe: 12 segments = 24 launches
                // of a few microseconds each behind a millisecond of inverse)
                std::vector<int> rlo, rhi, rslot;
                for (int sg = 0, start = 0; sg < nseg; start = seg_end[sg], ++sg) {
                    const int lo = std::max(start, m0), hi = std::min(seg_end[sg], m0 + nb);
                    if (hi > lo) { rlo.push_back(lo - m0); rhi.push_back(hi - m0); rslot.push_back(sg); }
                }
                if (launch_accumulate_ranges(c->stream, c->n, perm, wd + m0, perm ? c->W1 : c->G, c->d_ipiv, c->d_info + m0,
                                             (int)rlo.size(), rlo.data(), rhi.data(), rslot.data(), out, c->W2)) continue;
            }
            for (int sg = 0, start = 0; sg < nseg; start = seg_end[sg], ++sg) {
                const int lo = std::max(start, m0), hi = std::min(seg_end[sg], m0 + nb);
                if (hi <= lo) continue;
                if (perm)
                    launch_accumulate_perm(c->stream, c->n, hi - lo, wd + lo, c->W1 + (size_t)(lo - m0) * n2,
                                           c->d_ipiv + (size_t)(lo - m0) * 2 * c->n, c->d_info + lo, out + (size_t)sg * n2, c->W2);
                else
                    launch_accumulate(c->stream, (int)n2, hi - lo, wd + lo, c->G + (size_t)(lo - m0) * n2, out + (size_t)sg * n2, c->W2);
            }
        }
    }
    c->last_m = m;
    NEGF_HIP_CHECK(hipGetLastError());
    return NEGF_OK;
}

int negf_gr_int_seg_dev(negf_ctx* c, int handle, int m, const double* E_dev, const double* w_dev, int nseg,
                        const int* seg_end, double* out_dev)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out_dev || (m > 0 && (!E_dev || !w_dev)) || check_segments(m, nseg, seg_end)) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = ensure_mbuffers(c, m, p->n_contacts))) return rc;
    return gr_seg_core(c, p, m, reinterpret_cast<const cplx*>(E_dev), reinterpret_cast<const cplx*>(w_dev), nseg, seg_end,
                       reinterpret_cast<cplx*>(out_dev));
}

int negf_gr_int_seg(negf_ctx* c, int handle, int m, const double* E, const double* w, int nseg,
                    const int* seg_end, double* out, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out || (m > 0 && (!E || !w)) || check_segments(m, nseg, seg_end)) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    if ((rc = stage_grid(c, m, p->n_contacts, E, w))) return rc;
    // results: [nseg][n*n] on the device, then one download
    if ((size_t)nseg * n2 > c->seg_out_cap) {
        NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
        dev_free(c->d_seg_out); c->seg_out_cap = 0;
        if ((rc = dev_alloc(&c->d_seg_out, (size_t)nseg * n2))) return rc;
        c->seg_out_cap = (size_t)nseg * n2;
    }
    if ((rc = ensure_pinned(c, 2 * (size_t)m * sizeof(cplx) + nseg * n2 * sizeof(cplx) + (size_t)m * sizeof(int) + 64))) return rc;
    if ((rc = gr_seg_core(c, p, m, c->d_E, c->d_w, nseg, seg_end, c->d_seg_out))) return rc;
    // one download of all segment sums and the info, one synchronisation
    unsigned char* pout = c->h_pin + 2 * (size_t)m * sizeof(cplx);
    int* pinfo = reinterpret_cast<int*>(pout + nseg * n2 * sizeof(cplx));
    NEGF_HIP_CHECK(hipMemcpyAsync(pout, c->d_seg_out, nseg * n2 * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    if (m > 0) NEGF_HIP_CHECK(hipMemcpyAsync(pinfo, c->d_info, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    copy_bytes(out, pout, nseg * n2 * sizeof(cplx));
    rc = NEGF_OK;
    for (int i = 0; i < m; ++i) { if (info) info[i] = pinfo[i]; if (pinfo[i] != 0) rc = NEGF_ESINGULAR; }
    return rc;
}

// Adaptive nested quadrature with the refinement ON THE DEVICE (integratePointsAdaptiveANT, density.py:211-273): the m energies
// are the NEW nodes of consecutive levels of nint integrations (nlev[k] levels each, seg_end as in negf_gr_int_seg over all
// sum(nlev) levels, integral after integral); ratio[s] is the nested-weight ratio of level s -- NaN for the first level of an
// integration, which then starts from that level's sum; otherwise the integration continues from P_in[k].  One pass evaluates
// every level's sum, refine_levels_kernel (n > 512: one refine_wide_kernel launch per level) runs the reference's update and
// stopping test level by level, and only the result comes back: P_out [nint][n][n] (the value at the converged level, or after the last level), level_out [nint] (index of the
// converged level within the call, -1: not converged, continue with P_out as P_in), maxdp_out [sum(nlev)].  Saves the host the
// level sums (12 x n^2 per Fermi probe), the five numpy passes per level over them, and their download.
int negf_gr_int_refine(negf_ctx* c, int handle, int m, const double* E, const double* w, int nint, const int* nlev,
                       const int* seg_end, const double* ratio, double tol, const double* P_in, double* P_out,
                       int* level_out, double* maxdp_out, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!P_out || !level_out || !maxdp_out || !nlev || !ratio || nint <= 0 || nint > REF_MAX_INTS || (m > 0 && (!E || !w))) return NEGF_EINVAL;
    int nseg = 0;
    std::vector<int> first(nint + 1, 0);
    for (int k = 0; k < nint; ++k) { if (nlev[k] <= 0) return NEGF_EINVAL; nseg += nlev[k]; first[k + 1] = nseg; }
    if (nseg > REF_MAX_LEVELS || check_segments(m, nseg, seg_end)) return NEGF_EINVAL;
    for (int k = 0; k < nint; ++k) {
        for (int s = first[k] + 1; s < first[k + 1]; ++s) if (ratio[s] != ratio[s]) return NEGF_EINVAL;      // only a first level starts an integration
        if (ratio[first[k]] == ratio[first[k]] && !P_in) return NEGF_EINVAL;                                  // a continued one needs its running value
    }
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    const size_t gb = (size_t)m * sizeof(cplx), pb = (size_t)nint * n2 * sizeof(cplx);
    const size_t meta_b = (size_t)REF_MAX_LEVELS * 24 + (size_t)(2 * REF_MAX_INTS + 1 + REF_MAX_LEVELS) * 4;
    if ((rc = ensure_pinned(c, 2 * gb + pb + meta_b + (size_t)m * sizeof(int) + 256))) return rc;
    if ((rc = stage_grid(c, m, p->n_contacts, E, w))) return rc;
    if ((size_t)nseg * n2 > c->seg_out_cap) {
        NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
        dev_free(c->d_seg_out); c->seg_out_cap = 0;
        if ((rc = dev_alloc(&c->d_seg_out, (size_t)nseg * n2))) return rc;
        c->seg_out_cap = (size_t)nseg * n2;
    }
    if ((size_t)nint * n2 > c->ref_P_cap) {
        NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
        dev_free(c->d_ref_P); c->ref_P_cap = 0;
        if ((rc = dev_alloc(&c->d_ref_P, (size_t)nint * n2))) return rc;
        c->ref_P_cap = (size_t)nint * n2;
    }
    if (!c->d_ref_meta && (rc = dev_alloc(&c->d_ref_meta, meta_b))) return rc;
    // pinned layout: [E | w | P (in, then out) | ratio | maxdp | first | level | info]
    unsigned char* pP = c->h_pin + 2 * gb;
    unsigned char* pmeta = pP + pb;
    double* h_ratio = reinterpret_cast<double*>(pmeta);
    double* h_maxdp = h_ratio + REF_MAX_LEVELS;
    int* h_first = reinterpret_cast<int*>(h_maxdp + 2 * REF_MAX_LEVELS);          // (the host side mirrors the device layout)
    int* h_level = h_first + REF_MAX_INTS + 1;
    int* pinfo = h_level + REF_MAX_INTS + REF_MAX_LEVELS;
    double* d_ratio = reinterpret_cast<double*>(c->d_ref_meta);
    double* d_maxdp = d_ratio + REF_MAX_LEVELS;
    unsigned long long* d_maxbits = reinterpret_cast<unsigned long long*>(d_maxdp + REF_MAX_LEVELS);
    int* d_first = reinterpret_cast<int*>(d_maxbits + REF_MAX_LEVELS);
    int* d_level = d_first + REF_MAX_INTS + 1;
    int* d_nanflag = d_level + REF_MAX_INTS;
    std::memcpy(h_ratio, ratio, (size_t)nseg * sizeof(double));
    std::memcpy(h_first, first.data(), (size_t)(nint + 1) * sizeof(int));
    NEGF_HIP_CHECK(hipMemcpyAsync(d_ratio, h_ratio, (size_t)nseg * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NEGF_HIP_CHECK(hipMemcpyAsync(d_first, h_first, (size_t)(nint + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    for (int k = 0; k < nint; ++k)
        if (ratio[first[k]] == ratio[first[k]]) {                // continued: its running value goes up
            std::memcpy(pP + (size_t)k * n2 * sizeof(cplx), P_in + (size_t)k * n2 * 2, n2 * sizeof(cplx));
            NEGF_HIP_CHECK(hipMemcpyAsync(c->d_ref_P + (size_t)k * n2, pP + (size_t)k * n2 * sizeof(cplx), n2 * sizeof(cplx),
                                          hipMemcpyHostToDevice, c->stream));
        }
    if ((rc = gr_seg_core(c, p, m, c->d_E, c->d_w, nseg, seg_end, c->d_seg_out))) return rc;
    {
        ProfScope ps(c, "accumulate");
        if (c->n <= REF_MAX_N) {
            launch_refine_levels(c->stream, (int)n2, nint, c->d_seg_out, d_first, d_ratio, tol, c->d_ref_P, d_level, d_maxdp);
        } else {
            // one launch pair per level, queued without a host round trip (refine_wide_kernel)
            NEGF_HIP_CHECK(hipMemsetAsync(d_maxbits, 0, (size_t)nseg * sizeof(unsigned long long), c->stream));
            NEGF_HIP_CHECK(hipMemsetAsync(d_nanflag, 0, (size_t)nseg * sizeof(int), c->stream));
            NEGF_HIP_CHECK(hipMemsetAsync(d_level, 0xFF, (size_t)nint * sizeof(int), c->stream));            // -1: not converged
            NEGF_HIP_CHECK(hipMemsetAsync(d_maxdp, 0xFF, (size_t)nseg * sizeof(double), c->stream));         // all-ones: a NaN
            for (int k = 0; k < nint; ++k)
                for (int s = first[k]; s < first[k + 1]; ++s) {
                    cplx* Pk = c->d_ref_P + (size_t)k * n2;
                    const cplx* inc = c->d_seg_out + (size_t)s * n2;
                    if (ratio[s] != ratio[s])
                        NEGF_HIP_CHECK(hipMemcpyAsync(Pk, inc, n2 * sizeof(cplx), hipMemcpyDeviceToDevice, c->stream));
                    else
                        launch_refine_level_wide(c->stream, (int)n2, inc, ratio[s], Pk, s - first[k], tol, d_level + k,
                                                 d_maxbits + s, d_nanflag + s, d_maxdp + s);
                }
        }
    }
    NEGF_HIP_CHECK(hipGetLastError());
    NEGF_HIP_CHECK(hipMemcpyAsync(pP, c->d_ref_P, pb, hipMemcpyDeviceToHost, c->stream));
    NEGF_HIP_CHECK(hipMemcpyAsync(h_maxdp, d_maxdp, (size_t)nseg * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    NEGF_HIP_CHECK(hipMemcpyAsync(h_level, d_level, (size_t)nint * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (m > 0) NEGF_HIP_CHECK(hipMemcpyAsync(pinfo, c->d_info, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    copy_bytes(P_out, pP, pb);
    std::memcpy(maxdp_out, h_maxdp, (size_t)nseg * sizeof(double));
    std::memcpy(level_out, h_level, (size_t)nint * sizeof(int));
    rc = NEGF_OK;
    for (int i = 0; i < m; ++i) { if (info) info[i] = pinfo[i]; if (pinfo[i] != 0) rc = NEGF_ESINGULAR; }
    return rc;
}

// the same for GrLessInt (the adaptive bias-window integral, densityGrid, density.py:605-658)
int negf_gless_int_seg(negf_ctx* c, int handle, int ind, int m, const double* E, const double* w, int nseg,
                       const int* seg_end, double* out, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out || (m > 0 && (!E || !w)) || check_segments(m, nseg, seg_end)) return NEGF_EINVAL;
    const int contact = norm_contact(p, ind);
    if (contact == -2) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    if ((rc = stage_grid(c, m, p->n_contacts, E, w))) return rc;
    if ((size_t)nseg * n2 > c->seg_out_cap) {
        NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
        dev_free(c->d_seg_out); c->seg_out_cap = 0;
        if ((rc = dev_alloc(&c->d_seg_out, (size_t)nseg * n2))) return rc;
        c->seg_out_cap = (size_t)nseg * n2;
    }
    if ((rc = ensure_pinned(c, 2 * (size_t)m * sizeof(cplx) + nseg * n2 * sizeof(cplx) + (size_t)m * sizeof(int) + 64))) return rc;
    if ((rc = gless_core(c, p, contact, m, c->d_E, c->d_w, c->d_seg_out, nseg, seg_end))) return rc;
    unsigned char* pout = c->h_pin + 2 * (size_t)m * sizeof(cplx);
    int* pinfo = reinterpret_cast<int*>(pout + nseg * n2 * sizeof(cplx));
    NEGF_HIP_CHECK(hipMemcpyAsync(pout, c->d_seg_out, nseg * n2 * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    if (m > 0) NEGF_HIP_CHECK(hipMemcpyAsync(pinfo, c->d_info, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    copy_bytes(out, pout, nseg * n2 * sizeof(cplx));
    rc = NEGF_OK;
    for (int i = 0; i < m; ++i) { if (info) info[i] = pinfo[i]; if (pinfo[i] != 0) rc = NEGF_ESINGULAR; }
    return rc;
}

int negf_gless_int_seg_dev(negf_ctx* c, int handle, int ind, int m, const double* E_dev, const double* w_dev, int nseg,
                           const int* seg_end, double* out_dev)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out_dev || (m > 0 && (!E_dev || !w_dev)) || check_segments(m, nseg, seg_end)) return NEGF_EINVAL;
    const int contact = norm_contact(p, ind);
    if (contact == -2) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    return gless_core(c, p, contact, m, reinterpret_cast<const cplx*>(E_dev), reinterpret_cast<const cplx*>(w_dev),
                      reinterpret_cast<cplx*>(out_dev), nseg, seg_end);
}

int negf_gless_int(negf_ctx* c, int handle, int ind, int m, const double* E, const double* w,
                   double* out, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (!out || (m > 0 && (!E || !w))) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = stage_grid(c, m, p->n_contacts, E, w))) return rc;
    if ((rc = negf_gless_int_dev(c, handle, ind, m, reinterpret_cast<double*>(c->d_E),
                                 reinterpret_cast<double*>(c->d_w), reinterpret_cast<double*>(c->d_acc)))) return rc;
    return fetch_matrix_and_info(c, m, c->d_acc, out, info);
}

int negf_gr_batch(negf_ctx* c, int handle, int m, const double* E, double* G_out, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (m > 0 && (!E || !G_out)) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    if ((rc = stage_grid(c, m, p->n_contacts, E, nullptr))) return rc;
    if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
    for (int m0 = 0; m0 < m; m0 += c->batch) {
        const int nb = std::min(c->batch, m - m0);
        if ((rc = run_assemble_inverse(c, p, m0, nb, c->d_E))) return rc;
        if ((rc = download(c, reinterpret_cast<cplx*>(G_out) + n2 * m0, c->G, n2 * nb))) return rc;
    }
    c->last_m = m;
    return reduce_info(c, m, info);
}

int negf_transmission(negf_ctx* c, int handle, int contact_L, int contact_R, int spin_mode, int m,
                      const double* E, double* T, double* Tspin, int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (m > 0 && (!E || !T)) return NEGF_EINVAL;
    if (spin_mode == NEGF_SPIN_BLOCK && !Tspin) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = stage_grid(c, m, p->n_contacts, E, nullptr))) return rc;
    double* dT = c->d_scal;                 // [m]
    double* dTs = c->d_scal + c->m_cap;     // [m][4]
    if ((rc = negf_transmission_dev(c, handle, contact_L, contact_R, spin_mode, m,
                                    reinterpret_cast<double*>(c->d_E), dT, dTs))) return rc;
    if (spin_mode == NEGF_SPIN_BLOCK) {
        if ((rc = download(c, Tspin, dTs, (size_t)m * 4))) return rc;
        for (int i = 0; i < m; ++i) {
            // total = sum of the four components in order, as jnp.sum(T_spin) (transport.py:181)
            T[i] = ((Tspin[4 * i] + Tspin[4 * i + 1]) + Tspin[4 * i + 2]) + Tspin[4 * i + 3];
        }
    } else {
        if ((rc = download(c, T, dT, (size_t)m))) return rc;
    }
    return reduce_info(c, m, info);
}

int negf_dos(negf_ctx* c, int handle, int m, const double* E, double* dos_total, double* dos_site,
             int* info)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (m > 0 && (!E || !dos_total)) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = stage_grid(c, m, p->n_contacts, E, nullptr))) return rc;
    if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
    for (int m0 = 0; m0 < m; m0 += c->batch) {
        const int nb = std::min(c->batch, m - m0);
        if ((rc = run_assemble_inverse(c, p, m0, nb, c->d_E))) return rc;
        { ProfScope ps(c, "trace"); launch_dos(c->stream, c->n, nb, c->G, c->d_scal + m0, dos_site ? c->d_site : nullptr); }
        if (dos_site && (rc = download(c, dos_site + (size_t)m0 * c->n, c->d_site, (size_t)nb * c->n))) return rc;
    }
    if ((rc = download(c, dos_total, c->d_scal, (size_t)m))) return rc;
    c->last_m = m;
    return reduce_info(c, m, info);
}

int negf_sigma_eval(negf_ctx* c, int handle, int contact, int m, const double* E, double* sigma_out,
                    int* iters, int* converged)
{
    SigmaProvider* p = get_provider(c, handle);
    int rc = check_ready(c, p, m);
    if (rc) return rc;
    if (m > 0 && (!E || !sigma_out)) return NEGF_EINVAL;
    const int ct = norm_contact(p, contact);
    if (ct == -2) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    if ((rc = stage_grid(c, m, p->n_contacts, E, nullptr))) return rc;
    if ((rc = ensure_workspace(c, m, p->blk_stride))) return rc;
    cplx* out = reinterpret_cast<cplx*>(sigma_out);
    for (int m0 = 0; m0 < m; m0 += c->batch) {
        const int nb = std::min(c->batch, m - m0);
        switch (p->kind) {
        case SK_CONST: {
            const cplx* src = ct < 0 ? p->d_const_tot : p->d_const_c + n2 * ct;
            for (int b = 0; b < nb; ++b)
                if ((rc = download(c, out + n2 * (m0 + b), src, n2))) return rc;
            break;
        }
        case SK_PRECOMPUTED: {
            const cplx* src = (ct < 0 || !p->d_pre_c) ? p->d_pre_tot + n2 * m0 : nullptr;
            if (src) { if ((rc = download(c, out + n2 * m0, src, n2 * nb))) return rc; }
            else {
                const int pn = std::max(p->pre_nc, 1);
                for (int b = 0; b < nb; ++b)
                    if ((rc = download(c, out + n2 * (m0 + b), p->d_pre_c + n2 * ((size_t)(m0 + b) * pn + ct), n2))) return rc;
            }
            break;
        }
        case SK_CHAIN1D:
        case SK_BETHE: {
            if ((rc = run_sigma_blocks(c, p, nb, c->d_E + m0, c->d_iters + (size_t)m0 * p->n_contacts,
                                       c->d_conv + (size_t)m0 * p->n_contacts, m0))) return rc;
            launch_scatter_blocks(c->stream, c->n, nb, c->d_blk, p->blk_stride, p->n_contacts, p->d_nc,
                                  p->d_blk_off, p->d_inds_off, p->d_inds, ct, c->d_A);
            cplx* res = c->d_A;
            if (p->d_xi) {
                launch_zgemm(c->stream, c->n, c->n, c->n, nb, p->d_xi, c->n, 0, c->d_A, c->n, n2, 0, c->d_T2, c->n, n2);
                launch_zgemm(c->stream, c->n, c->n, c->n, nb, c->d_T2, c->n, n2, p->d_xi, c->n, 0, 0, c->d_A, c->n, n2);
            }
            if ((rc = download(c, out + n2 * m0, res, n2 * nb))) return rc;
            break;
        }
        default: return NEGF_EINVAL;
        }
    }
    if (p->kind == SK_CHAIN1D || p->kind == SK_BETHE) {
        if (iters && (rc = download(c, iters, c->d_iters, (size_t)m * p->n_contacts))) return rc;
        if (converged && (rc = download(c, converged, c->d_conv, (size_t)m * p->n_contacts))) return rc;
    } else {
        if (iters) for (int i = 0; i < m * p->n_contacts; ++i) iters[i] = 0;
        if (converged) for (int i = 0; i < m * p->n_contacts; ++i) converged[i] = 1;
    }
    NEGF_HIP_CHECK(hipGetLastError());
    return NEGF_OK;
}

int negf_bethe_raw(negf_ctx* c, const double* H, const double* Slist, const double* Vlist, double eta,
                   double conv, double mix, int max_iter, int force_iters, int which, int m,
                   const double* E, double* out, int* iters, int* converged)
{
    if (!c || !H || !Slist || !Vlist || m < 0 || (which != 1 && which != 2)) return NEGF_EINVAL;
    if (m > 0 && (!E || !out)) return NEGF_EINVAL;
    if (m == 0) return NEGF_OK;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    const int nd = which == 1 ? 12 : 9;
    double *dH = nullptr, *dS = nullptr, *dV = nullptr;
    cplx *dE = nullptr, *dout = nullptr;
    int *dit = nullptr, *dcv = nullptr;
    int rc = NEGF_OK;
    if ((rc = dev_alloc(&dH, 81)) || (rc = dev_alloc(&dS, 12 * 81)) || (rc = dev_alloc(&dV, 12 * 81)) ||
        (rc = dev_alloc(&dE, (size_t)m)) || (rc = dev_alloc(&dout, (size_t)m * nd * 81)) ||
        (rc = dev_alloc(&dit, (size_t)m)) || (rc = dev_alloc(&dcv, (size_t)m))) goto done;
    if ((rc = upload(c, dH, H, 81)) || (rc = upload(c, dS, Slist, 12 * 81)) || (rc = upload(c, dV, Vlist, 12 * 81)) ||
        (rc = upload(c, dE, reinterpret_cast<const cplx*>(E), (size_t)m))) goto done;
    {
        ProfScope ps(c, "bethe");
        launch_bethe_raw(c->stream, dH, dS, dV, eta, conv, mix, max_iter, force_iters, which, m, dE, dout, dit, dcv);
    }
    if ((rc = download(c, reinterpret_cast<cplx*>(out), dout, (size_t)m * nd * 81))) goto done;
    if (iters && (rc = download(c, iters, dit, (size_t)m))) goto done;
    if (converged && (rc = download(c, converged, dcv, (size_t)m))) goto done;
    if (hipGetLastError() != hipSuccess) rc = NEGF_EHIP;
done:
    dev_free(dH); dev_free(dS); dev_free(dV); dev_free(dE); dev_free(dout); dev_free(dit); dev_free(dcv);
    return rc;
}

// ------------------------------------------------------------ g(E) cache knob
int negf_set_chain_cache(negf_ctx* c, int max_grids)
{
    if (!c || max_grids < 0 || max_grids > 4096) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (max_grids < (int)c->gcache.size()) free_gcache(c);     // shrinking (or switching off) drops every entry
    c->gcache_max = max_grids;
    c->gcache.reserve((size_t)max_grids);                      // entries are handed out by pointer: no reallocation later
    return NEGF_OK;
}

int negf_set_chain_cache_bytes(negf_ctx* c, long long max_bytes)
{
    if (!c || max_bytes < 0) return NEGF_EINVAL;
    c->gcache_bytes_max = (size_t)max_bytes;
    return NEGF_OK;
}

int negf_chain_cache_clear(negf_ctx* c)
{
    if (!c) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    NEGF_HIP_CHECK(hipStreamSynchronize(c->stream));
    free_gcache(c);
    c->gcache.reserve((size_t)c->gcache_max);
    return NEGF_OK;
}

int negf_chain_cache_stats(negf_ctx* c, long long* hits, long long* misses, long long* entries, long long* bytes)
{
    if (!c) return NEGF_EINVAL;
    long long n = 0, b = 0;
    for (const auto& e : c->gcache)
        if (e.valid) { ++n; b += (long long)(e.g_cap * sizeof(cplx) + 2 * e.it_cap * sizeof(int)); }
    if (hits) *hits = (long long)c->gcache_hits;
    if (misses) *misses = (long long)c->gcache_misses;
    if (entries) *entries = n;
    if (bytes) *bytes = b;
    return NEGF_OK;
}

// ------------------------------------------------------------------ diagnostics
int negf_profile_enable(negf_ctx* c, int on) { if (!c) return NEGF_EINVAL; c->profiling = on != 0; return NEGF_OK; }
int negf_profile_reset(negf_ctx* c) { if (!c) return NEGF_EINVAL; prof_resolve(c); c->prof.clear(); return NEGF_OK; }
int negf_profile_read(negf_ctx* c, const char* family, double* total_ms, int* launches)
{
    if (!c || !family) return NEGF_EINVAL;
    prof_resolve(c);
    auto it = c->prof.find(family);
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    return NEGF_OK;
}

int negf_profile_read_flops(negf_ctx* c, const char* family, double* flops_algorithmic, double* flops_mfma_issued)
{
    if (!c || !family) return NEGF_EINVAL;
    auto it = c->prof.find(family);
    if (flops_algorithmic) *flops_algorithmic = it == c->prof.end() ? 0.0 : it->second.flops_alg;
    if (flops_mfma_issued) *flops_mfma_issued = it == c->prof.end() ? 0.0 : it->second.flops_mfma;
    return NEGF_OK;
}

int negf_selftest_mfma(negf_ctx* c, double* max_err)
{
    if (!c || !max_err) return NEGF_EINVAL;
    NEGF_HIP_CHECK(hipSetDevice(c->device));
    return run_mfma_selftest(c->stream, max_err);
}

}  // extern "C"
