// phx_api.cpp — C-ABI of libphx (include/phx.h): context, HBM buffers, kernel orchestration, taps.
//
// There is deliberately no CPU implementation of the path in this file: if HIP is unusable every
// entry point that would compute returns PHX_E_NODEVICE / PHX_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "phx_internal.h"
extern "C" {
#include "phx_dec.h"
}

namespace {

thread_local std::string g_create_error;

enum Stage { ST_MEMSET = 0, ST_FEATURES, ST_ORF_COUNT, ST_ORF_EMIT, ST_ORF_STATS, ST_SCORE, ST_NODES, ST_EDGE_COUNT, ST_EDGE_FILL, ST_SSSP, ST_CERTIFY, ST_COPY, ST_INORDER, ST_WAVE_PLAN };
const char *kStageName[PHX_N_STAGES] = {"memset", "features", "orf_count", "orf_emit", "orf_stats", "score", "nodes", "edges_count", "edges_fill", "sssp", "certify", "copies", "inorder", "wave_plan"};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// per-contig records on the host, in pinned memory (they cross PCIe several times per run)
struct MetaBuf {
    DMeta *p = nullptr;
    size_t n = 0, cap = 0;
    bool assign(size_t count) {
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = count + count / 4 + 16;
            if (hipHostMalloc((void **)&p, want * sizeof(DMeta), hipHostMallocDefault) != hipSuccess) { p = nullptr; n = 0; return false; }
            cap = want;
        }
        n = count;
        if (n) memset(p, 0, n * sizeof(DMeta));
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
    DMeta *begin() { return p; }
    DMeta *end() { return p + n; }
    const DMeta *begin() const { return p; }
    const DMeta *end() const { return p + n; }
    DMeta &operator[](size_t i) { return p[i]; }
    const DMeta &operator[](size_t i) const { return p[i]; }
    DMeta *data() { return p; }
};

} // namespace

// Worker threads that pack the letters of phx_upload (created at the first batch large enough to need them, kept until phx_destroy:
// starting a thread costs as much as packing a megabase).
struct StagePool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    const std::function<void()> *job = nullptr;
    uint64_t gen = 0;
    int active = 0;
    bool stop = false;
    void grow(int n) {
        while ((int)th.size() < n) {
            try { th.emplace_back([this, seen = gen]() mutable { loop(seen); }); } catch (...) { break; }
        }
    }
    void loop(uint64_t seen) {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const std::function<void()> *j = job;
            lk.unlock();
            (*j)();
            lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    void start(const std::function<void()> *j) { // every worker runs *j once; `finish` returns when all have
        std::lock_guard<std::mutex> lk(m);
        job = j; active = (int)th.size(); gen++;
        cv.notify_all();
    }
    void finish() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
    }
    ~StagePool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; cv.notify_all(); }
        for (std::thread &t : th) t.join();
    }
};

struct phx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr}; // side streams: independent SSSP classes overlap (0..2); k_wave_plan beside k_edges<true> (3)
    bool in_flight = false;      // phx_run_async has enqueued a run that phx_wait has not collected yet
    int pend_mask = 0;
    int64_t pend_lds[4] = {0, 0, 0, 0};
    hipEvent_t ev_fork = nullptr, ev_fork_plan = nullptr, ev_fork_nodes = nullptr, ev_join_nodes = nullptr, ev_fork_pre = nullptr, ev_join_pre = nullptr, ev_fork_orf = nullptr, ev_join_orf = nullptr, ev_fork_score = nullptr, ev_join_score = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    phx_params params;
    std::string err;
    // device constants
    DParams *d_params = nullptr;
    // batch
    int n = 0;
    bool uploaded = false, ran = false;
    int64_t totalL = 0, tot_orf = 0, tot_grp = 0, tot_node = 0, tot_edge = 0;
    int n_limbs = 0;
    MetaBuf meta;            // host copy of the per-contig records: the layout as the host set it; the device's values after fetch_meta()
    bool meta_stale = false; // a run has finished since c->meta was last fetched
    DRes *res = nullptr;     // pinned: status / gene count / first gene of every contig after a run
    size_t res_cap = 0;
    std::vector<uint32_t> ftab; // k_features' tables: voff (n + 1: first virtual word = first record - 1 of every contig), then wfirst (per 62 virtual words: the contig of the first)
    uint32_t vtotal = 0;        // virtual words of the batch (records without the two pads)
    bool defcod = false;        // the codon tables are the reference's defaults
    const void *attached = nullptr; // phx_attach: the caller's letters on the device (k_pack_planes turns them into records at the head of a run)
    hipEvent_t ev_upload = nullptr; bool upload_pending = false; // recorded behind the last copy of phx_upload
    hipEvent_t ev_layout = nullptr; bool layout_pending = false; // recorded behind push_layout's copies
    hipEvent_t ev_piece[2] = {nullptr, nullptr};                  // phx_upload: a piece's copy -> its k_features launch
    void *h_tiles = nullptr; size_t h_tiles_cap = 0;             // pinned copy of ftab
    bool eager_now = false;    // ... and this launch is that run (launch_once)
    bool eager_done = false;   // phx_upload has already reset the accumulators and run k_features for this batch: the next run starts behind them
    bool trna_clean = true;    // no phx_set_trnas with hits since the layout was set
    std::unique_ptr<StagePool> pool;
    // buffers
    DevBuf b_bridge, b_recs, b_meta, b_tiles, b_nbits, b_nbase, b_cbits, b_orf, b_ostat, b_oweight, b_owi, b_oflag, b_onode, b_grp, b_bits, b_cpre, b_bpre, b_item, b_iprev;
    DevBuf b_ewf, b_esrcf;   // fp64 weights and plain sources of the batch last run, recomputed for the edge tap (k_edges<true, true>)
    bool tapw_valid = false; // ... are those of the run whose results the context holds
    int64_t tot_nbits = 0, tot_bridge = 0;
    int64_t tot_words = 0, tot_items = 0;
    DevBuf b_win, b_wrole;
    DevBuf b_tnode, b_tedge, b_tnid, b_tbits; // tRNA masking (phx_set_trnas)
    std::vector<DTNode> h_tnode;               // host copy for the node tap
    bool has_trna = false;
    DevBuf b_cint, b_csig; // scratch of k_certify (per node)
    DevBuf b_eref;         // k_refine -> k_certify: bounds on the reference's integers, per edge (allocated when a certificate is first asked for)
    bool certify = true;   // phx_certified works (PHX_CREATE_NO_CERTIFY: it reports -1 and the scratch is not allocated)
    bool cert_done = false; // k_certify has run on the results the context holds
    bool exact = true;       // phx_download* solve uncertified contigs again on the host (PHX_CREATE_NO_EXACT: they do not)
    bool exact_done = false; // ... and that has happened for the results the context holds
    std::map<int, std::vector<DGene>> exact_genes; // contig -> its genes from the host re-solve (phx_exact.inc)
    std::vector<int> host_only;                    // contigs of the batch last run that no device kernel could solve (path sums beyond 1088 bits) and the host did: their status is PHX_S_OVERFLOW again when exactness is switched off
    int exact_failed = 0;    // contigs whose replay met an operation phx_dec.c does not restate (left as the device solved them)
    double cert_scale = 1.0;
    bool cert_wide = false; // test switch: every contig through k_certify_wide
    bool poison = false;    // test switch: new device buffers are filled with a byte pattern
    bool one_stream = false; // test switch: the side streams are the main stream
    bool learning = false;   // enqueue_run is sizing the buffers between the kernels (DCaps.flags bit 2)
    DevBuf b_tie;         // scratch of k_inorder; grows to what the contigs with equal-length alternative paths ask for
    int64_t tie_seen = 0; // largest DTotals.tie_need a run reported
    DevBuf b_ekey;        // phx_solve: rank of every edge in the caller's order
    DevBuf b_meta0;          // the per-contig records as a run starts (layout fields set, accumulators zero): copied over b_meta on the device at the start of every run
    bool meta0_dirty = true; // batch layout changed since b_meta0 was written
    int runs_on_layout = 0;  // completed runs since the batch layout last changed (a graph is captured from the second on)
    DevBuf b_node, b_parent, b_inoff, b_no, b_npos, b_ehit, b_mreach, b_olist, b_dist, b_esrc, b_ew, b_ewl, b_path, b_genes, b_gtot, b_tot, b_lpart, b_res, b_sord, b_gtab, b_erank;
    DTotals *h_tot = nullptr; // pinned
    bool have_plan = false;    // a run completed on this context: its buffers, solver classes and LDS sizes are the first guess for the next
    int last_mask = 0;
    int last_vmax = 0;       // largest node count of a contig in the last run (LDS tables of k_certify)
    int64_t last_lds[4] = {0, 0, 0, 0};
    int64_t max_len = 0;
    // steady-state runs replay a captured HIP graph of the whole enqueue (valid while batch layout, buffers and solver classes stand)
    bool graphs_enabled = true, graph_valid = false, tiles_dirty = true;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_flags = -1;
    void *h_stage = nullptr; // pinned staging for H2D of the bases (records)
    size_t h_stage_cap = 0;
    DGene *h_genes = nullptr; size_t h_genes_cap = 0; // pinned staging of the gene records (D2H at link rate)
    // profiling
    bool prof = false;
    uint32_t prof_mask = 0xffffffffu; // stages that are bracketed by events when prof is on
    int n_simd = 1024; // SIMDs of the device (4 per CU): places of k_sssp_wave
    bool force_global_sssp = false; // development switch: run every contig through the global-memory SSSP kernel
    bool no_wave = false, always_sync = false;
    bool no_fuse = false;          // PHX_CREATE_NO_FUSE: small batches through the staged kernels as well
    bool duo = true;               // 128-bit contigs: k_sssp_duo (feeder + solver wavefront) instead of k_sssp_wave<2> (PHX_CREATE_NO_DUO, env PHX_NO_DUO=1: off)
    int front_spins = 8000;        // FRONT_SPINS of k_front (env PHX_FRONT_SPINS at phx_create)
    bool side_score = true;        // k_score on a side stream beside k_node_attr and the edge count (env PHX_NO_SIDE_SCORE=1: in line)
    int orf_stream = 1;            // which side stream k_edges_orf takes (env PHX_ORF_STREAM: 0..3; -1: the main stream, in front of the edge fill)
    bool orf_rows = true;          // the ORF edges' rows by k_edges_orf beside the edge fill (env PHX_NO_ORF_ROWS=1: by k_edges<true> itself)
    bool eager_cert = true;        // phx_run_async puts the certificate behind the run (env PHX_NO_EAGER_CERT=1: it does not)
    bool pend_cert = false;        // phx_run_async put the certificate kernels behind the run in flight: phx_download* will find it done
    bool pend_front = false;       // the run in flight (or the captured graph) has k_front as its front end
    bool front_off = false;        // k_front once waited too long at a grid barrier on this context (its workgroups were not all resident): staged kernels from then on
    int64_t front_runs = 0;        // runs of this context whose front end was k_front (phx_front_runs)
    // small batches: a contig's shortest path by up to 16 wavefront pairs side by side, joined and proven by k_seg_merge (phx_sssp_seg.inc)
    bool seg_on = true;            // PHX_CREATE_NO_SEG, env PHX_NO_SEG=1: off
    bool seg_off = false;          // a contig of this batch could not be joined or proven (it was solved by one sweep in the same run): one sweep per contig until the next batch is uploaded
    bool seg_never = false;        //   ... and for good once that has happened to more than a quarter of the runs
    bool pend_seg = false;         // the run in flight uses segments
    int pend_seg_k = 0;            //   ... at most this many per contig (the stride of DBatch.segw: phx_seg_stats)
    bool seg_clean = false;        // a run of this batch has proven every contig's segments: later runs of it do not launch the one sweep behind them
    int seg_max_n = 32;            // batches of up to this many contigs (env PHX_SEG_MAX_N): a contig that cannot be proven (~1 %) costs its one sweep on top,
                                   // and a batch waits for it: beyond ~32 contigs that eats the gain
    int seg_margin_bp = 6000;      // sequence a segment sweeps in front of what it commits (env PHX_SEG_MARGIN_BP; observed need: <= 3.2 kb)
    int64_t seg_runs = 0, seg_aborts = 0, seg_fallbacks = 0; // phx_seg_runs; contigs, over the life of the context, that one sweep solved behind their segments
    DevBuf b_swin, b_swrole, b_sdist, b_segw;
    int64_t plan_timeouts = 0;     // contigs, over the life of the context, whose solver gave up waiting for the planner it follows (phx_plan_timeouts)
    bool plan_stream_off = false;  // ... after the first of them the solver is launched behind its planner again on this context
    float stage_ms[PHX_N_STAGES] = {0};
    int stage_n[PHX_N_STAGES] = {0};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> ev_pool;
};

static int seg_cap(const phx_ctx *c); // segments per contig at most in the batch at hand (below)
namespace {

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
            return PHX_E_HIP;                                                                                  \
        }                                                                                                      \
    } while (0)

int ensure(phx_ctx *c, DevBuf &b, size_t bytes) {
    // 2 KB of slack behind every array: the staging loads of k_sssp_wave read whole 16-byte / 256-node groups
    if (bytes + 2048 <= b.cap && b.p) return PHX_OK;
    c->graph_valid = false; // a buffer moves: the captured graph holds stale pointers
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    // test switch: fresh device memory is not zero once several processes share a GPU (they hand each other's freed pages around);
    // a kernel that reads what nobody wrote shows with this, on a box of its own too
    if (c->poison) HIPCHK(c, hipMemset(b.p, 0xA5, want));
    return PHX_OK;
}
void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

inline int code_of(char c) { return c == 'a' ? 0 : c == 'c' ? 1 : c == 't' ? 2 : c == 'g' ? 3 : -1; }

int check_params(const phx_params *p) {
    if (!p || p->minlen < 6 || p->n_start < 1 || p->n_start > PHX_MAX_CODONS || p->n_stop < 1 || p->n_stop > PHX_MAX_CODONS) return PHX_E_PARAM;
    for (int i = 0; i < p->n_start; i++) {
        for (int j = 0; j < 3; j++) if (code_of(p->start[i][j]) < 0) return PHX_E_PARAM;
        if (p->start[i][3] != 0) return PHX_E_PARAM;
        if (!(p->start_w[i] == p->start_w[i])) return PHX_E_PARAM;
        if (!memchr(p->start_w_text[i], 0, sizeof(p->start_w_text[i]))) return PHX_E_PARAM;
        if (p->start_w_text[i][0]) { dec_t t; if (dec_from_str(&t, p->start_w_text[i])) return PHX_E_PARAM; }
    }
    for (int i = 0; i < p->n_stop; i++) {
        for (int j = 0; j < 3; j++) if (code_of(p->stop[i][j]) < 0) return PHX_E_PARAM;
        if (p->stop[i][3] != 0) return PHX_E_PARAM;
    }
    return PHX_OK;
}

void build_dparams(const phx_params *p, DParams *d) {
    memset(d, 0, sizeof(*d));
    d->minlen = p->minlen;
    d->n_start = p->n_start;
    for (int i = 0; i < p->n_start; i++) d->start_w[i] = p->start_w[i];
    { dec_t q[PHX_MAX_CODONS]; dec_start_weights(p->n_start, p->start_w_text, p->start_w, q); for (int i = 0; i < p->n_start; i++) dec_to_dd(&q[i], &d->sw_hi[i], &d->sw_lo[i]); }
    auto codon_code = [](const char *c) { return code_of(c[0]) | (code_of(c[1]) << 2) | (code_of(c[2]) << 4); };
    auto rc_code = [](int ci) { // reverse complement of a codon code
        int c0 = ci & 3, c1 = (ci >> 2) & 3, c2 = (ci >> 4) & 3;
        return (c2 ^ 2) | ((c1 ^ 2) << 2) | ((c0 ^ 2) << 4);
    };
    int start_idx[64], stop_is[64];
    for (int i = 0; i < 64; i++) { start_idx[i] = -1; stop_is[i] = 0; }
    for (int i = p->n_start - 1; i >= 0; i--) start_idx[codon_code(p->start[i])] = i; // first occurrence wins, like dict lookup of the codon
    for (int i = 0; i < p->n_stop; i++) stop_is[codon_code(p->stop[i])] = 1;
    const int atg = codon_code("atg"), cat = codon_code("cat");
    for (int ci = 0; ci < 64; ci++) {
        const int rc = rc_code(ci);
        int cls = CLS_NONE, idx = 0;
        if (start_idx[ci] >= 0) { cls = CLS_FS; idx = start_idx[ci]; }      // functions.py:198
        else if (start_idx[rc] >= 0) { cls = CLS_RS; idx = start_idx[rc]; } // functions.py:200
        else if (stop_is[ci]) cls = CLS_FT;                                 // functions.py:202
        else if (stop_is[rc]) cls = CLS_RT;                                 // functions.py:215
        d->cls_tab[ci] = (uint8_t)(cls | (idx << 3) | (start_idx[rc] >= 0 ? 0x80 : 0));
        d->atg_tab[ci] = (uint8_t)((ci == atg ? 1 : 0) | (ci == cat ? 2 : 0) | (idx << 2)); // bits 2..5: index of the start codon, as in cls_tab
    }
}

// ---- profiling helpers ----
hipEvent_t get_event(phx_ctx *c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct StageTimer {
    phx_ctx *c; int st; bool on = false; hipEvent_t a = nullptr, b = nullptr; hipStream_t str = nullptr;
    StageTimer(phx_ctx *c_, int st_, hipStream_t on_stream = nullptr) : c(c_), st(st_) { // on_stream: a side stream's kernels (k_wave_plan beside the edge fill)
        on = c->prof && ((c->prof_mask >> st_) & 1u);
        str = on_stream ? on_stream : c->stream;
        if (on) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, str); }
    }
    ~StageTimer() {
        if (on) { (void)hipEventRecord(b, str); c->pending.push_back({st, {a, b}}); }
    }
};
void collect_timers(phx_ctx *c) {
    for (auto &p : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) { c->stage_ms[p.first] += ms; c->stage_n[p.first]++; }
        c->ev_pool.push_back(p.second.first);
        c->ev_pool.push_back(p.second.second);
    }
    c->pending.clear();
}

// elements the context's buffers can take (every buffer keeps 2 KB of slack, see ensure())
int64_t cap_of(const DevBuf &b, size_t elem, int64_t reserve) {
    if (!b.p || b.cap < 2048 + elem) return 0;
    const int64_t k = (int64_t)((b.cap - 2048) / elem) - reserve;
    return k > 0 ? k : 0;
}
// Batches beyond 4096 contigs keep the gene records in two halves of b_genes: every contig's own place in the first (no shared
// counter: 20 000 atomics on one address were 0.25 ms), the packed records in the second (k_gene_pack, ~0.03 ms).  Entries per half:
static inline bool gene_pack(const phx_ctx *c) { return c->n > 4096; }
// more contigs than k_sssp_wave has places at a time (one wavefront per SIMD): its workgroups take the contigs in the order of k_sssp_order
static inline bool sssp_ordered(const phx_ctx *c) { return c->n > c->n_simd; }
static inline int64_t gene_half(const phx_ctx *c) { return (int64_t)(c->b_genes.cap / sizeof(DGene)) / 2; }

void current_caps(const phx_ctx *c, DCaps *k) {
    const int limbs = c->n_limbs > 2 ? c->n_limbs : 2;
    k->orf = std::min(std::min(cap_of(c->b_orf, sizeof(DOrf), 1), cap_of(c->b_ostat, sizeof(DOrfStat), 1)), std::min(std::min(cap_of(c->b_oweight, 8, 1), cap_of(c->b_owi, 8, 1)), std::min(cap_of(c->b_onode, 4, 1), cap_of(c->b_oflag, 1, 8))));
    k->grp = std::min(cap_of(c->b_grp, sizeof(DGrp), 1), cap_of(c->b_genes, 2 * sizeof(DGene), 1) - (int64_t)c->h_tnode.size()); // a path k_inorder replaces may take new gene slots; tRNA features are genes too
    int64_t v = cap_of(c->b_node, sizeof(DNode), 8);
    for (const DevBuf *q : {&c->b_parent, &c->b_path, &c->b_olist}) v = std::min(v, cap_of(*q, 4, 8));
    v = std::min(v, cap_of(c->b_no, 8, 8));
    v = std::min(v, cap_of(c->b_npos, 4, 8));
    v = std::min(v, cap_of(c->b_ehit, 8, 8));
    v = std::min(v, cap_of(c->b_erank, 16, 8));
    v = std::min(v, cap_of(c->b_mreach, 16, 8));
    v = std::min(v, cap_of(c->b_inoff, 4, 8 + (int64_t)c->n + 1));
    v = std::min(v, cap_of(c->b_dist, 8 * (size_t)limbs, 8));
    if (c->certify) { v = std::min(v, cap_of(c->b_cint, 20, 8)); v = std::min(v, cap_of(c->b_csig, 16 * (size_t)limbs, 8)); }
    k->node = v;
    k->cb = cap_of(c->b_cbits, 8, 8);
    k->win = std::min(cap_of(c->b_win, sizeof(DWin), 8), cap_of(c->b_wrole, sizeof(uint2) * WIN_ROLES, 8));
    k->edge = std::min(cap_of(c->b_esrc, 4, 1), cap_of(c->b_ew, 8, 1));
    k->limbs = limbs;
    k->flags = (c->force_global_sssp ? 1 : 0) | (c->no_wave ? 2 : 0) | (c->learning ? 4 : 0);
}

void fill_batch(phx_ctx *c, DBatch *b) {
    memset(b, 0, sizeof(*b));
    b->n_contig = c->n;
    b->mean_len = c->n > 0 ? c->totalL / c->n : 0;
    b->meta = (DMeta *)c->b_meta.p;
    b->tot = (DTotals *)c->b_tot.p;
    b->lpart = (int64_t *)c->b_lpart.p;
    b->sord = sssp_ordered(c) ? (int32_t *)c->b_sord.p : nullptr;
    b->res = (DRes *)c->b_res.p;
    current_caps(c, &b->caps);
    b->params = c->d_params;
    b->recs = (uint32_t *)c->b_recs.p;
    b->voff = (const uint32_t *)c->b_tiles.p; b->wfirst = b->voff + (size_t)c->n + 1;
    b->vtotal = c->vtotal; b->defcod = c->defcod ? 1 : 0;
    b->nbits = (uint64_t *)c->b_nbits.p; b->nbase = (uint32_t *)c->b_nbase.p; b->cbits = (uint64_t *)c->b_cbits.p;
    b->bits = (uint64_t *)c->b_bits.p; b->item = (uint2 *)c->b_item.p; b->iprev = (int32_t *)c->b_iprev.p;
    b->cpre = (uint32_t *)c->b_cpre.p; b->bpre = (uint32_t *)c->b_bpre.p;
    b->bridge = (DBridge *)c->b_bridge.p;
    if (c->has_trna) { b->tnode = (const DTNode *)c->b_tnode.p; b->tedge = (const DTEdge *)c->b_tedge.p; b->tnid = (int32_t *)c->b_tnid.p; b->tbits = (uint64_t *)c->b_tbits.p; }
    b->orf = (DOrf *)c->b_orf.p; b->grp = (DGrp *)c->b_grp.p;
    b->ostat = (DOrfStat *)c->b_ostat.p; b->oweight = (double *)c->b_oweight.p; b->owi = (long long *)c->b_owi.p; b->oflag = (uint8_t *)c->b_oflag.p; b->onode = (int32_t *)c->b_onode.p;
    b->node = (DNode *)c->b_node.p; b->parent = (int32_t *)c->b_parent.p;
    b->in_off = (uint32_t *)c->b_inoff.p;
    b->no = (double *)c->b_no.p;
    b->npos = (int32_t *)c->b_npos.p;
    b->ehit = (uint64_t *)c->b_ehit.p;
    b->erank = (int4 *)c->b_erank.p;
    b->mreach = (uint32_t *)c->b_mreach.p;
    b->olist = (int32_t *)c->b_olist.p;
    b->win = (DWin *)c->b_win.p; b->wrole = (uint2 *)c->b_wrole.p;
    b->swin = (DWin *)c->b_swin.p; b->swrole = (uint2 *)c->b_swrole.p; b->sdist = (uint64_t *)c->b_sdist.p; b->segw = (int32_t *)c->b_segw.p;
    b->sdist_nodes = b->caps.node; b->seg = 0; b->seg_margin_bp = c->seg_margin_bp;
    { const size_t want = (size_t)seg_cap(c) * (size_t)c->n * 8, have = c->b_segw.p ? c->b_segw.cap / 4 : 0; b->segw_ints = (int32_t)std::min(want, have); if (!b->segw_ints) b->segw = nullptr; }
    b->dist = (uint64_t *)c->b_dist.p;
    b->dist_stride = c->n_limbs;
    b->esrc = (uint32_t *)c->b_esrc.p; b->ew = (long long *)c->b_ew.p; b->ewl = nullptr; b->ekey = nullptr;
    b->esrcf = nullptr; b->ewf = nullptr;
    b->gtab = (long long *)c->b_gtab.p;
    b->gtabf = b->gtab ? (uint16_t *)(b->gtab + ((size_t)c->n + 1) * GT_N) : nullptr;
    b->gap_code = (c->duo && c->n <= c->n_simd) ? 1 : 0; // coded gap edges (4 bytes, no weight) where the solver of the batch's 128-bit contigs is k_sssp_duo, which reads the gap table (DBatch.duo, enqueue_run)
    b->tie = (uint8_t *)c->b_tie.p; b->tie_cap = cap_of(c->b_tie, 1, 0);
    b->cint = (int32_t *)c->b_cint.p; b->csig = (uint64_t *)c->b_csig.p; b->cert_scale = c->cert_scale;
    b->eref = (DERef *)c->b_eref.p;
    b->path = (int32_t *)c->b_path.p;
    b->genes = (DGene *)c->b_genes.p;
    b->front_spins = c->front_spins;
    b->gpack = gene_pack(c) ? 1 : 0;
    b->genes_c = b->genes ? b->genes + gene_half(c) : nullptr;
    b->gene_total = (uint32_t *)c->b_gtot.p;
}

int settle(phx_ctx *c); // brings a run enqueued by phx_run_async to its end (below)
// the device's per-contig records, for the taps (a run only brings DRes over)
int fetch_meta(phx_ctx *c) {
    if (!c->meta_stale || c->n == 0) return PHX_OK;
    HIPCHK(c, hipMemcpy(c->meta.data(), c->b_meta.p, sizeof(DMeta) * (size_t)c->n, hipMemcpyDeviceToHost));
    c->meta_stale = false;
    return PHX_OK;
}

int set_batch_layout(phx_ctx *c, int32_t n, const int64_t *len_or_null, const int64_t *offsets_or_null) {
    if (c->layout_pending) { HIPCHK(c, hipEventSynchronize(c->ev_layout)); c->layout_pending = false; } // push_layout's copies read the records rewritten below
    c->eager_done = false; c->trna_clean = true;
    c->seg_off = false; c->seg_clean = false; // (segments get their chance on every new batch)
    c->uploaded = false; c->ran = false; c->graph_valid = false; c->n = 0; // whatever fails below leaves the context without a batch
    c->meta_stale = false;
    c->has_trna = false; c->h_tnode.clear();
    if (n < 0) return PHX_E_ARG;
    if (!c->meta.assign((size_t)n)) { c->err = "hipHostMalloc failed"; return PHX_E_NOMEM; }
    c->ftab.clear(); c->vtotal = 0;
    c->graph_valid = false; c->tiles_dirty = true; c->meta0_dirty = true; c->runs_on_layout = 0;
    c->max_len = 0;
    int64_t off = 0, words = 0, items = 0, nbw = 0, nbr = 0, recs = 1; // (record 0 is a pad)
    c->ftab.reserve((size_t)n + 1);
    for (int i = 0; i < n; i++) {
        int64_t L = len_or_null ? len_or_null[i] : offsets_or_null[i + 1] - offsets_or_null[i];
        if (L < 0 || L > 0x7ffffff0ll) return PHX_E_ARG;
        DMeta &m = c->meta[(size_t)i];
        c->max_len = std::max<int64_t>(c->max_len, L);
        m.off = offsets_or_null ? offsets_or_null[i] : off;
        m.L = (int32_t)L;
        const int nt = (int)((L + PHX_TILE - 1) / PHX_TILE);
        m.nw = 8 * nt; // bitmap words per (plane, frame): whole tiles of 1536 positions (the readers take 8-word groups)
        m.rec_off = recs; // 2 nw records of 96 positions and one all-outside record behind them
        c->ftab.push_back((uint32_t)(recs - 1));
        recs += 2 * (int64_t)m.nw + 1;
        if (recs > 0x7ffffff0ll) return PHX_E_ARG;
        m.bits_off = words; m.item_off = items; m.nbits_off = nbw;
        m.bridge_off = nbr; m.bridge_cap = (int32_t)(L / 500 + 2);
        nbr += m.bridge_cap;
        nbw += 9 * (int64_t)m.nw;
        words += PHX_BITMAP_WORDS_PER_NW * (int64_t)m.nw; items += 6 * (int64_t)m.nw;
        off += L;
        off = (off + 15) & ~(int64_t)15; // 16-byte aligned rows let the feature kernel store uint4
    }
    c->vtotal = (uint32_t)(recs - 1);
    c->ftab.push_back(c->vtotal);
    { // per 62 virtual words: the contig of virtual word max(62 blk - 1, 0)
        int ci = 0;
        for (int64_t blk = 0; blk * 62 < (int64_t)c->vtotal + 62; blk++) {
            const int64_t v = blk * 62 > 0 ? blk * 62 - 1 : 0;
            while (ci + 1 < n && v >= (int64_t)c->ftab[(size_t)ci + 1]) ci++;
            c->ftab.push_back((uint32_t)ci);
        }
    }
    c->tot_words = words; c->tot_items = items; c->tot_nbits = nbw; c->tot_bridge = nbr;
    c->totalL = offsets_or_null ? offsets_or_null[n] : off;
    c->n = n; // the callers set `uploaded` once the bases are where the kernels read them
    return PHX_OK;
}

int ensure_position_buffers(phx_ctx *c) {
    int rc;
    if ((rc = ensure(c, c->b_recs, ((size_t)c->vtotal + 2) * 36))) return rc;
    if ((rc = ensure(c, c->b_nbits, (size_t)(c->tot_nbits + 8) * 8))) return rc;
    if ((rc = ensure(c, c->b_nbase, (size_t)(c->tot_nbits / 3 + 8) * 4))) return rc;
    if ((rc = ensure(c, c->b_meta, sizeof(DMeta) * (size_t)(c->n + 1)))) return rc;
    if ((rc = ensure(c, c->b_tiles, 4 * (c->ftab.size() + 1)))) return rc;
    if ((rc = ensure(c, c->b_gtot, 64))) return rc;
    if ((rc = ensure(c, c->b_lpart, ((size_t)c->n / 256 + 2) * 32 + 512))) return rc; // (+ 48 time stamps of k_front in -DFRONT_PROFILE builds)
    if (sssp_ordered(c) && (rc = ensure(c, c->b_sord, (size_t)c->n * 4))) return rc;
    if ((rc = ensure(c, c->b_gtab, ((size_t)c->n + 1) * GT_N * 10))) return rc; // per contig: the weights of its coded gap edges (k_edges<true>)
    if ((rc = ensure(c, c->b_res, ((size_t)c->n + 1) * sizeof(DRes) + sizeof(DTotals)))) return rc; // (k_results appends the totals: one copy brings both to the host)
    if (c->res_cap < (size_t)c->n + 1) {
        if (c->res) (void)hipHostFree(c->res);
        c->res = nullptr; c->res_cap = 0;
        const size_t want = (size_t)c->n + (size_t)c->n / 4 + 16;
        if (hipHostMalloc((void **)&c->res, want * sizeof(DRes) + sizeof(DTotals), hipHostMallocDefault) != hipSuccess) { c->res = nullptr; c->err = "hipHostMalloc failed"; return PHX_E_NOMEM; }
        c->res_cap = want;
    }
    if ((rc = ensure(c, c->b_bits, (size_t)(c->tot_words + 8) * 8))) return rc;
    { // prefix popcounts (k_bit_prefix): of the class bitmaps one record per PHX_PRE_G words — 6 planes of nw / G + 1 records of 32 bytes —, of the bases 2 nw + 2 entries of 16 bytes per contig
        const size_t W = (size_t)(c->tot_words / PHX_BITMAP_WORDS_PER_NW);
        if ((rc = ensure(c, c->b_cpre, (6 * (W / PHX_PRE_G + (size_t)c->n) + 2) * 32))) return rc;
        if ((rc = ensure(c, c->b_bpre, (2 * W + 2 * (size_t)c->n + 4) * 16))) return rc; // per record of 96 positions: a, c, t, g in front of it
    }
    if ((rc = ensure(c, c->b_item, (size_t)(c->tot_items + 8) * 8))) return rc;
    if ((rc = ensure(c, c->b_iprev, (size_t)(c->tot_items + 8) * 4))) return rc;
    if ((rc = ensure(c, c->b_bridge, (size_t)(c->tot_bridge + 8) * sizeof(DBridge)))) return rc;
    return PHX_OK;
}

} // namespace

extern "C" {

int phx_version(void) { return PHX_VERSION; }

int phx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *phx_strerror(int code) {
    switch (code) {
    case PHX_OK: return "ok";
    case PHX_E_ARG: return "bad argument";
    case PHX_E_NODEVICE: return "no usable HIP device (libphx has no CPU path)";
    case PHX_E_HIP: return "HIP runtime error";
    case PHX_E_NOMEM: return "out of memory";
    case PHX_E_STATE: return "call sequence error";
    case PHX_E_PARAM: return "bad codon table / minlen";
    case PHX_E_IO: return "file could not be opened or read";
    case PHX_S_BADLETTER: return "letter outside the IUPAC nucleotide alphabet";
    case PHX_S_TOOSHORT: return "contig shorter than 6 bases";
    case PHX_S_PARALLEL: return "bridge edge duplicates a connect edge";
    case PHX_S_BADTRNA: return "a tRNA hit lies outside the contig";
    case PHX_S_OVERFLOW: return "integer path sums overflow";
    case PHX_S_LONGORF: return "an open reading frame of more than 65535 codons";
    case PHX_S_NEGCYCLE: return "relaxation did not converge";
    case PHX_S_NOPATH: return "target unreachable";
    default: return "unknown";
    }
}

const char *phx_last_error(const phx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void phx_default_params(phx_params *p) {
    memset(p, 0, sizeof(*p));
    p->minlen = 90;
    p->n_start = 3;
    strcpy(p->start[0], "atg"); strcpy(p->start[1], "gtg"); strcpy(p->start[2], "ttg");
    // file_handling.py:58-62: weights divided by their maximum
    p->start_w[0] = 0.85 / 0.85; p->start_w[1] = 0.10 / 0.85; p->start_w[2] = 0.05 / 0.85;
    strcpy(p->start_w_text[0], "0.85"); strcpy(p->start_w_text[1], "0.10"); strcpy(p->start_w_text[2], "0.05");
    p->n_stop = 3;
    strcpy(p->stop[0], "tag"); strcpy(p->stop[1], "tga"); strcpy(p->stop[2], "taa");
}

int phx_params_from_flags(const char *start_codons, const char *stop_codons, int32_t minlen, phx_params *p) {
    if (!p) return PHX_E_ARG;
    phx_default_params(p);
    p->minlen = minlen;
    auto lower3 = [](const std::string &t, char *out) {
        if (t.size() != 3) return false;
        for (int j = 0; j < 3; j++) { const char ch = (char)(t[(size_t)j] | 0x20); if (code_of(ch) < 0) return false; out[j] = ch; }
        out[3] = 0;
        return true;
    };
    auto split = [](const char *s) { std::vector<std::string> v; std::string cur; for (; *s; s++) { if (*s == ',') { v.push_back(cur); cur.clear(); } else cur.push_back(*s); } v.push_back(cur); return v; };
    if (start_codons) {
        p->n_start = 0;
        memset(p->start, 0, sizeof(p->start)); memset(p->start_w, 0, sizeof(p->start_w)); memset(p->start_w_text, 0, sizeof(p->start_w_text));
        double wv[PHX_MAX_CODONS];
        for (const std::string &item : split(start_codons)) {
            const size_t colon = item.find(':');
            if (colon == std::string::npos || item.find(':', colon + 1) != std::string::npos) return PHX_E_PARAM;
            char cod[4];
            if (!lower3(item.substr(0, colon), cod)) return PHX_E_PARAM;
            const std::string wt = item.substr(colon + 1);
            if (wt.empty() || wt.size() >= sizeof(p->start_w_text[0])) return PHX_E_PARAM;
            char *end = nullptr;
            const double w = phx_strtod_c(wt.c_str(), &end); // (the "C" locale whatever the host application's LC_NUMERIC is)
            if (!end || *end || !(w == w)) return PHX_E_PARAM;
            int at = -1;
            for (int i = 0; i < p->n_start; i++) if (!strcmp(p->start[i], cod)) at = i; // dict semantics: first place, last weight
            if (at < 0) { if (p->n_start >= PHX_MAX_CODONS) return PHX_E_PARAM; at = p->n_start++; strcpy(p->start[at], cod); }
            strcpy(p->start_w_text[at], wt.c_str());
            wv[at] = w;
        }
        if (p->n_start < 1) return PHX_E_PARAM;
        double mx = wv[0];
        for (int i = 1; i < p->n_start; i++) mx = wv[i] > mx ? wv[i] : mx;
        for (int i = 0; i < p->n_start; i++) p->start_w[i] = wv[i] / mx; // file_handling.py:58-62
    }
    if (stop_codons) {
        p->n_stop = 0;
        memset(p->stop, 0, sizeof(p->stop));
        for (const std::string &item : split(stop_codons)) {
            if (p->n_stop >= PHX_MAX_CODONS) return PHX_E_PARAM;
            if (!lower3(item, p->stop[p->n_stop])) return PHX_E_PARAM;
            p->n_stop++;
        }
    }
    return check_params(p);
}

int phx_create(const phx_params *params, int device, void *stream, phx_ctx **out) { return phx_create_ex(params, device, stream, stream ? PHX_CREATE_USE_STREAM : 0u, out); }

int phx_create_ex(const phx_params *params, int device, void *stream, uint32_t flags, phx_ctx **out) {
    if (!out || (flags & ~(PHX_CREATE_USE_STREAM | PHX_CREATE_NO_GRAPH | PHX_CREATE_SIZE_EVERY_RUN | PHX_CREATE_SOLVER_GLOBAL | PHX_CREATE_SOLVER_NO_WAVE | PHX_CREATE_NO_CERTIFY | PHX_CREATE_CERT_TIGHT | PHX_CREATE_CERT_WIDE | PHX_CREATE_POISON | PHX_CREATE_ONE_STREAM | PHX_CREATE_NO_EXACT | PHX_CREATE_NO_FUSE | PHX_CREATE_NO_DUO | PHX_CREATE_NO_SEG)) || (stream && !(flags & PHX_CREATE_USE_STREAM))) return PHX_E_ARG;
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_error = e != hipSuccess ? hipGetErrorString(e) : "device index out of range";
        return PHX_E_NODEVICE;
    }
    phx_ctx *c = new phx_ctx();
    c->device = device;
    c->params = *params;
    // what a context does is a matter of its arguments alone (no environment variable changes it): solver kernel selection, no HIP
    // graph, sizing mode every run are flags of phx_create_ex, used by tools/ and the tests that force a solver kernel
    c->force_global_sssp = (flags & PHX_CREATE_SOLVER_GLOBAL) != 0;
    c->no_wave = (flags & PHX_CREATE_SOLVER_NO_WAVE) != 0;
    c->graphs_enabled = (flags & PHX_CREATE_NO_GRAPH) == 0;
    c->always_sync = (flags & PHX_CREATE_SIZE_EVERY_RUN) != 0;
    c->certify = (flags & PHX_CREATE_NO_CERTIFY) == 0;
    c->exact = (flags & PHX_CREATE_NO_EXACT) == 0;
    c->no_fuse = (flags & PHX_CREATE_NO_FUSE) != 0;
    { const char *e = getenv("PHX_FRONT_SPINS"); if (e && *e) c->front_spins = atoi(e); }
    { const char *e = getenv("PHX_NO_EAGER_CERT"); c->eager_cert = !(e && e[0] == '1'); }
    { const char *e = getenv("PHX_NO_ORF_ROWS"); c->orf_rows = !(e && e[0] == '1'); }
    { const char *e = getenv("PHX_ORF_STREAM"); if (e && *e) { const int v = atoi(e); if (v >= -1 && v <= 3) c->orf_stream = v; } }
    { const char *e = getenv("PHX_NO_SIDE_SCORE"); c->side_score = !(e && e[0] == '1'); }
    { const char *e = getenv("PHX_NO_DUO"); c->duo = !(flags & PHX_CREATE_NO_DUO) && !(e && e[0] == '1'); }
    { const char *e = getenv("PHX_NO_SEG"); c->seg_on = !(flags & PHX_CREATE_NO_SEG) && !(e && e[0] == '1'); }
    { const char *e = getenv("PHX_SEG_MAX_N"); if (e && atoi(e) >= 0) c->seg_max_n = atoi(e); }
    { const char *e = getenv("PHX_SEG_MARGIN_BP"); if (e && atoi(e) > 0) c->seg_margin_bp = atoi(e); }
    c->cert_wide = (flags & PHX_CREATE_CERT_WIDE) != 0;
    c->poison = (flags & PHX_CREATE_POISON) != 0;
    c->cert_scale = (flags & PHX_CREATE_CERT_TIGHT) ? 68719476736.0 : 1.0; // 2^36
    auto fail = [&](int code) { g_create_error = c->err; phx_destroy(c); return code; };
    if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return fail(PHX_E_NODEVICE); }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_simd = 4 * cus; }
    if (flags & PHX_CREATE_USE_STREAM) { // as given; a null handle is HIP's null stream
        c->stream = (hipStream_t)stream; c->own_stream = false;
        if (!stream) c->graphs_enabled = false; // the legacy stream cannot be captured
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(PHX_E_HIP); }
        c->own_stream = true;
    }
    for (int a = 0; a < 4; a++) {
        if (flags & PHX_CREATE_ONE_STREAM) { c->aux[a] = c->stream; c->one_stream = true; } // test switch: nothing runs side by side
        else if (hipStreamCreateWithFlags(&c->aux[a], hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(PHX_E_HIP); }
        if (hipEventCreateWithFlags(&c->ev_join[a], hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(PHX_E_HIP); }
    }
    if (hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_layout, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_piece[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_piece[1], hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(PHX_E_HIP); }
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork_plan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork_nodes, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join_nodes, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork_pre, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join_pre, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork_orf, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join_orf, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork_score, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join_score, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(PHX_E_HIP); }
    DParams dp;
    build_dparams(params, &dp);
    { // the reference's default codon tables (file_handling.py:51-53): k_features then uses their formulas instead of the table walk
        phx_params d;
        phx_default_params(&d);
        DParams dd;
        build_dparams(&d, &dd);
        c->defcod = memcmp(dd.cls_tab, dp.cls_tab, sizeof dp.cls_tab) == 0; // (classes only: weights and minlen do not enter k_features)
    }
    if (hipMalloc((void **)&c->d_params, sizeof(DParams)) != hipSuccess) { c->err = "hipMalloc failed"; return fail(PHX_E_NOMEM); }
    if (hipMemcpy(c->d_params, &dp, sizeof(dp), hipMemcpyHostToDevice) != hipSuccess) { c->err = "hipMemcpy failed"; return fail(PHX_E_HIP); }
    *out = c;
    return PHX_OK;
}

void phx_destroy(phx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->in_flight = false;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->b_eref, &c->b_cint, &c->b_csig, &c->b_meta0, &c->b_tie, &c->b_ekey, &c->b_tnode, &c->b_tedge, &c->b_tnid, &c->b_tbits, &c->b_win, &c->b_wrole, &c->b_bridge, &c->b_recs, &c->b_meta, &c->b_tiles, &c->b_nbits, &c->b_nbase, &c->b_cbits, &c->b_orf, &c->b_ostat, &c->b_oweight, &c->b_owi, &c->b_oflag, &c->b_ewf, &c->b_esrcf, &c->b_onode, &c->b_grp, &c->b_bits, &c->b_cpre, &c->b_bpre, &c->b_item, &c->b_iprev,
                     &c->b_node, &c->b_parent, &c->b_inoff, &c->b_no, &c->b_npos, &c->b_ehit, &c->b_mreach, &c->b_olist, &c->b_dist, &c->b_esrc, &c->b_ew, &c->b_ewl, &c->b_path, &c->b_genes, &c->b_gtot, &c->b_tot, &c->b_lpart, &c->b_res, &c->b_sord, &c->b_gtab, &c->b_erank, &c->b_swin, &c->b_swrole, &c->b_sdist, &c->b_segw};
    for (DevBuf *b : all) release(*b);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->h_tot) (void)hipHostFree(c->h_tot);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_genes) (void)hipHostFree(c->h_genes);
    c->meta.release();
    if (c->res) (void)hipHostFree(c->res);
    c->res = nullptr; c->res_cap = 0;
    collect_timers(c);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (int a = 0; a < 4; a++) { if (c->aux[a] && !c->one_stream) (void)hipStreamDestroy(c->aux[a]); if (c->ev_join[a]) (void)hipEventDestroy(c->ev_join[a]); }
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->ev_layout) (void)hipEventDestroy(c->ev_layout);
    for (int k = 0; k < 2; k++) if (c->ev_piece[k]) (void)hipEventDestroy(c->ev_piece[k]);
    if (c->h_tiles) (void)hipHostFree(c->h_tiles);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_fork_plan) (void)hipEventDestroy(c->ev_fork_plan);
    if (c->ev_fork_nodes) (void)hipEventDestroy(c->ev_fork_nodes);
    if (c->ev_join_nodes) (void)hipEventDestroy(c->ev_join_nodes);
    if (c->ev_fork_pre) (void)hipEventDestroy(c->ev_fork_pre);
    if (c->ev_fork_orf) (void)hipEventDestroy(c->ev_fork_orf);
    if (c->ev_fork_score) (void)hipEventDestroy(c->ev_fork_score);
    if (c->ev_join_score) (void)hipEventDestroy(c->ev_join_score);
    if (c->ev_join_orf) (void)hipEventDestroy(c->ev_join_orf);
    if (c->ev_join_pre) (void)hipEventDestroy(c->ev_join_pre);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#ifndef PHX_PLAN_STREAM_MAX
#define PHX_PLAN_STREAM_MAX 800 // batches of up to this many contigs: k_sssp_wave<2,0> runs beside k_wave_plan<2,0> (run_pipeline).  Beyond, the planner is done before the edge fill is (1000 x 50 kb: 0.35 against 0.43 ms) and there is nothing to gain (measured with 1024: +0.7 % on the step, +2 % on two batches in flight)
#endif
#ifndef PHX_UPLOAD_THREADS
#define PHX_UPLOAD_THREADS 16
#endif
#ifndef PHX_UPLOAD_PIECE
#define PHX_UPLOAD_PIECE (4 << 20)
#endif
namespace { int push_layout(phx_ctx *c); }

int phx_upload(phx_ctx *c, int32_t n, const char *const *seq, const int64_t *len) {
    if (!c || n < 0 || (n > 0 && (!seq || !len))) return PHX_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_flight) (void)settle(c); // a run still in flight reads the buffers this call replaces
    if (c->upload_pending) { HIPCHK(c, hipEventSynchronize(c->ev_upload)); c->upload_pending = false; } // the previous batch's copies read the staging memory
    c->attached = nullptr;
    int rc = set_batch_layout(c, n, len, nullptr);
    if (rc) return rc;
    // The letters cross the link as the records the kernels read (phx_pack_planes: residue-split bit planes, 36 bytes per 96 bases; the
    // records of a contig are followed by its all-outside spacer; records 0 and vtotal + 1 are pads): the staging pass has to touch
    // every letter anyway and writes — and PCIe moves — three bits per base.
    const size_t T = ((size_t)c->vtotal + 2) * 36;
    if ((rc = ensure(c, c->b_recs, T))) return rc;
    if (c->h_stage_cap < T) {
        if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr; c->h_stage_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_stage, T + T / 8, hipHostMallocDefault));
        c->h_stage_cap = T + T / 8;
    }
    // Packed into pinned memory and sent in pieces of >= 4 Mbases (whole contigs; 2 MB copies reach ~45 of the link's 56 GB/s), so
    // that the DMA of one piece overlaps the packing of the next.  Large batches are packed by the context's worker threads, all of
    // them on the earliest unfinished piece (items of <= 96 Kbases, taken in order), so that the first copy starts after 1/threads
    // of a piece's packing time; the calling thread enqueues each piece as soon as its last item is done.
    StageTimer t(c, ST_COPY);
    // k_features needs nothing but the bases: it is launched piece by piece behind the copies, on a side stream, so that the link and
    // the kernel work side by side and the run that follows starts at the ORF scan.  (The accumulators it adds to are reset here, as
    // the head of a run does; a run that is repeated on this batch — or retried with other buffer sizes — does all of it again.)
    bool eager = c->n > 0 && c->vtotal > 0 && c->aux[1] && !c->one_stream && !c->prof;
    DBatch fb;
    if (eager) {
        if ((rc = ensure_position_buffers(c))) return rc;
        if ((rc = ensure(c, c->b_tot, sizeof(DTotals)))) return rc;
        if ((rc = ensure(c, c->b_meta0, sizeof(DMeta) * (size_t)(c->n + 1)))) return rc;
        if ((rc = push_layout(c))) return rc;
        hipStream_t s0 = c->stream;
        fill_batch(c, &fb);
        phxk_reset(&fb, c->b_meta0.p, (unsigned long long)(c->tot_nbits + 8), 0ull, s0); // (a new upload has no tRNA hits yet)
    }
    int n_sent = 0;
    struct Piece { int i0, i1; int64_t r0, r1; int items; };              // contigs [i0, i1) = records [r0, r1)
    struct Item { const char *in; int64_t n; uint32_t *out; int64_t nrec; int piece; };
    std::vector<Piece> pieces;
    std::vector<Item> items;
    uint32_t *stage = (uint32_t *)c->h_stage;
    {
        const int64_t piece = PHX_UPLOAD_PIECE / 96, chunk = 96 * 1024; // (records; letters)
        int64_t sent = 0; int first = 0; // record 0, the pad, goes with the first piece
        for (int i = 0; i < n; i++) {
            const int64_t r0 = c->meta[(size_t)i].rec_off, nrec = 2 * (int64_t)c->meta[(size_t)i].nw + 1;
            const int64_t end = i + 1 < n ? r0 + nrec : (int64_t)c->vtotal + 2; // (the pad behind the last record goes with the last piece)
            for (int64_t o = 0; o < len[i] || o == 0; o += chunk) {
                const bool last = o + chunk >= len[i];
                items.push_back(Item{seq[i] + o, std::min(chunk, len[i] - o), stage + (size_t)(r0 + o / 96) * 9, last ? nrec - o / 96 : chunk / 96, (int)pieces.size()});
            }
            if (end - sent >= piece || i + 1 == n) {
                int cnt = 0;
                for (size_t k = items.size(); k > 0 && items[k - 1].piece == (int)pieces.size(); k--) cnt++;
                pieces.push_back(Piece{first, i + 1, sent, end, cnt}); sent = end; first = i + 1;
            }
        }
        for (int k = 0; k < 9; k++) { stage[k] = k % 3 == 2 ? ~0u : 0u; stage[((size_t)c->vtotal + 1) * 9 + (size_t)k] = stage[k]; } // the two pads
    }
    auto send_piece = [&](const Piece &pc) {
        const size_t b0 = (size_t)pc.r0 * 36, b1 = (size_t)pc.r1 * 36;
        hipError_t e = hipMemcpyAsync((char *)c->b_recs.p + b0, (char *)c->h_stage + b0, b1 - b0, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && eager) { // this piece's records, as soon as its bases have landed
            hipEvent_t ev = c->ev_piece[n_sent & 1];
            const uint32_t v0 = c->ftab[(size_t)pc.i0], v1 = c->ftab[(size_t)pc.i1];
            if ((e = hipEventRecord(ev, c->stream)) == hipSuccess && (e = hipStreamWaitEvent(c->aux[1], ev, 0)) == hipSuccess && v1 > v0)
                phxk_features(&fb, v0, v1, c->aux[1]);
        }
        n_sent++;
        return e;
    };
    const int np = (int)pieces.size(), ni = (int)items.size();
    int nthreads = 0;
    if (c->totalL >= (2 << 20)) {
        const int hw = (int)std::thread::hardware_concurrency();
        nthreads = std::min(std::max(1, ni / 4), std::max(1, std::min(PHX_UPLOAD_THREADS, hw / 4)));
        if (!c->pool) c->pool.reset(new (std::nothrow) StagePool());
        if (c->pool) c->pool->grow(nthreads);
        nthreads = c->pool ? (int)c->pool->th.size() : 0; // no thread to be had: the calling thread packs everything
    }
    if (nthreads == 0) {
        int k = 0;
        for (const Piece &pc : pieces) {
            for (int e = k + pc.items; k < e; k++) phx_pack_planes(items[(size_t)k].in, items[(size_t)k].n, items[(size_t)k].out, items[(size_t)k].nrec);
            HIPCHK(c, send_piece(pc));
        }
    } else {
        std::atomic<int> next(0);
        std::unique_ptr<std::atomic<int>[]> left(new std::atomic<int>[(size_t)np]); // items of the piece not packed yet
        for (int k = 0; k < np; k++) left[(size_t)k].store(pieces[(size_t)k].items, std::memory_order_relaxed);
        const std::function<void()> work = [&]() {
            for (int k; (k = next.fetch_add(1)) < ni;) {
                const Item &it = items[(size_t)k];
                phx_pack_planes(it.in, it.n, it.out, it.nrec);
                left[(size_t)it.piece].fetch_sub(1, std::memory_order_acq_rel);
            }
        };
        c->pool->start(&work);
        hipError_t err = hipSuccess;
        for (int k = 0; k < np; k++) {
            while (left[(size_t)k].load(std::memory_order_acquire) > 0) std::this_thread::yield();
            if (err == hipSuccess) err = send_piece(pieces[(size_t)k]);
        }
        c->pool->finish(); // (the workers hold references to this frame)
        HIPCHK(c, err);
    }
    if (eager) { // the run follows the last k_features launch
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev_piece[0], c->aux[1]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_piece[0], 0));
        c->eager_done = true;
    }
    // No wait here: the letters are staged (the caller's strings are free again), the copies are ordered before the run on the
    // context's stream, and the next phx_upload waits for this event before it touches the staging memory.
    HIPCHK(c, hipEventRecord(c->ev_upload, c->stream));
    c->upload_pending = true;
    c->uploaded = true;
    return PHX_OK;
}

int phx_set_trnas(phx_ctx *c, const int64_t *offsets, const int32_t *start, const int32_t *stop) {
    if (!c) return PHX_E_ARG;
    if (!c->uploaded) return PHX_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_flight) (void)settle(c);
    if (!offsets && c->trna_clean) return PHX_OK; // no hits before, none now: the batch is as phx_upload left it
    c->trna_clean = false; c->eager_done = false; // (the records change: the run resets the accumulators and runs k_features itself)
    c->ran = false; c->graph_valid = false; c->meta0_dirty = true; c->runs_on_layout = 0;
    c->has_trna = false; c->h_tnode.clear();
    for (DMeta &m : c->meta) { m.n_tnode = 0; m.n_tedge = 0; m.tn_off = 0; m.te_off = 0; if (m.status == PHX_S_PARALLEL || m.status == PHX_S_BADTRNA) m.status = 0; }
    if (!offsets) return PHX_OK; // no tRNA finder: the graph is built without add_trnas' part (functions.py:493-495)
    const int n = c->n;
    if (offsets[0] != 0) return PHX_E_ARG;
    for (int i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return PHX_E_ARG;
    if (offsets[n] > 0 && (!start || !stop)) return PHX_E_ARG;
    std::vector<DTNode> tn;
    std::vector<DTEdge> te;
    for (int i = 0; i < n; i++) {
        DMeta &m = c->meta[(size_t)i];
        m.tn_off = (int64_t)tn.size(); m.te_off = (int64_t)te.size();
        const size_t n0 = tn.size(), e0 = te.size();
        std::vector<std::pair<int32_t, int32_t>> other; // other_end['t' + str(pos)]: last writer wins (functions.py:502-508)
        auto set_other = [&](int32_t pos, int32_t v) { for (auto &o : other) if (o.first == pos) { o.second = v; return; } other.push_back({pos, v}); };
        auto node = [&](int type, int frame, int32_t pos) -> int32_t { // Graph.add_node: an existing node is reused (nodes.py:7-12: identity = repr)
            const int32_t info = NINFO(type, frame);
            for (size_t k = n0; k < tn.size(); k++) if (tn[k].pos == pos && tn[k].info == info) return (int32_t)(k - n0);
            tn.push_back(DTNode{pos, info, -1, (int32_t)(tn.size() - n0)});
            return (int32_t)(tn.size() - 1 - n0);
        };
        bool parallel = false, outside = false;
        for (int64_t k = offsets[i]; k < offsets[i + 1]; k++) {
            const int32_t a = start[k], z = stop[k];
            int32_t s, t;
            // both nodes of a hit must sit on a base of the contig (the node bitmaps cover 1..L); a hit that does not — a finder run
            // with a circular topology, a truncated parse — fails this contig, not the batch
            if (a < z) { // functions.py:499-503
                if (a < 1 || a > m.L || z - 2 < 1 || z - 2 > m.L) { outside = true; break; }
                s = node(0, 4, a); t = node(1, 4, z - 2);
                set_other(z - 2, a); set_other(a, z - 2);
            } else { // functions.py:504-508
                if (z < 1 || z > m.L || a - 2 < 1 || a - 2 > m.L) { outside = true; break; }
                s = node(1, -4, z); t = node(0, -4, a - 2);
                set_other(a - 2, z); set_other(z, a - 2);
            }
            for (size_t q = e0; q < te.size(); q++) if (te[q].src == s && te[q].dst == t) parallel = true;
            te.push_back(DTEdge{s, t});
        }
        for (size_t k = n0; k < tn.size(); k++) for (auto &o : other) if (o.first == tn[k].pos) tn[k].other = o.second;
        m.n_tnode = (int32_t)(tn.size() - n0); m.n_tedge = (int32_t)(te.size() - e0);
        if (outside) { tn.resize(n0); te.resize(e0); m.n_tnode = 0; m.n_tedge = -1; m.status = PHX_S_BADTRNA; }
        else if (parallel) { m.status = PHX_S_PARALLEL; m.n_tedge = -1; } // the reference raises (graphs.py:74); marked for run_once
    }
    if (tn.empty()) return PHX_OK; // finders ran and found nothing: same graph as without
    int rc;
    if ((rc = ensure(c, c->b_tnode, sizeof(DTNode) * (tn.size() + 1)))) return rc;
    if ((rc = ensure(c, c->b_tedge, sizeof(DTEdge) * (te.size() + 1)))) return rc;
    if ((rc = ensure(c, c->b_tnid, 4 * (tn.size() + 1)))) return rc;
    if ((rc = ensure(c, c->b_tbits, (size_t)(c->tot_nbits / 3 * 4 + 8) * 8))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->b_tnode.p, tn.data(), sizeof(DTNode) * tn.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b_tedge.p, te.data(), sizeof(DTEdge) * te.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_tnode.swap(tn);
    c->has_trna = true;
    return PHX_OK;
}

int phx_attach(phx_ctx *c, int32_t n, const void *d_ascii, const int64_t *offsets) {
    if (!c || n < 0 || !offsets || (n > 0 && !d_ascii)) return PHX_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_flight) (void)settle(c);
    int rc = set_batch_layout(c, n, nullptr, offsets);
    if (rc) return rc;
    c->attached = d_ascii; // the caller's letters as they are: the head of every run turns them into records (k_pack_planes)
    c->uploaded = true;
    return PHX_OK;
}

static double host_contig_pstop(uint32_t gc, int L);

// Segments (phx_sssp_seg.inc): how many per contig at most in this batch (0: none).  The chip has a place per SIMD for a wavefront pair:
// 32 segments x 32 contigs ... 2 x 512.
static int seg_cap(const phx_ctx *c) {
    if (!c->seg_on || c->seg_off || c->seg_never || !c->duo || c->force_global_sssp || c->no_wave || c->n < 1 || c->n > c->seg_max_n || sssp_ordered(c)) return 0;
    if (c->max_len < (int64_t)c->seg_margin_bp + 4000) return 0; // (no contig long enough for two segments: the one sweep, which follows its planner, is 1-4 % faster then)
    const int k = std::min(phxk_seg_kmax(), c->n_simd / c->n);
    return k >= 2 ? k : 0;
}
static bool seg_wanted(const phx_ctx *c) { return seg_cap(c) > 0; }
// their buffers, sized by what the context's node / window buffers hold (call after those have grown)
static int ensure_seg(phx_ctx *c) {
    if (!seg_wanted(c)) return PHX_OK;
    DCaps k;
    current_caps(c, &k);
    if (k.node <= 0 || k.win <= 0 || k.node > (1ll << 22)) return PHX_OK; // (beyond 4 M nodes the slices would take GBs: such batches keep one sweep per contig)
    const size_t KM = (size_t)seg_cap(c);
    int rc;
    if ((rc = ensure(c, c->b_swin, KM * (size_t)(k.win + 8) * sizeof(DWin)))) return rc;
    if ((rc = ensure(c, c->b_swrole, KM * (size_t)(k.win + 8) * sizeof(uint2) * WIN_ROLES))) return rc;
    if ((rc = ensure(c, c->b_sdist, KM * (size_t)(k.node + 8) * 16))) return rc;
    if ((rc = ensure(c, c->b_segw, KM * (size_t)(c->n + 1) * 32))) return rc;
    return PHX_OK;
}
// ... and whether they hold this run
static bool seg_ready(const phx_ctx *c, const DCaps &k) {
    if (!seg_wanted(c) || k.node <= 0 || k.node > (1ll << 22)) return false;
    const size_t KM = (size_t)seg_cap(c);
    return c->b_swin.p && c->b_swin.cap >= KM * (size_t)(k.win + 8) * sizeof(DWin) && c->b_swrole.p && c->b_swrole.cap >= KM * (size_t)(k.win + 8) * sizeof(uint2) * WIN_ROLES &&
           c->b_sdist.p && c->b_sdist.cap >= KM * (size_t)(k.node + 8) * 16 && c->b_segw.p && c->b_segw.cap >= KM * (size_t)(c->n + 1) * 32;
}

namespace {

const int kRetry = 1000; // run_once: a buffer was too small for this batch (or a solver class was not launched): run again, sizing as we go

// One pass over the whole path.  `learn`: read the device-side totals back after each layout kernel and size the buffers
// from them (two extra host round trips: first run of a context, or a batch that outgrew it).  Otherwise everything is
// enqueued at once against the buffers the context already has; the layout kernels flag a batch that does not fit, later
// kernels then do nothing, and the caller runs again with `learn`.
// Everything a run puts on the stream(s), from the accumulator resets to the copies of the results.  With `learn` it
// stops twice to read the device-side totals and size the buffers; otherwise it only enqueues (and can be captured
// into a HIP graph).  mask / lds: solver classes launched and the LDS given to k_sssp_lds per limb class.
int enqueue_run(phx_ctx *c, bool learn, int &mask, int64_t lds[4]) {
    int rc;
    // (sizing mode: the layout kernels do not flag totals beyond the capacities, the host reads the totals and grows the buffers;
    //  until round 3 they did and the host cleared the flag with a 4-byte hipMemsetAsync between the kernels — which later
    //  kernels did not always see when several processes shared the GPU: they skipped their work on a context's first run)
    struct Learning { phx_ctx *c; Learning(phx_ctx *c_, bool on) : c(c_) { c->learning = on; } ~Learning() { c->learning = false; } } learning_scope(c, learn);
    const int n = c->n;
    hipStream_t s = c->stream;
    DBatch b;
    const bool head_done = c->eager_now; // phx_upload reset the accumulators and ran k_features behind its copies
    fill_batch(c, &b);
    {
        StageTimer t(c, ST_MEMSET);
        const unsigned long long tw = c->has_trna ? (unsigned long long)(c->tot_nbits / 3 * 4 + 8) : 0ull;
        if (!head_done) phxk_reset(&b, c->b_meta0.p, (unsigned long long)(c->tot_nbits + 8), tw, s); // bitmaps, totals, per-contig accumulators: one launch
        else if (c->has_trna) HIPCHK(c, hipMemsetAsync(c->b_tbits.p, 0, (size_t)tw * 8, s));
    }
    fill_batch(c, &b);
    if (!head_done) {
        StageTimer t(c, ST_FEATURES);
        if (c->attached) phxk_pack_planes(&b, c->attached, s);
        phxk_features(&b, 0u, c->vtotal, s);
    }
    // A small batch in steady state: everything from the ORF count to the edge fill in ONE launch (k_front, phx_front.inc) instead of
    // two dozen launch-bound kernels.  (Not while the buffers are being sized — that needs the host between the stages — and not
    // with the stage timers on.)
    const uint32_t front_stages = (1u << ST_ORF_COUNT) | (1u << ST_ORF_EMIT) | (1u << ST_ORF_STATS) | (1u << ST_SCORE) | (1u << ST_NODES) | (1u << ST_EDGE_COUNT) | (1u << ST_EDGE_FILL);
    const bool fuse = !learn && !(c->prof && (c->prof_mask & front_stages)) && !c->no_fuse && !c->front_off && phxk_front_blocks_y(&b) > 0;
    DTotals *ht = c->h_tot;
    c->pend_front = fuse; // (stands for the replays of the graph this enqueue is captured into: finish_once counts the runs)
    if (fuse) {
        b.defer_overlap = c->max_len < (1 << 21) ? 1 : 0;
        phxk_front(&b, s);
        HIPCHK(c, hipGetLastError());
        mask = c->last_mask;
        for (int k = 0; k < 4; k++) lds[k] = c->last_lds[k];
    } else {
    // word-prefix popcounts of the bitmaps (for k_orf_stats) on a side stream, beside the ORF scan
    HIPCHK(c, hipEventRecord(c->ev_fork_pre, s));
    HIPCHK(c, hipStreamWaitEvent(c->aux[0], c->ev_fork_pre, 0));
    phxk_bit_prefix(&b, c->aux[0]);
    HIPCHK(c, hipEventRecord(c->ev_join_pre, c->aux[0]));
    { StageTimer t(c, ST_ORF_COUNT); phxk_orf_count(&b, s); phxk_layout1(&b, s); }
    HIPCHK(c, hipGetLastError());
    if (learn) { // sync #1: totals of ORFs / groups / nodes -> buffers
        { StageTimer t(c, ST_COPY); HIPCHK(c, hipMemcpyAsync(ht, c->b_tot.p, sizeof(DTotals), hipMemcpyDeviceToHost, s)); }
        HIPCHK(c, hipStreamSynchronize(s));
        const size_t NV = (size_t)ht->node + 8, G = (size_t)ht->grp + 1;
        if ((rc = ensure(c, c->b_cbits, (size_t)(ht->cb + 8) * 8))) return rc;
        if ((rc = ensure(c, c->b_orf, sizeof(DOrf) * (size_t)(ht->orf + 1)))) return rc;
        if ((rc = ensure(c, c->b_ostat, sizeof(DOrfStat) * (size_t)(ht->orf + 1)))) return rc;
        if ((rc = ensure(c, c->b_oweight, 8 * (size_t)(ht->orf + 1)))) return rc;
        if ((rc = ensure(c, c->b_owi, 8 * (size_t)(ht->orf + 1)))) return rc;
        if ((rc = ensure(c, c->b_oflag, (size_t)(ht->orf + 16)))) return rc;
        if ((rc = ensure(c, c->b_onode, 4 * (size_t)(ht->orf + 1)))) return rc;
        if ((rc = ensure(c, c->b_grp, sizeof(DGrp) * G))) return rc;
        if ((rc = ensure(c, c->b_genes, 2 * sizeof(DGene) * (G + c->h_tnode.size() + 8)))) return rc;
        if ((rc = ensure(c, c->b_node, NV * sizeof(DNode)))) return rc;
        for (DevBuf *q : {&c->b_parent, &c->b_path, &c->b_olist})
            if ((rc = ensure(c, *q, NV * 4))) return rc;
        if ((rc = ensure(c, c->b_inoff, (NV + (size_t)n + 1) * 4))) return rc;
        if ((rc = ensure(c, c->b_no, NV * 8))) return rc;
        if ((rc = ensure(c, c->b_npos, NV * 4))) return rc;
        if ((rc = ensure(c, c->b_ehit, NV * 8))) return rc;
        if ((rc = ensure(c, c->b_erank, NV * 16))) return rc;
        if ((rc = ensure(c, c->b_mreach, NV * 16))) return rc;
        if ((rc = ensure(c, c->b_dist, NV * 8 * (size_t)std::max(c->n_limbs, 2)))) return rc;
        if (c->certify) {
            if ((rc = ensure(c, c->b_cint, NV * 20))) return rc;
            if ((rc = ensure(c, c->b_csig, NV * 16 * (size_t)std::max(c->n_limbs, 2)))) return rc;
        }
        const size_t NW = NV / 16 + 8 * (size_t)n + 16; // window records of k_sssp_wave, see k_layout1
        if ((rc = ensure(c, c->b_win, NW * sizeof(DWin)))) return rc;
        if ((rc = ensure(c, c->b_wrole, NW * sizeof(uint2) * WIN_ROLES))) return rc;
        if ((rc = ensure_seg(c))) return rc;
    }
    fill_batch(c, &b);
    { StageTimer t(c, ST_ORF_EMIT); phxk_orf_emit(&b, s); }
    // nodes (coverage, ranks, records, order) beside the ORF statistics: both only read what k_orf<true> wrote
    HIPCHK(c, hipEventRecord(c->ev_fork_nodes, s));
    HIPCHK(c, hipStreamWaitEvent(c->aux[1], c->ev_fork_nodes, 0));
    phxk_nodes(&b, c->aux[1]);
    HIPCHK(c, hipEventRecord(c->ev_join_nodes, c->aux[1]));
    { StageTimer t(c, ST_ORF_STATS); HIPCHK(c, hipStreamWaitEvent(s, c->ev_join_pre, 0)); phxk_orf_stats(&b, s); }
    // the ORF weights are needed by k_layout2 (integer class) and by the edge fill, not by the node attributes or the edge count: beside them
    // (large batches only: a side stream costs an event round trip — 10 us of a lone genome's 330 —, and with more streams than hardware queues a side
    //  stream's kernel may sit behind another side stream's: behind the planner, which outlasts the edge fill below ~500 contigs — Lambda 0.34 -> 0.39 ms,
    //  64 x 50 kb 0.78 -> 0.91 with k_edges_orf on its side stream, profiles/r06_side_streams_small.txt)
    const bool big_batch = c->n >= 600;
    const bool side_score = c->side_score && big_batch && !c->one_stream && c->aux[0] && !learn;
    if (side_score) {
        HIPCHK(c, hipEventRecord(c->ev_fork_score, s));
        HIPCHK(c, hipStreamWaitEvent(c->aux[0], c->ev_fork_score, 0));
        { StageTimer t(c, ST_SCORE, c->aux[0]); phxk_score(&b, c->aux[0]); }
        HIPCHK(c, hipEventRecord(c->ev_join_score, c->aux[0]));
    } else { StageTimer t(c, ST_SCORE); phxk_score(&b, s); }
    { StageTimer t(c, ST_NODES); HIPCHK(c, hipStreamWaitEvent(s, c->ev_join_nodes, 0)); phxk_node_attr(&b, s); }
    { StageTimer t(c, ST_EDGE_COUNT); phxk_edges_count(&b, s); if (side_score) HIPCHK(c, hipStreamWaitEvent(s, c->ev_join_score, 0)); phxk_layout2(&b, s); }
    HIPCHK(c, hipGetLastError());
    mask = c->last_mask;
    for (int k = 0; k < 4; k++) lds[k] = c->last_lds[k];
    if (learn) { // sync #2: edge total, widest integer class, solver classes
        { StageTimer t(c, ST_COPY); HIPCHK(c, hipMemcpyAsync(ht, c->b_tot.p, sizeof(DTotals), hipMemcpyDeviceToHost, s)); }
        HIPCHK(c, hipStreamSynchronize(s));
        c->n_limbs = std::max(c->n_limbs, std::max(ht->nlmax, 2));
        if ((rc = ensure(c, c->b_esrc, (size_t)(ht->edge + 1) * 4))) return rc;
        if ((rc = ensure(c, c->b_ew, (size_t)(ht->edge + 1) * 8))) return rc;
        if ((rc = ensure(c, c->b_dist, ((size_t)ht->node + 8) * 8 * (size_t)c->n_limbs))) return rc;
        if (c->certify && (rc = ensure(c, c->b_csig, ((size_t)ht->node + 8) * 16 * (size_t)c->n_limbs))) return rc;
        if ((rc = ensure_seg(c))) return rc; // (b_dist may have grown: the node capacity with it)
        mask = ht->class_mask;
        if (mask & 4) mask |= 8; // k_wave_plan (still to run) may move 128-bit contigs to the wavefront kernel's roomy configuration: this run launches it in any case
        for (int k = 0; k < 4; k++) lds[k] = ht->lds_need[k];
    }
    } // (!fuse)
    fill_batch(c, &b);
    // A contig's planner (one wavefront walking all its windows: 0.14 ms for Lambda, 0.25 ms for T4, 0.2-0.35 ms in a batch) outlasts the
    // edge fill it runs beside unless the batch is large (0.06 / 0.09 ms for a lone contig, 0.2 ms for 512 contigs), so the solver used to
    // start that much late.  The tight 128-bit solver is launched right behind the edge fill instead and follows the planner's progress
    // counter (DMeta.plan_prog).  Every planner wavefront has been resident for the whole edge fill by then (both kernels fit the chip
    // side by side up to one contig per SIMD); a solver that sees no progress for 20 ms hands its contig to the workgroup kernel.
    // Which class follows its planner: the 128-bit contigs' tight configuration whenever the batch has such contigs; a batch that is ALL
    // 256-bit or all 512-bit contigs (a lone genome with a 2000+ codon ORF) streams that class's pair instead.
    int stream_k = -1;
    // (safe while every planner wavefront is resident beside the edge fill: bounded by the device's SIMD count, not by a constant;
    //  and a context that has seen one time-out — another context or process held the SIMDs — goes back to waiting for the planner)
    if (c->n <= std::min(PHX_PLAN_STREAM_MAX, c->n_simd * 3 / 4) && !c->plan_stream_off && !b.sord && !c->one_stream && c->aux[3]) {
        if ((mask >> 2) & 1) stream_k = 0;
        else for (int k = 1; k <= 2; k++) if (((mask >> (4 * k + 2)) & 1) && !(mask & 0xffff & ~(15 << (4 * k)))) stream_k = k;
    }
    // Small batches: the 128-bit contigs of the wavefront solver in up to 16 segments each, all at once (phx_sssp_seg.inc); their planner
    // wavefronts are short (a sixteenth of a contig), so the solver is launched behind them as in a large batch.
    b.seg = (((mask >> 2) & 1) && seg_ready(c, b.caps)) ? seg_cap(c) : 0;
    c->pend_seg = b.seg != 0; c->pend_seg_k = b.seg;
    if (b.seg && stream_k == 0) stream_k = -1;
    b.seg_nofb = (b.seg && c->seg_clean && !learn) ? 1 : 0;
    // behind k_front (up to 4 contigs) no edge fill is left to hide the segments' planner wavefronts: their solvers are launched beside them and
    // follow the window counts (DBatch.segw, cleared by k_reset at the head of this run); 128 pairs + 128 planner wavefronts are resident at once
    // (also behind the staged kernels of up to 4 contigs: Lambda's planner wavefronts take 50 us, the edge fill they run beside 38)
    b.seg_stream = (b.seg && !head_done && !learn && c->n <= 4 && !c->plan_stream_off && !c->one_stream && c->aux[3]) ? 1 : 0;
    const bool stream_plan = stream_k >= 0;
    // two wavefronts per contig while every contig has its pair of SIMD places at once (256 registers each: four workgroups per CU); beyond that the
    // one-wavefront kernel keeps 1280 contigs resident against 1024 and wins: 1250 contigs 2.10 -> 1.93 ms, 2500 3.68 -> 3.58, 10 000 13.7 -> 13.3
    b.duo = (c->duo && c->n <= c->n_simd) ? 1 : 0;
    b.plan_stream = stream_k < 0 ? 0 : (2 << stream_k); // the limb count of the class that streams (2, 4, 8)
    b.defer_overlap = c->max_len < (1 << 21) ? 1 : 0; // node ids fit 21 bits (a contig has fewer nodes than positions)
    auto launch_plan = [&]() -> int { // the windows of the wavefront solver need the node records and in-edge counts only
        HIPCHK(c, hipEventRecord(c->ev_fork_plan, s));
        HIPCHK(c, hipStreamWaitEvent(c->aux[3], c->ev_fork_plan, 0));
        {
            StageTimer t(c, ST_WAVE_PLAN, c->aux[3]); // (on the side stream: runs beside "edges_fill" / "sssp", not in the sum of the main stream's stages)
            phxk_wave_plan(&b, ((mask >> 6) & 1) | (((mask >> 10) & 1) << 1), c->aux[3]); // bits 4*1+2, 4*2+2: 256- / 512-bit contigs for the wavefront kernel
            if (b.sord) phxk_sssp_order(&b, c->aux[3]);
        }
        HIPCHK(c, hipEventRecord(c->ev_join[3], c->aux[3]));
        return PHX_OK;
    };
    if ((rc = launch_plan())) return rc; // beside the edge fill (started after it, beside the solver: the fill gains what the solver loses)
    // the rows of the close CDS nodes (the ORF edges) by a thread per ORF on a side stream, beside the neighbour scans of the open nodes
    b.orf_rows = (!fuse && c->orf_rows && c->n >= 600 && !c->one_stream && c->aux[2]) ? 1 : 0; // (large batches only, see k_score above; aux[1]: the node kernels' stream, idle by now — behind the planner's stream or on aux[2] the kernel ran AFTER the planner, profiles/r06_orf_stream.txt)
    const bool orf_side = b.orf_rows && c->orf_stream >= 0;
    if (orf_side) {
        hipStream_t so = c->aux[c->orf_stream];
        HIPCHK(c, hipEventRecord(c->ev_fork_orf, s));
        HIPCHK(c, hipStreamWaitEvent(so, c->ev_fork_orf, 0));
        phxk_edges_orf(&b, so);
        HIPCHK(c, hipEventRecord(c->ev_join_orf, so));
    }
    if (!fuse) { StageTimer t(c, ST_EDGE_FILL); if (b.orf_rows && !orf_side) phxk_edges_orf(&b, s); phxk_edges_fill(&b, s); if (orf_side) HIPCHK(c, hipStreamWaitEvent(s, c->ev_join_orf, 0)); }
    {
        // one stream per limb class that occurs in the batch (the classes are disjoint sets of contigs); within it the
        // wavefront kernel first, then the kernels it may hand contigs to
        StageTimer t(c, ST_SSSP);
        if (stream_plan) phxk_sssp(&b, 2 << stream_k, 2, (size_t)lds[stream_k], s); // beside the planner, see above
        if (b.seg_stream) phxk_sssp(&b, 2, 2, (size_t)lds[0], s);                   // the segments' solvers, beside their planners
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[3], 0)); // k_wave_plan: also decides which contigs the wavefront kernel takes
        const int nl_of[4] = {2, 4, 8, 17};
        int nlaunch = 0, nclass = 0;
        bool used[3] = {false, false, false};
        for (int k = 0; k < 4; k++) nclass += ((mask >> (4 * k)) & 15) ? 1 : 0;
        // contigs that never enter the wavefront kernel (too dense for its windows: k_edges<false>; a window the planner could not
        // lay out: k_wave_plan, which has finished by now) are solved by the workgroup kernel on a side stream, beside the
        // wavefront kernel; the launch after the wavefront kernel then only takes what that kernel handed back while it ran
        bool early = false;
        auto early_k = [&](int k) { return ((mask >> (4 * k + 2)) & 1) && ((mask >> (4 * k + 1)) & 1) && (((mask >> (16 + k)) | (mask >> (20 + k))) & 1); }; // such contigs existed in the run this one is modelled on
        for (int k = 0; k < 4; k++) early = early || early_k(k);
        early = early && c->aux[2];
        // the wavefront kernel's roomy configuration takes other contigs than the tight one (k_wave_plan decided): beside it, on the side stream
        const bool roomy_side = ((mask >> 3) & 1) && c->aux[2] && !c->one_stream;
        if ((nclass > 1 || early || roomy_side) && c->aux[0]) HIPCHK(c, hipEventRecord(c->ev_fork, s)); // fork point: before any of the launches
        if (early || roomy_side) {
            HIPCHK(c, hipStreamWaitEvent(c->aux[2], c->ev_fork, 0));
            used[2] = true;
            if (roomy_side) phxk_sssp(&b, 2, 3, 0, c->aux[2]);
            if (early)
                for (int k = 3; k >= 0; k--)
                    if (early_k(k)) phxk_sssp(&b, nl_of[k], 1, (size_t)lds[k], c->aux[2]);
            HIPCHK(c, hipEventRecord(c->ev_join[2], c->aux[2]));
        }
        for (int k = 3; k >= 0; k--) { // widest integers first: fewest contigs, longest per-contig time
            if (!((mask >> (4 * k)) & 15)) continue;
            hipStream_t st = s;
            if (nlaunch > 0 && c->aux[0]) {
                const int a = (nlaunch - 1) % 2;
                if (!used[a]) { HIPCHK(c, hipStreamWaitEvent(c->aux[a], c->ev_fork, 0)); used[a] = true; }
                st = c->aux[a];
            }
            for (int mode = 3; mode >= 0; mode--) // 3: the wavefront kernel's roomy configuration (few contigs, if any), 2: its tight one
                if (((mask >> (4 * k + mode)) & 1) && !(mode == 3 && roomy_side) && !(stream_plan && k == stream_k && mode == 2)) {
                    if ((early && mode == 1 && early_k(k)) || (roomy_side && k == 0 && mode <= 1)) HIPCHK(c, hipStreamWaitEvent(st, c->ev_join[2], 0)); // after the side launches: it skips what the workgroup kernel solved there, and takes what the roomy wavefront kernel handed back
                    if (!(b.seg_stream && k == 0 && mode == 2)) phxk_sssp(&b, nl_of[k], mode, (size_t)lds[k], st);
                    if (b.seg && k == 0 && mode == 2) { // the segments' solvers: join, prove, parents; then one sweep for the contigs that could not be proven
                        phxk_seg_merge(&b, c->last_vmax, st);
                        if (!b.seg_nofb) phxk_seg_fallback(&b, st);
                    }
                }
            nlaunch++;
        }
        for (int a = 0; a < 2; a++)
            if (used[a]) { HIPCHK(c, hipEventRecord(c->ev_join[a], c->aux[a])); HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[a], 0)); }
        if (early || roomy_side) HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[2], 0));
    }
    {
        StageTimer t(c, ST_INORDER);
        int nlm = 0;
        for (int k = 0; k < 4; k++) nlm |= ((mask >> (4 * k)) & 15) ? 1 << k : 0;
        phxk_inorder(&b, nlm, s);
        if (b.gpack) phxk_gene_pack(&b, s);
    } // equal-length alternatives: the parents of the reference's relaxation order
    HIPCHK(c, hipGetLastError());
    { // per-contig records (statuses, offsets, gene counts) and the totals
        StageTimer t(c, ST_COPY);
        phxk_results(&b, s);
        HIPCHK(c, hipMemcpyAsync(c->res, c->b_res.p, sizeof(DRes) * (size_t)n + sizeof(DTotals), hipMemcpyDeviceToHost, s)); // records + the totals k_results put behind them (finish_once)
    }
    return PHX_OK;
}

void drop_graph(phx_ctx *c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr; c->graph = nullptr; c->graph_valid = false;
}

// The batch layout on the device, once per layout: the records a run starts from (b_meta0: offsets and lengths set, accumulators
// zero) and the tile table of k_features.  Asynchronous, from pinned memory; ev_layout guards the host copies.
int push_layout(phx_ctx *c) {
    hipStream_t s = c->stream;
    bool pushed = false;
    if (c->meta0_dirty) { // once per batch layout: the records a run starts from (offsets and lengths set, accumulators zero)
        for (DMeta &m : c->meta) {
            DMeta k = m;
            memset(&m, 0, sizeof(m));
            m.off = k.off; m.L = k.L; m.nw = k.nw; m.rec_off = k.rec_off; m.bits_off = k.bits_off; m.item_off = k.item_off; m.nbits_off = k.nbits_off;
            m.bridge_off = k.bridge_off; m.bridge_cap = k.bridge_cap;
            m.n_tnode = k.n_tnode; m.n_tedge = k.n_tedge; m.tn_off = k.tn_off; m.te_off = k.te_off;
            if ((k.status == PHX_S_PARALLEL || k.status == PHX_S_BADTRNA) && k.n_tedge < 0) { m.status = k.status; m.n_tedge = 0; } // two identical tRNA hits (ValueError graphs.py:74); a hit outside the contig
        }
        HIPCHK(c, hipMemcpyAsync(c->b_meta0.p, c->meta.data(), sizeof(DMeta) * (size_t)c->n, hipMemcpyHostToDevice, s)); // (pinned: set_batch_layout waits for ev_layout before it rewrites the records)
        c->meta0_dirty = false; pushed = true;
    }
    if (c->tiles_dirty) { // once per batch layout
        const size_t tb = 4 * c->ftab.size();
        if (c->h_tiles_cap < tb) {
            if (c->h_tiles) HIPCHK(c, hipHostFree(c->h_tiles));
            c->h_tiles = nullptr; c->h_tiles_cap = 0;
            HIPCHK(c, hipHostMalloc(&c->h_tiles, tb + tb / 4 + 4096, hipHostMallocDefault));
            c->h_tiles_cap = tb + tb / 4 + 4096;
        }
        if (tb) { memcpy(c->h_tiles, c->ftab.data(), tb); HIPCHK(c, hipMemcpyAsync(c->b_tiles.p, c->h_tiles, tb, hipMemcpyHostToDevice, s)); }
        c->tiles_dirty = false; pushed = true;
    }
    if (pushed) { HIPCHK(c, hipEventRecord(c->ev_layout, s)); c->layout_pending = true; }
    return PHX_OK;
}

// One pass over the whole path.  `learn`: read the device-side totals back after each layout kernel and size the buffers
// from them (two extra host round trips: first run of a context, or a batch that outgrew it).  Otherwise everything is
// enqueued at once against the buffers the context already has — as one HIP graph launch when the previous run's graph
// still applies (same batch layout, buffers, solver classes) —; the layout kernels flag a batch that does not fit,
// later kernels then do nothing, and the caller runs again with `learn`.
int launch_once(phx_ctx *c, bool learn) {
    int rc;
    c->tapw_valid = false; c->cert_done = false; c->exact_done = false; c->exact_genes.clear(); c->exact_failed = 0; c->host_only.clear();
    hipStream_t s = c->stream;
    c->eager_now = c->eager_done && !c->meta0_dirty && !c->tiles_dirty; // the first launch after such an upload only: a repeated or retried run does everything
    c->eager_done = false;
    if (learn) c->graph_valid = false; // sizes, strides or solver classes are being re-derived
    if ((rc = ensure_position_buffers(c))) return rc;
    if ((rc = ensure(c, c->b_tot, sizeof(DTotals)))) return rc;
    if (!c->h_tot) HIPCHK(c, hipHostMalloc((void **)&c->h_tot, sizeof(DTotals), hipHostMallocDefault));
    if ((rc = ensure(c, c->b_meta0, sizeof(DMeta) * (size_t)(c->n + 1)))) return rc;
    if ((rc = ensure(c, c->b_tie, (size_t)std::max<int64_t>(1 << 20, c->tie_seen + c->tie_seen / 4)))) return rc;
    if ((rc = push_layout(c))) return rc;
    if (!learn && (rc = ensure_seg(c))) return rc;
    int mask = 0;
    int64_t lds[4] = {0, 0, 0, 0};
    DCaps caps_now;
    current_caps(c, &caps_now);
    // capturing and instantiating a graph costs about 0.7 ms: it pays when the same batch layout is run again, not for a
    // batch that is uploaded, run once and replaced (phx_annotate on fresh contigs)
    const bool use_graph = !learn && !c->prof && !c->eager_now && c->graphs_enabled && (c->runs_on_layout >= 2 || c->graph_valid);
    bool launched = false;
    if (use_graph) {
        if (c->graph_valid && c->graph_exec && c->graph_flags == caps_now.flags) {
            mask = c->last_mask;
            for (int k = 0; k < 4; k++) lds[k] = c->last_lds[k];
            if (hipGraphLaunch(c->graph_exec, s) == hipSuccess) launched = true;
            else { (void)hipGetLastError(); drop_graph(c); c->graphs_enabled = false; }
        } else {
            drop_graph(c);
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                rc = enqueue_run(c, false, mask, lds);
                hipGraph_t g = nullptr;
                const hipError_t ee = hipStreamEndCapture(s, &g);
                if (rc == PHX_OK && ee == hipSuccess && g && hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0) == hipSuccess &&
                    hipGraphLaunch(c->graph_exec, s) == hipSuccess) {
                    c->graph = g; c->graph_valid = true; c->graph_flags = caps_now.flags;
                    launched = true;
                } else {
                    (void)hipGetLastError();
                    if (g) (void)hipGraphDestroy(g);
                    drop_graph(c);
                    c->graphs_enabled = false; // this runtime cannot capture the run: enqueue directly from now on
                }
            } else { (void)hipGetLastError(); c->graphs_enabled = false; }
        }
    }
    if (!launched && (rc = enqueue_run(c, learn, mask, lds))) return rc;
    c->pend_mask = mask;
    for (int k = 0; k < 4; k++) c->pend_lds[k] = lds[k];
    return PHX_OK;
}

// ... and what follows once the stream has drained: the totals the run left on the host say whether it fitted
int finish_once(phx_ctx *c) {
    hipStream_t s = c->stream;
    const int mask = c->pend_mask;
    const int64_t *lds = c->pend_lds;
    HIPCHK(c, hipStreamSynchronize(s));
    collect_timers(c);
    c->meta_stale = true;
    memcpy(c->h_tot, (const void *)(c->res + (size_t)c->n), sizeof(DTotals)); // the totals came with the per-contig records
    const DTotals *ht = c->h_tot;
    c->tie_seen = std::max(c->tie_seen, ht->tie_need);
#ifdef FRONT_PROFILE
    if (getenv("PHX_DEBUG_FRONT")) {
        int64_t st[40];
        (void)hipMemcpy(st, (int64_t *)c->b_lpart.p + 16, sizeof st, hipMemcpyDeviceToHost);
        fprintf(stderr, "k_front stamps (us, 100 MHz clock):");
        for (int k = 1; k < 26; k++) fprintf(stderr, " %.1f", (double)(st[k] - st[k - 1]) * 0.01);
        fprintf(stderr, "  total %.1f\n", (double)(st[25] - st[0]) * 0.01);
    }
#endif
    if (ht->front_abort) { // k_front's workgroups were not all resident (other contexts / processes held the CUs): nothing of this run counts
        c->front_off = true; c->graph_valid = false;
        return kRetry;
    }
    if (ht->seg_abort) { // k_seg_merge could not join or prove some contig's segments (a margin too short for this genome, a window the tight planner cannot lay out): nothing of this run counts
        c->seg_off = true; c->seg_clean = false; c->seg_aborts++; c->graph_valid = false;
        if (getenv("PHX_DEBUG_SEG")) fprintf(stderr, "segments: run given up (reasons %d), %d contigs\n", ht->seg_abort, c->n);
        if (c->seg_aborts > 4 && 4 * c->seg_aborts > c->seg_runs + c->seg_aborts) c->seg_never = true;
        return kRetry;
    }
    // A contig that fell back costs its planner and its one sweep BEHIND the segments of the others (0.13 + 0.27 ms for 50 kb: more than the
    // segments gain on any batch): the results of this run stand, later runs of the same batch take the one-sweep kernels (which hide the planner)
    if (c->pend_seg && ht->seg_fallbacks) { c->seg_off = true; c->graph_valid = false; }
    if (c->pend_seg && !ht->seg_fallbacks && !c->seg_clean) { c->seg_clean = true; c->graph_valid = false; } // (the next runs of this batch are captured without the sweep behind the segments)
    if (c->pend_seg) { c->seg_runs++; c->seg_fallbacks += ht->seg_fallbacks; if (ht->seg_fallbacks && getenv("PHX_DEBUG_SEG")) fprintf(stderr, "segments: %d of %d contigs solved by one sweep instead\n", ht->seg_fallbacks, c->n); }
    if (ht->plan_timeouts > 0) { c->plan_timeouts += ht->plan_timeouts; c->plan_stream_off = true; c->graph_valid = false; } // (the results stand: the workgroup kernel solved those contigs)
    if (ht->overflow) { c->graph_valid = false; return kRetry; }
    bool covered = ((ht->class_mask & ~mask) & 0xffff) == 0; // (bits 16+: which classes have contigs for the side launch of the workgroup kernel: a matter of speed only)
    for (int k = 0; k < 4; k++) covered = covered && ht->lds_need[k] <= lds[k];
    if (ht->class_mask != c->last_mask || ht->lds_need[0] != c->last_lds[0] || ht->lds_need[1] != c->last_lds[1] || ht->lds_need[2] != c->last_lds[2] || ht->lds_need[3] != c->last_lds[3])
        c->graph_valid = false; // the next run launches other solver kernels / LDS sizes
    c->last_mask = ht->class_mask;
    c->last_vmax = ht->vmax; // (sizes k_certify's LDS tables; that kernel runs on demand, outside the captured graph)
    for (int k = 0; k < 4; k++) c->last_lds[k] = ht->lds_need[k];
    if (!covered) return kRetry;
    c->tot_orf = ht->orf; c->tot_grp = ht->grp; c->tot_node = ht->node; c->tot_edge = ht->edge;
    c->have_plan = true;
    c->runs_on_layout++;
    if (c->pend_front) c->front_runs++;
    return PHX_OK;
}

int run_once(phx_ctx *c, bool learn) {
    const int rc = launch_once(c, learn);
    return rc ? rc : finish_once(c);
}

// a run enqueued by phx_run_async is brought to its end before anything else touches the context
int settle(phx_ctx *c) {
    if (!c->in_flight) return PHX_OK;
    c->in_flight = false;
    const bool with_cert = c->pend_cert;
    c->pend_cert = false;
    int rc = finish_once(c);
    if (rc == PHX_OK && with_cert) { c->cert_done = true; c->meta_stale = true; } // (the certificate's kernels and its copy of the records were behind the run on the stream)
    for (int attempt = 0; rc == kRetry && attempt < 3; attempt++) rc = run_once(c, true);
    if (rc == kRetry) { c->err = "batch layout did not settle"; return PHX_E_STATE; }
    if (rc) return rc;
    c->ran = true;
    return PHX_OK;
}

} // namespace

#ifdef PHX_DEV_REPORT
// development builds only (make EXTRA="-DPHX_DEV_REPORT -DWV_PROFILE"): what the profiling variants of the solver kernels left in
// the per-contig records, printed when PHX_DEBUG_WAVE / PHX_DEBUG_SSSP / PHX_DEBUG_CENSUS is set
static void dev_report(phx_ctx *c) {
    const int n = c->n;
    (void)fetch_meta(c);
    if (getenv("PHX_DEBUG_CENSUS")) { uint32_t t[4] = {0,0,0,0}; (void)hipMemcpy(t, c->b_gtot.p, 16, hipMemcpyDeviceToHost); fprintf(stderr, "census: max concurrent sssp workgroups %u (end %u)\n", t[2], t[1]); }
    if (getenv("PHX_DEBUG_WAVE")) {
        int nfb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nw = 0;
        for (int i = 0; i < n; i++) { if (c->meta[i].sssp_mode == 2 || c->meta[i].sssp_mode == 3) nw++; else if (c->meta[i].n_node > 2 && c->meta[i].sssp_nl == 2) nfb[c->meta[i].sssp_why & 7]++; }
        fprintf(stderr, "wave kernel: %d contigs done, handed back: plan %d spill %d no-convergence %d rollbacks %d other %d\n", nw, nfb[1], nfb[2], nfb[3], nfb[4], nfb[0]);
        std::vector<std::pair<double, int>> tt;
        long nroll = 0;
        for (int i = 0; i < n; i++) if (c->meta[i].sssp_mode == 2 || c->meta[i].sssp_mode == 3) { const DMeta &m = c->meta[i]; tt.push_back({(m.pmax[0] + m.pmax[1] + m.pmax[2] + m.pmax[3] + m.pmin[0] + m.pmin[1]) * 0.01, i}); nroll += m.sweeps - 1; }
        std::sort(tt.begin(), tt.end());
        if (!tt.empty()) {
            const DMeta &m = c->meta[tt.back().second];
            fprintf(stderr, "wave kernel (WV_PROFILE builds): per-contig us p50 %.0f p90 %.0f p99 %.0f max %.0f; rollbacks %ld; slowest contig %d: V=%d phases=%d sweeps=%d | dma-wait %.1f classes+setup %.1f gather %.1f plan+stage %.1f phases %.1f end+rest %.1f\n",
                    tt[tt.size() / 2].first, tt[tt.size() * 9 / 10].first, tt[tt.size() * 99 / 100].first, tt.back().first, nroll, tt.back().second, m.n_node, m.sssp_iters, m.sweeps,
                    m.pmax[0] * 0.01, m.pmax[1] * 0.01, m.pmax[2] * 0.01, m.pmax[3] * 0.01, m.pmin[0] * 0.01, m.pmin[1] * 0.01);
            const DMeta &q = c->meta[tt[tt.size() / 2].second];
            fprintf(stderr, "  slowest contig: 64-bit phases %u (redone in 128 bits: %u), 128-bit phases %u, base moves %u; median contig: %u (%u), %u, %u\n", m.tr[0], m.tr[1], m.tr[2], m.tr[3], q.tr[0], q.tr[1], q.tr[2], q.tr[3]);
            fprintf(stderr, "  median contig (-DWV_PROFILE_AB): helper-lane maxima summed over phases A %u B %u; phases with more than 4 helpers A %u B %u\n", q.tr[4], q.tr[5], q.tr[6], q.tr[7]);
            fprintf(stderr, "  median contig %d: phases %d: relax %.1f followers %.1f spill %.1f apply %.1f loop %.1f us | setup %.1f gather %.1f plan %.1f window-end %.1f path+genes %.1f\n", tt[tt.size() / 2].second, q.sssp_iters,
                    q.bg[0] * 0.01, q.bg[1] * 0.01, q.bg[2] * 0.01, q.bg[3] * 0.01, q.bg[4] * 0.01, q.pmax[1] * 0.01, q.pmax[2] * 0.01, q.pmax[3] * 0.01, q.pmin[1] * 0.01, q.bg[5] * 0.01);
        }
    }
    if (getenv("PHX_DEBUG_WAVE"))
        for (int i = 0; i < n && i < 4; i++) fprintf(stderr, "wave contig %d: V=%d phases=%d sweeps=%d | dma-wait %.1f classes+setup %.1f gather %.1f plan+stage %.1f phases %.1f end+rest %.1f us\n", i, c->meta[i].n_node, c->meta[i].sssp_iters, c->meta[i].sweeps,
                c->meta[i].pmax[0] * 0.01, c->meta[i].pmax[1] * 0.01, c->meta[i].pmax[2] * 0.01, c->meta[i].pmax[3] * 0.01, c->meta[i].pmin[0] * 0.01, c->meta[i].pmin[1] * 0.01);
    if (getenv("PHX_DEBUG_SSSP"))
        for (int i = 0; i < n && i < 6; i++) fprintf(stderr, "sssp contig %d: V=%d iters=%d sweeps=%d setup=%.1fus iter=%.1fus tail=%.1fus\n", i, c->meta[i].n_node, c->meta[i].sssp_iters, c->meta[i].sweeps, c->meta[i].pmax[0] * 0.01, c->meta[i].pmin[0] * 0.01, c->meta[i].sssp_why * 0.01);
}
#endif

static int enqueue_cert(phx_ctx *c); // (below, with the certificate)
int phx_run_async(phx_ctx *c) {
    if (!c) return PHX_E_ARG;
    if (!c->uploaded) return PHX_E_STATE;
    // the first run of a context (and one after a batch outgrew its buffers) sizes the buffers between kernels: synchronous
    if (!c->have_plan || c->always_sync || c->n == 0) return phx_run(c);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = settle(c);
    if (rc) return rc;
    c->ran = false;
    c->tot_orf = c->tot_grp = c->tot_node = c->tot_edge = 0;
    if ((rc = launch_once(c, false))) return rc;
    c->in_flight = true;
    // The downloads that follow ask for the certificate (the reference's genes, §5c of DESIGN.md): behind the run on the stream it is
    // done when they come, and with two batches in flight it runs beside the other context's kernels instead of holding the host up.
    // (Sized and launched as the run itself: on what the last run of the context saw; a run that is repeated leaves it to the downloads.)
    if (c->certify && c->exact && c->eager_cert) {
        DCaps k;
        current_caps(c, &k);
        if (ensure(c, c->b_eref, ((size_t)k.edge + 16) * sizeof(DERef)) == PHX_OK && enqueue_cert(c) == PHX_OK) c->pend_cert = true;
    }
    return PHX_OK;
}

int phx_wait(phx_ctx *c) {
    if (!c) return PHX_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_flight) return settle(c);
    return c->ran ? PHX_OK : PHX_E_STATE;
}

int phx_run(phx_ctx *c) {
    if (!c) return PHX_E_ARG;
    if (!c->uploaded) return PHX_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rs = settle(c); if (rs) return rs; }
    c->ran = false;
    c->tot_orf = c->tot_grp = c->tot_node = c->tot_edge = 0;
    if (c->n == 0) { c->ran = true; return PHX_OK; }
    int rc = run_once(c, !c->have_plan || c->always_sync);
    for (int attempt = 0; rc == kRetry && attempt < 3; attempt++) rc = run_once(c, true);
    if (rc == kRetry) { c->err = "batch layout did not settle"; return PHX_E_STATE; }
    if (rc) return rc;
#ifdef PHX_DEV_REPORT
    dev_report(c); // per-contig timing of builds with -DWV_PROFILE / -DSW_PROFILE (tools/prof_wave.sh)
#endif
    c->ran = true;
    return PHX_OK;
}

static int ensure_gene_stage(phx_ctx *c, size_t n) {
    if (n <= c->h_genes_cap) return PHX_OK;
    if (c->h_genes) HIPCHK(c, hipHostFree(c->h_genes));
    c->h_genes = nullptr; c->h_genes_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    HIPCHK(c, hipHostMalloc((void **)&c->h_genes, cap * sizeof(DGene), hipHostMallocDefault));
    c->h_genes_cap = cap;
    return PHX_OK;
}

static int ensure_exact(phx_ctx *c); // certificate, and the host re-solve of what it leaves open (below, behind the taps it reads)

int phx_download(phx_ctx *c, phx_result *out) {
    if (!c || (!out && c->n > 0)) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (!c->ran) return PHX_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rx = ensure_exact(c); if (rx) return rx; }
    int64_t total = 0;
    for (int i = 0; i < c->n; i++) total = std::max<int64_t>(total, c->res[i].gene_off + c->res[i].n_genes);
    { const int rg = ensure_gene_stage(c, (size_t)total); if (rg) return rg; }
    if (total) {
        HIPCHK(c, hipMemcpyAsync(c->h_genes, (const DGene *)c->b_genes.p + (gene_pack(c) ? gene_half(c) : 0), sizeof(DGene) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < c->n; i++) { out[i].status = 0; out[i].n_genes = 0; out[i].genes = nullptr; } // (so that a failure half-way leaves nothing dangling)
    for (int i = 0; i < c->n; i++) {
        const DRes &m = c->res[(size_t)i];
        out[i].status = m.status;
        const auto ex = c->exact_genes.find(i); // solved again on the host: those genes
        const DGene *src = ex != c->exact_genes.end() ? ex->second.data() : c->h_genes + (size_t)m.gene_off;
        out[i].n_genes = m.status < 0 ? 0 : (ex != c->exact_genes.end() ? (int32_t)ex->second.size() : m.n_genes);
        out[i].genes = nullptr;
        if (out[i].n_genes > 0) {
            out[i].genes = (phx_gene *)malloc(sizeof(phx_gene) * (size_t)out[i].n_genes);
            if (!out[i].genes) { out[i].n_genes = 0; phx_free_results(out, c->n); return PHX_E_NOMEM; }
            for (int k = 0; k < out[i].n_genes; k++) {
                const DGene &g = src[k];
                out[i].genes[k].left = g.left; out[i].genes[k].right = g.right;
                out[i].genes[k].strand = g.strand; out[i].genes[k].frame = g.frame;
                out[i].genes[k].score = g.score;
            }
        }
    }
    return PHX_OK;
}

int phx_download_flat(phx_ctx *c, phx_gene *genes, int64_t cap, int64_t *offsets, int32_t *status, int64_t *total_out) {
    if (!c || (c->n > 0 && (!offsets || !status))) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (!c->ran) return PHX_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rx = ensure_exact(c); if (rx) return rx; }
    auto count_of = [&](int i) -> int64_t { // genes of contig i: the host re-solve's where there was one
        const DRes &m = c->res[(size_t)i];
        if (m.status < 0) return 0;
        if (!c->exact_genes.empty()) { const auto ex = c->exact_genes.find(i); if (ex != c->exact_genes.end()) return (int64_t)ex->second.size(); }
        return m.n_genes;
    };
    int64_t total = 0, hi = 0;
    for (int i = 0; i < c->n; i++) { const DRes &m = c->res[(size_t)i]; total += count_of(i); hi = std::max<int64_t>(hi, m.gene_off + m.n_genes); }
    if (total_out) *total_out = total;
    if (!genes) { // size query
        int64_t o = 0;
        for (int i = 0; i < c->n; i++) { const DRes &m = c->res[(size_t)i]; offsets[i] = o; status[i] = m.status; o += count_of(i); }
        if (c->n >= 0 && offsets) offsets[c->n] = o;
        return PHX_OK;
    }
    if (cap < total) return PHX_E_ARG;
    { const int rg = ensure_gene_stage(c, (size_t)hi); if (rg) return rg; }
    if (hi) {
        HIPCHK(c, hipMemcpyAsync(c->h_genes, (const DGene *)c->b_genes.p + (gene_pack(c) ? gene_half(c) : 0), sizeof(DGene) * (size_t)hi, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    static_assert(sizeof(DGene) == sizeof(phx_gene), "device and ABI gene records have the same layout");
    int64_t o = 0;
    for (int i = 0; i < c->n; i++) {
        const DRes &m = c->res[(size_t)i];
        const int64_t k = count_of(i);
        offsets[i] = o; status[i] = m.status;
        const DGene *src = &c->h_genes[(size_t)m.gene_off];
        if (!c->exact_genes.empty()) { const auto ex = c->exact_genes.find(i); if (ex != c->exact_genes.end()) src = ex->second.data(); }
        if (k) memcpy(genes + o, src, sizeof(phx_gene) * (size_t)k);
        o += k;
    }
    offsets[c->n] = o;
    return PHX_OK;
}

// The certificate is computed when it is first asked for after a run (k_refine + k_certify on the state the run left on the device:
// distances, parent edges, path, edge records), not inside phx_run.  phx_download* ask for it (their gene lists are the reference's);
// a caller that only wants the device's lists at full rate creates the context with PHX_CREATE_NO_EXACT (or phx_set_exact(ctx, 0)):
// the downloads then skip it, phx_certified still computes it on demand.
static int enqueue_cert(phx_ctx *c) { // k_refine + k_certify + the records, on the context's stream (b_eref holds an entry for every edge the edge buffers hold)
    DBatch b;
    fill_batch(c, &b);
    int nlm = 0;
    for (int k = 0; k < 4; k++) nlm |= ((c->last_mask >> (4 * k)) & 15) ? 1 << k : 0;
    {
        StageTimer t(c, ST_CERTIFY);
        phxk_refine(&b, c->stream); // the flagged edges once more in double-double: flags cleared or bounds in b_eref
        phxk_certify(&b, nlm, c->cert_wide ? -1 : c->last_vmax, c->stream);
        phxk_results(&b, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->res, c->b_res.p, sizeof(DRes) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    return PHX_OK;
}
static int ensure_cert(phx_ctx *c) {
    if (!c->certify || c->cert_done || c->n == 0) return PHX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { DCaps k; current_caps(c, &k); const int re = ensure(c, c->b_eref, ((size_t)std::max<int64_t>(c->tot_edge, k.edge) + 16) * sizeof(DERef)); if (re) return re; } // (an entry for every edge the edge buffers hold: phx_run_async sizes it the same way, before the run's edge count is known)
    { const int rq = enqueue_cert(c); if (rq) return rq; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_timers(c);
    c->meta_stale = true;
    c->cert_done = true;
    return PHX_OK;
}

int phx_certified(phx_ctx *c, int8_t *cert) {
    if (!c || (!cert && c->n > 0)) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (!c->ran) return PHX_E_STATE;
    { const int rc = c->exact ? ensure_exact(c) : ensure_cert(c); if (rc) return rc; }
    for (int i = 0; i < c->n; i++) { const DRes &m = c->res[(size_t)i]; cert[i] = (int8_t)(!c->certify ? -1 : (m.status < 0 ? 1 : m.cert)); }
    return PHX_OK;
}

// ---- one process, several GPUs: a host thread and two contexts per device, batches round the devices ----
struct phx_pool {
    struct Lane { int device = 0; phx_ctx *ctx[2] = {nullptr, nullptr}; };
    std::vector<Lane> lanes;
    std::string err;
};

int phx_pool_create(const phx_params *params, int32_t n_dev, const int32_t *devices, uint32_t flags, phx_pool **out) {
    if (!out || n_dev < 1 || !devices || (flags & PHX_CREATE_USE_STREAM)) return PHX_E_ARG;
    *out = nullptr;
    phx_pool *p = new phx_pool();
    p->lanes.resize((size_t)n_dev);
    for (int i = 0; i < n_dev; i++) {
        p->lanes[(size_t)i].device = devices[i];
        for (int k = 0; k < 2; k++) {
            const int rc = phx_create_ex(params, devices[i], nullptr, flags, &p->lanes[(size_t)i].ctx[k]);
            if (rc) { phx_pool_destroy(p); return rc; }
        }
    }
    *out = p;
    return PHX_OK;
}

void phx_pool_destroy(phx_pool *p) {
    if (!p) return;
    for (auto &ln : p->lanes) for (int k = 0; k < 2; k++) if (ln.ctx[k]) phx_destroy(ln.ctx[k]);
    delete p;
}

const char *phx_pool_last_error(const phx_pool *p) { return p ? p->err.c_str() : ""; }

int phx_pool_annotate(phx_pool *p, int32_t n, const char *const *seq, const int64_t *len, int64_t batch_bases,
                      const int64_t *trna_offsets, const int32_t *trna_start, const int32_t *trna_stop, phx_result *out) {
    if (!p || n < 0 || (n > 0 && (!seq || !len || !out))) return PHX_E_ARG;
    if (n == 0) return PHX_OK;
    for (int i = 0; i < n; i++) { out[i].status = 0; out[i].n_genes = 0; out[i].genes = nullptr; }
    const size_t nl = p->lanes.size();
    // the contigs are dealt to the lanes (GPUs) greedily, longest first, by the bases a lane already holds — one T4
    // among two hundred phiX-sized contigs does not leave the other lanes idle —; inside a lane they keep their input order and are
    // cut into batches of at most batch_bases bases, at least two per lane, so that every GPU has two in flight.
    int64_t total = 0;
    for (int i = 0; i < n; i++) total += len[i] > 0 ? len[i] : 0;
    int64_t limit = batch_bases > 0 ? batch_bases : 400000000ll;
    if (nl > 1) limit = std::max<int64_t>(1, std::min<int64_t>(limit, (total + 2 * (int64_t)nl - 1) / (2 * (int64_t)nl)));
    std::vector<std::vector<int>> lane_idx(nl);
    std::vector<std::vector<std::pair<int, int>>> lane_cuts(nl); // per lane: [lo, hi) into lane_idx
    try {
        if (nl == 1) { lane_idx[0].resize((size_t)n); for (int i = 0; i < n; i++) lane_idx[0][(size_t)i] = i; }
        else {
            std::vector<int> order((size_t)n);
            for (int i = 0; i < n; i++) order[(size_t)i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
            std::vector<int64_t> load(nl, 0);
            for (int i : order) {
                size_t best = 0;
                for (size_t j = 1; j < nl; j++) if (load[j] < load[best]) best = j;
                lane_idx[best].push_back(i);
                load[best] += len[i] > 0 ? len[i] : 0;
            }
            for (auto &v : lane_idx) std::sort(v.begin(), v.end());
        }
        for (size_t j = 0; j < nl; j++) {
            const std::vector<int> &v = lane_idx[j];
            for (size_t lo = 0; lo < v.size();) {
                size_t hi = lo; int64_t size = 0;
                while (hi < v.size() && (hi == lo || size + len[v[hi]] <= limit)) { size += len[v[hi]]; hi++; }
                lane_cuts[j].push_back({(int)lo, (int)hi});
                lo = hi;
            }
        }
    } catch (const std::bad_alloc &) { p->err = "out of memory"; return PHX_E_NOMEM; }
    std::atomic<int> first_err{0};
    std::mutex em;
    auto lane_body = [&](size_t j) {
        phx_pool::Lane &ln = p->lanes[j];
        const std::vector<int> &idx = lane_idx[j];
        const std::vector<std::pair<int, int>> &cuts = lane_cuts[j];
        auto fail = [&](int rc, phx_ctx *c) { int z = 0; if (first_err.compare_exchange_strong(z, rc)) { std::lock_guard<std::mutex> g(em); p->err = c ? phx_last_error(c) : "out of memory"; } };
        std::vector<phx_result> tmp;
        auto collect = [&](size_t t) { // batch t ran on ctx[t & 1]; its results go to their contigs' places
            phx_ctx *c = ln.ctx[t & 1];
            const int lo = cuts[t].first, hi = cuts[t].second;
            tmp.assign((size_t)(hi - lo), phx_result{0, 0, nullptr});
            const int rc = phx_download(c, tmp.data());
            if (rc) { fail(rc, c); return; }
            for (int k = lo; k < hi; k++) out[idx[(size_t)k]] = tmp[(size_t)(k - lo)];
        };
        std::vector<const char *> bseq;
        std::vector<int64_t> blen, toffs;
        std::vector<int32_t> tst, tsp;
        for (size_t t = 0; t < cuts.size() && !first_err.load(); t++) {
            if (t >= 2) collect(t - 2);
            phx_ctx *c = ln.ctx[t & 1];
            const int lo = cuts[t].first, hi = cuts[t].second;
            bseq.clear(); blen.clear();
            for (int k = lo; k < hi; k++) { bseq.push_back(seq[idx[(size_t)k]]); blen.push_back(len[idx[(size_t)k]]); }
            int rc = phx_upload(c, hi - lo, bseq.data(), blen.data());
            if (!rc && trna_offsets) {
                toffs.assign(1, 0); tst.clear(); tsp.clear();
                for (int k = lo; k < hi; k++) {
                    const int i = idx[(size_t)k];
                    for (int64_t q = trna_offsets[i]; q < trna_offsets[i + 1]; q++) { tst.push_back(trna_start[q]); tsp.push_back(trna_stop[q]); }
                    toffs.push_back((int64_t)tst.size());
                }
                rc = phx_set_trnas(c, toffs.data(), tst.data(), tsp.data());
            }
            if (!rc) rc = phx_run_async(c);
            if (rc) { fail(rc, c); break; }
        }
        if (!first_err.load()) for (size_t t = cuts.size() >= 2 ? cuts.size() - 2 : 0; t < cuts.size(); t++) collect(t);
        else for (int k = 0; k < 2; k++) (void)phx_wait(ln.ctx[k]);
    };
    // (nothing may leave a lane's thread as an exception: a bad_alloc there would end the caller's process)
    auto lane_work = [&](size_t j) {
        try { lane_body(j); }
        catch (...) {
            int z = 0;
            if (first_err.compare_exchange_strong(z, PHX_E_NOMEM)) { std::lock_guard<std::mutex> g(em); p->err = "out of memory in a lane's host thread"; }
            for (int k = 0; k < 2; k++) (void)phx_wait(p->lanes[j].ctx[k]);
        }
    };
    std::vector<std::thread> th;
    for (size_t j = 1; j < nl; j++) { try { th.emplace_back(lane_work, j); } catch (...) { first_err = PHX_E_NOMEM; break; } }
    lane_work(0);
    for (std::thread &t : th) t.join();
    const int rc = first_err.load();
    if (rc) phx_free_results(out, n);
    return rc;
}

int phx_set_exact(phx_ctx *c, int on) {
    if (!c) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (!on) { // the device's own lists again, for every contig
        for (auto &kv : c->exact_genes) if (c->res && kv.first < c->n) c->res[(size_t)kv.first].cert = 0;
        for (int i : c->host_only) if (c->res && i < c->n) c->res[(size_t)i].status = PHX_S_OVERFLOW; // (the device has no genes for them)
        c->host_only.clear();
        c->exact_genes.clear(); c->exact_done = false;
    }
    c->exact = on != 0;
    return PHX_OK;
}

int phx_annotate(phx_ctx *c, int32_t n, const char *const *seq, const int64_t *len, phx_result *out) {
    int rc = phx_upload(c, n, seq, len);
    if (rc) return rc;
    if ((rc = phx_run_async(c))) return rc; // (with the certificate behind the run on the stream: the download asks for it; the first run of a context is synchronous)
    return phx_download(c, out);
}

void phx_free_results(phx_result *res, int32_t n) {
    if (!res) return;
    for (int i = 0; i < n; i++) { free(res[i].genes); res[i].genes = nullptr; res[i].n_genes = 0; }
}

// ---- taps ----
#define TAP_PRE(c, contig)                                            \
    if (!(c)) return PHX_E_ARG;                                       \
    if ((c)->in_flight) { const int rs_ = phx_wait(c); if (rs_) return rs_; } \
    if (!(c)->ran) return PHX_E_STATE;                                \
    if ((contig) < 0 || (contig) >= (c)->n) return PHX_E_ARG;         \
    HIPCHK(c, hipSetDevice((c)->device));                             \
    { const int rf_ = fetch_meta(c); if (rf_) return rf_; }           \
    const DMeta &m = (c)->meta[(size_t)(contig)];

extern "C" void phxk_sssp_only(const DBatch *b, int n_limbs, void *stream);
static void hipLaunchSsspOnly(const DBatch *b, int nl, hipStream_t s) { phxk_sssp_only(b, nl, (void *)s); }

static double host_contig_pstop(uint32_t gc, int L) {
    double fa = (double)((uint32_t)L - gc), fg = (double)gc, d = (double)((int64_t)L * 2);
    double Pa = fa / d, Pt = fa / d, Pg = fg / d;
    return Pt * Pa * Pa + Pt * Pg * Pa + Pt * Pa * Pg;
}

static int ensure_cert(phx_ctx *c);
int phx_tap_globals(phx_ctx *c, int32_t contig, phx_globals *out) {
    if (c && c->ran && !c->in_flight) { const int rc_ = ensure_cert(c); if (rc_) return rc_; }
    TAP_PRE(c, contig);
    if (!out) return PHX_E_ARG;
    memset(out, 0, sizeof(*out));
    out->L = m.L;
    out->status = m.status;
    out->n_limbs = m.sssp_nl;
    if (m.status < 0) return PHX_OK;
    out->pstop = host_contig_pstop(m.gc, m.L);
    double bgs = 0, trs = 0;
    for (int i = 0; i < 28; i++) { bgs += 1.0 + (double)m.bg[i]; trs += 1.0 + (double)m.tr[i]; }
    for (int i = 0; i < 28; i++) { out->background_rbs[i] = (1.0 + (double)m.bg[i]) / bgs; out->training_rbs[i] = (1.0 + (double)m.tr[i]) / trs; }
    double ymx = 1, ymn = 1;
    for (int i = 0; i < 4; i++) {
        out->pos_max[i] = 1.0 + (double)(i ? m.pmax[i] : 0u); out->pos_min[i] = 1.0 + (double)(i ? m.pmin[i] : 0u);
        ymx = std::max(ymx, out->pos_max[i]); ymn = std::max(ymn, out->pos_min[i]);
    }
    for (int i = 0; i < 4; i++) { out->pos_max[i] /= ymx; out->pos_min[i] /= ymn; }
    out->n_orf = m.n_orf; out->n_group = m.n_grp; out->n_node = m.n_node; out->n_edge = m.n_edge; out->n_bridge = m.n_bridge;
    out->sssp_sweeps = m.sweeps;
    out->sssp_iters = m.sssp_iters;
    out->sssp_kernel = m.sssp_mode;
    out->sssp_handed_back = m.sssp_why > 0 ? m.sssp_why : 0;
    out->tie = m.tie;
    out->certified = c->certify ? (c->exact_genes.count(contig) ? 2 : m.cert) : -1;
    for (int i = 0; i < 28; i++) { out->rbs_background_count[i] = m.bg[i]; out->rbs_training_count[i] = m.tr[i]; }
    for (int i = 0; i < 4; i++) { out->gc_max_count[i] = i ? m.pmax[i] : 0u; out->gc_min_count[i] = i ? m.pmin[i] : 0u; }
    out->gc_count = m.gc;
    return PHX_OK;
}

// the bases of a contig as codes (0 a, 1 c, 2 t, 3 g, 4 anything else), from the records the kernels read
static int fetch_codes(phx_ctx *c, const DMeta &m, std::vector<uint8_t> &codes) {
    const size_t L = (size_t)m.L, nrec = 2 * (size_t)m.nw;
    codes.assign(L, 4);
    if (!L) return PHX_OK;
    std::vector<uint32_t> rec(nrec * 9);
    HIPCHK(c, hipMemcpy(rec.data(), (const uint32_t *)c->b_recs.p + (size_t)m.rec_off * 9, rec.size() * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < L; p++) {
        const size_t r = p % 3, k = p / 3;
        const uint32_t *q = &rec[(k >> 5) * 9 + r * 3];
        const unsigned sh = (unsigned)(k & 31);
        codes[p] = ((q[2] >> sh) & 1u) ? (uint8_t)4 : (uint8_t)(((q[0] >> sh) & 1u) | (((q[1] >> sh) & 1u) << 1));
    }
    return PHX_OK;
}

int phx_tap_positions(phx_ctx *c, int32_t contig, uint8_t *cls, uint8_t *gcc, uint8_t *binF, uint8_t *binR) {
    TAP_PRE(c, contig);
    const size_t L = (size_t)m.L;
    if (cls) { // the device keeps the codon classes as four bitmaps; the start-codon index and the "rev_comp(codon) is a start" bit of a
               // forward start codon (no kernel keeps them per position) come from the bases and the class table
        const size_t nw = (size_t)m.nw;
        std::vector<uint64_t> bits((size_t)4 * 3 * nw);
        std::vector<uint8_t> cd;
        if (nw) HIPCHK(c, hipMemcpy(bits.data(), (uint64_t *)c->b_bits.p + m.bits_off, bits.size() * 8, hipMemcpyDeviceToHost));
        int rc = fetch_codes(c, m, cd);
        if (rc) return rc;
        DParams dp;
        build_dparams(&c->params, &dp);
        for (size_t p = 0; p < L; p++) {
            uint8_t v = 0;
            if (p + 3 <= L) {
                const size_t f = p % 3, k = p / 3;
                auto bit = [&](int id) { return (int)((bits[((size_t)id * 3 + f) * nw + (k >> 6)] >> (k & 63)) & 1ull); };
                const int cl = bit(0) ? CLS_FS : bit(1) ? CLS_RS : bit(2) ? CLS_FT : bit(3) ? CLS_RT : CLS_NONE;
                v = (uint8_t)cl;
                if (cd[p] < 4 && cd[p + 1] < 4 && cd[p + 2] < 4) {
                    const uint8_t t = dp.cls_tab[cd[p] | (cd[p + 1] << 2) | (cd[p + 2] << 4)];
                    if (cl == CLS_FS || cl == CLS_RS) v |= (uint8_t)(t & 0x78u);
                    v |= (uint8_t)(t & 0x80u);
                }
            }
            cls[p] = v;
        }
    }
    if (gcc) { // the device keeps the GC-frame classes bit-sliced: three comparison bitmaps per strand and frame
        const size_t nw = (size_t)m.nw;
        std::vector<uint64_t> bits((size_t)PHX_BITMAP_WORDS_PER_NW * nw);
        if (nw) HIPCHK(c, hipMemcpy(bits.data(), (uint64_t *)c->b_bits.p + m.bits_off, bits.size() * 8, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < L; p++) {
            const size_t f = p % 3, k = p / 3;
            uint8_t g = 0;
            if (p + 3 <= L) {
                auto bit = [&](int id) { return (int)((bits[((size_t)id * 3 + f) * nw + (k >> 6)] >> (k & 63)) & 1ull); };
                // planes a > b, b > c, a > c (and c > b, b > a, c > a for the reversed triple): gc_frame_plot.py:7-28
                auto cls = [&](int base) { const int A = bit(base), B = bit(base + 1), C = bit(base + 2); return ((A && C) ? 0 : ((!A && B) ? 1 : 2)) * 3 + ((!A && !C) ? 0 : ((A && !B) ? 1 : 2)); };
                const int fc = cls(PHX_PLANE_GCF), rc = cls(PHX_PLANE_GCR);
                g = (uint8_t)(fc | (rc << 4));
            }
            gcc[p] = g;
        }
    }
    if (binF || binR) { // the RBS bins of every window are no output of a run any more: k_features computes them once more for this
                        // contig and leaves them as bit planes ([record][strand][stream][bit], the same code as the run's)
        const size_t nrec = 2 * (size_t)m.nw;
        std::vector<uint32_t> tp(nrec * 30);
        if (nrec) {
            uint32_t *d = nullptr;
            HIPCHK(c, hipMalloc((void **)&d, tp.size() * 4));
            DBatch b;
            fill_batch(c, &b);
            const uint32_t v0 = c->ftab[(size_t)contig];
            phxk_features_tap(&b, contig, v0, v0 + (uint32_t)nrec, d, c->stream);
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipMemcpy(tp.data(), d, tp.size() * 4, hipMemcpyDeviceToHost);
            (void)hipFree(d);
            HIPCHK(c, e);
        }
        for (size_t p = 0; p < L; p++) {
            const size_t r = p % 3, k = p / 3, w = k >> 5;
            const unsigned sh = (unsigned)(k & 31);
            unsigned f = 0, rv = 0;
            for (int bb = 0; bb < 5; bb++) { f |= ((tp[((w * 2 + 0) * 3 + r) * 5 + bb] >> sh) & 1u) << bb; rv |= ((tp[((w * 2 + 1) * 3 + r) * 5 + bb] >> sh) & 1u) << bb; }
            if (binF) binF[p] = (uint8_t)f;
            if (binR) binR[p] = (uint8_t)rv;
        }
    }
    return PHX_OK;
}

// Reference insertion order of the stop-groups = ascending DGrp.evkey.  ref_rank[g] = rank of device
// group g, ref_first[g] = number of ORFs in the groups before it.
static void reference_order(const std::vector<DGrp> &grp, std::vector<int> &order, std::vector<int> &ref_rank, std::vector<int> &ref_first) {
    const size_t G = grp.size();
    order.resize(G); ref_rank.resize(G); ref_first.resize(G);
    for (size_t g = 0; g < G; g++) order[g] = (int)g;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return grp[(size_t)a].evkey < grp[(size_t)b].evkey; });
    int acc = 0;
    for (size_t r = 0; r < G; r++) { ref_rank[(size_t)order[r]] = (int)r; ref_first[(size_t)order[r]] = acc; acc += grp[(size_t)order[r]].n; }
}

int phx_tap_orfs(phx_ctx *c, int32_t contig, phx_orf *out) {
    TAP_PRE(c, contig);
    if (m.status < 0 || m.n_orf == 0) return PHX_OK;
    if (!out) return PHX_E_ARG;
    std::vector<DOrf> d((size_t)m.n_orf);
    std::vector<DOrfStat> ds((size_t)m.n_orf);
    std::vector<double> dw((size_t)m.n_orf);
    std::vector<DGrp> grp((size_t)m.n_grp);
    HIPCHK(c, hipMemcpy(d.data(), (DOrf *)c->b_orf.p + m.orf_off, sizeof(DOrf) * d.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ds.data(), (DOrfStat *)c->b_ostat.p + m.orf_off, sizeof(DOrfStat) * ds.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(dw.data(), (double *)c->b_oweight.p + m.orf_off, 8 * dw.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(grp.data(), (DGrp *)c->b_grp.p + m.grp_off, sizeof(DGrp) * grp.size(), hipMemcpyDeviceToHost));
    std::vector<int> order, ref_rank, ref_first;
    reference_order(grp, order, ref_rank, ref_first);
    phx_globals gl;
    phx_tap_globals(c, contig, &gl);
    std::vector<uint64_t> wide_bits;
    size_t k = 0;
    for (size_t rr = 0; rr < order.size(); rr++) {
        const DGrp &G = grp[(size_t)order[rr]];
        for (int j = 0; j < G.n; j++, k++) {
            const DOrf &r = d[(size_t)(G.orf_begin + j)];
            const DOrfStat &rs = ds[(size_t)(G.orf_begin + j)];
            phx_orf &o = out[k];
            o.start = r.start; o.stop = r.stop; o.frame = r.frame;
            o.length = r.frame > 0 ? r.stop + 2 - r.start + 1 : r.start + 2 - r.stop + 1;
            o.rbs = rs.rbs; o.startidx = rs.startidx; o.group = (int32_t)rr;
            uint32_t h[9];
            for (int i = 0; i < 9; i++) h[i] = rs.hist[i];
            if (rs.flags & 4u) { // more than 65535 codons: the 16-bit record is clipped (k_score counted again); the tap counts on the host
                if (wide_bits.empty()) {
                    wide_bits.resize((size_t)PHX_BITMAP_WORDS_PER_NW * (size_t)m.nw);
                    HIPCHK(c, hipMemcpy(wide_bits.data(), (uint64_t *)c->b_bits.p + m.bits_off, wide_bits.size() * 8, hipMemcpyDeviceToHost));
                }
                const bool fwd = r.frame > 0;
                const int f = (fwd ? r.frame : -r.frame) - 1;
                const int lo1 = fwd ? r.start : r.stop, hi1 = (fwd ? r.stop : r.start) + 2, ncod = (hi1 - lo1 + 1) / 3;
                const int k0 = (lo1 - 1 - f) / 3, k1 = k0 + ncod - 1, clo = fwd ? k0 : k0 + 1, chi = fwd ? k1 - 1 : k1;
                const size_t nw = (size_t)m.nw;
                const uint64_t *A = wide_bits.data() + ((size_t)((fwd ? PHX_PLANE_GCF : PHX_PLANE_GCR) * 3 + f)) * nw, *B = A + 3 * nw, *Cc = A + 6 * nw;
                for (int i = 0; i < 9; i++) h[i] = 0;
                for (int k = clo; k <= chi; k++) {
                    const int a = (int)((A[k >> 6] >> (k & 63)) & 1), bb = (int)((B[k >> 6] >> (k & 63)) & 1), cc = (int)((Cc[k >> 6] >> (k & 63)) & 1);
                    const int mx = (a && cc) ? 0 : ((!a && bb) ? 1 : 2), mn = (!a && !cc) ? 0 : ((a && !bb) ? 1 : 2); // gc_frame_plot.py:7-28
                    h[mx * 3 + mn]++;
                }
            }
            double S = 0;
            for (int a = 0; a < 3; a++) for (int cc = 0; cc < 3; cc++) { o.hist[a * 3 + cc] = (int32_t)h[a * 3 + cc]; S += (double)h[a * 3 + cc] * (gl.pos_max[a + 1] * gl.pos_min[cc + 1]); }
            o.S = S;
            o.pstop = rs.pstop;
            o.weight_rbs = gl.training_rbs[rs.rbs] / gl.background_rbs[rs.rbs];
            o.weight = dw[(size_t)(G.orf_begin + j)];
        }
    }
    return PHX_OK;
}

int phx_tap_nodes(phx_ctx *c, int32_t contig, phx_node *out) {
    TAP_PRE(c, contig);
    if (m.status < 0 || m.n_node == 0) return PHX_OK;
    if (!out) return PHX_E_ARG;
    const size_t V = (size_t)m.n_node;
    std::vector<DNode> nd(V);
    std::vector<double> no(V);
    std::vector<DGrp> grp((size_t)m.n_grp);
    std::vector<DOrf> orf((size_t)m.n_orf);
    HIPCHK(c, hipMemcpy(nd.data(), (DNode *)c->b_node.p + m.node_off, V * sizeof(DNode), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(no.data(), (double *)c->b_no.p + m.node_off, V * 8, hipMemcpyDeviceToHost));
    if (m.n_grp) HIPCHK(c, hipMemcpy(grp.data(), (DGrp *)c->b_grp.p + m.grp_off, sizeof(DGrp) * grp.size(), hipMemcpyDeviceToHost));
    if (m.n_orf) HIPCHK(c, hipMemcpy(orf.data(), (DOrf *)c->b_orf.p + m.orf_off, sizeof(DOrf) * orf.size(), hipMemcpyDeviceToHost));
    std::vector<int> order, ref_rank, ref_first;
    reference_order(grp, order, ref_rank, ref_first);
    for (size_t v = 0; v < V; v++) {
        out[v].pos = nd[v].pos; out[v].type = (int8_t)NTYPE(nd[v].info); out[v].frame = (int8_t)NFRAME(nd[v].info); out[v].pad = 0;
        out[v].other = nd[v].other; out[v].o = no[v];
        // reference insertion rank (functions.py:311-318): per ORF (source, target) in iter_orfs order
        int ref = -1;
        const int ntn = c->has_trna ? m.n_tnode : 0;
        if (v + 2 == V) ref = m.n_orf + m.n_grp + ntn;
        else if (v + 1 == V) ref = m.n_orf + m.n_grp + ntn + 1;
        else if (LINK_KIND(nd[v].link) == LINK_TRNA) ref = m.n_orf + m.n_grp + c->h_tnode[(size_t)m.tn_off + LINK_IDX(nd[v].link)].rank; // add_trnas runs after every CDS node exists (functions.py:357)
        else if (LINK_KIND(nd[v].link) == LINK_STOP) {
            const size_t g = (size_t)LINK_IDX(nd[v].link);
            ref = ref_first[g] + ref_rank[g] + (grp[g].frame > 0 ? 1 : 0);
        } else {
            const int k = (int)LINK_IDX(nd[v].link);
            const size_t g = (size_t)orf[(size_t)k].grp;
            const int mth = k - grp[g].orf_begin;
            const int base = ref_first[g] + ref_rank[g];
            ref = grp[g].frame > 0 ? (mth == 0 ? base : base + 1 + mth) : base + 1 + mth;
        }
        out[v].refidx = ref;
    }
    return PHX_OK;
}

// fp64 weights and plain source nodes of the whole batch, recomputed once per run for the taps (the run keeps the integers the
// solver adds, not the doubles they were truncated from)
static int ensure_tap_weights(phx_ctx *c) {
    if (c->tapw_valid) return PHX_OK;
    int rc;
    const size_t E = (size_t)c->tot_edge;
    if ((rc = ensure(c, c->b_esrcf, (E + 1) * 4))) return rc;
    if ((rc = ensure(c, c->b_ewf, (E + 1) * 8))) return rc;
    DBatch b;
    fill_batch(c, &b);
    b.esrcf = (uint32_t *)c->b_esrcf.p; b.ewf = (double *)c->b_ewf.p;
    b.defer_overlap = c->max_len < (1 << 21) ? 1 : 0;
    phxk_edges_tap(&b, c->stream);
    phxk_edges_expand(&b, 0, 0, c->stream); // the taps (and the host re-solve) read DBatch.ew of every edge: the coded gap edges' weights from the gap tables
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tapw_valid = true;
    return PHX_OK;
}

int phx_tap_edges(phx_ctx *c, int32_t contig, phx_edge *out) {
    TAP_PRE(c, contig);
    if (m.status < 0 || m.n_edge == 0) return PHX_OK;
    if (!out) return PHX_E_ARG;
    { const int rt = ensure_tap_weights(c); if (rt) return rt; }
    const size_t V = (size_t)m.n_node, E = (size_t)m.n_edge;
    std::vector<uint32_t> in_off(V + 1), esrc(E), esrci(E);
    std::vector<double> ew(E);
    std::vector<long long> ewi(E);
    HIPCHK(c, hipMemcpy(in_off.data(), (uint32_t *)c->b_inoff.p + m.node_off + contig, (V + 1) * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(esrc.data(), (uint32_t *)c->b_esrcf.p + m.edge_off, E * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ew.data(), (double *)c->b_ewf.p + m.edge_off, E * 8, hipMemcpyDeviceToHost));
    // the records the solver read: source | off-path << 30 | inexact << 31, and the integer trunc(w * 1000) in the encoding of ew_encode
    HIPCHK(c, hipMemcpy(esrci.data(), (uint32_t *)c->b_esrc.p + m.edge_off, E * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ewi.data(), (long long *)c->b_ew.p + m.edge_off, E * 8, hipMemcpyDeviceToHost));
    const bool have_ref = c->certify && c->cert_done && c->b_eref.p; // k_refine has looked at the flagged edges of this run
    std::vector<DERef> eref(have_ref ? E : 0);
    if (have_ref) HIPCHK(c, hipMemcpy(eref.data(), (DERef *)c->b_eref.p + m.edge_off, E * sizeof(DERef), hipMemcpyDeviceToHost));
    for (size_t v = 0; v < V; v++)
        for (uint32_t e = in_off[v]; e < in_off[v + 1]; e++) {
            esrc[e] = ESRC_NODE(esrc[e]); // (the tap variant carries the off-path flag of a source-node edge)
            out[e].src = (int32_t)esrc[e]; out[e].dst = (int32_t)v; out[e].w = ew[e];
            // the tap's weight and the solver's integer are two results of one computation: any difference is an error of the library
            const double t = std::trunc(ew[e] * 1000.0);
            const long long x = ewi[e];
            const bool wide = (((unsigned long long)x >> 63) ^ ((unsigned long long)x >> 62)) & 1ull;
            bool same = ESRC_NODE(esrci[e]) == esrc[e];
            if (wide) { long long bits; memcpy(&bits, &t, 8); same = same && ((x | (1ll << 62)) == bits); }
            else same = same && std::fabs(t) < 4611686018427387904.0 && (long long)t == x;
            if (!same) {
                char msg[256];
                snprintf(msg, sizeof(msg), "edge tap: recomputed weight differs from the solver's integer (contig %d edge %u: w %.17g, record %016llx, sources %u / %u)", contig, e, ew[e], (unsigned long long)x, esrci[e], esrc[e]);
                c->err = msg; return PHX_E_STATE;
            }
            out[e].inexact = (int32_t)(esrci[e] >> 31);
            out[e].pad = 0; out[e].d1 = 0; out[e].d2 = 0; out[e].err = 0;
            if (have_ref && out[e].inexact) { out[e].d1 = eref[e].d1; out[e].d2 = eref[e].d2; out[e].err = eref[e].err; }
        }
    return PHX_OK;
}

int phx_tap_path(phx_ctx *c, int32_t contig, int32_t *path, int32_t cap, int32_t *n_path, uint64_t *dist_limbs, int32_t cap_limbs) {
    TAP_PRE(c, contig);
    if (n_path) *n_path = 0;
    if (m.status < 0 || m.n_path == 0) return PHX_OK;
    if (n_path) *n_path = m.n_path;
    if (path) {
        if (cap < m.n_path) return PHX_E_ARG;
        HIPCHK(c, hipMemcpy(path, (int32_t *)c->b_path.p + m.node_off, (size_t)m.n_path * 4, hipMemcpyDeviceToHost));
    }
    if (dist_limbs) {
        if (cap_limbs < m.sssp_nl) return PHX_E_ARG;
        const size_t tgt = (size_t)m.node_off * (size_t)c->n_limbs + ((size_t)m.n_node - 1) * (size_t)m.sssp_nl;
        HIPCHK(c, hipMemcpy(dist_limbs, (uint64_t *)c->b_dist.p + tgt, (size_t)m.sssp_nl * 8, hipMemcpyDeviceToHost));
    }
    return PHX_OK;
}

int phx_tap_dist(phx_ctx *c, int32_t contig, uint64_t *dist_limbs, int64_t cap_words) {
    TAP_PRE(c, contig);
    if (m.status < 0 || m.n_node <= 2) return PHX_OK;
    const size_t words = (size_t)m.n_node * (size_t)m.sssp_nl;
    if (!dist_limbs || cap_words < (int64_t)words) return PHX_E_ARG;
    HIPCHK(c, hipMemcpy(dist_limbs, (uint64_t *)c->b_dist.p + (size_t)m.node_off * (size_t)c->n_limbs, words * 8, hipMemcpyDeviceToHost));
    return PHX_OK;
}

// ---- the solver alone ----
} // extern "C" (the replay is C++)
#include "phx_exact.inc"
extern "C" {

// everything the replay reads of contig i, through the taps (exact device output) + the solver's own integers and the bases
static int exact_fetch(phx_ctx *c, int32_t contig, ExactIn &in) {
    int rc;
    if ((rc = phx_tap_globals(c, contig, &in.gl))) return rc;
    in.L = (int)in.gl.L;
    in.nl = in.gl.n_limbs;
    in.nd.resize((size_t)std::max(in.gl.n_node, 0)); in.ed.resize((size_t)std::max(in.gl.n_edge, 0)); in.orf.resize((size_t)std::max(in.gl.n_orf, 0));
    in.gcc.resize((size_t)in.L); in.base.resize((size_t)in.L);
    if ((rc = phx_tap_nodes(c, contig, in.nd.data()))) return rc;
    if ((rc = phx_tap_edges(c, contig, in.ed.data()))) return rc;
    if ((rc = phx_tap_orfs(c, contig, in.orf.data()))) return rc;
    if ((rc = phx_tap_positions(c, contig, nullptr, in.gcc.data(), nullptr, nullptr))) return rc;
    const DMeta &m = c->meta[(size_t)contig];
    in.ewi.resize(in.ed.size());
    if (!in.ewi.empty()) HIPCHK(c, hipMemcpy(in.ewi.data(), (long long *)c->b_ew.p + m.edge_off, in.ewi.size() * 8, hipMemcpyDeviceToHost));
    if ((rc = fetch_codes(c, m, in.base))) return rc;
    return PHX_OK;
}

// The certificate, and for every contig it leaves open the reference's own arithmetic on the host (one worker thread per contig, at
// most 32): phx_download* and phx_certified call this; its results replace the contig's genes.
static int ensure_exact(phx_ctx *c) {
    if (!c->certify || !c->exact || c->n == 0) return PHX_OK; // (exact off: the downloads hand out the device's lists, no certificate is computed for them)
    { const int rc = ensure_cert(c); if (rc) return rc; }
    if (c->exact_done) return PHX_OK;
    try { // (nothing may cross the C-ABI as an exception: the vectors below and the worker threads can run out of memory)
    std::vector<int> todo;
    // what the certificate left open, and the contigs no device kernel could take (DMeta.sssp_mode 4: path sums beyond 1088 bits — the
    // only ones whose full record is needed here: the 0.5 KB per contig come over only when some contig reports that status)
    bool any_overflow = false;
    for (int i = 0; i < c->n && !any_overflow; i++) any_overflow = c->res[(size_t)i].status == PHX_S_OVERFLOW;
    if (any_overflow) { const int rm = fetch_meta(c); if (rm) return rm; }
    for (int i = 0; i < c->n; i++) {
        const DRes &r = c->res[(size_t)i];
        if ((r.status >= 0 && r.cert == 0) || (r.status == PHX_S_OVERFLOW && any_overflow && c->meta[(size_t)i].status == 0 && c->meta[(size_t)i].sssp_mode == 4)) todo.push_back(i);
    }
    if (!todo.empty()) {
        std::vector<ExactIn> in(todo.size());
        std::vector<ExactOut> out(todo.size());
        for (size_t k = 0; k < todo.size(); k++) { const int rc = exact_fetch(c, todo[k], in[k]); if (rc) return rc; }
        const phx_params par = c->params;
        std::atomic<size_t> next{0};
        std::atomic<bool> oom{false};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= todo.size()) return;
                try { exact_solve(in[k], par, out[k]); } catch (...) { out[k].failed = true; oom = true; }
            }
        };
        unsigned nt = std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 32) nt = 32;
        if (nt > todo.size()) nt = (unsigned)todo.size();
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) { try { th.emplace_back(work); } catch (...) { break; } }
        work();
        for (std::thread &t : th) t.join();
        if (oom.load()) { c->err = "out of memory in the host re-solve"; return PHX_E_NOMEM; }
        for (size_t k = 0; k < todo.size(); k++) {
            if (out[k].failed) { c->exact_failed++; continue; }
            c->exact_genes[todo[k]] = std::move(out[k].genes);
            c->res[(size_t)todo[k]].cert = 2;
            if (c->res[(size_t)todo[k]].status == PHX_S_OVERFLOW) { c->res[(size_t)todo[k]].status = 0; c->host_only.push_back(todo[k]); }
        }
    }
    } catch (const std::bad_alloc &) { c->err = "out of memory in the host re-solve"; return PHX_E_NOMEM; }
    c->exact_done = true;
    return PHX_OK;
}

} // extern "C"
#include "phx_dd.h"
extern "C" {
// test hooks for the double-double arithmetic of k_refine (phx_dd.h compiles for the host as well): tests/test_dec.py
int phx_dd_eval(const char *op, double ah, double al, double bh, double bl, double *rh, double *rl) {
    if (!op || !rh || !rl) return PHX_E_ARG;
    const dd_t a = dd_make(ah, al), b = dd_make(bh, bl);
    dd_t r = dd_make(0, 0);
    if (!strcmp(op, "add")) r = dd_add(a, b);
    else if (!strcmp(op, "sub")) r = dd_sub(a, b);
    else if (!strcmp(op, "mul")) r = dd_mul(a, b);
    else if (!strcmp(op, "div")) r = dd_div(a, b);
    else if (!strcmp(op, "exp")) r = dd_exp(a);
    else if (!strcmp(op, "log")) r = dd_log(a);
    else if (!strcmp(op, "repr")) { int ok; r = dd_repr_value(ah, &ok); if (!ok) return PHX_E_ARG; }
    else return PHX_E_ARG;
    *rh = r.hi; *rl = r.lo;
    return PHX_OK;
}
int phx_dd_shortest(double x, uint64_t *digits, int32_t *exp10) {
    if (!digits || !exp10) return PHX_E_ARG;
    int e = 0;
    const int nd = dd_shortest(x, digits, &e);
    *exp10 = e;
    return nd;
}

int phx_dump_text(phx_ctx *c, int32_t contig, char **text, int64_t *text_len) {
    if (!c || !text) return PHX_E_ARG;
    *text = nullptr;
    if (text_len) *text_len = 0;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (!c->ran) return PHX_E_STATE;
    if (contig < 0 || contig >= c->n) return PHX_E_ARG;
    ExactIn in;
    { const int rc = exact_fetch(c, contig, in); if (rc) return rc; }
    std::string out;
    if (in.gl.status >= 0 && !exact_dump(in, c->params, out)) { c->err = "dump: an operation outside what phx_dec.c restates"; return PHX_E_STATE; }
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) return PHX_E_NOMEM;
    memcpy(buf, out.data(), out.size()); buf[out.size()] = 0;
    *text = buf;
    if (text_len) *text_len = (int64_t)out.size();
    return PHX_OK;
}

int phx_solve(phx_ctx *c, int32_t V, int32_t E, const int32_t *src, const int32_t *dst, const uint64_t *w_limbs, int32_t n_limbs,
              int32_t source, int32_t target, int32_t *path_out, int32_t cap, int32_t *n_path, uint64_t *dist_limbs) {
    if (!c || V < 2 || E < 0 || (E > 0 && (!src || !dst || !w_limbs)) || !n_path) return PHX_E_ARG;
    if (V >= (1 << 29)) return PHX_E_ARG; // (a source word holds 29 bits of node id: bit 29 marks a coded gap edge, phx_internal.h)
    if (!(n_limbs == 2 || n_limbs == 4 || n_limbs == 8 || n_limbs == 17)) return PHX_E_ARG;
    if (source < 0 || source >= V || target < 0 || target >= V || source == target) return PHX_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->in_flight) (void)settle(c);
    *n_path = 0;
    // device node order: everything else in given order, then source (V-2), then target (V-1)
    std::vector<int32_t> to_dev((size_t)V), to_user((size_t)V);
    {
        int k = 0;
        for (int v = 0; v < V; v++) if (v != source && v != target) { to_dev[(size_t)v] = k; to_user[(size_t)k] = v; k++; }
        to_dev[(size_t)source] = V - 2; to_user[(size_t)V - 2] = source;
        to_dev[(size_t)target] = V - 1; to_user[(size_t)V - 1] = target;
    }
    std::vector<uint32_t> in_off((size_t)V + 1, 0), esrc((size_t)E);
    std::vector<uint64_t> ewl((size_t)E * (size_t)n_limbs);
    for (int e = 0; e < E; e++) {
        if (src[e] < 0 || src[e] >= V || dst[e] < 0 || dst[e] >= V) return PHX_E_ARG;
        in_off[(size_t)to_dev[(size_t)dst[e]] + 1]++;
    }
    for (int v = 0; v < V; v++) in_off[(size_t)v + 1] += in_off[(size_t)v];
    {
        std::vector<uint32_t> fill(in_off.begin(), in_off.end() - 1);
        for (int e = 0; e < E; e++) { // stable: in-edge order of a node = order of the caller's edge list
            uint32_t k = fill[(size_t)to_dev[(size_t)dst[e]]]++;
            esrc[k] = (uint32_t)to_dev[(size_t)src[e]];
            memcpy(&ewl[(size_t)k * (size_t)n_limbs], &w_limbs[(size_t)e * (size_t)n_limbs], (size_t)n_limbs * 8);
        }
    }
    std::vector<uint32_t> ekey((size_t)E);
    {
        std::vector<uint32_t> fill(in_off.begin(), in_off.end() - 1);
        for (int e = 0; e < E; e++) ekey[fill[(size_t)to_dev[(size_t)dst[e]]]++] = (uint32_t)e; // same stable placement as above: the caller's index of every device edge
    }
    DMeta m;
    memset(&m, 0, sizeof(m));
    m.n_node = V; m.n_edge = E; m.L = 1; m.sssp_nl = n_limbs; m.sssp_mode = 0;
    int rc;
    DevBuf &b_meta = c->b_meta;
    if ((rc = ensure(c, b_meta, sizeof(DMeta) * 2))) return rc;
    if ((rc = ensure(c, c->b_inoff, ((size_t)V + 2) * 4))) return rc;
    if ((rc = ensure(c, c->b_esrc, ((size_t)E + 1) * 4))) return rc;
    if ((rc = ensure(c, c->b_ekey, ((size_t)E + 1) * 4))) return rc;
    if ((rc = ensure(c, c->b_ew, ((size_t)E + 1) * 8))) return rc;
    if ((rc = ensure(c, c->b_ewl, ((size_t)E + 1) * 8 * (size_t)n_limbs))) return rc;
    if ((rc = ensure(c, c->b_dist, ((size_t)V + 1) * 8 * (size_t)n_limbs))) return rc;
    if ((rc = ensure(c, c->b_parent, ((size_t)V + 1) * 4))) return rc;
    if ((rc = ensure(c, c->b_path, ((size_t)V + 1) * 4))) return rc;
    if ((rc = ensure(c, c->b_tie, (size_t)V * 40 + (size_t)E * 12 + 64))) return rc; // k_inorder's worst case for one graph
    if ((rc = ensure(c, c->b_tot, sizeof(DTotals)))) return rc;
    c->uploaded = false; c->ran = false; // the batch buffers are being reused
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(c->b_tot.p, 0, sizeof(DTotals), s));
    HIPCHK(c, hipMemcpyAsync(b_meta.p, &m, sizeof(m), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->b_inoff.p, in_off.data(), ((size_t)V + 1) * 4, hipMemcpyHostToDevice, s));
    if (E) {
        HIPCHK(c, hipMemcpyAsync(c->b_esrc.p, esrc.data(), (size_t)E * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_ekey.p, ekey.data(), (size_t)E * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->b_ewl.p, ewl.data(), (size_t)E * 8 * (size_t)n_limbs, hipMemcpyHostToDevice, s));
    }
    DBatch b;
    memset(&b, 0, sizeof(b));
    b.n_contig = 1;
    b.meta = (DMeta *)b_meta.p;
    b.tot = (DTotals *)c->b_tot.p;
    b.in_off = (uint32_t *)c->b_inoff.p; b.esrc = (uint32_t *)c->b_esrc.p; b.ew = (long long *)c->b_ew.p;
    b.ewl = (const uint64_t *)c->b_ewl.p;
    b.ekey = (const uint32_t *)c->b_ekey.p;
    b.dist = (uint64_t *)c->b_dist.p; b.parent = (int32_t *)c->b_parent.p;
    b.path = (int32_t *)c->b_path.p; // no b.genes: the kernels stop at the path
    b.tie = (uint8_t *)c->b_tie.p; b.tie_cap = cap_of(c->b_tie, 1, 0);
    b.dist_stride = n_limbs;
    // relaxation (k_sssp), walk (k_path), parents of the caller's relaxation order where equal-length paths exist (k_inorder)
    hipLaunchSsspOnly(&b, n_limbs, s);
    HIPCHK(c, hipGetLastError());
    std::vector<int32_t> path((size_t)V);
    std::vector<uint64_t> dt((size_t)n_limbs);
    HIPCHK(c, hipMemcpyAsync(&m, b_meta.p, sizeof(m), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(path.data(), c->b_path.p, (size_t)V * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(dt.data(), (uint64_t *)c->b_dist.p + (size_t)(V - 1) * (size_t)n_limbs, (size_t)n_limbs * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (m.status < 0) return m.status;
    if (m.n_path < 2) return PHX_OK; // unreachable: *n_path = 0
    if (m.n_path > cap && path_out) return PHX_E_ARG;
    *n_path = m.n_path;
    if (path_out) for (int k = 0; k < m.n_path; k++) path_out[k] = to_user[(size_t)path[(size_t)k]];
    if (dist_limbs) memcpy(dist_limbs, dt.data(), (size_t)n_limbs * 8);
    return PHX_OK;
}

// ---- measurement ----
int phx_set_profiling(phx_ctx *c, int on) {
    if (!c) return PHX_E_ARG;
    c->prof = on != 0;
    c->prof_mask = 0xffffffffu;
    return PHX_OK;
}
int phx_set_profiling_stages(phx_ctx *c, uint32_t stage_mask) {
    if (!c) return PHX_E_ARG;
    c->prof = stage_mask != 0;
    c->prof_mask = stage_mask;
    return PHX_OK;
}
int phx_get_stage_ms(phx_ctx *c, float *ms, int32_t *launches, int reset) {
    if (!c) return PHX_E_ARG;
    for (int i = 0; i < PHX_N_STAGES; i++) { if (ms) ms[i] = c->stage_ms[i]; if (launches) launches[i] = c->stage_n[i]; }
    if (reset) for (int i = 0; i < PHX_N_STAGES; i++) { c->stage_ms[i] = 0; c->stage_n[i] = 0; }
    return PHX_OK;
}
const char *phx_stage_name(int k) { return k >= 0 && k < PHX_N_STAGES ? kStageName[k] : ""; }
int64_t phx_seg_runs(phx_ctx *c) {
    if (!c) return 0;
    if (c->in_flight && phx_wait(c)) return 0;
    return c->seg_runs;
}
int64_t phx_seg_fallbacks(phx_ctx *c) {
    if (!c) return 0;
    if (c->in_flight && phx_wait(c)) return 0;
    return c->seg_fallbacks;
}
// development: the records of the segments of contig `i` in the run last made — 8 ints each: windows | done flag, solver status, first node, end node, solver time
// (10 ns ticks), phases, packs, step-backs; returns the number of records written (0: the run did not use segments)
int phx_seg_stats(phx_ctx *c, int32_t i, int32_t *out, int32_t cap_records) {
    if (!c || !out) return 0;
    if (c->in_flight && phx_wait(c)) return 0;
    if (i < 0 || i >= c->n || !c->ran || !c->pend_seg || !c->b_segw.p) return 0;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    const int K = c->pend_seg_k;
    const int n = std::min(K, (int)cap_records);
    if (n <= 0) return 0;
    if (hipMemcpy(out, (const int32_t *)c->b_segw.p + (size_t)i * (size_t)K * 8, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}
int64_t phx_front_runs(phx_ctx *c) {
    if (!c) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    return c->front_off ? -c->front_runs - 1 : c->front_runs;
}

int64_t phx_plan_timeouts(phx_ctx *c) {
    if (!c) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    return c->plan_timeouts;
}

int phx_batch_sizes(phx_ctx *c, int64_t *L, int64_t *n_orf, int64_t *n_node, int64_t *n_edge) {
    if (!c) return PHX_E_ARG;
    if (c->in_flight) { const int rs = phx_wait(c); if (rs) return rs; }
    if (L) *L = c->totalL;
    if (n_orf) *n_orf = c->tot_orf;
    if (n_node) *n_node = c->tot_node;
    if (n_edge) *n_edge = c->tot_edge;
    return PHX_OK;
}

} // extern "C"
