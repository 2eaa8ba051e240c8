// psgdk.hip -- C-ABI implementation (see include/psgdk.h).  One translation unit; kernels live in the .hiph files.
//
// A plan lays every tensor of an optimizer (or one tensor, for the functional seam) out in two caller-owned arenas
// and pre-builds, once, the grouped-GEMM problem/tile tables of every stage.  A step is then ~26 grouped launches
// over ALL tensors (the reference issues ~100 ATen launches per tensor), with every scalar kept on the device.
#include <hip/hip_ext.h>
#include "host_util.hiph"
#include "descs.hiph"
#include "kernels_ew.hiph"
#include "kernels_dense.hiph"
#include "kernels_lra.hiph"
#include "kernels_lra_pk.hiph"
#include "kernels_lra_gen.hiph"
#include "kernels_gen.hiph"
#include "kernels_eq.hiph"
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <memory>

namespace {

struct Stage {                       // one grouped GEMM launch
    std::vector<GemmProblem> probs;
    GemmProblem* d_probs = nullptr;
    GemmTile* d_tiles = nullptr;
    unsigned n_tiles = 0;
    bool big = false;                // 256x256 / 8-wave tiling (gemm_nt_pipe_kernel)
    bool one_per_tile = false;       // tests / experiments: the staggered-phase kernel with one workgroup per tile (not persistent)
    bool ext = false;                // problems use GemmProblem::skip / GF_PROCR3 (PRO4P): the EXT instantiation, small tiling
    bool ksplit = false;             // small launches: 64 x 64 tiles, K split over the four waves (gemm_nt_ks_kernel)
};

struct FactorRef { int kind; int idx; };   // idx into dd (diag/scalar) or dn (dense)

}  // namespace

struct psgdk_plan {
    int n_tensors = 0, dtype = 0, use_momentum = 0;
    size_t esz = 2;
    double max_size = 0, max_skew = 0;
    std::vector<TensorDesc> td;
    std::vector<DiagDesc> dd;
    std::vector<DenseDesc> dn;
    std::vector<std::vector<FactorRef>> factors;    // per tensor, logical dim order
    std::vector<int> order;                          // number of factors per tensor (k of scale^(1/k))
    std::vector<int> dense_dim;                      // per dense factor: logical dim index
    std::vector<GenDesc> gd;                         // tensors with > 2 dims
    GenDesc* d_gd = nullptr;
    std::vector<int> gram_prob;                      // per dense factor: its problem in g_gram, or -1 (N-D tensors)
    size_t state_bytes = 0, work_bytes = 0;
    size_t zero_off = 0, zero_bytes = 0, hsumsq_off = 0, balnorm_off = 0, diag_mu_off = 0;
    // the balancing norms live inside the zero region: an update's one memset leaves them clean for the balancing step at its end
    // (the sums of h^2 stay outside: an update may run between precond_grad and the consumers of h)
    bool bal_clean = false;
    int max_diag_len = 0;
    unsigned char* state = nullptr;
    unsigned char* work = nullptr;
    // device tables owned by the plan
    TensorDesc* d_td = nullptr; DiagDesc* d_dd = nullptr; DenseDesc* d_dn = nullptr;
    EwTile* d_tiles_all = nullptr; unsigned n_tiles_all = 0;
    std::vector<unsigned> tile_begin;                // per tensor range in d_tiles_all
    EwTile* d_tiles_diag = nullptr; unsigned n_tiles_diag = 0;
    PtrTableCache ptrs_a, ptrs_b;                           // device copies of the callers' pointer tables (gradients; parameters / outputs)
    bool zero_clean = false, hsq_clean = false;             // psgdk_accumulate zeroed the update's accumulators / the sums of h^2 in its own pass
    hipStream_t clean_stream = nullptr;                     // ... on this stream: a consumer on another stream clears them itself
    std::vector<const void*> h_noise_a, h_noise_b;          // staging for explicit-noise pointer tables
    std::vector<void*> h_dump_g;                            // staging of psgdk_test_dump_noise's output table
    std::vector<int> h_balance;
    void** d_noise_g = nullptr; void** d_noise_spd = nullptr; void** d_noise_skh = nullptr;
    float* d_scale_diag = nullptr; float* d_scale_dense = nullptr;
    int* d_balance = nullptr;
    int max_dp = 0;
    bool nlb_coop = false;            // the cooperative one-launch norm bound is usable for this plan
    bool nlb_small = false;           // ... in its instantiation for plans whose widest factor is <= 128 (one workgroup per factor, 16 columns per wave)
    bool nlb_k32 = false;             // ... factors of 25 .. 32 K steps (bf16 d <= 1024, fp32 d <= 512): members of 128 columns with 32 K steps of registers
    int nlb_cols = 256;               // columns of A per member of the cooperative bound: 128 (nlb_small / nlb_narrow), 192 (twelve waves: round 6), 256
    bool nlb_narrow = false;          // ... with 128 columns of A per member instead of 256 (round 6): plans with few wide factors -- a rank's share
                                      //     of a sharded job, a model of a few layers -- whose 2 x as many members still fit the CUs
    NlbJob* d_nlb_jobs = nullptr; unsigned n_nlb_jobs = 0, nlb_lds = 0;
    unsigned long long* d_nlb_ts = nullptr;   // psgdk_test_nlb_stamps: NLB_TS_SLOTS words per workgroup (its address sits after the job table)
    // error word of the cooperative kernels: host-mapped pinned memory, so that the host can look at it WITHOUT synchronising
    // (read at the start of every update call; see nlb_check_error)
    volatile unsigned* h_err = nullptr; unsigned* d_err = nullptr;
    int64_t nlb_fallbacks = 0;        // how often a timeout moved this plan to the multi-launch route (0 or 1)
    bool nlb_unfused = false;        // PSGDK_NLB_FUSED=0 at plan creation: keep the multi-launch route (tests compare the two)
    int ew_dbg = 0;                  // psgdk_test_ew_mode
    bool no_fuse = false;            // psgdk_test_fuse_mode(0): psgdk_precond_grad_apply takes the two-pass route (A/B runs, parity tests)
    bool p_valid = false;
    bool x_valid = false, x_explicit = false; int x_source = 0; float x_damping = 0.f; uint64_t x_seed = 0, x_offset = 0;
    Stage g_P, g_upd_a, g_upd_b, g_gram, g_qupd, g_rq, g_rrq, g_app_a[2], g_app_b, g_nlb[2][4];
    // psgdk_precond_grad_apply (round 6): the apply's last product with the parameter update fused into its epilogue (GemmProblem::upd_p).
    // The stages are copies of g_app_a[src] / g_app_b with the callers' parameter pointers in them, rebuilt when those change (fused_key).
    Stage f_app_a, f_app_b;
    std::vector<int> app_a_tensor, app_b_tensor;     // per problem of g_app_a[.] / g_app_b: its tensor
    std::vector<const void*> fused_key;              // [params..., (void*)src] the fused stages were built for; empty = not built
    int fused_rebuilds = 0;
    int fuse_stagger = -1;           // GemmUpdArgs::stagger of the fused launches, 100 MHz ticks; -1 = the default (scaled with K)
    bool fused_any = false;                          // at least one tensor takes the fused epilogue (otherwise the call runs unfused)
    bool h_fused = false;                            // the work arena's h was produced by the fused stages: fused tensors' h is in LOGICAL
                                                     // orientation and already applied -- read / export / apply_update refuse it
    EwTile* d_tiles_rest = nullptr; unsigned n_tiles_rest = 0;      // streaming tiles of the tensors the fused epilogue does not cover
    FixDesc* d_fix = nullptr; unsigned n_fix = 0;
    std::vector<int> split_dense;                    // dense factors whose Gram is split-K
    // row shards (psgdk_plan_set_row_shard): tensors of this plan that are row blocks of a larger, row-sharded tensor.  Their dense
    // factor is replicated on the `members` owners of the blocks and fitted to the WHOLE tensor: the members' partial mode Grams
    // (fp32) and their maxima of the diagonal factor's term1 are exchanged between the two halves of a phased update.
    struct RowShard { int tensor; int64_t global_rows, row0; size_t rec_off; };
    std::vector<RowShard> shards;
    int shard_member = 0, shard_members = 0;
    size_t xchg_record_bytes = 0;                    // one member's record: per shard [dp x dp fp32 partial Gram][64 fp32 scalars]
    bool update_open = false;                        // between psgdk_update_precond_begin and _finish
    int shard_of(int t) const { for (size_t i = 0; i < shards.size(); ++i) if (shards[i].tensor == t) return (int)i; return -1; }
    // PSGDK_GEOM_EQ (psgd.py:278-336): A = (kron Q) Hvp in two products, Grams of A and B, Q -= mu triu(.) Q; the right
    // triangular solves run in two phases (column-side factors on V, then row-side factors on the transposed result)
    int geometry = PSGDK_GEOM_Q0P5EQ1P5;
    bool p_mode() const { return geometry == PSGDK_GEOM_QUAD4P || geometry == PSGDK_GEOM_PRO4P; }   // the factors ARE P (psgd.py:422-452, 486-513)
    // default geometry: the update + Procrustes chain runs in transposed space (see psgdk_plan_bind)
    bool chain_t() const { return geometry == PSGDK_GEOM_Q0P5EQ1P5; }
    Stage e_a1, e_a2, e_g1, e_g2, e_qupd;
    Stage v_qeq, v_quad2;                            // PSGDK_GEOM_QEQ: Q term1;  PSGDK_GEOM_QUAD: the second half step
    Stage v_qep_u, v_qep_t1, v_qep_t2;               // PSGDK_GEOM_QEP: Q term1, (Q term1) Q^T, c Q Q^T
    Stage v_pro_rq, v_pro_rrq, v_pro_rrrq;           // PSGDK_GEOM_PRO4P: the three products of procrustes_step3
    std::vector<int> e_gram_prob;                    // per dense factor: index into e_g1 / e_g2
    TrsmJob* d_trsm[2] = {nullptr, nullptr}; TrsmTile* d_trsm_tiles[2] = {nullptr, nullptr};
    unsigned n_trsm_tiles[2] = {0, 0};
    UinvJob* d_uinv = nullptr; unsigned n_uinv = 0;
    // optional live profiling of the grouped-GEMM launches (bench.py roofline line)
    bool prof = false;            // event pairs on the grouped-GEMM launches (ride on the dispatch packets: free)
    bool prof_calls = false;      // event pairs around every hot-path call (hipEventRecord: they fence the stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev, prof_call_ev;
    size_t prof_used = 0, prof_call_used = 0;
    std::vector<char> prof_fused;     // per profiled launch: it carried the fused parameter update (psgdk_profile_read_fused)

    std::vector<Stage*> all_stages() {
        std::vector<Stage*> v = {&g_P, &g_upd_a, &g_upd_b, &g_gram, &g_qupd, &g_rq, &g_rrq, &g_app_a[0], &g_app_a[1], &g_app_b, &f_app_a, &f_app_b};
        for (int c = 0; c < 2; ++c) for (int p = 0; p < 4; ++p) v.push_back(&g_nlb[c][p]);
        for (Stage* e : {&e_a1, &e_a2, &e_g1, &e_g2, &e_qupd, &v_qeq, &v_quad2, &v_qep_u, &v_qep_t1, &v_qep_t2, &v_pro_rq, &v_pro_rrq, &v_pro_rrrq}) v.push_back(e);
        return v;
    }

    ~psgdk_plan() {
        auto fr = [](void* p) { if (p) (void)hipFree(p); };
        fr(d_td); fr(d_dd); fr(d_dn); fr(d_tiles_all); fr(d_tiles_diag);
        fr(d_noise_g); fr(d_noise_spd); fr(d_noise_skh); fr(d_scale_diag); fr(d_scale_dense); fr(d_balance); fr(d_gd);
        for (auto& e : prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : prof_call_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (Stage* s : all_stages()) { fr(s->d_probs); fr(s->d_tiles); }
        for (int k = 0; k < 2; ++k) { fr(d_trsm[k]); fr(d_trsm_tiles[k]); }
        fr(d_uinv); fr(d_nlb_jobs); fr(d_nlb_ts); fr(d_tiles_rest); fr(d_fix);
        if (h_err) (void)hipHostFree((void*)h_err);
    }
};

namespace {

template <typename X>
int upload(X** dst, const std::vector<X>& v) {
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    if (v.empty()) return PSGDK_OK;
    HIPCHK(hipMalloc((void**)dst, v.size() * sizeof(X)));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(X), hipMemcpyHostToDevice));
    return PSGDK_OK;
}

int finish_stage(Stage& s) {
    TileTableBuilder tb;
    tb.bm = s.big ? GEMM_BIG_BM : (s.ksplit ? 64 : GEMM_BM);
    tb.bn = s.big ? GEMM_BIG_BN : (s.ksplit ? 64 : GEMM_BN);
    for (size_t i = 0; i < s.probs.size(); ++i) tb.add_problem((int)i, s.probs[i]);
    std::vector<GemmTile> tiles = tb.finish();
    s.n_tiles = (unsigned)tiles.size();
    int rc = upload(&s.d_probs, s.probs);
    if (rc) return rc;
    return upload(&s.d_tiles, tiles);
}

// A stage whose problems make at most this many tiles of 128 x 128 runs on 64 x 64 tiles with the K loop split over the workgroup's waves
// (gemm_nt_ks_kernel; round 4, measured: LeNet5's step 0.234 -> 0.195 ms, profiles/r04_a_ksplit.md)
static constexpr int64_t kKsplitMaxTiles = 64;
static int64_t big_min_tiles() {      // (read at every bind: the tests force the big tiling onto small plans with it)
    const char* e = getenv("PSGDK_BIG_MIN_TILES");
    return e ? (int64_t)atoll(e) : (int64_t)768;
}

// the staggered-phase kernel runs persistently (one workgroup per CU walks the tile table) unless the stage asks for one workgroup
// per tile (test hook)
static unsigned persistent_grid(unsigned n_tiles) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n - n % 8;          // a multiple of 8: a workgroup's stride keeps it on one XCD's queue of the interleaved table
    }();
    return (cus > 0 && n_tiles > (unsigned)cus) ? (unsigned)cus : n_tiles;
}

// (e0, e1: optional events that take the kernel's own start / stop timestamps -- hipExtLaunchKernelGGL attaches them to the dispatch
//  packet, so a profiled launch costs no extra packets on the stream, unlike a hipEventRecord pair, which fences it: 0.13 ms per step)
#define PSGDK_LAUNCH(KERNEL, GRID, BLOCK, ...)                                                               \
    do {                                                                                                     \
        if (e0) { ++g_psgdk_launches; hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, e0, e1, 0, __VA_ARGS__); }   \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, __VA_ARGS__);                                    \
    } while (0)
template <typename T>
void launch_stage_t(const Stage& s, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const GemmUpdArgs upd = GemmUpdArgs{0.f, 1.f, 0.f, 0, 0, 0}) {
    if (!s.n_tiles) return;
    if (s.big) PSGDK_LAUNCH(gemm_nt_pipe_kernel<T>, dim3(s.one_per_tile ? s.n_tiles : persistent_grid(s.n_tiles)), dim3(512),
                                 s.d_probs, s.d_tiles, (int)s.n_tiles, upd);
    else if (s.ksplit) PSGDK_LAUNCH(gemm_nt_ks_kernel<T>, dim3(s.n_tiles), dim3(256), s.d_probs, s.d_tiles);
    else if (s.ext) PSGDK_LAUNCH((gemm_nt_kernel<T, true>), dim3(s.n_tiles), dim3(256), s.d_probs, s.d_tiles, upd);
    else PSGDK_LAUNCH((gemm_nt_kernel<T, false>), dim3(s.n_tiles), dim3(256), s.d_probs, s.d_tiles, upd);
}
void launch_stage(psgdk_plan* p, const Stage& s, hipStream_t st, const GemmUpdArgs* upd = nullptr) {
    if (!s.n_tiles) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p->prof) {
        if (p->prof_used == p->prof_ev.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            p->prof_ev.emplace_back(a, b);
        }
        e0 = p->prof_ev[p->prof_used].first; e1 = p->prof_ev[p->prof_used].second;
        if (p->prof_fused.size() <= p->prof_used) p->prof_fused.resize(p->prof_used + 1);
        p->prof_fused[p->prof_used] = (upd && upd->lr > 0.f) ? 1 : 0;
        ++p->prof_used;
    }
    const GemmUpdArgs u = upd ? *upd : GemmUpdArgs{0.f, 1.f, 0.f, 0, 0, 0};
    if (p->dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st, e0, e1, u); else launch_stage_t<float>(s, st, e0, e1, u);
}

// profiling (psgdk_profile_enable): an event pair around one hot-path call -- recorded on entry and when the call returns
struct ProfCall {
    psgdk_plan* p; hipStream_t st; bool on;
    ProfCall(psgdk_plan* plan, void* stream) : p(plan), st((hipStream_t)stream), on(plan && plan->prof_calls && plan->state) {
        if (!on) return;
        if (p->prof_call_used == p->prof_call_ev.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            p->prof_call_ev.emplace_back(a, b);
        }
        (void)hipEventRecord(p->prof_call_ev[p->prof_call_used].first, st);
    }
    ~ProfCall() { if (on) (void)hipEventRecord(p->prof_call_ev[p->prof_call_used++].second, st); }
};

#define DISPATCH_T(plan, CALL)                         \
    do {                                               \
        if ((plan)->dtype == PSGDK_BF16) { typedef bf16_t T; CALL; } else { typedef float T; CALL; } \
    } while (0)

// (kron_i F_i) applied mode by mode to an N-D tensor (psgd.py:251-252, exprP): F_i = P_i = Q_i^T Q_i (dense, from the
// batched P stage) or diag(q_i^2).  src -> ping-pong buffers -> dst (dst may be one of the ping-pong buffers' owner h).
// Returns the buffer holding the result when dst == nullptr.
// what: 0 = P (Q^T Q / a^2, psgd.py:251-252 exprP), 1 = Q itself (a; exprA, psgd.py:248-249), 2 = the solves X inv(Q_i)
// (X / a; psgd.py:297-303).  The ping-pong pair `pp` keeps the result alive while another chain runs through the other pair.
template <typename T>
static const T* gen_apply_chain(psgdk_plan* P, const GenDesc& g, const T* src, T* dst, float* sumsq, hipStream_t st, int what = 0,
                                int pp = 0) {
    const int64_t numel = P->td[g.tensor].numel;
    const T* cur = src;
    int64_t A = 1;
    for (int i = 0; i < g.ndim; ++i) {
        const int s_ = g.dims[i];
        const int64_t B = numel / (A * s_);
        const bool last = (i == g.ndim - 1);
        T* out = (last && dst) ? dst : (T*)(P->work + (pp ? g.pp2_off[i & 1] : g.pp_off[i & 1]));
        const bool dense = g.fkind[i] == PSGDK_DENSE;
        const T* F = dense ? (what == 0 ? (const T*)(P->work + P->dn[g.fidx[i]].p_off) : (const T*)(P->state + P->dn[g.fidx[i]].q_off))
                           : (const T*)(P->state + P->dd[g.fidx[i]].a_off);
        const int ldf = dense ? P->dn[g.fidx[i]].dp : 0;
        if (what == 2) {
            const unsigned gb = (unsigned)std::min<int64_t>((A * B + 63) / 64, 4096);
            hipLaunchKernelGGL(gen_mode_solve_kernel<T>, dim3(gb), dim3(64), 0, st, cur, out, F, ldf, dense ? 1 : 0, (int)A, s_, (int)B);
        } else {
            const unsigned gb = (unsigned)std::min<int64_t>((numel + 255) / 256, 4096);
            hipLaunchKernelGGL(gen_mode_apply_kernel<T>, dim3(gb), dim3(256), 0, st, cur, out, F, ldf, dense ? 1 : 0, (int)A, s_, (int)B,
                               last ? sumsq : (float*)nullptr, (what == 1 || P->p_mode()) ? 1 : 0);
        }
        cur = out;
        A *= s_;
    }
    return cur;
}

// arena layout (offsets in the descriptors); depends on the geometry, so psgdk_plan_set_geometry redoes it
static void layout_arenas(psgdk_plan* P) {
    const size_t esz = P->esz;
    const int n_tensors = P->n_tensors;
    // ---- state arena: [L fp32 per factor][diag vectors][Q, Qt][ema] ----
    size_t so = 0;
    for (auto& G : P->dd) { G.L_off = so; so += 4; }
    for (auto& F : P->dn) { F.L_off = so; so += 4; }
    so = align256(so);
    for (auto& G : P->dd) { G.a_off = so; so += align256((size_t)round_up64(G.len) * esz); }
    for (auto& F : P->dn) {
        const size_t mb = align256((size_t)F.dp * F.dp * esz);
        F.q_off = so; so += mb; F.qt_off = so; so += mb;
    }
    for (auto& D : P->td) {
        D.ema_off = so;
        if (P->use_momentum) so += align256((size_t)D.Rp * D.Cp * esz);
    }
    P->state_bytes = align256(so);
    // ---- work arena ----
    size_t wo = 0;
    P->zero_off = wo;
    for (auto& F : P->dn) { F.sc_off = wo; wo += 64; }
    wo = align256(wo);
    for (auto& F : P->dn) { F.vsq_off = wo; wo += 2 * 4 * 64 * 4; }
    wo = align256(wo);
    for (auto& F : P->dn) {
        F.rowss_off = wo; wo += align256((size_t)F.dp * 4);
        F.rowss_skh_off = F.rowss_off;
        if (P->chain_t()) { F.rowss_skh_off = wo; wo += align256((size_t)F.dp * 4); }
    }
    for (auto& G : P->dd) { G.sum_off = wo; wo += align256((size_t)round_up64(G.len) * 4); }
    if (P->geometry == PSGDK_GEOM_EQ)
        for (auto& G : P->dd) { G.sum2_off = wo; wo += align256((size_t)round_up64(G.len) * 4); }
    P->balnorm_off = wo; wo += align256((size_t)n_tensors * 2 * 4);
    P->zero_bytes = wo - P->zero_off;
    P->hsumsq_off = wo; wo += align256((size_t)n_tensors * 4);
    P->diag_mu_off = wo; wo += align256((P->dd.size() + 1) * 4);
    for (auto& G : P->dd) P->max_diag_len = std::max(P->max_diag_len, G.len);
    for (auto& D : P->td) {
        const size_t mb = align256((size_t)D.Rp * D.Cp * esz);
        D.gc_off = wo; wo += mb; D.x_off = wo; wo += mb; D.h_off = wo; wo += mb;
        if (D.kind == TK_M1 || D.kind == TK_M2) { D.pgt_off = wo; wo += mb; }
        if (D.kind == TK_M2) { D.pg_off = wo; wo += mb; D.tt_off = wo; wo += mb; }
        if (P->geometry == PSGDK_GEOM_EQ) {
            D.v_off = wo; wo += mb;
            if (D.kind == TK_M1 || D.kind == TK_M2) { D.bt_off = wo; wo += mb; }
            if (D.kind == TK_M2) { D.bb_off = wo; wo += mb; }
        }
    }
    for (auto& F : P->dn) {
        const size_t mb = align256((size_t)F.dp * F.dp * esz);
        size_t* offs[] = {&F.p_off, &F.t1_off, &F.qn_off, &F.qtn_off, &F.r_off, &F.rq_off, &F.rqt_off};
        for (size_t* o : offs) { *o = wo; wo += mb; }
        F.va_off = wo; wo += align256((size_t)64 * F.dp * esz);
        F.vb_off = wo; wo += align256((size_t)64 * F.dp * esz);
        // split-K for the mode Gram when the contracted extent is long (keeps >= ~256 workgroups on a lone big tensor)
        const TensorDesc& D = P->td[F.tensor];
        const int K = F.is_row ? D.Cp : D.Rp;
        F.slab_off = 0; F.gpart_off = 0;
        if (D.kind == TK_GEN) { F.gpart_off = wo; wo += align256((size_t)PSGDK_GEN_GPART * 4); }
        if (P->geometry == PSGDK_GEOM_EQ) { F.uinv_off = wo; wo += align256((size_t)F.dp * 64 * 4); }
        if (D.kind != TK_GEN && (K > 4096 || P->shard_of(F.tensor) >= 0)) {      // (a row shard's Gram always leaves as fp32 partials)
            const int nks = (K + 3071) / 3072;
            F.slab_off = wo; wo += align256((size_t)nks * F.dp * F.dp * 4);
        }
    }
    for (auto& g : P->gd) {
        const size_t nb = align256((size_t)P->td[g.tensor].numel * esz);
        g.pp_off[0] = wo; wo += nb; g.pp_off[1] = wo; wo += nb;
        if (P->geometry == PSGDK_GEOM_EQ) { g.pp2_off[0] = wo; wo += nb; g.pp2_off[1] = wo; wo += nb; }
    }
    P->work_bytes = align256(wo);
}

bool is_dense_dim(int64_t size, int64_t numel, double max_size, double max_skew) {
    // psgd.py:208  -- diagonal iff size <= 1 or size > max_size or size**2 > max_skew * numel
    return !(size <= 1 || (double)size > max_size || (double)size * (double)size > max_skew * (double)numel);
}

}  // namespace

extern "C" {

int psgdk_version(void) { return PSGDK_VERSION; }

const char* psgdk_strerror(int status) {
    switch (status) {
        case PSGDK_OK: return "ok";
        case PSGDK_ERR_INVALID: return "invalid argument";
        case PSGDK_ERR_UNSUPPORTED: return "unsupported (not built yet)";
        case PSGDK_ERR_HIP: return "HIP runtime error (see psgdk_last_hip_error)";
        case PSGDK_ERR_STATE: return "call order violated";
        case PSGDK_ERR_NLB_TIMEOUT: return "a cooperative norm-bound launch timed out waiting for a sibling workgroup; the affected "
                                           "factors skipped that update, the plan now uses the multi-launch route -- repeat the call";
        default: return "unknown status";
    }
}

int psgdk_last_hip_error(void) { return g_last_hip_error; }

int psgdk_plan_create(psgdk_plan** out, int n_tensors, const int32_t* ndim, const int64_t* dims, double max_size,
                      double max_skew, int precond_dtype, int use_momentum) {
    if (!out || n_tensors <= 0 || !ndim || (precond_dtype != PSGDK_BF16 && precond_dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (!(max_size >= 0.0) || !(max_skew >= 0.0)) return PSGDK_ERR_INVALID;
    std::unique_ptr<psgdk_plan> P(new psgdk_plan());
    P->n_tensors = n_tensors; P->dtype = precond_dtype; P->use_momentum = use_momentum ? 1 : 0;
    P->esz = precond_dtype == PSGDK_BF16 ? 2 : 4;
    P->max_size = max_size; P->max_skew = max_skew;
    { const char* e = getenv("PSGDK_NLB_FUSED"); P->nlb_unfused = e && e[0] == '0'; }
    size_t dpos = 0;
    // ---- structure (init_kron's dense/diag rule) ----
    for (int t = 0; t < n_tensors; ++t) {
        const int nd = ndim[t];
        if (nd < 0 || nd > 26) return PSGDK_ERR_INVALID;           // psgd.py:197-198
        if (nd > 0 && !dims) return PSGDK_ERR_INVALID;
        int64_t numel = 1;
        for (int i = 0; i < nd; ++i) { if (dims[dpos + i] <= 0) return PSGDK_ERR_INVALID; numel *= dims[dpos + i]; }
        if (nd > PSGDK_GEN_MAXDIM) return PSGDK_ERR_UNSUPPORTED;
        TensorDesc D{};
        D.numel = numel; D.row_diag = D.col_diag = D.row_dense = D.col_dense = -1;
        std::vector<FactorRef> fr;
        if (nd > 2) {
            // N-D: contiguous logical array folded into rows of PSGDK_GEN_FOLD; factors applied mode by mode
            D.kind = TK_GEN; D.transposed = 0;
            D.lcols = D.C = D.Cp = PSGDK_GEN_FOLD;
            D.lrows = D.R = D.Rp = (int)((numel + PSGDK_GEN_FOLD - 1) / PSGDK_GEN_FOLD);
            GenDesc g{};
            g.tensor = t; g.ndim = nd;
            for (int i = 0; i < nd; ++i) {
                g.dims[i] = (int)dims[dpos + i];
                const bool dense = is_dense_dim(dims[dpos + i], numel, max_size, max_skew);
                g.fkind[i] = dense ? PSGDK_DENSE : PSGDK_DIAG;
                fr.push_back(FactorRef{dense ? PSGDK_DENSE : PSGDK_DIAG, -1});
            }
            P->gd.push_back(g);
        } else if (nd <= 1) {
            const int64_t n = nd == 0 ? 1 : dims[dpos];
            // a 1-D tensor of size n: dense iff n^2 <= max_skew * n (psgd.py:208) -- only possible for tiny n / huge skew
            const bool dense = nd == 1 && is_dense_dim(n, numel, max_size, max_skew);
            D.lrows = 1; D.lcols = (int)n; D.transposed = 0; D.R = 1; D.C = (int)n;
            if (dense) {
                // treat as an R=1 row with a dense column factor: reuse the matrix path (TK_M1 without row factor)
                D.kind = TK_M1; D.Rp = 64; D.Cp = (int)round_up64(n);
            } else {
                D.kind = TK_VEC; D.Rp = 1; D.Cp = (int)round_up64(n);
            }
            fr.push_back(FactorRef{dense ? PSGDK_DENSE : (nd == 0 ? PSGDK_SCALAR : PSGDK_DIAG), -1});
        } else {
            const int64_t m = dims[dpos], n = dims[dpos + 1];
            const bool d0 = is_dense_dim(m, numel, max_size, max_skew), d1 = is_dense_dim(n, numel, max_size, max_skew);
            D.lrows = (int)m; D.lcols = (int)n;
            D.transposed = (d0 && !d1) ? 1 : 0;
            D.R = D.transposed ? (int)n : (int)m; D.C = D.transposed ? (int)m : (int)n;
            D.Rp = (int)round_up64(D.R); D.Cp = (int)round_up64(D.C);
            D.kind = (d0 && d1) ? TK_M2 : ((d0 || d1) ? TK_M1 : TK_DD);
            fr.push_back(FactorRef{d0 ? PSGDK_DENSE : PSGDK_DIAG, -1});
            fr.push_back(FactorRef{d1 ? PSGDK_DENSE : PSGDK_DIAG, -1});
        }
        D.stream_id = (unsigned)t;
        D.numel_clip = numel; D.row0 = 0;
        D.wide = (!D.transposed && D.C >= 256 && D.R >= 4) ? 1 : 0;
        P->td.push_back(D);
        P->factors.push_back(fr);
        P->order.push_back(nd == 0 ? 1 : nd);
        dpos += nd;
    }
    // ---- factor lists ----
    for (int t = 0; t < n_tensors; ++t) {
        TensorDesc& D = P->td[t];
        auto& fr = P->factors[t];
        for (size_t i = 0; i < fr.size(); ++i) {
            // which canonical side does logical dim i sit on?
            bool is_row;
            const bool gen = D.kind == TK_GEN;
            if (fr.size() == 1 || gen) is_row = false;           // vectors / scalars: the column side
            else is_row = D.transposed ? (i == 1) : (i == 0);
            int len = is_row ? D.R : D.C;
            GenDesc* G_ = nullptr;
            if (gen) {
                for (auto& g : P->gd) if (g.tensor == t) G_ = &g;
                len = G_->dims[i];
            }
            if (fr[i].kind == PSGDK_DENSE) {
                DenseDesc F{};
                F.tensor = t; F.d = len; F.dp = (int)round_up64(len); F.is_row = is_row ? 1 : 0;
                F.c = (float)((double)D.numel / (double)len);
                fr[i].idx = (int)P->dn.size();
                if (gen) G_->fidx[i] = fr[i].idx; else (is_row ? D.row_dense : D.col_dense) = fr[i].idx;
                F.stream_id = 0x40000000u + ((unsigned)t * PSGDK_GEN_MAXDIM + (unsigned)i) * 2u;
                P->dn.push_back(F);
                P->dense_dim.push_back((int)i);
                P->max_dp = std::max(P->max_dp, F.dp);
            } else {
                DiagDesc G{};
                G.ext_max_off = -1;
                G.tensor = t; G.len = len; G.is_row = is_row ? 1 : 0;
                G.c = (float)((double)D.numel / (double)len);
                fr[i].idx = (int)P->dd.size();
                if (gen) G_->fidx[i] = fr[i].idx; else (is_row ? D.row_diag : D.col_diag) = fr[i].idx;
                P->dd.push_back(G);
            }
        }
    }
    layout_arenas(P.get());
    *out = P.release();
    return PSGDK_OK;
}

int psgdk_plan_set_stream_ids(psgdk_plan* plan, const uint32_t* ids) {
    if (!plan || !ids) return PSGDK_ERR_INVALID;
    if (plan->state) return PSGDK_ERR_STATE;      // descriptors are uploaded at bind time
    for (int t = 0; t < plan->n_tensors; ++t) {
        if (ids[t] >= 0x04000000u) return PSGDK_ERR_INVALID;
        plan->td[t].stream_id = ids[t];
    }
    for (size_t f = 0; f < plan->dn.size(); ++f)
        plan->dn[f].stream_id = 0x40000000u + (ids[plan->dn[f].tensor] * PSGDK_GEN_MAXDIM + (unsigned)plan->dense_dim[f]) * 2u;
    return PSGDK_OK;
}

int psgdk_plan_set_geometry(psgdk_plan* plan, int geometry) {
    if (!plan || geometry < PSGDK_GEOM_Q0P5EQ1P5 || geometry > PSGDK_GEOM_PRO4P) return PSGDK_ERR_INVALID;
    if (plan->state) return PSGDK_ERR_STATE;
    if (!plan->shards.empty() && geometry != PSGDK_GEOM_Q0P5EQ1P5 && geometry != PSGDK_GEOM_QEQ && geometry != PSGDK_GEOM_QUAD) return PSGDK_ERR_UNSUPPORTED;
    plan->geometry = geometry;
    layout_arenas(plan);
    return PSGDK_OK;
}

int psgdk_plan_set_row_shard(psgdk_plan* plan, int t, int64_t global_rows, int64_t row0, int member, int members) {
    if (!plan || t < 0 || t >= plan->n_tensors || members < 2 || member < 0 || member >= members) return PSGDK_ERR_INVALID;
    if (plan->state) return PSGDK_ERR_STATE;
    // the three geometries that share Pg, the mode Grams and the norm-bound step size (update_whiten_family): for all of them the rows of a
    // [diagonal, dense] matrix are independent given the dense factor, whose mode Gram is the only sum over the row blocks
    if (plan->geometry != PSGDK_GEOM_Q0P5EQ1P5 && plan->geometry != PSGDK_GEOM_QEQ && plan->geometry != PSGDK_GEOM_QUAD) return PSGDK_ERR_UNSUPPORTED;
    TensorDesc& D = plan->td[t];
    // a row block of a matrix with a diagonal factor on dim 0 and a dense one on dim 1, held as it is (rows contiguous in the caller's
    // tensor): the structure init_kron gives the WHOLE tensor must also be what the block got from its own shape (the host checks that
    // before it splits a tensor; here it is verified)
    if (D.kind != TK_M1 || D.transposed || D.row_diag < 0 || D.col_dense < 0) return PSGDK_ERR_INVALID;
    if (row0 < 0 || global_rows <= 0 || row0 + D.lrows > global_rows || plan->shard_of(t) >= 0) return PSGDK_ERR_INVALID;
    if (!plan->shards.empty() && (plan->shard_member != member || plan->shard_members != members)) return PSGDK_ERR_INVALID;
    if (!is_dense_dim(D.lcols, global_rows * D.lcols, plan->max_size, plan->max_skew) ||
        is_dense_dim(global_rows, global_rows * D.lcols, plan->max_size, plan->max_skew)) return PSGDK_ERR_INVALID;
    plan->shard_member = member; plan->shard_members = members;
    const double numel_g = (double)global_rows * (double)D.lcols;
    D.numel_clip = (long long)(global_rows * (int64_t)D.lcols);
    D.row0 = row0;
    plan->dd[D.row_diag].c = (float)(numel_g / (double)global_rows);
    plan->dn[D.col_dense].c = (float)(numel_g / (double)D.lcols);
    size_t ro = 0;
    for (auto& sh : plan->shards) ro = sh.rec_off + align256((size_t)plan->dn[plan->td[sh.tensor].col_dense].dp * plan->dn[plan->td[sh.tensor].col_dense].dp * 4 + 256);
    plan->shards.push_back(psgdk_plan::RowShard{t, global_rows, row0, ro});
    const int dp = plan->dn[D.col_dense].dp;
    plan->dd[D.row_diag].ext_max_off = (long long)(ro + (size_t)dp * dp * 4);
    plan->xchg_record_bytes = ro + align256((size_t)dp * dp * 4 + 256);
    layout_arenas(plan);
    return PSGDK_OK;
}

int psgdk_plan_exchange_bytes(const psgdk_plan* plan, size_t* record_bytes) {
    if (!plan || !record_bytes) return PSGDK_ERR_INVALID;
    *record_bytes = plan->xchg_record_bytes;
    return PSGDK_OK;
}

int psgdk_plan_destroy(psgdk_plan* plan) {
    delete plan;
    return PSGDK_OK;
}

int psgdk_plan_arena_bytes(const psgdk_plan* plan, size_t* state_bytes, size_t* work_bytes) {
    if (!plan || !state_bytes || !work_bytes) return PSGDK_ERR_INVALID;
    *state_bytes = plan->state_bytes; *work_bytes = plan->work_bytes;
    return PSGDK_OK;
}

int psgdk_plan_num_factors(const psgdk_plan* plan, int t, int* n_factors) {
    if (!plan || t < 0 || t >= plan->n_tensors || !n_factors) return PSGDK_ERR_INVALID;
    *n_factors = (int)plan->factors[t].size();
    return PSGDK_OK;
}

int psgdk_plan_factor_view(const psgdk_plan* plan, int t, int i, int* kind, size_t* q_offset, int64_t* d, int64_t* ld,
                           size_t* lipschitz_offset) {
    if (!plan || t < 0 || t >= plan->n_tensors || i < 0 || i >= (int)plan->factors[t].size()) return PSGDK_ERR_INVALID;
    const FactorRef& fr = plan->factors[t][i];
    if (kind) *kind = fr.kind;
    if (fr.kind == PSGDK_DENSE) {
        const DenseDesc& F = plan->dn[fr.idx];
        if (q_offset) *q_offset = F.q_off;
        if (d) *d = F.d;
        if (ld) *ld = F.dp;
        if (lipschitz_offset) *lipschitz_offset = F.L_off;
    } else {
        const DiagDesc& G = plan->dd[fr.idx];
        if (q_offset) *q_offset = G.a_off;
        if (d) *d = G.len;
        if (ld) *ld = G.len;
        if (lipschitz_offset) *lipschitz_offset = G.L_off;
    }
    return PSGDK_OK;
}

int psgdk_plan_ema_view(const psgdk_plan* plan, int t, size_t* offset, int64_t* rows, int64_t* cols, int64_t* ld,
                        int* transposed) {
    if (!plan || t < 0 || t >= plan->n_tensors) return PSGDK_ERR_INVALID;
    if (!plan->use_momentum) return PSGDK_ERR_STATE;
    const TensorDesc& D = plan->td[t];
    if (offset) *offset = D.ema_off;
    if (rows) *rows = D.lrows;
    if (cols) *cols = D.lcols;
    if (ld) *ld = D.Cp;
    if (transposed) *transposed = D.transposed;
    return PSGDK_OK;
}

static int nlb_plan_coop(psgdk_plan* P);

    // tiling per stage: the 256 x 256 kernel pays off once a launch holds at least three full rounds of its tiles (one
    // workgroup per CU; 560 tiles = 2.19 rounds cost three, and the 128 x 128 kernel's finer tiles then fill the chip
    // better: -2.4 % of the GPT-2-small step with the threshold at 768 instead of 512).  The subspace iteration
    // (M = 64) and the EQ Grams / update stay on the small tiling; EQ's A = (kron Q) Hvp is a full-size product like upd_a.
    // Both tilings accumulate K in the same order: same bits.
static void decide_tiling(psgdk_plan* P, Stage* s) {
    {
        int64_t nb = 0, nb_f2 = 0;
        for (const GemmProblem& g : s->probs) {
            const int64_t tm = (g.M + GEMM_BIG_BM - 1) / GEMM_BIG_BM, tn = (g.N + GEMM_BIG_BN - 1) / GEMM_BIG_BN;
            const int64_t nks = (g.flags & GF_SPLITK) ? (g.K + g.kchunk - 1) / g.kchunk : 1;
            const int64_t nt = ((g.flags & GF_SYM) ? tm * (tm + 1) / 2 : tm * tn) * nks;
            nb += nt;
            if (P->dtype == PSGDK_BF16 && gemm_is_fused2_problem<bf16_t>(g)) nb_f2 += nt;
        }
        s->big = nb >= big_min_tiles();
        // two-output problems with a fused update (Q', R Q, the symmetric Grams): only the 128 x 128 kernel has the register-
        // resident epilogue for them (in the 256 x 256 one it spills); it wins there although its main loop is slower --
        // GPT-2-medium (123 x 1024^3): Q' 446 -> 406 us, R Q 570 -> 429, mode Grams 531 -> 482 (profiles/r02_experiments).
        // (PSGDK_BIG_MIN_TILES set: the tests want the big tiling wherever it can run)
        if (s->big && s != &P->g_P && !getenv("PSGDK_BIG_MIN_TILES") && 2 * nb_f2 >= nb) s->big = false;     // (P = Q^T Q: 183 vs 197)
        // small launches: few 128 x 128 tiles in the whole launch -> 64 x 64 tiles, K split over the waves
        s->ksplit = false;
        if (!s->big && !s->ext && !s->probs.empty()) {
            const int bk = P->dtype == PSGDK_BF16 ? 64 : 32;
            int64_t n128 = 0;
            bool ok = true;
            for (const GemmProblem& g : s->probs) {
                const int64_t tm = (g.M + GEMM_BM - 1) / GEMM_BM, tn = (g.N + GEMM_BN - 1) / GEMM_BN;
                const int64_t nks = (g.flags & GF_SPLITK) ? (g.K + g.kchunk - 1) / g.kchunk : 1;
                n128 += ((g.flags & GF_SYM) ? tm * (tm + 1) / 2 : tm * tn) * nks;
                ok = ok && g.M % 64 == 0 && g.N % 64 == 0 && g.K % bk == 0 && g.K >= bk &&
                     (!(g.flags & GF_SPLITK) || g.kchunk % bk == 0);
            }
            s->ksplit = ok && n128 <= kKsplitMaxTiles;
        }
    }
}

int psgdk_plan_bind(psgdk_plan* plan, void* state_arena, void* work_arena) {
    if (!plan || !state_arena || !work_arena) return PSGDK_ERR_INVALID;
    if (((uintptr_t)state_arena & 255) || ((uintptr_t)work_arena & 255)) return PSGDK_ERR_INVALID;
    psgdk_plan* P = plan;
    P->state = (unsigned char*)state_arena; P->work = (unsigned char*)work_arena;
    P->p_valid = false;
    unsigned char* S = P->state; unsigned char* W = P->work;
    int rc;
    if ((rc = upload(&P->d_td, P->td))) return rc;
    if ((rc = upload(&P->d_dd, P->dd))) return rc;
    if ((rc = upload(&P->d_dn, P->dn))) return rc;
    if ((rc = nlb_plan_coop(P))) return rc;
    if ((rc = upload(&P->d_gd, P->gd))) return rc;
    // ---- elementwise tile tables ----
    std::vector<EwTile> all, diag;
    P->tile_begin.assign(P->n_tensors + 1, 0);
    for (int t = 0; t < P->n_tensors; ++t) {
        const TensorDesc& D = P->td[t];
        P->tile_begin[t] = (unsigned)all.size();
        const int th = D.wide ? 16 : 64, tw = D.wide ? 256 : 64;
        for (int tr = 0; tr < (D.R + th - 1) / th; ++tr)
            for (int tc = 0; tc < (D.C + tw - 1) / tw; ++tc) all.push_back(EwTile{t, tr, tc});
        if (D.kind == TK_VEC || D.kind == TK_DD)
            for (int tr = 0; tr < (D.R + 63) / 64; ++tr)
                for (int tc = 0; tc < (D.C + 63) / 64; ++tc) diag.push_back(EwTile{t, tr, tc});
    }
    P->tile_begin[P->n_tensors] = (unsigned)all.size();
    P->n_tiles_all = (unsigned)all.size(); P->n_tiles_diag = (unsigned)diag.size();
    if ((rc = upload(&P->d_tiles_all, all))) return rc;
    if ((rc = upload(&P->d_tiles_diag, diag))) return rc;
    auto alloc_ptrs = [&](void*** p, size_t n) -> int {
        if (*p) { (void)hipFree(*p); *p = nullptr; }
        HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(void*)));
        return PSGDK_OK;
    };
    if ((rc = alloc_ptrs(&P->d_noise_g, P->n_tensors))) return rc;
    if ((rc = alloc_ptrs(&P->d_noise_spd, P->dn.size()))) return rc;
    if ((rc = alloc_ptrs(&P->d_noise_skh, P->dn.size()))) return rc;
    if (P->d_balance) { (void)hipFree(P->d_balance); P->d_balance = nullptr; }
    HIPCHK(hipMalloc((void**)&P->d_balance, P->n_tensors * sizeof(int)));
    if (P->d_scale_diag) { (void)hipFree(P->d_scale_diag); P->d_scale_diag = nullptr; }
    if (P->d_scale_dense) { (void)hipFree(P->d_scale_dense); P->d_scale_dense = nullptr; }
    HIPCHK(hipMalloc((void**)&P->d_scale_diag, std::max<size_t>(P->dd.size(), 1) * 4));
    HIPCHK(hipMalloc((void**)&P->d_scale_dense, std::max<size_t>(P->dn.size(), 1) * 4));

    // ---- grouped GEMM stages (absolute pointers, so built at bind time) ----
    for (Stage* s : P->all_stages()) s->probs.clear();
    P->app_a_tensor.clear(); P->app_b_tensor.clear();
    P->split_dense.clear();
    P->gram_prob.clear();
    float* hsumsq = (float*)(W + P->hsumsq_off);
    for (size_t f = 0; f < P->dn.size(); ++f) {
        const DenseDesc& F = P->dn[f];
        const TensorDesc& D = P->td[F.tensor];
        float* sc = (float*)(W + F.sc_off);
        GemmProblem g{};
        // P = Q^T Q = Qt Qt^T
        g.A = S + F.qt_off; g.B = S + F.qt_off; g.C = W + F.p_off; g.Ct = g.C;
        g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = F.dp; g.alpha = 1.f; g.flags = GF_SYM;
        // triangular geometry: Q stays exactly upper triangular (psgd.py:316 applies triu), so Qt[m][k] = 0 for k > m
        if (P->geometry == PSGDK_GEOM_EQ) g.flags |= GF_KBAND_A_LO | GF_KBAND_B_LO;
        P->g_P.probs.push_back(g);
        // mode Gram term1 (psgd.py:405): col factor: Pgt Pgt^T; row factor: Pg Pg^T  (N-D tensors: kernels_gen.hiph)
        P->gram_prob.push_back(D.kind == TK_GEN ? -1 : (int)P->g_gram.probs.size());
        g = GemmProblem{};
        const unsigned char* Z = W + (F.is_row ? D.pg_off : D.pgt_off);
        const int K = F.is_row ? D.Cp : D.Rp;
        g.A = Z; g.B = Z; g.C = W + F.t1_off; g.Ct = g.C;
        g.M = g.N = F.dp; g.K = K; g.lda = g.ldb = K; g.ldc = g.ldct = F.dp; g.alpha = 1.f; g.flags = GF_SYM;
        // row sums of squares and max diagonal of term1 (psgd.py:59,61) straight from the epilogue
        g.row_sumsq = (float*)(W + F.rowss_off); g.diag_max = sc + DS_NF;
        if (F.slab_off) {
            g.flags |= GF_SPLITK; g.kchunk = 3072; g.slab = (float*)(W + F.slab_off);
            if (D.kind != TK_GEN) P->split_dense.push_back((int)f);
        }
        if (D.kind != TK_GEN) P->g_gram.probs.push_back(g);
        // subspace iteration of norm_lower_bound_spd (A = term1) and _skh (A = R): V <- V (A/nf), psgd.py:65-67
        for (int chain = 0; chain < 2; ++chain)
            for (int p = 0; p < 4; ++p) {
                g = GemmProblem{};
                float* vsq = (float*)(W + F.vsq_off) + (chain * 4 + p) * 64;
                g.A = W + ((p & 1) ? F.vb_off : F.va_off); g.B = W + (chain ? F.r_off : F.t1_off);
                g.C = W + ((p & 1) ? F.va_off : F.vb_off);
                g.M = 64; g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = F.dp; g.alpha = 1.f;
                g.alpha_dev = sc + (chain ? DS_INVNF_SKH : DS_INVNF_SPD);
                g.row_sumsq = vsq;
                if (p & 1) { g.row_scale = vsq - 64; g.flags = GF_RSQRT_ROWSCALE; }
                P->g_nlb[chain][p].probs.push_back(g);
            }
        if (P->chain_t()) {
            // The Q0.5EQ1.5 update + procrustes_step2 in TRANSPOSED space (round 3): with T = term1 symmetric and R antisymmetric, every
            // product of psgd.py:415 and :117-124 has an NT form that needs only the transposes Z = Q'^T, RQ^T, and R itself:
            //   Z     = Qt - mu (Qt T - c Qt)         NT(A = Qt,   B = T)        = (T Q)^T                   [one output]
            //   R     = Z - Z^T                        rsub_t_kernel (reads Z only)
            //   RQ^T  = s NT(A = Z,    B = R)          (R Q')[n][m] = sum_k R[n][k] Z[m][k]                  [one output]
            //   Qt   <- Z + a (RQ^T + a/2 RRQ^T),      RRQ^T = s NT(A = RQ^T, B = R): (R RQ)[n][m] = sum_k R[n][k] RQ^T[m][k]
            //           stored as Qt (primary) and Q (transposed copy), straight into the state.
            // Each accumulator sums the same products in the same K order as the Q-space form (operands swapped), so every output is
            // bit-identical to it; the stages read and write 3 / 3 / 4.5 of a factor's matrices where they moved 5 / 5 / 6 (Q', RQ
            // and the second copy of every intermediate are gone: 1.39 -> 0.99 GB per GPT-2-small step over the chain).
            g = GemmProblem{};
            g.A = S + F.qt_off; g.B = W + F.t1_off; g.C = W + F.qtn_off; g.Ct = nullptr;
            g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = g.ldq = F.dp; g.alpha = 1.f;
            g.flags = GF_QUPD; g.Qold = S + F.qt_off; g.mu_dev = sc + DS_MU; g.c = F.c;
            P->g_qupd.probs.push_back(g);
            g = GemmProblem{};
            g.A = W + F.qtn_off; g.B = W + F.r_off; g.C = W + F.rqt_off; g.Ct = nullptr;
            g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = F.dp; g.alpha = 1.f; g.alpha_dev = sc + DS_S; g.trace = sc + DS_TR1;
            // tr(R RQ) = <R, RQ^T>: available from THIS product's epilogue, before R RQ is formed
            g.dot_with = W + F.r_off; g.lddot = F.dp; g.dot_out = sc + DS_TR2; g.flags = GF_DOT_POS;
            P->g_rq.probs.push_back(g);
            g.dot_with = nullptr; g.dot_out = nullptr; g.trace = nullptr;
            g.A = W + F.rqt_off; g.B = W + F.r_off; g.C = S + F.qt_off; g.Ct = S + F.q_off;
            g.flags = GF_PROCR; g.X1 = W + F.qtn_off; g.X2 = W + F.rqt_off; g.ldq = F.dp; g.tr1_dev = sc + DS_TR1; g.tr2_dev = sc + DS_TR2;
            P->g_rrq.probs.push_back(g);
            continue;
        }
        // Q' = Q - mu (term1 Q - c Q)   (psgd.py:415)
        g = GemmProblem{};
        g.A = W + F.t1_off; g.B = S + F.qt_off; g.C = W + F.qn_off; g.Ct = W + F.qtn_off;
        g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = g.ldq = F.dp; g.alpha = 1.f;
        g.flags = GF_QUPD; g.Qold = S + F.q_off; g.mu_dev = sc + DS_MU; g.c = F.c;
        P->g_qupd.probs.push_back(g);
        // RQ = s R Q', RRQ = s R RQ  (psgd.py:118-120), traces for the line search (psgd.py:121-122)
        g = GemmProblem{};
        g.A = W + F.r_off; g.B = W + F.qtn_off; g.C = W + F.rq_off; g.Ct = W + F.rqt_off;
        g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = F.dp; g.alpha = 1.f; g.alpha_dev = sc + DS_S; g.trace = sc + DS_TR1;
        // tr(R RQ) = -<R, RQ> (R antisymmetric): available from THIS product's epilogue, before R RQ is formed
        g.dot_with = W + F.r_off; g.lddot = F.dp; g.dot_out = sc + DS_TR2;
        P->g_rq.probs.push_back(g);
        // Q = Q' + a (RQ + a/2 R RQ), straight into the state (Q and Qt); a from the device traces
        g.dot_with = nullptr; g.dot_out = nullptr; g.trace = nullptr;
        g.B = W + F.rqt_off; g.C = S + F.q_off; g.Ct = S + F.qt_off;
        g.flags = GF_PROCR; g.X1 = W + F.qn_off; g.X2 = W + F.rq_off; g.ldq = F.dp; g.tr1_dev = sc + DS_TR1; g.tr2_dev = sc + DS_TR2;
        P->g_rrq.probs.push_back(g);
    }
    for (int t = 0; t < P->n_tensors; ++t) {
        const TensorDesc& D = P->td[t];
        if (D.kind != TK_M1 && D.kind != TK_M2) continue;
        const DenseDesc& Fc = P->dn[D.col_dense];
        const void* rs = D.row_diag >= 0 ? (const void*)(S + P->dd[D.row_diag].a_off) : nullptr;
        float* rsum = D.row_diag >= 0 ? (float*)(W + P->dd[D.row_diag].sum_off) : nullptr;
        // first product: S * P_col (S = X for the update, ema / grad for the apply)
        GemmProblem g{};
        g.B = W + Fc.p_off; g.M = D.Rp; g.N = D.Cp; g.K = D.Cp; g.lda = D.Cp; g.ldb = Fc.dp; g.alpha = 1.f;
        if (D.kind == TK_M1) {
            g.row_scale = rs; g.flags = (rs && !P->p_mode()) ? GF_SQ_ROWSCALE : 0;
            GemmProblem u = g; u.A = W + D.x_off; u.Ct = W + D.pgt_off; u.ldct = D.Rp; u.row_sumsq = rsum; u.flags |= GF_TMAJOR;
            P->g_upd_a.probs.push_back(u);
            for (int src = 0; src < 2; ++src) {
                GemmProblem a = g; a.A = src == PSGDK_SRC_GRAD ? (const void*)(W + D.gc_off) : (const void*)(S + D.ema_off);
                a.C = W + D.h_off; a.ldc = D.Cp; a.sumsq = hsumsq + t;
                P->g_app_a[src].probs.push_back(a);
            }
            P->app_a_tensor.push_back(t);
        } else {
            const DenseDesc& Fr = P->dn[D.row_dense];
            GemmProblem u = g; u.A = W + D.x_off; u.Ct = W + D.tt_off; u.ldct = D.Rp; u.flags |= GF_TMAJOR;
            P->g_upd_a.probs.push_back(u);
            for (int src = 0; src < 2; ++src) {
                GemmProblem a = g; a.A = src == PSGDK_SRC_GRAD ? (const void*)(W + D.gc_off) : (const void*)(S + D.ema_off);
                a.Ct = W + D.tt_off; a.ldct = D.Rp; a.flags |= GF_TMAJOR;
                P->g_app_a[src].probs.push_back(a);
            }
            P->app_a_tensor.push_back(t);
            // second product: P_row * T, via T^T as the K-contiguous B operand
            GemmProblem b{};
            b.A = W + Fr.p_off; b.B = W + D.tt_off; b.M = D.Rp; b.N = D.Cp; b.K = D.Rp; b.lda = Fr.dp; b.ldb = D.Rp; b.alpha = 1.f;
            GemmProblem ub = b; ub.C = W + D.pg_off; ub.ldc = D.Cp; ub.Ct = W + D.pgt_off; ub.ldct = D.Rp;
            P->g_upd_b.probs.push_back(ub);
            GemmProblem ab = b; ab.C = W + D.h_off; ab.ldc = D.Cp; ab.sumsq = hsumsq + t;
            P->g_app_b.probs.push_back(ab);
            P->app_b_tensor.push_back(t);
        }
    }
    if (P->geometry == PSGDK_GEOM_QEQ || P->geometry == PSGDK_GEOM_QUAD || P->geometry == PSGDK_GEOM_QUAD4P)
        for (size_t f = 0; f < P->dn.size(); ++f) {
            const DenseDesc& F = P->dn[f];
            float* sc = (float*)(W + F.sc_off);
            GemmProblem g{};
            g.B = W + F.t1_off; g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = g.ldq = F.dp; g.alpha = 1.f;
            g.flags = GF_QUPD; g.mu_dev = sc + DS_MU; g.c = F.c;
            if (P->geometry == PSGDK_GEOM_QEQ) {      // Q' = Q - mu (Q term1 - c Q): term1 is symmetric, so it is its own B^T
                g.A = S + F.q_off; g.Qold = S + F.q_off; g.C = W + F.qn_off; g.Ct = W + F.qtn_off;
                P->v_qeq.probs.push_back(g);
            } else {                                   // p2 = p1 - mu/2 (p1 term1 - c p1), p1 = the first half step in qn
                g.A = W + F.qn_off; g.Qold = W + F.qn_off; g.C = W + F.rq_off; g.Ct = W + F.rqt_off;
                P->v_quad2.probs.push_back(g);
            }
        }
    if (P->geometry == PSGDK_GEOM_PRO4P) {
        for (Stage* e : {&P->v_pro_rq, &P->v_pro_rrq, &P->v_pro_rrrq, &P->g_nlb[1][0], &P->g_nlb[1][1], &P->g_nlb[1][2], &P->g_nlb[1][3]}) e->ext = true;
        for (size_t f = 0; f < P->dn.size(); ++f) {
            // procrustes_step3 (psgd.py:127-158) on Q' (qn / qtn): RQ = s R Q', RRQ = s R RQ, then Q' += a (RQ + a/2 (RRQ + a/4 s R RRQ))
            // with tr(RRQ) = -s <R, RQ> and tr(RRRQ) = -s <R, RRQ> from the epilogues of the first two products (R antisymmetric)
            const DenseDesc& F = P->dn[f];
            float* sc = (float*)(W + F.sc_off);
            for (int p_ = 0; p_ < 4; ++p_) P->g_nlb[1][p_].probs[f].skip = sc + DS_DONE;
            GemmProblem g{};
            g.A = W + F.r_off; g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = g.ldq = g.lddot = F.dp; g.alpha = 1.f;
            g.alpha_dev = sc + DS_S; g.skip = sc + DS_DONE; g.dot_with = W + F.r_off;
            GemmProblem a = g; a.B = W + F.qtn_off; a.C = W + F.rq_off; a.Ct = W + F.rqt_off; a.trace = sc + DS_TR1; a.dot_out = sc + DS_TR2;
            P->v_pro_rq.probs.push_back(a);
            GemmProblem b = g; b.B = W + F.rqt_off; b.C = W + F.p_off; b.Ct = W + F.t1_off; b.dot_out = sc + DS_TR3;      // RRQ, RRQ^T
            P->v_pro_rrq.probs.push_back(b);
            GemmProblem c = g; c.dot_with = nullptr; c.B = W + F.t1_off; c.C = W + F.qn_off; c.Ct = W + F.qtn_off; c.flags = GF_PROCR3;
            c.X1 = W + F.qn_off; c.X2 = W + F.rq_off; c.X3 = W + F.p_off; c.tr1_dev = sc + DS_TR1; c.tr2_dev = sc + DS_TR2; c.tr3_dev = sc + DS_TR3;
            P->v_pro_rrrq.probs.push_back(c);
        }
    }
    if (P->geometry == PSGDK_GEOM_QEP)
        for (size_t f = 0; f < P->dn.size(); ++f) {
            // term1 = Gram_i(Q_i Pg) = Q T1 Q^T with T1 the mode Gram of Pg; term2 = c Q Q^T (psgd.py:353-361)
            const DenseDesc& F = P->dn[f];
            float* sc = (float*)(W + F.sc_off);
            GemmProblem g{};
            g.M = g.N = g.K = F.dp; g.lda = g.ldb = g.ldc = g.ldct = g.ldq = F.dp; g.alpha = 1.f;
            GemmProblem u = g; u.A = S + F.q_off; u.B = W + F.t1_off; u.C = W + F.r_off;               // U = Q T1 (T1 symmetric)
            P->v_qep_u.probs.push_back(u);
            GemmProblem t1 = g; t1.A = W + F.r_off; t1.B = S + F.q_off; t1.C = t1.Ct = W + F.t1_off; t1.flags = GF_SYM;   // U Q^T
            P->v_qep_t1.probs.push_back(t1);
            GemmProblem t2 = g; t2.A = t2.B = S + F.q_off; t2.C = t2.Ct = W + F.rq_off; t2.flags = GF_SYM; t2.alpha = F.c;
            P->v_qep_t2.probs.push_back(t2);
            GemmProblem q = g;                                                                          // Q' = Q - mu (term1 - term2) Q
            q.A = W + F.r_off; q.B = S + F.qt_off; q.C = W + F.qn_off; q.Ct = W + F.qtn_off;
            q.flags = GF_QUPD; q.Qold = S + F.q_off; q.mu_dev = sc + DS_MU; q.c = 0.f;
            P->e_qupd.probs.push_back(q);
        }
    if (P->geometry == PSGDK_GEOM_EQ) {
        // ---- triangular geometry (psgd.py:278-336) ----
        P->e_gram_prob.assign(P->dn.size(), -1);
        std::vector<TrsmJob> jobs[2]; std::vector<TrsmTile> ttiles[2]; std::vector<UinvJob> uj;
        for (size_t f = 0; f < P->dn.size(); ++f) {
            const DenseDesc& F = P->dn[f];
            const TensorDesc& D = P->td[F.tensor];
            float* sc = (float*)(W + F.sc_off);
            uj.push_back(UinvJob{S + F.q_off, (float*)(W + F.uinv_off), F.d, F.dp});
            // term1 = Gram_i(A) from A / A^T, term2 = Gram_i(B) from B / B^T (psgd.py:306-307)
            const int K = F.is_row ? D.Cp : D.Rp;
            GemmProblem g{};
            g.M = g.N = F.dp; g.K = K; g.lda = g.ldb = K; g.ldc = g.ldct = F.dp; g.alpha = 1.f; g.flags = GF_SYM;
            if (F.slab_off) { g.flags |= GF_SPLITK; g.kchunk = 3072; g.slab = (float*)(W + F.slab_off); }
            if (D.kind == TK_GEN) { P->e_gram_prob[f] = -1; goto eq_qupd; }      // N-D tensors: Grams by kernels_gen.hiph
            P->e_gram_prob[f] = (int)P->e_g1.probs.size();
            {
            GemmProblem g1 = g; g1.A = g1.B = W + (F.is_row ? D.pg_off : D.pgt_off); g1.C = g1.Ct = W + F.t1_off;
            GemmProblem g2 = g; g2.A = g2.B = W + (F.is_row ? D.bb_off : D.bt_off); g2.C = g2.Ct = W + F.rq_off;
            P->e_g1.probs.push_back(g1); P->e_g2.probs.push_back(g2);
            }
        eq_qupd:
            // Q' = Q - mu triu(term1 - term2) Q  (psgd.py:316)
            GemmProblem q{};
            q.A = W + F.r_off; q.B = S + F.qt_off; q.C = W + F.qn_off; q.Ct = W + F.qtn_off;
            q.M = q.N = q.K = F.dp; q.lda = q.ldb = q.ldc = q.ldct = q.ldq = F.dp; q.alpha = 1.f;
            // (both operands triangular: triu(.) is zero left of the diagonal, Q^T right of it)
            q.flags = GF_QUPD | GF_KBAND_A_UP | GF_KBAND_B_LO; q.Qold = S + F.q_off; q.mu_dev = sc + DS_MU; q.c = 0.f;
            P->e_qupd.probs.push_back(q);
        }
        for (int t = 0; t < P->n_tensors; ++t) {
            const TensorDesc& D = P->td[t];
            if (D.kind != TK_M1 && D.kind != TK_M2) continue;
            const DenseDesc& Fc = P->dn[D.col_dense];
            const DiagDesc* Gr = D.row_diag >= 0 ? &P->dd[D.row_diag] : nullptr;
            // A = (row factor) Hvp Qc^T: first the column side
            GemmProblem a{};
            a.A = W + D.x_off; a.B = S + Fc.q_off; a.M = D.Rp; a.N = D.Cp; a.K = D.Cp; a.lda = D.Cp; a.ldb = Fc.dp; a.alpha = 1.f;
            a.flags = GF_TMAJOR | GF_KBAND_B_UP; a.ldct = D.Rp;          // Q upper triangular: Q[n][k] = 0 for k < n
            TrsmJob j{};
            j.in = W + D.v_off; j.ld_in = D.Cp; j.U = S + Fc.q_off; j.Ut = S + Fc.qt_off; j.uinv = (const float*)(W + Fc.uinv_off);
            j.rows = D.R; j.dp = Fc.dp;
            if (D.kind == TK_M1) {
                a.Ct = W + D.pgt_off;
                if (Gr) { a.row_scale = S + Gr->a_off; a.row_sumsq = (float*)(W + Gr->sum_off); }
                j.row_div = Gr ? (const void*)(S + Gr->a_off) : nullptr;
                j.out_t = W + D.bt_off; j.ld_t = D.Rp;
                j.row_ss = Gr ? (float*)(W + Gr->sum2_off) : nullptr;
            } else {
                const DenseDesc& Fr = P->dn[D.row_dense];
                a.Ct = W + D.tt_off;
                GemmProblem b{};
                b.A = S + Fr.q_off; b.B = W + D.tt_off; b.M = D.Rp; b.N = D.Cp; b.K = D.Rp; b.lda = Fr.dp; b.ldb = D.Rp; b.alpha = 1.f;
                b.C = W + D.pg_off; b.ldc = D.Cp; b.Ct = W + D.pgt_off; b.ldct = D.Rp;
                b.flags = GF_KBAND_A_UP;
                P->e_a2.probs.push_back(b);
                // V Qc^{-1} -> (.)^T in tt (free again after the second product), then (.)^T Qr^{-1} = B^T, and B
                j.out_t = W + D.tt_off; j.ld_t = D.Rp;
                TrsmJob k{};
                k.in = W + D.tt_off; k.ld_in = D.Rp; k.U = S + Fr.q_off; k.Ut = S + Fr.qt_off; k.uinv = (const float*)(W + Fr.uinv_off);
                k.rows = D.C; k.dp = Fr.dp;
                k.out_nat = W + D.bt_off; k.ld_nat = D.Rp; k.out_t = W + D.bb_off; k.ld_t = D.Cp;
                for (int pnl = 0; pnl < D.Cp / 64; ++pnl) ttiles[1].push_back(TrsmTile{(int)jobs[1].size(), pnl});
                jobs[1].push_back(k);
            }
            P->e_a1.probs.push_back(a);
            for (int pnl = 0; pnl < D.Rp / 64; ++pnl) ttiles[0].push_back(TrsmTile{(int)jobs[0].size(), pnl});
            jobs[0].push_back(j);
        }
        for (int k = 0; k < 2; ++k) {
            P->n_trsm_tiles[k] = (unsigned)ttiles[k].size();
            if ((rc = upload(&P->d_trsm[k], jobs[k]))) return rc;
            if ((rc = upload(&P->d_trsm_tiles[k], ttiles[k]))) return rc;
        }
        P->n_uinv = (unsigned)uj.size();
        if ((rc = upload(&P->d_uinv, uj))) return rc;
    }
    for (Stage* s : {&P->g_P, &P->g_upd_a, &P->g_upd_b, &P->g_gram, &P->g_qupd, &P->g_rq, &P->g_rrq, &P->g_app_a[0], &P->g_app_a[1],
                     &P->g_app_b, &P->e_a1, &P->e_a2}) decide_tiling(P, s);
    P->fused_key.clear();      // the fused-update stages (psgdk_precond_grad_apply) are rebuilt from the new tables at their next call
    for (Stage* s : P->all_stages())
        if ((rc = finish_stage(*s))) return rc;
    return PSGDK_OK;
}

int psgdk_init_state(psgdk_plan* plan, double scale, void* stream) {
    if (!plan || !(scale > 0.0)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(plan->state, 0, plan->state_bytes, st));
    std::vector<float> sd(plan->dd.size()), sn(plan->dn.size());
    for (size_t i = 0; i < plan->dd.size(); ++i) sd[i] = (float)std::pow(scale, 1.0 / plan->order[plan->dd[i].tensor]);
    for (size_t i = 0; i < plan->dn.size(); ++i) sn[i] = (float)std::pow(scale, 1.0 / plan->order[plan->dn[i].tensor]);
    if (!sd.empty()) HIPCHK(hipMemcpyAsync(plan->d_scale_diag, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, st));
    if (!sn.empty()) HIPCHK(hipMemcpyAsync(plan->d_scale_dense, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, st));
    const unsigned nb = (unsigned)(plan->dd.size() + plan->dn.size());
    DISPATCH_T(plan, hipLaunchKernelGGL(init_factors_kernel<T>, dim3(nb), dim3(256), 0, st, plan->d_dd, (int)plan->dd.size(),
                                        plan->d_dn, (int)plan->dn.size(), plan->d_scale_diag, plan->d_scale_dense, plan->state));
    HIPCHK(hipGetLastError());
    plan->p_valid = false;
    return PSGDK_OK;
}

int psgdk_state_changed(psgdk_plan* plan, void* stream) {
    if (!plan) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    if (!plan->dn.empty()) {
        const unsigned nb = (unsigned)(plan->max_dp / 64);
        DISPATCH_T(plan, hipLaunchKernelGGL(transpose_q_kernel<T>, dim3(nb, nb, (unsigned)plan->dn.size()), dim3(256), 0, st,
                                            plan->d_dn, plan->state));
        HIPCHK(hipGetLastError());
    }
    plan->p_valid = false;
    return PSGDK_OK;
}

// balancing (psgd.py:266-275, drawn at psgd.py:318 / 418)
static int run_balance(psgdk_plan* P, const uint8_t* balance_mask, hipStream_t st) {
    if (balance_mask) {
        std::vector<int>& which = P->h_balance;      // plan-owned staging for the async upload
        which.clear();
        for (int t = 0; t < P->n_tensors; ++t)      // (row shards: psgdk_balance_phase, around the caller's max over the members)
            if (balance_mask[t] && P->factors[t].size() > 1 && P->td[t].kind != TK_GEN && P->shard_of(t) < 0) which.push_back(t);
        for (size_t gi = 0; gi < P->gd.size(); ++gi)
            if (balance_mask[P->gd[gi].tensor])
                DISPATCH_T(P, hipLaunchKernelGGL(gen_balance_kernel<T>, dim3(1), dim3(256), 0, st, P->d_gd, (int)gi, P->d_dd, P->d_dn, P->state));
        if (!which.empty()) {
            BalanceList inl{};
            if (which.size() <= 15) { inl.n = (int)which.size(); for (size_t i = 0; i < which.size(); ++i) inl.t[i] = which[i]; }
            else HIPCHK(hipMemcpyAsync(P->d_balance, which.data(), which.size() * sizeof(int), hipMemcpyHostToDevice, st));
            float* balnorm = (float*)(P->work + P->balnorm_off);
            if (!P->bal_clean) HIPCHK(hipMemsetAsync(balnorm, 0, 2 * which.size() * sizeof(float), st));
            P->bal_clean = false;
            const dim3 bg(64, (unsigned)(2 * which.size()));
            for (int phase = 0; phase < 2; ++phase)
                DISPATCH_T(P, hipLaunchKernelGGL(balance_kernel<T>, bg, dim3(256), 0, st, P->d_td, P->d_dd, P->d_dn,
                                                 P->d_balance, inl, P->state, balnorm, phase));
        }
    }
    return PSGDK_OK;
}

static int upload_ptrs(void** dst, std::vector<const void*>& cache, const void* const* src, size_t n, hipStream_t st) {
    if (cache.size() == n && std::memcmp(cache.data(), src, n * sizeof(void*)) == 0) return PSGDK_OK;
    cache.assign(src, src + n);
    // the source is the plan-owned cache (stable until the next differing call), not the caller's array
    HIPCHK(hipMemcpyAsync(dst, cache.data(), n * sizeof(void*), hipMemcpyHostToDevice, st));
    return PSGDK_OK;
}

int psgdk_accumulate(psgdk_plan* plan, const void* const* grads, int grad_dtype, const void* const* params,
                     int param_dtype, float coupled_wd, float beta, int keep_grad, const psgdk_damp* damp, void* stream) {
    if (!plan || !grads) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    ProfCall prof_call(plan, stream);
    if ((grad_dtype != PSGDK_BF16 && grad_dtype != PSGDK_F32) || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (coupled_wd != 0.f && !params) return PSGDK_ERR_INVALID;
    if (!(beta >= 0.f && beta < 1.f)) return PSGDK_ERR_INVALID;
    for (int t = 0; t < plan->n_tensors; ++t) if (!grads[t] || (coupled_wd != 0.f && !params[t])) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rcp;
    void** d_grads = nullptr; void** d_params = nullptr;
    if ((rcp = plan->ptrs_a.get(grads, plan->n_tensors, st, &d_grads))) return rcp;
    if (coupled_wd != 0.f && (rcp = plan->ptrs_b.get((const void* const*)params, plan->n_tensors, st, &d_params))) return rcp;
    const int keep = (keep_grad || !plan->use_momentum) ? 1 : 0;
    // optional fusion of the update's damped input X (psgd.py:402-403) into this pass (saves one read of the source)
    plan->x_valid = false;
    const void* const* ng = nullptr;
    int do_x = 0, x_from_grad = 0;
    float damping = 0.f; uint64_t seed = 0, offset = 0;
    if (damp) {
        if ((damp->source != PSGDK_SRC_EMA && damp->source != PSGDK_SRC_GRAD) || !(damp->damping >= 0.f)) return PSGDK_ERR_INVALID;
        if (damp->source == PSGDK_SRC_EMA && !plan->use_momentum) return PSGDK_ERR_INVALID;
        if (damp->noise) {
            if (!damp->noise->g_noise) return PSGDK_ERR_INVALID;
            HIPCHK(hipMemcpyAsync(plan->d_noise_g, damp->noise->g_noise, plan->n_tensors * sizeof(void*), hipMemcpyHostToDevice, st));
            ng = (const void* const*)plan->d_noise_g;
        }
        do_x = 1; x_from_grad = damp->source == PSGDK_SRC_GRAD; damping = damp->damping; seed = damp->seed; offset = damp->offset;
    }
    // psgdk_test_ew_mode (test hook, tools/ew_bench.py / ew_check.py): 1 = damped input without the noise (the pass's HBM floor; wrong
    // results), 2 = the general path for every tile (same results as the compile-time-resolved path, bit for bit)
    if (plan->ew_dbg & 1) do_x |= do_x << 1;
    if (plan->ew_dbg & 2) do_x |= do_x << 2;
    DISPATCH_T(plan, hipLaunchKernelGGL(accumulate_kernel<T>, dim3(plan->n_tiles_all), dim3(256), 0, st, plan->d_td,
                                        plan->d_tiles_all, (const void* const*)d_grads, (const void* const*)d_params,
                                        plan->state, plan->work, grad_dtype, param_dtype, coupled_wd, beta, plan->use_momentum, keep,
                                        do_x, x_from_grad, damping, ng, seed, offset, plan->geometry == PSGDK_GEOM_EQ ? 1 : 0,
                                        (unsigned long long)plan->hsumsq_off, (unsigned)plan->n_tensors,
                                        (unsigned long long)plan->zero_off, (unsigned long long)(damp ? plan->zero_bytes : 0)));
    HIPCHK(hipGetLastError());
    // this pass also cleared the sums of h^2 (the precond_grad of this step) and, when an update was announced, that update's
    // accumulators (row sums, scalars, arrival counters, balancing norms): one memset launch less for each
    plan->hsq_clean = true;
    plan->zero_clean = damp != nullptr;
    plan->clean_stream = st;
    if (damp) {
        plan->x_valid = true; plan->x_source = damp->source; plan->x_damping = damp->damping;
        plan->x_seed = damp->seed; plan->x_offset = damp->offset; plan->x_explicit = damp->noise != nullptr;
    }
    return PSGDK_OK;
}


// norm_lower_bound_spd (chain 0: A = term1 -> L, mu) / _skh (chain 1: A = R -> s) of every dense factor (psgd.py:46-93).
// One cooperative launch (nlb_coop_kernel: A read once, kept in registers by the workgroups of the factor) when the widest
// factor fits its register budget (bf16: dp <= 768, fp32: dp <= 384) and all workgroups are resident at once; otherwise start
// block, four grouped-GEMM products and the scalars as separate launches.  PSGDK_NLB_FUSED=0 at plan creation forces the
// latter (the tests compare the two).
static int nlb_plan_coop(psgdk_plan* P) {
    P->nlb_coop = false;
    if (P->dn.empty()) return PSGDK_OK;
    const int kstep = P->dtype == PSGDK_BF16 ? 32 : 16;
    // K steps of registers a member holds: 24 (768 bf16 / 384 fp32) with 256 or 128 columns per member, 32 (1024 / 512: GPT-2-medium's factors)
    // with 128 columns only -- 256 columns x 32 K steps would be the whole register file
    const int ksteps = P->max_dp / kstep;
    if (ksteps > 32) return PSGDK_OK;
    P->nlb_k32 = ksteps > 24;
    // members of a factor are dealt to one XCD (workgroup b is observed to run on XCD b % 8; speed only -- their slabs of A
    // then share an L2 -- the exchange protocol does not depend on it): per-XCD lists, longest-first packing
    std::vector<std::vector<NlbJob>> xcd(8);
    std::vector<int> order(P->dn.size());
    for (size_t f = 0; f < order.size(); ++f) order[f] = (int)f;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return P->dn[a].dp > P->dn[b].dp; });
    // plans whose widest factor is <= 128 wide (LeNet5, conv nets): every factor is ONE workgroup's, 16 columns per wave (all eight waves
    // work on a 128-wide factor) and as many K steps of registers as such a factor has -- the general instantiation loaded 24 K steps
    // (sized for 768) of which 16 were dead re-reads: 10.5 of the 39 us a bound took on LeNet5's plan (profiles/r04_a)
    P->nlb_small = P->max_dp <= 128;
    P->nlb_narrow = false;
    // one workgroup per CU (512 threads, ~250 VGPRs): all siblings are resident only if an XCD's share fits its CUs
    int dev = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus < 8) return PSGDK_OK;
    size_t len = 0;
    // Members of 128 columns where the plan is small enough for twice the members (a slab of A is half as long to pull into registers, a
    // product half as many matrix instructions, a published piece half the stores: tools/nlb_stamps.py), members of 256 otherwise.
    // PSGDK_NLB_NARROW=0 keeps the wide members (the GPU suite compares the two on one plan).
    const char* env_narrow = getenv("PSGDK_NLB_NARROW");
    // (round 6, second step) members of 192 columns -- twelve waves of 16 -- for the plans in between: GPT-2-small's 62 factors of 768 make 248
    // workgroups of four members each, where members of 128 would need 372
    for (int cols_per_wg : {128, 192, 256}) {
        if (cols_per_wg < 256 && !P->nlb_small && !P->nlb_k32 && env_narrow && env_narrow[0] == '0') continue;
        if (cols_per_wg > 128 && (P->nlb_k32 || P->nlb_small)) break;
        for (auto& l : xcd) l.clear();
        for (int f : order) {
            const int S = (P->dn[f].dp + cols_per_wg - 1) / cols_per_wg;
            size_t best = 0;
            for (size_t x = 1; x < 8; ++x) if (xcd[x].size() < xcd[best].size()) best = x;
            for (int m = 0; m < S; ++m) xcd[best].push_back(NlbJob{f, m, S, 0});
        }
        len = 0;
        for (auto& l : xcd) len = std::max(len, l.size());
        if (len <= (size_t)(cus / 8)) { P->nlb_narrow = cols_per_wg == 128 && !P->nlb_small; P->nlb_cols = cols_per_wg; break; }
    }
    if (len > (size_t)(cus / 8)) return PSGDK_OK;
    std::vector<NlbJob> jobs(8 * len, NlbJob{-1, 0, 0, 0});
    for (size_t x = 0; x < 8; ++x)
        for (size_t k = 0; k < xcd[x].size(); ++k) jobs[k * 8 + x] = xcd[x][k];
    // one entry more than the grid reads: the address of the stamp buffer of the instrumented instantiation (TS; test hook only)
    const size_t n_jobs = jobs.size();
    if (P->d_nlb_ts) { (void)hipFree(P->d_nlb_ts); P->d_nlb_ts = nullptr; }
    HIPCHK(hipMalloc((void**)&P->d_nlb_ts, n_jobs * NLB_TS_SLOTS * sizeof(unsigned long long)));
    {
        NlbJob tail{0, 0, 0, 0};
        static_assert(sizeof(NlbJob) >= sizeof(void*), "the stamp buffer's address must fit a job entry");
        std::memcpy(&tail, &P->d_nlb_ts, sizeof(void*));
        jobs.push_back(tail);
    }
    int rc = upload(&P->d_nlb_jobs, jobs);
    if (rc) return rc;
    if (!P->h_err) {
        void* h = nullptr; void* d = nullptr;
        HIPCHK(hipHostMalloc(&h, 64, hipHostMallocMapped));
        std::memset(h, 0, 64);
        HIPCHK(hipHostGetDevicePointer(&d, h, 0));
        P->h_err = (volatile unsigned*)h; P->d_err = (unsigned*)d;
    }
    P->n_nlb_jobs = (unsigned)n_jobs;
    // the subspace block (32 rows of the widest factor + 16 bytes each) + eight wave-private publish stages of 32 x (32 T + 16 bytes)
    P->nlb_lds = (unsigned)(32 * ((size_t)P->max_dp * P->esz + 16) + 8 * 32 * (32 * P->esz + 16));
    for (const void* k : {(const void*)nlb_coop_kernel<bf16_t, 2, 24>, (const void*)nlb_coop_kernel<float, 2, 24>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 4>, (const void*)nlb_coop_kernel<float, 1, 8>,
                          (const void*)nlb_coop_kernel<bf16_t, 2, 24, true>, (const void*)nlb_coop_kernel<float, 2, 24, true>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 4, true>, (const void*)nlb_coop_kernel<float, 1, 8, true>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 24>, (const void*)nlb_coop_kernel<float, 1, 24>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 24, true>, (const void*)nlb_coop_kernel<float, 1, 24, true>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 32>, (const void*)nlb_coop_kernel<float, 1, 32>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 24, false, 12>, (const void*)nlb_coop_kernel<float, 1, 24, false, 12>,
                          (const void*)nlb_coop_kernel<bf16_t, 1, 24, true, 12>, (const void*)nlb_coop_kernel<float, 1, 24, true, 12>})
        HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    P->nlb_coop = !P->nlb_unfused;      // (the job table exists either way: psgdk_test_nlb runs both routes on one plan)
    return PSGDK_OK;
}
static int run_nlb(psgdk_plan* P, int chain, const void* const* noise, uint64_t seed, uint64_t offset, float lr, float betaL,
                   int add_c, int pro_iter, hipStream_t st, int route = -1, int fault = 0, bool stamps = false) {
    const unsigned F = (unsigned)P->dn.size();
    if (route < 0 ? P->nlb_coop : (route == 1)) {
        const bool bf = P->dtype == PSGDK_BF16;
        if (stamps && P->nlb_k32) return PSGDK_ERR_UNSUPPORTED;      // (no instrumented instantiation of the 32-K-step variant)
        const void* k = stamps ? (P->nlb_small ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 4, true> : (const void*)nlb_coop_kernel<float, 1, 8, true>)
                                  : P->nlb_cols == 192 ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 24, true, 12> : (const void*)nlb_coop_kernel<float, 1, 24, true, 12>)
                                  : P->nlb_narrow ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 24, true> : (const void*)nlb_coop_kernel<float, 1, 24, true>)
                                                  : (bf ? (const void*)nlb_coop_kernel<bf16_t, 2, 24, true> : (const void*)nlb_coop_kernel<float, 2, 24, true>))
                      : P->nlb_small ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 4> : (const void*)nlb_coop_kernel<float, 1, 8>)
                      : P->nlb_cols == 192 ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 24, false, 12> : (const void*)nlb_coop_kernel<float, 1, 24, false, 12>)
                      : P->nlb_k32 ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 32> : (const void*)nlb_coop_kernel<float, 1, 32>)
                      : P->nlb_narrow ? (bf ? (const void*)nlb_coop_kernel<bf16_t, 1, 24> : (const void*)nlb_coop_kernel<float, 1, 24>)
                                      : (bf ? (const void*)nlb_coop_kernel<bf16_t, 2, 24> : (const void*)nlb_coop_kernel<float, 2, 24>);
        const DenseDesc* dn = P->d_dn; const NlbJob* jobs = P->d_nlb_jobs; unsigned* err = P->d_err;
        unsigned char* state = P->state; unsigned char* work = P->work;
        void* args[] = {&dn, &jobs, &err, &state, &work, &chain, &noise, &seed, &offset, &lr, &betaL, &add_c, &pro_iter, &fault};
        ++g_psgdk_launches;
        HIPCHK(hipLaunchKernel(k, dim3(P->n_nlb_jobs), dim3(P->nlb_cols == 192 ? 768 : 512), args, P->nlb_lds, st));
        return PSGDK_OK;
    }
    DISPATCH_T(P, hipLaunchKernelGGL(nlb_init_kernel<T>, dim3(8, F), dim3(256), 0, st, P->d_dn, P->work, chain, noise, seed, offset, pro_iter));
    for (int p = 0; p < 4; ++p) launch_stage(P, P->g_nlb[chain][p], st);
    DISPATCH_T(P, hipLaunchKernelGGL(nlb_finalize_kernel<T>, dim3(F), dim3(64), 0, st, P->d_dn, P->state, P->work, chain, lr, betaL, add_c,
                                     pro_iter >= 0 ? 1 : 0));
    return PSGDK_OK;
}

// Did a cooperative norm-bound launch of an EARLIER call give up waiting for a sibling (its factors skipped that update; the
// state is valid)?  Non-blocking: the kernels write the word straight into host memory.  Reported once; the plan then stays on
// the multi-launch route.
static int nlb_check_error(psgdk_plan* P) {
    if (P->h_err && *P->h_err != 0u) {
        *P->h_err = 0u;
        P->nlb_coop = false;
        ++P->nlb_fallbacks;
        return PSGDK_ERR_NLB_TIMEOUT;
    }
    return PSGDK_OK;
}

static int ensure_P(psgdk_plan* plan, hipStream_t st) {
    if (!plan->p_valid) {
        if (plan->p_mode()) {      // P := Q (exprA applies every factor once; Q is symmetric in this geometry)
            if (!plan->dn.empty())
                DISPATCH_T(plan, hipLaunchKernelGGL(copy_q_to_p_kernel<T>, dim3(16, (unsigned)plan->dn.size()), dim3(256), 0, st, plan->d_dn,
                                                    plan->state, plan->work));
        } else
        launch_stage(plan, plan->g_P, st);
        HIPCHK(hipGetLastError());
        plan->p_valid = true;
    }
    return PSGDK_OK;
}

// The three geometries that share Pg = (kron Q^T Q)(G + damped noise), the mode Grams and the spectral-norm step
// normalisation: Q0.5EQ1.5 (psgd.py:394-419), QEQ (psgd.py:367-391), QUAD (psgd.py:455-483).
// phase: -1 = the whole update; 0 / 1 = the two halves of a PHASED update of a plan with row shards (psgdk_update_precond_begin / _finish):
// 0 ends with the members' partial mode Grams and diagonal maxima in record `shard_member` of `xchg`; 1 starts from all members' records.
static int update_whiten_family(psgdk_plan* plan, int variant, int source, float lr, float betaL, float damping,
                                const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                                const uint8_t* balance_mask, void* stream, int phase = -1, void* xchg_ = nullptr) {
    if (!plan || (source != PSGDK_SRC_EMA && source != PSGDK_SRC_GRAD)) return PSGDK_ERR_INVALID;
    if (!plan->state || plan->geometry != variant) return PSGDK_ERR_STATE;
    if (phase < 0 && !plan->shards.empty()) return PSGDK_ERR_STATE;         // row shards need the exchange between the two halves
    if (phase >= 0 && (plan->shards.empty() || !xchg_ || ((uintptr_t)xchg_ & 255))) return PSGDK_ERR_INVALID;
    if (phase >= 0 && plan->update_open != (phase == 1)) return PSGDK_ERR_STATE;
    unsigned char* xchg = (unsigned char*)xchg_;
    const bool do_a = phase != 1;
    ProfCall prof_call(plan, stream);
    if (source == PSGDK_SRC_EMA && !plan->use_momentum) return PSGDK_ERR_INVALID;
    if (!(lr > 0.f) || !(betaL >= 0.f && betaL <= 1.f) || !(damping >= 0.f)) return PSGDK_ERR_INVALID;
    const bool need_skh = variant == PSGDK_GEOM_Q0P5EQ1P5 || variant == PSGDK_GEOM_PRO4P;
    if (noise && (!noise->g_noise || (!plan->dn.empty() && (!noise->spd_noise || (need_skh && !noise->skh_noise))))) return PSGDK_ERR_INVALID;
    const float lr_eff = variant == PSGDK_GEOM_QUAD ? 0.5f * lr : lr;       // QUAD takes two half steps (psgd.py:473,479-480)
    const bool quadlike = variant == PSGDK_GEOM_QUAD || variant == PSGDK_GEOM_QUAD4P;   // QUAD4P: two full steps on P itself
    const bool qep = variant == PSGDK_GEOM_QEP;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    const unsigned F = (unsigned)P->dn.size();
    int rc;
    if (do_a && (rc = nlb_check_error(P))) return rc;
    if (noise) {
        HIPCHK(hipMemcpyAsync(P->d_noise_g, noise->g_noise, P->n_tensors * sizeof(void*), hipMemcpyHostToDevice, st));
        if (F) {
            std::vector<const void*>& a = P->h_noise_a; std::vector<const void*>& b = P->h_noise_b;
            a.resize(F); b.resize(F);
            for (unsigned f = 0; f < F; ++f) {
                const int slot = P->dn[f].tensor * PSGDK_GEN_MAXDIM + P->dense_dim[f];
                a[f] = noise->spd_noise[slot]; b[f] = need_skh ? noise->skh_noise[slot] : nullptr;
                if (!a[f] || (need_skh && !b[f])) return PSGDK_ERR_INVALID;
            }
            HIPCHK(hipMemcpyAsync(P->d_noise_spd, a.data(), F * sizeof(void*), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(P->d_noise_skh, b.data(), F * sizeof(void*), hipMemcpyHostToDevice, st));
        }
    }
    const void* const* ng = noise ? (const void* const*)P->d_noise_g : nullptr;
    const void* const* nspd = noise ? (const void* const*)P->d_noise_spd : nullptr;
    const void* const* nskh = noise ? (const void* const*)P->d_noise_skh : nullptr;

    if (do_a) {
        if (!P->zero_clean || P->clean_stream != st) HIPCHK(hipMemsetAsync(P->work + P->zero_off, 0, P->zero_bytes, st));
        P->zero_clean = false;
        P->bal_clean = true;
        if (qep) {      // balancing is not optional for QEP and comes first (psgd.py:346-347)
            std::vector<uint8_t> all(P->n_tensors, 1);
            if ((rc = run_balance(P, all.data(), st))) return rc;
            P->p_valid = false;
        }
        // damped input X (psgd.py:402-403), unless psgdk_accumulate already produced exactly this X
        const bool x_ready = P->x_valid && P->x_source == source && P->x_damping == damping && P->x_seed == seed &&
                             P->x_offset == offset && P->x_explicit == (noise != nullptr);
        P->x_valid = false;
        if (!x_ready)
            DISPATCH_T(P, hipLaunchKernelGGL(make_x_kernel<T>, dim3(P->n_tiles_all), dim3(256), 0, st, P->d_td, P->d_tiles_all, ng,
                                             P->state, P->work, source == PSGDK_SRC_GRAD ? 1 : 0, damping, seed, offset));
        // Pg = (kron Q^T Q) X and the mode Grams (psgd.py:403-405)
        if ((rc = ensure_P(P, st))) return rc;
        {
            // the persistent 256 x 256 launch of the update's first product: workgroups that walk one tile fewer start up to ~0.9 tile times
            // late (GemmUpdArgs::stagger; a tile of K columns takes ~44 K ns): their output bursts (Pg^T, 128 KiB per tile) fall between the
            // others'.  GPT-2-small: -10 us per step (r6j); psgdk_test_fuse_mode(.., 0) switches it off for A/B runs
            int kmax = 0;
            if (P->g_upd_a.big) for (const GemmProblem& g : P->g_upd_a.probs) kmax = std::max(kmax, g.K);
            GemmUpdArgs sa{0.f, 1.f, 0.f, 0, P->fuse_stagger == 0 ? 0 : std::min(4 * kmax, 16000), 0};
            launch_stage(P, P->g_upd_a, st, sa.stagger ? &sa : nullptr);
        }
        launch_stage(P, P->g_upd_b, st);
        if (P->n_tiles_diag)
            DISPATCH_T(P, hipLaunchKernelGGL(diag_tensor_kernel<T>, dim3(P->n_tiles_diag), dim3(256), 0, st, P->d_td, P->d_dd,
                                             P->d_tiles_diag, P->state, P->work, 0, 0, (float*)(P->work + P->hsumsq_off), P->p_mode() ? 1 : 0));
        // N-D tensors: Pg mode by mode, then every mode's Gram (dense -> term1 + its row stats; diagonal -> the sum vector)
        for (const GenDesc& g : P->gd) {
            const TensorDesc& D = P->td[g.tensor];
            DISPATCH_T(P, {
                const T* Pg = gen_apply_chain<T>(P, g, (const T*)(P->work + D.x_off), (T*)nullptr, (float*)nullptr, st);
                int64_t A = 1;
                for (int i = 0; i < g.ndim; ++i) {
                    const int s_ = g.dims[i];
                    const int64_t B = D.numel / (A * s_);
                    if (g.fkind[i] == PSGDK_DENSE) {
                        const DenseDesc& Fd = P->dn[g.fidx[i]];
                        T* T1 = (T*)(P->work + Fd.t1_off);
                        // chunks of the (a, b) range: enough workgroups to pull the tensor out of HBM at speed (about a thousand), at
                        // least 4096 terms each, and Z s^2 partial sums within the factor's scratch
                        const int64_t blocks = (int64_t)s_ * ((s_ + 63) / 64);
                        const int64_t AB = D.numel / s_;
                        int64_t Z = std::max<int64_t>(1, std::min<int64_t>(256, 1024 / blocks));
                        Z = std::min<int64_t>(Z, std::max<int64_t>(1, AB / 4096));
                        Z = std::min<int64_t>(Z, std::max<int64_t>(1, (int64_t)PSGDK_GEN_GPART / ((int64_t)s_ * s_)));
                        if (Z > 1 && Fd.gpart_off) {
                            float* gp = (float*)(P->work + Fd.gpart_off);
                            hipLaunchKernelGGL(gen_gram_kernel<T>, dim3(s_, (s_ + 63) / 64, (unsigned)Z), dim3(256), 0, st, Pg, (int)A, s_, (int)B, 1, T1,
                                               Fd.dp, (float*)nullptr, gp);
                            hipLaunchKernelGGL(gen_gram_finish_kernel<T>, dim3(s_), dim3(256), 0, st, (const float*)gp, (int)Z, s_, T1, Fd.dp);
                        } else
                        hipLaunchKernelGGL(gen_gram_kernel<T>, dim3(s_, (s_ + 63) / 64), dim3(256), 0, st, Pg, (int)A, s_, (int)B, 1, T1, Fd.dp,
                                           (float*)nullptr, (float*)nullptr);
                        hipLaunchKernelGGL(gen_rowstats_kernel<T>, dim3(s_), dim3(256), 0, st, (const T*)T1, s_, Fd.dp,
                                           (float*)(P->work + Fd.rowss_off), (float*)(P->work + Fd.sc_off) + DS_NF);
                    } else {
                        hipLaunchKernelGGL(gen_gram_kernel<T>, dim3(s_, 1), dim3(256), 0, st, Pg, (int)A, s_, (int)B, 0, (T*)nullptr, 0,
                                           (float*)(P->work + P->dd[g.fidx[i]].sum_off));
                    }
                    A *= s_;
                }
            });
        }
        if (F) {
            launch_stage(P, P->g_gram, st);
            for (int f : P->split_dense) {
                const DenseDesc& D = P->dn[f];
                const GemmProblem& g = P->g_gram.probs[P->gram_prob[f]];
                const int nks = (g.K + g.kchunk - 1) / g.kchunk;
                const unsigned nblk = (unsigned)((D.dp / 64) * (D.dp / 64 + 1) / 2);
                const int sh = P->shard_of(D.tensor);
                if (sh >= 0)      // a row shard's Gram is PARTIAL: the fp32 sum of its slabs goes into this member's exchange record
                    hipLaunchKernelGGL(slab_sum_upper_kernel, dim3(nblk), dim3(256), 0, st, (const float*)g.slab, nks, D.dp,
                                       (float*)(xchg + (size_t)P->shard_member * P->xchg_record_bytes + P->shards[sh].rec_off));
                else
                    DISPATCH_T(P, hipLaunchKernelGGL(splitk_reduce_sym_kernel<T>, dim3(nblk), dim3(256), 0, st,
                                                     (const float*)g.slab, (T*)g.C, D.dp, D.dp, nks, 1.0f, g.row_sumsq, g.diag_max, D.d, (size_t)D.dp * D.dp));
            }
        }
        if (phase == 0) {
            // ... and its own maximum of the diagonal factor's term1 (the rest of that factor's update is row-local)
            DISPATCH_T(P, hipLaunchKernelGGL(diag_update_kernel<T>, dim3((unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd, P->state, P->work,
                                             (float*)(P->work + P->diag_mu_off), 2, lr_eff, betaL, 0, xchg, (unsigned long long)P->xchg_record_bytes,
                                             P->shard_members, P->shard_member));
            HIPCHK(hipGetLastError());
            P->update_open = true;
            return PSGDK_OK;
        }
    }
    if (phase == 1) {
        // the shards' mode Grams from ALL members' partials (summed in member order: every member forms the same term1), rounded once,
        // with the row statistics the unsplit epilogue would have produced
        P->update_open = false;
        for (const auto& sh : P->shards) {
            const int f = P->td[sh.tensor].col_dense;
            const DenseDesc& D = P->dn[f];
            const GemmProblem& g = P->g_gram.probs[P->gram_prob[f]];
            DISPATCH_T(P, hipLaunchKernelGGL(splitk_reduce_sym_kernel<T>, dim3((unsigned)((D.dp / 64) * (D.dp / 64 + 1) / 2)), dim3(256), 0, st,
                                             (const float*)(xchg + sh.rec_off), (T*)g.C, D.dp, D.dp, P->shard_members, 1.0f, g.row_sumsq, g.diag_max,
                                             D.d, P->xchg_record_bytes / 4));
        }
    }
    if (F) {
        const dim3 grows((unsigned)(P->max_dp / 64), F);
        // ell = ||term1||_lb + numel/d, L, mu (psgd.py:413-414 -> 46-68); row stats of term1 came with the Gram
        if (qep) {
            // term1 = Q T1 Q^T (-> t1), term2 = c Q Q^T (-> rq); S = term1 + term2 (-> t1), D = term1 - term2 (-> r)
            launch_stage(P, P->v_qep_u, st);
            launch_stage(P, P->v_qep_t1, st);
            launch_stage(P, P->v_qep_t2, st);
            DISPATCH_T(P, hipLaunchKernelGGL(zero_scalar_kernel, dim3((F + 63) / 64), dim3(64), 0, st, P->d_dn, P->work, (int)F, (int)DS_NF));
            DISPATCH_T(P, hipLaunchKernelGGL(eq_combine_kernel<T>, grows, dim3(256), 0, st, P->d_dn, P->work, 0));
        }
        if ((rc = run_nlb(P, 0, nspd, seed, offset, lr_eff, betaL, qep ? 0 : 1, -1, st))) return rc;
        if (qep) {
            launch_stage(P, P->e_qupd, st);       // Q' = Q - mu (term1 - term2) Q (psgd.py:364)
            DISPATCH_T(P, hipLaunchKernelGGL(eq_commit_q_kernel<T>, dim3(16, F), dim3(256), 0, st, P->d_dn, P->state, P->work));
        } else if (variant == PSGDK_GEOM_Q0P5EQ1P5) {
            // Q' = Q - mu (term1 Q - c Q) (psgd.py:415)
            launch_stage(P, P->g_qupd, st);
            // procrustes_step2 (psgd.py:416 -> 101-124); its line search and AXPY are fused into the R RQ product
            {
                const unsigned nb = (unsigned)(P->max_dp / 64);
                DISPATCH_T(P, hipLaunchKernelGGL(rsub_t_kernel<T>, dim3(nb * (nb + 1) / 2, F), dim3(256), 0, st, P->d_dn, P->work));
            }
            if ((rc = run_nlb(P, 1, nskh, seed, offset, lr, betaL, 1, -1, st))) return rc;
            launch_stage(P, P->g_rq, st);
            launch_stage(P, P->g_rrq, st);
        } else if (variant == PSGDK_GEOM_PRO4P) {
            // P' = P - mu (term1 P - c P), then up to 10 procrustes_step3 rotations per factor; each factor stops once it is
            // Hermitian to 1e-3 (device flag, psgd.py:447-450); finally into the state
            launch_stage(P, P->g_qupd, st);
            for (int k = 0; k < 10; ++k) {
                DISPATCH_T(P, hipLaunchKernelGGL(pro_reset_kernel<T>, dim3(F), dim3(64), 0, st, P->d_dn, P->work));
                DISPATCH_T(P, hipLaunchKernelGGL(rsub_kernel<T>, grows, dim3(256), 0, st, P->d_dn, P->work, 1));
                if (k > 0) DISPATCH_T(P, hipLaunchKernelGGL(pro_decide_kernel<T>, dim3(F), dim3(64), 0, st, P->d_dn, P->work));
                if ((rc = run_nlb(P, 1, nskh, seed, offset, lr, betaL, 1, k, st))) return rc;
                launch_stage(P, P->v_pro_rq, st);
                launch_stage(P, P->v_pro_rrq, st);
                launch_stage(P, P->v_pro_rrrq, st);
            }
            DISPATCH_T(P, hipLaunchKernelGGL(eq_commit_q_kernel<T>, dim3(16, F), dim3(256), 0, st, P->d_dn, P->state, P->work));
        } else if (variant == PSGDK_GEOM_QEQ) {
            // Q' = Q - mu (Q term1 - c Q) (psgd.py:388), then into the state
            launch_stage(P, P->v_qeq, st);
            DISPATCH_T(P, hipLaunchKernelGGL(eq_commit_q_kernel<T>, dim3(16, F), dim3(256), 0, st, P->d_dn, P->state, P->work));
        } else {
            // p = q - mu/2 (term1 q - c q);  p = p - mu/2 (p term1 - c p);  q = (p + p^T)/2  (psgd.py:479-481)
            launch_stage(P, P->g_qupd, st);
            launch_stage(P, P->v_quad2, st);
            DISPATCH_T(P, hipLaunchKernelGGL(quad_symmetrize_kernel<T>, dim3(16, F), dim3(256), 0, st, P->d_dn, P->state, P->work));
        }
    }
    // diagonal factors (psgd.py:406-410); after every GEMM that still reads the old diagonals
    if (!P->dd.empty() && qep) {
        float* mu = (float*)(P->work + P->diag_mu_off);
        const unsigned chunks = (unsigned)std::max(1, std::min(16, (P->max_diag_len + 4095) / 4096));
        DISPATCH_T(P, hipLaunchKernelGGL(qep_diag_update_kernel<T>, dim3((unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd, P->state,
                                         P->work, mu, 0, lr, betaL));
        DISPATCH_T(P, hipLaunchKernelGGL(qep_diag_update_kernel<T>, dim3(chunks, (unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd,
                                         P->state, P->work, mu, 1, lr, betaL));
    } else if (!P->dd.empty())
    {
        float* mu = (float*)(P->work + P->diag_mu_off);
        const unsigned chunks = (unsigned)std::max(1, std::min(16, (P->max_diag_len + 4095) / 4096));
        // short diagonals (one workgroup walks a whole factor): both phases in one launch
        DISPATCH_T(P, hipLaunchKernelGGL(diag_update_kernel<T>, dim3((unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd, P->state,
                                         P->work, mu, chunks == 1 ? 3 : 0, lr_eff, betaL, quadlike ? 1 : 0, xchg,
                                         (unsigned long long)P->xchg_record_bytes, P->shard_members, P->shard_member));
        if (chunks > 1)
            DISPATCH_T(P, hipLaunchKernelGGL(diag_update_kernel<T>, dim3(chunks, (unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd,
                                             P->state, P->work, mu, 1, lr_eff, betaL, quadlike ? 1 : 0));
    }
    if (!qep && (rc = run_balance(P, balance_mask, st))) return rc;
    HIPCHK(hipGetLastError());
    P->p_valid = false;
    return PSGDK_OK;
}

int psgdk_update_precond_q0p5eq1p5(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                                   uint64_t seed, uint64_t offset, const uint8_t* balance_mask, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_Q0P5EQ1P5, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream);
}
int psgdk_update_precond_begin(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise, uint64_t seed,
                               uint64_t offset, void* exchange, void* stream) {
    if (!plan) return PSGDK_ERR_INVALID;
    return update_whiten_family(plan, plan->geometry, source, lr, betaL, damping, noise, seed, offset, nullptr, stream, 0, exchange);
}
int psgdk_update_precond_finish(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise, uint64_t seed,
                                uint64_t offset, const void* exchange, const uint8_t* balance_mask, void* stream) {
    if (!plan) return PSGDK_ERR_INVALID;
    return update_whiten_family(plan, plan->geometry, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream, 1,
                                (void*)exchange);
}
// balance_kron_precond (psgd.py:266-275) of ROW SHARDS in two calls: phase 0 leaves max |q| of the shard's two factors in the plan's
// balancing slots (PSGDK_INFO_BALNORM_OFFSET: 2 floats per flagged shard, in tensor order; the caller takes the maximum over the
// members), phase 1 rescales with what it finds there.  The same mask in both calls; only row shards may be flagged.
int psgdk_balance_phase(psgdk_plan* plan, const uint8_t* mask, int phase, void* stream) {
    if (!plan || !mask || (phase != 0 && phase != 1)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    std::vector<int>& which = P->h_balance;
    which.clear();
    for (int t = 0; t < P->n_tensors; ++t)
        if (mask[t]) { if (P->shard_of(t) < 0) return PSGDK_ERR_INVALID; which.push_back(t); }
    if (which.empty()) return PSGDK_OK;
    if (which.size() > 15) return PSGDK_ERR_UNSUPPORTED;
    BalanceList inl{};
    inl.n = (int)which.size();
    for (size_t i = 0; i < which.size(); ++i) inl.t[i] = which[i];
    float* balnorm = (float*)(P->work + P->balnorm_off);
    if (phase == 0) HIPCHK(hipMemsetAsync(balnorm, 0, 2 * which.size() * sizeof(float), st));
    P->bal_clean = false;
    DISPATCH_T(P, hipLaunchKernelGGL(balance_kernel<T>, dim3(64, (unsigned)(2 * which.size())), dim3(256), 0, st, P->d_td, P->d_dd, P->d_dn,
                                     P->d_balance, inl, P->state, balnorm, phase));
    HIPCHK(hipGetLastError());
    if (phase == 1) P->p_valid = false;
    return PSGDK_OK;
}
int psgdk_update_precond_qeq(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                             uint64_t seed, uint64_t offset, const uint8_t* balance_mask, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_QEQ, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream);
}
int psgdk_update_precond_quad(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                              uint64_t seed, uint64_t offset, const uint8_t* balance_mask, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_QUAD, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream);
}
int psgdk_update_precond_quad4p(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                                uint64_t seed, uint64_t offset, const uint8_t* balance_mask, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_QUAD4P, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream);
}
int psgdk_update_precond_pro4p(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                               uint64_t seed, uint64_t offset, const uint8_t* balance_mask, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_PRO4P, source, lr, betaL, damping, noise, seed, offset, balance_mask, stream);
}
int psgdk_update_precond_qep(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise,
                             uint64_t seed, uint64_t offset, void* stream) {
    return update_whiten_family(plan, PSGDK_GEOM_QEP, source, lr, betaL, damping, noise, seed, offset, nullptr, stream);
}

// shape of the bf16 solve: 0 = 32-row panels, two workgroups per CU; 1 = 64-row panels (test hook: psgdk_test_trsm_bench)
#define TRSM_SHAPE_DEFAULT 0
static int trsm_shape() { return TRSM_SHAPE_DEFAULT; }
static int launch_trsm_bf16(const TrsmJob* jobs, const TrsmTile* tiles, unsigned n_tiles, int max_dp, hipStream_t st, int dbg = 0,
                            int shape = -1) {
    if (shape < 0) shape = trsm_shape();
    const int rows = shape == 0 ? 32 : 64;
    const unsigned shm = (unsigned)rows * (unsigned)(max_dp + 8) * 2u + 64u * (unsigned)(rows + 4) * 4u;
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)eq_trsm_bf16_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)eq_trsm_bf16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const unsigned n_pan = n_tiles * (64u / (unsigned)rows);
    const unsigned grid = ((n_pan + 7) / 8) * 8;
    if (shape == 1) hipLaunchKernelGGL(eq_trsm_bf16_kernel<64>, dim3(grid), dim3(256), shm, st, jobs, tiles, (int)n_tiles, dbg);
    else hipLaunchKernelGGL(eq_trsm_bf16_kernel<32>, dim3(grid), dim3(256), shm, st, jobs, tiles, (int)n_tiles, dbg);
    return PSGDK_OK;
}

int psgdk_update_precond_eq(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                            const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                            const uint8_t* balance_mask, void* stream) {
    if (!plan || (source != PSGDK_SRC_EMA && source != PSGDK_SRC_GRAD)) return PSGDK_ERR_INVALID;
    if (!plan->state || plan->geometry != PSGDK_GEOM_EQ) return PSGDK_ERR_STATE;
    ProfCall prof_call(plan, stream);
    if (source == PSGDK_SRC_EMA && !plan->use_momentum) return PSGDK_ERR_INVALID;
    if (!(lr > 0.f) || !(betaL >= 0.f && betaL <= 1.f) || !(damping >= 0.f)) return PSGDK_ERR_INVALID;
    if (noise && (!noise->g_noise || (!plan->dn.empty() && !noise->spd_noise))) return PSGDK_ERR_INVALID;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    const unsigned F = (unsigned)P->dn.size();
    { const int rce = nlb_check_error(P); if (rce) return rce; }
    if (noise) {
        HIPCHK(hipMemcpyAsync(P->d_noise_g, noise->g_noise, P->n_tensors * sizeof(void*), hipMemcpyHostToDevice, st));
        if (F) {
            P->h_noise_a.resize(F);
            for (unsigned f = 0; f < F; ++f) {
                P->h_noise_a[f] = noise->spd_noise[P->dn[f].tensor * PSGDK_GEN_MAXDIM + P->dense_dim[f]];
                if (!P->h_noise_a[f]) return PSGDK_ERR_INVALID;
            }
            HIPCHK(hipMemcpyAsync(P->d_noise_spd, P->h_noise_a.data(), F * sizeof(void*), hipMemcpyHostToDevice, st));
        }
    }
    const void* const* ng = noise ? (const void* const*)P->d_noise_g : nullptr;
    const void* const* nspd = noise ? (const void* const*)P->d_noise_spd : nullptr;
    if (!P->zero_clean || P->clean_stream != st) HIPCHK(hipMemsetAsync(P->work + P->zero_off, 0, P->zero_bytes, st));
    P->zero_clean = false;
    P->bal_clean = true;
    // V and Hvp = S + (damping + eps|S|) V (psgd.py:334-336), unless psgdk_accumulate already wrote exactly this pair
    const bool x_ready = P->x_valid && P->x_source == source && P->x_damping == damping && P->x_seed == seed &&
                         P->x_offset == offset && P->x_explicit == (noise != nullptr);
    P->x_valid = false;
    if (!x_ready)
        DISPATCH_T(P, hipLaunchKernelGGL(eq_make_xv_kernel<T>, dim3(P->n_tiles_all), dim3(256), 0, st, P->d_td, P->d_tiles_all, ng,
                                         P->state, P->work, source == PSGDK_SRC_GRAD ? 1 : 0, damping, seed, offset));
    // A = (kron Q) Hvp (psgd.py:295) and B = V x_i Q_i^{-T} (psgd.py:297-303)
    launch_stage(P, P->e_a1, st);
    launch_stage(P, P->e_a2, st);
    if (P->n_uinv)
        DISPATCH_T(P, hipLaunchKernelGGL(eq_uinv_kernel<T>, dim3((unsigned)(P->max_dp / 64), P->n_uinv), dim3(64), 0, st, P->d_uinv));
    for (int k = 0; k < 2; ++k)
        if (P->n_trsm_tiles[k]) {
            if (P->dtype == PSGDK_BF16 && P->max_dp <= EQ_TRSM_BF16_MAX_DP) {
                const int rc = launch_trsm_bf16(P->d_trsm[k], P->d_trsm_tiles[k], P->n_trsm_tiles[k], P->max_dp, st);
                if (rc) return rc;
            } else
                DISPATCH_T(P, hipLaunchKernelGGL(eq_trsm_kernel<T>, dim3(P->n_trsm_tiles[k]), dim3(256), 0, st, P->d_trsm[k], P->d_trsm_tiles[k]));
        }
    if (P->n_tiles_diag)
        DISPATCH_T(P, hipLaunchKernelGGL(eq_diag_tensor_kernel<T>, dim3(P->n_tiles_diag), dim3(256), 0, st, P->d_td, P->d_dd,
                                         P->d_tiles_diag, P->state, P->work));
    // N-D tensors: A and B mode by mode (two ping-pong pairs), then every mode's two Grams
    for (const GenDesc& g : P->gd) {
        const TensorDesc& D = P->td[g.tensor];
        DISPATCH_T(P, {
            const T* At = gen_apply_chain<T>(P, g, (const T*)(P->work + D.x_off), (T*)nullptr, (float*)nullptr, st, 1, 0);
            const T* Bt = gen_apply_chain<T>(P, g, (const T*)(P->work + D.v_off), (T*)nullptr, (float*)nullptr, st, 2, 1);
            int64_t A = 1;
            for (int i = 0; i < g.ndim; ++i) {
                const int s_ = g.dims[i];
                const int64_t B = D.numel / (A * s_);
                if (g.fkind[i] == PSGDK_DENSE) {
                    const DenseDesc& Fd = P->dn[g.fidx[i]];
                    const dim3 gg(s_, (s_ + 63) / 64);
                    hipLaunchKernelGGL(gen_gram_kernel<T>, gg, dim3(256), 0, st, At, (int)A, s_, (int)B, 1, (T*)(P->work + Fd.t1_off), Fd.dp,
                                       (float*)nullptr);
                    hipLaunchKernelGGL(gen_gram_kernel<T>, gg, dim3(256), 0, st, Bt, (int)A, s_, (int)B, 1, (T*)(P->work + Fd.rq_off), Fd.dp,
                                       (float*)nullptr);
                } else {
                    const DiagDesc& Gd = P->dd[g.fidx[i]];
                    hipLaunchKernelGGL(gen_gram_kernel<T>, dim3(s_, 1), dim3(256), 0, st, At, (int)A, s_, (int)B, 0, (T*)nullptr, 0,
                                       (float*)(P->work + Gd.sum_off));
                    hipLaunchKernelGGL(gen_gram_kernel<T>, dim3(s_, 1), dim3(256), 0, st, Bt, (int)A, s_, (int)B, 0, (T*)nullptr, 0,
                                       (float*)(P->work + Gd.sum2_off));
                }
                A *= s_;
            }
        });
    }
    if (F) {
        // term1, term2 (psgd.py:306-307)
        for (int which = 0; which < 2; ++which) {
            Stage& G = which ? P->e_g2 : P->e_g1;
            launch_stage(P, G, st);
            for (unsigned f = 0; f < F; ++f) {
                if (P->e_gram_prob[f] < 0) continue;
                const GemmProblem& g = G.probs[P->e_gram_prob[f]];
                if (!(g.flags & GF_SPLITK)) continue;
                const DenseDesc& D = P->dn[f];
                const int nks = (g.K + g.kchunk - 1) / g.kchunk;
                DISPATCH_T(P, hipLaunchKernelGGL(splitk_reduce_sym_kernel<T>, dim3((unsigned)((D.dp / 64) * (D.dp / 64 + 1) / 2)), dim3(256), 0,
                                                 st, (const float*)g.slab, (T*)g.C, D.dp, D.dp, nks, 1.0f, (float*)nullptr, (float*)nullptr, D.d, (size_t)D.dp * D.dp));
            }
        }
        const dim3 grows((unsigned)(P->max_dp / 64), F);
        DISPATCH_T(P, hipLaunchKernelGGL(eq_combine_kernel<T>, grows, dim3(256), 0, st, P->d_dn, P->work, 1));
        // ell = ||term1 + term2||_lb, L, mu (psgd.py:314-315 -> 46-68)
        { const int rcn = run_nlb(P, 0, nspd, seed, offset, lr, betaL, 0, -1, st); if (rcn) return rcn; }
        // Q -= mu triu(term1 - term2) Q (psgd.py:316)
        launch_stage(P, P->e_qupd, st);
        DISPATCH_T(P, hipLaunchKernelGGL(eq_commit_q_kernel<T>, dim3(16, F), dim3(256), 0, st, P->d_dn, P->state, P->work));
    }
    if (!P->dd.empty()) {
        float* mu = (float*)(P->work + P->diag_mu_off);
        DISPATCH_T(P, hipLaunchKernelGGL(eq_diag_update_kernel<T>, dim3((unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd, P->state,
                                         P->work, mu, 0, lr, betaL));
        const unsigned chunks = (unsigned)std::max(1, std::min(16, (P->max_diag_len + 4095) / 4096));
        DISPATCH_T(P, hipLaunchKernelGGL(eq_diag_update_kernel<T>, dim3(chunks, (unsigned)P->dd.size()), dim3(1024), 0, st, P->d_dd,
                                         P->state, P->work, mu, 1, lr, betaL));
    }
    int rc;
    if ((rc = run_balance(P, balance_mask, st))) return rc;
    HIPCHK(hipGetLastError());
    P->p_valid = false;
    return PSGDK_OK;
}

int psgdk_precond_grad(psgdk_plan* plan, int source, void* stream) {
    if (!plan || (source != PSGDK_SRC_EMA && source != PSGDK_SRC_GRAD)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    if (source == PSGDK_SRC_EMA && !plan->use_momentum) return PSGDK_ERR_INVALID;
    ProfCall prof_call(plan, stream);
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!P->hsq_clean || P->clean_stream != st) HIPCHK(hipMemsetAsync(P->work + P->hsumsq_off, 0, (size_t)P->n_tensors * 4, st));
    P->hsq_clean = false;
    P->h_fused = false;
    if ((rc = ensure_P(P, st))) return rc;
    launch_stage(P, P->g_app_a[source], st);
    launch_stage(P, P->g_app_b, st);
    for (const GenDesc& g : P->gd) {
        const TensorDesc& D = P->td[g.tensor];
        DISPATCH_T(P, {
            const T* src = source == PSGDK_SRC_GRAD ? (const T*)(P->work + D.gc_off) : (const T*)(P->state + D.ema_off);
            gen_apply_chain<T>(P, g, src, (T*)(P->work + D.h_off), (float*)(P->work + P->hsumsq_off) + g.tensor, st);
        });
    }
    if (P->n_tiles_diag)
        DISPATCH_T(P, hipLaunchKernelGGL(diag_tensor_kernel<T>, dim3(P->n_tiles_diag), dim3(256), 0, st, P->d_td, P->d_dd,
                                         P->d_tiles_diag, P->state, P->work, 1, source == PSGDK_SRC_GRAD ? 1 : 0,
                                         (float*)(P->work + P->hsumsq_off), P->p_mode() ? 1 : 0));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_apply_update(psgdk_plan* plan, void* const* params, int param_dtype, float lr, float decoupled_wd,
                       float max_avg_amp, float max_elem_amp, void* stream) {
    if (!plan || !params || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    if (plan->h_fused) return PSGDK_ERR_STATE;      // the last h was consumed by psgdk_precond_grad_apply (fused tensors: applied, logical orientation)
    if (!(lr > 0.f) || !(decoupled_wd >= 0.f) || !(max_elem_amp >= max_avg_amp) || !(max_avg_amp > 0.f)) return PSGDK_ERR_INVALID;
    for (int t = 0; t < plan->n_tensors; ++t) if (!params[t]) return PSGDK_ERR_INVALID;
    ProfCall prof_call(plan, stream);
    hipStream_t st = (hipStream_t)stream;
    int rcp;
    void** d_params = nullptr;
    if ((rcp = plan->ptrs_b.get((const void* const*)params, plan->n_tensors, st, &d_params))) return rcp;
    DISPATCH_T(plan, hipLaunchKernelGGL(emit_kernel<T>, dim3(plan->n_tiles_all), dim3(256), 0, st, plan->d_td,
                                        plan->d_tiles_all,
                                        (void* const*)d_params, param_dtype, plan->work,
                                        (const float*)(plan->work + plan->hsumsq_off), 0, 1, lr, decoupled_wd, max_avg_amp, max_elem_amp, (void*)nullptr));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

// precond_grad + apply_update in one call, with the parameter update FUSED into the epilogue of the apply's last product wherever a
// tensor allows it (..._ddp.py:150-157; GemmProblem::upd_p has the arithmetic).  A tensor takes the fused epilogue when its h comes out
// of a grouped-GEMM product (one or two dense factors, <= 2-D), the parameters are fp32, 16-byte aligned, and its logical row length is
// a multiple of 4; every other tensor (1-D, N-D, odd row lengths) is updated by the streaming pass as before, in the same call.  Small
// plans whose products run on the K-split kernel, and bf16 parameters, take the unfused route whole.  The RMS clip is applied
// speculatively (scale 1) and corrected per tensor by clip_fix_kernel when it engages: the corrected parameter differs from the
// two-pass result by at most one fp32 rounding (p' + lr c1 - lr c2 instead of p keep - lr c2).
static int build_fused(psgdk_plan* P, int source, void* const* params) {
    unsigned char* W = P->work;
    P->f_app_a.probs = P->g_app_a[source].probs;
    P->f_app_b.probs = P->g_app_b.probs;
    std::vector<char> fused(P->n_tensors, 0);
    std::vector<FixDesc> fix;
    auto fuse = [&](GemmProblem& g, int t) {
        const TensorDesc& D = P->td[t];
        if (((uintptr_t)params[t] & 15) || (D.lcols & 3) || D.kind == TK_GEN) return;
        g.upd_p = (float*)params[t]; g.upd_ld = D.lcols; g.upd_nr = D.lrows; g.upd_nc = D.lcols;
        if (D.transposed) {      // held transposed: produce h t-major = in the parameter's own orientation ([C][Rp])
            g.Ct = g.C; g.ldct = D.Rp; g.C = nullptr; g.flags |= GF_TMAJOR;
        }
        fused[t] = 1;
        fix.push_back(FixDesc{t, D.lrows, D.lcols, D.transposed ? D.Rp : D.Cp, (long long)D.numel_clip, (unsigned long long)D.h_off, (float*)params[t]});
    };
    // (the tiling of a stage does not depend on the fusion; a stage on the K-split kernel of small launches keeps the two-pass update
    //  for its tensors: that kernel's epilogue does not carry the fused form)
    decide_tiling(P, &P->f_app_a);
    decide_tiling(P, &P->f_app_b);
    if (!P->f_app_a.ksplit)
        for (size_t i = 0; i < P->f_app_a.probs.size(); ++i)
            if (P->td[P->app_a_tensor[i]].kind == TK_M1) fuse(P->f_app_a.probs[i], P->app_a_tensor[i]);
    if (!P->f_app_b.ksplit)
        for (size_t i = 0; i < P->f_app_b.probs.size(); ++i) fuse(P->f_app_b.probs[i], P->app_b_tensor[i]);
    P->fused_any = !fix.empty();
    if (!P->fused_any) return PSGDK_OK;
    int rc;
    if ((rc = finish_stage(P->f_app_a)) || (rc = finish_stage(P->f_app_b))) return rc;
    std::vector<EwTile> rest;
    for (int t = 0; t < P->n_tensors; ++t) {
        if (fused[t]) continue;
        const TensorDesc& D = P->td[t];
        const int th = D.wide ? 16 : 64, tw = D.wide ? 256 : 64;
        for (int tr = 0; tr < (D.R + th - 1) / th; ++tr)
            for (int tc = 0; tc < (D.C + tw - 1) / tw; ++tc) rest.push_back(EwTile{t, tr, tc});
    }
    P->n_tiles_rest = (unsigned)rest.size(); P->n_fix = (unsigned)fix.size();
    if ((rc = upload(&P->d_tiles_rest, rest)) || (rc = upload(&P->d_fix, fix))) return rc;
    (void)W;
    return PSGDK_OK;
}

int psgdk_precond_grad_apply(psgdk_plan* plan, int source, void* const* params, int param_dtype, float lr, float decoupled_wd,
                             float max_avg_amp, float max_elem_amp, void* stream) {
    if (!plan || !params || (source != PSGDK_SRC_EMA && source != PSGDK_SRC_GRAD) || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32))
        return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    if (source == PSGDK_SRC_EMA && !plan->use_momentum) return PSGDK_ERR_INVALID;
    if (!(lr > 0.f) || !(decoupled_wd >= 0.f) || !(max_elem_amp >= max_avg_amp) || !(max_avg_amp > 0.f)) return PSGDK_ERR_INVALID;
    for (int t = 0; t < plan->n_tensors; ++t) if (!params[t]) return PSGDK_ERR_INVALID;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    bool fusable = param_dtype == PSGDK_F32 && P->shards.empty() && !P->no_fuse;
    if (fusable) {
        // the stages carry the callers' parameter pointers: rebuilt when a pointer, the source or the plan's tables change (a synchronous
        // table upload; parameters keep their storage from step to step, so this happens once)
        std::vector<const void*> key(params, params + P->n_tensors);
        key.push_back((const void*)(uintptr_t)(source + 1));
        if (key != P->fused_key && P->fused_rebuilds < 16) {
            // (a caller that hands over fresh pointers every step -- packed shadows of strided parameters -- would pay a synchronous table
            //  upload per step: after 16 rebuilds the plan stays on the two-call route)
            ++P->fused_rebuilds;
            if ((rc = build_fused(P, source, params))) return rc;
            P->fused_key = key;
        }
        fusable = P->fused_any && key == P->fused_key;
    }
    if (!fusable) {
        if ((rc = psgdk_precond_grad(plan, source, stream))) return rc;
        return psgdk_apply_update(plan, params, param_dtype, lr, decoupled_wd, max_avg_amp, max_elem_amp, stream);
    }
    ProfCall prof_call(plan, stream);
    if (!P->hsq_clean || P->clean_stream != st) HIPCHK(hipMemsetAsync(P->work + P->hsumsq_off, 0, (size_t)P->n_tensors * 4, st));
    P->hsq_clean = false;
    if ((rc = ensure_P(P, st))) return rc;
    // stagger: workgroups that walk one tile fewer start up to ~one tile time late (measured on GPT-2-small, K = 768: 4000 ticks of 10 ns:
    // 389 -> 375 us per fused apply; scaled with K for other widths; psgdk_test_fuse_mode overrides)
    int stagger = P->fuse_stagger > 0 ? (P->fuse_stagger & 0xfffff) : P->fuse_stagger;
    if (stagger < 0) {
        int kmax = 0;
        for (const GemmProblem& g : P->f_app_a.probs) kmax = std::max(kmax, g.K);
        stagger = std::min(5 * kmax, 16000);
    }
    const GemmUpdArgs upd{lr, 1.0f - decoupled_wd * lr, max_elem_amp, decoupled_wd != 0.f ? 1 : 0, stagger & 0xffffff, 0};
    launch_stage(P, P->f_app_a, st, &upd);
    launch_stage(P, P->f_app_b, st, &upd);
    P->h_fused = true;
    for (const GenDesc& g : P->gd) {
        const TensorDesc& D = P->td[g.tensor];
        DISPATCH_T(P, {
            const T* src = source == PSGDK_SRC_GRAD ? (const T*)(P->work + D.gc_off) : (const T*)(P->state + D.ema_off);
            gen_apply_chain<T>(P, g, src, (T*)(P->work + D.h_off), (float*)(P->work + P->hsumsq_off) + g.tensor, st);
        });
    }
    if (P->n_tiles_diag)
        DISPATCH_T(P, hipLaunchKernelGGL(diag_tensor_kernel<T>, dim3(P->n_tiles_diag), dim3(256), 0, st, P->d_td, P->d_dd,
                                         P->d_tiles_diag, P->state, P->work, 1, source == PSGDK_SRC_GRAD ? 1 : 0,
                                         (float*)(P->work + P->hsumsq_off), P->p_mode() ? 1 : 0));
    void** d_params = nullptr;
    if ((rc = P->ptrs_b.get((const void* const*)params, P->n_tensors, st, &d_params))) return rc;
    // the tensors the fused epilogue does not cover: clip + update as psgdk_apply_update does it ...
    if (P->n_tiles_rest)
        DISPATCH_T(P, hipLaunchKernelGGL(emit_kernel<T>, dim3(P->n_tiles_rest), dim3(256), 0, st, P->d_td, P->d_tiles_rest,
                                         (void* const*)d_params, param_dtype, P->work, (const float*)(P->work + P->hsumsq_off), 0, 1, lr,
                                         decoupled_wd, max_avg_amp, max_elem_amp, (void*)nullptr));
    // ... and the correction of the fused tensors whose RMS clip engaged (normally none: every workgroup scans the sums and leaves)
    DISPATCH_T(P, hipLaunchKernelGGL(clip_fix_kernel<T>, dim3(std::min(P->n_fix * 8u, 1024u)), dim3(256), 0, st, (const FixDesc*)P->d_fix, (int)P->n_fix,
                                     P->work, (const float*)(P->work + P->hsumsq_off), lr, max_avg_amp, max_elem_amp));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

// ---- the exchange step of the sharded path: parameter update of ALL tensors from the gathered flat h ---------------------
struct psgdk_flat {
    int n = 0;
    FlatDesc* d_fd = nullptr; FlatChunk* d_chunks = nullptr; unsigned n_chunks = 0;
    void** d_ptrs = nullptr;
    std::vector<const void*> h_ptrs;
    // psgdk_flat_set_clip_groups: piece -> clip group (-1: none), the groups, where the members' partial sums lie inside the gathered buffer
    int* d_piece_group = nullptr; FlatGroup* d_groups = nullptr; int n_groups = 0, members = 0; long long member_stride = 0;
    ~psgdk_flat() {
        if (d_fd) (void)hipFree(d_fd); if (d_chunks) (void)hipFree(d_chunks); if (d_ptrs) (void)hipFree(d_ptrs);
        if (d_piece_group) (void)hipFree(d_piece_group); if (d_groups) (void)hipFree(d_groups);
    }
};

int psgdk_flat_create(psgdk_flat** out, int n, const int64_t* numel, const int64_t* h_offset) {
    if (!out || n <= 0 || !numel || !h_offset) return PSGDK_ERR_INVALID;
    std::unique_ptr<psgdk_flat> F(new psgdk_flat());
    F->n = n;
    std::vector<FlatDesc> fd(n);
    std::vector<FlatChunk> ck;
    for (int t = 0; t < n; ++t) {
        if (numel[t] < 0 || h_offset[t] < 0 || numel[t] > 0x7fffffffLL) return PSGDK_ERR_INVALID;
        fd[t] = FlatDesc{(long long)numel[t], (long long)h_offset[t]};
        for (int64_t s0 = 0; s0 < numel[t]; s0 += 2048) ck.push_back(FlatChunk{t, (int)s0});
    }
    F->n_chunks = (unsigned)ck.size();
    int rc;
    if ((rc = upload(&F->d_fd, fd)) || (rc = upload(&F->d_chunks, ck))) return rc;
    HIPCHK(hipMalloc((void**)&F->d_ptrs, (size_t)n * sizeof(void*)));
    *out = F.release();
    return PSGDK_OK;
}

int psgdk_flat_destroy(psgdk_flat* flat) { delete flat; return PSGDK_OK; }

int psgdk_flat_apply(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                     float decoupled_wd, void* stream) {
    if (!flat || !params || !h_flat || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32) ||
        (h_dtype != PSGDK_BF16 && h_dtype != PSGDK_F32) || !(lr > 0.f) || !(decoupled_wd >= 0.f)) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;      // (params[t] == NULL: tensor t is skipped this step)
    int rc;
    if ((rc = upload_ptrs(flat->d_ptrs, flat->h_ptrs, (const void* const*)params, flat->n, st))) return rc;
    if (flat->n_chunks)
        hipLaunchKernelGGL(flat_apply_kernel, dim3(flat->n_chunks), dim3(256), 0, st, flat->d_fd, flat->d_chunks, (void* const*)flat->d_ptrs,
                           param_dtype, h_flat, h_dtype, lr, 1.0f - decoupled_wd * lr, (const float*)nullptr, 1.f, 0.f, 0.f);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_flat_set_clip_groups(psgdk_flat* flat, int n_groups, const int32_t* piece_group, const int64_t* sum_off_bytes,
                               int64_t member_stride_bytes, int members, const int64_t* numel_clip) {
    if (!flat || n_groups < 0) return PSGDK_ERR_INVALID;
    if (flat->d_piece_group) { (void)hipFree(flat->d_piece_group); flat->d_piece_group = nullptr; }
    if (flat->d_groups) { (void)hipFree(flat->d_groups); flat->d_groups = nullptr; }
    flat->n_groups = 0;
    if (n_groups == 0) return PSGDK_OK;
    if (!piece_group || !sum_off_bytes || !numel_clip || members < 1 || member_stride_bytes < 0 || (member_stride_bytes & 3)) return PSGDK_ERR_INVALID;
    std::vector<int> pg(piece_group, piece_group + flat->n);
    std::vector<FlatGroup> gs(n_groups);
    for (int t = 0; t < flat->n; ++t) if (pg[t] < -1 || pg[t] >= n_groups) return PSGDK_ERR_INVALID;
    for (int g = 0; g < n_groups; ++g) {
        if (sum_off_bytes[g] < 0 || (sum_off_bytes[g] & 3) || numel_clip[g] <= 0) return PSGDK_ERR_INVALID;
        gs[g] = FlatGroup{(long long)sum_off_bytes[g], (long long)numel_clip[g]};
    }
    int rc;
    if ((rc = upload(&flat->d_piece_group, pg)) || (rc = upload(&flat->d_groups, gs))) return rc;
    flat->n_groups = n_groups; flat->members = members; flat->member_stride = member_stride_bytes;
    return PSGDK_OK;
}

int psgdk_flat_apply_groups(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                            float decoupled_wd, float max_avg_amp, float max_elem_amp, void* stream) {
    if (!flat || !params || !h_flat || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32) || (h_dtype != PSGDK_BF16 && h_dtype != PSGDK_F32) ||
        !(lr > 0.f) || !(decoupled_wd >= 0.f) || !(max_elem_amp >= max_avg_amp) || !(max_avg_amp > 0.f)) return PSGDK_ERR_INVALID;
    if (!flat->n_groups) return PSGDK_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = upload_ptrs(flat->d_ptrs, flat->h_ptrs, (const void* const*)params, flat->n, st))) return rc;
    if (flat->n_chunks)
        hipLaunchKernelGGL(flat_apply_kernel, dim3(flat->n_chunks), dim3(256), 0, st, flat->d_fd, flat->d_chunks, (void* const*)flat->d_ptrs,
                           param_dtype, h_flat, h_dtype, lr, 1.0f - decoupled_wd * lr, (const float*)nullptr, 1.f, max_avg_amp, max_elem_amp,
                           (const int*)flat->d_piece_group, (const FlatGroup*)flat->d_groups, flat->member_stride, flat->members);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_flat_apply_clipped(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                             const float* h_sumsq_dev, int64_t h_numel, float max_avg_amp, float max_elem_amp, void* stream) {
    if (!flat || !params || !h_flat || !h_sumsq_dev || h_numel <= 0 || (param_dtype != PSGDK_BF16 && param_dtype != PSGDK_F32) ||
        (h_dtype != PSGDK_BF16 && h_dtype != PSGDK_F32) || !(lr > 0.f) || !(max_elem_amp >= max_avg_amp) || !(max_avg_amp > 0.f))
        return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = upload_ptrs(flat->d_ptrs, flat->h_ptrs, (const void* const*)params, flat->n, st))) return rc;
    if (flat->n_chunks)
        hipLaunchKernelGGL(flat_apply_kernel, dim3(flat->n_chunks), dim3(256), 0, st, flat->d_fd, flat->d_chunks, (void* const*)flat->d_ptrs,
                           param_dtype, h_flat, h_dtype, lr, 1.0f, h_sumsq_dev, (float)h_numel, max_avg_amp, max_elem_amp);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_flat_gather(psgdk_flat* flat, const void* const* grads, int grad_dtype, void* g_flat, int flat_dtype, void* m_flat, float beta,
                      float* sum_g4_dev, void* stream) {
    if (!flat || !grads || !g_flat || (grad_dtype != PSGDK_BF16 && grad_dtype != PSGDK_F32) ||
        (flat_dtype != PSGDK_BF16 && flat_dtype != PSGDK_F32) || !(beta >= 0.f && beta < 1.f)) return PSGDK_ERR_INVALID;
    for (int t = 0; t < flat->n; ++t) if (!grads[t]) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = upload_ptrs(flat->d_ptrs, flat->h_ptrs, grads, flat->n, st))) return rc;
    if (sum_g4_dev) HIPCHK(hipMemsetAsync(sum_g4_dev, 0, sizeof(float), st));
    if (flat->n_chunks)
        hipLaunchKernelGGL(flat_gather_kernel, dim3(flat->n_chunks), dim3(256), 0, st, flat->d_fd, flat->d_chunks,
                           (const void* const*)flat->d_ptrs, grad_dtype, g_flat, flat_dtype, m_flat, beta, sum_g4_dev);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}


int psgdk_read_precond_grad(psgdk_plan* plan, int t, void* out, int out_dtype, int clip, float max_avg_amp,
                            float max_elem_amp, void* stream) {
    if (!plan || t < 0 || t >= plan->n_tensors || !out || (out_dtype != PSGDK_BF16 && out_dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    if (plan->h_fused) return PSGDK_ERR_STATE;      // the last h was consumed by psgdk_precond_grad_apply (fused tensors: applied, logical orientation)
    hipStream_t st = (hipStream_t)stream;
    const unsigned b = plan->tile_begin[t], e = plan->tile_begin[t + 1];
    DISPATCH_T(plan, hipLaunchKernelGGL(emit_kernel<T>, dim3(e - b), dim3(256), 0, st, plan->d_td, plan->d_tiles_all + b,
                                        (void* const*)nullptr, out_dtype, plan->work,
                                        (const float*)(plan->work + plan->hsumsq_off), 1, clip, 0.f, 0.f, max_avg_amp, max_elem_amp, out));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_export_precond_grad(psgdk_plan* plan, void* const* outs, int out_dtype, int clip, float max_avg_amp, float max_elem_amp,
                              void* stream) {
    if (!plan || !outs || (out_dtype != PSGDK_BF16 && out_dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    if (plan->h_fused) return PSGDK_ERR_STATE;      // the last h was consumed by psgdk_precond_grad_apply (fused tensors: applied, logical orientation)
    for (int t = 0; t < plan->n_tensors; ++t) if (!outs[t]) return PSGDK_ERR_INVALID;
    ProfCall prof_call(plan, stream);
    hipStream_t st = (hipStream_t)stream;
    int rcp;
    void** d_outs = nullptr;
    if ((rcp = plan->ptrs_b.get((const void* const*)outs, plan->n_tensors, st, &d_outs))) return rcp;
    DISPATCH_T(plan, hipLaunchKernelGGL(emit_kernel<T>, dim3(plan->n_tiles_all), dim3(256), 0, st, plan->d_td, plan->d_tiles_all,
                                        (void* const*)d_outs, out_dtype, plan->work,
                                        (const float*)(plan->work + plan->hsumsq_off), 2, clip, 0.f, 0.f, max_avg_amp, max_elem_amp, (void*)nullptr));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_plan_info(const psgdk_plan* plan, int what, int64_t* value) {
    if (!plan || !value) return PSGDK_ERR_INVALID;
    switch (what) {
        case PSGDK_INFO_NLB_COOP: *value = plan->nlb_coop ? 1 : 0; return PSGDK_OK;
        case PSGDK_INFO_NLB_FALLBACKS: *value = plan->nlb_fallbacks; return PSGDK_OK;
        case PSGDK_INFO_DENSE_FACTORS: *value = (int64_t)plan->dn.size(); return PSGDK_OK;
        case PSGDK_INFO_MAX_DENSE_DIM: *value = plan->max_dp; return PSGDK_OK;
        case PSGDK_INFO_HSUMSQ_OFFSET: *value = (int64_t)plan->hsumsq_off; return PSGDK_OK;
        case PSGDK_INFO_BALNORM_OFFSET: *value = (int64_t)plan->balnorm_off; return PSGDK_OK;
        case PSGDK_INFO_UPDATE_FUSED: *value = plan->h_fused ? (int64_t)plan->n_fix : 0; return PSGDK_OK;
        case PSGDK_INFO_NLB_MEMBER_COLS: *value = !plan->nlb_coop ? 0 : (plan->nlb_small ? 128 : plan->nlb_cols); return PSGDK_OK;
    }
    return PSGDK_ERR_INVALID;
}

int psgdk_profile_enable(psgdk_plan* plan, int enable) {
    if (!plan) return PSGDK_ERR_INVALID;
    plan->prof = (enable & 1) != 0;
    plan->prof_calls = (enable & 2) != 0;      // (enable = 1: the launches only; 3: the calls too)
    return PSGDK_OK;
}

int psgdk_profile_read(psgdk_plan* plan, double* gemm_ms, int64_t* gemm_launches, int reset) {
    if (!plan || !gemm_ms || !gemm_launches) return PSGDK_ERR_INVALID;
    double tot = 0.0;
    for (size_t i = 0; i < plan->prof_used; ++i) {
        HIPCHK(hipEventSynchronize(plan->prof_ev[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, plan->prof_ev[i].first, plan->prof_ev[i].second));
        tot += ms;
    }
    *gemm_ms = tot; *gemm_launches = (int64_t)plan->prof_used;
    if (reset) plan->prof_used = 0;
    return PSGDK_OK;
}

int psgdk_profile_read_fused(psgdk_plan* plan, double* fused_ms, int64_t* fused_launches) {
    if (!plan || !fused_ms || !fused_launches) return PSGDK_ERR_INVALID;
    double tot = 0.0; int64_t n = 0;
    for (size_t i = 0; i < plan->prof_used; ++i) {
        if (!plan->prof_fused[i]) continue;
        HIPCHK(hipEventSynchronize(plan->prof_ev[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, plan->prof_ev[i].first, plan->prof_ev[i].second));
        tot += ms; ++n;
    }
    *fused_ms = tot; *fused_launches = n;
    return PSGDK_OK;
}

int psgdk_profile_read_calls(psgdk_plan* plan, double* call_ms, int64_t* calls, int reset) {
    if (!plan || !call_ms || !calls) return PSGDK_ERR_INVALID;
    double tot = 0.0;
    for (size_t i = 0; i < plan->prof_call_used; ++i) {
        HIPCHK(hipEventSynchronize(plan->prof_call_ev[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, plan->prof_call_ev[i].first, plan->prof_call_ev[i].second));
        tot += ms;
    }
    *call_ms = tot; *calls = (int64_t)plan->prof_call_used;
    if (reset) plan->prof_call_used = 0;
    return PSGDK_OK;
}

int psgdk_fill_normal(void* out, int dtype, int64_t n, uint64_t seed, uint64_t offset, uint32_t stream_id,
                      void* stream) {
    if (!out || n < 0 || (dtype != PSGDK_BF16 && dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (n == 0) return PSGDK_OK;
    const int block = 256;
    const int grid = (int)std::min<int64_t>((n + block - 1) / block, 256 * 8);
    hipLaunchKernelGGL(fill_normal_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, out, dtype, n, seed, offset,
                       stream_id);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_test_dump_noise(psgdk_plan* plan, uint64_t seed, uint64_t offset, void* const* g_out, void* const* spd_out,
                          void* const* skh_out, int pro_iter, void* stream) {
    if (!plan || pro_iter < -1 || pro_iter > 9) return PSGDK_ERR_INVALID;
    if (!plan->state) return PSGDK_ERR_STATE;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    const unsigned F = (unsigned)P->dn.size();
    if (g_out) {
        P->h_dump_g.assign(g_out, g_out + P->n_tensors);
        HIPCHK(hipMemcpyAsync(P->d_noise_g, P->h_dump_g.data(), P->n_tensors * sizeof(void*), hipMemcpyHostToDevice, st));
        DISPATCH_T(P, hipLaunchKernelGGL(dump_g_noise_kernel<T>, dim3(P->n_tiles_all), dim3(256), 0, st, P->d_td, P->d_tiles_all,
                                         (void* const*)P->d_noise_g, seed, offset));
    }
    if (F && (spd_out || skh_out)) {
        std::vector<const void*>& a = P->h_noise_a; std::vector<const void*>& b = P->h_noise_b;
        a.assign(F, nullptr); b.assign(F, nullptr);
        for (unsigned f = 0; f < F; ++f) {
            const int slot = P->dn[f].tensor * PSGDK_GEN_MAXDIM + P->dense_dim[f];
            if (spd_out) a[f] = spd_out[slot];
            if (skh_out) b[f] = skh_out[slot];
        }
        HIPCHK(hipMemcpyAsync(P->d_noise_spd, a.data(), F * sizeof(void*), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(P->d_noise_skh, b.data(), F * sizeof(void*), hipMemcpyHostToDevice, st));
        DISPATCH_T(P, hipLaunchKernelGGL(dump_nlb_noise_kernel<T>, dim3(8, F, 2), dim3(256), 0, st, P->d_dn, (void* const*)P->d_noise_spd,
                                         (void* const*)P->d_noise_skh, seed, offset, pro_iter));
    }
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_test_launch_count(int64_t* count, int reset) {
    if (!count) return PSGDK_ERR_INVALID;
    *count = (int64_t)g_psgdk_launches;
    if (reset) g_psgdk_launches = 0;
    return PSGDK_OK;
}

int psgdk_test_fuse_mode(psgdk_plan* plan, int on, int stagger_ticks) {
    if (!plan || stagger_ticks < -1) return PSGDK_ERR_INVALID;
    plan->no_fuse = !on;
    if (stagger_ticks >= 0) plan->fuse_stagger = stagger_ticks;
    return PSGDK_OK;
}

int psgdk_test_ew_mode(psgdk_plan* plan, int mode) {
    if (!plan || mode < 0 || mode > 3) return PSGDK_ERR_INVALID;
    plan->ew_dbg = mode;
    return PSGDK_OK;
}

int psgdk_test_nlb(psgdk_plan* plan, int chain, int route, uint64_t seed, uint64_t offset, float* out_vsq, void* out_v,
                   int inject_fault, void* stream) {
    if (!plan || (chain != 0 && chain != 1) || (route != 0 && route != 1)) return PSGDK_ERR_INVALID;
    if (!plan->state || plan->dn.empty()) return PSGDK_ERR_STATE;
    if (route == 1 && !plan->d_nlb_jobs) return PSGDK_ERR_STATE;        // the plan has no cooperative launch (wide factors)
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    const unsigned F = (unsigned)P->dn.size();
    P->zero_clean = false;
    // what a real update zeroes before the bound: this chain's row sums and arrival counter
    hipLaunchKernelGGL(test_nlb_reset_kernel, dim3(F), dim3(64), 0, st, P->d_dn, P->work, chain);
    int rc = run_nlb(P, chain, nullptr, seed, offset, 0.1f, 0.9f, 1, -1, st, route, inject_fault ? 4096 : 0);
    if (rc) return rc;
    DISPATCH_T(P, hipLaunchKernelGGL(test_nlb_collect_kernel<T>, dim3(F), dim3(256), 0, st, P->d_dn, P->work, chain, P->max_dp, out_vsq,
                                     (T*)out_v));
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_test_nlb_stamps(psgdk_plan* plan, int chain, uint64_t seed, uint64_t offset, unsigned long long* out, int max_blocks,
                          int* n_blocks, void* stream) {
    if (!plan || (chain != 0 && chain != 1) || !out || !n_blocks || max_blocks <= 0) return PSGDK_ERR_INVALID;
    if (!plan->state || plan->dn.empty() || !plan->d_nlb_jobs || !plan->d_nlb_ts) return PSGDK_ERR_STATE;
    psgdk_plan* P = plan;
    hipStream_t st = (hipStream_t)stream;
    const unsigned F = (unsigned)P->dn.size();
    const size_t words = (size_t)P->n_nlb_jobs * NLB_TS_SLOTS;
    P->zero_clean = false;
    HIPCHK(hipMemsetAsync(P->d_nlb_ts, 0, words * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(test_nlb_reset_kernel, dim3(F), dim3(64), 0, st, P->d_dn, P->work, chain);
    int rc = run_nlb(P, chain, nullptr, seed, offset, 0.1f, 0.9f, 1, -1, st, 1, 0, true);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    const int nb = (int)std::min<size_t>((size_t)max_blocks, (size_t)P->n_nlb_jobs);
    HIPCHK(hipMemcpyAsync(out, P->d_nlb_ts, (size_t)nb * NLB_TS_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *n_blocks = nb;
    return PSGDK_OK;
}

int psgdk_test_gemm_nt(const void* A, const void* B, void* C, void* Ct, int dtype, int M, int N, int K, int lda,
                       int ldb, int ldc, int ldct, int symmetric, void* stream) {
    if (!A || !B || (!C && !Ct) || (M % 64) || (N % 64) || (K % 64) || M <= 0 || N <= 0 || K <= 0)
        return PSGDK_ERR_INVALID;
    if (dtype != PSGDK_BF16 && dtype != PSGDK_F32) return PSGDK_ERR_INVALID;
    Stage s;
    GemmProblem P{};
    P.A = A; P.B = B; P.C = C; P.Ct = Ct;
    P.M = M; P.N = N; P.K = K; P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldct = ldct;
    P.alpha = 1.0f; P.flags = (symmetric & 1) ? GF_SYM : 0;
    if (symmetric & 1) { if (M != N || !C) return PSGDK_ERR_INVALID; P.Ct = C; P.ldct = ldc; }
    if (!C && Ct) P.flags |= GF_TMAJOR;      // as psgdk_plan_bind does for transposed-only outputs
    s.big = (symmetric & 1024) != 0;          // test hook: bit 10 selects the 256x256 tiling, bit 25 the 64 x 64 K-split one
    s.ksplit = (symmetric & (1 << 25)) != 0;
    s.probs.push_back(P);
    int rc = finish_stage(s);
    hipStream_t st = (hipStream_t)stream;
    if (!rc) {
        if (dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st); else launch_stage_t<float>(s, st);
        if (hipGetLastError() != hipSuccess) rc = PSGDK_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) rc = PSGDK_ERR_HIP;
        if (rc == PSGDK_ERR_HIP) g_last_hip_error = (int)hipGetLastError();
    }
    if (s.d_probs) (void)hipFree(s.d_probs);
    if (s.d_tiles) (void)hipFree(s.d_tiles);
    return rc;
}

// experiments: a stage of a bound plan launched `iters` times back to back (hipEvent-timed), as the step launches it or with another
// main loop / launch shape: variant 0 = as bound, 1 = lock-step 256x256, 2 = staggered with one workgroup per tile, 3 = 128x128 tiling
int psgdk_test_stage_bench(psgdk_plan* plan, int which, int variant, int iters, float* avg_ms, void* stream) {
    if (!plan || !plan->state || iters <= 0 || !avg_ms) return PSGDK_ERR_INVALID;
    Stage* list[] = {&plan->g_upd_a, &plan->g_app_a[0], &plan->g_gram, &plan->g_qupd, &plan->g_rq, &plan->g_rrq, &plan->g_P, &plan->g_upd_b, &plan->g_app_b};
    if (which < 0 || which >= (int)(sizeof(list) / sizeof(list[0]))) return PSGDK_ERR_INVALID;
    Stage s = *list[which];              // shallow copy: shares the device tables unless the tiling changes
    Stage alt;
    if (variant == 1) return PSGDK_ERR_UNSUPPORTED;      // (the lock-step 256x256 main loop: removed in round 5)
    if (variant == 2) s.one_per_tile = true;
    if (variant == 3 && s.big) { alt.probs = s.probs; alt.big = false; int rc = finish_stage(alt); if (rc) return rc; s = alt; }
    if (variant == 4 && !s.big) { alt.probs = s.probs; alt.big = true; int rc = finish_stage(alt); if (rc) return rc; s = alt; }
    if (variant == 14) { alt.probs = s.probs; int rc = finish_stage(alt); if (rc) return rc; s = alt; }     // the 128 x 128 tiling, whatever was bound
    if (variant == 56) {      // as bound, output stores register-direct (16 rows x 64 B) instead of staged through the LDS
        alt.probs = s.probs; alt.big = s.big; alt.ksplit = s.ksplit; alt.ext = s.ext;
        for (auto& q : alt.probs) q.flags |= GF_DBG_HALFLINES;
        int rc = finish_stage(alt); if (rc) return rc; s = alt;
    }
    if ((variant == 22 || variant == 24 || variant == 26) && s.big) {      // the 256 x 256 kernel: no epilogue, no stores, register-direct stores
        alt.probs = s.probs; alt.big = true;
        for (auto& q : alt.probs) {
            if (variant == 22) q.flags |= GF_DBG_NOEPI;
            if (variant == 24) q.flags |= GF_DBG_NOSTORE;
            if (variant == 26) q.flags |= GF_DBG_HALFLINES;
        }
        int rc = finish_stage(alt); if (rc) return rc; s = alt;
    }
    if (variant >= 5 && variant <= 8) {      // same tiling, parts of the epilogue's work stripped / the output discarded
        alt.probs = s.probs; alt.big = s.big;
        for (auto& q : alt.probs) {
            if (variant == 5 || variant == 6) { q.row_sumsq = nullptr; q.sumsq = nullptr; }
            if (variant == 6) { q.row_scale = nullptr; q.flags &= ~(GF_SQ_ROWSCALE | GF_RSQRT_ROWSCALE); }
            if (variant == 7) q.flags |= GF_DBG_NOSTORE;
            if (variant == 8) q.flags |= GF_DBG_NOEPI;
        }
        int rc = finish_stage(alt); if (rc) return rc; s = alt;
    }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) { if (plan->dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st); else launch_stage_t<float>(s, st); }
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) { if (plan->dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st); else launch_stage_t<float>(s, st); }
    (void)hipEventRecord(e1, st);
    int rc = PSGDK_OK;
    if (hipEventSynchronize(e1) != hipSuccess) rc = PSGDK_ERR_HIP;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (alt.d_probs) (void)hipFree(alt.d_probs);
    if (alt.d_tiles) (void)hipFree(alt.d_tiles);
    return rc;
}

// experiments: ONE asynchronous launch of a one-problem stage (tables cached per argument set, never freed: tools only)
int psgdk_test_gemm_launch(const void* A, const void* B, void* C, void* Ct, int dtype, int M, int N, int K, int flags, void* stream) {
    struct Key { const void* a; const void* b; void* c; void* ct; int dtype, M, N, K, flags; };
    static std::vector<std::pair<Key, Stage*>> cache;
    Stage* s = nullptr;
    for (auto& e : cache)
        if (e.first.a == A && e.first.b == B && e.first.c == C && e.first.ct == Ct && e.first.dtype == dtype && e.first.M == M &&
            e.first.N == N && e.first.K == K && e.first.flags == flags) s = e.second;
    if (!s) {
        s = new Stage();
        GemmProblem P{};
        P.A = A; P.B = B; P.C = C; P.Ct = Ct; P.M = M; P.N = N; P.K = K; P.lda = K; P.ldb = K; P.ldc = N; P.ldct = M; P.alpha = 1.f;
        P.flags = flags & ~(1024 | 2048 | 16384 | (3 << 24));
        if (!C && Ct) P.flags |= GF_TMAJOR;
        s->probs.push_back(P);
        s->big = (flags & 1024) != 0; s->one_per_tile = (flags & 16384) != 0;
        s->ksplit = (flags & (1 << 25)) != 0;
        int rc = finish_stage(*s);
        if (rc) return rc;
        cache.push_back({Key{A, B, C, Ct, dtype, M, N, K, flags}, s});
    }
    if (dtype == PSGDK_BF16) launch_stage_t<bf16_t>(*s, (hipStream_t)stream); else launch_stage_t<float>(*s, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_test_gemm_bench(const void* A, const void* B, void* C, void* Ct, int dtype, int M, int N, int K, int batch,
                          int symmetric, int iters, float* avg_ms, void* stream) {
    if (!A || !B || (!C && !Ct) || (M % 64) || (N % 64) || (K % 64) || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || iters <= 0 || !avg_ms)
        return PSGDK_ERR_INVALID;
    const size_t esz = dtype == PSGDK_BF16 ? 2 : 4;
    Stage s;
    for (int b = 0; b < batch; ++b) {
        GemmProblem P{};
        P.A = (const unsigned char*)A + (size_t)b * M * K * esz;
        P.B = (const unsigned char*)B + (size_t)b * N * K * esz;
        P.C = C ? (unsigned char*)C + (size_t)b * M * N * esz : nullptr;
        P.Ct = Ct ? (unsigned char*)Ct + (size_t)b * M * N * esz : nullptr;
        P.M = M; P.N = N; P.K = K; P.lda = K; P.ldb = K; P.ldc = N; P.ldct = M;
        P.alpha = 1.0f; P.flags = (symmetric & 1 ? GF_SYM : 0) | (symmetric & ~1);
        if (symmetric & 1) { P.Ct = P.C; P.ldct = P.ldc; }
        if (!P.C && P.Ct) P.flags |= GF_TMAJOR;
        s.probs.push_back(P);
    }
    s.big = (symmetric & 1024) != 0;
    s.one_per_tile = (symmetric & 16384) != 0;
    s.ksplit = (symmetric & (1 << 25)) != 0;
    for (auto& q : s.probs) q.flags &= ~(1024 | 2048 | 16384 | (63 << 24) | (1 << 23));
    int rc = finish_stage(s);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (!rc) {
        for (int i = 0; i < 3; ++i) { if (dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st); else launch_stage_t<float>(s, st); }
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) { if (dtype == PSGDK_BF16) launch_stage_t<bf16_t>(s, st); else launch_stage_t<float>(s, st); }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) rc = PSGDK_ERR_HIP;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (s.d_probs) (void)hipFree(s.d_probs);
    if (s.d_tiles) (void)hipFree(s.d_tiles);
    return rc;
}

int psgdk_test_tile_queues(int n, const int32_t* M, const int32_t* N, const int32_t* K, const int32_t* sym, int big,
                           int64_t* queue_tiles, int64_t* queue_cost, int64_t* table_len) {
    if (n <= 0 || !M || !N || !K || !queue_tiles || !queue_cost || !table_len) return PSGDK_ERR_INVALID;
    TileTableBuilder tb;
    tb.bm = big ? GEMM_BIG_BM : GEMM_BM;
    tb.bn = big ? GEMM_BIG_BN : GEMM_BN;
    std::vector<GemmProblem> probs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (M[i] <= 0 || N[i] <= 0 || K[i] <= 0) return PSGDK_ERR_INVALID;
        probs[i].M = M[i]; probs[i].N = N[i]; probs[i].K = K[i];
        probs[i].flags = (sym && sym[i]) ? GF_SYM : 0;
        tb.add_problem(i, probs[i]);
    }
    const std::vector<GemmTile> table = tb.finish();
    *table_len = (int64_t)table.size();
    for (int x = 0; x < 8; ++x) { queue_tiles[x] = 0; queue_cost[x] = 0; }
    // read the queues back from the table the kernel would see: entry b belongs to XCD b % 8
    for (size_t b = 0; b < table.size(); ++b) {
        const GemmTile& t = table[b];
        if (t.prob < 0) continue;
        if (t.prob >= n) return PSGDK_ERR_STATE;
        queue_tiles[b % 8] += 1;
        queue_cost[b % 8] += probs[t.prob].K;
    }
    return PSGDK_OK;
}

int psgdk_test_trsm_right(const void* Y, const void* U, const void* Ut, void* out_nat, void* out_t, int dtype, int rows, int d,
                          void* stream) {
    if (Ut && (dtype != PSGDK_BF16 || round_up64(d) > EQ_TRSM_BF16_MAX_DP)) return PSGDK_ERR_INVALID;
    if (!Y || !U || (!out_nat && !out_t) || rows <= 0 || d <= 0 || (dtype != PSGDK_BF16 && dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int dp = (int)round_up64(d), rp = (int)round_up64(rows);
    float* uinv = nullptr; UinvJob* dj = nullptr; TrsmJob* tj = nullptr; TrsmTile* tt = nullptr;
    HIPCHK(hipMalloc((void**)&uinv, (size_t)dp * 64 * 4));
    std::vector<UinvJob> uj = {UinvJob{U, uinv, d, dp}};
    TrsmJob j{};
    j.in = Y; j.ld_in = dp; j.U = U; j.Ut = Ut; j.uinv = uinv; j.out_nat = out_nat; j.ld_nat = dp; j.out_t = out_t; j.ld_t = rp; j.rows = rows; j.dp = dp;
    std::vector<TrsmJob> jobs = {j};
    std::vector<TrsmTile> tiles;
    for (int p = 0; p < rp / 64; ++p) tiles.push_back(TrsmTile{0, p});
    int rc;
    if ((rc = upload(&dj, uj)) || (rc = upload(&tj, jobs)) || (rc = upload(&tt, tiles))) return rc;
    if (dtype == PSGDK_BF16) {
        hipLaunchKernelGGL(eq_uinv_kernel<bf16_t>, dim3(dp / 64, 1), dim3(64), 0, st, dj);
        if (Ut) { if ((rc = launch_trsm_bf16(tj, tt, (unsigned)tiles.size(), dp, st))) return rc; }
        else hipLaunchKernelGGL(eq_trsm_kernel<bf16_t>, dim3((unsigned)tiles.size()), dim3(256), 0, st, tj, tt);
    } else {
        hipLaunchKernelGGL(eq_uinv_kernel<float>, dim3(dp / 64, 1), dim3(64), 0, st, dj);
        hipLaunchKernelGGL(eq_trsm_kernel<float>, dim3((unsigned)tiles.size()), dim3(256), 0, st, tj, tt);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(uinv); (void)hipFree(dj); (void)hipFree(tj); (void)hipFree(tt);
    return PSGDK_OK;
}

// timing of the two solve kernels alone (tools/trsm_bench.py): `iters` launches between two events; dbg: see eq_trsm_bf16_kernel
int psgdk_test_trsm_bench(const void* Y, const void* U, const void* Ut, void* out_nat, void* out_t, int rows, int d, int iters, int dbg,
                          float* avg_ms, void* stamps, void* stream) {
    if (!Y || !U || (!out_nat && !out_t) || rows <= 0 || d <= 0 || iters <= 0 || !avg_ms) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int dp = (int)round_up64(d), rp = (int)round_up64(rows);
    if (Ut && dp > EQ_TRSM_BF16_MAX_DP) return PSGDK_ERR_INVALID;
    float* uinv = nullptr; UinvJob* dj = nullptr; TrsmJob* tj = nullptr; TrsmTile* tt = nullptr;
    HIPCHK(hipMalloc((void**)&uinv, (size_t)dp * 64 * 4));
    std::vector<UinvJob> uj = {UinvJob{U, uinv, d, dp}};
    TrsmJob j{};
    j.in = Y; j.ld_in = dp; j.U = U; j.Ut = Ut; j.uinv = uinv; j.out_nat = out_nat; j.ld_nat = dp; j.out_t = out_t; j.ld_t = rp; j.rows = rows; j.dp = dp;
    if (dbg & 16) { if (!stamps) return PSGDK_ERR_INVALID; j.row_ss = (float*)stamps; }
    std::vector<TrsmJob> jobs = {j};
    std::vector<TrsmTile> tiles;
    for (int p = 0; p < rp / 64; ++p) tiles.push_back(TrsmTile{0, p});
    int rc;
    if ((rc = upload(&dj, uj)) || (rc = upload(&tj, jobs)) || (rc = upload(&tt, tiles))) return rc;
    hipLaunchKernelGGL(eq_uinv_kernel<bf16_t>, dim3(dp / 64, 1), dim3(64), 0, st, dj);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int it = -2; it < iters; ++it) {
        if (it == 0) HIPCHK(hipEventRecord(e0, st));
        if (Ut) { if ((rc = launch_trsm_bf16(tj, tt, (unsigned)tiles.size(), dp, st, dbg & 23, (dbg & 8) ? 1 : 0))) return rc; }
        else hipLaunchKernelGGL(eq_trsm_kernel<bf16_t>, dim3((unsigned)tiles.size()), dim3(256), 0, st, tj, tt);
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(uinv); (void)hipFree(dj); (void)hipFree(tj); (void)hipFree(tt);
    return PSGDK_OK;
}


}  // extern "C"

// ===================================================================================================================
// LRA preconditioner (psgd.py:987-1072)
// ===================================================================================================================
struct psgdk_lra {
    int64_t N = 0; int r = 0; int dtype = 0; size_t esz = 4;
    void* U = nullptr; void* V = nullptr; void* d = nullptr; float* Luvd = nullptr;
    unsigned char* work = nullptr;
    size_t sm_off = 0, v_off = 0, h_off = 0, qh_off = 0, iq_off = 0, diff_off = 0, y_off = 0, work_bytes = 0;
    int64_t row0 = 0;          // psgdk_lra_set_row_shard: this object's rows are rows [row0, row0 + N) of a larger, row-sharded preconditioner
    bool sharded = false;
    // Gram recurrence (psgdk_lra_set_gram_recurrence; lra_gram_recur_kernel): the Grams of the current factors, carried from update to update
    float* gst = nullptr;      // device: [UtU | VtV | VtU | rotated VtU of the update in flight], 4 x RM x RM fp32
    int gram_every = 0;        // 0 = off: every update reads the factors for its Grams (psgd.py:1006 as written)
    int gram_age = -1;         // updates since the Grams in gst were last read from the factors; -1 = gst is not valid
    int64_t pk_rows = 0;       // rows the packed two-rows-per-thread kernels (kernels_lra_pk.hiph) covered in the last update / apply call (0: none)
    ~psgdk_lra() { if (gst) (void)hipFree(gst); }
};

// The launches of kernels_lra_pk.hiph (bf16 factors of even rank <= 16, one thread per row pair): the packed kernels over the first
// floor(N / 512) * 512 rows, the one-row kernels of kernels_lra.hiph over the tail (their partial sums meet in the same scratch slots).
// false = not this instantiation: the caller goes on to the one-row kernels for everything.
// (the packed kernels are instantiated for the even rank itself, RE: every request of a block is then a whole access of every thread)
#define LRAPK_RE(r_, ...) do { switch (r_) { case 2: { constexpr int RE = 2; __VA_ARGS__; } break; case 4: { constexpr int RE = 4; __VA_ARGS__; } break; \
                                             case 6: { constexpr int RE = 6; __VA_ARGS__; } break; case 8: { constexpr int RE = 8; __VA_ARGS__; } break;  \
                                             case 10: { constexpr int RE = 10; __VA_ARGS__; } break; case 12: { constexpr int RE = 12; __VA_ARGS__; } break; \
                                             case 14: { constexpr int RE = 14; __VA_ARGS__; } break; default: { constexpr int RE = 16; __VA_ARGS__; } break; } } while (0)
struct LrapkGeom { unsigned gb1, gb2, gbr, shm1, shm2, shmr;        // packed kernels: grids / dynamic LDS for one factor, two, the rotation
                   unsigned tb1, tb2, tbr, tshm1, tshm2, tshmr;     // tail, one-row kernels
                   int64_t nf; };                                   // rows the packed kernels own
template <typename T, int TPR, int RC>
static bool lrapk_update_phase(psgdk_lra* L, int phase, const LrapkGeom& G, T* U, T* V, T* d, const LraVH<T>& vh, T* Qh, T* iq, T* diff, float* sm,
                               unsigned shm_s1, bool recur, bool grams_carried, int update_u, float lr, float betaL, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && TPR == 1) {
        const int64_t N = L->N, nf = G.nf, nt = N - nf; const int r = L->r;
        LraVH<T> vt = vh;           // the tail's rows: the same vectors from row nf on (Philox counters of the WHOLE vector)
        vt.g = vh.g + nf; vt.noise = vh.noise ? vh.noise + nf : nullptr; vt.row0 = vh.row0 + nf;
        T* Ut = U + nf * r; T* Vt = V + nf * r;
        switch (phase) {
        case 1:
            hipLaunchKernelGGL((lra_small1_kernel<T, TPR>), dim3(1), dim3(256), shm_s1, st, sm, r, recur ? L->gst + 3 * LraCfg<TPR>::MS : (float*)nullptr);
            LRAPK_RE(r, hipLaunchKernelGGL((lrapk_rotate_kernel<RE>), dim3(G.gbr), dim3(LRA_THREADS), G.shmr, st, U, V, (const T*)d, vh, nf, r, sm));
            if (nt) hipLaunchKernelGGL((lra_rotate_kernel<T, TPR, RC>), dim3(G.tbr), dim3(LRA_THREADS), G.tshmr, st, Ut, Vt, (const T*)(d + nf), vt, nt, r, sm);
            break;
        case 2:
            hipLaunchKernelGGL((lra_small2_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, r);
            LRAPK_RE(r, hipLaunchKernelGGL((lrapk_pass3_kernel<RE>), dim3(G.gb2), dim3(LRA_THREADS), G.shm2, st, (const T*)U, (const T*)V, (const T*)d, vh, Qh, iq, nf, r, sm));
            if (nt) hipLaunchKernelGGL((lra_pass3_kernel<T, TPR, RC>), dim3(G.tb2), dim3(LRA_THREADS), G.tshm2, st, (const T*)Ut, (const T*)Vt, (const T*)(d + nf), vt,
                                       Qh + nf, iq + nf, nt, r, sm);
            break;
        case 3:
            hipLaunchKernelGGL((lra_small3_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, r);
            LRAPK_RE(r, hipLaunchKernelGGL((lrapk_pass4_kernel<RE>), dim3(G.gb2), dim3(LRA_THREADS), G.shm2, st, (const T*)U, (const T*)V, (const T*)d, vh, (const T*)Qh,
                               (const T*)iq, diff, nf, r, sm));
            if (nt) hipLaunchKernelGGL((lra_pass4_kernel<T, TPR, RC>), dim3(G.tb2), dim3(LRA_THREADS), G.tshm2, st, (const T*)Ut, (const T*)Vt, (const T*)(d + nf), vt,
                                       (const T*)(Qh + nf), (const T*)(iq + nf), diff + nf, nt, r, sm);
            break;
        case 4:
            hipLaunchKernelGGL((lra_small4_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, L->Luvd, r, update_u ? 1 : 0, lr, betaL);
            LRAPK_RE(r, hipLaunchKernelGGL((lrapk_pass5_kernel<RE>), dim3(G.gb1), dim3(LRA_THREADS), G.shm1, st, U, V, d, (const T*)Qh, (const T*)iq, (const T*)diff, nf, r,
                               update_u ? 1 : 0, (const float*)sm));
            if (nt) hipLaunchKernelGGL((lra_pass5_kernel<T, TPR, RC>), dim3(G.tb1), dim3(LRA_THREADS), G.tshm1, st, Ut, Vt, d + nf, (const T*)(Qh + nf),
                                       (const T*)(iq + nf), (const T*)(diff + nf), nt, r, update_u ? 1 : 0, (const float*)sm);
            if (recur) {      // the Grams of the factors as pass 5 leaves them, for the next update
                hipLaunchKernelGGL((lra_gram_recur_kernel<T, TPR>), dim3(1), dim3(256), 0, st, (const float*)sm, (const float*)(L->gst + 3 * LraCfg<TPR>::MS),
                                   L->gst, r, update_u ? 1 : 0);
                L->gram_age = grams_carried ? L->gram_age + 1 : 1;
            }
            break;
        default:
            return false;      // (phase 0, the Grams, keeps its kernel: it runs once in `gram_every` updates)
        }
        return true;
    } else {
        return false;
    }
}
template <typename T, int TPR, int RC>
static bool lrapk_apply_phase(psgdk_lra* L, int phase, const LrapkGeom& G, const T* g, T* y, T* out, float* sm, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && TPR == 1) {
        const int64_t nf = G.nf, nt = L->N - nf; const int r = L->r;
        const T* U = (const T*)L->U; const T* V = (const T*)L->V; const T* d = (const T*)L->d;
        LRAPK_RE(r, hipLaunchKernelGGL((lrapk_apply_kernel<RE>), dim3(G.gb1), dim3(LRA_THREADS), G.shm1, st, U, V, d, g, y, out, nf, r, phase, sm));
        if (nt) hipLaunchKernelGGL((lra_apply_kernel<T, TPR, RC>), dim3(G.tb1), dim3(LRA_THREADS), G.tshm1, st, U + nf * r, V + nf * r, d + nf, g + nf, y + nf,
                                   out + nf, nt, r, phase, sm);
        return true;
    } else {
        return false;
    }
}

extern "C" {

static int lra_sm_total(int tpr) { return tpr == 1 ? LraCfg<1>::TOTAL : (tpr == 2 ? LraCfg<2>::TOTAL : LraCfg<4>::TOTAL); }

int psgdk_lra_create(psgdk_lra** out, int64_t N, int r, int dtype) {
    if (!out || N <= 0 || r < 0 || (r > 0 && r >= N) || (dtype != PSGDK_BF16 && dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    // r <= 64: the three tuned rank classes; above: the general path (kernels_lra_gen.hiph), whose lanes keep 16 columns each
    if (r > LRAG_RMAX) return PSGDK_ERR_UNSUPPORTED;       // (valid upstream -- any rank -- but an N x 1024 factor pair is 8 KB per element)
    psgdk_lra* L = new psgdk_lra();
    L->N = N; L->r = r; L->dtype = dtype; L->esz = dtype == PSGDK_BF16 ? 2 : 4;
    size_t wo = 0;
    L->sm_off = wo; wo += align256((size_t)(r > LRA_RMAX ? lrag_layout(r).TOTAL : lra_sm_total(lra_tpr_of_rank(r))) * 4);
    const size_t nb = align256((size_t)N * L->esz);
    L->v_off = L->h_off = 0;      // (v and h are no longer materialised: LraVH)
    L->qh_off = wo; wo += nb; L->iq_off = wo; wo += nb;
    L->diff_off = wo; wo += nb; L->y_off = wo; wo += nb;
    L->work_bytes = wo;
    *out = L;
    return PSGDK_OK;
}

int psgdk_lra_destroy(psgdk_lra* lra) { delete lra; return PSGDK_OK; }

int psgdk_lra_work_bytes(const psgdk_lra* lra, size_t* work_bytes) {
    if (!lra || !work_bytes) return PSGDK_ERR_INVALID;
    *work_bytes = lra->work_bytes;
    return PSGDK_OK;
}

int psgdk_lra_bind(psgdk_lra* lra, void* U, void* V, void* d, float* Luvd, void* work) {
    if (!lra || !d || !Luvd || !work || (lra->r > 0 && (!U || !V))) return PSGDK_ERR_INVALID;
    lra->U = U; lra->V = V; lra->d = d; lra->Luvd = Luvd; lra->work = (unsigned char*)work;
    lra->gram_age = -1;
    return PSGDK_OK;
}
int psgdk_lra_set_gram_recurrence(psgdk_lra* lra, int every) {
    if (!lra || every < 0) return PSGDK_ERR_INVALID;
    if (every > 0 && lra->r > LRA_RMAX) return PSGDK_ERR_UNSUPPORTED;      // (the tuned rank classes only)
    if (every > 0 && lra->r > 0 && !lra->gst) {
        const size_t rm = (size_t)LRA_CB * lra_tpr_of_rank(lra->r);
        HIPCHK(hipMalloc((void**)&lra->gst, 4 * rm * rm * sizeof(float)));
    }
    lra->gram_every = every;
    lra->gram_age = -1;
    return PSGDK_OK;
}
int psgdk_lra_state_changed(psgdk_lra* lra) {
    if (!lra) return PSGDK_ERR_INVALID;
    lra->gram_age = -1;
    return PSGDK_OK;
}

// element type x rank class (threads per row: 1, 2, 4 for ranks up to 16, 32, 64)
// (RC: the columns a thread's register arrays hold -- r rounded up to a multiple of 4 in the one-thread-per-row class, so that the reference's
//  default rank 10 carries 12-wide arrays instead of 16-wide ones: the row passes are bound by what their registers let them keep in flight)
#define LRA_TPR_(tpr_, rc_, ...) do { if ((tpr_) == 1) { constexpr int TPR = 1;                                                                  \
                                          if ((rc_) <= 4) { constexpr int RC = 4; __VA_ARGS__; } else if ((rc_) <= 8) { constexpr int RC = 8; __VA_ARGS__; } \
                                          else if ((rc_) <= 12) { constexpr int RC = 12; __VA_ARGS__; } else { constexpr int RC = 16; __VA_ARGS__; } }       \
                                      else if ((tpr_) == 2) { constexpr int TPR = 2; constexpr int RC = LRA_CB; __VA_ARGS__; }                     \
                                      else { constexpr int TPR = 4; constexpr int RC = LRA_CB; __VA_ARGS__; } } while (0)
#define LRA_T(L, ...) do { const int tpr_ = lra_tpr_of_rank((L)->r); const int rc_ = (L)->r;                      \
                           if ((L)->dtype == PSGDK_BF16) { typedef bf16_t T; LRA_TPR_(tpr_, rc_, __VA_ARGS__); }  \
                           else { typedef float T; LRA_TPR_(tpr_, rc_, __VA_ARGS__); } } while (0)

// launch geometry of the LRA row passes: dynamic LDS for `mats` row buffers of (256 / tpr) x r floats (+ `fixed` bytes of
// static LDS), and as many workgroups as are resident at once (grid-stride loops; <= 8 workgroups of 4 waves per CU)
static void lra_geometry(int64_t N, int r, int mats, unsigned fixed, unsigned* grid, unsigned* shm) {
    const int rows = LRA_ROWS / lra_tpr_of_rank(r);
    const unsigned bytes = (unsigned)std::max(16, mats * rows * lra_lds_stride(r) * 4);      // (padded row stride for r > 16)
    const unsigned per_cu = std::max(1u, std::min(8u, (160u * 1024u) / (bytes + fixed + 1024u)));
    const int64_t blocks = (N + rows - 1) / rows;
    *grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * (int64_t)per_cu));
    *shm = bytes;
}

// the packed two-rows-per-thread passes (kernels_lra_pk.hiph): bf16 factors, one thread per row pair, 512 rows per workgroup and iteration
static void lrapk_geometry(int64_t N, int r, int mats, unsigned fixed, unsigned* grid, unsigned* shm) {
    const unsigned bytes = (unsigned)std::max(16, mats * LRAPK_ROWS * r * 2);
    const unsigned per_cu = std::max(1u, std::min(8u, (160u * 1024u) / (bytes + fixed + 1024u)));
    const int64_t blocks = (N + LRAPK_ROWS - 1) / LRAPK_ROWS;
    *grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * (int64_t)per_cu));
    *shm = bytes;
}
// ... taken when every N-vector the passes touch can be read two elements at a time (PSGDK_LRA_PK=0: the one-row kernels, for A/B runs and tests)
static void lra_geometry(int64_t N, int r, int mats, unsigned fixed, unsigned* grid, unsigned* shm);
static LrapkGeom lrapk_geom(int64_t N, int r) {
    LrapkGeom G{};
    G.nf = (N / LRAPK_ROWS) * LRAPK_ROWS;
    const int64_t nt = N - G.nf;
    const unsigned rot_fixed = 2u * 16u * 16u * 4u;
    lrapk_geometry(G.nf, r, 1, 0, &G.gb1, &G.shm1); lrapk_geometry(G.nf, r, 2, 0, &G.gb2, &G.shm2); lrapk_geometry(G.nf, r, 2, rot_fixed, &G.gbr, &G.shmr);
    if (nt) { lra_geometry(nt, r, 1, 0, &G.tb1, &G.tshm1); lra_geometry(nt, r, 2, 0, &G.tb2, &G.tshm2); lra_geometry(nt, r, 2, rot_fixed, &G.tbr, &G.tshmr); }
    return G;
}
static bool lrapk_ok(const psgdk_lra* L, const void* a, const void* b, const void* c) {
    if (L->dtype != PSGDK_BF16 || L->r < 2 || L->r > 16 || (L->r & 1) || L->N < LRAPK_ROWS) return false;
    const char* e = getenv("PSGDK_LRA_PK");
    if (e && e[0] == '0') return false;
    const uintptr_t vec = (uintptr_t)L->d | (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)L->work;
    const uintptr_t mat = (uintptr_t)L->U | (uintptr_t)L->V;
    return (vec & 3) == 0 && (mat & 15) == 0;
}

// ranks above 64: the general path (kernels_lra_gen.hiph), stage for stage the tuned one
// phase: -1 = the whole update; 0 .. 4 = the phases of a row-sharded preconditioner (round 6: the general path cut where a reduction over
// ALL rows is complete, like the tuned rank classes -- psgdk_lra_phase_segments names the words to reduce over the ranks in between):
//   0: clear, Grams U^T U, V^T V   1: small1, rotation (+ V^T (d h), U^T (v / d)), Grams of the rotated factors   2: small2, pass 3
//   3: small3, pass 4 (maxima)      4: small4, pass 5
static int lra_update_general(psgdk_lra* L, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u, float lr,
                              float betaL, float damping, hipStream_t st, int phase = -1) {
    const int64_t N = L->N; const int r = L->r;
    const LragLayout Y = lrag_layout(r);
    float* sm = (float*)(L->work + L->sm_off);
    const auto on = [phase](int p) { return phase < 0 || phase == p; };
    if (on(0)) HIPCHK(hipMemsetAsync(sm, 0, (size_t)(Y.SC + 7) * 4, st));       // everything but the sum of h^2 of the last apply
    const unsigned tiles = (unsigned)((r + 15) / 16);
    const unsigned slices = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4096 / ((int64_t)tiles * tiles) + 1, 1024), (N + 255) / 256));
    const unsigned gw = (unsigned)std::max<int64_t>(1, std::min<int64_t>((N + 3) / 4, 2048));
    const unsigned shm_rot = (unsigned)(4 * 2 * r * sizeof(float));
#define LRAG_T(...) do { if (L->dtype == PSGDK_BF16) { typedef bf16_t T; __VA_ARGS__; } else { typedef float T; __VA_ARGS__; } } while (0)
    LRAG_T({
        T* Qh = (T*)(L->work + L->qh_off); T* iq = (T*)(L->work + L->iq_off); T* diff = (T*)(L->work + L->diff_off);
        T* U = (T*)L->U; T* V = (T*)L->V; T* d = (T*)L->d;
        const LraVH<T> vh{(const T*)g, (const T*)v_noise, damping, seed, offset, L->row0};
        if (on(0))
            hipLaunchKernelGGL(lrag_gram_kernel<T>, dim3(tiles, tiles, slices), dim3(256), 0, st, (const T*)U, (const T*)V, N, r, sm + Y.UTU, sm + Y.VTV,
                               (float*)nullptr, 3);
        if (on(1)) {
            hipLaunchKernelGGL(lrag_small1_kernel<T>, dim3(1), dim3(256), 0, st, sm, Y);
            if (shm_rot > 64u * 1024u)
                HIPCHK(hipFuncSetAttribute((const void*)lrag_rotate_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_rot));
            hipLaunchKernelGGL(lrag_rotate_kernel<T>, dim3(gw), dim3(256), shm_rot, st, U, V, (const T*)d, vh, N, sm, Y);
            hipLaunchKernelGGL(lrag_gram_kernel<T>, dim3(tiles, tiles, slices), dim3(256), 0, st, (const T*)U, (const T*)V, N, r, sm + Y.UTU2, sm + Y.VTV2,
                               sm + Y.VTU, 7);
        }
        if (on(2)) {
            hipLaunchKernelGGL(lrag_small2_kernel<T>, dim3(1), dim3(256), 0, st, sm, Y);
            hipLaunchKernelGGL(lrag_pass3_kernel<T>, dim3(gw), dim3(256), 0, st, (const T*)U, (const T*)V, (const T*)d, vh, Qh, iq, N, sm, Y);
        }
        if (on(3)) {
            hipLaunchKernelGGL(lrag_small3_kernel<T>, dim3(1), dim3(64), 0, st, sm, Y);
            hipLaunchKernelGGL(lrag_pass4_kernel<T>, dim3(gw), dim3(256), 0, st, (const T*)U, (const T*)V, (const T*)d, vh, (const T*)Qh, (const T*)iq,
                               diff, N, sm, Y);
        }
        if (on(4)) {
            hipLaunchKernelGGL(lrag_small4_kernel<T>, dim3(1), dim3(64), 0, st, sm, L->Luvd, Y, update_u ? 1 : 0, lr, betaL);
            hipLaunchKernelGGL(lrag_pass5_kernel<T>, dim3(gw), dim3(256), 0, st, U, V, d, (const T*)Qh, (const T*)iq, (const T*)diff, N,
                               update_u ? 1 : 0, (const float*)sm, Y);
        }
    });
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}
static int lra_apply_general(psgdk_lra* L, const void* g, void* out, hipStream_t st, int phase = -1) {      // phase: -1 = all three stages, 0 .. 2 = one
    const int64_t N = L->N; const int r = L->r;
    const LragLayout Y = lrag_layout(r);
    float* sm = (float*)(L->work + L->sm_off);
    if (phase <= 0) {
        HIPCHK(hipMemsetAsync(sm + Y.VTX2, 0, (size_t)(2 * r) * 4, st));
        HIPCHK(hipMemsetAsync(sm + Y.SC + 7, 0, 4, st));
    }
    const unsigned gw = (unsigned)std::max<int64_t>(1, std::min<int64_t>((N + 3) / 4, 2048));
    LRAG_T({
        T* y = (T*)(L->work + L->y_off);
        for (int stage = 0; stage < 3; ++stage)
            if (phase < 0 || phase == stage)
                hipLaunchKernelGGL(lrag_apply_kernel<T>, dim3(gw), dim3(256), 0, st, (const T*)L->U, (const T*)L->V, (const T*)L->d, (const T*)g, y,
                                   (T*)out, N, stage, sm, Y);
    });
#undef LRAG_T
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

// The update as five phases and the apply as three: a phase ends where a reduction over ALL rows is complete in the scratch block `sm`
// (the row passes add their partial sums / maxima there with atomics; the one-workgroup `small` kernels that consume them open the next
// phase).  One GPU runs the phases back to back (psgdk_lra_update_whiten, psgdk_lra_precond_grad: the launches of rounds 1 - 4, in their
// order); a row-sharded preconditioner (psgdk_lra_set_row_shard) reduces the phase's slots over the ranks in between.
//   update phase 0: clear, Grams (psgd.py:1006)                      -> UTU, VTV, VTU                     (sum)
//                1: small1 (rotation, :1007-1015), rotate             -> V^T (d h), U^T (v / d)            (sum)
//                2: small2 (LU, first solve, :1020-1024), pass 3      -> a^T U, b^T U, a^T V, b^T V, |a|^2, |b|^2   (sum)
//                3: small3 (second solve, :1025), pass 4              -> max |Ph h|, max |v invPv|         (max)
//                4: small4 (Lipschitz constants, coefficients, :1031-1052), pass 5 (the rank-1 updates)
//   apply phase 0: clear, V^T (d g);  1: y, U^T y;  2: out, sum of out^2      (sum each)
static int lra_update_phase_t(psgdk_lra* L, int phase, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u,
                              float lr, float betaL, float damping, hipStream_t st) {
    float* sm = (float*)(L->work + L->sm_off);
    const int64_t N = L->N; const int r = L->r;
    const int tpr = lra_tpr_of_rank(r), rm = 16 * tpr;
    unsigned gb1, gb2, gbr, shm1, shm2, shmr;
    lra_geometry(N, r, 1, 0, &gb1, &shm1); lra_geometry(N, r, 2, 0, &gb2, &shm2);
    lra_geometry(N, r, 2, 2u * rm * rm * 4u, &gbr, &shmr);                       // the rotation keeps M_u, M_v in LDS as well
    const unsigned shm_s1 = 7u * rm * rm * 4u;
    // only the update's own slots: HSQ (||h||^2 of the last psgdk_lra_precond_grad, read by psgdk_flat_apply_clipped) and the apply's
    // reduction slots behind it survive an update that runs between precond_grad and the clipped parameter update
    // (update_preconditioner_first=False, psgd.py:1172-1183)
    if (phase == 0) {
        HIPCHK(hipMemsetAsync(sm, 0, (size_t)(tpr == 1 ? LraCfg<1>::HSQ : (tpr == 2 ? LraCfg<2>::HSQ : LraCfg<4>::HSQ)) * 4, st));
        HIPCHK(hipMemsetAsync(sm + (tpr == 1 ? LraCfg<1>::AB : (tpr == 2 ? LraCfg<2>::AB : LraCfg<4>::AB)), 0, 4, st));
    }
    // Gram recurrence: this update takes its Grams from the previous update's recurrence instead of reading U and V (row shards keep the pass:
    // their sums run over the ranks)
    const bool recur = L->gram_every > 0 && r > 0 && !L->sharded && L->gst != nullptr;
    const bool grams_carried = recur && L->gram_age >= 0 && L->gram_age < L->gram_every;
    const bool pk = lrapk_ok(L, g, v_noise, nullptr);
    const LrapkGeom pkg = pk ? lrapk_geom(N, r) : LrapkGeom{};
    L->pk_rows = pk ? pkg.nf : 0;
    LRA_T(L, {
        T* Qh = (T*)(L->work + L->qh_off); T* iq = (T*)(L->work + L->iq_off); T* diff = (T*)(L->work + L->diff_off);
        T* U = (T*)L->U; T* V = (T*)L->V; T* d = (T*)L->d;
        // (no preparation pass: v and h are rebuilt from g where they are used -- LraVH)
        const LraVH<T> vh{(const T*)g, (const T*)v_noise, damping, seed, offset, L->row0};
        if (pk && phase >= 1 &&
            lrapk_update_phase<T, TPR, RC>(L, phase, pkg, U, V, d, vh, Qh, iq, diff, sm, shm_s1, recur, grams_carried, update_u, lr, betaL, st)) {
            HIPCHK(hipGetLastError());
            return PSGDK_OK;
        }
        switch (phase) {
        case 0:
            if (grams_carried)      // UTU, VTV, VTU are the first three matrices of the scratch block, in gst's order
                HIPCHK(hipMemcpyAsync(sm, L->gst, (size_t)3 * LraCfg<TPR>::MS * 4, hipMemcpyDeviceToDevice, st));
            else if (r > 0) hipLaunchKernelGGL((lra_gram_kernel<T, TPR>), dim3(gb2), dim3(LRA_THREADS), shm2, st, (const T*)U, (const T*)V, N, r, sm);
            break;
        case 1:
            if (r > 0) {
                if (shm_s1 > 64u * 1024u)
                    HIPCHK(hipFuncSetAttribute((const void*)lra_small1_kernel<T, TPR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_s1));
                hipLaunchKernelGGL((lra_small1_kernel<T, TPR>), dim3(1), dim3(256), shm_s1, st, sm, r, recur ? L->gst + 3 * LraCfg<TPR>::MS : (float*)nullptr);
            }
            if constexpr (TPR == 1) {
                hipLaunchKernelGGL((lra_rotate_kernel<T, TPR, RC>), dim3(gbr), dim3(LRA_THREADS), shmr, st, U, V, (const T*)d, vh, N, r, sm);
            } else {      // wider rank classes: the rotation on the fp32 matrix cores
                const int rows_m = LRA_ROWS / TPR;
                const unsigned gm = (unsigned)std::max<int64_t>(1, std::min<int64_t>((N + rows_m - 1) / rows_m, 256 * 3));
                hipLaunchKernelGGL((lra_rotate_mfma_kernel<T, TPR>), dim3(gm), dim3(LRA_THREADS), 0, st, U, V, (const T*)d, vh, N, r, sm);
            }
            break;
        case 2:
            hipLaunchKernelGGL((lra_small2_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, r);
            hipLaunchKernelGGL((lra_pass3_kernel<T, TPR, RC>), dim3(gb2), dim3(LRA_THREADS), shm2, st, (const T*)U, (const T*)V, (const T*)d, vh,
                               Qh, iq, N, r, sm);
            break;
        case 3:
            hipLaunchKernelGGL((lra_small3_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, r);
            hipLaunchKernelGGL((lra_pass4_kernel<T, TPR, RC>), dim3(gb2), dim3(LRA_THREADS), shm2, st, (const T*)U, (const T*)V, (const T*)d, vh,
                               (const T*)Qh, (const T*)iq, diff, N, r, sm);
            break;
        default:
            hipLaunchKernelGGL((lra_small4_kernel<T, TPR>), dim3(1), dim3(64), 0, st, sm, L->Luvd, r, update_u ? 1 : 0, lr, betaL);
            hipLaunchKernelGGL((lra_pass5_kernel<T, TPR, RC>), dim3(gb1), dim3(LRA_THREADS), shm1, st, U, V, d, (const T*)Qh, (const T*)iq, (const T*)diff,
                               N, r, update_u ? 1 : 0, (const float*)sm);
            if (recur) {      // the Grams of the factors as pass 5 leaves them, for the next update
                hipLaunchKernelGGL((lra_gram_recur_kernel<T, TPR>), dim3(1), dim3(256), 0, st, (const float*)sm, (const float*)(L->gst + 3 * LraCfg<TPR>::MS),
                                   L->gst, r, update_u ? 1 : 0);
                L->gram_age = grams_carried ? L->gram_age + 1 : 1;
            }
            break;
        }
    });
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

static int lra_apply_phase_t(psgdk_lra* L, int phase, const void* g, void* out, hipStream_t st) {
    float* sm = (float*)(L->work + L->sm_off);
    unsigned gb1, shm1;
    lra_geometry(L->N, L->r, 1, 0, &gb1, &shm1);
    const bool pk = lrapk_ok(L, g, out, nullptr);
    const LrapkGeom pkg = pk ? lrapk_geom(L->N, L->r) : LrapkGeom{};
    L->pk_rows = pk ? pkg.nf : 0;
    LRA_T(L, {
        if (phase == 0) HIPCHK(hipMemsetAsync(sm + LraCfg<TPR>::HSQ, 0, (size_t)(LraCfg<TPR>::TOTAL - LraCfg<TPR>::HSQ) * 4, st));
        T* y = (T*)(L->work + L->y_off);
        if (pk && lrapk_apply_phase<T, TPR, RC>(L, phase, pkg, (const T*)g, y, (T*)out, sm, st)) {
            HIPCHK(hipGetLastError());
            return PSGDK_OK;
        }
        hipLaunchKernelGGL((lra_apply_kernel<T, TPR, RC>), dim3(gb1), dim3(LRA_THREADS), shm1, st, (const T*)L->U, (const T*)L->V, (const T*)L->d,
                           (const T*)g, y, (T*)out, L->N, L->r, phase, sm);
    });
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

static int lra_update_args_ok(const psgdk_lra* lra, const void* g, float lr, float betaL, float damping) {
    if (!lra || !g) return PSGDK_ERR_INVALID;
    if (!lra->work) return PSGDK_ERR_STATE;
    if (!(lr > 0.f) || !(betaL >= 0.f && betaL <= 1.f) || !(damping >= 0.f)) return PSGDK_ERR_INVALID;
    return PSGDK_OK;
}

int psgdk_lra_update_whiten(psgdk_lra* lra, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u,
                            float lr, float betaL, float damping, void* stream) {
    int rc = lra_update_args_ok(lra, g, lr, betaL, damping);
    if (rc) return rc;
    if (lra->sharded) return PSGDK_ERR_STATE;        // a row shard's reductions are incomplete without the exchanges: psgdk_lra_update_phase
    hipStream_t st = (hipStream_t)stream;
    if (lra->r > LRA_RMAX) return lra_update_general(lra, g, v_noise, seed, offset, update_u, lr, betaL, damping, st);
    for (int phase = 0; phase < 5; ++phase)
        if ((rc = lra_update_phase_t(lra, phase, g, v_noise, seed, offset, update_u, lr, betaL, damping, st))) return rc;
    return PSGDK_OK;
}

int psgdk_lra_precond_grad(psgdk_lra* lra, const void* g, void* out, void* stream) {
    if (!lra || !g || !out) return PSGDK_ERR_INVALID;
    if (!lra->work) return PSGDK_ERR_STATE;
    if (lra->sharded) return PSGDK_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    if (lra->r > LRA_RMAX) return lra_apply_general(lra, g, out, st);
    for (int phase = 0; phase < 3; ++phase) {
        const int rc = lra_apply_phase_t(lra, phase, g, out, st);
        if (rc) return rc;
    }
    return PSGDK_OK;
}

// ---- row shards of one LRA preconditioner (last row) -----------------------------------------------------
int psgdk_lra_set_row_shard(psgdk_lra* lra, int64_t row0) {
    if (!lra || row0 < 0) return PSGDK_ERR_INVALID;
    lra->row0 = row0; lra->sharded = true;      // (every rank: the tuned classes and, since round 6, the general path are cut into the same phases)
    return PSGDK_OK;
}

int psgdk_lra_update_phase(psgdk_lra* lra, int phase, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u,
                           float lr, float betaL, float damping, void* stream) {
    const int rc = lra_update_args_ok(lra, g, lr, betaL, damping);
    if (rc) return rc;
    if (phase < 0 || phase >= PSGDK_LRA_UPDATE_PHASES) return PSGDK_ERR_INVALID;
    if (lra->r > LRA_RMAX)
        return lra_update_general(lra, g, v_noise, seed, offset, update_u, lr, betaL, damping, (hipStream_t)stream, phase);
    return lra_update_phase_t(lra, phase, g, v_noise, seed, offset, update_u, lr, betaL, damping, (hipStream_t)stream);
}

int psgdk_lra_apply_phase(psgdk_lra* lra, int phase, const void* g, void* out, void* stream) {
    if (!lra || !g || !out || phase < 0 || phase >= PSGDK_LRA_APPLY_PHASES) return PSGDK_ERR_INVALID;
    if (!lra->work) return PSGDK_ERR_STATE;
    if (lra->r > LRA_RMAX) return lra_apply_general(lra, g, out, (hipStream_t)stream, phase);
    return lra_apply_phase_t(lra, phase, g, out, (hipStream_t)stream);
}

// the slots of the scratch block (fp32 words, relative to the start of the work buffer) that the phase's row pass has accumulated and the
// next phase reads: up to PSGDK_LRA_MAX_SEGMENTS runs of words, each summed (op 0) or maximised (op 1) over the shards
int psgdk_lra_phase_segments(const psgdk_lra* lra, int kind, int phase, int* n_segments, int64_t* word_offset, int* words, int* op) {
    if (!lra || !n_segments || !word_offset || !words || !op) return PSGDK_ERR_INVALID;
    const int tpr = lra_tpr_of_rank(lra->r);
    const int64_t base = (int64_t)(lra->sm_off / 4);
    int n = 0;
    auto seg = [&](int off, int cnt, int o) { word_offset[n] = base + off; words[n] = cnt; op[n] = o; ++n; };
    if (lra->r > LRA_RMAX) {      // the general path's scratch layout (kernels_lra_gen.hiph): r x r matrices and r-vectors without padding
        const LragLayout Y = lrag_layout(lra->r);
        const int r = lra->r;
        if (kind == 0) {
            if (phase == 0) seg(Y.UTU, 2 * Y.R2, 0);                                              // U^T U, V^T V
            else if (phase == 1) { seg(Y.VTU, Y.R2, 0); seg(Y.UTU2, 2 * Y.R2, 0); seg(Y.T1, 2 * r, 0); }   // rotated Grams; V^T (d h), U^T (v / d)
            else if (phase == 2) { seg(Y.ATU, 4 * r, 0); seg(Y.SC, 2, 0); }
            else if (phase == 3) seg(Y.SC + 2, 2, 1);
            else if (phase != 4) return PSGDK_ERR_INVALID;
        } else if (kind == 1) {
            if (phase == 0) seg(Y.VTX2, r, 0);
            else if (phase == 1) seg(Y.UTY, r, 0);
            else if (phase == 2) seg(Y.SC + 7, 1, 0);
            else return PSGDK_ERR_INVALID;
        } else return PSGDK_ERR_INVALID;
        *n_segments = n;
        return PSGDK_OK;
    }
#define LRA_SEGS(TPR_)                                                                                                    \
    { using C = LraCfg<TPR_>;                                                                                              \
      if (kind == 0) {                                                                                                     \
          if (phase == 0) seg(C::UTU, 3 * C::MS, 0);                                                                       \
          else if (phase == 1) seg(C::VTX, 2 * C::RM, 0);                                                                  \
          else if (phase == 2) { seg(C::ATU, 4 * C::RM, 0); seg(C::NA2, 2, 0); }                                           \
          else if (phase == 3) seg(C::MAX1, 2, 1);                                                                         \
          else if (phase != 4) return PSGDK_ERR_INVALID;                                                                   \
      } else if (kind == 1) {                                                                                              \
          if (phase == 0) seg(C::VTX2, C::RM, 0);                                                                          \
          else if (phase == 1) seg(C::UTY, C::RM, 0);                                                                      \
          else if (phase == 2) seg(C::HSQ, 1, 0);                                                                          \
          else return PSGDK_ERR_INVALID;                                                                                   \
      } else return PSGDK_ERR_INVALID; }
    if (tpr == 1) LRA_SEGS(1) else if (tpr == 2) LRA_SEGS(2) else LRA_SEGS(4)
#undef LRA_SEGS
    *n_segments = n;
    return PSGDK_OK;
}

int psgdk_lra_last_sumsq(const psgdk_lra* lra, const float** dev_ptr) {
    if (!lra || !dev_ptr) return PSGDK_ERR_INVALID;
    if (!lra->work) return PSGDK_ERR_STATE;
    if (lra->r > LRA_RMAX) { *dev_ptr = (const float*)(lra->work + lra->sm_off) + lrag_layout(lra->r).SC + 7; return PSGDK_OK; }
    const int tpr = lra_tpr_of_rank(lra->r);
    *dev_ptr = (const float*)(lra->work + lra->sm_off) + (tpr == 1 ? LraCfg<1>::HSQ : (tpr == 2 ? LraCfg<2>::HSQ : LraCfg<4>::HSQ));
    return PSGDK_OK;
}

int psgdk_lra_info(const psgdk_lra* lra, int what, int64_t* value) {
    if (!lra || !value) return PSGDK_ERR_INVALID;
    switch (what) {
    case PSGDK_LRA_INFO_PACKED_ROWS: *value = lra->pk_rows; return PSGDK_OK;
    case PSGDK_LRA_INFO_GRAM_AGE: *value = lra->gram_age; return PSGDK_OK;
    default: return PSGDK_ERR_INVALID;
    }
}

}  // extern "C"
