/*
 * psgdk.h -- C ABI of the MI355X-native PSGD preconditioner engine (libpsgdk.so).
 *
 * This is the drop-in boundary for ONE hot path of lixilinx/psgd_torch: the Kron "Q0.5EQ1.5" whitening
 * preconditioner update + apply (and, secondarily, the LRA update + apply).  The reference has no FFI -- its seam
 * is three Python functions selected at wrapped_as_torch_optimizer_for_ddp.py:84-86 plus psgd.init_kron
 * (..._ddp.py:131-135).  Each entry point below names the reference code it replaces (file:line under the
 * reference repo).  All pointers are raw device pointers (or host arrays of device pointers where stated);
 * no torch types cross this boundary.  Every call is stream-ordered on `stream` (a hipStream_t passed as void*),
 * performs no hidden host synchronisation, never frees or allocates caller memory, and returns an int status
 * (no C++ exceptions cross the ABI).
 *
 * Differences from the reference seam, on purpose (MI355X-first):
 *   - calls are BATCHED over all tensors of a plan (one grouped launch per stage instead of ~100 ATen launches
 *     per tensor); a 1-tensor plan gives the reference's per-tensor functional behaviour;
 *   - randomness is explicit: either caller-supplied noise buffers (parity testing) or a counter-based Philox
 *     stream (seed, offset), so replicas stay identical without broadcasting RNG state (..._ddp.py:88-104);
 *   - state lives in two caller-allocated arenas (persistent `state`, scratch `work`) whose layout the plan
 *     defines: every matrix is zero-padded to multiples of 64 in both dims, each dense factor Q is stored together
 *     with its transpose Qt, and a 2-D tensor with exactly one dense factor is held with the dense dim last.
 */
#ifndef PSGDK_H
#define PSGDK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSGDK_VERSION 404    /* round 6 (404): psgdk_lra_info (which row kernels the last LRA call took);
                                round 6 (403): psgdk_flat_set_clip_groups / psgdk_flat_apply_groups, psgdk_export_precond_grad clip = 2 (deferred clip of row shards),
                                row shards for the QEQ / QUAD geometries, PSGDK_INFO_NLB_MEMBER_COLS;
                                round 6 (402): psgdk_precond_grad_apply (the parameter update fused into the apply's last product);
                                round 5 (401): row shards of the LRA preconditioner (psgdk_lra_set_row_shard, psgdk_lra_update_phase / _apply_phase / _phase_segments);
                                round 4 (400): row shards (psgdk_plan_set_row_shard, psgdk_update_precond_begin / _finish, psgdk_balance_phase),
                                psgdk_profile_read_calls; (300: test hooks moved to psgdk_test.h; 200: PSGDK_MAX_DIMS 8 -> 26, PSGDK_ERR_NLB_TIMEOUT) */
#define PSGDK_MAX_DIMS 26     /* most dims of one tensor: the reference's own limit (einsum letters, psgd.py:197-198) */

/* status codes */
enum {
    PSGDK_OK = 0,
    PSGDK_ERR_INVALID = 1,     /* bad argument (mirrors the reference's assert/ValueError sites) */
    PSGDK_ERR_UNSUPPORTED = 2, /* valid in the reference, not built (LRA rank > 1024; row shards outside the Q0.5EQ1.5 / QEQ / QUAD geometries) */
    PSGDK_ERR_HIP = 3,         /* a HIP runtime call failed; see psgdk_last_hip_error() */
    PSGDK_ERR_STATE = 4,       /* call order violated (e.g. arenas not bound) */
    PSGDK_ERR_NLB_TIMEOUT = 5  /* returned ONCE by the first psgdk_update_precond_* call after a cooperative norm-bound launch of an
                                  earlier call gave up waiting for a sibling workgroup (bounded spin): the dense factors concerned
                                  skipped that one preconditioner update (state stays valid), and the plan has switched to the
                                  multi-launch route for good.  Nothing was enqueued by the call that returns it: repeat the call. */
};

/* element types */
enum { PSGDK_BF16 = 0, PSGDK_F32 = 1 };
/* factor kinds, psgd.py:208 */
enum { PSGDK_DIAG = 0, PSGDK_DENSE = 1, PSGDK_SCALAR = 2 };
/* which buffer a stage reads: the momentum EMA or the (cast) gradient, ..._ddp.py:145,150 */
enum { PSGDK_SRC_EMA = 0, PSGDK_SRC_GRAD = 1 };

typedef struct psgdk_plan psgdk_plan;

int psgdk_version(void);
const char* psgdk_strerror(int status);
int psgdk_last_hip_error(void);

/* ---- planning: replaces psgd.init_kron's structural half (psgd.py:161-263, dense/diag rule psgd.py:208) -----
 * n_tensors tensors; tensor t has ndim[t] dims (AFTER squeeze(), ..._ddp.py:124) listed consecutively in `dims`.
 * precond_dtype: PSGDK_BF16 | PSGDK_F32 (..._ddp.py:41,58).  use_momentum: allocate the EMA buffers (..._ddp.py:137).
 * Tensors with more than 26 (= PSGDK_MAX_DIMS) dims -> PSGDK_ERR_INVALID (psgd.py:197-198).
 * Tensors with <= 2 dims take the grouped-GEMM path; 3..PSGDK_MAX_DIMS dims a generic mode-product path. */
int psgdk_plan_create(psgdk_plan** out, int n_tensors, const int32_t* ndim, const int64_t* dims, double max_size,
                      double max_skew, int precond_dtype, int use_momentum);
int psgdk_plan_destroy(psgdk_plan* plan);
/* Optional, before psgdk_plan_bind: GLOBAL ids (< 2^28) of the plan's tensors for the Philox noise streams, so that a
 * rank owning a subset of an optimizer's tensors draws exactly what a single GPU owning all of them would (default:
 * the tensor's index in the plan).  New relative to the reference, which keeps replicas in lock-step by broadcasting
 * torch RNG state (wrapped_as_torch_optimizer_for_ddp.py:88-104). */
int psgdk_plan_set_stream_ids(psgdk_plan* plan, const uint32_t* ids);
/* Optional, before psgdk_plan_arena_bytes / psgdk_plan_bind: the update geometry the plan will be driven with -- the dQ
 * argument of psgd.init_kron (psgd.py:161).  PSGDK_GEOM_Q0P5EQ1P5 (default; dense Q, psgd.py:394-419) or PSGDK_GEOM_EQ
 * (upper-triangular Q, psgd.py:278-336; needs extra work buffers), PSGDK_GEOM_QEQ (psgd.py:367-391), PSGDK_GEOM_QUAD (symmetric Q, psgd.py:455-483), PSGDK_GEOM_QEP
 * (psgd.py:339-364), PSGDK_GEOM_QUAD4P (psgd.py:486-513: the factors ARE P; psgdk_precond_grad then applies every
 * factor once, as KronWhiten does for this choice, psgd.py:573; init scale is squared by the caller like psgd.py:186-187).
 * PSGDK_GEOM_PRO4P (psgd.py:422-452): fits P with dP = P^0.5 E P; after the gradient step every dense factor takes up to
 * ten procrustes_step3 rotations (psgd.py:127-158) and stops, per factor and on the device, once it is Hermitian to 1e-3.  Each update entry point below requires the plan to
 * carry its geometry (else PSGDK_ERR_STATE). */
#define PSGDK_GEOM_Q0P5EQ1P5 0
#define PSGDK_GEOM_EQ 1
#define PSGDK_GEOM_QEQ 2
#define PSGDK_GEOM_QUAD 3
#define PSGDK_GEOM_QEP 4
#define PSGDK_GEOM_QUAD4P 5
#define PSGDK_GEOM_PRO4P 6
int psgdk_plan_set_geometry(psgdk_plan* plan, int geometry);

/* arena sizes in bytes; caller allocates both zero-filled, 256-byte aligned, and binds them. */
int psgdk_plan_arena_bytes(const psgdk_plan* plan, size_t* state_bytes, size_t* work_bytes);
int psgdk_plan_bind(psgdk_plan* plan, void* state_arena, void* work_arena);

/* introspection so the host can expose Q / L / ema as strided views of the state arena (the reference keeps them in
 * optimizer.state[p]["QL"], ["ema"], ..._ddp.py:129-137).  Offsets are bytes from the state arena base.
 * Dense factor: rows = cols = d, row stride `ld` elements.  Diag/scalar factor: rows = 1, cols = d. */
int psgdk_plan_num_factors(const psgdk_plan* plan, int t, int* n_factors);
int psgdk_plan_factor_view(const psgdk_plan* plan, int t, int i, int* kind, size_t* q_offset, int64_t* d,
                           int64_t* ld, size_t* lipschitz_offset);
/* EMA view: logical shape (rows, cols) of the squeezed tensor; element (r, c) lives at
 * offset + (transposed ? c*ld + r : r*ld + c) * elem_size. */
int psgdk_plan_ema_view(const psgdk_plan* plan, int t, size_t* offset, int64_t* rows, int64_t* cols, int64_t* ld,
                        int* transposed);

/* ---- init_kron's numeric half (psgd.py:200,207,210,228): Q_i = scale^(1/k) I | ones, L_i = 0, ema = 0 ------ */
int psgdk_init_state(psgdk_plan* plan, double scale, void* stream);
/* after the HOST wrote into Q views (load_state_dict, tests): rebuild the internal transposes / cached P. */
int psgdk_state_changed(psgdk_plan* plan, void* stream);

/* ---- momentum + cast: replaces ..._ddp.py:117-143 (coupled weight decay, squeeze+cast, EMA with warm-up beta) --
 * grads/params: HOST arrays of n_tensors DEVICE pointers (contiguous tensors in their logical layout; params may be
 * NULL when coupled_wd == 0).  beta = min(t/(t+1), momentum) is computed by the caller per ..._ddp.py:141.
 * Produces ema <- beta*ema + (1-beta)*cast(g + coupled_wd*p) (if the plan has momentum); keep_grad != 0 (forced when
 * the plan has no momentum) also keeps cast(g) for PSGDK_SRC_GRAD consumers (whiten_grad=True, ..._ddp.py:145). */
struct psgdk_noise;
/* Optional fusion hint: when the caller knows psgdk_update_precond_q0p5eq1p5 will follow with exactly these arguments,
 * psgdk_accumulate also writes the update's damped input G + (damping + eps|G|) * noise (psgd.py:402-403) in the same
 * pass; the update call then skips that stage.  NULL = no fusion (the update computes it itself). */
typedef struct psgdk_damp {
    int source;                     /* PSGDK_SRC_EMA | PSGDK_SRC_GRAD */
    float damping;
    const struct psgdk_noise* noise;   /* explicit g_noise (only that member is read here) or NULL for Philox */
    uint64_t seed, offset;
} psgdk_damp;
int psgdk_accumulate(psgdk_plan* plan, const void* const* grads, int grad_dtype, const void* const* params,
                     int param_dtype, float coupled_wd, float beta, int keep_grad, const psgdk_damp* damp, void* stream);

/* explicit noise for parity testing (all device pointers, element type = precond dtype, logical layouts):
 *   g_noise[t]            : numel(t) values, the randn_like(G) of psgd.py:403
 *   spd_noise[t*PSGDK_MAX_DIMS+i] : 32 x d_i values, the randn(32,d) of psgd.py:62 for dense factor i of tensor t
 *   skh_noise[t*PSGDK_MAX_DIMS+i] : 32 x d_i values, the randn(32,d) of psgd.py:87
 * A NULL psgdk_noise* selects the built-in counter-based Philox4x32 stream keyed by (seed, offset). */
typedef struct psgdk_noise {
    const void* const* g_noise;
    const void* const* spd_noise;
    const void* const* skh_noise;
} psgdk_noise;

/* ---- replaces psgd.update_precond_kron_whiten_q0p5eq1p5 (psgd.py:394-419) with its helpers
 * norm_lower_bound_spd (psgd.py:46-68), procrustes_step2 (psgd.py:101-124) -> norm_lower_bound_skh (psgd.py:71-93),
 * balance_kron_precond (psgd.py:266-275).  In place on Q and L of every tensor of the plan.
 * balance_mask: HOST array of n_tensors bytes; nonzero = the caller's rand([]) < 0.01 draw (psgd.py:418) fired for
 * that tensor (NULL = never). */
int psgdk_update_precond_q0p5eq1p5(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                                   const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                                   const uint8_t* balance_mask, void* stream);

/* ---- replace psgd.update_precond_kron_whiten_qeq (psgd.py:367-391: Q -= lr/L (Q term1 - c Q), no Procrustes step) and
 * psgd.update_precond_kron_whiten_quad (psgd.py:455-483: two half steps p = q - lr/2/L (term1 q - c q),
 * p = p - lr/2/L (p term1 - c p), q = (p + p^T)/2; diagonal factors q *= (1 - lr/2/L (term1 - c))^2).  Same arguments as
 * psgdk_update_precond_q0p5eq1p5; noise->skh_noise is not read. */
int psgdk_update_precond_qeq(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                             const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                             const uint8_t* balance_mask, void* stream);
int psgdk_update_precond_quad(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                              const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                              const uint8_t* balance_mask, void* stream);
/* psgd.update_precond_kron_whiten_quad4p (psgd.py:486-513): as _quad with full steps lr/L, on P itself. */
int psgdk_update_precond_quad4p(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                                const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                                const uint8_t* balance_mask, void* stream);
/* psgd.update_precond_kron_whiten_pro4p (psgd.py:422-452).  Explicit noise: skh_noise[slot] holds the draws of the successive
 * procrustes_step3 calls of that factor back to back, 10 x 32 x d values (unused ones are never read). */
int psgdk_update_precond_pro4p(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                               const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                               const uint8_t* balance_mask, void* stream);
/* psgd.update_precond_kron_whiten_qep (psgd.py:339-364): balancing of every tensor FIRST and on every call (not optional,
 * psgd.py:346-347); per factor term1 = Gram_i(Q_i Pg), term2 = (numel/d) Q Q^T, ell = ||term1 + term2||_lb,
 * Q -= lr/L (term1 - term2) Q (diagonal: q *= 1 - lr/L (term1 - term2)).  No gate draw, hence no balance_mask. */
int psgdk_update_precond_qep(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                             const psgdk_noise* noise, uint64_t seed, uint64_t offset, void* stream);

/* ---- replaces psgd.update_precond_kron_whiten_eq -> update_precond_kron_eq (psgd.py:330-336 -> 278-319), the
 * triangular geometry dQ = E*Q:  V = noise, Hvp = G + (damping + eps|G|) V;  A = (kron Q) Hvp (exprA);
 * B = V x_i Q_i^{-T} by right triangular solves carried out in fp32 (psgd.py:288-303);  per factor
 * term1 = Gram_i(A), term2 = Gram_i(B);  dense: ell = ||term1+term2||_lb (psgd.py:46-68), Q -= lr/L triu(term1-term2) Q;
 * diagonal: ell = max(term1+term2), q -= lr/L (term1-term2) q;  then the 1% balancing (psgd.py:318-319).
 * The plan must have been created with psgdk_plan_set_geometry(plan, PSGDK_GEOM_EQ) (else PSGDK_ERR_STATE).
 * noise->g_noise is V (probe AND damping noise, psgd.py:334-336); noise->spd_noise as above; skh_noise is not read. */
int psgdk_update_precond_eq(psgdk_plan* plan, int source, float lr, float betaL, float damping,
                            const psgdk_noise* noise, uint64_t seed, uint64_t offset,
                            const uint8_t* balance_mask, void* stream);

/* ---- replaces psgd.precond_grad_kron (psgd.py:322-327): h_t = (kron_i Q_i^T Q_i) src_t for every tensor; h stays
 * in the work arena (consumed by psgdk_apply_update / psgdk_read_precond_grad). */
int psgdk_precond_grad(psgdk_plan* plan, int source, void* stream);

/* ---- replaces ..._ddp.py:117-120 (decoupled weight decay) and 153-157 (RMS clip, element clip, p -= lr*h) ----
 * params: HOST array of n_tensors DEVICE pointers (logical layout). */
int psgdk_apply_update(psgdk_plan* plan, void* const* params, int param_dtype, float lr, float decoupled_wd,
                       float max_avg_amp, float max_elem_amp, void* stream);

/* ---- psgdk_precond_grad followed by psgdk_apply_update, as ONE call (..._ddp.py:150-157) ----
 * Same arguments, same result as the two calls.  Where a tensor allows it (its h leaves a grouped-GEMM product; fp32 parameters,
 * 16-byte aligned, logical row length a multiple of 4) the parameter update runs INSIDE the epilogue of that product -- the h tile is
 * in registers there, so the separate pass over h and p (10 B / parameter) disappears; every other tensor is updated by the streaming
 * pass within the same call.  The RMS clip (..._ddp.py:153-155) needs the whole tensor's sum of h^2, which the product itself
 * accumulates: the fused update is applied with clip scale 1 and a tensor whose RMS turns out above max_avg_amp is corrected afterwards
 * (p + lr clamp(h) - lr clamp(h scale): at most one fp32 rounding away from the two-call result; identical bits when no clip engages).
 * Afterwards h is CONSUMED: psgdk_apply_update / psgdk_read_precond_grad / psgdk_export_precond_grad return PSGDK_ERR_STATE until the next
 * psgdk_precond_grad.  Plans with row shards, bf16 parameters and small plans on the K-split kernel take the two-call route inside. */
int psgdk_precond_grad_apply(psgdk_plan* plan, int source, void* const* params, int param_dtype, float lr, float decoupled_wd,
                             float max_avg_amp, float max_elem_amp, void* stream);

/* Sharded path (build-side design): the clipped preconditioned gradients (wrapped_as_torch_optimizer_for_ddp.py:153-156)
 * of ALL tensors of the plan, exported in one launch to caller buffers outs[t] (logical contiguous order, element type
 * out_dtype) -- normally this rank's slices of the flat all-gather buffer.  clip: 0 = as they are, 1 = clipped, 2 (round 6) = clipped except
 * ROW SHARDS, which leave unclipped for psgdk_flat_apply_groups (below) to clip after the exchange. */
int psgdk_export_precond_grad(psgdk_plan* plan, void* const* outs, int out_dtype, int clip, float max_avg_amp, float max_elem_amp,
                              void* stream);
/* copy h of tensor t to `out` (logical layout, contiguous, out_dtype) -- the return value of precond_grad_kron.
 * clip != 0 applies the ..._ddp.py:153-156 clipping first (used by the sharded path to ship clipped h). */
int psgdk_read_precond_grad(psgdk_plan* plan, int t, void* out, int out_dtype, int clip, float max_avg_amp,
                            float max_elem_amp, void* stream);

/* ---- the exchange step of the sharded (N > 1) path.  After the all-gather of the clipped preconditioned gradients every rank
 * applies p <- p (1 - decoupled_wd lr) - lr h (..._ddp.py:120,157) to ALL n tensors in one launch, reading h from the flat
 * exchange buffer (tensor t at element offset h_offset[t], numel[t] elements, dtype h_dtype; offsets that are multiples of
 * 8 take the 16-byte path).  New relative to the reference, which runs replicas only. */
typedef struct psgdk_flat psgdk_flat;
int psgdk_flat_create(psgdk_flat** out, int n, const int64_t* numel, const int64_t* h_offset);
int psgdk_flat_destroy(psgdk_flat* flat);
int psgdk_flat_apply(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                     float decoupled_wd, void* stream);
/* Deferred clip of ROW SHARDS (round 6, version 403): the RMS clip of a row-split tensor (..._ddp.py:153-155) needs the sum of h^2 over ALL its
 * row blocks.  Instead of a collective of its own between psgdk_precond_grad and the export, every member exports its block UNCLIPPED
 * (psgdk_export_precond_grad with clip = 2) and puts its partial sum -- the fp32 word at PSGDK_INFO_HSUMSQ_OFFSET + 4 t -- into its segment of
 * the exchange buffer; after the gather psgdk_flat_apply_groups clips the blocks where it applies them.  psgdk_flat_set_clip_groups declares
 * which pieces of the flat layout belong to which row-split tensor (piece_group[t] = group, or -1: applied as it is), where member 0's partial
 * sum of each group lies inside h_flat (sum_off_bytes[g], a multiple of 4) and how far apart the members' words are (member_stride_bytes: the
 * segment size); the sums are added in member order (the same bits on every rank), numel_clip[g] is the WHOLE tensor's element count.  Same
 * arithmetic as the owner-side clip: scale = max_avg_amp / rms when rms > max_avg_amp, rounded to h_dtype, clamped to +-max_elem_amp. */
int psgdk_flat_set_clip_groups(psgdk_flat* flat, int n_groups, const int32_t* piece_group, const int64_t* sum_off_bytes,
                               int64_t member_stride_bytes, int members, const int64_t* numel_clip);
int psgdk_flat_apply_groups(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                            float decoupled_wd, float max_avg_amp, float max_elem_amp, void* stream);
/* LRAWhiten.step's vector work around the preconditioner (psgd.py:1142-1155, 1179-1187), one launch each over ALL parameters of
 * the flat layout (h_offset[t] = running sum of numel: the concatenation order of psgd.py:1142):
 * psgdk_flat_gather: g_flat[off_t + i] = grads[t][i] (cast to flat_dtype); if m_flat: m <- beta m + (1 - beta) g in place
 *   (psgd.py:1153); if sum_g4_dev: *sum_g4_dev = sum g^4 (the on-the-fly scale of d, psgd.py:1144-1145).
 * psgdk_flat_apply_clipped: params[t] -= lr * clip(h)[off_t..]: h scaled by max_avg_amp / rms(h) when rms(h) = sqrt(*h_sumsq_dev /
 *   h_numel) exceeds max_avg_amp, then clamped to +-max_elem_amp (psgd.py:1179-1187).  psgdk_lra_last_sumsq returns the device
 *   word into which psgdk_lra_precond_grad accumulated the sum of squares of its last output. */
int psgdk_flat_gather(psgdk_flat* flat, const void* const* grads, int grad_dtype, void* g_flat, int flat_dtype, void* m_flat, float beta,
                      float* sum_g4_dev, void* stream);
int psgdk_flat_apply_clipped(psgdk_flat* flat, void* const* params, int param_dtype, const void* h_flat, int h_dtype, float lr,
                             const float* h_sumsq_dev, int64_t h_numel, float max_avg_amp, float max_elem_amp, void* stream);

/* fill `out` with the engine's N(0,1) stream (same generator the fused kernels use), for statistical tests. */
int psgdk_fill_normal(void* out, int dtype, int64_t n, uint64_t seed, uint64_t offset, uint32_t stream_id,
                      void* stream);

/* ===================================================================================================================
 * LRA preconditioner Q = (I + U V^T) diag(d) on the concatenated parameter vector (psgd.py:987-1072).
 * U, V: N x r row-major, d: N, element type = dtype (PSGDK_BF16 | PSGDK_F32), all caller-owned device memory;
 * Luvd: 3 fp32 device scalars (Lu, Lv, Ld; psgd.py:1123).  r <= 64: three tuned rank classes (1, 2, 4 threads per row for r <= 16, 32, 64); 64 < r <= 1024: a general path
 * (one wavefront per row, r x r matrices in global memory) -- the same stages and rounding points, written for generality.
 * =================================================================================================================== */
typedef struct psgdk_lra psgdk_lra;
int psgdk_lra_create(psgdk_lra** out, int64_t N, int r, int dtype);
int psgdk_lra_destroy(psgdk_lra* lra);
int psgdk_lra_work_bytes(const psgdk_lra* lra, size_t* work_bytes);
int psgdk_lra_bind(psgdk_lra* lra, void* U, void* V, void* d, float* Luvd, void* work);
/* The Grams U^T U, V^T V, V^T U of psgd.py:1006 without reading the factors (round 6; ranks <= 64, not for row shards).  An update
 * determines the Grams of the factors it leaves behind: the rotation's follow analytically, and the rank-1 step of psgd.py:1043 / 1052
 * changes them by outer products of r-vectors the update reduces anyway (a^T U, b^T U, a^T V, b^T V, |a|^2, |b|^2, a^T b).  With every > 0
 * psgdk_lra_update_whiten carries them from update to update (fp32, r x r) and reads the factors for them only on the first update, every
 * `every` updates (rounding drift; 16 is what the Python host uses) and after psgdk_lra_bind / psgdk_lra_state_changed: 2 of the 15 matrix
 * passes of an update + apply disappear.  every = 0 (default): psgd.py:1006 as written.  A caller that writes U or V ITSELF between two
 * updates must say so (psgdk_lra_state_changed): the engine cannot see it. */
int psgdk_lra_set_gram_recurrence(psgdk_lra* lra, int every);
int psgdk_lra_state_changed(psgdk_lra* lra);
/* replaces psgd.update_precond_lra_whiten (psgd.py:1066-1072) -> update_precond_lra (psgd.py:994-1052), in place on
 * U, V, d, Luvd.  g: the N-vector to whiten; v_noise: the randn_like(g) draw of psgd.py:1070 or NULL for Philox
 * (seed, offset); update_u: the caller's rand([]) < 0.5 coin of psgd.py:1035 (nonzero = update U, else V). */
int psgdk_lra_update_whiten(psgdk_lra* lra, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u,
                            float lr, float betaL, float damping, void* stream);
/* replaces psgd.precond_grad_lra (psgd.py:1055-1063): out = Q^T Q g. */
int psgdk_lra_precond_grad(psgdk_lra* lra, const void* g, void* out, void* stream);
/* device word holding the sum of squares of the last psgdk_lra_precond_grad output (see psgdk_flat_apply_clipped) */
int psgdk_lra_last_sumsq(const psgdk_lra* lra, const float** dev_ptr);
/* What the last psgdk_lra_update_* / psgdk_lra_precond_grad / _apply_phase call did (host-side bookkeeping, no device access):
 *   PSGDK_LRA_INFO_PACKED_ROWS  rows covered by the two-rows-per-thread bf16 kernels (csrc/kernels_lra_pk.hiph: bf16 factors, even rank
 *                               2 .. 16, N >= 512, 4-byte aligned N-vectors): floor(N / 512) * 512, or 0 when the one-row kernels ran
 *                               everything (also with PSGDK_LRA_PK=0 in the environment, the A/B switch of the tests);
 *   PSGDK_LRA_INFO_GRAM_AGE     updates since the carried Grams were last read from the factors (-1: not valid / recurrence off). */
#define PSGDK_LRA_INFO_PACKED_ROWS 0
#define PSGDK_LRA_INFO_GRAM_AGE 1
int psgdk_lra_info(const psgdk_lra* lra, int what, int64_t* value);

/* ---- row shards of ONE LRA preconditioner (last row; new relative to the reference, whose LRA runs replicas only).
 * U, V, d are cut by rows over the ranks: an object created with N = the shard's row count and declared by psgdk_lra_set_row_shard(row0)
 * holds rows [row0, row0 + N) of the whole factors; g, v_noise and out are the shard's slices.  Every stage of psgd.py:994-1063 is row-local
 * except its r x r / r-vector / scalar reductions, which the row passes accumulate in the fp32 scratch block at the start of the work
 * buffer.  So the update runs as PSGDK_LRA_UPDATE_PHASES phases and the apply as PSGDK_LRA_APPLY_PHASES (the SAME launches, in the same
 * order, that psgdk_lra_update_whiten / psgdk_lra_precond_grad issue back to back), and between two phases the caller reduces the words
 * psgdk_lra_phase_segments names over the ranks -- op 0: sum, op 1: maximum -- and writes the result back on every rank (one collective
 * per phase: 4 per update, the fifth phase ends with nothing to exchange; 3 per apply, the last one being the sum of out^2 the RMS clip
 * reads through psgdk_lra_last_sumsq).  The Philox counters of the damping noise run over the WHOLE vector (row0 + local row), so the
 * shards draw what one GPU would.  Every rank the engine holds (round 6: the general path above rank 64 is cut into the same phases); on a declared shard the one-call
 * forms return PSGDK_ERR_STATE.  psgd_torch_amd/lra_sharded.py drives this over torch.distributed. */
#define PSGDK_LRA_UPDATE_PHASES 5
#define PSGDK_LRA_APPLY_PHASES 3
#define PSGDK_LRA_MAX_SEGMENTS 4
int psgdk_lra_set_row_shard(psgdk_lra* lra, int64_t row0);
int psgdk_lra_update_phase(psgdk_lra* lra, int phase, const void* g, const void* v_noise, uint64_t seed, uint64_t offset, int update_u,
                           float lr, float betaL, float damping, void* stream);
int psgdk_lra_apply_phase(psgdk_lra* lra, int phase, const void* g, void* out, void* stream);
/* kind 0: update, 1: apply.  word_offset[i] (fp32 words from the start of the work buffer), words[i], op[i] for i < *n_segments
 * (<= PSGDK_LRA_MAX_SEGMENTS; 0 for the update's last phase). */
int psgdk_lra_phase_segments(const psgdk_lra* lra, int kind, int phase, int* n_segments, int64_t* word_offset, int* words, int* op);

/* ---- introspection (bench.py / tests; no reference counterpart): how the plan runs.  NLB_COOP: the norm lower bounds
 * (psgd.py:46-93) run as one cooperative launch per bound instead of start block + 4 grouped-GEMM products + scalars (set at
 * psgdk_plan_bind; cleared for good when a launch times out, see PSGDK_ERR_NLB_TIMEOUT); NLB_FALLBACKS: how often that
 * happened.  The workgroups of a cooperative launch exchange their slabs with a placement-independent device-scope protocol
 * and bound every spin, so concurrent work on other streams can delay a launch but neither hang nor corrupt it. */
#define PSGDK_INFO_NLB_COOP 0
#define PSGDK_INFO_NLB_FALLBACKS 1
#define PSGDK_INFO_DENSE_FACTORS 2
#define PSGDK_INFO_MAX_DENSE_DIM 3   /* padded to a multiple of 64 */
#define PSGDK_INFO_HSUMSQ_OFFSET 4   /* byte offset in the WORK arena of the fp32 sums of h^2, one per tensor (written by psgdk_precond_grad,
                                        read by the clip of psgdk_apply_update / psgdk_export_precond_grad): a row shard's entry is summed
                                        over the members by the caller in between */
#define PSGDK_INFO_BALNORM_OFFSET 5  /* byte offset in the WORK arena of the balancing slots of psgdk_balance_phase */
#define PSGDK_INFO_UPDATE_FUSED 6    /* how many tensors' parameter updates ran inside a GEMM epilogue in the last psgdk_precond_grad_apply
                                        (0: it took the two-call route, or the last h came from psgdk_precond_grad) */
#define PSGDK_INFO_NLB_MEMBER_COLS 7  /* columns of a factor one workgroup of the cooperative bound covers: 256, or 128 where the plan is small
                                        enough for twice the members (round 6), or 128 with one workgroup per factor (widest factor <= 128);
                                        0 without a cooperative launch */
int psgdk_plan_info(const psgdk_plan* plan, int what, int64_t* value);

/* ---- row shards: the sharded (multi-GPU) path's split of a DOMINANT tensor (new; the reference only replicates -- SURVEY 8e.  GPT-2's
 * tied embedding is 31 % of the model's elements and 20 % of a step's FLOPs: owned by one rank it caps the scaling of everything).
 * Tensor t of this plan is declared to be rows [row0, row0 + rows_t) of a (global_rows x cols) matrix whose dim-0 factor is diagonal and
 * whose dim-1 factor is dense (psgd.py:208 applied to the WHOLE matrix); `members` plans, one per rank, hold the blocks.  Then
 *   - term2 = numel / size (psgd.py:407,412) and the RMS clip (..._ddp.py:153-155) use the whole matrix's element count; the damping
 *     noise's Philox counters run over the whole matrix (a shard draws what one GPU would);
 *   - the dense factor is REPLICATED on the members and fitted to the whole matrix: the update is cut in two halves around ONE exchange,
 *     psgdk_update_precond_begin ... all-gather of `exchange` ... psgdk_update_precond_finish.  `exchange` holds `members` records of
 *     psgdk_plan_exchange_bytes() bytes each (256-byte aligned; record m written by member m's begin): per shard the fp32 partial mode
 *     Gram sum_rows Pg^T Pg (psgd.py:405) and the member's max of the diagonal factor's term1 (psgd.py:408).  finish sums the partial
 *     Grams in member order (identical bits on every member), rounds once, and continues with psgd.py:406-416; the diagonal factor's rows
 *     are updated locally with the maximum over all members;
 *   - balancing (psgd.py:418, 266-275) of a shard needs max |q| over all members' rows: psgdk_balance_phase(0), caller's max over the
 *     members of the 2 floats per shard at PSGDK_INFO_BALNORM_OFFSET, psgdk_balance_phase(1); psgdk_update_precond_finish ignores the
 *     balance flags of shards;
 *   - psgdk_precond_grad leaves the shard's OWN sum of h^2 at PSGDK_INFO_HSUMSQ_OFFSET + 4 t: the caller sums it over the members before
 *     psgdk_export_precond_grad / psgdk_apply_update clip.
 * Geometries Q0.5EQ1.5, QEQ and QUAD (round 6: the three that share Pg, the mode Grams and the norm-bound step -- psgd.py:394-419, 367-391,
 * 455-483; begin / finish dispatch on the plan's geometry); the others return PSGDK_ERR_UNSUPPORTED.  Before psgdk_plan_bind.  The one-call
 * psgdk_update_precond_* refuse a plan with shards. */
int psgdk_plan_set_row_shard(psgdk_plan* plan, int t, int64_t global_rows, int64_t row0, int member, int members);
int psgdk_plan_exchange_bytes(const psgdk_plan* plan, size_t* record_bytes);
int psgdk_update_precond_begin(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise, uint64_t seed,
                               uint64_t offset, void* exchange, void* stream);
int psgdk_update_precond_finish(psgdk_plan* plan, int source, float lr, float betaL, float damping, const psgdk_noise* noise, uint64_t seed,
                                uint64_t offset, const void* exchange, const uint8_t* balance_mask, void* stream);
int psgdk_balance_phase(psgdk_plan* plan, const uint8_t* mask, int phase, void* stream);

/* ---- live profiling of the grouped-GEMM launches (bench.py roofline line): with bit 0 of `enable` set, every grouped-GEMM launch
 * carries a start / stop event pair on its own dispatch packet (hipExtLaunchKernelGGL: nothing extra on the stream);
 * psgdk_profile_read synchronises those events and returns the summed launch time (ms) and the launch count since the last reset.
 * Bit 1: the hot-path calls are bracketed too (below; recorded on the stream, ~4 us each). */
int psgdk_profile_enable(psgdk_plan* plan, int enable);
int psgdk_profile_read(psgdk_plan* plan, double* gemm_ms, int64_t* gemm_launches, int reset);
/* of the launches psgdk_profile_read would report: those that carried the fused parameter update (psgdk_precond_grad_apply) -- their time holds
 * the update's memory traffic as well as the product.  Call BEFORE a resetting psgdk_profile_read. */
int psgdk_profile_read_fused(psgdk_plan* plan, double* fused_ms, int64_t* fused_launches);
/* bit 1 of the same switch brackets every hot-path CALL (accumulate, update, precond_grad, apply / export) by an event pair: the summed device
 * time between the first and the last kernel of each call, i.e. the engine's kernel time per step without the host in it. */
int psgdk_profile_read_calls(psgdk_plan* plan, double* call_ms, int64_t* calls, int reset);

/* Kernel-level test / benchmark hooks (psgdk_test_*) are declared in psgdk_test.h: exported by the same library, not part of the
 * drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* PSGDK_H */
