/* eve_moe.h -- C ABI of libeve_moe.so: MI355X (gfx950) kernels for EVE's modality-aware
 * Mixture-of-Experts layer.
 *
 * The reference (ssyze/EVE) has no FFI for this path: the whole layer is stock PyTorch
 * ops composed in models/sharded_moe.py, models/experts.py and
 * models/multiway_transformer.py.  Each entry point below replaces one such composition;
 * the comment on it cites the reference lines it stands in for (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory the caller owns and keeps alive until
 *     the stream has passed the call (borrowed, never freed or retained here);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on it: no hipMalloc/hipFree, no host synchronisation;
 *   - return value 0 = success, negative = EVE_ERR_*; eve_moe_last_error() gives the
 *     message of the calling thread's most recent failure;
 *   - `dtype` is EVE_F32 or EVE_BF16 and names the element type of activations and
 *     expert weights; gate tensors (logits, gates, wg, noise, weights) are always fp32;
 *   - token order is the row-major (batch, position) flattening of [B,T,M] -> [S,M];
 *   - a "slot" is a row e*C + c of the expert-major capacity buffer [E, C, M].
 */
#ifndef EVE_MOE_H_
#define EVE_MOE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVE_MOE_ABI_VERSION 11

enum { EVE_F32 = 0, EVE_BF16 = 1 };

/* modality routing of MOELayer.forward, models/sharded_moe.py:350-364 */
enum {
  EVE_MODE_NONE = 0,   /* num_tokens=None: no modality embedding            (:360-364) */
  EVE_MODE_VL = 1,     /* positions [0,n_vis) + vision_embed, rest + text   (:355-358) */
  EVE_MODE_VISION = 2, /* every token + vision_embed                        (:351-352) */
  EVE_MODE_TEXT = 3    /* every token + text_embed                          (:353-354) */
};

/* expert activation, models/multiway_transformer.py:47 (ACT2FN[hidden_act]) */
enum { EVE_ACT_QUICK_GELU = 0, EVE_ACT_GELU = 1, EVE_ACT_RELU = 2, EVE_ACT_IDENTITY = 3 };

enum {
  EVE_OK = 0,
  EVE_ERR_ARG = -1,       /* bad shape / dtype / null pointer / misalignment */
  EVE_ERR_WORKSPACE = -2, /* workspace too small */
  EVE_ERR_LAUNCH = -3,    /* HIP launch error */
  EVE_ERR_UNSUPPORTED = -4
};

int eve_moe_abi_version(void);
const char* eve_moe_last_error(void);

/* Hex digest (sha256, first 16 hex digits) of the sources this library was built from: every file of
 * eve_amd/csrc/ and this header, as computed by eve_amd/build.py::source_hash.  The Python binding recomputes
 * it from the checkout and refuses a library built from other sources (a stale .so fails loudly instead of
 * being tested in place of the committed code). */
const char* eve_moe_source_hash(void);

/* Capacity of one expert queue: max(ceil_f32(f32(S/E) * f32(2*capacity_factor)), min_capacity).
 * Host-only arithmetic.  Replaces _capacity, models/sharded_moe.py:150-160 as called at :177. */
int eve_capacity(int S, int E, double capacity_factor, int min_capacity);

/* ---------------------------------------------------------------------------- gate */

/* logits[s,e] = sum_m f32(x[s,m]) * wg[e,m],  x[s,:] = round_dtype(h[s,:] + embed(s)).
 * Replaces the modality add (models/sharded_moe.py:350-359) feeding TopKGate.forward's
 * `self.wg(input.float())` (:286-292).  vision_embed/text_embed have h's dtype and may be
 * NULL when mode == EVE_MODE_NONE.  M % 8 == 0. */
int eve_gate_logits_fwd(const void* h, const void* vision_embed, const void* text_embed, const float* wg,
                        float* logits, int S, int T, int n_vis, int mode, int M, int E, int dtype, void* stream);

/* Bytes of scratch eve_gate_top2_fwd needs for S tokens and E experts. */
size_t eve_gate_workspace_bytes(int S, int E);

/* Fused top-2 gating.  Replaces top2gating, models/sharded_moe.py:172-235, with `noise`
 * standing in for the Gumbel draw of :186, producing index-form routing instead of the
 * dense [S,E,C] combine_weights / dispatch_mask:
 *   gates      [S,E]  softmax(logits)                                             (:175)
 *   idx        [S,2]  first / second expert of each token                    (:180,:189)
 *   loc        [S,2]  position in that expert's queue, BEFORE the capacity test (:193-196)
 *   slot       [S,2]  idx*C+loc where the token is dispatched (loc < C and weight != 0),
 *                     else -1                                              (:207-208,:233)
 *   weight     [S,2]  combine weights renormalised after drops (0 if dropped) (:217-223)
 *   slot_token [E*C]  inverse map: token held by each slot, -1 if empty
 *   slot_weight[E*C]  that token's combine weight for the slot
 *   exp_counts [E]    int64 first-choice counts before drops                      (:199)
 *   rows       [E]    min(C, first + second choices): slots [0,rows) of expert e are the
 *                     only ones that can be occupied
 *   l_aux      [1]    load-balance loss                                       (:202-204)
 * `noise` holds the Gumbel variates themselves (noise_is_uniform == 0) or the standard-uniform draws in
 * [0,1) they are made of (noise_is_uniform == 1; transformed inside the kernel with the arithmetic of
 * eve_gumbel_from_uniform, bit for bit).  noise_is_uniform == 2 (EVE_NOISE_DEVICE_PHILOX, since ABI 11): the
 * gate draws the uniform values itself and `noise` points at a 16-byte, 8-byte-aligned device generator state
 * `uint64_t {seed, offset}` instead of a tensor: element s*E+e of the (never materialised) [S,E] draw is word
 * (s*E+e) & 3 of Philox4x32-10 keyed by `seed` at counter `offset + ((s*E+e) >> 2)` (counter words 2, 3 zero),
 * mapped to [0,1) as (word >> 8) * 2^-24 and then transformed as above; the call ADVANCES `offset` by
 * ceil(S*E/4) on the stream (so the state is written although the parameter is const-qualified), which makes
 * consecutive calls -- and replays of a HIP graph that holds the call -- draw fresh noise with no host work.
 * It stands for the draw of models/sharded_moe.py:186 when the caller does not need torch's generator stream.
 * slot_weight needs no clearing by the caller.
 * Routing indices and drops are bit-exact w.r.t. the reference for identical
 * (logits, noise); fp outputs agree to fp32 rounding.  Deterministic (no atomics). */
enum { EVE_NOISE_GUMBEL = 0, EVE_NOISE_UNIFORM = 1, EVE_NOISE_DEVICE_PHILOX = 2 };  /* values of noise_is_uniform */
int eve_gate_top2_fwd(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C,
                      float* gates, int32_t* idx, int32_t* loc, int32_t* slot, float* weight,
                      int32_t* slot_token, float* slot_weight,
                      int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace, size_t workspace_bytes,
                      void* stream);

/* eve_gate_top2_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference; it exists only for
 * those callers.  Same inputs, outputs, kernels and results as eve_gate_top2_fwd, but the gate's scratch
 * (block histograms, in-block ranks, per-block gate sums) lives in a buffer of
 * EVE_GATE_NOALLOC_SCRATCH_BYTES built into the library's code object instead of a caller workspace; the
 * call performs no hipMalloc / hipFree and no allocating library routine.  Each GPU the library runs on
 * holds its own copy of that buffer until the process ends; its address is looked up on the first call on
 * a device (make that call outside graph capture).  Shapes whose eve_gate_workspace_bytes(S, E) exceeds the
 * buffer (about S > 110 000 tokens at E = 32) are rejected with EVE_ERR_WORKSPACE.  All calls on one GPU
 * share the buffer: calls that overlap on one GPU (different streams or threads) race; calls on one stream
 * are ordered and safe.  Nothing in the buffer has to start at zero (every word is written before it is
 * read within a call). */
#define EVE_GATE_NOALLOC_SCRATCH_BYTES (2u << 20)
int eve_gate_top2_fwd_noalloc(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C,
                              float* gates, int32_t* idx, int32_t* loc, int32_t* slot, float* weight,
                              int32_t* slot_token, float* slot_weight, int64_t* exp_counts, int32_t* rows,
                              float* l_aux, void* stream);

/* noise[i] = Gumbel(0,1) variate of the standard-uniform draw uniform[i] in [0,1): the arithmetic of
 * torch.distributions.Gumbel(0,1).rsample -- low + u*(high-low) with low = tiny, high = 1-eps, then
 * -log(-log(.)), each step rounded to fp32 -- that the reference's gumbel_rsample performs
 * (models/sharded_moe.py:70-77, used at :186), in one pass instead of ten elementwise ones. */
int eve_gumbel_from_uniform(const float* uniform, float* noise, long n, void* stream);

/* Backward of eve_gate_top2_fwd w.r.t. logits (what autograd derives for :172-235):
 * d_weight [S,2] and d_l_aux [1] (device scalar) in, dlogits [S,E] out.  Either gradient may be NULL (= zero:
 * nothing downstream used the combine weights / the auxiliary loss). */
int eve_gate_top2_bwd(const float* gates, const int32_t* idx, const int32_t* loc, const float* d_weight,
                      const float* d_l_aux, const int64_t* exp_counts, int S, int E, int C, float* dlogits,
                      void* stream);

/* dwg[e,m] = sum_s dlogits[s,e] * f32(x[s,m]) (x as in eve_gate_logits_fwd): weight gradient
 * of TopKGate.wg, models/sharded_moe.py:271,:292.  workspace: eve_gate_wgrad_workspace_bytes. */
size_t eve_gate_wgrad_workspace_bytes(int S, int M, int E);
int eve_gate_wgrad(const void* h, const void* vision_embed, const void* text_embed, const float* dlogits, float* dwg,
                   int S, int T, int n_vis, int mode, int M, int E, int dtype, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ dispatch / combine */

/* xd[slot,:] = round_dtype(h[tok,:] + embed(tok)) for tok = slot_token[slot] >= 0, else 0.
 * Replaces the modality add plus einsum("sec,sm->ecm", dispatch_mask, reshaped_input),
 * models/sharded_moe.py:350-367.  Writes all E*C rows.  `slot` [S,2] is the forward map of eve_gate_top2_fwd
 * (the inverse of slot_token); with it each token row is read once for both of its slots.  It may be NULL:
 * the rows are then gathered slot by slot from slot_token alone (same result, a token's row read twice). */
int eve_dispatch_fwd(const void* h, const void* vision_embed, const void* text_embed, const int32_t* slot,
                     const int32_t* slot_token, void* xd, int S, int T, int n_vis, int mode, int M, int num_slots,
                     int dtype, void* stream);

/* dh[s,:] = dxd[slot[s,0],:] + dxd[slot[s,1],:] + sum_e dlogits[s,e] * wg[e,:]  (slots < 0 skipped;
 * dlogits may be NULL).  Backward of :350-367 plus the input gradient of the gate's Linear (:292). */
int eve_dispatch_bwd(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh, int S,
                     int M, int E, int dtype, void* stream);

/* dvision[m] = sum over vision tokens of dh[s,m]; dtext likewise (fp32 outputs; either may
 * be NULL).  Gradient of MOELayer.vision_embed / text_embed, models/sharded_moe.py:339-342. */
size_t eve_embed_grad_workspace_bytes(int S, int M);
int eve_embed_grad(const void* dh, float* dvision, float* dtext, int S, int T, int n_vis, int mode, int M, int dtype,
                   void* workspace, size_t workspace_bytes, void* stream);

/* eve_dispatch_bwd and eve_embed_grad in one pass over dh (the backward of the routing front end,
 * models/sharded_moe.py:350-367 as autograd differentiates it): dh as eve_dispatch_bwd (dlogits and wg
 * required), and -- when mode != EVE_MODE_NONE -- dvision / dtext [M] as eve_embed_grad (either may be
 * NULL), stored as embed_dtype (EVE_F32, or EVE_BF16: the fp32 sums rounded once, what a bf16 parameter's
 * gradient holds).  workspace: eve_embed_grad_workspace_bytes(S, M) (unused when no embedding gradient
 * is wanted). */
int eve_route_bwd(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh,
                  void* dvision, void* dtext, int S, int T, int n_vis, int mode, int M, int E, int dtype,
                  int embed_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* out[s,:] = weight[s,0]*y[slot[s,0],:] + weight[s,1]*y[slot[s,1],:]  (fp32 accumulate).
 * Replaces einsum("sec,ecm->sm", combine_weights, expert_output), models/sharded_moe.py:377. */
int eve_combine_fwd(const void* y, const int32_t* slot, const float* weight, void* out, int S, int M, int dtype,
                    void* stream);

/* Backward of eve_combine_fwd: dy[slot,:] = slot_weight[slot]*dout[slot_token[slot],:] (0 for
 * empty slots; all num_slots rows written) and d_weight[s,k] = <dout[s,:], y[slot[s,k],:]>. */
int eve_combine_bwd(const void* dout, const void* y, const int32_t* slot, const int32_t* slot_token,
                    const float* slot_weight, void* dy, float* d_weight, int S, int M, int num_slots, int dtype,
                    void* stream);

/* ----------------------------------------------------------------------- expert FFNs */

/* `flags` of eve_ffn_fwd / eve_ffn_bwd (no counterpart in the reference, whose experts are stock GEMM calls).
 * EVE_FFN_SHARED_GPU: the caller runs other work -- the RCCL exchanges of the pipelined expert-parallel path --
 * on the GPU beside this call's GEMMs, so its launches use one workgroup per output tile, which the dispatcher
 * places on whichever CUs are free; without it forward / dgrad launches are persistent, one workgroup per CU
 * walking the tiles, which assumes the whole chip (a CU held by another kernel would push its workgroup into a
 * second round and double the launch).  A per-call argument: calls on different streams or threads do not
 * affect each other.  Results are identical either way. */
enum { EVE_FFN_SHARED_GPU = 1 };

/* Grouped expert FFN forward.  `x` is [G, R, M]: G groups of R rows; group g is processed
 * by local expert g % E_local (so both the single-GPU layout [E, C, M] and the
 * expert-parallel layout [ep, E_local, C, M] of models/sharded_moe.py:370 are covered).
 *   z = x @ w1[e]^T + b1[e]   [G,R,I]  (the `z` buffer is what eve_ffn_bwd needs of it; NULL = do not store, inference)
 *   a = act(z)                [G,R,I]  (saved for backward)
 *   y = a @ w2[e]^T + b2[e]   [G,R,M]
 * The `z` buffer is OPAQUE to the caller: pass it to eve_ffn_bwd of the same shape and dtype unchanged.  The fp32 path
 * (and bf16 widths outside the MFMA tiling) stores the pre-activation itself; the bf16 MFMA path stores act'(z), the
 * only thing the backward pass uses z for (evaluated in the forward epilogue from the sigmoid it computes anyway,
 * so that the backward epilogue is one multiply per element).
 * w1 [E_local,I,M], b1 [E_local,I], w2 [E_local,M,I], b2 [E_local,M] in `dtype`.
 * rows[g] (may be NULL = R) limits the rows of group g that are computed; rows beyond it
 * are left untouched.  Replaces Experts.forward (models/experts.py:18-28) over
 * MultiwayTransformerMLP.forward (models/multiway_transformer.py:51-55). */
int eve_ffn_fwd(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z, void* a,
                void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act, int dtype,
                int flags, void* stream);

/* Grouped expert FFN backward.  In: dy [G,R,M] and the tensors of the forward.  Out:
 *   dz  [G,R,I]  scratch (gradient at the pre-activation), dx [G,R,M],
 *   dw1 [E_local,I,M], db1 [E_local,I], dw2 [E_local,M,I], db2 [E_local,M]  (overwritten,
 *   summed over the groups that share an expert; `dtype` elements, fp32 accumulation).
 * Rows >= rows[g] contribute nothing and their dx/dz rows are left untouched. */
int eve_ffn_bwd(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2,
                void* dz, void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G,
                int E_local, int R, int M, int I, int act, int dtype, int flags, void* workspace,
                size_t workspace_bytes, void* stream);

/* Scratch for eve_ffn_bwd (256-byte aligned device memory, contents irrelevant).  bf16 MFMA path: fp32 partial
 * column sums of dz, taken in the epilogue of the GEMM that produces dz and reduced in a fixed order to db1 (the
 * bias gradients are formed outside the weight-gradient GEMMs), plus -- when the two weight-gradient GEMMs are split
 * along K because few local experts (expert parallelism, the two-group modality split) leave too few output tiles
 * for the CUs -- the slices' fp32 partials, summed in a fixed order.  0 for the fp32 path. */
size_t eve_ffn_bwd_workspace_bytes(int G, int E_local, int R, int M, int I, int dtype);

/* ------------------------------------------------------------- the whole layer in one call per direction
 *
 * MOELayer.forward on ONE rank (ep_size == 1: every expert local), models/sharded_moe.py:345-381 over
 * TopKGate.forward (:283-297), top2gating (:172-235) and Experts.forward (models/experts.py:18-28), and what
 * autograd derives from them for the backward.  eve_moe_fwd / eve_moe_bwd enqueue exactly the kernels of the
 * per-stage entry points above, in the same order with the same arguments (results are bit-identical to calling
 * those one by one); every intermediate lives in two caller-provided buffers, so a host binding pays two foreign
 * calls and a handful of allocations per step instead of about twenty-five of each.
 *
 *   saved      eve_moe_saved_bytes(shape, need_backward): what outlives the forward -- the routing tensors of
 *              eve_gate_top2_fwd (always) and, when a backward will follow, xd / z / a / y.  The caller keeps it
 *              untouched until eve_moe_bwd has run.  eve_moe_saved_offsets gives the byte offset of each tensor
 *              inside it (EVE_MOE_SAVED_*), for callers that want `slot`, `exp_counts`, ... .
 *   workspace  eve_moe_workspace_bytes(shape, pass): scratch of ONE call (logits, gate scratch, dy / dz / dxd,
 *              dlogits, the reductions' partials); contents irrelevant before, garbage after.
 * Both 256-byte aligned device memory.  Expert-parallel layers (ep_size > 1) have a collective between dispatch and
 * the FFNs: they use the two halves further down (eve_moe_route_fwd ... eve_moe_route_bwd) around eve_ffn_fwd/bwd. */
typedef struct eve_moe_shape {
  int S;           /* tokens B*T                                                                       */
  int T;           /* tokens per sample (modality split of mode VL); ignored for EVE_MODE_NONE          */
  int n_vis;       /* vision tokens per sample (MOELayer.num_vision_tokens)                             */
  int mode;        /* EVE_MODE_*                                                                        */
  int M, I;        /* model / expert hidden width                                                       */
  int E;           /* experts (all local)                                                               */
  int C;           /* capacity: eve_capacity(S, E, capacity_factor, min_capacity)                       */
  int act;         /* EVE_ACT_*                                                                         */
  int dtype;       /* activations, expert weights, modality embeddings                                  */
  int embed_dtype; /* element type the modality-embedding GRADIENTS are stored as (eve_route_bwd)       */
} eve_moe_shape;

enum { EVE_MOE_PASS_INFERENCE = 0, EVE_MOE_PASS_FORWARD = 1, EVE_MOE_PASS_BACKWARD = 2,
       EVE_MOE_PASS_ROUTE_FWD = 3, EVE_MOE_PASS_ROUTE_BWD = 4 /* the expert-parallel halves below */ };

/* order of eve_moe_saved_offsets' output; the last four exist only with need_backward (offset (size_t)-1 otherwise) */
enum {
  EVE_MOE_SAVED_GATES = 0,   /* [S,E] f32  */
  EVE_MOE_SAVED_IDX,         /* [S,2] i32  */
  EVE_MOE_SAVED_LOC,         /* [S,2] i32  */
  EVE_MOE_SAVED_SLOT,        /* [S,2] i32  */
  EVE_MOE_SAVED_WEIGHT,      /* [S,2] f32  */
  EVE_MOE_SAVED_SLOT_TOKEN,  /* [E*C] i32  */
  EVE_MOE_SAVED_SLOT_WEIGHT, /* [E*C] f32  */
  EVE_MOE_SAVED_ROWS,        /* [E]   i32  */
  EVE_MOE_SAVED_EXP_COUNTS,  /* [E]   i64  */
  EVE_MOE_SAVED_XD,          /* [E*C,M]    */
  EVE_MOE_SAVED_Z,           /* [E*C,I]    */
  EVE_MOE_SAVED_A,           /* [E*C,I]    */
  EVE_MOE_SAVED_Y,           /* [E*C,M]    */
  EVE_MOE_SAVED_FIELDS
};

size_t eve_moe_saved_bytes(const eve_moe_shape* shape, int need_backward);
size_t eve_moe_workspace_bytes(const eve_moe_shape* shape, int pass);
int eve_moe_saved_offsets(const eve_moe_shape* shape, int need_backward, size_t* offsets, int n_offsets);

/* out [S,M] and l_aux [1] of MOELayer.forward (:345-381): eve_gate_logits_fwd -> eve_gate_top2_fwd ->
 * eve_dispatch_fwd -> eve_ffn_fwd -> eve_combine_fwd.  h [S,M]; vision_embed / text_embed [M] (`dtype`; may be
 * NULL for EVE_MODE_NONE); wg [E,M] fp32; noise [S,E] as in eve_gate_top2_fwd; w1/b1/w2/b2 as in eve_ffn_fwd
 * with E_local = E.  need_backward == 0: inference (z is not stored, the activation buffers live in the
 * workspace, pass EVE_MOE_PASS_INFERENCE sizes it); != 0: EVE_MOE_PASS_FORWARD.
 * exp_counts_host (since ABI 11, NULL = none): a second place where the gate's closing kernel writes exp_counts [E]
 * int64 -- meant for pinned, device-visible HOST memory (hipHostMalloc), so that the `exp_counts.to('cpu')` the
 * reference's layer returns (models/sharded_moe.py:199 via :381, layer.py:93) needs no copy of its own on the
 * stream (inside a captured step: no copy node between the forward's and the backward's kernels); valid for the
 * host once the stream has passed the call.  Any device-writable int64 [E] buffer will do. */
int eve_moe_fwd(const eve_moe_shape* shape, const void* h, const void* vision_embed, const void* text_embed,
                const float* wg, const float* noise, int noise_is_uniform, const void* w1, const void* b1,
                const void* w2, const void* b2, void* out, float* l_aux, int64_t* exp_counts_host, void* saved,
                size_t saved_bytes, int need_backward, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of eve_moe_fwd: eve_combine_bwd -> eve_ffn_bwd -> eve_gate_top2_bwd -> eve_route_bwd ->
 * eve_gate_wgrad.  In: dout [S,M], d_l_aux [1] device scalar (NULL = the auxiliary loss was not used), the
 * forward's inputs and its `saved` buffer.  Out (all overwritten): dh [S,M]; dwg [E,M] fp32 (NULL = not wanted);
 * dvision / dtext [M] as shape->embed_dtype (NULL = not wanted; ignored for modes that do not use the
 * embedding); dw1 / db1 / dw2 / db2 as eve_ffn_bwd.  workspace: EVE_MOE_PASS_BACKWARD. */
int eve_moe_bwd(const eve_moe_shape* shape, const void* dout, const float* d_l_aux, const void* h,
                const void* vision_embed, const void* text_embed, const float* wg, const void* w1, const void* w2,
                const void* saved, size_t saved_bytes, void* dh, float* dwg, void* dvision, void* dtext, void* dw1,
                void* db1, void* dw2, void* db2, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------- the layer's two halves under expert parallelism (ep_size > 1)
 *
 * MOELayer.forward with the all-to-all exchanges of models/sharded_moe.py:368-379 (`_AllToAll`, :84-100) between
 * dispatch and the experts and between the experts and combine: everything BEFORE the first exchange and everything
 * AFTER the second one as one call each per direction, so that an expert-parallel step is about six foreign calls
 * (these four + eve_ffn_fwd + eve_ffn_bwd per chunk of local experts) plus the collectives, instead of one call per
 * stage.  Same kernels, same arguments, same results as the per-stage entry points (and as eve_moe_fwd / eve_moe_bwd
 * around their FFN calls).  `saved` here is the ROUTING part of the blob only, eve_moe_saved_bytes(shape, 0) with the
 * offsets of eve_moe_saved_offsets(shape, 0, ...); xd, y, dy and dxd are the caller's [E*C, M] exchange buffers in the
 * reference's [ep, E_local, C, M] layout (:370).  shape->E is the GLOBAL expert count, shape->I is not used.
 *
 *   eve_moe_route_fwd   eve_gate_logits_fwd -> eve_gate_top2_fwd -> eve_dispatch_fwd: xd [E*C,M], l_aux [1]; workspace
 *                       EVE_MOE_PASS_ROUTE_FWD
 *   eve_moe_finish_fwd  eve_combine_fwd of the expert outputs y [E*C,M] that came back: out [S,M]
 *   eve_moe_finish_bwd  eve_combine_bwd: dy [E*C,M] (to be sent to the experts' owners), dweight [S,2] fp32
 *   eve_moe_route_bwd   eve_gate_top2_bwd -> eve_route_bwd + eve_gate_wgrad on dxd [E*C,M] (the experts' input gradient
 *                       that came back), dweight (from eve_moe_finish_bwd; NULL = none) and d_l_aux [1] (NULL = none):
 *                       dh [S,M], dwg [E,M] fp32 (NULL = not wanted), dvision / dtext [M] as shape->embed_dtype (NULL =
 *                       not wanted); workspace EVE_MOE_PASS_ROUTE_BWD */
int eve_moe_route_fwd(const eve_moe_shape* shape, const void* h, const void* vision_embed, const void* text_embed,
                      const float* wg, const float* noise, int noise_is_uniform, void* xd, float* l_aux, void* saved,
                      size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
int eve_moe_finish_fwd(const eve_moe_shape* shape, const void* y, const void* saved, size_t saved_bytes, void* out,
                       void* stream);
int eve_moe_finish_bwd(const eve_moe_shape* shape, const void* dout, const void* y, const void* saved, size_t saved_bytes,
                       void* dy, float* dweight, void* stream);
int eve_moe_route_bwd(const eve_moe_shape* shape, const void* dxd, const float* dweight, const float* d_l_aux,
                      const void* h, const void* vision_embed, const void* text_embed, const float* wg,
                      const void* saved, size_t saved_bytes, void* dh, float* dwg, void* dvision, void* dtext,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------- hard modality split of the non-MoE layers
 * Reference models/multiway_transformer.py:366-370: vision positions [:, :n_vis] go through
 * norm2_imag + mlp_imag, the last T - n_vis positions through norm2_text + mlp_text, results concatenated
 * (:386-387 / :402-403 are the single-modality cases n_vis = T / n_vis = 0).  The two MLPs run as one
 * eve_ffn_fwd/bwd with two groups on a [2, R, M] buffer: token (b,t), t < n_vis, is row b*n_vis + t of
 * group 0, otherwise row b*(T-n_vis) + (t-n_vis) of group 1; R >= B * max(n_vis, T-n_vis). */

/* xg[group rows] = LayerNorm(h[b,t]) with the token's modality's affine (gamma/beta have h's dtype), fp32
 * statistics, biased variance, eps inside the square root as torch.nn.LayerNorm; mean / rstd [B*T] fp32 are
 * saved for the backward.  M <= 1024. */
int eve_split_ln_fwd(const void* h, const void* gamma_v, const void* beta_v, const void* gamma_t, const void* beta_t,
                     void* xg, float* mean, float* rstd, int B, int T, int n_vis, int R, int M, float eps, int dtype,
                     void* stream);

size_t eve_split_ln_workspace_bytes(int S, int M);

/* Backward of eve_split_ln_fwd: dxg [2,R,M] in; dh [B*T, M] and dparams [4][M] fp32 =
 * (dgamma_v, dbeta_v, dgamma_t, dbeta_t) out (two-stage fixed-order reduction, deterministic). */
int eve_split_ln_bwd(const void* dxg, const void* h, const float* mean, const float* rstd, const void* gamma_v,
                     const void* gamma_t, void* dh, float* dparams, int B, int T, int n_vis, int R, int M, int dtype,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Row permutation between token order [B*T, M] and the group buffer [2, R, M]: to_groups != 0 copies token
 * rows into their group rows, 0 copies group rows back to token order (the reference's slices and
 * torch.cat at :366-370).  It is its own transpose, so each direction is the other's backward. */
int eve_split_permute(const void* src, void* dst, int B, int T, int n_vis, int R, int M, int to_groups, int dtype,
                      void* stream);

/* ------------------------------------------------ layer-scale residual branch of the encoder layer
 * out = residual + (x * gamma) * row_scale[s / T]: `residual + self.drop_path(hidden_states * self.gamma_1)` and
 * `... * self.gamma_2` of MultiwayTransformerEnocderLayer.forward (models/multiway_transformer.py:341, :408) in ONE
 * pass over the rows instead of a multiply, the DropPath multiply and an add.  x, residual, out [S,M]; gamma [M]
 * (NULL: layer scale off); row_scale [S/T] = mask / keep_prob of DropPath per sample (NULL: eval mode or
 * drop_prob 0); all of `dtype`.  The products and the sum are rounded separately in fp32 (the reference's op chain
 * bit for bit in fp32; bf16 rounds once at the end). */
int eve_scale_residual_fwd(const void* x, const void* gamma, const void* row_scale, const void* residual, void* out, int S,
                           int T, int M, int dtype, void* stream);

/* Backward w.r.t. x and gamma (the gradient w.r.t. residual is dout itself): dx = (dout * row_scale) * gamma [S,M],
 * dgamma [M] fp32 = column sums of (dout * row_scale) * x (two-stage fixed-order reduction, deterministic;
 * workspace: eve_scale_residual_workspace_bytes).  gamma NULL: dx = dout * row_scale, no dgamma, no workspace. */
size_t eve_scale_residual_workspace_bytes(int S, int M);
int eve_scale_residual_bwd(const void* dout, const void* x, const void* gamma, const void* row_scale, void* dx,
                           float* dgamma, int S, int T, int M, int dtype, void* workspace, size_t workspace_bytes,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVE_MOE_H_ */
