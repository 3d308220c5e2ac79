// Whole-layer entry points: one C call per direction of the single-rank modality-aware MoE layer
//   forward   modality add + gate logits -> top-2 gating -> dispatch -> expert FFNs -> combine
//   backward  combine' -> expert FFNs' -> routing' (dispatch', gate', gate-Linear input gradient, modality
//             embedding gradients) -> gate weight gradient
// They enqueue exactly the kernels of the per-stage entry points (include/eve_moe.h), in the same order and with
// the same arguments, on the caller's stream; every intermediate lives in two caller-provided blobs: `saved`
// (what the backward needs from the forward) and `workspace` (scratch of one call).  A host binding then pays
// two foreign calls and a handful of allocations per step instead of about twenty-five of each, which is what
// bounds an eagerly issued step (reference: the whole of MOELayer.forward, models/sharded_moe.py:345-381, and
// its autograd backward).
#include "common.h"

namespace eve {

enum SavedField {
  SV_GATES = 0, SV_IDX, SV_LOC, SV_SLOT, SV_WEIGHT, SV_SLOT_TOKEN, SV_SLOT_WEIGHT, SV_ROWS, SV_EXP_COUNTS, SV_XD, SV_Z, SV_A,
  SV_Y, SV_NFIELDS
};

struct Layout {
  size_t off[24];
  size_t total;
};

static inline size_t esize(int dtype) { return dtype == EVE_BF16 ? 2 : 4; }

// saved: routing always; the activation buffers only when a backward will follow (else they are scratch)
static Layout saved_layout(const eve_moe_shape& p, bool need_backward) {
  Layout L{};
  size_t o = 0;
  auto put = [&](int f, size_t bytes) { L.off[f] = o; o = align_up(o + bytes, 256); };
  const size_t S = p.S, E = p.E, C = p.C, M = p.M, I = p.I, b = esize(p.dtype), n = E * C;
  put(SV_GATES, S * E * 4);
  put(SV_IDX, S * 2 * 4);
  put(SV_LOC, S * 2 * 4);
  put(SV_SLOT, S * 2 * 4);
  put(SV_WEIGHT, S * 2 * 4);
  put(SV_SLOT_TOKEN, n * 4);
  put(SV_SLOT_WEIGHT, n * 4);
  put(SV_ROWS, E * 4);
  put(SV_EXP_COUNTS, E * 8);
  if (need_backward) {
    put(SV_XD, n * M * b);
    put(SV_Z, n * I * b);
    put(SV_A, n * I * b);
    put(SV_Y, n * M * b);
  }
  L.total = o;
  return L;
}

enum WsField { WS_LOGITS = 0, WS_GATE, WS_XD, WS_A, WS_Y, WS_DY, WS_DZ, WS_DXD, WS_DLOGITS, WS_DWEIGHT, WS_WGRAD, WS_EMBED, WS_FFN, WS_NFIELDS };

static Layout fwd_ws_layout(const eve_moe_shape& p, bool need_backward) {
  Layout L{};
  size_t o = 0;
  auto put = [&](int f, size_t bytes) { L.off[f] = o; o = align_up(o + bytes, 256); };
  const size_t S = p.S, E = p.E, C = p.C, M = p.M, I = p.I, b = esize(p.dtype), n = E * C;
  put(WS_LOGITS, S * E * 4);
  put(WS_GATE, eve_gate_workspace_bytes(p.S, p.E));
  if (!need_backward) {
    put(WS_XD, n * M * b);
    put(WS_A, n * I * b);
    put(WS_Y, n * M * b);
  }
  L.total = o;
  return L;
}

static Layout bwd_ws_layout(const eve_moe_shape& p) {
  Layout L{};
  size_t o = 0;
  auto put = [&](int f, size_t bytes) { L.off[f] = o; o = align_up(o + bytes, 256); };
  const size_t S = p.S, E = p.E, C = p.C, M = p.M, I = p.I, b = esize(p.dtype), n = E * C;
  put(WS_DY, n * M * b);
  put(WS_DZ, n * I * b);
  put(WS_DXD, n * M * b);
  put(WS_DLOGITS, S * E * 4);
  put(WS_DWEIGHT, S * 2 * 4);
  put(WS_WGRAD, eve_gate_wgrad_workspace_bytes(p.S, p.M, p.E));
  put(WS_EMBED, eve_embed_grad_workspace_bytes(p.S, p.M));
  put(WS_FFN, eve_ffn_bwd_workspace_bytes(p.E, p.E, p.C, p.M, p.I, p.dtype));
  L.total = o;
  return L;
}

// expert-parallel halves: scratch of eve_moe_route_fwd (logits, gate scratch) and of eve_moe_route_bwd (dlogits and
// the partials of the closing reductions)
static Layout route_fwd_ws_layout(const eve_moe_shape& p) {
  Layout L{};
  size_t o = 0;
  auto put = [&](int f, size_t bytes) { L.off[f] = o; o = align_up(o + bytes, 256); };
  put(WS_LOGITS, (size_t)p.S * p.E * 4);
  put(WS_GATE, eve_gate_workspace_bytes(p.S, p.E));
  L.total = o;
  return L;
}
static Layout route_bwd_ws_layout(const eve_moe_shape& p) {
  Layout L{};
  size_t o = 0;
  auto put = [&](int f, size_t bytes) { L.off[f] = o; o = align_up(o + bytes, 256); };
  put(WS_DLOGITS, (size_t)p.S * p.E * 4);
  put(WS_WGRAD, eve_gate_wgrad_workspace_bytes(p.S, p.M, p.E));
  put(WS_EMBED, eve_embed_grad_workspace_bytes(p.S, p.M));
  L.total = o;
  return L;
}

static int check_shape(const char* who, const eve_moe_shape* p) {
  EVE_REQUIRE(p, EVE_ERR_ARG, "%s: null shape", who);
  EVE_REQUIRE(p->S > 0 && p->M > 0 && p->I > 0 && p->E > 0 && p->C > 0, EVE_ERR_ARG, "%s: bad shape S=%d M=%d I=%d E=%d C=%d",
              who, p->S, p->M, p->I, p->E, p->C);
  EVE_REQUIRE(p->E <= 32, EVE_ERR_UNSUPPORTED, "%s: E=%d > 32 (use the per-stage entry points)", who, p->E);
  EVE_REQUIRE(p->dtype == EVE_F32 || p->dtype == EVE_BF16, EVE_ERR_ARG, "%s: bad dtype %d", who, p->dtype);
  EVE_REQUIRE(p->mode >= EVE_MODE_NONE && p->mode <= EVE_MODE_TEXT, EVE_ERR_ARG, "%s: bad mode %d", who, p->mode);
  if (p->mode != EVE_MODE_NONE)
    EVE_REQUIRE(p->T > 0 && p->S % p->T == 0 && p->n_vis >= 0, EVE_ERR_ARG, "%s: S=%d not a multiple of T=%d", who, p->S, p->T);
  EVE_REQUIRE((long long)p->E * p->C < (1ll << 31), EVE_ERR_ARG, "%s: E*C overflows int32", who);
  return EVE_OK;
}

}  // namespace eve

using namespace eve;

extern "C" size_t eve_moe_saved_bytes(const eve_moe_shape* p, int need_backward) {
  if (!p || p->S <= 0 || p->E <= 0 || p->C <= 0 || p->M <= 0 || p->I <= 0) return 0;
  return saved_layout(*p, need_backward != 0).total;
}

extern "C" size_t eve_moe_workspace_bytes(const eve_moe_shape* p, int pass) {
  if (!p || p->S <= 0 || p->E <= 0 || p->C <= 0 || p->M <= 0 || p->I <= 0) return 0;
  switch (pass) {
    case EVE_MOE_PASS_INFERENCE: return fwd_ws_layout(*p, false).total;
    case EVE_MOE_PASS_FORWARD: return fwd_ws_layout(*p, true).total;
    case EVE_MOE_PASS_BACKWARD: return bwd_ws_layout(*p).total;
    case EVE_MOE_PASS_ROUTE_FWD: return route_fwd_ws_layout(*p).total;
    case EVE_MOE_PASS_ROUTE_BWD: return route_bwd_ws_layout(*p).total;
    default: return 0;
  }
}

extern "C" int eve_moe_saved_offsets(const eve_moe_shape* p, int need_backward, size_t* offsets, int n) {
  if (int rc = check_shape("eve_moe_saved_offsets", p)) return rc;
  EVE_REQUIRE(offsets && n >= SV_NFIELDS, EVE_ERR_ARG, "eve_moe_saved_offsets: need room for %d offsets", (int)SV_NFIELDS);
  const Layout L = saved_layout(*p, need_backward != 0);
  for (int f = 0; f < SV_NFIELDS; ++f) offsets[f] = (need_backward || f < SV_XD) ? L.off[f] : (size_t)-1;
  return EVE_OK;
}

extern "C" int eve_moe_fwd(const eve_moe_shape* p, const void* h, const void* vision_embed, const void* text_embed,
                           const float* wg, const float* noise, int noise_is_uniform, const void* w1, const void* b1,
                           const void* w2, const void* b2, void* out, float* l_aux, int64_t* exp_counts_host,
                           void* saved, size_t saved_bytes, int need_backward, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (int rc = check_shape("eve_moe_fwd", p)) return rc;
  EVE_REQUIRE(h && wg && noise && w1 && b1 && w2 && b2 && out && l_aux && saved && workspace, EVE_ERR_ARG, "eve_moe_fwd: null pointer");
  const bool nb = need_backward != 0;
  const Layout sv = saved_layout(*p, nb), ws = fwd_ws_layout(*p, nb);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_fwd: saved %zu < %zu (eve_moe_saved_bytes)", saved_bytes, sv.total);
  EVE_REQUIRE(workspace_bytes >= ws.total, EVE_ERR_WORKSPACE, "eve_moe_fwd: workspace %zu < %zu (eve_moe_workspace_bytes)", workspace_bytes, ws.total);
  EVE_REQUIRE((((uintptr_t)saved | (uintptr_t)workspace) & 255) == 0, EVE_ERR_ARG, "eve_moe_fwd: saved / workspace must be 256-byte aligned");
  char* s = static_cast<char*>(saved);
  char* w = static_cast<char*>(workspace);
  float* logits = reinterpret_cast<float*>(w + ws.off[WS_LOGITS]);
  float* gates = reinterpret_cast<float*>(s + sv.off[SV_GATES]);
  int32_t* idx = reinterpret_cast<int32_t*>(s + sv.off[SV_IDX]);
  int32_t* loc = reinterpret_cast<int32_t*>(s + sv.off[SV_LOC]);
  int32_t* slot = reinterpret_cast<int32_t*>(s + sv.off[SV_SLOT]);
  float* weight = reinterpret_cast<float*>(s + sv.off[SV_WEIGHT]);
  int32_t* slot_token = reinterpret_cast<int32_t*>(s + sv.off[SV_SLOT_TOKEN]);
  float* slot_weight = reinterpret_cast<float*>(s + sv.off[SV_SLOT_WEIGHT]);
  int32_t* rows = reinterpret_cast<int32_t*>(s + sv.off[SV_ROWS]);
  int64_t* exp_counts = reinterpret_cast<int64_t*>(s + sv.off[SV_EXP_COUNTS]);
  void* xd = nb ? s + sv.off[SV_XD] : w + ws.off[WS_XD];
  void* z = nb ? s + sv.off[SV_Z] : nullptr;
  void* a = nb ? s + sv.off[SV_A] : w + ws.off[WS_A];
  void* y = nb ? s + sv.off[SV_Y] : w + ws.off[WS_Y];
  const int n = p->E * p->C;
  if (eve::gate_logits_top2_fusable(p->S, p->M, p->E, p->dtype)) {  // logits + selection in one launch, then the assignment
    if (int rc = eve::gate_logits_top2_fused(h, vision_embed, text_embed, wg, noise, noise_is_uniform, p->S, p->T, p->n_vis,
                                             p->mode, p->M, p->E, p->C, gates, idx, loc, slot, weight, slot_token, slot_weight,
                                             exp_counts, rows, l_aux, w + ws.off[WS_GATE],
                                             eve_gate_workspace_bytes(p->S, p->E), stream, exp_counts_host)) return rc;
  } else {
    if (int rc = eve_gate_logits_fwd(h, vision_embed, text_embed, wg, logits, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->dtype, stream)) return rc;
    if (int rc = eve::gate_top2_fwd_mirrored(logits, noise, noise_is_uniform, p->S, p->E, p->C, gates, idx, loc, slot, weight, slot_token,
                                             slot_weight, exp_counts, rows, l_aux, w + ws.off[WS_GATE],
                                             eve_gate_workspace_bytes(p->S, p->E), stream, exp_counts_host)) return rc;
  }
  if (int rc = eve_dispatch_fwd(h, vision_embed, text_embed, slot, slot_token, xd, p->S, p->T, p->n_vis, p->mode, p->M, n, p->dtype, stream)) return rc;
  if (int rc = eve_ffn_fwd(xd, w1, b1, w2, b2, z, a, y, rows, p->E, p->E, p->C, p->M, p->I, p->act, p->dtype, 0, stream)) return rc;
  return eve_combine_fwd(y, slot, weight, out, p->S, p->M, p->dtype, stream);
}

extern "C" int eve_moe_bwd(const eve_moe_shape* p, const void* dout, const float* d_l_aux, const void* h, const void* vision_embed,
                           const void* text_embed, const float* wg, const void* w1, const void* w2, const void* saved,
                           size_t saved_bytes, void* dh, float* dwg, void* dvision, void* dtext, void* dw1, void* db1, void* dw2,
                           void* db2, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_shape("eve_moe_bwd", p)) return rc;
  EVE_REQUIRE(dout && h && wg && w1 && w2 && saved && dh && dw1 && db1 && dw2 && db2 && workspace, EVE_ERR_ARG, "eve_moe_bwd: null pointer");
  EVE_REQUIRE(p->embed_dtype == EVE_F32 || p->embed_dtype == EVE_BF16, EVE_ERR_ARG, "eve_moe_bwd: bad embed_dtype %d", p->embed_dtype);
  const Layout sv = saved_layout(*p, true), ws = bwd_ws_layout(*p);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_bwd: saved %zu < %zu (forward must have run with need_backward)", saved_bytes, sv.total);
  EVE_REQUIRE(workspace_bytes >= ws.total, EVE_ERR_WORKSPACE, "eve_moe_bwd: workspace %zu < %zu (eve_moe_workspace_bytes)", workspace_bytes, ws.total);
  EVE_REQUIRE((((uintptr_t)saved | (uintptr_t)workspace) & 255) == 0, EVE_ERR_ARG, "eve_moe_bwd: saved / workspace must be 256-byte aligned");
  const char* s = static_cast<const char*>(saved);
  char* w = static_cast<char*>(workspace);
  const float* gates = reinterpret_cast<const float*>(s + sv.off[SV_GATES]);
  const int32_t* idx = reinterpret_cast<const int32_t*>(s + sv.off[SV_IDX]);
  const int32_t* loc = reinterpret_cast<const int32_t*>(s + sv.off[SV_LOC]);
  const int32_t* slot = reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT]);
  const int32_t* slot_token = reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT_TOKEN]);
  const float* slot_weight = reinterpret_cast<const float*>(s + sv.off[SV_SLOT_WEIGHT]);
  const int32_t* rows = reinterpret_cast<const int32_t*>(s + sv.off[SV_ROWS]);
  const int64_t* exp_counts = reinterpret_cast<const int64_t*>(s + sv.off[SV_EXP_COUNTS]);
  const void* xd = s + sv.off[SV_XD];
  const void* z = s + sv.off[SV_Z];
  const void* a = s + sv.off[SV_A];
  const void* y = s + sv.off[SV_Y];
  void* dy = w + ws.off[WS_DY];
  void* dz = w + ws.off[WS_DZ];
  void* dxd = w + ws.off[WS_DXD];
  float* dlogits = reinterpret_cast<float*>(w + ws.off[WS_DLOGITS]);
  float* dweight = reinterpret_cast<float*>(w + ws.off[WS_DWEIGHT]);
  const int n = p->E * p->C;
  // (the gate's backward needs nothing but d weight: where the shapes allow, the combine backward forms d logits in
  // the same launch instead of a 5 us launch of its own between the expert GEMMs and the route backward)
  const bool gate_in_combine = combine_gate_bwd_fusable(dout, y, slot, dy, idx, loc, p->M, p->E, p->dtype);
  if (gate_in_combine) {
    if (int rc = combine_bwd_gate_bwd(dout, y, slot, slot_token, slot_weight, dy, dweight, gates, idx, loc, d_l_aux, exp_counts,
                                      dlogits, p->S, p->M, p->E, p->C, n, p->dtype, stream)) return rc;
  } else {
    if (int rc = eve_combine_bwd(dout, y, slot, slot_token, slot_weight, dy, dweight, p->S, p->M, n, p->dtype, stream)) return rc;
  }
  if (int rc = eve_ffn_bwd(dy, xd, z, a, w1, w2, dz, dxd, dw1, db1, dw2, db2, rows, p->E, p->E, p->C, p->M, p->I, p->act, p->dtype, 0,
                           w + ws.off[WS_FFN], eve_ffn_bwd_workspace_bytes(p->E, p->E, p->C, p->M, p->I, p->dtype), stream)) return rc;
  if (!gate_in_combine) {
    if (int rc = eve_gate_top2_bwd(gates, idx, loc, dweight, d_l_aux, exp_counts, p->S, p->E, p->C, dlogits, stream)) return rc;
  }
  const bool want_v = dvision && (p->mode == EVE_MODE_VL || p->mode == EVE_MODE_VISION);
  const bool want_t = dtext && (p->mode == EVE_MODE_VL || p->mode == EVE_MODE_TEXT);
  // route backward and gate weight gradient are independent of each other: one launch for both, one for their closing
  // sums, where the shape takes the kernels that combine (bit-identical to the two calls below)
  if (dwg && route_wgrad_fusable(dxd, slot, dh, h, p->M, p->E, p->dtype))
    return route_bwd_gate_wgrad(dxd, slot, dlogits, wg, dh, want_v ? dvision : nullptr, want_t ? dtext : nullptr, h, vision_embed,
                                text_embed, dwg, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->embed_dtype, w + ws.off[WS_EMBED],
                                eve_embed_grad_workspace_bytes(p->S, p->M), w + ws.off[WS_WGRAD],
                                eve_gate_wgrad_workspace_bytes(p->S, p->M, p->E), stream);
  if (int rc = eve_route_bwd(dxd, slot, dlogits, wg, dh, want_v ? dvision : nullptr, want_t ? dtext : nullptr, p->S, p->T, p->n_vis, p->mode,
                             p->M, p->E, p->dtype, p->embed_dtype, w + ws.off[WS_EMBED], eve_embed_grad_workspace_bytes(p->S, p->M), stream)) return rc;
  if (dwg) {
    if (int rc = eve_gate_wgrad(h, vision_embed, text_embed, dlogits, dwg, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->dtype,
                                w + ws.off[WS_WGRAD], eve_gate_wgrad_workspace_bytes(p->S, p->M, p->E), stream)) return rc;
  }
  return EVE_OK;
}

// ------------------------------------------------------------------ expert-parallel halves (ep_size > 1)
// The layer of one rank with the exchange and the expert FFNs cut out (they sit between the halves, on the host
// side: RCCL all-to-all, eve_ffn_fwd/bwd on the rows received): the same kernels with the same arguments as
// eve_moe_fwd / eve_moe_bwd enqueue around their FFN calls.  `saved` is the routing part of the blob
// (eve_moe_saved_bytes(shape, 0)); xd / y / dy / dxd are the caller's [E*C, M] exchange buffers.

extern "C" int eve_moe_route_fwd(const eve_moe_shape* p, const void* h, const void* vision_embed, const void* text_embed,
                                 const float* wg, const float* noise, int noise_is_uniform, void* xd, float* l_aux,
                                 void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_shape("eve_moe_route_fwd", p)) return rc;
  EVE_REQUIRE(h && wg && noise && xd && l_aux && saved && workspace, EVE_ERR_ARG, "eve_moe_route_fwd: null pointer");
  const Layout sv = saved_layout(*p, false), ws = route_fwd_ws_layout(*p);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_route_fwd: saved %zu < %zu (eve_moe_saved_bytes(shape, 0))", saved_bytes, sv.total);
  EVE_REQUIRE(workspace_bytes >= ws.total, EVE_ERR_WORKSPACE, "eve_moe_route_fwd: workspace %zu < %zu (EVE_MOE_PASS_ROUTE_FWD)", workspace_bytes, ws.total);
  EVE_REQUIRE((((uintptr_t)saved | (uintptr_t)workspace) & 255) == 0, EVE_ERR_ARG, "eve_moe_route_fwd: saved / workspace must be 256-byte aligned");
  char* s = static_cast<char*>(saved);
  char* w = static_cast<char*>(workspace);
  float* logits = reinterpret_cast<float*>(w + ws.off[WS_LOGITS]);
  int32_t* slot = reinterpret_cast<int32_t*>(s + sv.off[SV_SLOT]);
  int32_t* slot_token = reinterpret_cast<int32_t*>(s + sv.off[SV_SLOT_TOKEN]);
  float* gates = reinterpret_cast<float*>(s + sv.off[SV_GATES]);
  int32_t* idx = reinterpret_cast<int32_t*>(s + sv.off[SV_IDX]);
  int32_t* loc = reinterpret_cast<int32_t*>(s + sv.off[SV_LOC]);
  float* weight = reinterpret_cast<float*>(s + sv.off[SV_WEIGHT]);
  float* slot_weight = reinterpret_cast<float*>(s + sv.off[SV_SLOT_WEIGHT]);
  int64_t* exp_counts = reinterpret_cast<int64_t*>(s + sv.off[SV_EXP_COUNTS]);
  int32_t* rows = reinterpret_cast<int32_t*>(s + sv.off[SV_ROWS]);
  if (eve::gate_logits_top2_fusable(p->S, p->M, p->E, p->dtype)) {  // as in eve_moe_fwd
    if (int rc = eve::gate_logits_top2_fused(h, vision_embed, text_embed, wg, noise, noise_is_uniform, p->S, p->T, p->n_vis,
                                             p->mode, p->M, p->E, p->C, gates, idx, loc, slot, weight, slot_token, slot_weight,
                                             exp_counts, rows, l_aux, w + ws.off[WS_GATE],
                                             eve_gate_workspace_bytes(p->S, p->E), stream, nullptr)) return rc;
  } else {
    if (int rc = eve_gate_logits_fwd(h, vision_embed, text_embed, wg, logits, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->dtype, stream)) return rc;
    if (int rc = eve_gate_top2_fwd(logits, noise, noise_is_uniform, p->S, p->E, p->C, gates, idx, loc, slot, weight, slot_token,
                                   slot_weight, exp_counts, rows, l_aux, w + ws.off[WS_GATE],
                                   eve_gate_workspace_bytes(p->S, p->E), stream)) return rc;
  }
  return eve_dispatch_fwd(h, vision_embed, text_embed, slot, slot_token, xd, p->S, p->T, p->n_vis, p->mode, p->M, p->E * p->C, p->dtype, stream);
}

extern "C" int eve_moe_finish_fwd(const eve_moe_shape* p, const void* y, const void* saved, size_t saved_bytes, void* out,
                                  void* stream) {
  if (int rc = check_shape("eve_moe_finish_fwd", p)) return rc;
  EVE_REQUIRE(y && saved && out, EVE_ERR_ARG, "eve_moe_finish_fwd: null pointer");
  const Layout sv = saved_layout(*p, false);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_finish_fwd: saved %zu < %zu", saved_bytes, sv.total);
  const char* s = static_cast<const char*>(saved);
  return eve_combine_fwd(y, reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT]), reinterpret_cast<const float*>(s + sv.off[SV_WEIGHT]),
                         out, p->S, p->M, p->dtype, stream);
}

extern "C" int eve_moe_finish_bwd(const eve_moe_shape* p, const void* dout, const void* y, const void* saved, size_t saved_bytes,
                                  void* dy, float* dweight, void* stream) {
  if (int rc = check_shape("eve_moe_finish_bwd", p)) return rc;
  EVE_REQUIRE(dout && y && saved && dy && dweight, EVE_ERR_ARG, "eve_moe_finish_bwd: null pointer");
  const Layout sv = saved_layout(*p, false);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_finish_bwd: saved %zu < %zu", saved_bytes, sv.total);
  const char* s = static_cast<const char*>(saved);
  return eve_combine_bwd(dout, y, reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT]),
                         reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT_TOKEN]),
                         reinterpret_cast<const float*>(s + sv.off[SV_SLOT_WEIGHT]), dy, dweight, p->S, p->M, p->E * p->C, p->dtype, stream);
}

extern "C" int eve_moe_route_bwd(const eve_moe_shape* p, const void* dxd, const float* dweight, const float* d_l_aux, const void* h,
                                 const void* vision_embed, const void* text_embed, const float* wg, const void* saved,
                                 size_t saved_bytes, void* dh, float* dwg, void* dvision, void* dtext, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (int rc = check_shape("eve_moe_route_bwd", p)) return rc;
  EVE_REQUIRE(dxd && h && wg && saved && dh && workspace, EVE_ERR_ARG, "eve_moe_route_bwd: null pointer");
  EVE_REQUIRE(p->embed_dtype == EVE_F32 || p->embed_dtype == EVE_BF16, EVE_ERR_ARG, "eve_moe_route_bwd: bad embed_dtype %d", p->embed_dtype);
  const Layout sv = saved_layout(*p, false), ws = route_bwd_ws_layout(*p);
  EVE_REQUIRE(saved_bytes >= sv.total, EVE_ERR_WORKSPACE, "eve_moe_route_bwd: saved %zu < %zu", saved_bytes, sv.total);
  EVE_REQUIRE(workspace_bytes >= ws.total, EVE_ERR_WORKSPACE, "eve_moe_route_bwd: workspace %zu < %zu (EVE_MOE_PASS_ROUTE_BWD)", workspace_bytes, ws.total);
  EVE_REQUIRE((((uintptr_t)saved | (uintptr_t)workspace) & 255) == 0, EVE_ERR_ARG, "eve_moe_route_bwd: saved / workspace must be 256-byte aligned");
  const char* s = static_cast<const char*>(saved);
  char* w = static_cast<char*>(workspace);
  const int32_t* slot = reinterpret_cast<const int32_t*>(s + sv.off[SV_SLOT]);
  float* dlogits = reinterpret_cast<float*>(w + ws.off[WS_DLOGITS]);
  if (int rc = eve_gate_top2_bwd(reinterpret_cast<const float*>(s + sv.off[SV_GATES]), reinterpret_cast<const int32_t*>(s + sv.off[SV_IDX]),
                                 reinterpret_cast<const int32_t*>(s + sv.off[SV_LOC]), dweight, d_l_aux,
                                 reinterpret_cast<const int64_t*>(s + sv.off[SV_EXP_COUNTS]), p->S, p->E, p->C, dlogits, stream)) return rc;
  const bool want_v = dvision && (p->mode == EVE_MODE_VL || p->mode == EVE_MODE_VISION);
  const bool want_t = dtext && (p->mode == EVE_MODE_VL || p->mode == EVE_MODE_TEXT);
  if (dwg && route_wgrad_fusable(dxd, slot, dh, h, p->M, p->E, p->dtype))
    return route_bwd_gate_wgrad(dxd, slot, dlogits, wg, dh, want_v ? dvision : nullptr, want_t ? dtext : nullptr, h, vision_embed,
                                text_embed, dwg, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->embed_dtype, w + ws.off[WS_EMBED],
                                eve_embed_grad_workspace_bytes(p->S, p->M), w + ws.off[WS_WGRAD],
                                eve_gate_wgrad_workspace_bytes(p->S, p->M, p->E), stream);
  if (int rc = eve_route_bwd(dxd, slot, dlogits, wg, dh, want_v ? dvision : nullptr, want_t ? dtext : nullptr, p->S, p->T, p->n_vis, p->mode,
                             p->M, p->E, p->dtype, p->embed_dtype, w + ws.off[WS_EMBED], eve_embed_grad_workspace_bytes(p->S, p->M), stream)) return rc;
  if (dwg) {
    if (int rc = eve_gate_wgrad(h, vision_embed, text_embed, dlogits, dwg, p->S, p->T, p->n_vis, p->mode, p->M, p->E, p->dtype,
                                w + ws.off[WS_WGRAD], eve_gate_wgrad_workspace_bytes(p->S, p->M, p->E), stream)) return rc;
  }
  return EVE_OK;
}
