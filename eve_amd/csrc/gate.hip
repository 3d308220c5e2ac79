// Gate kernels: modality-add + gate logits, fused top-2 gating (softmax, two argmaxes,
// ordered per-expert queue positions, capacity drop, renormalised weights, l_aux), and
// their backward.  gfx950, wave64.  Reference semantics: models/sharded_moe.py:172-235,
// :283-297, :350-364 (see include/eve_moe.h for the per-entry-point citations).
#include "common.h"

namespace eve {

constexpr float kEps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps, sharded_moe.py:221
constexpr int kGateBlock = 256;                  // tokens per block in the select / assign passes

// =============================================================== logits = x @ wg^T
// One wave handles TOK tokens at a time; lane l owns 4-element granules l, l+64, ... of
// the M axis (NG granules per lane), so x stays in registers while the wave walks the E
// rows of wg (L2-resident, read once per TOK tokens).
template <typename T, int NG, int TOK>
__global__ __launch_bounds__(256) void gate_logits_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                          const T* __restrict__ temb, const float* __restrict__ wg,
                                                          float* __restrict__ logits, int S, int T_, int n_vis,
                                                          int mode, int M, int E) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ngran = M >> 2;
  for (int s0 = wave * TOK; s0 < S; s0 += nwaves * TOK) {
    float x[TOK][NG][4];
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const int s = s0 + t;
      const bool live = s < S;
      const T* emb = nullptr;
      if (mode != EVE_MODE_NONE) emb = token_modality(live ? s : 0, T_, n_vis, mode) ? temb : vemb;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int gr = lane + g * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[t][g][j] = 0.f;
        if (live && gr < ngran) {
          const T* p = h + (size_t)s * M + gr * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = to_f32<T>(p[j]);
            if (emb) v = to_f32<T>(from_f32<T>(v + to_f32<T>(emb[gr * 4 + j])));  // round like the dtype add
            x[t][g][j] = v;
          }
        }
      }
    }
    for (int e = 0; e < E; ++e) {
      float acc[TOK];
#pragma unroll
      for (int t = 0; t < TOK; ++t) acc[t] = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int gr = lane + g * 64;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < ngran) w = *reinterpret_cast<const float4*>(wg + (size_t)e * M + gr * 4);
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
          acc[t] = fmaf(x[t][g][0], w.x, acc[t]);
          acc[t] = fmaf(x[t][g][1], w.y, acc[t]);
          acc[t] = fmaf(x[t][g][2], w.z, acc[t]);
          acc[t] = fmaf(x[t][g][3], w.w, acc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const float r = wave_sum(acc[t]);
        if (lane == 0 && s0 + t < S) logits[(size_t)(s0 + t) * E + e] = r;
      }
    }
  }
}

// ===================================================================== top-2 gating
// Pass 1 (select): one thread per token.  softmax, first / second choice, rank of the token
// among earlier tokens OF ITS BLOCK with the same choice (wave ballots + LDS), per-block
// histograms and per-block column sums of the gates.  Also clears slot_token.
// Workspace layout (ints unless noted), nblk = ceil(S / kGateBlock):
//   hist   [nblk][2][E]   block histograms of first / second choices
//   boff   [nblk][2][E]   exclusive prefix of hist over blocks (pass 2)
//   rank   [S][2]         in-block rank
//   gsum   [nblk][E]      float, per-block column sums of gates
//   tot    [2][E]         totals
__global__ __launch_bounds__(kGateBlock) void gate_select_kernel(
    const float* __restrict__ logits, const float* __restrict__ noise, int S, int E, float* __restrict__ gates,
    int32_t* __restrict__ idx, int32_t* __restrict__ hist, int32_t* __restrict__ rank, float* __restrict__ gsum,
    int32_t* __restrict__ slot_token, int num_slots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nw = kGateBlock / 64;
  int32_t* wcnt = reinterpret_cast<int32_t*>(smem);                  // [nw][2][E]
  float* gcol = reinterpret_cast<float*>(wcnt + nw * 2 * E);         // [nw][E] wave partial sums

  // clear the inverse map (consumed by pass 3, which runs after this kernel completes)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_slots; i += gridDim.x * blockDim.x) slot_token[i] = -1;

  const int s = blockIdx.x * kGateBlock + threadIdx.x;
  const bool live = s < S;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const float* row = logits + (size_t)(live ? s : 0) * E;
  const float* nrow = noise + (size_t)(live ? s : 0) * E;

  // softmax exactly as exp(x - max) / sum; first choice = argmax of the GATES (ties -> lowest e)
  float mx = -INFINITY;
  for (int e = 0; e < E; ++e) mx = fmaxf(mx, row[e]);
  float sum = 0.f;
  for (int e = 0; e < E; ++e) sum += expf(row[e] - mx);
  int i1 = 0;
  float best = -INFINITY;
  for (int e = 0; e < E; ++e) {
    const float g = expf(row[e] - mx) / sum;
    if (live) gates[(size_t)s * E + e] = g;
    if (g > best) { best = g; i1 = e; }
  }
  // second choice = argmax of logits + noise with the first choice masked out
  int i2 = -1;
  float best2 = -INFINITY;
  for (int e = 0; e < E; ++e) {
    if (e == i1) continue;
    const float v = row[e] + nrow[e];
    if (i2 < 0 || v > best2) { best2 = v; i2 = e; }
  }
  if (i2 < 0) i2 = 0;  // E == 1: masked_fill leaves all -inf, argmax -> 0
  if (!live) { i1 = -1; i2 = -1; }

  // ranks inside the wave, counts per wave
  int r1 = 0, r2 = 0;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int e = 0; e < E; ++e) {
    const unsigned long long m1 = __ballot(i1 == e);
    const unsigned long long m2 = __ballot(i2 == e);
    if (i1 == e) r1 = __popcll(m1 & lt);
    if (i2 == e) r2 = __popcll(m2 & lt);
    if (lane == 0) {
      wcnt[(wave * 2 + 0) * E + e] = __popcll(m1);
      wcnt[(wave * 2 + 1) * E + e] = __popcll(m2);
    }
    // column sum of gates over the wave's tokens (fixed order -> deterministic)
    float g = live ? gates[(size_t)s * E + e] : 0.f;
    g = wave_sum(g);
    if (lane == 0) gcol[wave * E + e] = g;
  }
  __syncthreads();
  if (live) {
    for (int w = 0; w < wave; ++w) {
      r1 += wcnt[(w * 2 + 0) * E + i1];
      r2 += wcnt[(w * 2 + 1) * E + i2];
    }
    idx[2 * s] = i1;
    idx[2 * s + 1] = i2;
    rank[2 * s] = r1;
    rank[2 * s + 1] = r2;
  }
  for (int t = threadIdx.x; t < 2 * E; t += blockDim.x) {
    int c = 0;
    for (int w = 0; w < nw; ++w) c += wcnt[w * 2 * E + t];
    hist[(size_t)blockIdx.x * 2 * E + t] = c;
  }
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float g = 0.f;
    for (int w = 0; w < nw; ++w) g += gcol[w * E + e];
    gsum[(size_t)blockIdx.x * E + e] = g;
  }
}

// Pass 2 (scan): single block.  Exclusive prefix of the block histograms over blocks for each
// (choice, expert); totals; exp_counts; rows; l_aux.
__global__ __launch_bounds__(1024) void gate_scan_kernel(const int32_t* __restrict__ hist, int32_t* __restrict__ boff,
                                                        const float* __restrict__ gsum, int32_t* __restrict__ tot,
                                                        int64_t* __restrict__ exp_counts, int32_t* __restrict__ rows,
                                                        float* __restrict__ l_aux, int S, int E, int C, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* stot = reinterpret_cast<int32_t*>(smem);   // [2E]
  float* sme = reinterpret_cast<float*>(stot + 2 * E);  // [E]
  const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int pair = wave; pair < 2 * E; pair += nw) {
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
      const int b = b0 + lane;
      const int v = b < nblk ? hist[(size_t)b * 2 * E + pair] : 0;
      const int inc = wave_incl_scan(v);
      if (b < nblk) boff[(size_t)b * 2 * E + pair] = carry + inc - v;
      carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) stot[pair] = carry;
  }
  for (int e = wave; e < E; e += nw) {  // fixed lane-strided order, then butterfly: deterministic
    float g = 0.f;
    for (int b = lane; b < nblk; b += 64) g += gsum[(size_t)b * E + e];
    g = wave_sum(g);
    if (lane == 0) sme[e] = g;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * E; t += blockDim.x) tot[t] = stot[t];
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    exp_counts[e] = (int64_t)stot[e];
    const int filled = stot[e] + stot[E + e];
    rows[e] = filled < C ? filled : C;
  }
  if (threadIdx.x == 0) {
    // l_aux = mean_e(mean_s gates * mean_s mask1) * E * E     (sharded_moe.py:202-204)
    float acc = 0.f;
    for (int e = 0; e < E; ++e) {
      const float me = sme[e] / (float)S;
      const float ce = (float)stot[e] / (float)S;
      acc += me * ce;
    }
    l_aux[0] = acc / (float)E * (float)E * (float)E;
  }
}

// Pass 3 (assign): one thread per token.  Queue positions, capacity test, renormalised
// weights, token->slot and slot->token maps.
__global__ __launch_bounds__(kGateBlock) void gate_assign_kernel(
    const float* __restrict__ gates, const int32_t* __restrict__ idx, const int32_t* __restrict__ rank,
    const int32_t* __restrict__ boff, const int32_t* __restrict__ tot, int S, int E, int C, int32_t* __restrict__ loc,
    int32_t* __restrict__ slot, float* __restrict__ weight, int32_t* __restrict__ slot_token,
    float* __restrict__ slot_weight) {
  const int s = blockIdx.x * kGateBlock + threadIdx.x;
  if (s >= S) return;
  const int i1 = idx[2 * s], i2 = idx[2 * s + 1];
  const int32_t* bo = boff + (size_t)blockIdx.x * 2 * E;
  const int l1 = bo[i1] + rank[2 * s];
  const int l2 = bo[E + i2] + rank[2 * s + 1] + tot[i2];  // 2nd choices queue behind ALL 1st choices
  const bool k1 = l1 < C, k2 = l2 < C;
  const float g1 = k1 ? gates[(size_t)s * E + i1] : 0.f;
  const float g2 = k2 ? gates[(size_t)s * E + i2] : 0.f;
  const float den = fmaxf(g1 + g2, kEps);
  const float w1 = g1 / den, w2 = g2 / den;
  const int s1 = (k1 && w1 != 0.f) ? i1 * C + l1 : -1;
  const int s2 = (k2 && w2 != 0.f) ? i2 * C + l2 : -1;
  loc[2 * s] = l1;
  loc[2 * s + 1] = l2;
  slot[2 * s] = s1;
  slot[2 * s + 1] = s2;
  weight[2 * s] = w1;
  weight[2 * s + 1] = w2;
  if (s1 >= 0) { slot_token[s1] = s; slot_weight[s1] = w1; }
  if (s2 >= 0) { slot_token[s2] = s; slot_weight[s2] = w2; }
}

// Backward: one thread per token.
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                       const int32_t* __restrict__ loc,
                                                       const float* __restrict__ d_weight,
                                                       const float* __restrict__ d_l_aux,
                                                       const int64_t* __restrict__ exp_counts, int S, int E, int C,
                                                       float* __restrict__ dlogits) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const float* g = gates + (size_t)s * E;
  const int i1 = idx[2 * s], i2 = idx[2 * s + 1];
  const bool k1 = loc[2 * s] < C, k2 = loc[2 * s + 1] < C;
  const float g1 = k1 ? g[i1] : 0.f, g2 = k2 ? g[i2] : 0.f;
  const float dw1 = d_weight[2 * s], dw2 = d_weight[2 * s + 1];
  const float sum = g1 + g2;
  float dg1, dg2;
  if (sum >= kEps) {  // clamp inactive: w_k = g_k / (g1 + g2)
    const float inv2 = 1.f / (sum * sum);
    dg1 = (dw1 - dw2) * g2 * inv2;
    dg2 = (dw2 - dw1) * g1 * inv2;
  } else {  // clamp active: w_k = g_k / eps
    dg1 = dw1 / kEps;
    dg2 = dw2 / kEps;
  }
  if (!k1) dg1 = 0.f;
  if (!k2) dg2 = 0.f;
  const float dl = d_l_aux ? d_l_aux[0] : 0.f;
  const float aux_scale = dl * (float)E / ((float)S * (float)S);
  float dot = 0.f;
  for (int e = 0; e < E; ++e) {
    float dg = aux_scale * (float)exp_counts[e];
    if (e == i1) dg += dg1;
    if (e == i2) dg += dg2;
    dot = fmaf(g[e], dg, dot);
  }
  for (int e = 0; e < E; ++e) {
    float dg = aux_scale * (float)exp_counts[e];
    if (e == i1) dg += dg1;
    if (e == i2) dg += dg2;
    dlogits[(size_t)s * E + e] = g[e] * (dg - dot);
  }
}

// ================================================================ dwg = dlogits^T @ x
// Stage 1: block b owns tokens [b*TS, (b+1)*TS); thread t owns 4 columns (granule t of the
// M axis, looped when M/4 > blockDim) x all E experts.  Partials go to ws[b][E][M].
// Stage 2: sum partials over blocks in fixed order.
template <typename T, int EMAX>
__global__ __launch_bounds__(256) void gate_wgrad_partial_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                                 const T* __restrict__ temb,
                                                                 const float* __restrict__ dlogits,
                                                                 float* __restrict__ part, int S, int T_, int n_vis,
                                                                 int mode, int M, int E, int TS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dl = reinterpret_cast<float*>(smem);  // [TS][EMAX]
  const int s_begin = blockIdx.x * TS;
  const int s_end = min(S, s_begin + TS);
  for (int i = threadIdx.x; i < TS * EMAX; i += blockDim.x) {
    const int t = i / EMAX, e = i % EMAX;
    const int s = s_begin + t;
    dl[i] = (s < s_end && e < E) ? dlogits[(size_t)s * E + e] : 0.f;
  }
  __syncthreads();
  const int ngran = M >> 2;
  for (int gr = threadIdx.x; gr < ngran; gr += blockDim.x) {
    float acc[EMAX][4];
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[e][j] = 0.f;
    for (int s = s_begin; s < s_end; ++s) {
      const T* p = h + (size_t)s * M + gr * 4;
      float x[4];
      const T* emb = nullptr;
      if (mode != EVE_MODE_NONE) emb = token_modality(s, T_, n_vis, mode) ? temb : vemb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = to_f32<T>(p[j]);
        if (emb) v = to_f32<T>(from_f32<T>(v + to_f32<T>(emb[gr * 4 + j])));
        x[j] = v;
      }
      const float* d = dl + (s - s_begin) * EMAX;
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        const float dv = d[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[e][j] = fmaf(dv, x[j], acc[e][j]);
      }
    }
    float* out = part + (size_t)blockIdx.x * E * M;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
      if (e < E) *reinterpret_cast<float4*>(out + (size_t)e * M + gr * 4) = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int n, int nparts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
#pragma unroll 8
  for (int b = 0; b < nparts; ++b) acc += part[(size_t)b * n + i];
  out[i] = acc;
}

// =================================================================== host launchers
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
static int launch_logits(const void* h, const void* ve, const void* te, const float* wg, float* logits, int S, int T_,
                         int n_vis, int mode, int M, int E, hipStream_t st) {
  const int ngran = M / 4;
  const int ng = ceil_div(ngran, 64);
  constexpr int TOK = 4;
  const int waves_needed = ceil_div(S, TOK);
  const int blocks = max(1, min(ceil_div(waves_needed, 4), 2048));
#define EVE_LOGITS_CASE(NG)                                                                                          \
  case NG:                                                                                                          \
    hipLaunchKernelGGL((gate_logits_kernel<T, NG, TOK>), dim3(blocks), dim3(256), 0, st, (const T*)h, (const T*)ve, \
                       (const T*)te, wg, logits, S, T_, n_vis, mode, M, E);                                         \
    break;
  switch (ng) {
    EVE_LOGITS_CASE(1)
    EVE_LOGITS_CASE(2)
    EVE_LOGITS_CASE(3)
    EVE_LOGITS_CASE(4)
    EVE_LOGITS_CASE(5)
    EVE_LOGITS_CASE(6)
    EVE_LOGITS_CASE(7)
    EVE_LOGITS_CASE(8)
    default:
      set_error("eve_gate_logits_fwd: M=%d > 2048 not supported", M);
      return EVE_ERR_UNSUPPORTED;
  }
#undef EVE_LOGITS_CASE
  EVE_CHECK_LAUNCH("gate_logits_kernel");
  return EVE_OK;
}

struct GateWs {
  int nblk;
  size_t hist, boff, rank, gsum, tot, total;
};
static GateWs gate_ws_layout(int S, int E) {
  GateWs w;
  w.nblk = ceil_div(S, kGateBlock);
  size_t off = 0;
  w.hist = off; off = align_up(off + (size_t)w.nblk * 2 * E * 4, 256);
  w.boff = off; off = align_up(off + (size_t)w.nblk * 2 * E * 4, 256);
  w.rank = off; off = align_up(off + (size_t)S * 2 * 4, 256);
  w.gsum = off; off = align_up(off + (size_t)w.nblk * E * 4, 256);
  w.tot = off;  off = align_up(off + (size_t)2 * E * 4, 256);
  w.total = off;
  return w;
}

}  // namespace eve

using namespace eve;

extern "C" int eve_gate_logits_fwd(const void* h, const void* vision_embed, const void* text_embed, const float* wg,
                                   float* logits, int S, int T, int n_vis, int mode, int M, int E, int dtype,
                                   void* stream) {
  EVE_REQUIRE(h && wg && logits, EVE_ERR_ARG, "eve_gate_logits_fwd: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && M > 0 && M % 8 == 0, EVE_ERR_ARG, "eve_gate_logits_fwd: bad shape S=%d E=%d M=%d (M%%8)", S, E, M);
  EVE_REQUIRE(mode >= EVE_MODE_NONE && mode <= EVE_MODE_TEXT, EVE_ERR_ARG, "eve_gate_logits_fwd: bad mode %d", mode);
  if (mode != EVE_MODE_NONE) {
    EVE_REQUIRE(vision_embed && text_embed, EVE_ERR_ARG, "eve_gate_logits_fwd: modality embeddings required for mode %d", mode);
    EVE_REQUIRE(T > 0 && S % T == 0 && n_vis >= 0, EVE_ERR_ARG, "eve_gate_logits_fwd: S=%d not a multiple of T=%d", S, T);
  }
  if (dtype == EVE_F32) return launch_logits<float>(h, vision_embed, text_embed, wg, logits, S, T, n_vis, mode, M, E, S_(stream));
  if (dtype == EVE_BF16) return launch_logits<bf16_t>(h, vision_embed, text_embed, wg, logits, S, T, n_vis, mode, M, E, S_(stream));
  set_error("eve_gate_logits_fwd: bad dtype %d", dtype);
  return EVE_ERR_ARG;
}

extern "C" size_t eve_gate_workspace_bytes(int S, int E) {
  if (S <= 0 || E <= 0) return 0;
  return gate_ws_layout(S, E).total;
}

extern "C" int eve_gate_top2_fwd(const float* logits, const float* noise, int S, int E, int C, float* gates,
                                 int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token,
                                 float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(logits && noise && gates && idx && loc && slot && weight && slot_token && slot_weight && exp_counts &&
                  rows && l_aux && workspace,
              EVE_ERR_ARG, "eve_gate_top2_fwd: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && C > 0, EVE_ERR_ARG, "eve_gate_top2_fwd: bad shape S=%d E=%d C=%d", S, E, C);
  EVE_REQUIRE(E <= 1024, EVE_ERR_UNSUPPORTED, "eve_gate_top2_fwd: E=%d > 1024", E);
  EVE_REQUIRE((long long)E * C < (1ll << 31), EVE_ERR_ARG, "eve_gate_top2_fwd: E*C overflows int32");
  const GateWs w = gate_ws_layout(S, E);
  EVE_REQUIRE(workspace_bytes >= w.total, EVE_ERR_WORKSPACE, "eve_gate_top2_fwd: workspace %zu < %zu", workspace_bytes, w.total);
  EVE_REQUIRE(((uintptr_t)workspace & 255) == 0, EVE_ERR_ARG, "eve_gate_top2_fwd: workspace must be 256-byte aligned");
  char* ws = static_cast<char*>(workspace);
  int32_t* hist = reinterpret_cast<int32_t*>(ws + w.hist);
  int32_t* boff = reinterpret_cast<int32_t*>(ws + w.boff);
  int32_t* rank = reinterpret_cast<int32_t*>(ws + w.rank);
  float* gsum = reinterpret_cast<float*>(ws + w.gsum);
  int32_t* tot = reinterpret_cast<int32_t*>(ws + w.tot);
  hipStream_t st = S_(stream);
  const int nw = kGateBlock / 64;
  const size_t lds1 = (size_t)nw * 2 * E * 4 + (size_t)nw * E * 4;
  hipLaunchKernelGGL(gate_select_kernel, dim3(w.nblk), dim3(kGateBlock), lds1, st, logits, noise, S, E, gates, idx,
                     hist, rank, gsum, slot_token, E * C);
  EVE_CHECK_LAUNCH("gate_select_kernel");
  const size_t lds2 = (size_t)3 * E * 4;
  hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(1024), lds2, st, hist, boff, gsum, tot, exp_counts, rows, l_aux, S,
                     E, C, w.nblk);
  EVE_CHECK_LAUNCH("gate_scan_kernel");
  hipLaunchKernelGGL(gate_assign_kernel, dim3(w.nblk), dim3(kGateBlock), 0, st, gates, idx, rank, boff, tot, S, E, C,
                     loc, slot, weight, slot_token, slot_weight);
  EVE_CHECK_LAUNCH("gate_assign_kernel");
  return EVE_OK;
}

extern "C" int eve_gate_top2_bwd(const float* gates, const int32_t* idx, const int32_t* loc, const float* d_weight,
                                 const float* d_l_aux, const int64_t* exp_counts, int S, int E, int C, float* dlogits,
                                 void* stream) {
  EVE_REQUIRE(gates && idx && loc && d_weight && exp_counts && dlogits, EVE_ERR_ARG, "eve_gate_top2_bwd: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && C > 0, EVE_ERR_ARG, "eve_gate_top2_bwd: bad shape");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(S, 256)), dim3(256), 0, S_(stream), gates, idx, loc, d_weight,
                     d_l_aux, exp_counts, S, E, C, dlogits);
  EVE_CHECK_LAUNCH("gate_bwd_kernel");
  return EVE_OK;
}

static const int kWgradTS = 64;

extern "C" size_t eve_gate_wgrad_workspace_bytes(int S, int M, int E) {
  if (S <= 0) return 0;
  return (size_t)ceil_div(S, kWgradTS) * E * M * sizeof(float);
}

extern "C" int eve_gate_wgrad(const void* h, const void* vision_embed, const void* text_embed, const float* dlogits,
                              float* dwg, int S, int T, int n_vis, int mode, int M, int E, int dtype, void* workspace,
                              size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(h && dlogits && dwg && workspace, EVE_ERR_ARG, "eve_gate_wgrad: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && M % 8 == 0, EVE_ERR_ARG, "eve_gate_wgrad: bad shape");
  EVE_REQUIRE(E <= 32, EVE_ERR_UNSUPPORTED, "eve_gate_wgrad: E=%d > 32 not supported", E);
  if (mode != EVE_MODE_NONE) EVE_REQUIRE(vision_embed && text_embed && T > 0, EVE_ERR_ARG, "eve_gate_wgrad: embeddings required");
  const size_t need = eve_gate_wgrad_workspace_bytes(S, M, E);
  EVE_REQUIRE(workspace_bytes >= need, EVE_ERR_WORKSPACE, "eve_gate_wgrad: workspace %zu < %zu", workspace_bytes, need);
  const int nb = ceil_div(S, kWgradTS);
  float* part = static_cast<float*>(workspace);
  hipStream_t st = S_(stream);
  const size_t lds = (size_t)kWgradTS * 32 * 4;
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((gate_wgrad_partial_kernel<float, 32>), dim3(nb), dim3(256), lds, st, (const float*)h,
                       (const float*)vision_embed, (const float*)text_embed, dlogits, part, S, T, n_vis, mode, M, E,
                       kWgradTS);
  else if (dtype == EVE_BF16)
    hipLaunchKernelGGL((gate_wgrad_partial_kernel<bf16_t, 32>), dim3(nb), dim3(256), lds, st, (const bf16_t*)h,
                       (const bf16_t*)vision_embed, (const bf16_t*)text_embed, dlogits, part, S, T, n_vis, mode, M, E,
                       kWgradTS);
  else {
    set_error("eve_gate_wgrad: bad dtype %d", dtype);
    return EVE_ERR_ARG;
  }
  EVE_CHECK_LAUNCH("gate_wgrad_partial_kernel");
  const int n = E * M;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, part, dwg, n, nb);
  EVE_CHECK_LAUNCH("sum_partials_kernel");
  return EVE_OK;
}
