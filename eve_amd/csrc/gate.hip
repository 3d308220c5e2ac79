// Gate kernels: modality-add + gate logits, fused top-2 gating (softmax, two argmaxes,
// ordered per-expert queue positions, capacity drop, renormalised weights, l_aux), and
// their backward.  gfx950, wave64.  Reference semantics: models/sharded_moe.py:172-235,
// :283-297, :350-364 (see include/eve_moe.h for the per-entry-point citations).
#include <mutex>

#include "common.h"
#include "gate_bwd32.h"
#include "gate_wgrad_bf16.h"

namespace eve {

constexpr float kEps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps, sharded_moe.py:221
constexpr int kGateBlock = 64;                   // tokens per block (one wave) in the select / assign passes

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
// Gumbel(0,1) variate of one standard-uniform draw with torch.distributions.Gumbel's arithmetic
// (sharded_moe.py:70-77,:186): u = tiny + r * ((1 - eps) - tiny), then 0 + (-1) * log(0 + (-1) * log(u)),
// every step rounded to fp32, no contraction.
__device__ __forceinline__ float gumbel_of_uniform(float r) {
  const float low = 1.17549435e-38f;                 // finfo(float32).tiny
  const float high = 1.0f - 1.1920928955078125e-07f;  // 1 - finfo(float32).eps
  const float span = __fsub_rn(high, low);
  const float u = __fadd_rn(low, __fmul_rn(r, span));
  const float a = __fadd_rn(0.f, __fmul_rn(-1.f, logf(u)));
  return __fadd_rn(0.f, __fmul_rn(-1.f, logf(a)));
}

// ---- noise source 2: the gate draws its own standard-uniform values --------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the generator family behind curand / torch's device generator), keyed by the
// 64-bit seed, counter = offset + (element >> 2) in words 0/1 (words 2/3 zero), element = s * E + e of the [S,E]
// noise tensor, value = word (element & 3) of the block; uniform = (word >> 8) * 2^-24 in [0,1).  `noise` then
// points at the 16-byte device state {seed, offset} (include/eve_moe.h), read here and advanced by
// ceil(S * E / 4) in the kernel that follows the selection (gate_assign32_kernel / gate_scan_kernel).
constexpr int kNoisePhilox = 2;
struct Philox4 { unsigned w[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long ctr, unsigned long long seed) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float philox_unit(unsigned w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
__device__ __forceinline__ float philox_uniform_at(const unsigned long long* st, unsigned long long element) {
  const Philox4 b = philox4x32_10(st[1] + (element >> 2), st[0]);
  const unsigned k = (unsigned)element & 3u;
  return philox_unit(k == 0 ? b.w[0] : k == 1 ? b.w[1] : k == 2 ? b.w[2] : b.w[3]);
}
// (thread 0 of block 0 of the kernel after the selection; a vector store)
__device__ __forceinline__ void philox_advance(unsigned long long* st, unsigned long long inc) {
  if (st && blockIdx.x == 0 && threadIdx.x == 0) st[1] += inc;
}

// Pass 1 (select): one thread per token.  softmax, first / second choice, rank of the token
// among earlier tokens OF ITS BLOCK with the same choice (wave ballots + LDS), per-block
// histograms and per-block column sums of the gates.  Also clears slot_token.
// Workspace layout (ints unless noted), nblk = ceil(S / kGateBlock):
//   hist   [2][E][nblk]   block histograms of first / second choices (block index fastest)
//   boff   [2][E][nblk]   exclusive prefix of hist over blocks (pass 2)
//   rank   [S][2]         in-block rank
//   gsum   [E][nblk]      float, per-block column sums of gates
//   tot    [2][E]         totals
__global__ __launch_bounds__(kGateBlock) void gate_select_kernel(
    const float* __restrict__ logits, const float* __restrict__ noise, int noise_is_uniform, int S, int E,
    float* __restrict__ gates, int32_t* __restrict__ idx, int32_t* __restrict__ hist, int32_t* __restrict__ rank,
    float* __restrict__ gsum, int32_t* __restrict__ slot_token, float* __restrict__ slot_weight, int num_slots,
    int use_tile) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(kGateBlock == 64, "one wave per block");
  float* tile = reinterpret_cast<float*>(smem);  // [64][E+1] gates of the block's tokens (use_tile)

  // clear the inverse map (consumed by pass 3, which runs after this kernel completes)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_slots; i += gridDim.x * blockDim.x) {
    slot_token[i] = -1;
    slot_weight[i] = 0.f;
  }

  const int s = blockIdx.x * kGateBlock + threadIdx.x;
  const bool live = s < S;
  const int lane = lane_id();
  const float* row = logits + (size_t)(live ? s : 0) * E;
  const float* nrow = noise + (noise_is_uniform == kNoisePhilox ? 0 : (size_t)(live ? s : 0) * E);
  float* grow = gates + (size_t)(live ? s : 0) * E;

  // softmax exactly as exp(x - max) / sum; first choice = argmax of the GATES (ties -> lowest e);
  // second choice = argmax of logits + noise with the first choice masked out
  float mx = -INFINITY;
  for (int e = 0; e < E; ++e) mx = fmaxf(mx, row[e]);
  float sum = 0.f;
  for (int e = 0; e < E; ++e) sum += expf(row[e] - mx);
  int i1 = 0;
  float best = -INFINITY;
  for (int e = 0; e < E; ++e) {
    const float g = expf(row[e] - mx) / sum;
    if (live) grow[e] = g;
    if (use_tile) tile[threadIdx.x * (E + 1) + e] = live ? g : 0.f;
    if (g > best) { best = g; i1 = e; }
  }
  int i2 = -1;
  float best2 = -INFINITY;
  for (int e = 0; e < E; ++e) {
    if (e == i1) continue;
    const float nv = noise_is_uniform == kNoisePhilox
                         ? philox_uniform_at(reinterpret_cast<const unsigned long long*>(noise), (unsigned long long)(live ? s : 0) * E + e)
                         : nrow[e];
    const float v = row[e] + (noise_is_uniform ? gumbel_of_uniform(nv) : nv);
    if (i2 < 0 || v > best2) { best2 = v; i2 = e; }
  }
  if (i2 < 0) i2 = 0;  // E == 1: masked_fill leaves all -inf, argmax -> 0
  if (!live) { i1 = -1; i2 = -1; }

  // rank of each token among the block's earlier tokens with the same choice; block histograms
  int r1 = 0, r2 = 0;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int e = 0; e < E; ++e) {
    const unsigned long long m1 = __ballot(i1 == e);
    const unsigned long long m2 = __ballot(i2 == e);
    if (i1 == e) r1 = __popcll(m1 & lt);
    if (i2 == e) r2 = __popcll(m2 & lt);
    if (lane == 0) {
      hist[(size_t)e * gridDim.x + blockIdx.x] = __popcll(m1);
      hist[(size_t)(E + e) * gridDim.x + blockIdx.x] = __popcll(m2);
    }
    if (!use_tile) {  // column sum of gates over the block's tokens, butterfly (fixed order)
      float g = live ? grow[e] : 0.f;
      g = wave_sum(g);
      if (lane == 0) gsum[(size_t)e * gridDim.x + blockIdx.x] = g;
    }
  }
  if (live) {
    idx[2 * s] = i1;
    idx[2 * s + 1] = i2;
    rank[2 * s] = r1;
    rank[2 * s + 1] = r2;
  }
  if (use_tile) {  // column sums from the LDS tile: lane e walks the 64 rows in order
    __syncthreads();
    for (int e = lane; e < E; e += 64) {
      float g = 0.f;
      for (int r = 0; r < kGateBlock; ++r) g += tile[r * (E + 1) + e];
      gsum[(size_t)e * gridDim.x + blockIdx.x] = g;
    }
  }
}

// value of lane (l ^ 1) / (l ^ 2): quad-permute DPP moves, no LDS crossbar trip
template <int CTRL>
__device__ __forceinline__ float quad_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;
template <int CTRL>
__device__ __forceinline__ void argmax_quad(float& v, int& i) {
  const float ov = quad_f<CTRL>(v);
  const int oi = quad_i<CTRL>(i);
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

// Pass 1 for E <= 32: 256 threads per 64-token block, FOUR lanes per token, each owning eight consecutive
// experts in registers (two 16-byte loads per operand when E == 32, so a wave reads 2 KiB contiguous
// rows): max / sum / arg-max are seven in-register steps plus two quad-permute DPP steps.  `noise` holds
// Gumbel variates, or (noise_is_uniform) the standard-uniform draws they are made of, transformed here.
// Part B is the ballot ranking of the kernel above with the experts split over the four waves.  Same
// outputs and workspace layout; also clears slot_token / slot_weight for pass 3.
// The body of gate_select32_kernel for the block's first 256 threads (waves 0-3; a larger block's other waves only
// take part in the clearing loop and the barriers).  LDS_LOGITS: the block's logits come from `lds_logits` [64][33]
// (gate_logits_select_kernel) instead of global memory -- same values, same arithmetic.
template <bool E32, bool LDS_LOGITS>
__device__ __forceinline__ void gate_select32_body(
    const float* __restrict__ logits, const float* lds_logits, const float* __restrict__ noise, int noise_is_uniform, int S, int E,
    float* __restrict__ gates, int32_t* __restrict__ idx, int32_t* __restrict__ hist, int32_t* __restrict__ rank,
    float* __restrict__ gsum, int32_t* __restrict__ slot_token, float* __restrict__ slot_weight, int num_slots,
    int block_major, int* ci1, int* ci2, float (*tile)[33], float (*gpart)[32]) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_slots; i += gridDim.x * blockDim.x) {
    slot_token[i] = -1;
    slot_weight[i] = 0.f;
  }

  const int lane = lane_id(), wave = threadIdx.x >> 6;
  if (wave < 4) {
    const int q = lane & 3, t = wave * 16 + (lane >> 2);  // token inside the block
    const int s = blockIdx.x * kGateBlock + t;
    const bool live = s < S;
    const int e0 = q * 8;
    float x[8], nz[8];
    const bool own_draws = noise_is_uniform == kNoisePhilox;
    const unsigned long long* rng = reinterpret_cast<const unsigned long long*>(noise);
    if (E32) {
      if constexpr (LDS_LOGITS) {  // (the block's own logits, left in LDS by the kernel that formed them)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = lds_logits[t * 33 + e0 + j];
      } else {
        const float4* lp = reinterpret_cast<const float4*>(logits + (size_t)(live ? s : 0) * 32 + e0);
        const float4 a = lp[0], b = lp[1];
        x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
      }
      if (own_draws) {  // the lane's eight elements are two whole Philox blocks
        const unsigned long long c0 = rng[1] + ((unsigned long long)(live ? s : 0) * 32 + e0) / 4;
        const Philox4 p = philox4x32_10(c0, rng[0]), r = philox4x32_10(c0 + 1, rng[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) nz[j] = philox_unit(p.w[j]), nz[4 + j] = philox_unit(r.w[j]);
      } else {
        const float4* np = reinterpret_cast<const float4*>(noise + (size_t)(live ? s : 0) * 32 + e0);
        const float4 c = np[0], d = np[1];
        nz[0] = c.x, nz[1] = c.y, nz[2] = c.z, nz[3] = c.w, nz[4] = d.x, nz[5] = d.y, nz[6] = d.z, nz[7] = d.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t o = (size_t)(live ? s : 0) * E + min(e0 + j, E - 1);
        if constexpr (LDS_LOGITS) x[j] = lds_logits[t * 33 + min(e0 + j, E - 1)];
        else x[j] = logits[o];
        nz[j] = own_draws ? philox_uniform_at(rng, o) : noise[o];
      }
    }
    bool valid[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) valid[j] = live && (E32 || e0 + j < E);
    // softmax exactly as exp(x - max) / sum
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, valid[j] ? x[j] : -INFINITY);
    mx = fmaxf(mx, quad_f<kQuadXor1>(mx));
    mx = fmaxf(mx, quad_f<kQuadXor2>(mx));
    float ex[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ex[j] = valid[j] ? expf(x[j] - mx) : 0.f;
    float sum = ((ex[0] + ex[4]) + (ex[2] + ex[6])) + ((ex[1] + ex[5]) + (ex[3] + ex[7]));
    sum += quad_f<kQuadXor1>(sum);
    sum += quad_f<kQuadXor2>(sum);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = ex[j] / sum;
    if (E32) {
      if (live) {
        float4* gp = reinterpret_cast<float4*>(gates + (size_t)s * 32 + e0);
        gp[0] = make_float4(g[0], g[1], g[2], g[3]);
        gp[1] = make_float4(g[4], g[5], g[6], g[7]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (valid[j]) gates[(size_t)s * E + e0 + j] = g[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[t][e0 + j] = valid[j] ? g[j] : 0.f;
    // first choice = argmax of the GATES (ties -> lowest e)
    float v1 = -INFINITY;
    int i1 = e0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = valid[j] ? g[j] : -INFINITY;
      if (v > v1) { v1 = v; i1 = e0 + j; }
    }
    argmax_quad<kQuadXor1>(v1, i1);
    argmax_quad<kQuadXor2>(v1, i1);
    // second choice = argmax of logits + noise with the first choice masked out
    float v2 = -INFINITY;
    int i2 = e0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float n = noise_is_uniform ? gumbel_of_uniform(nz[j]) : nz[j];
      const float v = (valid[j] && e0 + j != i1) ? x[j] + n : -INFINITY;
      if (v > v2) { v2 = v; i2 = e0 + j; }
    }
    argmax_quad<kQuadXor1>(v2, i2);
    argmax_quad<kQuadXor2>(v2, i2);
    if (q == 0) {
      ci1[t] = live ? i1 : -1;
      ci2[t] = live ? i2 : -1;  // E == 1: every candidate is -inf and the tie goes to expert 0, as torch.argmax
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {  // per-block column sums of the gates (for l_aux), fixed order: 8 rows per thread, then 8 partials
    const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float p = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) p += tile[rg * 8 + r][e];
    gpart[rg][e] = p;
  }
  __syncthreads();
  if (threadIdx.x < E) {
    float p = gpart[0][threadIdx.x];
#pragma unroll
    for (int r = 1; r < 8; ++r) p += gpart[r][threadIdx.x];
    gsum[block_major ? (size_t)blockIdx.x * E + threadIdx.x : (size_t)threadIdx.x * gridDim.x + blockIdx.x] = p;
  }

  // part B: lane = token
  if (wave >= 4) return;
  const int s = blockIdx.x * kGateBlock + lane;
  const int i1 = ci1[lane], i2 = ci2[lane];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int x = wave; x < E; x += 4) {
    const unsigned long long m1 = __ballot(i1 == x);
    const unsigned long long m2 = __ballot(i2 == x);
    if (i1 == x) rank[2 * s] = __popcll(m1 & lt);
    if (i2 == x) rank[2 * s + 1] = __popcll(m2 & lt);
    if (lane == 0) {  // hist[2E][nblk] for the scan kernel, hist[nblk][2E] for gate_assign32_kernel
      hist[block_major ? (size_t)blockIdx.x * 2 * E + x : (size_t)x * gridDim.x + blockIdx.x] = __popcll(m1);
      hist[block_major ? (size_t)blockIdx.x * 2 * E + E + x : (size_t)(E + x) * gridDim.x + blockIdx.x] = __popcll(m2);
    }
  }
  if (wave == 0 && s < S) {
    idx[2 * s] = i1;
    idx[2 * s + 1] = i2;
  }
}

template <bool E32>
__global__ __launch_bounds__(256) void gate_select32_kernel(
    const float* __restrict__ logits, const float* __restrict__ noise, int noise_is_uniform, int S, int E,
    float* __restrict__ gates, int32_t* __restrict__ idx, int32_t* __restrict__ hist, int32_t* __restrict__ rank,
    float* __restrict__ gsum, int32_t* __restrict__ slot_token, float* __restrict__ slot_weight, int num_slots,
    int block_major) {
  __shared__ int ci1[kGateBlock], ci2[kGateBlock];
  __shared__ float tile[kGateBlock][33];
  __shared__ float gpart[8][32];
  gate_select32_body<E32, false>(logits, nullptr, noise, noise_is_uniform, S, E, gates, idx, hist, rank, gsum, slot_token,
                                 slot_weight, num_slots, block_major, ci1, ci2, tile, gpart);
}

// Pass 2 (scan): single block.  Exclusive prefix of the block histograms over blocks for each
// (choice, expert); totals; exp_counts; rows; l_aux.
__global__ __launch_bounds__(1024) void gate_scan_kernel(const int32_t* __restrict__ hist, int32_t* __restrict__ boff,
                                                        const float* __restrict__ gsum, int32_t* __restrict__ tot,
                                                        int64_t* __restrict__ exp_counts, int32_t* __restrict__ rows,
                                                        float* __restrict__ l_aux, int S, int E, int C, int nblk,
                                                        unsigned long long* __restrict__ rng_state,
                                                        unsigned long long rng_inc, int64_t* __restrict__ exp_counts_mirror) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  philox_advance(rng_state, rng_inc);
  int32_t* stot = reinterpret_cast<int32_t*>(smem);   // [2E]
  float* sme = reinterpret_cast<float*>(stot + 2 * E);  // [E]
  const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // Each wave scans one (choice, expert) row of hist at a time, 256 blocks per step: lane l owns
  // blocks 4l .. 4l+3 (one 16-byte load), the 64 lane totals are scanned across the wave, and the
  // lane writes its four exclusive prefixes with one 16-byte store.  nblk is padded to a multiple of 4
  // in the workspace layout, so the vector accesses stay inside the row.
  constexpr int R = 4;  // rows whose loads are in flight together
  // column sums of the gates: wave w owns experts w, w + nw, ...; their loads are issued first so that they
  // are in flight together with the histogram loads below (one memory round trip instead of two)
  constexpr int GE = 2, GB = 4;  // experts per wave and 64-block strides held in registers
  float greg[GE][GB];
  const bool g_in_regs = E <= GE * nw && nblk <= GB * 64;
  if (g_in_regs) {
#pragma unroll
    for (int i = 0; i < GE; ++i)
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int e = wave + i * nw, b = lane + 64 * k;
        greg[i][k] = (e < E && b < nblk) ? gsum[(size_t)e * nblk + b] : 0.f;
      }
  }
  for (int pair0 = wave; pair0 < 2 * E; pair0 += R * nw) {
    int carry[R];
#pragma unroll
    for (int r = 0; r < R; ++r) carry[r] = 0;
    for (int c0 = 0; c0 < nblk; c0 += 256) {
      const int b = c0 + 4 * lane;
      int v[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int pair = min(pair0 + r * nw, 2 * E - 1);
        const int32_t* hrow = hist + (size_t)pair * nblk;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[r][j] = hrow[min(b + j, nblk - 1)];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int pair = pair0 + r * nw;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b + j >= nblk) v[r][j] = 0;
        const int run = v[r][0] + v[r][1] + v[r][2] + v[r][3];
        const int inc = wave_incl_scan(run);
        int pre = carry[r] + inc - run;
        if (pair < 2 * E) {
          int32_t* brow = boff + (size_t)pair * nblk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (b + j < nblk) brow[b + j] = pre;
            pre += v[r][j];
          }
        }
        carry[r] += __shfl(inc, 63, 64);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (lane == 0 && pair0 + r * nw < 2 * E) stot[pair0 + r * nw] = carry[r];
  }
  if (g_in_regs) {  // fixed lane-strided order, then butterfly: deterministic
#pragma unroll
    for (int i = 0; i < GE; ++i) {
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < GB; ++k) g += greg[i][k];
      g = wave_sum(g);
      if (lane == 0 && wave + i * nw < E) sme[wave + i * nw] = g;
    }
  } else {
    for (int e = wave; e < E; e += nw) {
      float g = 0.f;
      for (int b = lane; b < nblk; b += 64) g += gsum[(size_t)e * nblk + b];
      g = wave_sum(g);
      if (lane == 0) sme[e] = g;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * E; t += blockDim.x) tot[t] = stot[t];
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    exp_counts[e] = (int64_t)stot[e];
    if (exp_counts_mirror) exp_counts_mirror[e] = (int64_t)stot[e];
    const int filled = stot[e] + stot[E + e];
    rows[e] = filled < C ? filled : C;
  }
  // l_aux = mean_e(mean_s gates * mean_s mask1) * E * E     (sharded_moe.py:202-204): the products in
  // parallel, their sum by one thread in expert order
  float* term = sme + E;  // [E]
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const float me = sme[e] / (float)S;
    const float ce = (float)stot[e] / (float)S;
    term[e] = me * ce;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int e = 0; e < E; ++e) acc += term[e];
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
  const int nblk = gridDim.x;
  const int l1 = boff[(size_t)i1 * nblk + blockIdx.x] + rank[2 * s];
  const int l2 = boff[(size_t)(E + i2) * nblk + blockIdx.x] + rank[2 * s + 1] + tot[i2];  // behind ALL 1st choices
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

// Passes 2 + 3 in one kernel for E <= 32 and at most 256 blocks (S <= 16384): every block derives its own
// queue offsets -- for each (choice, expert) column of the BLOCK-MAJOR histogram hist[blk][2E], the sum over
// the blocks before it and the total -- with 512 threads (thread = column x one eighth of the blocks, up to
// 32 independent coalesced loads each, partials combined through LDS in a fixed order), then assigns its
// 64 tokens as gate_assign_kernel does.  Block 0 also writes exp_counts, rows and l_aux (gsum is block-major
// [blk][E] here as well).  hist is re-read once per block (L2-resident, nblk^2 * 2E ints in all); that
// saves the single-block scan kernel, its launch and one dependent pass.  The scan + assign pair remains
// for longer inputs and for E > 32.
constexpr int kAssign32MaxBlocks = 256;
__global__ __launch_bounds__(512) void gate_assign32_kernel(
    const float* __restrict__ gates, const int32_t* __restrict__ idx, const int32_t* __restrict__ rank,
    const int32_t* __restrict__ hist, const float* __restrict__ gsum, int S, int E, int C, int32_t* __restrict__ loc,
    int32_t* __restrict__ slot, float* __restrict__ weight, int32_t* __restrict__ slot_token,
    float* __restrict__ slot_weight, int64_t* __restrict__ exp_counts, int32_t* __restrict__ rows,
    float* __restrict__ l_aux, unsigned long long* __restrict__ rng_state, unsigned long long rng_inc,
    int64_t* __restrict__ exp_counts_mirror) {
  __shared__ int32_t pbefore[8][64], pall[8][64];
  philox_advance(rng_state, rng_inc);
  __shared__ int32_t pre[64], tot[64];
  __shared__ float gpart[16][32];
  __shared__ float term[32];
  const int nblk = gridDim.x, me = blockIdx.x;
  {
    const int r = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int per = (nblk + 7) >> 3, b0 = part * per, b1 = min(nblk, b0 + per);
    const int rc = min(r, 2 * E - 1);
    int v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = hist[(size_t)min(b0 + k, nblk - 1) * (2 * E) + rc];  // clamped, masked below
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int b = b0 + k;
      all += b < b1 ? v[k] : 0;
      before += (b < b1 && b < me) ? v[k] : 0;
    }
    pbefore[part][r] = before;
    pall[part][r] = all;
  }
  float gacc = 0.f;
  if (me == 0) {  // column sums of the gates over all blocks: 16 partial sums per expert, fixed order
    const int e = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int per = (nblk + 15) >> 4, b0 = part * per, b1 = min(nblk, b0 + per);
    const int ec = min(e, E - 1);
    float g[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = gsum[(size_t)min(b0 + k, nblk - 1) * E + ec];
#pragma unroll
    for (int k = 0; k < 16; ++k) gacc += (b0 + k < b1) ? g[k] : 0.f;
    gpart[part][e] = gacc;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    int before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      before += pbefore[q][threadIdx.x];
      all += pall[q][threadIdx.x];
    }
    pre[threadIdx.x] = before;
    tot[threadIdx.x] = all;
  }
  __syncthreads();
  if (me == 0) {
    // l_aux = mean_e(mean_s gates * mean_s mask1) * E * E     (sharded_moe.py:202-204)
    if (threadIdx.x < E) {
      const int e = threadIdx.x;
      float gs = gpart[0][e];
#pragma unroll
      for (int q = 1; q < 16; ++q) gs += gpart[q][e];
      term[e] = (gs / (float)S) * ((float)tot[e] / (float)S);
      exp_counts[e] = (int64_t)tot[e];
      if (exp_counts_mirror) exp_counts_mirror[e] = (int64_t)tot[e];
      const int filled = tot[e] + tot[E + e];
      rows[e] = filled < C ? filled : C;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float acc = 0.f;
      for (int e = 0; e < E; ++e) acc += term[e];
      l_aux[0] = acc / (float)E * (float)E * (float)E;
    }
  }
  const int s = me * kGateBlock + threadIdx.x;
  if (threadIdx.x >= kGateBlock || s >= S) return;
  const int i1 = idx[2 * s], i2 = idx[2 * s + 1];
  const int l1 = pre[i1] + rank[2 * s];
  const int l2 = pre[E + i2] + rank[2 * s + 1] + tot[i2];  // behind ALL 1st choices
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
  const float dw1 = d_weight ? d_weight[2 * s] : 0.f, dw2 = d_weight ? d_weight[2 * s + 1] : 0.f;
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

// Backward for E <= 32: one lane per (token, expert), two tokens per wave; rows move as coalesced
// 128-byte loads / stores and the softmax-Jacobian dot product is a butterfly over the 32-lane half.
__global__ __launch_bounds__(256) void gate_bwd32_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ loc,
                                                         const float* __restrict__ d_weight,
                                                         const float* __restrict__ d_l_aux,
                                                         const int64_t* __restrict__ exp_counts, int S, int E, int C,
                                                         float* __restrict__ dlogits) {
  const int lane = lane_id();
  const int e = lane & 31, tk = lane >> 5;
  const int ec = min(e, E - 1);
  const float dl = d_l_aux ? d_l_aux[0] : 0.f;
  const float aux_scale = dl * (float)E / ((float)S * (float)S);
  const float aux = aux_scale * (float)exp_counts[ec];
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = 2 * wid; s0 < S; s0 += 2 * nw) {
    const int s = s0 + tk;
    const bool live = s < S, valid = live && e < E;
    const int sc = live ? s : S - 1;
    const float g = valid ? gates[(size_t)sc * E + ec] : 0.f;
    const int i1 = idx[2 * sc], i2 = idx[2 * sc + 1];
    const bool k1 = loc[2 * sc] < C, k2 = loc[2 * sc + 1] < C;
    const float dw1 = d_weight ? d_weight[2 * sc] : 0.f, dw2 = d_weight ? d_weight[2 * sc + 1] : 0.f;
    const float dl_e = gate_bwd32_token(g, i1, i2, k1, k2, dw1, dw2, aux, e, tk);
    if (valid) dlogits[(size_t)s * E + e] = dl_e;
  }
}

// ================================================================ dwg = dlogits^T @ x
// Stage 1: block (b, c) owns tokens [b*TS, (b+1)*TS) and experts [c*EMAX, (c+1)*EMAX); thread t owns 4 columns
// (granule t of the M axis, looped when M/4 > blockDim) x those experts.  Partials go to ws[b][E][M].
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
  const int e0 = blockIdx.y * EMAX;
  for (int i = threadIdx.x; i < TS * EMAX; i += blockDim.x) {
    const int t = i / EMAX, e = e0 + i % EMAX;
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
      if (e0 + e < E)
        *reinterpret_cast<float4*>(out + (size_t)(e0 + e) * M + gr * 4) = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
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

// ======================================================== fp32-MFMA versions (E <= 32)
// v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain at the VALU rate, but one instruction does a
// 32x32x2 block, which removes the per-element operand traffic of the VALU kernels above.
typedef __attribute__((ext_vector_type(16))) float f32x16;

// modality-added, dtype-rounded activation element
template <typename T>
__device__ __forceinline__ float x_elem(const T* __restrict__ hp, const T* __restrict__ emb, int col) {
  float v = to_f32<T>(hp[col]);
  if (emb) v = to_f32<T>(from_f32<T>(v + to_f32<T>(emb[col])));
  return v;
}

// logits tile: D[32 tokens x 32 experts] += A[32 tokens x 2 k] * B[2 k x 32 experts].  One block per 32
// tokens; its 4 waves split the K axis and their partial tiles are summed through LDS in wave order
// (deterministic).  Lane (i = lane & 31, hh = lane >> 5) supplies, for each 16-wide k block, the 8
// contiguous elements k0 + 8*hh .. +8 of token i (A) and of expert i (B): MFMA step j multiplies
// k0+j and k0+8+j.
template <typename T, bool EMB>
__global__ __launch_bounds__(256) void gate_logits_mfma_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                               const T* __restrict__ temb,
                                                               const float* __restrict__ wg,
                                                               float* __restrict__ logits, int S, int T_, int n_vis,
                                                               int mode, int M, int E) {
  __shared__ float red[4][16][64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * 32;
  const int i = lane & 31, hh = lane >> 5;
  const int s = min(s0 + i, S - 1);
  const int kq = ((M / 16 + 3) / 4) * 16;              // K elements per wave (multiple of 16)
  const int k_lo = wave * kq, k_hi = min(M, k_lo + kq);
  const T* hp = h + (size_t)s * M;
  const T* emb = nullptr;  // EMB is a template flag so that no load sits behind a per-lane branch
  if constexpr (EMB) emb = token_modality(s, T_, n_vis, mode) ? temb : vemb;
  const float* wp = wg + (size_t)min(i, E - 1) * M;
  const float wmask = i < E ? 1.f : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int KU = 4;  // 16-wide k blocks whose loads are issued together (latency hiding)
  if constexpr (Vec16<T>::N == 8) {
    // bf16 activations: x = bf16(h + embed) is a bf16 MFMA operand as it is; wg goes in as its exact
    // three-way bf16 split (see split3_bf16)
    for (int k0 = k_lo; k0 < k_hi; k0 += 16 * KU) {
      Vec16<T> xa[KU];
      float b[KU][8];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = min(k0 + 16 * u, M - 16) + 8 * hh;  // clamp: blocks past the range are masked below
        xa[u] = load16<T>(hp + k);
        if constexpr (EMB) {
          const Vec16<T> ev = load16<T>(emb + k);
          float xs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xs[j] = xa[u].get(j) + ev.get(j);
          xa[u].set_all(xs);
        }
        const float m = (k0 + 16 * u < k_hi) ? wmask : 0.f;
        const float4 w0 = *reinterpret_cast<const float4*>(wp + k), w1 = *reinterpret_cast<const float4*>(wp + k + 4);
        b[u][0] = w0.x * m; b[u][1] = w0.y * m; b[u][2] = w0.z * m; b[u][3] = w0.w * m;
        b[u][4] = w1.x * m; b[u][5] = w1.y * m; b[u][6] = w1.z * m; b[u][7] = w1.w * m;
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        gate_bf16x8 bh, bm, bl;
        split3_bf16(b[u], bh, bm, bl);
        const gate_bf16x8 av = __builtin_bit_cast(gate_bf16x8, xa[u].raw);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc, 0, 0, 0);
      }
    }
  } else {
    for (int k0 = k_lo; k0 < k_hi; k0 += 16 * KU) {
      float a[KU][8], b[KU][8];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = min(k0 + 16 * u, M - 16) + 8 * hh;  // clamp: blocks past the range are masked below
        const Vec16<T> hv0 = load16<T>(hp + k);
        const Vec16<T> hv1 = load16<T>(hp + k + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[u][j] = hv0.get(j); a[u][4 + j] = hv1.get(j); }
        if constexpr (EMB) {
          const Vec16<T> e0 = load16<T>(emb + k), e1 = load16<T>(emb + k + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[u][j] += e0.get(j); a[u][4 + j] += e1.get(j); }
        }
        const float m = (k0 + 16 * u < k_hi) ? wmask : 0.f;
        const float4 w0 = *reinterpret_cast<const float4*>(wp + k), w1 = *reinterpret_cast<const float4*>(wp + k + 4);
        b[u][0] = w0.x * m; b[u][1] = w0.y * m; b[u][2] = w0.z * m; b[u][3] = w0.w * m;
        b[u][4] = w1.x * m; b[u][5] = w1.y * m; b[u][6] = w1.z * m; b[u][7] = w1.w * m;
      }
#pragma unroll
      for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], b[u][j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // D: column (expert) = lane & 31, row (token) = (r & 3) + 8 * (r >> 2) + 4 * hh; wave w finishes
  // registers 4w .. 4w+3
  if (i < E) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave * 4 + q;
      const float v = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (s0 + row < S) logits[(size_t)(s0 + row) * E + i] = v;
    }
  }
}

// bf16 activations, round 3: the same arithmetic per logit as gate_logits_mfma_kernel's bf16 branch (bf16 x, wg as
// its exact three-way bf16 split, products of an fp32 FMA), but a workgroup WALKS token tiles.  One 512-thread
// workgroup per CU; its 8 waves split the K axis (KB 16-wide k blocks each) and keep their slice of wg -- split once
// per workgroup, not once per 32 tokens: the split is ~10 VALU instructions per element and was redone by every one
// of the S / 32 blocks -- in registers across the tiles; the rows of the next tile are in flight while the current
// one is multiplied, reduced over the waves through LDS in wave order (deterministic) and stored.
// (body shared by the two kernels below; TO_LDS: the logits of the workgroup's tiles are also left in `lds_logits`
// [tiles * 32][33], `logits` may then be NULL)
template <bool EMB, int KB, bool TO_LDS>
__device__ __forceinline__ void gate_logits_walk_body(const bf16_t* __restrict__ h, const bf16_t* __restrict__ vemb,
                                                      const bf16_t* __restrict__ temb, const float* __restrict__ wg,
                                                      float* __restrict__ logits, int S, int T_, int n_vis, int mode, int M,
                                                      int E, int ntiles, int tiles_per_block, float (*red)[16][64],
                                                      float* lds_logits) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int i = lane & 31, hh = lane >> 5;
  const int k_lo = wave * KB * 16;
  const int t_begin = blockIdx.x * tiles_per_block, t_end = min(ntiles, t_begin + tiles_per_block);
  if (t_begin >= t_end) return;
  // per-lane k offsets of the KB blocks (clamped: blocks past M are masked through wg)
  int kof[KB];
#pragma unroll
  for (int u = 0; u < KB; ++u) kof[u] = min(k_lo + 16 * u, M - 16) + 8 * hh;
  // rows of the tile being multiplied (cur) and of the next one (nxt, in flight meanwhile)
  uint4 cur[KB], nxt[KB];
  {
    const bf16_t* hp = h + (size_t)min(t_begin * 32 + i, S - 1) * M;
#pragma unroll
    for (int u = 0; u < KB; ++u) cur[u] = *reinterpret_cast<const uint4*>(hp + kof[u]);
    const bf16_t* hq = h + (size_t)min((t_begin + 1) * 32 + i, S - 1) * M;  // (clamped: unused past the last tile)
#pragma unroll
    for (int u = 0; u < KB; ++u) nxt[u] = *reinterpret_cast<const uint4*>(hq + kof[u]);
  }
  // my slice of wg (expert i), split once
  gate_bf16x8 bh[KB], bm[KB], bl[KB];
  {
    const float* wp = wg + (size_t)min(i, E - 1) * M;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const float m = (i < E && k_lo + 16 * u < M) ? 1.f : 0.f;
      const float4 w0 = *reinterpret_cast<const float4*>(wp + kof[u]), w1 = *reinterpret_cast<const float4*>(wp + kof[u] + 4);
      const float b[8] = {w0.x * m, w0.y * m, w0.z * m, w0.w * m, w1.x * m, w1.y * m, w1.z * m, w1.w * m};
      split3_bf16(b, bh[u], bm[u], bl[u]);
    }
  }
  for (int t = t_begin; t < t_end; ++t) {
    uint4 ev[KB];
    if constexpr (EMB) {
      const int s = min(t * 32 + i, S - 1);
      const bf16_t* emb = token_modality(s, T_, n_vis, mode) ? temb : vemb;
#pragma unroll
      for (int u = 0; u < KB; ++u) ev[u] = *reinterpret_cast<const uint4*>(emb + kof[u]);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      uint4 raw = cur[u];
      if constexpr (EMB) {  // x = bf16(h + embed), one rounding
        const uint32_t *xw = &raw.x, *ew = &ev[u].x;
        uint32_t o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float lo = __uint_as_float(xw[p] << 16) + __uint_as_float(ew[p] << 16);
          const float hi = __uint_as_float(xw[p] & 0xffff0000u) + __uint_as_float(ew[p] & 0xffff0000u);
          typedef __attribute__((ext_vector_type(2))) float f2;
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          const f2 pr = {lo, hi};
          o[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, b2));
        }
        raw = make_uint4(o[0], o[1], o[2], o[3]);
      }
      const gate_bf16x8 av = __builtin_bit_cast(gate_bf16x8, raw);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl[u], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bm[u], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh[u], acc, 0, 0, 0);
    }
    // next tile's rows move up (they have had this tile's multiply to land); the tile after it goes out
#pragma unroll
    for (int u = 0; u < KB; ++u) cur[u] = nxt[u];
    if (t + 2 < t_end) {
      const bf16_t* hq = h + (size_t)min((t + 2) * 32 + i, S - 1) * M;
#pragma unroll
      for (int u = 0; u < KB; ++u) nxt[u] = *reinterpret_cast<const uint4*>(hq + kof[u]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // D: column (expert) = lane & 31, row (token) = (r & 3) + 8 * (r >> 2) + 4 * hh; wave w finishes registers 2w, 2w+1
    if (i < E) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = wave * 2 + q;
        float v = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) v += red[w][r][lane];
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (logits && t * 32 + row < S) logits[(size_t)(t * 32 + row) * E + i] = v;
        if constexpr (TO_LDS) lds_logits[((t - t_begin) * 32 + row) * 33 + i] = v;
      }
    }
    __syncthreads();
  }
}

template <bool EMB, int KB>
__global__ __launch_bounds__(512) void gate_logits_walk_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ vemb,
                                                               const bf16_t* __restrict__ temb,
                                                               const float* __restrict__ wg, float* __restrict__ logits,
                                                               int S, int T_, int n_vis, int mode, int M, int E,
                                                               int ntiles, int tiles_per_block) {
  __shared__ float red[8][16][64];
  gate_logits_walk_body<EMB, KB, false>(h, vemb, temb, wg, logits, S, T_, n_vis, mode, M, E, ntiles, tiles_per_block, red, nullptr);
}

// The walking logits kernel with the gate's selection pass in its tail (the fused layer calls; E <= 32 and at most
// kAssign32MaxBlocks blocks of 64 tokens): workgroup b forms the logits of tokens [64 b, 64 b + 64) -- two tiles --,
// leaves them in LDS and runs gate_select32_body on them: the logits never travel through memory and the selection
// costs no launch of its own (7.5 us for 6 MB, all of it latency).  Same arithmetic per logit and per selection as
// gate_logits_walk_kernel followed by gate_select32_kernel, hence the same results bit for bit.
template <bool EMB, int KB, bool E32>
__global__ __launch_bounds__(512) void gate_logits_select_kernel(
    const bf16_t* __restrict__ h, const bf16_t* __restrict__ vemb, const bf16_t* __restrict__ temb,
    const float* __restrict__ wg, float* __restrict__ logits, int S, int T_, int n_vis, int mode, int M, int E, int ntiles,
    const float* __restrict__ noise, int noise_is_uniform, float* __restrict__ gates, int32_t* __restrict__ idx,
    int32_t* __restrict__ hist, int32_t* __restrict__ rank, float* __restrict__ gsum, int32_t* __restrict__ slot_token,
    float* __restrict__ slot_weight, int num_slots) {
  __shared__ float red[8][16][64];
  __shared__ float lds_logits[kGateBlock * 33];
  __shared__ int ci1[kGateBlock], ci2[kGateBlock];
  __shared__ float tile[kGateBlock][33];
  __shared__ float gpart[8][32];
  gate_logits_walk_body<EMB, KB, true>(h, vemb, temb, wg, logits, S, T_, n_vis, mode, M, E, ntiles, 2, red, lds_logits);
  __syncthreads();
  gate_select32_body<E32, true>(nullptr, lds_logits, noise, noise_is_uniform, S, E, gates, idx, hist, rank, gsum, slot_token,
                                slot_weight, num_slots, 1, ci1, ci2, tile, gpart);
}

// dwg partials: D[32 experts x 32 "virtual" cols] += A[32 experts x 2 tokens] * B[2 tokens x 32 cols].
// Block (x = token chunk of TS tokens, y = group of 256 columns); its 4 waves split the chunk's tokens
// and are summed through LDS in a fixed order (deterministic).  Lane i loads the 8 contiguous elements
// at column 256 y + 8 i of a token with one (bf16) or two (fp32) 16-byte loads and feeds element t to
// MFMA t, i.e. accumulator t's column j stands for real column 256 y + 8 j + t.
template <typename T, bool EMB>
__global__ __launch_bounds__(256, 2) void gate_wgrad_mfma_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                              const T* __restrict__ temb,
                                                              const float* __restrict__ dlogits,
                                                              float* __restrict__ part, int S, int T_, int n_vis,
                                                              int mode, int M, int E, int TS) {
  __shared__ float red[2][8][16][64];  // two waves' accumulators
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int i = lane & 31, hh = lane >> 5;
  const int col0 = blockIdx.y * 256 + i * 8;
  const bool col_ok = col0 < M;   // M % 8 == 0
  const int colc = col_ok ? col0 : 0;  // lanes past the last column load column 0 and store nothing
  const int ic = min(i, E - 1);
  const int per_wave = ((TS / 4 + 1) / 2) * 2;  // tokens per wave, even
  const int s_begin = min(S, blockIdx.x * TS + wave * per_wave);
  const int s_end = min(min(S, blockIdx.x * TS + TS), s_begin + per_wave);
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // Loads are unconditional (token index clamped, dead tokens masked through their dlogits factor) so
  // that all PU token pairs' loads are in flight together; a branch per load would serialise them.
  constexpr int PU = 4;  // token pairs whose loads are issued together
  for (int s2 = s_begin; s2 < s_end; s2 += 2 * PU) {
    float a[PU], b[PU][8];
    Vec16<T> hv[PU][Vec16<T>::N == 8 ? 1 : 2], ev[PU][Vec16<T>::N == 8 ? 1 : 2];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int s = s2 + 2 * u + hh;
      const int sc = min(s, S - 1);
      a[u] = dlogits[(size_t)sc * E + ic];
      const T* hp = h + (size_t)sc * M + colc;
      hv[u][0] = load16<T>(hp);
      if constexpr (Vec16<T>::N != 8) hv[u][1] = load16<T>(hp + 4);
      if constexpr (EMB) {
        const T* emb = (token_modality(sc, T_, n_vis, mode) ? temb : vemb) + colc;
        ev[u][0] = load16<T>(emb);
        if constexpr (Vec16<T>::N != 8) ev[u][1] = load16<T>(emb + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int s = s2 + 2 * u + hh;
      a[u] = (s < s_end && i < E) ? a[u] : 0.f;
      if constexpr (Vec16<T>::N == 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          b[u][j] = hv[u][0].get(j);
          if constexpr (EMB) b[u][j] = to_f32<T>(from_f32<T>(b[u][j] + ev[u][0].get(j)));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          b[u][j] = hv[u][0].get(j);
          b[u][4 + j] = hv[u][1].get(j);
          if constexpr (EMB) { b[u][j] += ev[u][0].get(j); b[u][4 + j] += ev[u][1].get(j); }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PU; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
  }
  // (w0 + w2) + (w1 + w3)
  if (wave >= 2) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 2][t][r][lane] = acc[t][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[wave][t][r][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[0][t][r][lane] = acc[t][r];
  }
  __syncthreads();
  if (wave != 0 || !col_ok) return;
  float* out = part + (size_t)blockIdx.x * E * M;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;  // expert
    if (row >= E) continue;
    float* o = out + (size_t)row * M + col0;
    *reinterpret_cast<float4*>(o) = make_float4(acc[0][r] + red[0][0][r][lane], acc[1][r] + red[0][1][r][lane],
                                                acc[2][r] + red[0][2][r][lane], acc[3][r] + red[0][3][r][lane]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4][r] + red[0][4][r][lane], acc[5][r] + red[0][5][r][lane],
                                                    acc[6][r] + red[0][6][r][lane], acc[7][r] + red[0][7][r][lane]);
  }
}

// bf16 activations: the body lives in gate_wgrad_bf16.h (shared with permute.hip's combined launch)
template <bool EMB>
__global__ __launch_bounds__(256, 2) void gate_wgrad_bf16_kernel(const bf16_t* __restrict__ h,
                                                              const bf16_t* __restrict__ vemb,
                                                              const bf16_t* __restrict__ temb,
                                                              const float* __restrict__ dlogits,
                                                              float* __restrict__ part, int S, int T_, int n_vis,
                                                              int mode, int M, int E, int TS) {
  gate_wgrad_bf16_body<EMB>(h, vemb, temb, dlogits, part, S, T_, n_vis, mode, M, E, TS, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void sum_partials4_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            int n, int nparts) {
  __shared__ float red[4][64];
  sum_partials4_body(part, out, n, nparts, blockIdx.x, red);
}

// =================================================================== host launchers
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static int gate_cu_count() {  // of the current device
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;
  return n;
}

template <typename T>
static int launch_logits(const void* h, const void* ve, const void* te, const float* wg, float* logits, int S, int T_,
                         int n_vis, int mode, int M, int E, hipStream_t st) {
  if constexpr (Vec16<T>::N == 8) {
    // bf16: the tile-walking kernel (one workgroup per CU, wg split once per workgroup); K slices of up to 8 blocks
    const int kb = ceil_div(M / 16, 8);
    if (E <= 32 && M % 16 == 0 && kb <= 8) {
      const int ntiles = ceil_div(S, 32);
      const int tpb = ceil_div(ntiles, gate_cu_count());
      const dim3 grid(ceil_div(ntiles, tpb));
#define EVE_WALK(KB_)                                                                                                   \
  case KB_:                                                                                                             \
    if (mode != EVE_MODE_NONE)                                                                                          \
      hipLaunchKernelGGL((gate_logits_walk_kernel<true, KB_>), grid, dim3(512), 0, st, (const bf16_t*)h,               \
                         (const bf16_t*)ve, (const bf16_t*)te, wg, logits, S, T_, n_vis, mode, M, E, ntiles, tpb);      \
    else                                                                                                                \
      hipLaunchKernelGGL((gate_logits_walk_kernel<false, KB_>), grid, dim3(512), 0, st, (const bf16_t*)h,              \
                         (const bf16_t*)ve, (const bf16_t*)te, wg, logits, S, T_, n_vis, mode, M, E, ntiles, tpb);      \
    break;
      switch (kb) {
        EVE_WALK(1) EVE_WALK(2) EVE_WALK(3) EVE_WALK(4) EVE_WALK(5) EVE_WALK(6) EVE_WALK(7) EVE_WALK(8)
      }
#undef EVE_WALK
      EVE_CHECK_LAUNCH("gate_logits_walk_kernel");
      return EVE_OK;
    }
  }
  if (E <= 32 && M % 16 == 0) {
    if (mode != EVE_MODE_NONE)
      hipLaunchKernelGGL((gate_logits_mfma_kernel<T, true>), dim3(ceil_div(S, 32)), dim3(256), 0, st, (const T*)h,
                         (const T*)ve, (const T*)te, wg, logits, S, T_, n_vis, mode, M, E);
    else
      hipLaunchKernelGGL((gate_logits_mfma_kernel<T, false>), dim3(ceil_div(S, 32)), dim3(256), 0, st, (const T*)h,
                         (const T*)ve, (const T*)te, wg, logits, S, T_, n_vis, mode, M, E);
    EVE_CHECK_LAUNCH("gate_logits_mfma_kernel");
    return EVE_OK;
  }
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

// ---- Gumbel(0,1) noise from standard-uniform draws -----------------------------------------------
// The reference draws torch.distributions.Gumbel(0,1).rsample (sharded_moe.py:70-77,:186), which torch
// evaluates as ~10 elementwise kernels over one torch.rand() draw: u = tiny + r * ((1 - eps) - tiny),
// then 0 + (-1) * log(0 + (-1) * log(u)), every step rounded to fp32.  This is the same arithmetic, in
// the same order and without contraction, in one pass.
__global__ void gumbel_from_uniform_kernel(const float* __restrict__ r, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = gumbel_of_uniform(r[i]);
}

extern "C" int eve_gumbel_from_uniform(const float* uniform, float* noise, long n, void* stream) {
  EVE_REQUIRE(uniform && noise, EVE_ERR_ARG, "eve_gumbel_from_uniform: null pointer");
  EVE_REQUIRE(n > 0, EVE_ERR_ARG, "eve_gumbel_from_uniform: n=%ld", n);
  const int block = 256;
  const long want = (n + block - 1) / block;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(gumbel_from_uniform_kernel, dim3(grid), dim3(block), 0, S_(stream), uniform, noise, n);
  EVE_CHECK_LAUNCH("gumbel_from_uniform_kernel");
  return EVE_OK;
}

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

// Launch sequence of the fused gate; `ws` is device scratch laid out by gate_ws_layout (every word of it is
// written before it is read, so its previous contents never matter).
static int gate_top2_launch(const float* logits, const float* noise, int noise_is_uniform, int S, int E,
                            int C, float* gates, int32_t* idx, int32_t* loc, int32_t* slot, float* weight,
                            int32_t* slot_token, float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux,
                            char* ws, const GateWs& w, hipStream_t st, int64_t* exp_counts_mirror = nullptr) {
  int32_t* hist = reinterpret_cast<int32_t*>(ws + w.hist);
  int32_t* boff = reinterpret_cast<int32_t*>(ws + w.boff);
  int32_t* rank = reinterpret_cast<int32_t*>(ws + w.rank);
  float* gsum = reinterpret_cast<float*>(ws + w.gsum);
  int32_t* tot = reinterpret_cast<int32_t*>(ws + w.tot);
  const int fused_assign = E <= 32 && w.nblk <= kAssign32MaxBlocks ? 1 : 0;
  // noise source 2: `noise` is the device generator state; the kernel after the selection advances it
  unsigned long long* rng_state = noise_is_uniform == kNoisePhilox
                                      ? reinterpret_cast<unsigned long long*>(const_cast<float*>(noise)) : nullptr;
  const unsigned long long rng_inc = ((unsigned long long)S * E + 3) / 4;
  if (E <= 32) {
    if (E == 32)
      hipLaunchKernelGGL(gate_select32_kernel<true>, dim3(w.nblk), dim3(256), 0, st, logits, noise, noise_is_uniform, S, E,
                         gates, idx, hist, rank, gsum, slot_token, slot_weight, E * C, fused_assign);
    else
      hipLaunchKernelGGL(gate_select32_kernel<false>, dim3(w.nblk), dim3(256), 0, st, logits, noise, noise_is_uniform, S, E,
                         gates, idx, hist, rank, gsum, slot_token, slot_weight, E * C, fused_assign);
    EVE_CHECK_LAUNCH("gate_select32_kernel");
  } else {
    const int use_tile = E <= 128 ? 1 : 0;
    const size_t lds1 = use_tile ? (size_t)kGateBlock * (E + 1) * 4 : 0;
    hipLaunchKernelGGL(gate_select_kernel, dim3(w.nblk), dim3(kGateBlock), lds1, st, logits, noise, noise_is_uniform, S, E,
                       gates, idx, hist, rank, gsum, slot_token, slot_weight, E * C, use_tile);
    EVE_CHECK_LAUNCH("gate_select_kernel");
  }
  if (fused_assign) {
    hipLaunchKernelGGL(gate_assign32_kernel, dim3(w.nblk), dim3(512), 0, st, gates, idx, rank, hist, gsum, S, E, C, loc,
                       slot, weight, slot_token, slot_weight, exp_counts, rows, l_aux, rng_state, rng_inc, exp_counts_mirror);
    EVE_CHECK_LAUNCH("gate_assign32_kernel");
    return EVE_OK;
  }
  const size_t lds2 = (size_t)4 * E * 4;
  hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(1024), lds2, st, hist, boff, gsum, tot, exp_counts, rows, l_aux, S,
                     E, C, w.nblk, rng_state, rng_inc, exp_counts_mirror);
  EVE_CHECK_LAUNCH("gate_scan_kernel");
  hipLaunchKernelGGL(gate_assign_kernel, dim3(w.nblk), dim3(kGateBlock), 0, st, gates, idx, rank, boff, tot, S, E, C,
                     loc, slot, weight, slot_token, slot_weight);
  EVE_CHECK_LAUNCH("gate_assign_kernel");
  return EVE_OK;
}

#define EVE_GATE_TOP2_CHECK_ARGS(who)                                                                                  \
  EVE_REQUIRE(logits && noise && gates && idx && loc && slot && weight && slot_token && slot_weight && exp_counts &&  \
                  rows && l_aux,                                                                                       \
              EVE_ERR_ARG, who ": null pointer");                                                                      \
  EVE_REQUIRE(S > 0 && E > 0 && C > 0, EVE_ERR_ARG, who ": bad shape S=%d E=%d C=%d", S, E, C);                        \
  EVE_REQUIRE(noise_is_uniform >= 0 && noise_is_uniform <= 2, EVE_ERR_ARG, who ": noise_is_uniform=%d (0, 1 or 2)",    \
              noise_is_uniform);                                                                                       \
  EVE_REQUIRE(noise_is_uniform != 2 || ((uintptr_t)noise & 7) == 0, EVE_ERR_ARG,                                       \
              who ": the generator state (noise_is_uniform == 2) must be 8-byte aligned");                             \
  EVE_REQUIRE(E <= 1024, EVE_ERR_UNSUPPORTED, who ": E=%d > 1024", E);                                                 \
  EVE_REQUIRE((long long)E * C < (1ll << 31), EVE_ERR_ARG, who ": E*C overflows int32")

namespace eve {
int gate_top2_fwd_mirrored(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C, float* gates,
                           int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token,
                           float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace,
                           size_t workspace_bytes, void* stream, int64_t* exp_counts_mirror);
}
extern "C" int eve_gate_top2_fwd(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C, float* gates,
                                 int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token,
                                 float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return eve::gate_top2_fwd_mirrored(logits, noise, noise_is_uniform, S, E, C, gates, idx, loc, slot, weight, slot_token,
                                     slot_weight, exp_counts, rows, l_aux, workspace, workspace_bytes, stream, nullptr);
}

// eve_gate_top2_fwd with a second copy of exp_counts written by the same kernel (fused.hip: eve_moe_fwd's
// exp_counts_host); arguments already checked by the caller's shape check except what is checked here
namespace eve {
int gate_top2_fwd_mirrored(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C, float* gates,
                           int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token,
                           float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace,
                           size_t workspace_bytes, void* stream, int64_t* exp_counts_mirror) {
  EVE_GATE_TOP2_CHECK_ARGS("eve_gate_top2_fwd");
  EVE_REQUIRE(workspace, EVE_ERR_ARG, "eve_gate_top2_fwd: null workspace");
  const GateWs w = gate_ws_layout(S, E);
  EVE_REQUIRE(workspace_bytes >= w.total, EVE_ERR_WORKSPACE, "eve_gate_top2_fwd: workspace %zu < %zu", workspace_bytes, w.total);
  EVE_REQUIRE(((uintptr_t)workspace & 255) == 0, EVE_ERR_ARG, "eve_gate_top2_fwd: workspace must be 256-byte aligned");
  return gate_top2_launch(logits, noise, noise_is_uniform, S, E, C, gates, idx, loc, slot, weight, slot_token, slot_weight,
                          exp_counts, rows, l_aux, static_cast<char*>(workspace), w, S_(stream), exp_counts_mirror);
}
}  // namespace eve

// ---- eve_gate_logits_fwd + eve_gate_top2_fwd of the fused layer calls in TWO launches (fused.hip) ----------------
// gate_logits_select_kernel (logits and selection) + gate_assign32_kernel.  Only for the shapes those kernels are
// written for; the caller asks gate_logits_top2_fusable first and otherwise makes the two calls.
namespace eve {
bool gate_logits_top2_fusable(int S, int M, int E, int dtype) {
  return dtype == EVE_BF16 && E == 32 && S > 0 && M > 0 && M % 16 == 0 && ceil_div(M / 16, 8) <= 8 &&
         ceil_div(S, kGateBlock) <= kAssign32MaxBlocks;
}

int gate_logits_top2_fused(const void* h, const void* ve, const void* te, const float* wg, const float* noise,
                           int noise_is_uniform, int S, int T_, int n_vis, int mode, int M, int E, int C, float* gates,
                           int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token, float* slot_weight,
                           int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace, size_t workspace_bytes,
                           void* stream, int64_t* exp_counts_mirror) {
  const float* logits = wg;  // (the argument check below wants a non-null `logits`; none exists on this path)
  EVE_GATE_TOP2_CHECK_ARGS("eve_moe_fwd (gate)");
  EVE_REQUIRE(h && wg && workspace, EVE_ERR_ARG, "eve_moe_fwd (gate): null pointer");
  EVE_REQUIRE(gate_logits_top2_fusable(S, M, E, EVE_BF16), EVE_ERR_UNSUPPORTED, "eve_moe_fwd (gate): shape not fusable");
  EVE_REQUIRE(mode >= EVE_MODE_NONE && mode <= EVE_MODE_TEXT, EVE_ERR_ARG, "eve_moe_fwd (gate): bad mode %d", mode);
  if (mode != EVE_MODE_NONE) {
    EVE_REQUIRE(ve && te, EVE_ERR_ARG, "eve_moe_fwd (gate): modality embeddings required for mode %d", mode);
    EVE_REQUIRE(T_ > 0 && S % T_ == 0 && n_vis >= 0, EVE_ERR_ARG, "eve_moe_fwd (gate): S=%d not a multiple of T=%d", S, T_);
  }
  const GateWs w = gate_ws_layout(S, E);
  EVE_REQUIRE(workspace_bytes >= w.total, EVE_ERR_WORKSPACE, "eve_moe_fwd (gate): workspace %zu < %zu", workspace_bytes, w.total);
  EVE_REQUIRE(((uintptr_t)workspace & 255) == 0, EVE_ERR_ARG, "eve_moe_fwd (gate): workspace must be 256-byte aligned");
  char* ws = static_cast<char*>(workspace);
  int32_t* hist = reinterpret_cast<int32_t*>(ws + w.hist);
  int32_t* rank = reinterpret_cast<int32_t*>(ws + w.rank);
  float* gsum = reinterpret_cast<float*>(ws + w.gsum);
  hipStream_t st = S_(stream);
  const int ntiles = ceil_div(S, 32), kb = ceil_div(M / 16, 8);
#define EVE_WALKSEL(KB_)                                                                                               \
  case KB_:                                                                                                            \
    if (mode != EVE_MODE_NONE)                                                                                         \
      hipLaunchKernelGGL((gate_logits_select_kernel<true, KB_, true>), dim3(w.nblk), dim3(512), 0, st, (const bf16_t*)h, \
                         (const bf16_t*)ve, (const bf16_t*)te, wg, (float*)nullptr, S, T_, n_vis, mode, M, E, ntiles,  \
                         noise, noise_is_uniform, gates, idx, hist, rank, gsum, slot_token, slot_weight, E * C);       \
    else                                                                                                               \
      hipLaunchKernelGGL((gate_logits_select_kernel<false, KB_, true>), dim3(w.nblk), dim3(512), 0, st, (const bf16_t*)h, \
                         (const bf16_t*)ve, (const bf16_t*)te, wg, (float*)nullptr, S, T_, n_vis, mode, M, E, ntiles,  \
                         noise, noise_is_uniform, gates, idx, hist, rank, gsum, slot_token, slot_weight, E * C);       \
    break;
  switch (kb) {
    EVE_WALKSEL(1) EVE_WALKSEL(2) EVE_WALKSEL(3) EVE_WALKSEL(4) EVE_WALKSEL(5) EVE_WALKSEL(6) EVE_WALKSEL(7) EVE_WALKSEL(8)
  }
#undef EVE_WALKSEL
  EVE_CHECK_LAUNCH("gate_logits_select_kernel");
  unsigned long long* rng_state = noise_is_uniform == kNoisePhilox
                                      ? reinterpret_cast<unsigned long long*>(const_cast<float*>(noise)) : nullptr;
  const unsigned long long rng_inc = ((unsigned long long)S * E + 3) / 4;
  hipLaunchKernelGGL(gate_assign32_kernel, dim3(w.nblk), dim3(512), 0, st, gates, idx, rank, hist, gsum, S, E, C, loc, slot,
                     weight, slot_token, slot_weight, exp_counts, rows, l_aux, rng_state, rng_inc, exp_counts_mirror);
  EVE_CHECK_LAUNCH("gate_assign32_kernel");
  return EVE_OK;
}
}  // namespace eve

// ---- the same gate for callers that cannot allocate (include/eve_moe.h: eve_gate_top2_fwd_noalloc) ------------
// The scratch of the gate kernels lives in the code object instead of in caller memory.  Every GPU the library
// runs on holds its own copy of this array at its own address until the process ends; `scratch` in host code
// is only the symbol's host-side stand-in, never a device address: the address is looked up once per device
// (that device current) and only the looked-up pointer is handed to kernels.
static __device__ __align__(256) unsigned char scratch[EVE_GATE_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices];  // per device, filled on the first call with that device current

int gate_scratch_for_current_device(char** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("eve_gate_top2_fwd_noalloc: hipGetDevice failed or device %d out of range: %s", dev, hipGetErrorString(e));
    return EVE_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      set_error("eve_gate_top2_fwd_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
      return EVE_ERR_LAUNCH;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = static_cast<char*>(g_scratch_ptr[dev]);
  return EVE_OK;
}
}  // namespace

extern "C" int eve_gate_top2_fwd_noalloc(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C,
                                         float* gates, int32_t* idx, int32_t* loc, int32_t* slot, float* weight,
                                         int32_t* slot_token, float* slot_weight, int64_t* exp_counts, int32_t* rows,
                                         float* l_aux, void* stream) {
  EVE_GATE_TOP2_CHECK_ARGS("eve_gate_top2_fwd_noalloc");
  const GateWs w = gate_ws_layout(S, E);
  EVE_REQUIRE(w.total <= (size_t)EVE_GATE_NOALLOC_SCRATCH_BYTES, EVE_ERR_WORKSPACE,
              "eve_gate_top2_fwd_noalloc: S=%d tokens x E=%d experts need %zu bytes of scratch, the built-in buffer holds %zu; "
              "use eve_gate_top2_fwd with a caller workspace",
              S, E, w.total, (size_t)EVE_GATE_NOALLOC_SCRATCH_BYTES);
  char* ws = nullptr;
  const int rc = gate_scratch_for_current_device(&ws);
  if (rc != EVE_OK) return rc;
  return gate_top2_launch(logits, noise, noise_is_uniform, S, E, C, gates, idx, loc, slot,
                          weight, slot_token, slot_weight, exp_counts, rows, l_aux, ws, w, S_(stream));
}

extern "C" int eve_gate_top2_bwd(const float* gates, const int32_t* idx, const int32_t* loc, const float* d_weight,
                                 const float* d_l_aux, const int64_t* exp_counts, int S, int E, int C, float* dlogits,
                                 void* stream) {
  EVE_REQUIRE(gates && idx && loc && exp_counts && dlogits, EVE_ERR_ARG, "eve_gate_top2_bwd: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && C > 0, EVE_ERR_ARG, "eve_gate_top2_bwd: bad shape");
  if (E <= 32) {
    const int grid = max(1, min(ceil_div(S, 8), 4096));
    hipLaunchKernelGGL(gate_bwd32_kernel, dim3(grid), dim3(256), 0, S_(stream), gates, idx, loc, d_weight, d_l_aux,
                       exp_counts, S, E, C, dlogits);
    EVE_CHECK_LAUNCH("gate_bwd32_kernel");
    return EVE_OK;
  }
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(S, 256)), dim3(256), 0, S_(stream), gates, idx, loc, d_weight,
                     d_l_aux, exp_counts, S, E, C, dlogits);
  EVE_CHECK_LAUNCH("gate_bwd_kernel");
  return EVE_OK;
}

static const int kWgradTS = 64;       // VALU kernel
static const int kWgradMfmaTS = 96;   // fp32-MFMA kernel: tokens per block (24 per wave)
static_assert(kWgradBf16TS >= kWgradMfmaTS, "the workspace sized for the fp32 kernel's partials covers the bf16 kernel's");
static inline bool wgrad_use_mfma(int M, int E) { return E <= 32 && M % 8 == 0; }

extern "C" size_t eve_gate_wgrad_workspace_bytes(int S, int M, int E) {
  if (S <= 0) return 0;
  return (size_t)ceil_div(S, wgrad_use_mfma(M, E) ? kWgradMfmaTS : kWgradTS) * E * M * sizeof(float);
}

extern "C" int eve_gate_wgrad(const void* h, const void* vision_embed, const void* text_embed, const float* dlogits,
                              float* dwg, int S, int T, int n_vis, int mode, int M, int E, int dtype, void* workspace,
                              size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(h && dlogits && dwg && workspace, EVE_ERR_ARG, "eve_gate_wgrad: null pointer");
  EVE_REQUIRE(S > 0 && E > 0 && M % 8 == 0, EVE_ERR_ARG, "eve_gate_wgrad: bad shape");
  if (mode != EVE_MODE_NONE) EVE_REQUIRE(vision_embed && text_embed && T > 0, EVE_ERR_ARG, "eve_gate_wgrad: embeddings required");
  const size_t need = eve_gate_wgrad_workspace_bytes(S, M, E);
  EVE_REQUIRE(workspace_bytes >= need, EVE_ERR_WORKSPACE, "eve_gate_wgrad: workspace %zu < %zu", workspace_bytes, need);
  float* part = static_cast<float*>(workspace);
  hipStream_t st = S_(stream);
  if (wgrad_use_mfma(M, E)) {
    EVE_REQUIRE(dtype == EVE_F32 || dtype == EVE_BF16, EVE_ERR_ARG, "eve_gate_wgrad: bad dtype %d", dtype);
    const int nbm = ceil_div(S, dtype == EVE_BF16 ? kWgradBf16TS : kWgradMfmaTS);
    const dim3 grid(nbm, ceil_div(M, 256));
#define EVE_WG_LAUNCH(TT, EMB)                                                                                    \
  hipLaunchKernelGGL((gate_wgrad_mfma_kernel<TT, EMB>), grid, dim3(256), 0, st, (const TT*)h, (const TT*)vision_embed, \
                     (const TT*)text_embed, dlogits, part, S, T, n_vis, mode, M, E, kWgradMfmaTS)
    if (dtype == EVE_F32) {
      if (mode != EVE_MODE_NONE) EVE_WG_LAUNCH(float, true); else EVE_WG_LAUNCH(float, false);
    } else if (mode != EVE_MODE_NONE) {
      hipLaunchKernelGGL(gate_wgrad_bf16_kernel<true>, grid, dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)vision_embed,
                         (const bf16_t*)text_embed, dlogits, part, S, T, n_vis, mode, M, E, kWgradBf16TS);
    } else {
      hipLaunchKernelGGL(gate_wgrad_bf16_kernel<false>, grid, dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)vision_embed,
                         (const bf16_t*)text_embed, dlogits, part, S, T, n_vis, mode, M, E, kWgradBf16TS);
    }
#undef EVE_WG_LAUNCH
    EVE_CHECK_LAUNCH("gate_wgrad_mfma_kernel");
    const int nn = E * M;
    hipLaunchKernelGGL(sum_partials4_kernel, dim3(ceil_div(nn, 64)), dim3(256), 0, st, part, dwg, nn, nbm);
    EVE_CHECK_LAUNCH("sum_partials4_kernel");
    return EVE_OK;
  }
  const int nb = ceil_div(S, kWgradTS);  // more than 32 experts (or M % 8 != 0): VALU kernel, 32 experts per block
  const size_t lds = (size_t)kWgradTS * 32 * 4;
  const dim3 vgrid(nb, ceil_div(E, 32));
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((gate_wgrad_partial_kernel<float, 32>), vgrid, dim3(256), lds, st, (const float*)h,
                       (const float*)vision_embed, (const float*)text_embed, dlogits, part, S, T, n_vis, mode, M, E,
                       kWgradTS);
  else if (dtype == EVE_BF16)
    hipLaunchKernelGGL((gate_wgrad_partial_kernel<bf16_t, 32>), vgrid, dim3(256), lds, st, (const bf16_t*)h,
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
