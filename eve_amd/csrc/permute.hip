// Dispatch / combine permutes: the coalesced row gathers that replace the reference's dense
// one-hot einsums (models/sharded_moe.py:367 and :377) and their autograd backward, plus the
// modality-embedding gradient.  HBM-bound row movers: one wave per row, 16 B per lane.
#include "common.h"
#include "gate_bwd32.h"
#include "gate_wgrad_bf16.h"

namespace eve {

// ----------------------------------------------------------------------------- dispatch fwd
// xd[slot] = round(h[tok] + emb(tok)) or 0.  One wave per slot row.
template <typename T>
__global__ __launch_bounds__(256) void dispatch_fwd_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                           const T* __restrict__ temb,
                                                           const int32_t* __restrict__ slot_token, T* __restrict__ xd,
                                                           int T_, int n_vis, int mode, int M, int num_slots) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int slot = wave; slot < num_slots; slot += nwaves) {
    const int tok = slot_token[slot];
    T* dst = xd + (size_t)slot * M;
    if (tok < 0) {
      for (int c = lane; c < chunks; c += 64) store16<T>(dst + c * N, zero16<T>());
      continue;
    }
    const T* src = h + (size_t)tok * M;
    const T* emb = nullptr;
    if (mode != EVE_MODE_NONE) emb = token_modality(tok, T_, n_vis, mode) ? temb : vemb;
    for (int c = lane; c < chunks; c += 64) {
      Vec16<T> v = load16<T>(src + c * N);
      if (emb) {
        const Vec16<T> e = load16<T>(emb + c * N);
        float xs[N];
#pragma unroll
        for (int j = 0; j < N; ++j) xs[j] = v.get(j) + e.get(j);
        v.set_all(xs);
      }
      store16<T>(dst + c * N, v);
    }
  }
}

// ----------------------------------------------------------------------------- dispatch bwd
// dh[s] = dxd[slot1] + dxd[slot2] + dlogits[s,:] @ wg.  A block owns TOKB consecutive tokens and
// each thread one 16-byte column chunk, so every wg chunk fetched (L2-resident) is used for TOKB
// tokens; the block's dlogits rows sit in LDS and are read as broadcasts.
constexpr int kDbTok = 8;
template <typename T>
__global__ __launch_bounds__(128) void dispatch_bwd_kernel(const T* __restrict__ dxd, const int32_t* __restrict__ slot,
                                                           const float* __restrict__ dlogits,
                                                           const float* __restrict__ wg, T* __restrict__ dh, int S,
                                                           int M, int E) {
  constexpr int N = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dl = reinterpret_cast<float*>(smem);  // [E][kDbTok]
  const int s0 = blockIdx.x * kDbTok;
  const int ntok = min(kDbTok, S - s0);
  if (dlogits) {
    for (int i = threadIdx.x; i < E * kDbTok; i += blockDim.x) {
      const int e = i / kDbTok, t = i % kDbTok;
      dl[i] = t < ntok ? dlogits[(size_t)(s0 + t) * E + e] : 0.f;
    }
    __syncthreads();
  }
  const int chunks = M / N;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    float acc[kDbTok][N];
#pragma unroll
    for (int t = 0; t < kDbTok; ++t) {
#pragma unroll
      for (int j = 0; j < N; ++j) acc[t][j] = 0.f;
      if (t < ntok) {
        const int s1 = slot[2 * (s0 + t)], s2 = slot[2 * (s0 + t) + 1];
        if (s1 >= 0) {
          const Vec16<T> v = load16<T>(dxd + (size_t)s1 * M + c * N);
#pragma unroll
          for (int j = 0; j < N; ++j) acc[t][j] += v.get(j);
        }
        if (s2 >= 0) {
          const Vec16<T> v = load16<T>(dxd + (size_t)s2 * M + c * N);
#pragma unroll
          for (int j = 0; j < N; ++j) acc[t][j] += v.get(j);
        }
      }
    }
    if (dlogits) {
      for (int e = 0; e < E; ++e) {
        float w[N];
        const float* wp = wg + (size_t)e * M + c * N;
#pragma unroll
        for (int j = 0; j < N; j += 4) {
          const float4 wv = *reinterpret_cast<const float4*>(wp + j);
          w[j] = wv.x; w[j + 1] = wv.y; w[j + 2] = wv.z; w[j + 3] = wv.w;
        }
        const float4 d0 = *reinterpret_cast<const float4*>(dl + e * kDbTok);
        const float4 d1 = *reinterpret_cast<const float4*>(dl + e * kDbTok + 4);
        const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int t = 0; t < kDbTok; ++t)
#pragma unroll
          for (int j = 0; j < N; ++j) acc[t][j] = fmaf(d[t], w[j], acc[t][j]);
      }
    }
#pragma unroll
    for (int t = 0; t < kDbTok; ++t) {
      if (t >= ntok) continue;
      Vec16<T> o;
      o.set_all(acc[t]);
      store16<T>(dh + (size_t)(s0 + t) * M + c * N, o);
    }
  }
}

// ------------------------------------------------------------------------------ combine fwd
template <typename T>
__global__ __launch_bounds__(256) void combine_fwd_kernel(const T* __restrict__ y, const int32_t* __restrict__ slot,
                                                          const float* __restrict__ weight, T* __restrict__ out, int S,
                                                          int M) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int s = wave; s < S; s += nwaves) {
    const int s1 = slot[2 * s], s2 = slot[2 * s + 1];
    const float w1 = weight[2 * s], w2 = weight[2 * s + 1];
    for (int c = lane; c < chunks; c += 64) {
      float acc[N];
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] = 0.f;
      if (s1 >= 0) {
        const Vec16<T> v = load16<T>(y + (size_t)s1 * M + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = w1 * v.get(j);
      }
      if (s2 >= 0) {
        const Vec16<T> v = load16<T>(y + (size_t)s2 * M + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = fmaf(w2, v.get(j), acc[j]);
      }
      Vec16<T> o;
      o.set_all(acc);
      store16<T>(out + (size_t)s * M + c * N, o);
    }
  }
}

// ------------------------------------------------------------------------------ combine bwd
// One pass over dout: half a wave per token reads its dout row once and, for each of its (up to two)
// slots, forms d_weight[s,k] = <dout[s], y[slot_k]> and writes dy[slot_k] = slot_weight * dout[s].  Slots
// that hold no token are then zero-filled (dy is read whole by the expert GEMMs of other ranks, and by the
// weight-gradient GEMMs up to rows[e]).
template <typename T>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                          const int32_t* __restrict__ slot,
                                                          const int32_t* __restrict__ slot_token,
                                                          const float* __restrict__ slot_weight, T* __restrict__ dy,
                                                          float* __restrict__ d_weight, int S, int M, int num_slots) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id(), l32 = lane & 31, half = lane >> 5;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int s0 = 2 * wave; s0 < S; s0 += 2 * nwaves) {
    const int s = s0 + half;
    const bool live = s < S;
    const int s1 = live ? slot[2 * s] : -1, s2 = live ? slot[2 * s + 1] : -1;
    const float w1 = s1 >= 0 ? slot_weight[s1] : 0.f, w2 = s2 >= 0 ? slot_weight[s2] : 0.f;
    const T* drow = dout + (size_t)(live ? s : 0) * M;
    const T* y1 = y + (size_t)max(s1, 0) * M;
    const T* y2 = y + (size_t)max(s2, 0) * M;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
    for (int c = l32; c < chunks; c += 32) {
      const Vec16<T> d = load16<T>(drow + c * N);
      const Vec16<T> v1 = load16<T>(y1 + c * N);  // unconditional (row 0 when there is no slot): loads stay in flight together
      const Vec16<T> v2 = load16<T>(y2 + c * N);
      Vec16<T> o1, o2;
      float p1[N], p2[N];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float dj = d.get(j);
        a1 = fmaf(dj, v1.get(j), a1);
        a2 = fmaf(dj, v2.get(j), a2);
        p1[j] = w1 * dj;
        p2[j] = w2 * dj;
      }
      o1.set_all(p1);
      o2.set_all(p2);
      if (s1 >= 0) store16<T>(dy + (size_t)s1 * M + c * N, o1);
      if (s2 >= 0) store16<T>(dy + (size_t)s2 * M + c * N, o2);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      a1 += __shfl_xor(a1, o, 64);
      a2 += __shfl_xor(a2, o, 64);
    }
    if (l32 == 0 && live) {
      d_weight[2 * s] = s1 >= 0 ? a1 : 0.f;
      d_weight[2 * s + 1] = s2 >= 0 ? a2 : 0.f;
    }
  }
  for (int base = wave * 64; base < num_slots; base += nwaves * 64) {  // empty slots
    const int sl = base + lane;
    unsigned long long empty = __ballot(sl < num_slots && slot_token[min(sl, num_slots - 1)] < 0);
    while (empty) {
      const int k = __ffsll((long long)empty) - 1;
      empty &= empty - 1;
      T* dst = dy + (size_t)(base + k) * M;
      for (int c = lane; c < chunks; c += 64) store16<T>(dst + c * N, zero16<T>());
    }
  }
}

// ------------------------------------------------- row movers for M = 32 * NP * (16-byte chunks)
// The model widths (768 = 3 passes, 1024 = 4, ... of 32 lanes x 16 bytes in bf16) get kernels whose trip counts are
// compile-time: half a wave owns a row, a lane NP chunks of it, and all of a row's loads are issued before the
// first use -- with two rows per wave and a grid of several waves per SIMD that keeps enough bytes in flight
// for an HBM-bound mover.  Other widths take the generic kernels above.

// dispatch, TOKEN-major: token s's row (+ embedding) is read ONCE and written to its (up to two) slots -- the
// slot-major kernel above reads it once per slot, far apart in time (different experts), so from HBM twice.
// Slots that hold no token are zero-filled afterwards, as combine_bwd does.
template <typename T, int NP>
__global__ __launch_bounds__(256) void dispatch_fwd_tok_kernel(const T* __restrict__ h, const T* __restrict__ vemb,
                                                               const T* __restrict__ temb, const int32_t* __restrict__ slot,
                                                               const int32_t* __restrict__ slot_token, T* __restrict__ xd,
                                                               int S, int T_, int n_vis, int mode, int num_slots) {
  constexpr int N = Vec16<T>::N, M = 32 * NP * N;
  const int lane = lane_id(), l32 = lane & 31, half = lane >> 5;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = 2 * wave; s0 < S; s0 += 2 * nwaves) {
    const int s = min(s0 + half, S - 1);
    const bool live = s0 + half < S;
    const int2 sl = *reinterpret_cast<const int2*>(slot + 2 * s);
    const T* src = h + (size_t)s * M;
    Vec16<T> v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = load16<T>(src + (p * 32 + l32) * N);
    if (mode != EVE_MODE_NONE) {
      const T* emb = token_modality(s, T_, n_vis, mode) ? temb : vemb;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const Vec16<T> e = load16<T>(emb + (p * 32 + l32) * N);
        float xs[N];
#pragma unroll
        for (int j = 0; j < N; ++j) xs[j] = v[p].get(j) + e.get(j);
        v[p].set_all(xs);
      }
    }
    if (live && sl.x >= 0) {
      T* dst = xd + (size_t)sl.x * M;
#pragma unroll
      for (int p = 0; p < NP; ++p) store16<T>(dst + (p * 32 + l32) * N, v[p]);
    }
    if (live && sl.y >= 0) {
      T* dst = xd + (size_t)sl.y * M;
#pragma unroll
      for (int p = 0; p < NP; ++p) store16<T>(dst + (p * 32 + l32) * N, v[p]);
    }
  }
  for (int base = wave * 64; base < num_slots; base += nwaves * 64) {  // empty slots
    const int sl = base + lane;
    unsigned long long empty = __ballot(sl < num_slots && slot_token[min(sl, num_slots - 1)] < 0);
    while (empty) {
      const int k = __ffsll((long long)empty) - 1;
      empty &= empty - 1;
      T* dst = xd + (size_t)(base + k) * M;
      for (int c = lane; c < M / N; c += 64) store16<T>(dst + c * N, zero16<T>());
    }
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void combine_fwd_np_kernel(const T* __restrict__ y, const int32_t* __restrict__ slot,
                                                             const float* __restrict__ weight, T* __restrict__ out, int S) {
  constexpr int N = Vec16<T>::N, M = 32 * NP * N;
  const int lane = lane_id(), l32 = lane & 31, half = lane >> 5;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = 2 * wave; s0 < S; s0 += 2 * nwaves) {
    const int s = min(s0 + half, S - 1);
    const int2 sl = *reinterpret_cast<const int2*>(slot + 2 * s);
    const float2 w = *reinterpret_cast<const float2*>(weight + 2 * s);
    const T* y1 = y + (size_t)max(sl.x, 0) * M;
    const T* y2 = y + (size_t)max(sl.y, 0) * M;
    Vec16<T> v1[NP], v2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {  // unconditional (row 0 when there is no slot): every load is in flight at once
      v1[p] = load16<T>(y1 + (p * 32 + l32) * N);
      v2[p] = load16<T>(y2 + (p * 32 + l32) * N);
    }
    if (s0 + half < S) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          acc[j] = sl.x >= 0 ? w.x * v1[p].get(j) : 0.f;
          if (sl.y >= 0) acc[j] = fmaf(w.y, v2[p].get(j), acc[j]);
        }
        Vec16<T> o;
        o.set_all(acc);
        store16<T>(out + (size_t)s * M + (p * 32 + l32) * N, o);
      }
    }
  }
}

// (d weight, d l_aux -> d logits of the token, E <= 32: what the fused backward passes so that the gate's backward
// needs no launch of its own; gates == nullptr when the kernel is instantiated without it)
struct GateBwdArgs {
  const float* gates; const int32_t* idx; const int32_t* loc; const float* d_l_aux; const int64_t* exp_counts;
  float* dlogits; int E, C;
};

template <typename T, int NP, bool GB = false>
__global__ __launch_bounds__(256) void combine_bwd_np_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                             const int32_t* __restrict__ slot,
                                                             const int32_t* __restrict__ slot_token,
                                                             const float* __restrict__ slot_weight, T* __restrict__ dy,
                                                             float* __restrict__ d_weight, int S, int num_slots,
                                                             GateBwdArgs gb) {
  constexpr int N = Vec16<T>::N, M = 32 * NP * N;
  const int lane = lane_id(), l32 = lane & 31, half = lane >> 5;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  // GB: the gate's backward for the token, in the same half wave with lane = expert (gate_bwd32.h)
  float aux = 0.f;
  if constexpr (GB) {
    const float dl = gb.d_l_aux ? gb.d_l_aux[0] : 0.f;
    aux = dl * (float)gb.E / ((float)S * (float)S) * (float)gb.exp_counts[min(l32, gb.E - 1)];
  }
  for (int s0 = 2 * wave; s0 < S; s0 += 2 * nwaves) {
    const int s = min(s0 + half, S - 1);
    const bool live = s0 + half < S;
    const int2 sl = *reinterpret_cast<const int2*>(slot + 2 * s);
    const int s1 = live ? sl.x : -1, s2 = live ? sl.y : -1;
    const T* drow = dout + (size_t)s * M;
    const T* y1 = y + (size_t)max(s1, 0) * M;
    const T* y2 = y + (size_t)max(s2, 0) * M;
    Vec16<T> d[NP], v1[NP], v2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      d[p] = load16<T>(drow + (p * 32 + l32) * N);
      v1[p] = load16<T>(y1 + (p * 32 + l32) * N);
      v2[p] = load16<T>(y2 + (p * 32 + l32) * N);
    }
    const float w1 = s1 >= 0 ? slot_weight[s1] : 0.f, w2 = s2 >= 0 ? slot_weight[s2] : 0.f;
    // (GB: the token's gate row and choices, requested with the rows so that the tail does not wait for them)
    float gate_g = 0.f;
    int2 gate_ix = make_int2(0, 0), gate_lc = make_int2(0, 0);
    if constexpr (GB) {
      gate_g = live && l32 < gb.E ? gb.gates[(size_t)s * gb.E + min(l32, gb.E - 1)] : 0.f;
      gate_ix = *reinterpret_cast<const int2*>(gb.idx + 2 * s);
      gate_lc = *reinterpret_cast<const int2*>(gb.loc + 2 * s);
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      Vec16<T> o1, o2;
      float p1[N], p2[N];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float dj = d[p].get(j);
        a1 = fmaf(dj, v1[p].get(j), a1);
        a2 = fmaf(dj, v2[p].get(j), a2);
        p1[j] = w1 * dj;
        p2[j] = w2 * dj;
      }
      o1.set_all(p1);
      o2.set_all(p2);
      if (s1 >= 0) store16<T>(dy + (size_t)s1 * M + (p * 32 + l32) * N, o1);
      if (s2 >= 0) store16<T>(dy + (size_t)s2 * M + (p * 32 + l32) * N, o2);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      a1 += __shfl_xor(a1, o, 64);
      a2 += __shfl_xor(a2, o, 64);
    }
    if (l32 == 0 && live) {
      d_weight[2 * s] = s1 >= 0 ? a1 : 0.f;
      d_weight[2 * s + 1] = s2 >= 0 ? a2 : 0.f;
    }
    if constexpr (GB) {
      const int e = l32, E = gb.E;
      const float dl_e = gate_bwd32_token(gate_g, gate_ix.x, gate_ix.y, gate_lc.x < gb.C, gate_lc.y < gb.C, s1 >= 0 ? a1 : 0.f,
                                          s2 >= 0 ? a2 : 0.f, aux, e, half);
      if (live && e < E) gb.dlogits[(size_t)s * E + e] = dl_e;
    }
  }
  for (int base = wave * 64; base < num_slots; base += nwaves * 64) {  // empty slots
    const int sl = base + lane;
    unsigned long long empty = __ballot(sl < num_slots && slot_token[min(sl, num_slots - 1)] < 0);
    while (empty) {
      const int k = __ffsll((long long)empty) - 1;
      empty &= empty - 1;
      T* dst = dy + (size_t)(base + k) * M;
      for (int c = lane; c < M / N; c += 64) store16<T>(dst + c * N, zero16<T>());
    }
  }
}

// ------------------------------------------------------------------ modality embed gradient
// Column sums of dh by modality.  Stage 1: block = chunk of TS tokens, thread = one 16-byte column
// chunk -> part[chunk][2][M]; stage 2: fixed-order sum over chunks (8 row groups x 32 columns per
// block, LDS tree).  Deterministic.
template <typename T>
__global__ __launch_bounds__(128) void embed_grad_partial_kernel(const T* __restrict__ dh, float* __restrict__ part,
                                                                 int S, int T_, int n_vis, int mode, int M, int TS) {
  constexpr int N = Vec16<T>::N;
  const int chunks = M / N;
  const int s0 = blockIdx.x * TS, s1 = min(S, s0 + TS);
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    float av[N], at[N];
#pragma unroll
    for (int j = 0; j < N; ++j) av[j] = at[j] = 0.f;
    for (int s = s0; s < s1; ++s) {
      const Vec16<T> v = load16<T>(dh + (size_t)s * M + c * N);
      if (token_modality(s, T_, n_vis, mode)) {
#pragma unroll
        for (int j = 0; j < N; ++j) at[j] += v.get(j);
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) av[j] += v.get(j);
      }
    }
    float* pv = part + ((size_t)blockIdx.x * 2 + 0) * M + c * N;
    float* pt = part + ((size_t)blockIdx.x * 2 + 1) * M + c * N;
#pragma unroll
    for (int j = 0; j < N; ++j) { pv[j] = av[j]; pt[j] = at[j]; }
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void embed_grad_final_kernel(const float* __restrict__ part, TO* __restrict__ dv,
                                                               TO* __restrict__ dt, int M, int nchunks) {
  __shared__ float red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;   // over 2*M columns: [vision M | text M] per chunk row
  float acc = 0.f;
  if (col < 2 * M)
    for (int b = ty; b < nchunks; b += 8) acc += part[(size_t)b * 2 * M + col];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < 2 * M) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][tx];
    if (col < M) { if (dv) dv[col] = from_f32<TO>(s); }
    else if (dt) dt[col - M] = from_f32<TO>(s);
  }
}

// ------------------------------------------------------- fused routing backward (E <= 32)
// dh[s] = dxd[slot1] + dxd[slot2] + dlogits[s,:] @ wg, plus the per-block column sums of the ROUNDED dh
// by modality (the modality-embedding gradients), in one pass over dh.  A block owns 32 tokens:
//   1. G[32 tokens x M] = dlogits @ wg on the matrix cores (fp32 activations: v_mfma_f32_32x32x2_f32, exact
//      fp32 FMAs; bf16 activations: both operands split into bf16 hi + lo halves, three bf16 MFMAs),
//      each wave every eighth 32-column tile, written to LDS;
//   2. thread per (token, 16-byte column chunk): G + the two gathered dxd rows -> dh, the rounded values
//      back into G;
//   3. thread per column: sums of G over the block's vision / text tokens -> part[block][2][M].
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int kRbTok = 32;
constexpr int kRbThreads = 512;
// blockIdx.y selects a slab of Mb columns (a multiple of 32), so that a block's LDS image stays small
// enough for several blocks per CU: the kernel is latency-bound and lives on occupancy.
template <typename T>
__global__ __launch_bounds__(kRbThreads) void route_bwd_kernel(const T* __restrict__ dxd, const int32_t* __restrict__ slot,
                                                               const float* __restrict__ dlogits,
                                                               const float* __restrict__ wg, T* __restrict__ dh,
                                                               float* __restrict__ part, int S, int T_, int n_vis,
                                                               int mode, int M, int E, int Mb) {
  constexpr int N = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* G = reinterpret_cast<float*>(smem);
  const int ld = Mb + 4;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * kRbTok;
  const int c0 = blockIdx.y * Mb;            // first column of the slab
  const int mb = min(Mb, M - c0);            // columns in it (multiple of N)
  __shared__ unsigned text_mask;  // bit t: token s0 + t is a text token
  __shared__ float dl[kRbTok][33];  // the block's dlogits rows: read coalesced once, not per wave and lane
  for (int id = threadIdx.x; id < kRbTok * E; id += kRbThreads) {
    const int t = id / E, e = id - t * E;
    dl[t][e] = dlogits[(size_t)min(s0 + t, S - 1) * E + e];
  }
  if (threadIdx.x < 64) {
    const bool tx = lane < kRbTok && token_modality(min(s0 + lane, S - 1), T_, n_vis, mode) != 0;
    const unsigned long long m = __ballot(tx);
    if (threadIdx.x == 0) text_mask = (unsigned)m;
  }
  // The gathers of phase 2 are issued first (up to PRE items per thread, held in registers) so that they
  // are in flight while phase 1 runs: one memory round trip per block instead of two.
  constexpr int PRE = 3;
  const int chunks = mb / N;
  const int items = kRbTok * chunks;
  Vec16<T> p1[PRE], p2[PRE];
  int ps1[PRE], ps2[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int id = min((int)threadIdx.x + k * kRbThreads, items - 1);
    const int t = id / chunks, c = id - t * chunks;
    const int sc = min(s0 + t, S - 1);
    ps1[k] = slot[2 * sc];
    ps2[k] = slot[2 * sc + 1];
    p1[k] = load16<T>(dxd + (size_t)max(ps1[k], 0) * M + c0 + c * N);
    p2[k] = load16<T>(dxd + (size_t)max(ps2[k], 0) * M + c0 + c * N);
  }
  __syncthreads();  // dl
  if constexpr (sizeof(T) == 2) {
    // phase 1, bf16 activations: each fp32 operand as hi + lo bf16 halves, hi*hi + hi*lo + lo*hi on
    // v_mfma_f32_32x32x16_bf16 (fp32 accumulate) -- products good to ~2^-17, far below the bf16 rounding
    // of dh, at a fifth of the fp32 MFMA's issue time
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    const int i = lane & 31, hh = lane >> 5;
    bf16x8_t ah[2], al[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = ks * 16 + 8 * hh + j;
        const float v = e < E ? dl[i][min(e, E - 1)] : 0.f;
        const __bf16 h = static_cast<__bf16>(v);
        ah[ks][j] = h;
        al[ks][j] = static_cast<__bf16>(v - static_cast<float>(h));
      }
    const int ntiles = (mb + 31) / 32;
    for (int tile = wave; tile < ntiles; tile += kRbThreads / 64) {
      const int col = tile * 32 + i;
      const float* wp = wg + c0 + min(col, mb - 1);
      float w[2][8];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[ks][j] = wp[(size_t)min(ks * 16 + 8 * hh + j, E - 1) * M];  // a = 0 masks e >= E
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const __bf16 h = static_cast<__bf16>(w[ks][j]);
          bh[j] = h;
          bl[j] = static_cast<__bf16>(w[ks][j] - static_cast<float>(h));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc, 0, 0, 0);
      }
      if (col < mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) G[((r & 3) + 8 * (r >> 2) + 4 * hh) * ld + col] = acc[r];
      }
    }
  } else {  // phase 1, fp32 activations: exact fp32 FMAs
    const int i = lane & 31, hh = lane >> 5;
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = 2 * j + hh;
      const float v = dl[i][min(e, E - 1)];
      a[j] = e < E ? v : 0.f;
    }
    const int ntiles = (mb + 31) / 32;
    for (int tile = wave; tile < ntiles; tile += kRbThreads / 64) {
      const int col = tile * 32 + i;
      const float* wp = wg + c0 + min(col, mb - 1);
      float b[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) b[j] = wp[(size_t)min(2 * j + hh, E - 1) * M];  // a[j] = 0 masks e >= E
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
      if (col < mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) G[((r & 3) + 8 * (r >> 2) + 4 * hh) * ld + col] = acc[r];
      }
    }
  }
  __syncthreads();
  auto finish = [&](int id, int s1, int s2, const Vec16<T>& v1, const Vec16<T>& v2) {
    const int t = id / chunks, c = id - t * chunks;
    const int s = s0 + t;
    float* g = G + t * ld + c * N;
    Vec16<T> o;
    float ov[N];
#pragma unroll
    for (int j = 0; j < N; ++j) ov[j] = (g[j] + (s1 >= 0 ? v1.get(j) : 0.f)) + (s2 >= 0 ? v2.get(j) : 0.f);
    o.set_all(ov);
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = s < S ? o.get(j) : 0.f;
    if (s < S) store16<T>(dh + (size_t)s * M + c0 + c * N, o);
  };
  // phase 2
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int id = threadIdx.x + k * kRbThreads;
    if (id < items) finish(id, ps1[k], ps2[k], p1[k], p2[k]);
  }
#pragma unroll 2
  for (int id = threadIdx.x + PRE * kRbThreads; id < items; id += kRbThreads) {  // wide slabs: the rest
    const int t = id / chunks, c = id - t * chunks;
    const int sc = min(s0 + t, S - 1);
    const int s1 = slot[2 * sc], s2 = slot[2 * sc + 1];
    const Vec16<T> v1 = load16<T>(dxd + (size_t)max(s1, 0) * M + c0 + c * N);
    const Vec16<T> v2 = load16<T>(dxd + (size_t)max(s2, 0) * M + c0 + c * N);
    finish(id, s1, s2, v1, v2);
  }
  if (!part) return;  // block-uniform
  __syncthreads();
  const unsigned tm = text_mask;
  for (int col = threadIdx.x; col < mb; col += kRbThreads) {  // phase 3
    float av = 0.f, at = 0.f;
#pragma unroll 8
    for (int t = 0; t < kRbTok; ++t) {
      const float v = G[t * ld + col];
      if ((tm >> t) & 1u) at += v; else av += v;
    }
    part[((size_t)blockIdx.x * 2 + 0) * M + c0 + col] = av;
    part[((size_t)blockIdx.x * 2 + 1) * M + c0 + col] = at;
  }
}

// ---- the same, bf16 activations, without block-wide phases: a WAVE owns 32 tokens x 64 columns ------------------
// The block kernel above runs load -> barrier -> MFMA -> barrier -> add/store -> barrier -> column sums, three
// dependent memory round trips per block and two rounds of blocks: it is bound by that latency chain, not by HBM.
// Here a wave works alone on its item: its gathers (lanes consecutive ALONG the row: 8 lanes x 16 bytes = one
// 128-byte line per token and slot, 8 lines per instruction) are in flight from the start, the 32 x 64 product goes
// through a wave-private fp32 image in LDS into that row layout, and a wave that has a second item has its loads in
// flight while the first one's column sums run.  The product is formed transposed (MFMA rows = columns of wg,
// permuted so that a lane holds 16 consecutive columns of one token: 16-byte LDS writes), from wg fragments that
// the block splits into bf16 hi / lo halves once.  Column sums by modality: token by token, as the block kernel.
typedef __attribute__((ext_vector_type(8))) __bf16 rw_bf16x8;
constexpr int kRwCols = 64;   // columns of a wave's item = 2 MFMA tiles
constexpr int kRwLd = 68;     // floats per token row of a wave's image (+4: rows start in different banks)
// (bx, by, gx) = the block's position and the x extent of the kernel's (gx, column spans) grid: the block / grid
// indices of the kernel of its own, a block range of the launch that also carries the gate weight gradient's blocks.
// wfrag: uint4[8 * 64], fragment (tile q, k step ks, hi/lo) x lane; image: float[4][32 * kRwLd]; both the block's.
__device__ __forceinline__ void route_bwd_bf16_body(const bf16_t* __restrict__ dxd, const int32_t* __restrict__ slot,
                                                    const float* __restrict__ dlogits, const float* __restrict__ wg,
                                                    bf16_t* __restrict__ dh, float* __restrict__ part, int S, int T_,
                                                    int n_vis, int mode, int M, int E, int ngroups, int bx, int by, int gx,
                                                    uint4* wfrag, float (*image)[32 * kRwLd]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c0 = by * kRwCols;
  const int t = lane & 31, hh = lane >> 5;  // MFMA view: token t, accumulator half hh
  const int tq = lane >> 3, ch = lane & 7;  // row view: tokens 8 i + tq (i < 4), 16-byte chunk ch of the 64 columns
  const int colr = min(c0 + 8 * ch, M - 8);
  const bool col_ok = c0 + 8 * ch < M;
  int grp = bx * 4 + wave;
  const int gstride = gx * 4;
  int2 sl[4];
  Vec16<bf16_t> v1[4], v2[4];
  float4 dl4[4];
  auto load_item = [&]() {  // the item's global loads, all in flight at once
    const int s0 = grp * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = min(s0 + 8 * i + tq, S - 1);
      sl[i] = *reinterpret_cast<const int2*>(slot + 2 * s);
      v1[i] = load16<bf16_t>(dxd + (size_t)max(sl[i].x, 0) * M + colr);
      v2[i] = load16<bf16_t>(dxd + (size_t)max(sl[i].y, 0) * M + colr);
    }
    // dlogits row of token t, experts 8 hh .. + 7 and 16 + 8 hh .. + 7 (the B operand: k = expert)
    const float* dr = dlogits + (size_t)min(s0 + t, S - 1) * E;
    if (E == 32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dl4[k] = *reinterpret_cast<const float4*>(dr + (k >> 1) * 16 + 8 * hh + (k & 1) * 4);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = (k >> 1) * 16 + 8 * hh + (k & 1) * 4;
        dl4[k] = make_float4(e < E ? dr[e] : 0.f, e + 1 < E ? dr[e + 1] : 0.f, e + 2 < E ? dr[e + 2] : 0.f,
                             e + 3 < E ? dr[e + 3] : 0.f);
      }
    }
  };
  bool have = grp < ngroups;
  if (have) load_item();
  for (int p = threadIdx.x; p < 4 * 64; p += 256) {
    const int l = p & 63, qk = p >> 6, q = qk >> 1, ks = qk & 1;
    const int i = l & 31, kh = l >> 5;
    // MFMA row i of tile q is accumulator register r = 4 (i >> 3) + (i & 3) of the lanes with hh = (i >> 2) & 1,
    // and stands for column 32 q + 16 hh + r of the span
    const int col = c0 + 32 * q + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    rw_bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = ks * 16 + 8 * kh + j;
      const float v = (e < E && col < M) ? wg[(size_t)e * M + col] : 0.f;
      const __bf16 hb = static_cast<__bf16>(v);
      hi[j] = hb;
      lo[j] = static_cast<__bf16>(v - static_cast<float>(hb));
    }
    wfrag[(qk * 2 + 0) * 64 + l] = __builtin_bit_cast(uint4, hi);
    wfrag[(qk * 2 + 1) * 64 + l] = __builtin_bit_cast(uint4, lo);
  }
  __syncthreads();
  float* img = image[wave];
  while (have) {
    const int grp_c = grp, s0 = grp * 32;
    rw_bf16x8 dhi[2], dlo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 f = dl4[ks * 2 + (j >> 2)];
        const float v = (j & 3) == 0 ? f.x : (j & 3) == 1 ? f.y : (j & 3) == 2 ? f.z : f.w;
        const __bf16 hb = static_cast<__bf16>(v);
        dhi[ks][j] = hb;
        dlo[ks][j] = static_cast<__bf16>(v - static_cast<float>(hb));
      }
    // (only the column sums by modality read it; without them T_ may be 0 and token_modality would divide by it)
    const unsigned tm = part ? (unsigned)__ballot(lane < 32 && s0 + t < S && token_modality(min(s0 + t, S - 1), T_, n_vis, mode) != 0) : 0u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const rw_bf16x8 wh = __builtin_bit_cast(rw_bf16x8, wfrag[((q * 2 + ks) * 2 + 0) * 64 + lane]);
        const rw_bf16x8 wl = __builtin_bit_cast(rw_bf16x8, wfrag[((q * 2 + ks) * 2 + 1) * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, dlo[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, dhi[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, dhi[ks], acc, 0, 0, 0);
      }
      float* g = img + t * kRwLd + 32 * q + 16 * hh;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(g + 4 * k) = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tok = 8 * i + tq, s = s0 + tok;
      float* g = img + tok * kRwLd + 8 * ch;
      const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      if (sl[i].x < 0) v1[i] = zero16<bf16_t>();  // row 0 stood in for "no slot": one select per chunk, not per element
      if (sl[i].y < 0) v2[i] = zero16<bf16_t>();
      float ov[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = (gv[j] + v1[i].get(j)) + v2[i].get(j);
      Vec16<bf16_t> o;
      o.set_all(ov);
      const bool live = s < S;
      if (live && col_ok) store16<bf16_t>(dh + (size_t)s * M + colr, o);
      if (part) {  // the rounded values (zero for rows past the end) back into the image for the column sums
        *reinterpret_cast<float4*>(g) = live ? make_float4(o.get(0), o.get(1), o.get(2), o.get(3)) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(g + 4) = live ? make_float4(o.get(4), o.get(5), o.get(6), o.get(7)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // the next item's loads go out before this one's column sums
    grp += gstride;
    have = grp < ngroups;
    if (have) load_item();
    if (part) {  // lane = column: sums over the item's tokens by modality, token by token
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      float av = 0.f, at = 0.f;
      const unsigned livem = s0 + 32 <= S ? 0xffffffffu : ((1u << (S - s0)) - 1u);  // rows past the end hold zeros
      if (tm == 0u || tm == livem) {  // one modality in the item (most items): a plain sum, same order
        float a = 0.f;
#pragma unroll 8
        for (int tt = 0; tt < 32; ++tt) a += img[tt * kRwLd + lane];
        if (tm == 0u) av = a; else at = a;
      } else {
#pragma unroll 8
        for (int tt = 0; tt < 32; ++tt) {
          const float x = img[tt * kRwLd + lane];
          if ((tm >> tt) & 1u) at += x; else av += x;
        }
      }
      if (c0 + lane < M) {
        part[((size_t)grp_c * 2 + 0) * M + c0 + lane] = av;
        part[((size_t)grp_c * 2 + 1) * M + c0 + lane] = at;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is rewritten by the next item
    __builtin_amdgcn_wave_barrier();
  }
#endif
}

__global__ __launch_bounds__(256) void route_bwd_bf16_kernel(const bf16_t* __restrict__ dxd, const int32_t* __restrict__ slot,
                                                             const float* __restrict__ dlogits, const float* __restrict__ wg,
                                                             bf16_t* __restrict__ dh, float* __restrict__ part, int S, int T_,
                                                             int n_vis, int mode, int M, int E, int ngroups) {
  __shared__ uint4 wfrag[8 * 64];
  __shared__ __attribute__((aligned(16))) float image[4][32 * kRwLd];
  route_bwd_bf16_body(dxd, slot, dlogits, wg, dh, part, S, T_, n_vis, mode, M, E, ngroups, blockIdx.x, blockIdx.y,
                      gridDim.x, wfrag, image);
}

// The route backward and the gate weight gradient in ONE launch (the fused eve_moe_bwd): both only need dlogits, each
// on its own is bound by its latency chain (37 % / 20 % of the HBM rate, waves parked half of their time), so their
// blocks are dealt into one grid -- evenly interleaved, nw weight-gradient blocks among nr route blocks -- and fill
// each other's waits.  Same bodies, same arithmetic, same results as the two kernels of their own.
struct RouteWgradArgs {
  const bf16_t* dxd; const int32_t* slot; const float* dlogits; const float* wg; bf16_t* dh; float* part_embed;
  const bf16_t* h; const bf16_t* vemb; const bf16_t* temb; float* part_wg;
  int S, T, n_vis, mode, M, E, ngroups;
  int rbx, nr;    // route blocks: (rbx, column spans) flattened, nr of them
  int wbx, nw;    // weight-gradient blocks: (wbx token chunks, groups of 256 columns) flattened, nw of them
};
template <bool EMB>
__global__ __launch_bounds__(256) void route_wgrad_bf16_kernel(RouteWgradArgs a) {
  __shared__ uint4 wfrag[8 * 64];
  __shared__ __attribute__((aligned(16))) float image[4][32 * kRwLd];
  const unsigned b = blockIdx.x, total = (unsigned)(a.nr + a.nw);
  const unsigned w0 = (unsigned)((unsigned long long)b * a.nw / total), w1 = (unsigned)((unsigned long long)(b + 1) * a.nw / total);
  if (w1 > w0) {  // weight-gradient block number w0
    gate_wgrad_bf16_body<EMB>(a.h, a.vemb, a.temb, a.dlogits, a.part_wg, a.S, a.T, a.n_vis, a.mode, a.M, a.E, kWgradBf16TS,
                              (int)(w0 % a.wbx), (int)(w0 / a.wbx));
  } else {        // route block number b - w0
    const unsigned r = b - w0;
    route_bwd_bf16_body(a.dxd, a.slot, a.dlogits, a.wg, a.dh, a.part_embed, a.S, a.T, a.n_vis, a.mode, a.M, a.E, a.ngroups,
                        (int)(r % a.rbx), (int)(r / a.rbx), a.rbx, wfrag, image);
  }
}

// fixed-order sum of the per-block partials: block bx = 32 columns x 32 slices of the partial rows (`red`: float[32][33])
template <typename TO>
__device__ __forceinline__ void embed_grad_final32_body(const float* __restrict__ part, TO* __restrict__ dv,
                                                        TO* __restrict__ dt, int M, int nchunks, int bx, float (*red)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = bx * 32 + tx;   // over 2*M columns: [vision M | text M] per chunk row
  float acc = 0.f;
  if (col < 2 * M) {
#pragma unroll 4
    for (int b = ty; b < nchunks; b += 32) acc += part[(size_t)b * 2 * M + col];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < 2 * M) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][tx];
    if (col < M) { if (dv) dv[col] = from_f32<TO>(s); }
    else if (dt) dt[col - M] = from_f32<TO>(s);
  }
}

template <typename TO>
__global__ __launch_bounds__(1024) void embed_grad_final32_kernel(const float* __restrict__ part, TO* __restrict__ dv,
                                                                  TO* __restrict__ dt, int M, int nchunks) {
  __shared__ float red[32][33];
  embed_grad_final32_body<TO>(part, dv, dt, M, nchunks, blockIdx.x, red);
}

// Both closing reductions of the fused backward in one launch: blocks [0, n1) the modality-embedding gradients
// (embed_grad_final32), the rest the gate weight gradient's partial sums (sum_partials4: 64 outputs per block, on
// the block's first 256 threads).
template <typename TO>
__global__ __launch_bounds__(1024) void final_sums_kernel(const float* __restrict__ part_embed, TO* __restrict__ dv,
                                                          TO* __restrict__ dt, int M, int nchunks, int n1,
                                                          const float* __restrict__ part_wg, float* __restrict__ dwg,
                                                          int n_wg, int nparts_wg) {
  __shared__ float red[32][33];
  if ((int)blockIdx.x < n1) embed_grad_final32_body<TO>(part_embed, dv, dt, M, nchunks, blockIdx.x, red);
  else sum_partials4_body(part_wg, dwg, n_wg, nparts_wg, (int)blockIdx.x - n1, reinterpret_cast<float(*)[64]>(&red[0][0]));
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int row_grid(int rows) { return max(1, min(ceil_div(rows, 4), 4096)); }

static int check_rows(const char* who, int M, int dtype) {
  const int n = dtype == EVE_BF16 ? 8 : 4;
  if (dtype != EVE_F32 && dtype != EVE_BF16) { set_error("%s: bad dtype %d", who, dtype); return EVE_ERR_ARG; }
  if (M <= 0 || M % n != 0) { set_error("%s: M=%d must be a positive multiple of %d", who, M, n); return EVE_ERR_ARG; }
  return EVE_OK;
}

// passes of 32 lanes x 16 bytes that make up a row when the width has a compile-time kernel, else 0
static int row_passes(int M, int dtype) {
  const int per_pass = 32 * (dtype == EVE_BF16 ? 8 : 4);
  if (M % per_pass) return 0;
  const int np = M / per_pass;
  return (np >= 1 && np <= 4) || np == 6 || np == 8 ? np : 0;
}
static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
// two rows per wave, four waves per block, every wave resident at once up to 8 blocks per CU of a 256-CU part
static inline int pair_grid(int rows) { return max(1, min(ceil_div(rows, 8), 4096)); }

}  // namespace eve

using namespace eve;

// KERNEL<TT, NP> for the runtime pass count np
#define EVE_LAUNCH_NP(KERNEL, TT, np, grid, st, ...)                                              \
  do {                                                                                            \
    switch (np) {                                                                                 \
      case 1: hipLaunchKernelGGL((KERNEL<TT, 1>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
      case 2: hipLaunchKernelGGL((KERNEL<TT, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
      case 3: hipLaunchKernelGGL((KERNEL<TT, 3>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
      case 4: hipLaunchKernelGGL((KERNEL<TT, 4>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
      case 6: hipLaunchKernelGGL((KERNEL<TT, 6>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
      default: hipLaunchKernelGGL((KERNEL<TT, 8>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
    }                                                                                             \
  } while (0)

#define EVE_LAUNCH_NP_GB(KERNEL, TT, np, grid, st, ...)                                                \
  do {                                                                                                 \
    switch (np) {                                                                                      \
      case 1: hipLaunchKernelGGL((KERNEL<TT, 1, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
      case 2: hipLaunchKernelGGL((KERNEL<TT, 2, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
      case 3: hipLaunchKernelGGL((KERNEL<TT, 3, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<TT, 4, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
      case 6: hipLaunchKernelGGL((KERNEL<TT, 6, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
      default: hipLaunchKernelGGL((KERNEL<TT, 8, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
    }                                                                                                  \
  } while (0)

#define EVE_DISPATCH_DTYPE(dtype, KERNEL, grid, block, st, ...)                                     \
  do {                                                                                              \
    if ((dtype) == EVE_F32) hipLaunchKernelGGL((KERNEL<float>), grid, block, 0, st, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<bf16_t>), grid, block, 0, st, __VA_ARGS__);                     \
  } while (0)

extern "C" int eve_dispatch_fwd(const void* h, const void* vision_embed, const void* text_embed, const int32_t* slot,
                                const int32_t* slot_token, void* xd, int S, int T, int n_vis, int mode, int M,
                                int num_slots, int dtype, void* stream) {
  EVE_REQUIRE(h && slot_token && xd, EVE_ERR_ARG, "eve_dispatch_fwd: null pointer");
  EVE_REQUIRE(S > 0 && num_slots > 0, EVE_ERR_ARG, "eve_dispatch_fwd: bad shape");
  if (int rc = check_rows("eve_dispatch_fwd", M, dtype)) return rc;
  if (mode != EVE_MODE_NONE)
    EVE_REQUIRE(vision_embed && text_embed && T > 0 && S % T == 0, EVE_ERR_ARG, "eve_dispatch_fwd: embeddings / T required for mode %d", mode);
  hipStream_t st = S_(stream);
  const int np = row_passes(M, dtype);
  if (slot && np && aligned_to(slot, 8) && aligned_to(h, 16) && aligned_to(xd, 16) &&
      (mode == EVE_MODE_NONE || (aligned_to(vision_embed, 16) && aligned_to(text_embed, 16)))) {
    // token-major: each token row is read once for both of its slots
    if (dtype == EVE_F32)
      EVE_LAUNCH_NP(dispatch_fwd_tok_kernel, float, np, dim3(pair_grid(S)), st, (const float*)h, (const float*)vision_embed,
                    (const float*)text_embed, slot, slot_token, (float*)xd, S, T, n_vis, mode, num_slots);
    else
      EVE_LAUNCH_NP(dispatch_fwd_tok_kernel, bf16_t, np, dim3(pair_grid(S)), st, (const bf16_t*)h, (const bf16_t*)vision_embed,
                    (const bf16_t*)text_embed, slot, slot_token, (bf16_t*)xd, S, T, n_vis, mode, num_slots);
    EVE_CHECK_LAUNCH("dispatch_fwd_tok_kernel");
    return EVE_OK;
  }
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((dispatch_fwd_kernel<float>), dim3(row_grid(num_slots)), dim3(256), 0, st, (const float*)h,
                       (const float*)vision_embed, (const float*)text_embed, slot_token, (float*)xd, T, n_vis, mode, M,
                       num_slots);
  else
    hipLaunchKernelGGL((dispatch_fwd_kernel<bf16_t>), dim3(row_grid(num_slots)), dim3(256), 0, st, (const bf16_t*)h,
                       (const bf16_t*)vision_embed, (const bf16_t*)text_embed, slot_token, (bf16_t*)xd, T, n_vis, mode,
                       M, num_slots);
  EVE_CHECK_LAUNCH("dispatch_fwd_kernel");
  return EVE_OK;
}

extern "C" int eve_dispatch_bwd(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh,
                                int S, int M, int E, int dtype, void* stream) {
  EVE_REQUIRE(dxd && slot && dh, EVE_ERR_ARG, "eve_dispatch_bwd: null pointer");
  EVE_REQUIRE(!dlogits || wg, EVE_ERR_ARG, "eve_dispatch_bwd: wg required with dlogits");
  EVE_REQUIRE(S > 0 && E > 0, EVE_ERR_ARG, "eve_dispatch_bwd: bad shape");
  if (int rc = check_rows("eve_dispatch_bwd", M, dtype)) return rc;
  hipStream_t st = S_(stream);
  const dim3 grid(ceil_div(S, kDbTok));
  const size_t lds = (size_t)E * kDbTok * sizeof(float);
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((dispatch_bwd_kernel<float>), grid, dim3(128), lds, st, (const float*)dxd, slot, dlogits, wg,
                       (float*)dh, S, M, E);
  else
    hipLaunchKernelGGL((dispatch_bwd_kernel<bf16_t>), grid, dim3(128), lds, st, (const bf16_t*)dxd, slot, dlogits, wg,
                       (bf16_t*)dh, S, M, E);
  EVE_CHECK_LAUNCH("dispatch_bwd_kernel");
  return EVE_OK;
}

extern "C" int eve_combine_fwd(const void* y, const int32_t* slot, const float* weight, void* out, int S, int M,
                               int dtype, void* stream) {
  EVE_REQUIRE(y && slot && weight && out, EVE_ERR_ARG, "eve_combine_fwd: null pointer");
  EVE_REQUIRE(S > 0, EVE_ERR_ARG, "eve_combine_fwd: bad shape");
  if (int rc = check_rows("eve_combine_fwd", M, dtype)) return rc;
  hipStream_t st = S_(stream);
  const int np = row_passes(M, dtype);
  if (np && aligned_to(slot, 8) && aligned_to(weight, 8) && aligned_to(y, 16) && aligned_to(out, 16)) {
    if (dtype == EVE_F32)
      EVE_LAUNCH_NP(combine_fwd_np_kernel, float, np, dim3(pair_grid(S)), st, (const float*)y, slot, weight, (float*)out, S);
    else
      EVE_LAUNCH_NP(combine_fwd_np_kernel, bf16_t, np, dim3(pair_grid(S)), st, (const bf16_t*)y, slot, weight, (bf16_t*)out, S);
    EVE_CHECK_LAUNCH("combine_fwd_np_kernel");
    return EVE_OK;
  }
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((combine_fwd_kernel<float>), dim3(row_grid(S)), dim3(256), 0, st, (const float*)y, slot, weight,
                       (float*)out, S, M);
  else
    hipLaunchKernelGGL((combine_fwd_kernel<bf16_t>), dim3(row_grid(S)), dim3(256), 0, st, (const bf16_t*)y, slot,
                       weight, (bf16_t*)out, S, M);
  EVE_CHECK_LAUNCH("combine_fwd_kernel");
  return EVE_OK;
}

extern "C" int eve_combine_bwd(const void* dout, const void* y, const int32_t* slot, const int32_t* slot_token,
                               const float* slot_weight, void* dy, float* d_weight, int S, int M, int num_slots,
                               int dtype, void* stream) {
  EVE_REQUIRE(dout && y && slot && slot_token && slot_weight && dy && d_weight, EVE_ERR_ARG, "eve_combine_bwd: null pointer");
  EVE_REQUIRE(S > 0 && num_slots > 0, EVE_ERR_ARG, "eve_combine_bwd: bad shape");
  if (int rc = check_rows("eve_combine_bwd", M, dtype)) return rc;
  hipStream_t st = S_(stream);
  const int np = row_passes(M, dtype);
  if (np && aligned_to(slot, 8) && aligned_to(dout, 16) && aligned_to(y, 16) && aligned_to(dy, 16)) {
    if (dtype == EVE_F32)
      EVE_LAUNCH_NP(combine_bwd_np_kernel, float, np, dim3(pair_grid(S)), st, (const float*)dout, (const float*)y, slot,
                    slot_token, slot_weight, (float*)dy, d_weight, S, num_slots, GateBwdArgs{});
    else
      EVE_LAUNCH_NP(combine_bwd_np_kernel, bf16_t, np, dim3(pair_grid(S)), st, (const bf16_t*)dout, (const bf16_t*)y, slot,
                    slot_token, slot_weight, (bf16_t*)dy, d_weight, S, num_slots, GateBwdArgs{});
    EVE_CHECK_LAUNCH("combine_bwd_np_kernel");
    return EVE_OK;
  }
  const int grid = max(1, min(ceil_div(S, 8), 8192));
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((combine_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)dout, (const float*)y, slot,
                       slot_token, slot_weight, (float*)dy, d_weight, S, M, num_slots);
  else
    hipLaunchKernelGGL((combine_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)y,
                       slot, slot_token, slot_weight, (bf16_t*)dy, d_weight, S, M, num_slots);
  EVE_CHECK_LAUNCH("combine_bwd_kernel");
  return EVE_OK;
}

// eve_combine_bwd + eve_gate_top2_bwd as ONE launch (the fused backward, fused.hip): the half wave that forms a
// token's d weight goes on to its d logits, lane = expert (gate_bwd32.h: the function gate_bwd32_kernel runs, so the
// result is bit-identical to the two calls).  The gate's backward depends on nothing but d weight; as a launch of its
// own it sat between the expert GEMMs and the route backward, 5 us of latency for 2 MB.  `combine_gate_bwd_fusable`:
// the shapes that take the row-mover kernel and the 32-expert gate kernel.
namespace eve {
bool combine_gate_bwd_fusable(const void* dout, const void* y, const int32_t* slot, const void* dy, const int32_t* idx,
                              const int32_t* loc, int M, int E, int dtype) {
  return E <= 32 && row_passes(M, dtype) != 0 && aligned_to(slot, 8) && aligned_to(dout, 16) && aligned_to(y, 16) &&
         aligned_to(dy, 16) && aligned_to(idx, 8) && aligned_to(loc, 8);
}

int combine_bwd_gate_bwd(const void* dout, const void* y, const int32_t* slot, const int32_t* slot_token, const float* slot_weight,
                         void* dy, float* d_weight, const float* gates, const int32_t* idx, const int32_t* loc,
                         const float* d_l_aux, const int64_t* exp_counts, float* dlogits, int S, int M, int E, int C,
                         int num_slots, int dtype, void* stream) {
  EVE_REQUIRE(dout && y && slot && slot_token && slot_weight && dy && d_weight && gates && idx && loc && exp_counts && dlogits,
              EVE_ERR_ARG, "eve_moe_bwd (combine + gate backward): null pointer");
  EVE_REQUIRE(S > 0 && num_slots > 0 && E > 0 && E <= 32 && C > 0, EVE_ERR_ARG, "eve_moe_bwd (combine + gate backward): bad shape");
  const int np = row_passes(M, dtype);
  EVE_REQUIRE(np != 0, EVE_ERR_UNSUPPORTED, "eve_moe_bwd (combine + gate backward): width %d", M);
  hipStream_t st = S_(stream);
  const GateBwdArgs gb{gates, idx, loc, d_l_aux, exp_counts, dlogits, E, C};
  if (dtype == EVE_F32)
    EVE_LAUNCH_NP_GB(combine_bwd_np_kernel, float, np, dim3(pair_grid(S)), st, (const float*)dout, (const float*)y, slot,
                     slot_token, slot_weight, (float*)dy, d_weight, S, num_slots, gb);
  else
    EVE_LAUNCH_NP_GB(combine_bwd_np_kernel, bf16_t, np, dim3(pair_grid(S)), st, (const bf16_t*)dout, (const bf16_t*)y, slot,
                     slot_token, slot_weight, (bf16_t*)dy, d_weight, S, num_slots, gb);
  EVE_CHECK_LAUNCH("combine_bwd_np_kernel (with the gate backward)");
  return EVE_OK;
}
}  // namespace eve

static const int kEmbedTS = 32;

extern "C" size_t eve_embed_grad_workspace_bytes(int S, int M) {
  if (S <= 0 || M <= 0) return 0;
  return (size_t)ceil_div(S, kEmbedTS) * 2 * M * sizeof(float);
}

extern "C" int eve_embed_grad(const void* dh, float* dvision, float* dtext, int S, int T, int n_vis, int mode, int M,
                              int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(dh && workspace, EVE_ERR_ARG, "eve_embed_grad: null pointer");
  EVE_REQUIRE(mode != EVE_MODE_NONE, EVE_ERR_ARG, "eve_embed_grad: mode NONE has no embeddings");
  EVE_REQUIRE(S > 0 && T > 0 && S % T == 0, EVE_ERR_ARG, "eve_embed_grad: bad shape");
  if (int rc = check_rows("eve_embed_grad", M, dtype)) return rc;
  const size_t need = eve_embed_grad_workspace_bytes(S, M);
  EVE_REQUIRE(workspace_bytes >= need, EVE_ERR_WORKSPACE, "eve_embed_grad: workspace %zu < %zu", workspace_bytes, need);
  const int nchunks = ceil_div(S, kEmbedTS);
  float* part = static_cast<float*>(workspace);
  hipStream_t st = S_(stream);
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((embed_grad_partial_kernel<float>), dim3(nchunks), dim3(128), 0, st, (const float*)dh, part, S,
                       T, n_vis, mode, M, kEmbedTS);
  else
    hipLaunchKernelGGL((embed_grad_partial_kernel<bf16_t>), dim3(nchunks), dim3(128), 0, st, (const bf16_t*)dh, part,
                       S, T, n_vis, mode, M, kEmbedTS);
  EVE_CHECK_LAUNCH("embed_grad_partial_kernel");
  hipLaunchKernelGGL(embed_grad_final_kernel<float>, dim3(ceil_div(2 * M, 32)), dim3(256), 0, st, part, dvision, dtext, M,
                     nchunks);
  EVE_CHECK_LAUNCH("embed_grad_final_kernel");
  return EVE_OK;
}

// column slab per block: the whole row up to 256 columns, else about half of it (a multiple of 32)
static inline int route_bwd_slab(int M) { return M <= 256 ? ceil_div(M, 32) * 32 : ceil_div(ceil_div(M, 2), 32) * 32; }
static inline size_t route_bwd_lds(int M) { return (size_t)kRbTok * (route_bwd_slab(M) + 4) * sizeof(float); }

extern "C" int eve_route_bwd(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh,
                             void* dvision, void* dtext, int S, int T, int n_vis, int mode, int M, int E, int dtype,
                             int embed_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(dxd && slot && dlogits && wg && dh, EVE_ERR_ARG, "eve_route_bwd: null pointer");
  EVE_REQUIRE(embed_dtype == EVE_F32 || embed_dtype == EVE_BF16, EVE_ERR_ARG, "eve_route_bwd: bad embed_dtype %d", embed_dtype);
  EVE_REQUIRE(S > 0 && E > 0, EVE_ERR_ARG, "eve_route_bwd: bad shape");
  if (int rc = check_rows("eve_route_bwd", M, dtype)) return rc;
  const bool want_embed = mode != EVE_MODE_NONE && (dvision || dtext);
  if (want_embed) {
    EVE_REQUIRE(T > 0 && S % T == 0, EVE_ERR_ARG, "eve_route_bwd: S=%d not a multiple of T=%d", S, T);
    const size_t need = eve_embed_grad_workspace_bytes(S, M);
    EVE_REQUIRE(workspace && workspace_bytes >= need, EVE_ERR_WORKSPACE, "eve_route_bwd: workspace %zu < %zu",
                workspace_bytes, need);
  }
  if (E > 32 || route_bwd_lds(M) > 160 * 1024) {  // shapes the fused kernel does not cover: the two-pass path
    if (int rc = eve_dispatch_bwd(dxd, slot, dlogits, wg, dh, S, M, E, dtype, stream)) return rc;
    if (!want_embed) return EVE_OK;
    if (embed_dtype == EVE_F32)
      return eve_embed_grad(dh, (float*)dvision, (float*)dtext, S, T, n_vis, mode, M, dtype, workspace, workspace_bytes, stream);
    // bf16 embedding gradients: the same two stages, the final sums rounded once
    const int nchunks = ceil_div(S, kEmbedTS);
    float* part = static_cast<float*>(workspace);
    if (dtype == EVE_F32)
      hipLaunchKernelGGL((embed_grad_partial_kernel<float>), dim3(nchunks), dim3(128), 0, S_(stream), (const float*)dh,
                         part, S, T, n_vis, mode, M, kEmbedTS);
    else
      hipLaunchKernelGGL((embed_grad_partial_kernel<bf16_t>), dim3(nchunks), dim3(128), 0, S_(stream),
                         (const bf16_t*)dh, part, S, T, n_vis, mode, M, kEmbedTS);
    EVE_CHECK_LAUNCH("embed_grad_partial_kernel");
    hipLaunchKernelGGL(embed_grad_final_kernel<bf16_t>, dim3(ceil_div(2 * M, 32)), dim3(256), 0, S_(stream), part,
                       (bf16_t*)dvision, (bf16_t*)dtext, M, nchunks);
    EVE_CHECK_LAUNCH("embed_grad_final_kernel");
    return EVE_OK;
  }
  static_assert(kRbTok == kEmbedTS, "eve_embed_grad_workspace_bytes sizes the partials of both paths");
  hipStream_t st = S_(stream);
  const int nblk = ceil_div(S, kRbTok);
  if (dtype == EVE_BF16 && aligned_to(slot, 8) && aligned_to(dxd, 16) && aligned_to(dh, 16)) {
    float* part = want_embed ? static_cast<float*>(workspace) : nullptr;
    // one round of blocks: 4 waves per block, 3 blocks per CU (LDS), 256 CUs; a wave walks its items
    const int spans = ceil_div(M, kRwCols);
    const int bx = max(1, min(ceil_div(nblk, 4), max(1, 768 / spans)));
    hipLaunchKernelGGL(route_bwd_bf16_kernel, dim3(bx, spans), dim3(256), 0, st, (const bf16_t*)dxd, slot, dlogits, wg,
                       (bf16_t*)dh, part, S, T, n_vis, mode, M, E, nblk);
    EVE_CHECK_LAUNCH("route_bwd_bf16_kernel");
    if (want_embed) {
      if (embed_dtype == EVE_F32)
        hipLaunchKernelGGL(embed_grad_final32_kernel<float>, dim3(ceil_div(2 * M, 32)), dim3(1024), 0, st, part,
                           (float*)dvision, (float*)dtext, M, nblk);
      else
        hipLaunchKernelGGL(embed_grad_final32_kernel<bf16_t>, dim3(ceil_div(2 * M, 32)), dim3(1024), 0, st, part,
                           (bf16_t*)dvision, (bf16_t*)dtext, M, nblk);
      EVE_CHECK_LAUNCH("embed_grad_final32_kernel");
    }
    return EVE_OK;
  }
  const size_t lds = route_bwd_lds(M);
  const int Mb = route_bwd_slab(M);
  float* part = want_embed ? static_cast<float*>(workspace) : nullptr;
#define EVE_RB_LAUNCH(TT)                                                                                          \
  do {                                                                                                             \
    auto kern = route_bwd_kernel<TT>;                                                                              \
    if (lds > 64 * 1024) { /* a per-device attribute: set on every launch (cheap next to the kernel) */            \
      EVE_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds) == hipSuccess, EVE_ERR_LAUNCH, "eve_route_bwd: cannot reserve %zu bytes of LDS", lds); \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, dim3(nblk, ceil_div(M, Mb)), dim3(kRbThreads), lds, st, (const TT*)dxd, slot, dlogits,  \
                       wg, (TT*)dh, part, S, T, n_vis, mode, M, E, Mb);                                            \
  } while (0)
  if (dtype == EVE_F32) EVE_RB_LAUNCH(float); else EVE_RB_LAUNCH(bf16_t);
#undef EVE_RB_LAUNCH
  EVE_CHECK_LAUNCH("route_bwd_kernel");
  if (want_embed) {
    if (embed_dtype == EVE_F32)
      hipLaunchKernelGGL(embed_grad_final32_kernel<float>, dim3(ceil_div(2 * M, 32)), dim3(1024), 0, st, part,
                         (float*)dvision, (float*)dtext, M, nblk);
    else
      hipLaunchKernelGGL(embed_grad_final32_kernel<bf16_t>, dim3(ceil_div(2 * M, 32)), dim3(1024), 0, st, part,
                         (bf16_t*)dvision, (bf16_t*)dtext, M, nblk);
    EVE_CHECK_LAUNCH("embed_grad_final32_kernel");
  }
  return EVE_OK;
}

namespace eve {

// The fused backward's closing stage: eve_route_bwd + eve_gate_wgrad as TWO launches instead of four (see
// route_wgrad_bf16_kernel / final_sums_kernel).  `route_wgrad_fusable` says whether the shape takes this path (the
// same conditions under which the two entry points pick the kernels it combines); results are bit-identical to the
// two calls.
bool route_wgrad_fusable(const void* dxd, const int32_t* slot, const void* dh, const void* h, int M, int E, int dtype) {
  return dtype == EVE_BF16 && E <= 32 && M % 8 == 0 && route_bwd_lds(M) <= 160 * 1024 && aligned_to(slot, 8) &&
         aligned_to(dxd, 16) && aligned_to(dh, 16) && aligned_to(h, 4);
}

int route_bwd_gate_wgrad(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh, void* dvision,
                         void* dtext, const void* h, const void* vision_embed, const void* text_embed, float* dwg, int S, int T,
                         int n_vis, int mode, int M, int E, int embed_dtype, void* ws_embed, size_t ws_embed_bytes,
                         void* ws_wgrad, size_t ws_wgrad_bytes, void* stream) {
  EVE_REQUIRE(dxd && slot && dlogits && wg && dh && h && dwg && ws_wgrad, EVE_ERR_ARG, "eve_moe_bwd (route + gate wgrad): null pointer");
  EVE_REQUIRE(embed_dtype == EVE_F32 || embed_dtype == EVE_BF16, EVE_ERR_ARG, "eve_moe_bwd: bad embed_dtype %d", embed_dtype);
  EVE_REQUIRE(S > 0 && E > 0 && E <= 32 && M % 8 == 0, EVE_ERR_ARG, "eve_moe_bwd (route + gate wgrad): bad shape");
  if (mode != EVE_MODE_NONE)
    EVE_REQUIRE(vision_embed && text_embed && T > 0, EVE_ERR_ARG, "eve_moe_bwd (route + gate wgrad): embeddings required");
  const bool want_embed = mode != EVE_MODE_NONE && (dvision || dtext);
  if (want_embed) {
    EVE_REQUIRE(S % T == 0, EVE_ERR_ARG, "eve_moe_bwd: S=%d not a multiple of T=%d", S, T);
    const size_t need = eve_embed_grad_workspace_bytes(S, M);
    EVE_REQUIRE(ws_embed && ws_embed_bytes >= need, EVE_ERR_WORKSPACE, "eve_moe_bwd: embed workspace %zu < %zu", ws_embed_bytes, need);
  }
  const size_t need_wg = eve_gate_wgrad_workspace_bytes(S, M, E);
  EVE_REQUIRE(ws_wgrad_bytes >= need_wg, EVE_ERR_WORKSPACE, "eve_moe_bwd: gate wgrad workspace %zu < %zu", ws_wgrad_bytes, need_wg);
  hipStream_t st = S_(stream);
  RouteWgradArgs a;
  a.dxd = (const bf16_t*)dxd; a.slot = slot; a.dlogits = dlogits; a.wg = wg; a.dh = (bf16_t*)dh;
  a.part_embed = want_embed ? static_cast<float*>(ws_embed) : nullptr;
  a.h = (const bf16_t*)h; a.vemb = (const bf16_t*)vision_embed; a.temb = (const bf16_t*)text_embed;
  a.part_wg = static_cast<float*>(ws_wgrad);
  a.S = S; a.T = T; a.n_vis = n_vis; a.mode = mode; a.M = M; a.E = E;
  a.ngroups = ceil_div(S, kRbTok);
  const int spans = ceil_div(M, kRwCols);
  a.rbx = max(1, min(ceil_div(a.ngroups, 4), max(1, 768 / spans)));  // as eve_route_bwd
  a.nr = a.rbx * spans;
  a.wbx = ceil_div(S, kWgradBf16TS);                                  // as eve_gate_wgrad
  a.nw = a.wbx * ceil_div(M, 256);
  if (mode != EVE_MODE_NONE) hipLaunchKernelGGL(route_wgrad_bf16_kernel<true>, dim3(a.nr + a.nw), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(route_wgrad_bf16_kernel<false>, dim3(a.nr + a.nw), dim3(256), 0, st, a);
  EVE_CHECK_LAUNCH("route_wgrad_bf16_kernel");
  const int n1 = want_embed ? ceil_div(2 * M, 32) : 0, n2 = ceil_div(E * M, 64);
  if (embed_dtype == EVE_F32)
    hipLaunchKernelGGL(final_sums_kernel<float>, dim3(n1 + n2), dim3(1024), 0, st, a.part_embed, (float*)dvision, (float*)dtext, M,
                       a.ngroups, n1, a.part_wg, dwg, E * M, a.wbx);
  else
    hipLaunchKernelGGL(final_sums_kernel<bf16_t>, dim3(n1 + n2), dim3(1024), 0, st, a.part_embed, (bf16_t*)dvision,
                       (bf16_t*)dtext, M, a.ngroups, n1, a.part_wg, dwg, E * M, a.wbx);
  EVE_CHECK_LAUNCH("final_sums_kernel");
  return EVE_OK;
}

}  // namespace eve
