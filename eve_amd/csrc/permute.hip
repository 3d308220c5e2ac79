// Dispatch / combine permutes: the coalesced row gathers that replace the reference's dense
// one-hot einsums (models/sharded_moe.py:367 and :377) and their autograd backward, plus the
// modality-embedding gradient.  HBM-bound row movers: one wave per row, 16 B per lane.
#include "common.h"

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
#pragma unroll
        for (int j = 0; j < N; ++j) v.set(j, v.get(j) + e.get(j));
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
#pragma unroll
      for (int j = 0; j < N; ++j) o.set(j, acc[t][j]);
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
#pragma unroll
      for (int j = 0; j < N; ++j) o.set(j, acc[j]);
      store16<T>(out + (size_t)s * M + c * N, o);
    }
  }
}

// ------------------------------------------------------------------------------ combine bwd
// (a) per slot: dy[slot] = slot_weight * dout[tok] (0 when empty);
template <typename T>
__global__ __launch_bounds__(256) void combine_bwd_dy_kernel(const T* __restrict__ dout,
                                                             const int32_t* __restrict__ slot_token,
                                                             const float* __restrict__ slot_weight,
                                                             T* __restrict__ dy, int M, int num_slots) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int sl = wave; sl < num_slots; sl += nwaves) {
    const int tok = slot_token[sl];
    T* dst = dy + (size_t)sl * M;
    if (tok < 0) {
      for (int c = lane; c < chunks; c += 64) store16<T>(dst + c * N, zero16<T>());
      continue;
    }
    const float w = slot_weight[sl];
    const T* src = dout + (size_t)tok * M;
    for (int c = lane; c < chunks; c += 64) {
      Vec16<T> v = load16<T>(src + c * N);
#pragma unroll
      for (int j = 0; j < N; ++j) v.set(j, w * v.get(j));
      store16<T>(dst + c * N, v);
    }
  }
}

// (b) per token: d_weight[s,k] = <dout[s], y[slot_k]>.
template <typename T>
__global__ __launch_bounds__(256) void combine_bwd_dw_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                             const int32_t* __restrict__ slot,
                                                             float* __restrict__ d_weight, int S, int M) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int s = wave; s < S; s += nwaves) {
    const int s1 = slot[2 * s], s2 = slot[2 * s + 1];
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < chunks; c += 64) {
      const Vec16<T> d = load16<T>(dout + (size_t)s * M + c * N);
      if (s1 >= 0) {
        const Vec16<T> v = load16<T>(y + (size_t)s1 * M + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) a1 = fmaf(d.get(j), v.get(j), a1);
      }
      if (s2 >= 0) {
        const Vec16<T> v = load16<T>(y + (size_t)s2 * M + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) a2 = fmaf(d.get(j), v.get(j), a2);
      }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
      d_weight[2 * s] = a1;
      d_weight[2 * s + 1] = a2;
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

__global__ __launch_bounds__(256) void embed_grad_final_kernel(const float* __restrict__ part, float* __restrict__ dv,
                                                               float* __restrict__ dt, int M, int nchunks) {
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
    if (col < M) { if (dv) dv[col] = s; }
    else if (dt) dt[col - M] = s;
  }
}

// ------------------------------------------------------- fused routing backward (E <= 32)
// dh[s] = dxd[slot1] + dxd[slot2] + dlogits[s,:] @ wg, plus the per-block column sums of the ROUNDED dh
// by modality (the modality-embedding gradients), in one pass over dh.  A block owns 32 tokens:
//   1. G[32 tokens x M] = dlogits @ wg on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 FMAs),
//      each wave every fourth 32-column tile, written to LDS;
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
  {  // phase 1
    const int i = lane & 31, hh = lane >> 5;
    const int tok = min(s0 + i, S - 1);
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = 2 * j + hh;
      const float v = dlogits[(size_t)tok * E + min(e, E - 1)];
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
  {  // phase 2: loads are unconditional (clamped) so that several items' gathers are in flight together
    const int chunks = mb / N;
    const int items = kRbTok * chunks;
#pragma unroll 4
    for (int id = threadIdx.x; id < items; id += kRbThreads) {
      const int t = id / chunks, c = id - t * chunks;
      const int s = s0 + t, sc = min(s, S - 1);
      const int s1 = slot[2 * sc], s2 = slot[2 * sc + 1];
      const Vec16<T> v1 = load16<T>(dxd + (size_t)max(s1, 0) * M + c0 + c * N);
      const Vec16<T> v2 = load16<T>(dxd + (size_t)max(s2, 0) * M + c0 + c * N);
      float* g = G + t * ld + c * N;
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float v = (g[j] + (s1 >= 0 ? v1.get(j) : 0.f)) + (s2 >= 0 ? v2.get(j) : 0.f);
        o.set(j, v);
      }
#pragma unroll
      for (int j = 0; j < N; ++j) g[j] = s < S ? o.get(j) : 0.f;
      if (s < S) store16<T>(dh + (size_t)s * M + c0 + c * N, o);
    }
  }
  if (!part) return;  // block-uniform
  __syncthreads();
  for (int col = threadIdx.x; col < mb; col += kRbThreads) {  // phase 3
    float av = 0.f, at = 0.f;
    for (int t = 0; t < kRbTok; ++t) {
      const float v = G[t * ld + col];
      if (token_modality(min(s0 + t, S - 1), T_, n_vis, mode)) at += v; else av += v;
    }
    part[((size_t)blockIdx.x * 2 + 0) * M + c0 + col] = av;
    part[((size_t)blockIdx.x * 2 + 1) * M + c0 + col] = at;
  }
}

// fixed-order sum of the per-block partials: block = 32 columns x 32 slices of the partial rows
__global__ __launch_bounds__(1024) void embed_grad_final32_kernel(const float* __restrict__ part, float* __restrict__ dv,
                                                                  float* __restrict__ dt, int M, int nchunks) {
  __shared__ float red[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;   // over 2*M columns: [vision M | text M] per chunk row
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
    if (col < M) { if (dv) dv[col] = s; }
    else if (dt) dt[col - M] = s;
  }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int row_grid(int rows) { return max(1, min(ceil_div(rows, 4), 4096)); }

static int check_rows(const char* who, int M, int dtype) {
  const int n = dtype == EVE_BF16 ? 8 : 4;
  if (dtype != EVE_F32 && dtype != EVE_BF16) { set_error("%s: bad dtype %d", who, dtype); return EVE_ERR_ARG; }
  if (M <= 0 || M % n != 0) { set_error("%s: M=%d must be a positive multiple of %d", who, M, n); return EVE_ERR_ARG; }
  return EVE_OK;
}

}  // namespace eve

using namespace eve;

#define EVE_DISPATCH_DTYPE(dtype, KERNEL, grid, block, st, ...)                                     \
  do {                                                                                              \
    if ((dtype) == EVE_F32) hipLaunchKernelGGL((KERNEL<float>), grid, block, 0, st, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<bf16_t>), grid, block, 0, st, __VA_ARGS__);                     \
  } while (0)

extern "C" int eve_dispatch_fwd(const void* h, const void* vision_embed, const void* text_embed,
                                const int32_t* slot_token, void* xd, int S, int T, int n_vis, int mode, int M,
                                int num_slots, int dtype, void* stream) {
  EVE_REQUIRE(h && slot_token && xd, EVE_ERR_ARG, "eve_dispatch_fwd: null pointer");
  EVE_REQUIRE(S > 0 && num_slots > 0, EVE_ERR_ARG, "eve_dispatch_fwd: bad shape");
  if (int rc = check_rows("eve_dispatch_fwd", M, dtype)) return rc;
  if (mode != EVE_MODE_NONE)
    EVE_REQUIRE(vision_embed && text_embed && T > 0 && S % T == 0, EVE_ERR_ARG, "eve_dispatch_fwd: embeddings / T required for mode %d", mode);
  hipStream_t st = S_(stream);
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
  if (dtype == EVE_F32) {
    hipLaunchKernelGGL((combine_bwd_dy_kernel<float>), dim3(row_grid(num_slots)), dim3(256), 0, st, (const float*)dout,
                       slot_token, slot_weight, (float*)dy, M, num_slots);
    hipLaunchKernelGGL((combine_bwd_dw_kernel<float>), dim3(row_grid(S)), dim3(256), 0, st, (const float*)dout,
                       (const float*)y, slot, d_weight, S, M);
  } else {
    hipLaunchKernelGGL((combine_bwd_dy_kernel<bf16_t>), dim3(row_grid(num_slots)), dim3(256), 0, st,
                       (const bf16_t*)dout, slot_token, slot_weight, (bf16_t*)dy, M, num_slots);
    hipLaunchKernelGGL((combine_bwd_dw_kernel<bf16_t>), dim3(row_grid(S)), dim3(256), 0, st, (const bf16_t*)dout,
                       (const bf16_t*)y, slot, d_weight, S, M);
  }
  EVE_CHECK_LAUNCH("combine_bwd kernels");
  return EVE_OK;
}

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
  hipLaunchKernelGGL(embed_grad_final_kernel, dim3(ceil_div(2 * M, 32)), dim3(256), 0, st, part, dvision, dtext, M,
                     nchunks);
  EVE_CHECK_LAUNCH("embed_grad_final_kernel");
  return EVE_OK;
}

// column slab per block: the whole row up to 256 columns, else about half of it (a multiple of 32)
static inline int route_bwd_slab(int M) { return M <= 256 ? ceil_div(M, 32) * 32 : ceil_div(ceil_div(M, 2), 32) * 32; }
static inline size_t route_bwd_lds(int M) { return (size_t)kRbTok * (route_bwd_slab(M) + 4) * sizeof(float); }

extern "C" int eve_route_bwd(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh,
                             float* dvision, float* dtext, int S, int T, int n_vis, int mode, int M, int E, int dtype,
                             void* workspace, size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(dxd && slot && dlogits && wg && dh, EVE_ERR_ARG, "eve_route_bwd: null pointer");
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
    if (want_embed) return eve_embed_grad(dh, dvision, dtext, S, T, n_vis, mode, M, dtype, workspace, workspace_bytes, stream);
    return EVE_OK;
  }
  static_assert(kRbTok == kEmbedTS, "eve_embed_grad_workspace_bytes sizes the partials of both paths");
  hipStream_t st = S_(stream);
  const int nblk = ceil_div(S, kRbTok);
  const size_t lds = route_bwd_lds(M);
  const int Mb = route_bwd_slab(M);
  float* part = want_embed ? static_cast<float*>(workspace) : nullptr;
#define EVE_RB_LAUNCH(TT)                                                                                          \
  do {                                                                                                             \
    auto kern = route_bwd_kernel<TT>;                                                                              \
    static size_t attr_bytes = 0;                                                                                  \
    if (lds > 64 * 1024 && lds > attr_bytes) {                                                                     \
      EVE_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds) == hipSuccess, EVE_ERR_LAUNCH, "eve_route_bwd: cannot reserve %zu bytes of LDS", lds); \
      attr_bytes = lds;                                                                                            \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, dim3(nblk, ceil_div(M, Mb)), dim3(kRbThreads), lds, st, (const TT*)dxd, slot, dlogits,  \
                       wg, (TT*)dh, part, S, T, n_vis, mode, M, E, Mb);                                            \
  } while (0)
  if (dtype == EVE_F32) EVE_RB_LAUNCH(float); else EVE_RB_LAUNCH(bf16_t);
#undef EVE_RB_LAUNCH
  EVE_CHECK_LAUNCH("route_bwd_kernel");
  if (want_embed) {
    hipLaunchKernelGGL(embed_grad_final32_kernel, dim3(ceil_div(2 * M, 32)), dim3(1024), 0, st, part, dvision, dtext, M, nblk);
    EVE_CHECK_LAUNCH("embed_grad_final32_kernel");
  }
  return EVE_OK;
}
