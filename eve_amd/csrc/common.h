// Shared helpers for the eve_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eve_moe.h"

namespace eve {

typedef __bf16 bf16_t;

constexpr int kWave = 64;

// ---- error plumbing -------------------------------------------------------------
void set_error(const char* fmt, ...);

#define EVE_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      ::eve::set_error(__VA_ARGS__);        \
      return (code);                        \
    }                                       \
  } while (0)

#define EVE_CHECK_LAUNCH(name)                                                         \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) {                                                            \
      ::eve::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));          \
      return EVE_ERR_LAUNCH;                                                           \
    }                                                                                  \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- element conversion ---------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return static_cast<float>(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }  // RNE

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
  __device__ __forceinline__ void set(int i, float v) { (&raw.x)[i] = v; }
  __device__ __forceinline__ void set_all(const float (&v)[4]) { raw = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    uint32_t w = (&raw.x)[i >> 1];
    uint32_t bits = (i & 1) ? (w & 0xffff0000u) : (w << 16);
    return __uint_as_float(bits);
  }
  __device__ __forceinline__ void set(int i, float v) {
    bf16_t b = static_cast<bf16_t>(v);
    uint32_t h = (uint32_t)__builtin_bit_cast(unsigned short, b);
    uint32_t& w = (&raw.x)[i >> 1];
    w = (i & 1) ? ((w & 0x0000ffffu) | (h << 16)) : ((w & 0xffff0000u) | h);
  }
  // all eight at once: one v_cvt_pk_bf16_f32 per pair (element-wise set() costs four merge ops per value)
  __device__ __forceinline__ void set_all(const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    uint32_t w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f2 t = {v[2 * p], v[2 * p + 1]};
      w[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, b2));
    }
    raw = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> v;
  v.raw = *reinterpret_cast<const decltype(v.raw)*>(p);
  return v;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<decltype(v.raw)*>(p) = v.raw;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  Vec16<T> v;
  v.raw.x = 0; v.raw.y = 0; v.raw.z = 0; v.raw.w = 0;
  return v;
}
template <> __device__ __forceinline__ Vec16<float> zero16<float>() {
  Vec16<float> v;
  v.raw = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

// ---- wave primitives (64 lanes) ---------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}

// modality of token s (row-major (b,t) flattening): 0 vision, 1 text.  mode: EVE_MODE_*
__device__ __forceinline__ int token_modality(int s, int T, int n_vis, int mode) {
  if (mode == EVE_MODE_VISION) return 0;
  if (mode == EVE_MODE_TEXT) return 1;
  return (s % T) < n_vis ? 0 : 1;
}

// permute.hip: the closing stage of the fused backward (eve_moe_bwd), eve_route_bwd + eve_gate_wgrad in two launches
// gate.hip: eve_gate_top2_fwd whose closing kernel also writes exp_counts to `exp_counts_mirror` (NULL = no mirror)
int gate_top2_fwd_mirrored(const float* logits, const float* noise, int noise_is_uniform, int S, int E, int C, float* gates,
                           int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token,
                           float* slot_weight, int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace,
                           size_t workspace_bytes, void* stream, int64_t* exp_counts_mirror);
// gate.hip: eve_gate_logits_fwd + eve_gate_top2_fwd as two launches (logits with the selection in their tail, then the
// assignment), for the shapes gate_logits_top2_fusable accepts; same results as the two calls
bool gate_logits_top2_fusable(int S, int M, int E, int dtype);
int gate_logits_top2_fused(const void* h, const void* ve, const void* te, const float* wg, const float* noise,
                           int noise_is_uniform, int S, int T_, int n_vis, int mode, int M, int E, int C, float* gates,
                           int32_t* idx, int32_t* loc, int32_t* slot, float* weight, int32_t* slot_token, float* slot_weight,
                           int64_t* exp_counts, int32_t* rows, float* l_aux, void* workspace, size_t workspace_bytes,
                           void* stream, int64_t* exp_counts_mirror);
bool combine_gate_bwd_fusable(const void* dout, const void* y, const int32_t* slot, const void* dy, const int32_t* idx,
                              const int32_t* loc, int M, int E, int dtype);
int combine_bwd_gate_bwd(const void* dout, const void* y, const int32_t* slot, const int32_t* slot_token, const float* slot_weight,
                         void* dy, float* d_weight, const float* gates, const int32_t* idx, const int32_t* loc,
                         const float* d_l_aux, const int64_t* exp_counts, float* dlogits, int S, int M, int E, int C,
                         int num_slots, int dtype, void* stream);
bool route_wgrad_fusable(const void* dxd, const int32_t* slot, const void* dh, const void* h, int M, int E, int dtype);
int route_bwd_gate_wgrad(const void* dxd, const int32_t* slot, const float* dlogits, const float* wg, void* dh, void* dvision,
                         void* dtext, const void* h, const void* vision_embed, const void* text_embed, float* dwg, int S, int T,
                         int n_vis, int mode, int M, int E, int embed_dtype, void* ws_embed, size_t ws_embed_bytes,
                         void* ws_wgrad, size_t ws_wgrad_bytes, void* stream);

}  // namespace eve
