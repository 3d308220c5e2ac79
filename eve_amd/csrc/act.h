// Expert activations (models/multiway_transformer.py:47, ACT2FN[hidden_act]) and derivatives.
#pragma once
#include "common.h"

namespace eve {

template <bool FAST> __device__ __forceinline__ float exp_(float x) { if constexpr (FAST) return __expf(x); else return expf(x); }
// 1/x: v_rcp_f32 (1 ulp) on the fast path, IEEE division otherwise
template <bool FAST> __device__ __forceinline__ float rcp_(float x) { if constexpr (FAST) return __builtin_amdgcn_rcpf(x); else return 1.f / x; }

// sigmoid(1.702 z); the fast path folds the exponent's scale into one multiply: 2^(z * (-1.702 log2 e))
template <bool FAST> __device__ __forceinline__ float qg_sigmoid(float z) {
  if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * -2.45546696f));
  else return 1.f / (1.f + expf(-1.702f * z));
}

template <int ACT, bool FAST> __device__ __forceinline__ float act_fwd(float z) {
  if constexpr (ACT == EVE_ACT_QUICK_GELU) {
    return z * qg_sigmoid<FAST>(z);
  } else if constexpr (ACT == EVE_ACT_GELU) {
    return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
  } else if constexpr (ACT == EVE_ACT_RELU) {
    return z > 0.f ? z : 0.f;
  } else {
    return z;
  }
}

template <int ACT, bool FAST> __device__ __forceinline__ float act_bwd(float z) {
  if constexpr (ACT == EVE_ACT_QUICK_GELU) {
    const float s = qg_sigmoid<FAST>(z);
    return s * (1.f + 1.702f * z * (1.f - s));
  } else if constexpr (ACT == EVE_ACT_GELU) {
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * exp_<FAST>(-0.5f * z * z);
  } else if constexpr (ACT == EVE_ACT_RELU) {
    return z > 0.f ? 1.f : 0.f;
  } else {
    return 1.f;
  }
}

// Two elements at a time on v_pk_*_f32 (a wave's VALU issues one instruction per ~4 cycles whatever it is, so the
// packed forms halve the issue time of the GEMM epilogues, which are bound by it): forward activation `a` and the
// derivative `g` = act'(z) from ONE sigmoid.  The MFMA expert-FFN kernels save g (not z) for the backward pass,
// whose epilogue is then a multiply: with t = z * (-1.702 log2 e), s = 1 / (1 + 2^t):
//   a = z s,   g = s (1 + 1.702 z (1 - s)) = s + 1.702 (a - a s).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int ACT, bool WANT_G>
__device__ __forceinline__ void act_fwd_grad2(f32x2_t z, f32x2_t& a, f32x2_t& g) {
  if constexpr (ACT == EVE_ACT_QUICK_GELU) {
    const f32x2_t t = z * -2.45546696f;
    f32x2_t d;
    d.x = __builtin_amdgcn_exp2f(t.x);
    d.y = __builtin_amdgcn_exp2f(t.y);
    d = d + 1.f;
    f32x2_t s;
    s.x = __builtin_amdgcn_rcpf(d.x);
    s.y = __builtin_amdgcn_rcpf(d.y);
    a = z * s;
    if constexpr (WANT_G) {
      const f32x2_t k = {1.702f, 1.702f};
      g = __builtin_elementwise_fma(k, a - a * s, s);  // s (1 + 1.702 z (1 - s)) with z s = a
    }
  } else {
    a.x = act_fwd<ACT, true>(z.x);
    a.y = act_fwd<ACT, true>(z.y);
    if constexpr (WANT_G) {
      g.x = act_bwd<ACT, true>(z.x);
      g.y = act_bwd<ACT, true>(z.y);
    }
  }
}

}  // namespace eve
