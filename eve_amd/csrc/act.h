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

}  // namespace eve
