// Backward of top2gating for one (token, expert) pair, E <= 32: d weight, d l_aux -> d logits (the formulae of SURVEY
// 8 A5: renormalised weights with the eps clamp of sharded_moe.py:221, the auxiliary loss of :196-201, softmax
// Jacobian).  A 32-lane half of a wave owns a token, lane = expert; shared by gate_bwd32_kernel (gate.hip) and by the
// tail of combine_bwd_np_kernel (permute.hip), which forms d weight for the same token in the same lanes -- one
// function, so that the fused launch is bit-identical to the two.
#pragma once
#include "common.h"

namespace eve {

constexpr float kGateEps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps, sharded_moe.py:221

// g: gates[s][e] (0 for e >= E); i1, i2: the token's two experts; k1, k2: kept (loc < C); dw1, dw2: d weight;
// aux: d l_aux * E / S^2 * exp_counts[e]; tk: which half of the wave the token occupies
__device__ __forceinline__ float gate_bwd32_token(float g, int i1, int i2, bool k1, bool k2, float dw1, float dw2, float aux,
                                                  int e, int tk) {
  const float gi1 = __shfl(g, tk * 32 + i1, 64), gi2 = __shfl(g, tk * 32 + i2, 64);
  const float g1 = k1 ? gi1 : 0.f, g2 = k2 ? gi2 : 0.f;
  const float sum = g1 + g2;
  float dg1, dg2;
  if (sum >= kGateEps) {  // clamp inactive: w_k = g_k / (g1 + g2)
    const float inv2 = 1.f / (sum * sum);
    dg1 = (dw1 - dw2) * g2 * inv2;
    dg2 = (dw2 - dw1) * g1 * inv2;
  } else {  // clamp active: w_k = g_k / eps
    dg1 = dw1 / kGateEps;
    dg2 = dw2 / kGateEps;
  }
  if (!k1) dg1 = 0.f;
  if (!k2) dg2 = 0.f;
  float dg = aux;
  if (e == i1) dg += dg1;
  if (e == i2) dg += dg2;
  float dot = g * dg;
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
  return g * (dg - dot);
}

}  // namespace eve
