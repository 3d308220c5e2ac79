// Device bodies of the gate weight-gradient kernels, shared by gate.hip (kernels of their own: eve_gate_wgrad) and
// permute.hip (launches that run them beside the route backward's blocks: the fused eve_moe_bwd).
#pragma once
#include "common.h"

namespace eve {

typedef __attribute__((ext_vector_type(16))) float gate_f32x16;

// Exact three-way split of eight fp32 values into bf16 parts, w = hi + mid + lo with every part the
// TRUNCATED top 8 significant bits of what is left (24 = 8 + 8 + 8, the subtractions are exact): a bf16
// MFMA over (x, hi), (x, mid), (x, lo) with bf16 x then forms the same exact products as an fp32 FMA and
// accumulates them in fp32, at 3 x 32 issue cycles per 16 k instead of 8 x 64 for v_mfma_f32_32x32x2_f32.
typedef __attribute__((ext_vector_type(8))) __bf16 gate_bf16x8;
__device__ __forceinline__ void split3_bf16(const float (&w)[8], gate_bf16x8& hi, gate_bf16x8& mid, gate_bf16x8& lo) {
  uint32_t ph[4], pm[4], pl[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint32_t h[2], m[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float x = w[2 * p + q];
      h[q] = __float_as_uint(x) & 0xffff0000u;
      const float r = x - __uint_as_float(h[q]);
      m[q] = __float_as_uint(r) & 0xffff0000u;
      const float r2 = r - __uint_as_float(m[q]);
      l[q] = __float_as_uint(r2);  // at most 8 significant bits left: exact in bf16
    }
    ph[p] = (h[0] >> 16) | h[1];
    pm[p] = (m[0] >> 16) | m[1];
    pl[p] = (l[0] >> 16) | (l[1] & 0xffff0000u);
  }
  hi = __builtin_bit_cast(gate_bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
  mid = __builtin_bit_cast(gate_bf16x8, make_uint4(pm[0], pm[1], pm[2], pm[3]));
  lo = __builtin_bit_cast(gate_bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
}

// bf16 activations: dwg partials on v_mfma_f32_32x32x16_bf16, 16 tokens per instruction.  x = bf16(h + embed)
// is an exact bf16 operand and dlogits goes in as its exact three-way bf16 split (split3_bf16), so the
// products are those of an fp32 FMA.  Block = TS tokens x 256 columns; the four waves split the COLUMNS
// (64 each: two MFMA tiles, lane i owns columns 2i and 2i+1 of its wave's slab), so no cross-wave
// reduction and 32 accumulator registers; a lane loads one dword per token row -- a half-wave reads 128
// contiguous bytes of a row -- and the B operand of tile t is half t of the 8 rows' dwords.
// (bx, by) = the block's token chunk and group of 256 columns: the block indices of the kernel of its own, positions in
// a block range of a launch shared with another kernel's blocks (permute.hip: route_wgrad_bf16_kernel).
template <bool EMB>
__device__ __forceinline__ void gate_wgrad_bf16_body(const bf16_t* __restrict__ h, const bf16_t* __restrict__ vemb,
                                                     const bf16_t* __restrict__ temb, const float* __restrict__ dlogits,
                                                     float* __restrict__ part, int S, int T_, int n_vis, int mode, int M,
                                                     int E, int TS, int bx, int by) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int i = lane & 31, hh = lane >> 5;
  const int col0 = by * 256 + wave * 64 + 2 * i;
  const bool col_ok = col0 < M;        // M % 8 == 0: a column pair is inside or outside as a whole
  const int colc = col_ok ? col0 : 0;  // lanes past the last column load column 0 and store nothing
  const int ic = min(i, E - 1);
  const int s_begin = bx * TS, s_end = min(S, s_begin + TS);
  uint32_t ev_v = 0, ev_t = 0;
  if constexpr (EMB) {
    ev_v = *reinterpret_cast<const uint32_t*>(vemb + colc);
    ev_t = *reinterpret_cast<const uint32_t*>(temb + colc);
  }
  gate_f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  constexpr int ST = 4;  // 16-token steps whose loads are issued together
  for (int s16 = s_begin; s16 < s_end; s16 += 16 * ST) {
    float a[ST][8];
    uint32_t xv[ST][8];
#pragma unroll
    for (int u = 0; u < ST; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // unconditional loads (clamped); dead tokens are masked through a
        const int sc = min(s16 + 16 * u + 8 * hh + j, S - 1);
        a[u][j] = dlogits[(size_t)sc * E + ic];
        xv[u][j] = *reinterpret_cast<const uint32_t*>(h + (size_t)sc * M + colc);
      }
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int sj0 = s16 + 16 * u + 8 * hh;
      int pos = 0;
      if constexpr (EMB) pos = mode == EVE_MODE_VL ? sj0 % T_ : 0;  // position inside the sequence
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[u][j] = (sj0 + j < s_end && i < E) ? a[u][j] : 0.f;
        if constexpr (EMB) {
          const bool text = mode == EVE_MODE_TEXT || (mode == EVE_MODE_VL && pos >= n_vis);
          const uint32_t e = text ? ev_t : ev_v;
          const float x0 = __uint_as_float(xv[u][j] << 16) + __uint_as_float(e << 16);
          const float x1 = __uint_as_float(xv[u][j] & 0xffff0000u) + __uint_as_float(e & 0xffff0000u);
          const uint32_t b0 = (uint32_t)__builtin_bit_cast(unsigned short, static_cast<bf16_t>(x0));
          const uint32_t b1 = (uint32_t)__builtin_bit_cast(unsigned short, static_cast<bf16_t>(x1));
          xv[u][j] = b0 | (b1 << 16);
          pos = pos + 1 == T_ ? 0 : pos + 1;
        }
      }
      gate_bf16x8 ah, am, al;
      split3_bf16(a[u], ah, am, al);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const uint32_t w0 = xv[u][2 * p], w1 = xv[u][2 * p + 1];
          w[p] = t ? ((w0 >> 16) | (w1 & 0xffff0000u)) : ((w0 & 0xffffu) | (w1 << 16));
        }
        const gate_bf16x8 bt = __builtin_bit_cast(gate_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bt, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bt, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bt, acc[t], 0, 0, 0);
      }
    }
  }
  if (!col_ok) return;
  float* out = part + (size_t)bx * E * M;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;  // expert
    if (row < E) *reinterpret_cast<float2*>(out + (size_t)row * M + col0) = make_float2(acc[0][r], acc[1][r]);
  }
}

// out[i] = sum over parts of part[p][i], in a fixed order: 4 slices of the parts summed by 4 thread
// groups, then the slices in order.  Block bx = 64 outputs x 4 slices = threads 0..255 (`red`: the block's float[4][64]).
__device__ __forceinline__ void sum_partials4_body(const float* __restrict__ part, float* __restrict__ out, int n,
                                                   int nparts, int bx, float (*red)[64]) {
  const bool mine = threadIdx.x < 256;
  const int o = threadIdx.x & 63, sl = (threadIdx.x >> 6) & 3;
  const int i = bx * 64 + o;
  const int per = (nparts + 3) / 4;
  const int p0 = sl * per, p1 = min(nparts, p0 + per);
  float acc = 0.f;
  if (mine && i < n) {
#pragma unroll 8
    for (int p = p0; p < p1; ++p) acc += part[(size_t)p * n + i];
  }
  if (mine) red[sl][o] = acc;
  __syncthreads();
  if (mine && sl == 0 && i < n) out[i] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

constexpr int kWgradBf16TS = 128;  // bf16-MFMA kernel: tokens per block (eight 16-token steps)

}  // namespace eve
