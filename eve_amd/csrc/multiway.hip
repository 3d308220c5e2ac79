// Hard modality split of the non-MoE layers (reference models/multiway_transformer.py:366-370, and the
// single-modality branches :386-387, :402-403): vision positions go through norm2_imag + mlp_imag, the last
// n_text positions of every sequence through norm2_text + mlp_text.  The two MLPs run as ONE grouped GEMM
// (eve_ffn_fwd/bwd with two groups); the kernels here are the memory-bound passes around it:
//   split_ln_fwd     LayerNorm with the token's modality's affine, written straight into its group's rows
//   split_ln_bwd     its backward (input gradient + per-modality gamma/beta partial sums, two-stage)
//   split_permute    group rows <-> token order (the reference's slice + cat)
// Token (b,t) with t < n_vis is row b*n_vis + t of group 0, otherwise row b*(T-n_vis) + (t-n_vis) of group 1;
// group g starts at row g*R of the [2,R,M] buffer.
#include "common.h"

namespace eve {

__device__ __forceinline__ int split_dest(int s, int T, int n_vis, int R) {
  const int b = s / T, t = s - b * T;
  return t < n_vis ? b * n_vis + t : R + b * (T - n_vis) + (t - n_vis);
}

// A row of up to 1024 elements lives in one wave's registers: lane owns 16-byte chunks lane, lane+64, ...
template <typename T> struct RowCfg { static constexpr int N = Vec16<T>::N; static constexpr int MAXC = 1024 / (64 * N); };

template <typename T>
__global__ __launch_bounds__(256) void split_ln_fwd_kernel(const T* __restrict__ h, const T* __restrict__ gamma_v,
                                                           const T* __restrict__ beta_v, const T* __restrict__ gamma_t,
                                                           const T* __restrict__ beta_t, T* __restrict__ xg,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int S, int T_, int n_vis, int R, int M, float eps) {
  constexpr int N = RowCfg<T>::N, MAXC = RowCfg<T>::MAXC;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int s = wave; s < S; s += nwaves) {
    const int t = s % T_;
    const bool text = t >= n_vis;
    const T* g = text ? gamma_t : gamma_v;
    const T* be = text ? beta_t : beta_v;
    float x[MAXC][N];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < chunks) {
        const Vec16<T> v = load16<T>(h + (size_t)s * M + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) { x[k][j] = v.get(j); sum += x[k][j]; }
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) x[k][j] = 0.f;
      }
    }
    const float mean = wave_sum(sum) / (float)M;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (lane + 64 * k < chunks) {
#pragma unroll
        for (int j = 0; j < N; ++j) { const float d = x[k][j] - mean; sq += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)M + eps);
    T* dst = xg + (size_t)split_dest(s, T_, n_vis, R) * M;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < chunks) {
        const Vec16<T> gv = load16<T>(g + c * N), bv = load16<T>(be + c * N);
        Vec16<T> o;
        float ov[N];
#pragma unroll
        for (int j = 0; j < N; ++j) ov[j] = (x[k][j] - mean) * rstd * gv.get(j) + bv.get(j);
        o.set_all(ov);
        store16<T>(dst + c * N, o);
      }
    }
    if (lane == 0) { mean_out[s] = mean; rstd_out[s] = rstd; }
  }
}

// Backward.  Block = 32 consecutive tokens, 8 per wave.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),
// g = dy * gamma; per-lane partial sums of dy * xhat (dgamma) and dy (dbeta) per modality are reduced over the
// block's 4 waves in wave order and written to part[block][4][M] = (dgamma_v, dbeta_v, dgamma_t, dbeta_t).
constexpr int kLnTok = 32;
template <typename T>
__global__ __launch_bounds__(256) void split_ln_bwd_kernel(const T* __restrict__ dxg, const T* __restrict__ h,
                                                           const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in,
                                                           const T* __restrict__ gamma_v, const T* __restrict__ gamma_t,
                                                           T* __restrict__ dh, float* __restrict__ part, int S, int T_,
                                                           int n_vis, int R, int M) {
  constexpr int N = RowCfg<T>::N, MAXC = RowCfg<T>::MAXC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][4][M]
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int chunks = M / N;
  float ag[2][MAXC][N], ab[2][MAXC][N];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
#pragma unroll
      for (int j = 0; j < N; ++j) ag[m][k][j] = ab[m][k][j] = 0.f;
  for (int r = 0; r < kLnTok / 4; ++r) {
    const int s = blockIdx.x * kLnTok + wave * (kLnTok / 4) + r;
    if (s >= S) break;  // wave-uniform
    const bool text = (s % T_) >= n_vis;
    const T* g = text ? gamma_t : gamma_v;
    const T* src = dxg + (size_t)split_dest(s, T_, n_vis, R) * M;
    const float mean = mean_in[s], rstd = rstd_in[s];
    float xh[MAXC][N], gg[MAXC][N], dy[MAXC][N];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < chunks) {
        const Vec16<T> xv = load16<T>(h + (size_t)s * M + c * N), dv = load16<T>(src + c * N), gv = load16<T>(g + c * N);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          xh[k][j] = (xv.get(j) - mean) * rstd;
          dy[k][j] = dv.get(j);
          gg[k][j] = dy[k][j] * gv.get(j);
          c1 += gg[k][j];
          c2 += gg[k][j] * xh[k][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) xh[k][j] = gg[k][j] = dy[k][j] = 0.f;
      }
    }
    c1 = wave_sum(c1) / (float)M;
    c2 = wave_sum(c2) / (float)M;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < chunks) {
        Vec16<T> o;
        float ov[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          ov[j] = rstd * (gg[k][j] - c1 - xh[k][j] * c2);
          if (text) { ag[1][k][j] += dy[k][j] * xh[k][j]; ab[1][k][j] += dy[k][j]; }
          else { ag[0][k][j] += dy[k][j] * xh[k][j]; ab[0][k][j] += dy[k][j]; }
        }
        o.set_all(ov);
        store16<T>(dh + (size_t)s * M + c * N, o);
      }
    }
  }
  float* mine = red + (size_t)wave * 4 * M;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < chunks) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        mine[0 * M + c * N + j] = ag[0][k][j];
        mine[1 * M + c * N + j] = ab[0][k][j];
        mine[2 * M + c * N + j] = ag[1][k][j];
        mine[3 * M + c * N + j] = ab[1][k][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * M; i += blockDim.x)
    part[(size_t)blockIdx.x * 4 * M + i] = ((red[i] + red[4 * M + i]) + red[8 * M + i]) + red[12 * M + i];
}

// out[i] = sum over blocks of part[b][i], fixed order: 32 columns x 32 slices of the partial rows per block
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            int ncols, int nchunks) {
  __shared__ float red[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (col < ncols) {
#pragma unroll 4
    for (int b = ty; b < nchunks; b += 32) acc += part[(size_t)b * ncols + col];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < ncols) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][tx];
    out[col] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void split_permute_kernel(const T* __restrict__ src, T* __restrict__ dst, int S, int T_,
                                                            int n_vis, int R, int M, int to_groups) {
  constexpr int N = Vec16<T>::N;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int chunks = M / N;
  for (int s = wave; s < S; s += nwaves) {
    const size_t grow = (size_t)split_dest(s, T_, n_vis, R) * M, trow = (size_t)s * M;
    const T* from = src + (to_groups ? trow : grow);
    T* to = dst + (to_groups ? grow : trow);
    for (int c = lane; c < chunks; c += 64) store16<T>(to + c * N, load16<T>(from + c * N));
  }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int check_split(const char* who, int B, int T, int n_vis, int R, int M, int dtype, bool rows_in_regs) {
  if (dtype != EVE_F32 && dtype != EVE_BF16) { set_error("%s: bad dtype %d", who, dtype); return EVE_ERR_ARG; }
  const int n = dtype == EVE_BF16 ? 8 : 4;
  if (B <= 0 || T <= 0 || n_vis < 0 || n_vis > T) { set_error("%s: bad shape B=%d T=%d n_vis=%d", who, B, T, n_vis); return EVE_ERR_ARG; }
  if (M <= 0 || M % n != 0) { set_error("%s: M=%d must be a positive multiple of %d", who, M, n); return EVE_ERR_ARG; }
  if (rows_in_regs && M > 1024) { set_error("%s: M=%d > 1024 is not supported", who, M); return EVE_ERR_UNSUPPORTED; }
  const long need = (long)B * (n_vis > T - n_vis ? n_vis : T - n_vis);
  if (R < need) { set_error("%s: R=%d rows per group < %ld", who, R, need); return EVE_ERR_ARG; }
  return EVE_OK;
}

}  // namespace eve

using namespace eve;

extern "C" int eve_split_ln_fwd(const void* h, const void* gamma_v, const void* beta_v, const void* gamma_t,
                                const void* beta_t, void* xg, float* mean, float* rstd, int B, int T, int n_vis, int R,
                                int M, float eps, int dtype, void* stream) {
  EVE_REQUIRE(h && gamma_v && beta_v && gamma_t && beta_t && xg && mean && rstd, EVE_ERR_ARG, "eve_split_ln_fwd: null pointer");
  if (int rc = check_split("eve_split_ln_fwd", B, T, n_vis, R, M, dtype, true)) return rc;
  const int S = B * T;
  const int grid = max(1, min(ceil_div(S, 4), 4096));
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((split_ln_fwd_kernel<float>), dim3(grid), dim3(256), 0, S_(stream), (const float*)h,
                       (const float*)gamma_v, (const float*)beta_v, (const float*)gamma_t, (const float*)beta_t,
                       (float*)xg, mean, rstd, S, T, n_vis, R, M, eps);
  else
    hipLaunchKernelGGL((split_ln_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, S_(stream), (const bf16_t*)h,
                       (const bf16_t*)gamma_v, (const bf16_t*)beta_v, (const bf16_t*)gamma_t, (const bf16_t*)beta_t,
                       (bf16_t*)xg, mean, rstd, S, T, n_vis, R, M, eps);
  EVE_CHECK_LAUNCH("split_ln_fwd_kernel");
  return EVE_OK;
}

extern "C" size_t eve_split_ln_workspace_bytes(int S, int M) {
  if (S <= 0 || M <= 0) return 0;
  return (size_t)ceil_div(S, kLnTok) * 4 * M * sizeof(float);
}

extern "C" int eve_split_ln_bwd(const void* dxg, const void* h, const float* mean, const float* rstd, const void* gamma_v,
                                const void* gamma_t, void* dh, float* dparams, int B, int T, int n_vis, int R, int M,
                                int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(dxg && h && mean && rstd && gamma_v && gamma_t && dh && dparams && workspace, EVE_ERR_ARG,
              "eve_split_ln_bwd: null pointer");
  if (int rc = check_split("eve_split_ln_bwd", B, T, n_vis, R, M, dtype, true)) return rc;
  const int S = B * T;
  const size_t need = eve_split_ln_workspace_bytes(S, M);
  EVE_REQUIRE(workspace_bytes >= need, EVE_ERR_WORKSPACE, "eve_split_ln_bwd: workspace %zu < %zu", workspace_bytes, need);
  const int nblk = ceil_div(S, kLnTok);
  const size_t lds = (size_t)16 * M * sizeof(float);
  float* part = static_cast<float*>(workspace);
  hipStream_t st = S_(stream);
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((split_ln_bwd_kernel<float>), dim3(nblk), dim3(256), lds, st, (const float*)dxg, (const float*)h,
                       mean, rstd, (const float*)gamma_v, (const float*)gamma_t, (float*)dh, part, S, T, n_vis, R, M);
  else
    hipLaunchKernelGGL((split_ln_bwd_kernel<bf16_t>), dim3(nblk), dim3(256), lds, st, (const bf16_t*)dxg,
                       (const bf16_t*)h, mean, rstd, (const bf16_t*)gamma_v, (const bf16_t*)gamma_t, (bf16_t*)dh, part, S,
                       T, n_vis, R, M);
  EVE_CHECK_LAUNCH("split_ln_bwd_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(4 * M, 32)), dim3(1024), 0, st, part, dparams, 4 * M, nblk);
  EVE_CHECK_LAUNCH("colsum_final_kernel");
  return EVE_OK;
}

extern "C" int eve_split_permute(const void* src, void* dst, int B, int T, int n_vis, int R, int M, int to_groups,
                                 int dtype, void* stream) {
  EVE_REQUIRE(src && dst, EVE_ERR_ARG, "eve_split_permute: null pointer");
  if (int rc = check_split("eve_split_permute", B, T, n_vis, R, M, dtype, false)) return rc;
  const int S = B * T;
  const int grid = max(1, min(ceil_div(S, 4), 4096));
  if (dtype == EVE_F32)
    hipLaunchKernelGGL((split_permute_kernel<float>), dim3(grid), dim3(256), 0, S_(stream), (const float*)src, (float*)dst,
                       S, T, n_vis, R, M, to_groups);
  else
    hipLaunchKernelGGL((split_permute_kernel<bf16_t>), dim3(grid), dim3(256), 0, S_(stream), (const bf16_t*)src,
                       (bf16_t*)dst, S, T, n_vis, R, M, to_groups);
  EVE_CHECK_LAUNCH("split_permute_kernel");
  return EVE_OK;
}
