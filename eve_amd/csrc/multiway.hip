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

// ------------------------------------------------------------ layer-scale residual branch
// out = residual + (x * gamma) * row_scale: `residual + self.drop_path(hidden_states * self.gamma_k)` of the encoder
// layer (reference models/multiway_transformer.py:341 / :408) in one pass instead of a multiply, the DropPath
// multiply and an add.  Each product and the sum are rounded to fp32 separately (no contraction), so the fp32 result
// is the reference's op chain bit for bit; bf16 rounds once at the end.  row_scale [B] = mask / keep_prob of DropPath
// (NULL = no DropPath), gamma NULL = no layer scale.  Thread per 16-byte chunk, grid-stride: a plain stream.
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                                 const T* __restrict__ row_scale, const T* __restrict__ residual,
                                                                 T* __restrict__ out, long nchunks, int chunks_per_row, int T_) {
  constexpr int N = Vec16<T>::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const long row = i / chunks_per_row;
    const int c = (int)(i - row * chunks_per_row);
    const Vec16<T> xv = load16<T>(x + i * N), rv = load16<T>(residual + i * N);
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = xv.get(j);
    if (gamma) {
      const Vec16<T> g = load16<T>(gamma + c * N);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = v[j] * g.get(j);
    }
    if (row_scale) {
      const float rs = to_f32<T>(row_scale[row / T_]);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = v[j] * rs;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = rv.get(j) + v[j];
    Vec16<T> o;
    o.set_all(v);
    store16<T>(out + i * N, o);
  }
}

// Backward: dx = (dout * row_scale) * gamma and the per-block partial column sums of (dout * row_scale) * x (the
// gradient of gamma).  Block = kSrTok rows; thread (rsub, c) walks rows rsub, rsub + nsub, ... of the block for
// column chunk c; the row sub-groups are combined through LDS in a fixed order.
constexpr int kSrTok = 16;
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                                 const T* __restrict__ gamma, const T* __restrict__ row_scale,
                                                                 T* __restrict__ dx, float* __restrict__ part, int S, int T_, int M) {
  constexpr int N = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [nsub][M]
  const int chunks = M / N;
  const int nsub = max(1, (int)blockDim.x / chunks);
  const int r0 = blockIdx.x * kSrTok, r1 = min(S, r0 + kSrTok);
  for (int c0 = 0; c0 < chunks; c0 += blockDim.x) {  // one trip unless a row has more chunks than the block threads
    const int rsub = chunks >= (int)blockDim.x ? 0 : threadIdx.x / chunks;
    const int c = chunks >= (int)blockDim.x ? c0 + threadIdx.x : threadIdx.x % chunks;
    const bool on = c < chunks && rsub < nsub;
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    if (on) {
      Vec16<T> g;
      if (gamma) g = load16<T>(gamma + c * N);
      constexpr int U = 4;  // rows whose loads are issued together
      for (int rb = r0 + rsub; rb < r1; rb += U * nsub) {
        Vec16<T> dv[U], xv[U];
        float rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = min(rb + u * nsub, r1 - 1);  // clamped: rows past the end are loaded again and not stored
          const size_t o = (size_t)r * M + (size_t)c * N;
          dv[u] = load16<T>(dout + o);
          if (gamma) xv[u] = load16<T>(x + o);
          rs[u] = row_scale ? to_f32<T>(row_scale[r / T_]) : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = rb + u * nsub;
          if (r >= r1) break;
          const size_t o = (size_t)r * M + (size_t)c * N;
          float d[N];
#pragma unroll
          for (int j = 0; j < N; ++j) d[j] = dv[u].get(j);
          if (row_scale) {
#pragma unroll
            for (int j = 0; j < N; ++j) d[j] = d[j] * rs[u];
          }
          Vec16<T> ov;
          if (gamma) {
            float dxv[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
              acc[j] += d[j] * xv[u].get(j);
              dxv[j] = d[j] * g.get(j);
            }
            ov.set_all(dxv);
          } else {
            ov.set_all(d);
          }
          store16<T>(dx + o, ov);
        }
      }
    }
    if (!part) continue;  // block-uniform
    if (on) {
#pragma unroll
      for (int j = 0; j < N; ++j) red[rsub * M + c * N + j] = acc[j];
    }
    __syncthreads();
    const int cols_here = min(chunks - c0, (int)blockDim.x) * N;
    for (int i = threadIdx.x; i < cols_here; i += blockDim.x) {
      float sum = red[c0 * N + i];
      for (int k = 1; k < nsub; ++k) sum += red[k * M + c0 * N + i];
      part[(size_t)blockIdx.x * M + c0 * N + i] = sum;
    }
    __syncthreads();
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

extern "C" size_t eve_scale_residual_workspace_bytes(int S, int M) {
  if (S <= 0 || M <= 0) return 0;
  return (size_t)ceil_div(S, kSrTok) * M * sizeof(float);
}

static int check_scale_residual(const char* who, int S, int T, int M, int dtype, bool has_row_scale) {
  if (dtype != EVE_F32 && dtype != EVE_BF16) { set_error("%s: bad dtype %d", who, dtype); return EVE_ERR_ARG; }
  const int n = dtype == EVE_BF16 ? 8 : 4;
  if (S <= 0 || M <= 0 || M % n != 0) { set_error("%s: bad shape S=%d M=%d (M %% %d)", who, S, M, n); return EVE_ERR_ARG; }
  if (has_row_scale && (T <= 0 || S % T != 0)) { set_error("%s: S=%d not a multiple of T=%d", who, S, T); return EVE_ERR_ARG; }
  return EVE_OK;
}

extern "C" int eve_scale_residual_fwd(const void* x, const void* gamma, const void* row_scale, const void* residual,
                                      void* out, int S, int T, int M, int dtype, void* stream) {
  EVE_REQUIRE(x && residual && out, EVE_ERR_ARG, "eve_scale_residual_fwd: null pointer");
  if (int rc = check_scale_residual("eve_scale_residual_fwd", S, T, M, dtype, row_scale != nullptr)) return rc;
  const int n = dtype == EVE_BF16 ? 8 : 4;
  const long nchunks = (long)S * (M / n);
  const int grid = (int)max(1L, min((nchunks + 255) / 256, 8192L));
  if (dtype == EVE_F32)
    hipLaunchKernelGGL(scale_residual_fwd_kernel<float>, dim3(grid), dim3(256), 0, S_(stream), (const float*)x, (const float*)gamma,
                       (const float*)row_scale, (const float*)residual, (float*)out, nchunks, M / n, max(T, 1));
  else
    hipLaunchKernelGGL(scale_residual_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, S_(stream), (const bf16_t*)x,
                       (const bf16_t*)gamma, (const bf16_t*)row_scale, (const bf16_t*)residual, (bf16_t*)out, nchunks, M / n,
                       max(T, 1));
  EVE_CHECK_LAUNCH("scale_residual_fwd_kernel");
  return EVE_OK;
}

extern "C" int eve_scale_residual_bwd(const void* dout, const void* x, const void* gamma, const void* row_scale, void* dx,
                                      float* dgamma, int S, int T, int M, int dtype, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  EVE_REQUIRE(dout && dx, EVE_ERR_ARG, "eve_scale_residual_bwd: null pointer");
  EVE_REQUIRE(!gamma || (x && dgamma), EVE_ERR_ARG, "eve_scale_residual_bwd: x and dgamma are required with gamma");
  if (int rc = check_scale_residual("eve_scale_residual_bwd", S, T, M, dtype, row_scale != nullptr)) return rc;
  const size_t need = gamma ? eve_scale_residual_workspace_bytes(S, M) : 0;
  EVE_REQUIRE(!need || (workspace && workspace_bytes >= need), EVE_ERR_WORKSPACE, "eve_scale_residual_bwd: workspace %zu < %zu",
              workspace_bytes, need);
  const int n = dtype == EVE_BF16 ? 8 : 4;
  const int chunks = M / n;
  const int nsub = chunks >= 256 ? 1 : 256 / chunks;
  const size_t lds = (size_t)nsub * M * sizeof(float);
  EVE_REQUIRE(lds <= 64 * 1024, EVE_ERR_UNSUPPORTED, "eve_scale_residual_bwd: M=%d too wide", M);
  const int nblk = ceil_div(S, kSrTok);
  float* part = gamma ? static_cast<float*>(workspace) : nullptr;
  hipStream_t st = S_(stream);
  if (dtype == EVE_F32)
    hipLaunchKernelGGL(scale_residual_bwd_kernel<float>, dim3(nblk), dim3(256), lds, st, (const float*)dout, (const float*)x,
                       (const float*)gamma, (const float*)row_scale, (float*)dx, part, S, max(T, 1), M);
  else
    hipLaunchKernelGGL(scale_residual_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), lds, st, (const bf16_t*)dout, (const bf16_t*)x,
                       (const bf16_t*)gamma, (const bf16_t*)row_scale, (bf16_t*)dx, part, S, max(T, 1), M);
  EVE_CHECK_LAUNCH("scale_residual_bwd_kernel");
  if (gamma) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(M, 32)), dim3(1024), 0, st, part, dgamma, M, nblk);
    EVE_CHECK_LAUNCH("colsum_final_kernel");
  }
  return EVE_OK;
}
