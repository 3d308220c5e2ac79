// Grouped expert FFN (models/experts.py:18-28 over models/multiway_transformer.py:51-55):
// entry points eve_ffn_fwd / eve_ffn_bwd.  bf16 goes to the MFMA kernels in ffn_mfma.hip;
// fp32 (the 1e-5 parity path) and shapes the MFMA tiling does not cover run on the generic
// strided grouped GEMM below (fp32 FMA, LDS-tiled).
#include "act.h"
#include "common.h"
#include "ffn_mfma.h"

namespace eve {

enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_ACT = 2, EPI_DACT = 3 };

struct GemmDesc {
  const void* A;
  const void* B;
  void* C;
  void* C2;          // EPI_BIAS_ACT: activation output
  const void* bias;  // [E_local][N]
  const void* Zin;   // EPI_DACT: pre-activation, same layout as C
  long a_g, a_m, a_k;
  long b_e, b_k, b_n;
  long c_g, c_m;
  int Mext, N, K;
  int G, E_local;
  const int32_t* rows;
};

constexpr int GT = 64, GK = 16;

// WGRAD=false: block (nt, mt, g): C[g] = A[g] (rows[g] x K) . B[g % E_local] (K x N)
// WGRAD=true : block (nt, mt, e): C[e] = sum over groups g == e (mod E_local) of
//              A[g]^T-like (Mext x rows[g]) . B[g] (rows[g] x N), both addressed by strides.
template <typename T, int EPI, int ACT, bool WGRAD>
__global__ __launch_bounds__(256) void grouped_gemm_generic(GemmDesc d) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n0 = blockIdx.x * GT, m0 = blockIdx.y * GT;
  const int z = blockIdx.z;
  const T* A = static_cast<const T*>(d.A);
  const T* B = static_cast<const T*>(d.B);
  int mext = d.Mext;
  if (!WGRAD && d.rows) mext = min(mext, d.rows[z]);
  if (m0 >= mext) return;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  const int nseg = WGRAD ? d.G / d.E_local : 1;
  for (int seg = 0; seg < nseg; ++seg) {
    const int g = WGRAD ? seg * d.E_local + z : z;
    const int e = WGRAD ? z : z % d.E_local;
    int kext = d.K;
    if (WGRAD && d.rows) kext = min(kext, d.rows[g]);
    const T* Ag = A + (size_t)g * d.a_g;
    const T* Bg = B + (WGRAD ? (size_t)g * d.b_e : (size_t)e * d.b_e);
    for (int k0 = 0; k0 < kext; k0 += GK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m, k;
        if (d.a_k == 1) { k = threadIdx.x & 15; m = (threadIdx.x >> 4) + 16 * i; }
        else { m = threadIdx.x & 63; k = (threadIdx.x >> 6) + 4 * i; }
        float v = 0.f;
        if (m0 + m < mext && k0 + k < kext) v = to_f32<T>(Ag[(size_t)(m0 + m) * d.a_m + (size_t)(k0 + k) * d.a_k]);
        As[k][m] = v;
        int n, kb;
        if (d.b_k == 1) { kb = threadIdx.x & 15; n = (threadIdx.x >> 4) + 16 * i; }
        else { n = threadIdx.x & 63; kb = (threadIdx.x >> 6) + 4 * i; }
        float w = 0.f;
        if (n0 + n < d.N && k0 + kb < kext) w = to_f32<T>(Bg[(size_t)(k0 + kb) * d.b_k + (size_t)(n0 + n) * d.b_n]);
        Bs[kb][n] = w;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < GK; ++k) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
      __syncthreads();
    }
  }

  const int e_out = WGRAD ? z : z % d.E_local;
  T* C = static_cast<T*>(d.C) + (size_t)z * d.c_g;
  T* C2 = d.C2 ? static_cast<T*>(d.C2) + (size_t)z * d.c_g : nullptr;
  const T* Zin = d.Zin ? static_cast<const T*>(d.Zin) + (size_t)z * d.c_g : nullptr;
  const T* bias = d.bias ? static_cast<const T*>(d.bias) + (size_t)e_out * d.N : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= mext) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= d.N) continue;
      float v = acc[i][j];
      const size_t o = (size_t)m * d.c_m + n;
      if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_ACT) v += to_f32<T>(bias[n]);
      if constexpr (EPI == EPI_BIAS_ACT) {
        if (d.C) C[o] = from_f32<T>(v);
        C2[o] = from_f32<T>(act_fwd<ACT, false>(v));
      } else if constexpr (EPI == EPI_DACT) {
        C[o] = from_f32<T>(v * act_bwd<ACT, false>(to_f32<T>(Zin[o])));
      } else {
        C[o] = from_f32<T>(v);
      }
    }
  }
}

// Column sums for bias gradients: out[e][n] = sum over groups g == e (mod E_local), rows r <
// rows[g] of X[g][r][n].  One thread per column, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ X, T* __restrict__ out,
                                                        const int32_t* __restrict__ rows, int G, int E_local, int R,
                                                        int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (n >= N) return;
  float acc = 0.f;
  for (int g = e; g < G; g += E_local) {
    const int r_end = rows ? min(R, rows[g]) : R;
    const T* p = X + (size_t)g * R * N + n;
#pragma unroll 4
    for (int r = 0; r < r_end; ++r) acc += to_f32<T>(p[(size_t)r * N]);
  }
  out[(size_t)e * N + n] = from_f32<T>(acc);
}

template <typename T, int EPI, bool WGRAD>
static int launch_generic(const GemmDesc& d, int act, hipStream_t st) {
  dim3 grid(ceil_div(d.N, GT), ceil_div(d.Mext, GT), WGRAD ? d.E_local : d.G);
#define EVE_ACT_CASE(A)                                                                              \
  case A:                                                                                            \
    hipLaunchKernelGGL((grouped_gemm_generic<T, EPI, A, WGRAD>), grid, dim3(256), 0, st, d);         \
    break;
  if constexpr (EPI == EPI_BIAS_ACT || EPI == EPI_DACT) {
    switch (act) {
      EVE_ACT_CASE(EVE_ACT_QUICK_GELU)
      EVE_ACT_CASE(EVE_ACT_GELU)
      EVE_ACT_CASE(EVE_ACT_RELU)
      EVE_ACT_CASE(EVE_ACT_IDENTITY)
      default:
        set_error("eve_ffn: unknown activation %d", act);
        return EVE_ERR_ARG;
    }
  } else {
    hipLaunchKernelGGL((grouped_gemm_generic<T, EPI, EVE_ACT_IDENTITY, WGRAD>), grid, dim3(256), 0, st, d);
  }
#undef EVE_ACT_CASE
  EVE_CHECK_LAUNCH("grouped_gemm_generic");
  return EVE_OK;
}

template <typename T>
static int ffn_fwd_generic(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z,
                           void* a, void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act,
                           hipStream_t st) {
  GemmDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows; d.Mext = R;
  // z = x @ w1^T + b1 ; a = act(z)
  d.A = x; d.a_g = (long)R * M; d.a_m = M; d.a_k = 1;
  d.B = w1; d.b_e = (long)I * M; d.b_k = 1; d.b_n = M;
  d.C = z; d.C2 = a; d.c_g = (long)R * I; d.c_m = I; d.bias = b1; d.N = I; d.K = M;
  if (int rc = launch_generic<T, EPI_BIAS_ACT, false>(d, act, st)) return rc;
  // y = a @ w2^T + b2
  d.A = a; d.a_g = (long)R * I; d.a_m = I; d.a_k = 1;
  d.B = w2; d.b_e = (long)M * I; d.b_k = 1; d.b_n = I;
  d.C = y; d.C2 = nullptr; d.c_g = (long)R * M; d.c_m = M; d.bias = b2; d.N = M; d.K = I;
  return launch_generic<T, EPI_BIAS, false>(d, act, st);
}

template <typename T>
static int ffn_bwd_generic(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2,
                           void* dz, void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G,
                           int E_local, int R, int M, int I, int act, hipStream_t st) {
  GemmDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows;
  // dz = (dy @ w2) * act'(z)            [G,R,I];  w2 is [E,M,I]: B(k=m, n=i)
  d.Mext = R;
  d.A = dy; d.a_g = (long)R * M; d.a_m = M; d.a_k = 1;
  d.B = w2; d.b_e = (long)M * I; d.b_k = I; d.b_n = 1;
  d.C = dz; d.c_g = (long)R * I; d.c_m = I; d.Zin = z; d.N = I; d.K = M;
  if (int rc = launch_generic<T, EPI_DACT, false>(d, act, st)) return rc;
  // dx = dz @ w1                         [G,R,M];  w1 is [E,I,M]: B(k=i, n=m)
  d.A = dz; d.a_g = (long)R * I; d.a_m = I; d.a_k = 1;
  d.B = w1; d.b_e = (long)I * M; d.b_k = M; d.b_n = 1;
  d.C = dx; d.c_g = (long)R * M; d.c_m = M; d.Zin = nullptr; d.N = M; d.K = I;
  if (int rc = launch_generic<T, EPI_STORE, false>(d, act, st)) return rc;
  // dw2[e] = sum_g dy[g]^T @ a[g]        [E,M,I]: A(m=m, k=r) = dy[g][r][m]; B(k=r, n=i) = a[g][r][i]
  d.Mext = M;
  d.A = dy; d.a_g = (long)R * M; d.a_m = 1; d.a_k = M;
  d.B = a; d.b_e = (long)R * I; d.b_k = I; d.b_n = 1;
  d.C = dw2; d.c_g = (long)M * I; d.c_m = I; d.N = I; d.K = R;
  if (int rc = launch_generic<T, EPI_STORE, true>(d, act, st)) return rc;
  // dw1[e] = sum_g dz[g]^T @ x[g]        [E,I,M]
  d.Mext = I;
  d.A = dz; d.a_g = (long)R * I; d.a_m = 1; d.a_k = I;
  d.B = x; d.b_e = (long)R * M; d.b_k = M; d.b_n = 1;
  d.C = dw1; d.c_g = (long)I * M; d.c_m = M; d.N = M; d.K = R;
  if (int rc = launch_generic<T, EPI_STORE, true>(d, act, st)) return rc;
  // bias grads
  hipLaunchKernelGGL((bias_grad_kernel<T>), dim3(ceil_div(I, 256), E_local), dim3(256), 0, st, (const T*)dz, (T*)db1,
                     rows, G, E_local, R, I);
  hipLaunchKernelGGL((bias_grad_kernel<T>), dim3(ceil_div(M, 256), E_local), dim3(256), 0, st, (const T*)dy, (T*)db2,
                     rows, G, E_local, R, M);
  EVE_CHECK_LAUNCH("bias_grad_kernel");
  return EVE_OK;
}

static int check_ffn(const char* who, int G, int E_local, int R, int M, int I, int act, int dtype) {
  if (dtype != EVE_F32 && dtype != EVE_BF16) { set_error("%s: bad dtype %d", who, dtype); return EVE_ERR_ARG; }
  if (G <= 0 || E_local <= 0 || G % E_local != 0 || R <= 0 || M <= 0 || I <= 0) {
    set_error("%s: bad shape G=%d E_local=%d R=%d M=%d I=%d", who, G, E_local, R, M, I);
    return EVE_ERR_ARG;
  }
  if (act < EVE_ACT_QUICK_GELU || act > EVE_ACT_IDENTITY) { set_error("%s: unknown activation %d", who, act); return EVE_ERR_ARG; }
  return EVE_OK;
}

}  // namespace eve

using namespace eve;

extern "C" int eve_ffn_fwd(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z,
                           void* a, void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act,
                           int dtype, int flags, void* stream) {
  EVE_REQUIRE(x && w1 && b1 && w2 && b2 && a && y, EVE_ERR_ARG, "eve_ffn_fwd: null pointer");
  if (int rc = check_ffn("eve_ffn_fwd", G, E_local, R, M, I, act, dtype)) return rc;
  EVE_REQUIRE((flags & ~EVE_FFN_SHARED_GPU) == 0, EVE_ERR_ARG, "eve_ffn_fwd: unknown flags 0x%x", flags);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == EVE_BF16) {
    if (ffn_mfma_supported(M, I))
      return ffn_fwd_mfma(x, w1, b1, w2, b2, z, a, y, rows, G, E_local, R, M, I, act, (flags & EVE_FFN_SHARED_GPU) != 0, st);
    return ffn_fwd_generic<bf16_t>(x, w1, b1, w2, b2, z, a, y, rows, G, E_local, R, M, I, act, st);
  }
  return ffn_fwd_generic<float>(x, w1, b1, w2, b2, z, a, y, rows, G, E_local, R, M, I, act, st);
}

extern "C" size_t eve_ffn_bwd_workspace_bytes(int G, int E_local, int R, int M, int I, int dtype) {
  if (G <= 0 || E_local <= 0 || G % E_local != 0 || R <= 0 || dtype != EVE_BF16 || !ffn_mfma_supported(M, I)) return 0;
  return ffn_bwd_mfma_workspace_bytes(G, E_local, R, M, I);
}

extern "C" int eve_ffn_bwd(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2,
                           void* dz, void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G,
                           int E_local, int R, int M, int I, int act, int dtype, int flags, void* workspace,
                           size_t workspace_bytes, void* stream) {
  EVE_REQUIRE(dy && x && z && a && w1 && w2 && dz && dx && dw1 && db1 && dw2 && db2, EVE_ERR_ARG, "eve_ffn_bwd: null pointer");
  if (int rc = check_ffn("eve_ffn_bwd", G, E_local, R, M, I, act, dtype)) return rc;
  EVE_REQUIRE((flags & ~EVE_FFN_SHARED_GPU) == 0, EVE_ERR_ARG, "eve_ffn_bwd: unknown flags 0x%x", flags);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == EVE_BF16) {
    if (ffn_mfma_supported(M, I))
      return ffn_bwd_mfma(dy, x, z, a, w1, w2, dz, dx, dw1, db1, dw2, db2, rows, G, E_local, R, M, I, act,
                          (flags & EVE_FFN_SHARED_GPU) != 0, workspace, workspace_bytes, st);
    return ffn_bwd_generic<bf16_t>(dy, x, z, a, w1, w2, dz, dx, dw1, db1, dw2, db2, rows, G, E_local, R, M, I, act, st);
  }
  return ffn_bwd_generic<float>(dy, x, z, a, w1, w2, dz, dx, dw1, db1, dw2, db2, rows, G, E_local, R, M, I, act, st);
}
