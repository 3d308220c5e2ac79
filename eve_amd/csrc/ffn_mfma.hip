// bf16 MFMA grouped GEMMs for the expert FFN (gfx950, v_mfma_f32_32x32x16_bf16).
//
// One kernel template covers the six GEMMs of an FFN forward + backward:
//   forward   z = x  @ w1^T, y  = a  @ w2^T      A row-major [rows,K], B row-major [N,K]
//   dgrad     da = dy @ w2 , dx = dz @ w1        A row-major [rows,K], B k-major  [K,N]
//   wgrad     dw2 = dy^T @ a, dw1 = dz^T @ x     A k-major  [K,M]  , B k-major  [K,N]
// "k-major" operands (the contraction index is the slow, row index in memory) are staged in
// LDS as they lie in memory and transposed on the way to the MFMA by ds_read_b64_tr_b16.
//
// Block = 256 threads (4 waves, 2x2), tile 128x128x64, each wave a 64x64 output (2x2 MFMA
// tiles of 32x32).  Global -> registers -> LDS staging with the next tile's loads in flight
// under the current tile's MFMAs, two LDS stages, one barrier per K step.  LDS images are
// XOR-swizzled per 16-byte chunk so that both ds_read_b128 (row-major tiles) and the
// transposed reads (k-major tiles) are bank-conflict free.  The epilogue goes through LDS
// (fp32) so that global stores are 16 bytes per lane along rows, and fuses bias, activation,
// activation derivative and -- in the weight-gradient kernels -- the bias gradient.
#include "ffn_mfma.h"

#include "act.h"

namespace eve {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { MEPI_STORE = 0, MEPI_BIAS = 1, MEPI_BIAS_ACT = 2, MEPI_DACT = 3 };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB, same for both tile kinds
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int EPI_LD = 68;                    // floats per row of a wave's 64x64 epilogue image
constexpr int EPI_BYTES = 4 * 64 * EPI_LD * 4;
constexpr int LDS_BYTES = EPI_BYTES > 2 * STAGE_BYTES ? EPI_BYTES : 2 * STAGE_BYTES;

struct MfmaDesc {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  bf16_t* C2;
  const bf16_t* bias;
  const bf16_t* Zin;
  bf16_t* dbias;
  long a_g, b_g, c_g;  // element strides per group (A; B when WGRAD) / per expert (B, C when WGRAD)
  int lda, ldb, ldc;
  int Mext, N, K;
  int G, E_local;
  const int32_t* rows;
};

// ---- LDS images -------------------------------------------------------------------
// row-major tile: [128 rows][64 k] bf16, 128 B per row, 8 chunks of 16 B
__device__ __forceinline__ int row_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// k-major tile: [64 k][128 cols] bf16, 256 B per row, 16 chunks of 16 B
__device__ __forceinline__ int col_tile_off(int k, int chunk) { return k * 256 + ((chunk ^ ((k & 3) << 2)) << 4); }

template <bool COL>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ origin, int ld, int rows_valid, uint4 (&regs)[4]) {
  constexpr int CPR = COL ? 16 : 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int row = idx / CPR, c = idx % CPR;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < rows_valid) v = *reinterpret_cast<const uint4*>(origin + (size_t)row * ld + c * 8);
    regs[i] = v;
  }
}

template <bool COL>
__device__ __forceinline__ void tile_store(char* tile, const uint4 (&regs)[4]) {
  constexpr int CPR = COL ? 16 : 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int row = idx / CPR, c = idx % CPR;
    const int off = COL ? col_tile_off(row, c) : row_tile_off(row, c);
    *reinterpret_cast<uint4*>(tile + off) = regs[i];
  }
}

// Fragment of the 32-row (or 32-column) sub-tile starting at r0, K sub-step ks (16 deep).
template <bool COL>
__device__ __forceinline__ bf16x8 frag_load(const char* tile, int r0, int ks, int lane) {
  if constexpr (!COL) {
    const int row = r0 + (lane & 31);
    const int c = ks * 2 + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(tile + row_tile_off(row, c));
  } else {
    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(col) block; lane 4q+p of the group gives the
    // address of k-row q, columns 4p..4p+3; lane i receives column i, k-rows 0..3.
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = lane >> 5;
    const int e = r0 + 16 * (g & 1) + 4 * p;
    s16x4 v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = ks * 16 + 8 * h + 4 * r + q;
      const int off = col_tile_off(k, e >> 3) + ((e & 7) << 1);
      v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) char*)(tile + off));
    }
    s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
  }
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t w, int hi) {
  return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16_t a = static_cast<bf16_t>(lo), b = static_cast<bf16_t>(hi);
  return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

template <bool A_COL, bool B_COL, int EPI, int ACT, bool WGRAD, bool BIAS_GRAD>
__global__ __launch_bounds__(256, 2) void grouped_gemm_mfma(MfmaDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int z = blockIdx.z;

  int mext = d.Mext;
  if (!WGRAD && d.rows) mext = min(mext, d.rows[z]);
  if (m0 >= mext) return;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const bool do_bias_grad = BIAS_GRAD && blockIdx.x == 0;

  const int nseg = WGRAD ? d.G / d.E_local : 1;
  for (int seg = 0; seg < nseg; ++seg) {
    const int g = WGRAD ? seg * d.E_local + z : z;
    const int e = WGRAD ? z : z % d.E_local;
    int kext = d.K;
    if (WGRAD && d.rows) kext = min(kext, d.rows[g]);
    const int nk = (kext + BK - 1) / BK;
    if (nk == 0) continue;
    const bf16_t* Ag = d.A + (size_t)g * d.a_g;
    const bf16_t* Bg = d.B + (size_t)(WGRAD ? g : e) * d.b_g;
    // tile origin and number of valid LDS rows for K step t
    auto a_origin = [&](int t) { return A_COL ? Ag + (size_t)(t * BK) * d.lda + m0 : Ag + (size_t)m0 * d.lda + t * BK; };
    auto b_origin = [&](int t) { return B_COL ? Bg + (size_t)(t * BK) * d.ldb + n0 : Bg + (size_t)n0 * d.ldb + t * BK; };
    auto a_valid = [&](int t) { return A_COL ? kext - t * BK : mext - m0; };
    auto b_valid = [&](int t) { return B_COL ? kext - t * BK : BN; };

    uint4 ra[4], rb[4];
    tile_load<A_COL>(a_origin(0), d.lda, a_valid(0), ra);
    tile_load<B_COL>(b_origin(0), d.ldb, b_valid(0), rb);
    __syncthreads();  // previous segment's / nothing's readers are done with stage 0
    tile_store<A_COL>(smem, ra);
    tile_store<B_COL>(smem + TILE_BYTES, rb);
    if (do_bias_grad) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t* w = &ra[i].x;
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += bf16_bits_to_f32(w[j >> 1], j & 1);
      }
    }
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
      const char* sa = smem + (t & 1) * STAGE_BYTES;
      const char* sb = sa + TILE_BYTES;
      const bool more = t + 1 < nk;
      if (more) {
        tile_load<A_COL>(a_origin(t + 1), d.lda, a_valid(t + 1), ra);
        tile_load<B_COL>(b_origin(t + 1), d.ldb, b_valid(t + 1), rb);
      }
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag_load<A_COL>(sa, wm * 64 + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[j] = frag_load<B_COL>(sb, wn * 64 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        char* na = smem + ((t + 1) & 1) * STAGE_BYTES;
        tile_store<A_COL>(na, ra);
        tile_store<B_COL>(na + TILE_BYTES, rb);
        if (do_bias_grad) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t* w = &ra[i].x;
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += bf16_bits_to_f32(w[j >> 1], j & 1);
          }
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ epilogue
  // (the K loop ended with a barrier: nobody reads the staging buffers any more)
  const int e_out = WGRAD ? z : z % d.E_local;
  float* ep = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
  {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float bv = 0.f;
      if constexpr (EPI == MEPI_BIAS || EPI == MEPI_BIAS_ACT)
        bv = to_f32<bf16_t>(d.bias[(size_t)e_out * d.N + n0 + wn * 64 + j * 32 + c]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          ep[row * EPI_LD + j * 32 + c] = acc[i][j][r] + bv;
        }
    }
  }
  __syncthreads();
  bf16_t* Cg = d.C ? d.C + (size_t)z * d.c_g : nullptr;
  bf16_t* C2g = d.C2 ? d.C2 + (size_t)z * d.c_g : nullptr;
  const bf16_t* Zg = d.Zin ? d.Zin + (size_t)z * d.c_g : nullptr;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int item = it * 64 + lane;
    const int row = item >> 3, ch = item & 7;
    const int m = m0 + wm * 64 + row;
    if (m >= mext) continue;
    const int n = n0 + wn * 64 + ch * 8;
    const float4 lo = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ch * 8);
    const float4 hi = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ch * 8 + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const size_t o = (size_t)m * d.ldc + n;
    if constexpr (EPI == MEPI_BIAS_ACT) {
      if (Cg) *reinterpret_cast<uint4*>(Cg + o) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_fwd<ACT, true>(v[j]);
      *reinterpret_cast<uint4*>(C2g + o) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    } else {
      if constexpr (EPI == MEPI_DACT) {
        const uint4 zz = *reinterpret_cast<const uint4*>(Zg + o);
        const uint32_t* w = &zz.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= act_bwd<ACT, true>(bf16_bits_to_f32(w[j >> 1], j & 1));
      }
      *reinterpret_cast<uint4*>(Cg + o) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
  }

  if constexpr (BIAS_GRAD) {
    if (do_bias_grad) {  // block-uniform
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);  // [16 row groups][128 columns]
      const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) red[rg * 128 + c * 8 + j] = bsum[j];
      __syncthreads();
      if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * 128 + threadIdx.x];
        d.dbias[(size_t)z * d.Mext + m0 + threadIdx.x] = static_cast<bf16_t>(s);
      }
    }
  }
}

bool ffn_mfma_supported(int M, int I) { return M % 128 == 0 && I % 128 == 0; }

template <bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BIAS_GRAD>
static int launch(const MfmaDesc& d, int act, hipStream_t st) {
  dim3 grid(d.N / BN, ceil_div(d.Mext, BM), WGRAD ? d.E_local : d.G);
#define EVE_MFMA_ACT(A)                                                                                              \
  case A:                                                                                                            \
    hipLaunchKernelGGL((grouped_gemm_mfma<A_COL, B_COL, EPI, A, WGRAD, BIAS_GRAD>), grid, dim3(256), LDS_BYTES, st, d); \
    break;
  if constexpr (EPI == MEPI_BIAS_ACT || EPI == MEPI_DACT) {
    switch (act) {
      EVE_MFMA_ACT(EVE_ACT_QUICK_GELU)
      EVE_MFMA_ACT(EVE_ACT_GELU)
      EVE_MFMA_ACT(EVE_ACT_RELU)
      EVE_MFMA_ACT(EVE_ACT_IDENTITY)
      default:
        set_error("ffn_mfma: unknown activation %d", act);
        return EVE_ERR_ARG;
    }
  } else {
    hipLaunchKernelGGL((grouped_gemm_mfma<A_COL, B_COL, EPI, EVE_ACT_IDENTITY, WGRAD, BIAS_GRAD>), grid, dim3(256),
                       LDS_BYTES, st, d);
  }
#undef EVE_MFMA_ACT
  EVE_CHECK_LAUNCH("grouped_gemm_mfma");
  return EVE_OK;
}

int ffn_fwd_mfma(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z, void* a,
                 void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act, hipStream_t st) {
  MfmaDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows; d.Mext = R;
  // z = x @ w1^T + b1 ; a = act(z)
  d.A = (const bf16_t*)x; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)w1; d.b_g = (long)I * M; d.ldb = M;
  d.C = (bf16_t*)z; d.C2 = (bf16_t*)a; d.c_g = (long)R * I; d.ldc = I; d.bias = (const bf16_t*)b1; d.N = I; d.K = M;
  if (int rc = launch<false, false, MEPI_BIAS_ACT, false, false>(d, act, st)) return rc;
  // y = a @ w2^T + b2
  d.A = (const bf16_t*)a; d.a_g = (long)R * I; d.lda = I;
  d.B = (const bf16_t*)w2; d.b_g = (long)M * I; d.ldb = I;
  d.C = (bf16_t*)y; d.C2 = nullptr; d.c_g = (long)R * M; d.ldc = M; d.bias = (const bf16_t*)b2; d.N = M; d.K = I;
  return launch<false, false, MEPI_BIAS, false, false>(d, act, st);
}

int ffn_bwd_mfma(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2, void* dz,
                 void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G, int E_local, int R,
                 int M, int I, int act, hipStream_t st) {
  MfmaDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows;
  // dz = (dy @ w2) * act'(z): A = dy [R,M] row-major, B = w2 [M,I] k-major
  d.Mext = R;
  d.A = (const bf16_t*)dy; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)w2; d.b_g = (long)M * I; d.ldb = I;
  d.C = (bf16_t*)dz; d.c_g = (long)R * I; d.ldc = I; d.Zin = (const bf16_t*)z; d.N = I; d.K = M;
  if (int rc = launch<false, true, MEPI_DACT, false, false>(d, act, st)) return rc;
  // dx = dz @ w1: A = dz [R,I] row-major, B = w1 [I,M] k-major
  d.A = (const bf16_t*)dz; d.a_g = (long)R * I; d.lda = I;
  d.B = (const bf16_t*)w1; d.b_g = (long)I * M; d.ldb = M;
  d.C = (bf16_t*)dx; d.c_g = (long)R * M; d.ldc = M; d.Zin = nullptr; d.N = M; d.K = I;
  if (int rc = launch<false, true, MEPI_STORE, false, false>(d, act, st)) return rc;
  // dw2[e] = sum_g dy[g]^T @ a[g]  ([M,I]);  db2[e] = column sums of dy
  d.Mext = M;
  d.A = (const bf16_t*)dy; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)a; d.b_g = (long)R * I; d.ldb = I;
  d.C = (bf16_t*)dw2; d.c_g = (long)M * I; d.ldc = I; d.N = I; d.K = R; d.dbias = (bf16_t*)db2;
  if (int rc = launch<true, true, MEPI_STORE, true, true>(d, act, st)) return rc;
  // dw1[e] = sum_g dz[g]^T @ x[g]  ([I,M]);  db1[e] = column sums of dz
  d.Mext = I;
  d.A = (const bf16_t*)dz; d.a_g = (long)R * I; d.lda = I;
  d.B = (const bf16_t*)x; d.b_g = (long)R * M; d.ldb = M;
  d.C = (bf16_t*)dw1; d.c_g = (long)I * M; d.ldc = M; d.N = M; d.K = R; d.dbias = (bf16_t*)db1;
  return launch<true, true, MEPI_STORE, true, true>(d, act, st);
}

}  // namespace eve
