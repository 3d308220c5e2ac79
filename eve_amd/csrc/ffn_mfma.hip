// bf16 MFMA grouped GEMMs for the expert FFN (gfx950, v_mfma_f32_32x32x16_bf16).
//
// One kernel template covers the six GEMMs of an FFN forward + backward:
//   forward   z = x  @ w1^T, y  = a  @ w2^T      A row-major [rows,K], B row-major [N,K]
//   dgrad     da = dy @ w2 , dx = dz @ w1        A row-major [rows,K], B k-major  [K,N]
//   wgrad     dw2 = dy^T @ a, dw1 = dz^T @ x     A k-major  [K,M]  , B k-major  [K,N]
// "k-major" operands (the contraction index is the slow, row index in memory) are staged in
// LDS as they lie in memory and transposed on the way to the MFMA by ds_read_b64_tr_b16.
//
// Block tile BM x 128 x 64 with BM = 256 (8 waves, 4x2) or 128 (4 waves, 2x2); each wave owns a
// 64x64 output (2x2 MFMA tiles of 32x32).  Tiles travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), NS stages deep: NS-1 tiles are in flight while one is being
// multiplied, waited for with a COUNTED vmcnt and a raw s_barrier so the younger DMAs stay in
// flight across the barrier.  LDS images are XOR-swizzled per 16-byte chunk -- applied to the
// DMA's per-lane SOURCE address, since an LDS-DMA writes lane-linear -- so that ds_read_b128
// (row-major tiles) and the transposed reads (k-major tiles) are bank-conflict free.  The
// epilogue goes through LDS (fp32) so that global stores are 16 bytes per lane along rows, and
// fuses bias, activation, activation derivative and -- in the weight-gradient kernels -- the
// bias gradient (column sums of the k-major A tile).
#include <atomic>
#include "ffn_mfma.h"

#include <cstdlib>
#include <type_traits>

#include "act.h"

namespace eve {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { MEPI_STORE = 0, MEPI_BIAS = 1, MEPI_BIAS_ACT = 2, MEPI_DACT = 3, MEPI_STORE_F32 = 4 };

// MFMA shape of the ping-pong kernels: 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16 (same flops
// per cycle and the same fragment traffic; the chip holds a higher clock with the 16x16 shape under load)
#ifndef EVE_MFMA16
#define EVE_MFMA16 0
#endif
constexpr bool kM16 = EVE_MFMA16 != 0;
constexpr int kTR = kM16 ? 16 : 32;   // edge of one MFMA output tile
constexpr int kKS = kM16 ? 32 : 16;   // K depth of one MFMA

constexpr int BN = 256, BK = 64;
constexpr int WTM = 4, WTN = 2;  // MFMA tiles per wave: 128 x 64 outputs
constexpr int EPI_LD = 68;       // floats per row of a wave's 32x64 epilogue image

template <int BM, int NS> struct Cfg {
  static constexpr int kWaves = (BM / 128) * (BN / 64);  // waves as (BM/128) x 4, 128x64 outputs each
  static constexpr int kThreads = kWaves * 64;
  static constexpr int kATile = BM * BK * 2;              // bytes
  static constexpr int kBTile = BN * BK * 2;
  static constexpr int kStage = kATile + kBTile;
  static constexpr int kEpi = kWaves * 32 * EPI_LD * 4;
  static constexpr int kLds = NS * kStage > kEpi ? NS * kStage : kEpi;
  static constexpr int kDmaA = kATile / 1024 / kWaves;   // DMA instructions per wave per tile
  static constexpr int kDmaB = kBTile / 1024 / kWaves;
};

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
  // weight-gradient split-K (ksplit > 1): split j of expert e accumulates a slice of the K (row) range into
  // fp32 partials Cpart[j][e][Mext][N] / dbias_part[j][e][Mext]; wgrad_reduce_kernel sums the splits in order
  int ksplit;
  float* Cpart;
  float* dbias_part;
  // MEPI_DACT launches: column sums of the OUTPUT (dz) for the bias gradient db1, taken in the epilogue while the
  // values are in registers: fp32 partials colsum[(z * colsum_slots + slot) * N + n], slot = wave row block of the
  // group (m0 / rows-per-wave + wave row); bias_grads_kernel adds them up in a fixed order.  nullptr = off.
  float* colsum;
  int colsum_slots;
  // forward / dgrad launches with TWO groups of unequal height (the modality split: 12 608 vision rows beside
  // 2 560 text rows): tiles are enumerated over the rows that exist (tile_of_block) instead of over
  // G x ceil(Mext / BM) slots, most of which would be empty for the short group and leave its XCDs idle
  int compact;
  // tile order inside a group: 0 = row tile fastest (consecutive tiles share their B column panel), 1 = column tile
  // fastest (they share their A row panel).  The panel to share is the one of the LARGER operand: the smaller one
  // stays in the XCD's L2 anyway.  dw1 = dz^T x has A = dz [rows, I] four times the size of B = x [rows, M]: walked
  // row tile fastest, its twelve A panels (5.8 MB per expert, more than the 4 MiB L2) were fetched three times over.
  int nfast;
  // weight-gradient launches of such groups (one K segment per expert, rows[] given): expert e's K range is cut
  // into ceil(rows[e] / L) equal slices, L = ceil(sum of rows / kslots) -- at most kslots + E_local slices in all,
  // numbered expert-major -- each with an fp32 partial Cpart[slice][Mext][N]; wgrad_reduce_dyn_kernel adds an
  // expert's slices in order.  0 = off (ksplit decides).
  int kslots;
#ifdef EVE_PP_STAMP
  unsigned long long* dbg;  // diagnostic build (tools/pp_stamps.hip): [block 0 / 1000][wave 8][64] cycle stamps (low 32 bits)
  int same_tile;            // diagnostic: every K step fetches K tile 0 (everything hits in L2)
#endif
};

struct MfmaPair {
  MfmaDesc d[2];
  int nobias;  // weight-gradient launch: no tile takes the bias gradient (it is formed outside, see ffn_bwd_mfma)
  int n0;      // blocks (tiles) of d[0]; a multiple of 8 (the XCD-aware tile order counts from each problem's first block)
  int ntotal;  // tiles of both problems (persistent kernel: the grid is smaller and walks them)
};


// ---- block id -> tile ------------------------------------------------------------------------------------
// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so block b
// and b+8 share an L2.  Every XCD walks a contiguous run of (group, tile) items -- whole groups when there are
// at least 8 of them (sorted expert-major, so groups that share weights sit on one XCD), a slice of a group's
// tiles when there are fewer -- and, inside a group, its tiles with the row tile varying fastest: the ~32 tiles
// an XCD runs at a time then share their A row panels and B column panels through its L2 instead of each
// streaming them from HBM.  (Speed only: correctness does not depend on where a block runs.)
struct TileCoord { int z, split, m0, n0, mext; };

__device__ __forceinline__ int dyn_slice_rows(const MfmaDesc& d) {
  int total = 0;
  for (int e = 0; e < d.E_local; ++e) total += min(d.K, d.rows[e]);
  const int L = (total + d.kslots - 1) / d.kslots;
  return max(64, (L + 63) & ~63);
}
__device__ __forceinline__ int dyn_nslices(int rows_e, int L) { return max(1, (rows_e + L - 1) / L); }

// Forward / dgrad launches (and weight-gradient launches without dynamic slices): false = the block has no tile.
// CMP: the compact enumeration (MfmaDesc::compact) -- a kernel variant of its own, so that the loops over rows[] do
// not cost the many-group kernels registers.
template <int BM, bool WGRAD, bool CMP = false>
__device__ __forceinline__ bool tile_of_block(const MfmaDesc& d, unsigned bid, TileCoord& T) {
  const int tiles_n = d.N / BN;
  const int xcd = bid & 7, local = bid >> 3;
  T.split = 0;
  if constexpr (!WGRAD && CMP) {  // exactly two groups (closed form: no loop over rows[], two scalar loads)
    const int mext0 = min(d.Mext, d.rows[0]), mext1 = min(d.Mext, d.rows[1]);
    const int tm0 = (mext0 + BM - 1) / BM, tm1 = (mext1 + BM - 1) / BM;
    const int cnt0 = tm0 * tiles_n, total = cnt0 + tm1 * tiles_n;
    const int per_xcd = (total + 7) >> 3;
    const int item = xcd * per_xcd + local;
    if (local >= per_xcd || item >= total) return false;
    const bool g1 = item >= cnt0;
    const int it = g1 ? item - cnt0 : item, tm = g1 ? tm1 : tm0;
    T.z = g1 ? 1 : 0; T.m0 = (it % tm) * BM; T.n0 = (it / tm) * BN; T.mext = g1 ? mext1 : mext0;
    return true;
  }
  const int tiles_m = (d.Mext + BM - 1) / BM;
  const int tiles_per_group = tiles_m * tiles_n;
  const int nsplit = WGRAD ? max(d.ksplit, 1) : 1;
  const int ngroups = WGRAD ? d.E_local * nsplit : d.G;
  const int per_xcd = (ngroups * tiles_per_group + 7) >> 3;
  const int item = xcd * per_xcd + local;
  if (local >= per_xcd || item >= ngroups * tiles_per_group) return false;
  const int p = item / tiles_per_group;
  const int tile = item % tiles_per_group;
  if (WGRAD) {
    T.z = p % d.E_local;   // p = split * E_local + expert
    T.split = p / d.E_local;
  } else {
    const int ep = d.G / d.E_local;
    T.z = (p % ep) * d.E_local + p / ep;  // p enumerates (expert, source) pairs expert-major
  }
  if (WGRAD && d.nfast) { T.n0 = (tile % tiles_n) * BN; T.m0 = (tile / tiles_n) * BM; }
  else { T.m0 = (tile % tiles_m) * BM; T.n0 = (tile / tiles_m) * BN; }
  T.mext = d.Mext;
  if (!WGRAD && d.rows) T.mext = min(T.mext, d.rows[T.z]);
  return T.m0 < T.mext;
}

// In-kernel cycle stamps, compiled only into the diagnostic build; they go to a buffer of their own.
#ifdef EVE_PP_STAMP
// stamp k of K tile t goes to lane 5t+k of one VGPR (no memory traffic inside the loop); tiles 0..11
#define EVE_STAMP(k)                                                                                     \
  do {                                                                                                   \
    if (t < 10) {                                                                                        \
      const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime();                                        \
      stamp_v = lane == t * 6 + (k) ? c_ : stamp_v;                                                      \
    }                                                                                                    \
  } while (0)
#else
#define EVE_STAMP(k) do { } while (0)
#endif

// ---- LDS images -------------------------------------------------------------------
// row-major tile: [W rows][64 k] bf16, 128 B per row, 8 chunks of 16 B
__device__ __forceinline__ int row_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// k-major tile: [64 k][W cols] bf16, 2W B per row, W/8 chunks of 16 B
template <int W> __device__ __forceinline__ int col_tile_off(int k, int chunk) {
  return k * (2 * W) + ((chunk ^ ((k & 3) << 2)) << 4);
}

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins below: the host pass sees an empty kernel body
// Direct global -> LDS copy of one tile by LDS-DMA (buffer_load_dwordx4 ... lds: 16 B per lane,
// 1 KiB per wave instruction).  The LDS image of one instruction is lane-linear, so the chunk swizzle
// is applied to the per-lane SOURCE offset.  Loads go through a buffer descriptor whose extent is
// the valid part of the operand: lanes that fall outside it read zeros, which is what zero-fills
// the rows past the end of a weight-gradient segment (and keeps every access inside the tensor).
// voff[i] is the loop-invariant per-lane byte offset of DMA i inside a tile (tile_dma_offsets).
template <bool COL, int W, int WAVES>
__device__ __forceinline__ void tile_dma_offsets(int ld, int wave, int lane, uint32_t (&voff)[W * BK * 2 / 1024 / WAVES]) {
  constexpr int PIECES = W * BK * 2 / 1024;
#pragma unroll
  for (int i = 0; i < PIECES / WAVES; ++i) {
    const int P = (i * WAVES + wave) * 64 + lane;
    int row, c;
    if constexpr (COL) {
      constexpr int CPR = W / 8;
      row = P / CPR;
      c = (P % CPR) ^ ((row & 3) << 2);
    } else {
      row = P >> 3;
      c = (P & 7) ^ ((row >> 1) & 7);
    }
    voff[i] = ((uint32_t)row * (uint32_t)ld + (uint32_t)c * 8u) * 2u;
  }
}

template <int W, int WAVES>
__device__ __forceinline__ void tile_dma(__amdgpu_buffer_rsrc_t rsrc, uint32_t origin_bytes,
                                         const uint32_t (&voff)[W * BK * 2 / 1024 / WAVES], char* tile, int wave) {
  constexpr int PIECES = W * BK * 2 / 1024;
#pragma unroll
  for (int i = 0; i < PIECES / WAVES; ++i) {
    const int piece = i * WAVES + wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + piece * 1024), 16,
                                             voff[i] + origin_bytes, 0, 0, 0);
  }
}

#endif  // __HIP_DEVICE_COMPILE__

// Fragment of the 32-row (or 32-column) sub-tile starting at r0, K sub-step ks (16 deep).
template <bool COL, int W>
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
      const int off = col_tile_off<W>(k, e >> 3) + ((e & 7) << 1);
      v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) char*)(tile + off));
    }
    s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
  }
}

// Same for v_mfma_f32_16x16x32_bf16: 16-row (or 16-column) sub-tile starting at r0, K sub-step ks (32 deep);
// lane l supplies row / column l & 15 and k = 8 (l >> 4) .. + 8.
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <bool M16> using acc_tile_t = typename std::conditional<M16, f32x4, f32x16>::type;
template <bool COL, int W>
__device__ __forceinline__ bf16x8 frag_load16(const char* tile, int r0, int ks, int lane) {
  if constexpr (!COL) {
    const int row = r0 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row_tile_off(row, c));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int e = r0 + 4 * p;
    s16x4 v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = ks * 32 + 8 * g + 4 * r + q;
      const int off = col_tile_off<W>(k, e >> 3) + ((e & 7) << 1);
      v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) char*)(tile + off));
    }
    s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
  }
}

// ---- transposed fragment reads of the ping-pong kernels ---------------------------------------------------
// ds_read_b64_tr_b16 as inline asm.  Through the builtin the compiler's waitcnt pass puts an s_waitcnt
// vmcnt(0) in front of every such read that follows an LDS-DMA in program order (it cannot see that the
// DMA targets another ring slot), which forbids issuing the fetch of a phase before or between its reads;
// an asm read carries no such dependence, so the fetch is interleaved with the reads and both latencies
// overlap.  The price: the compiler does not count these reads in lgkmcnt, so every user ends its read
// phase with `s_waitcnt lgkmcnt(0)` followed by frag_fence() on the registers before their first use.
// Address of a lane inside the tile for the 32-column sub-tile at r0 (see frag_load): k-row 8h + q of K
// sub-step 0, columns from 4p; K sub-step ks and the second half of the fragment are immediate offsets.
template <int W>
__device__ __forceinline__ uint32_t tr_lane_offset(int r0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = lane >> 5;
  const int e = r0 + 16 * (g & 1) + 4 * p;
  return (uint32_t)(col_tile_off<W>(8 * h + q, e >> 3) + ((e & 7) << 1));
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) {
  return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}
template <int W>
__device__ __forceinline__ bf16x8 frag_load_tr(uint32_t addr, int ks) {
#if defined(__HIP_DEVICE_COMPILE__)
  s16x4 v0, v1;
#define EVE_TR_CASE(n)                                                                                       \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v0) : "v"(addr), "n"((n) * 16 * 2 * W));          \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v1) : "v"(addr), "n"(((n) * 16 + 4) * 2 * W));
  switch (ks) {  // ks is a constant after unrolling; the offset must be one for the assembler
    case 0: EVE_TR_CASE(0) break;
    case 1: EVE_TR_CASE(1) break;
    case 2: EVE_TR_CASE(2) break;
    default: EVE_TR_CASE(3) break;
  }
#undef EVE_TR_CASE
  s16x8 w = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, w);
#else
  return bf16x8{};
#endif
}
// orders the first use of asm-loaded fragments after the s_waitcnt that precedes this call
template <int NA, int NB, int KI>
__device__ __forceinline__ void frag_fence(bf16x8 (&fa)[KI][NA], bf16x8 (&fb)[KI][NB]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int ks = 0; ks < KI; ++ks) {
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(fa[ks][i]));
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(fb[ks][j]));
  }
#endif
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t w, int hi) {
  return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16));
}
// accumulator + bias where the epilogue has one ("x + 0.f" is not a no-op to the compiler: it is -0's sign)
template <int EPI> __device__ __forceinline__ float add_bias(float a, float b) {
  if constexpr (EPI == MEPI_BIAS || EPI == MEPI_BIAS_ACT) return a + b;
  else return a;
}
// one v_cvt_pk_bf16_f32 (round to nearest even); written as a vector conversion so that the compiler does not pair
// the scalar conversions its own way and then shuffle 16-bit halves back into place
typedef __attribute__((ext_vector_type(2))) float pk_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 pk_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const pk_f32x2 t = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, pk_bf16x2));
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// wait until at most `tiles` of this wave's youngest tiles (PER DMA instructions each) are in flight
template <int PER> __device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  if (tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
  else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
  else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Eight consecutive columns of one output row, read back from the staging image as two float4 (accumulator + bias),
// finished two elements at a time on packed fp32 instructions (the activation epilogues are bound by the VALU's
// instruction issue: in-kernel stamps put the conversions / activation of a half slab at the sum of both SIMD
// partners' instruction counts x 4 cycles).
//   MEPI_BIAS_ACT: out2 = a = act(z);  out1 = g = act'(z), what the backward pass needs of z -- saved in its place
//                  (the `z` tensor of ffn_fwd_mfma), so that the MEPI_DACT epilogue is one multiply per element
//                  instead of a second sigmoid (exp + rcp + 6 more instructions per element);
//   MEPI_DACT:     out1 = v * g with g read from the saved tensor (`zw`: eight bf16); `cs` += the unrounded values
//                  of rows inside the extent (column sums for the bias gradient db1);
//   otherwise:     out1 = v.
template <int EPI, int ACT>
__device__ __forceinline__ void epi_row8(const float4& lo, const float4& hi, const uint4& zw, bool want_g, bool in_extent,
                                         f32x2_t (&cs)[4], uint4& out1, uint4& out2) {
  f32x2_t v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
  uint32_t o1[4], o2[4];
  if constexpr (EPI == MEPI_BIAS_ACT) {
    if (want_g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2_t a, g;
        act_fwd_grad2<ACT, true>(v[j], a, g);
        o1[j] = pack_bf16x2(g.x, g.y);
        o2[j] = pack_bf16x2(a.x, a.y);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2_t a, g;
        act_fwd_grad2<ACT, false>(v[j], a, g);
        o1[j] = 0;
        o2[j] = pack_bf16x2(a.x, a.y);
      }
    }
  } else {
    if constexpr (EPI == MEPI_DACT) {
      const uint32_t* w = &zw.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2_t g = {bf16_bits_to_f32(w[j], 0), bf16_bits_to_f32(w[j], 1)};
        v[j] = v[j] * g;
      }
      if (in_extent) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] = cs[j] + v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { o1[j] = pack_bf16x2(v[j].x, v[j].y); o2[j] = 0; }
  }
  out1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
  out2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

// Column sums of a wave's output block for the bias gradient (MEPI_DACT): lane (irow = lane >> 3, ich = lane & 7)
// holds the sums over its rows of columns ncol .. ncol + 7; the eight row lanes are added by a butterfly (fixed
// order) and the lanes of row lane 0 store the wave's 64 sums as this (group, slot)'s partial.
__device__ __forceinline__ void colsum_store(const f32x2_t (&cs2)[4], const MfmaDesc& d, int z, int slot, int ncol, int lane) {
  if (!d.colsum) return;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = (j & 1) ? cs2[j >> 1].y : cs2[j >> 1].x;
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    cs[j] = v;
  }
  if (lane < 8) {
    float* o = d.colsum + ((size_t)z * d.colsum_slots + slot) * d.N + ncol;
    *reinterpret_cast<float4*>(o) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// Epilogue of one block tile.  Per wave, one 32-row slab (MFMA row tile i) at a time through a private
// fp32 LDS image, so that global stores are 16 B per lane along rows.  Only the wave itself touches
// its image; waves are laid out (wave >> 2) x (wave & 3) over 128 x 64 outputs each.
template <int EPI, int ACT, int WR = WTM, bool M16 = false>
__device__ __forceinline__ void tile_epilogue(const acc_tile_t<M16> (&acc)[M16 ? 2 * WR : WR][M16 ? 4 : WTN], char* smem, const MfmaDesc& d, int z, int e_out,
                                              int m0, int n0, int mext, int wave, int lane) {
  const int wm = wave >> 2, wn = wave & 3;
  float* ep = reinterpret_cast<float*>(smem) + wave * 32 * EPI_LD;
  bf16_t* Cg = d.C ? d.C + (size_t)z * d.c_g : nullptr;
  bf16_t* C2g = d.C2 ? d.C2 + (size_t)z * d.c_g : nullptr;
  const bf16_t* Zg = d.Zin ? d.Zin + (size_t)z * d.c_g : nullptr;
  constexpr int NTN = M16 ? 4 : WTN, TW = M16 ? 16 : 32;  // column tiles per wave and their width
  float bv[NTN];
#pragma unroll
  for (int j = 0; j < NTN; ++j) {
    bv[j] = 0.f;
    if constexpr (EPI == MEPI_BIAS || EPI == MEPI_BIAS_ACT)
      bv[j] = to_f32<bf16_t>(d.bias[(size_t)e_out * d.N + n0 + wn * 64 + j * TW + (lane & (TW - 1))]);
  }
  // item `it` of a slab: lane -> (row it*8 + lane/8, 8 columns from (lane%8)*8)
  const int irow = lane >> 3, ich = lane & 7;
  const int ncol = n0 + wn * 64 + ich * 8;
  // DACT reads the saved pre-activation z: the loads of slab i+1 are issued before slab i is worked
  // on, so that their latency is paid once per tile, not once per item.  Rows past the extent are
  // clamped for the load (their results are never stored).
  uint4 zbuf[2][4];
  auto load_z = [&](int i, uint4 (&dst)[4]) {
    if constexpr (EPI == MEPI_DACT) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min(m0 + wm * (WR * 32) + i * 32 + it * 8 + irow, mext - 1);
        dst[it] = *reinterpret_cast<const uint4*>(Zg + (size_t)m * d.ldc + ncol);
      }
    }
  };
  load_z(0, zbuf[0]);
  f32x2_t cs[4];  // DACT: this lane's share of the column sums of the stored values (rows it owns, 8 columns)
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = f32x2_t{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    if (i + 1 < WR) load_z(i + 1, zbuf[(i + 1) & 1]);
    if constexpr (M16) {  // D of 16x16: column lane & 15, rows 4 (lane >> 4) + r
      const int c = lane & 15, g4 = 4 * (lane >> 4);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < NTN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) ep[(ii * 16 + g4 + r) * EPI_LD + j * 16 + c] = add_bias<EPI>(acc[2 * i + ii][j][r], bv[j]);
    } else {
      const int c = lane & 31, h = lane >> 5;
#pragma unroll
      for (int j = 0; j < NTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          ep[row * EPI_LD + j * 32 + c] = add_bias<EPI>(acc[i][j][r], bv[j]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + irow;
      const int m = m0 + wm * (WR * 32) + i * 32 + row;
      const float4 lo = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ich * 8);
      const float4 hi = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ich * 8 + 4);
      const size_t o = (size_t)m * d.ldc + ncol;
      if constexpr (EPI == MEPI_STORE_F32) {
        if (m < mext) {  // fp32 partial of a split-K weight gradient; z = split * E_local + expert
          float* o32 = d.Cpart + (size_t)z * d.c_g + o;
          *reinterpret_cast<float4*>(o32) = lo;
          *reinterpret_cast<float4*>(o32 + 4) = hi;
        }
      } else {
        uint4 o1, o2;
        epi_row8<EPI, ACT>(lo, hi, zbuf[i & 1][it], Cg != nullptr, m < mext, cs, o1, o2);
        if constexpr (EPI == MEPI_BIAS_ACT) {
          if (Cg && m < mext) *reinterpret_cast<uint4*>(Cg + o) = o1;
          if (m < mext) *reinterpret_cast<uint4*>(C2g + o) = o2;
        } else {
          if (m < mext) *reinterpret_cast<uint4*>(Cg + o) = o1;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (EPI == MEPI_DACT) colsum_store(cs, d, z, m0 / (WR * 32) + wm, ncol, lane);
}

// Epilogue of the plain-store launches on TRANSPOSED accumulators.  With the MFMA operands swapped
// (D' = B A^T, i.e. C^T) a lane holds ROW l & 31 of a 32-row slab and, per 32-column tile j and register
// group gq = r >> 2, the four CONSECUTIVE columns 8 gq + 4 (l >> 5) + (r & 3).  The finished values
// (fp32 accumulator + fp32 bias, rounded once) are therefore staged as bf16 with one 8-byte LDS store per four
// values into a row-major [32][64] image (pitch 136 B: the stores are conflict-free), and read back as 16-byte
// row chunks for the same coalesced global stores as tile_epilogue: 8 ds_write_b64 + 8 ds_read_b64 per slab
// instead of 32 ds_write_b32 + 8 ds_read_b128 through an fp32 image -- LDS stores move 64-85 B/clk, and the
// fp32 image was most of the plain epilogue's time.  `img`: this wave's 4 352-byte staging area.
constexpr int kImgPitch = 136;
template <int EPI, int WR>
__device__ __forceinline__ void tile_epilogue_t(const f32x16 (&acc)[WR][WTN], char* img, const MfmaDesc& d, int z, int e_out,
                                                int m0, int n0, int mext, int wave, int lane) {
  static_assert(EPI == MEPI_STORE || EPI == MEPI_BIAS, "plain epilogues only");
  const int wm = wave >> 2, wn = wave & 3, c = lane & 31, hh = lane >> 5;
  bf16_t* Cg = d.C + (size_t)z * d.c_g;
  float bv[WTN][4][4];
#pragma unroll
  for (int j = 0; j < WTN; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      if constexpr (EPI == MEPI_BIAS) {
        const uint2 b = *reinterpret_cast<const uint2*>(d.bias + (size_t)e_out * d.N + n0 + wn * 64 + j * 32 + 8 * gq + 4 * hh);
        bv[j][gq][0] = bf16_bits_to_f32(b.x, 0); bv[j][gq][1] = bf16_bits_to_f32(b.x, 1);
        bv[j][gq][2] = bf16_bits_to_f32(b.y, 0); bv[j][gq][3] = bf16_bits_to_f32(b.y, 1);
      } else {
        bv[j][gq][0] = bv[j][gq][1] = bv[j][gq][2] = bv[j][gq][3] = 0.f;
      }
    }
  const int irow = lane >> 3, ich = lane & 7;
  const int ncol = n0 + wn * 64 + ich * 8;
  char* wr = img + c * kImgPitch + 8 * hh;
#pragma unroll
  for (int i = 0; i < WR; ++i) {
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const uint2 v = make_uint2(pack_bf16x2(add_bias<EPI>(acc[i][j][4 * gq], bv[j][gq][0]),
                                               add_bias<EPI>(acc[i][j][4 * gq + 1], bv[j][gq][1])),
                                   pack_bf16x2(add_bias<EPI>(acc[i][j][4 * gq + 2], bv[j][gq][2]),
                                               add_bias<EPI>(acc[i][j][4 * gq + 3], bv[j][gq][3])));
        *reinterpret_cast<uint2*>(wr + (j * 32 + 8 * gq) * 2) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + irow;
      const int m = m0 + wm * (WR * 32) + i * 32 + row;
      const uint2 lo = *reinterpret_cast<const uint2*>(img + row * kImgPitch + ich * 16);
      const uint2 hi = *reinterpret_cast<const uint2*>(img + row * kImgPitch + ich * 16 + 8);
      if (m < mext) *reinterpret_cast<uint4*>(Cg + (size_t)m * d.ldc + ncol) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// Bias gradient of a weight-gradient tile: reduce the per-thread partial column sums of the k-major
// A tiles (thread t owns the 8 columns of chunk t % (BM/8)) and store BM values.  Block-uniform call.
template <int BM, int THREADS>
__device__ __forceinline__ void bias_grad_store(const float (&bsum)[8], char* smem, const MfmaDesc& d, int z, int m0) {
  __syncthreads();
  constexpr int CPR = BM / 8, RG = THREADS / CPR;  // row groups that share a chunk column
  float* red = reinterpret_cast<float*>(smem);      // [RG][BM]
  const int c = threadIdx.x % CPR, rg = threadIdx.x / CPR;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rg * BM + c * 8 + j] = bsum[j];
  __syncthreads();
  if (threadIdx.x < BM) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) s += red[r * BM + threadIdx.x];
    if (d.ksplit > 1) d.dbias_part[(size_t)z * d.Mext + m0 + threadIdx.x] = s;  // z = split * E_local + expert
    else d.dbias[(size_t)z * d.Mext + m0 + threadIdx.x] = static_cast<bf16_t>(s);
  }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int BM, int NS, bool A_COL, bool B_COL, int EPI, int ACT, bool WGRAD, bool BIAS_GRAD>
__global__ __launch_bounds__((BM / 128) * 256) void grouped_gemm_mfma(MfmaDesc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C_ = Cfg<BM, NS>;
  constexpr int D = NS - 1;  // tiles in flight ahead of the one being multiplied
  static_assert(D >= 1 && D <= 3, "prefetch depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave >> 2, wn = wave & 3;
  TileCoord tc;
  if (!tile_of_block<BM, WGRAD>(d, blockIdx.x, tc)) return;  // XCD-aware tile order
  const int z = tc.z, m0 = tc.m0, n0 = tc.n0, mext = tc.mext;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const bool do_bias_grad = BIAS_GRAD && n0 == 0;

  uint32_t voff_a[C_::kDmaA], voff_b[C_::kDmaB];
  tile_dma_offsets<A_COL, BM, C_::kWaves>(d.lda, wave, lane, voff_a);
  tile_dma_offsets<B_COL, BN, C_::kWaves>(d.ldb, wave, lane, voff_b);

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
    // byte offset of K tile t's origin inside the group's A / the expert's (or group's) B
    auto a_origin = [&](int t) { return (uint32_t)(A_COL ? (t * BK) * d.lda + m0 : m0 * d.lda + t * BK) * 2u; };
    auto b_origin = [&](int t) { return (uint32_t)(B_COL ? (t * BK) * d.ldb + n0 : n0 * d.ldb + t * BK) * 2u; };
    // valid extent of the operands: whole [R, K] group / weight matrix, or the first kext rows of a
    // k-major activation in the weight-gradient kernels (rows past it then read as zero)
    const uint32_t a_bytes = (uint32_t)(A_COL ? (WGRAD ? kext : d.K) * d.lda : d.Mext * d.lda) * 2u;
    const uint32_t b_bytes = (uint32_t)(B_COL ? (WGRAD ? kext : d.K) * d.ldb : d.N * d.ldb) * 2u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, b_bytes, 0x00020000);
    auto stage = [&](int t) {
      char* sa_ = smem + (t % NS) * C_::kStage;
      tile_dma<BM, C_::kWaves>(a_rsrc, a_origin(t), voff_a, sa_, wave_u);
      tile_dma<BN, C_::kWaves>(b_rsrc, b_origin(t), voff_b, sa_ + C_::kATile, wave_u);
    };

    // Software pipeline over half K tiles (32 deep): while the MFMAs of one half run, the LDS reads
    // of the next half are already in flight into the other fragment set, across the barrier too:
    //   read H1(t) | MFMA H0(t) | wait tile t+1, barrier, DMA tile t+D, read H0(t+1) | MFMA H1(t) | ...
    constexpr int KH = BK / 32;  // k16 steps per half
    bf16x8 a0[KH][WTM], b0[KH][WTN];
    bf16x8 a1[WGRAD ? 1 : KH][WGRAD ? 1 : WTM], b1[WGRAD ? 1 : KH][WGRAD ? 1 : WTN];
    auto read_half = [&](const char* sa, bf16x8 (&fa)[KH][WTM], bf16x8 (&fb)[KH][WTN], int half) {
      const char* sb = sa + C_::kATile;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[ks][j] = frag_load<B_COL, BN>(sb, wn * 64 + j * 32, half * KH + ks, lane);
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[ks][i] = frag_load<A_COL, BM>(sa, wm * 128 + i * 32, half * KH + ks, lane);
      }
    };
    auto mfma_half = [&](const bf16x8 (&fa)[KH][WTM], const bf16x8 (&fb)[KH][WTN]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    auto bias_grad_tile = [&](const char* sa) {
      if constexpr (BIAS_GRAD) {
        if (do_bias_grad) {  // column sums of the k-major A tile, read back from LDS
          constexpr int CPR = BM / 8;
#pragma unroll
          for (int i = 0; i < C_::kATile / 16 / C_::kThreads; ++i) {
            const int idx = threadIdx.x + C_::kThreads * i;
            const uint4 v = *reinterpret_cast<const uint4*>(sa + col_tile_off<BM>(idx / CPR, idx % CPR));
            const uint32_t* w = &v.x;
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += bf16_bits_to_f32(w[j >> 1], j & 1);
          }
        }
      }
    };

    __syncthreads();  // a previous segment's readers are done with every stage
    for (int t = 0; t < D && t < nk; ++t) stage(t);
    wait_tiles_in_flight<C_::kDmaA + C_::kDmaB>(min(D - 1, nk - 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile 0 is in LDS
    if (D < nk) stage(D);
    if constexpr (!WGRAD) {
      // The last K tile is peeled off: with it inside the loop the two paths into the second MFMA
      // group join, and the compiler's conservative lgkmcnt waits at the join make that group wait
      // for the fragment reads just issued for the NEXT half tile.
      read_half(smem, a0, b0, 0);
      for (int t = 0; t + 1 < nk; ++t) {
        const char* sa = smem + (t % NS) * C_::kStage;
        read_half(sa, a1, b1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        // tile t+1 must have landed; tiles t+2 .. t+D (where they exist) may stay in flight
        wait_tiles_in_flight<C_::kDmaA + C_::kDmaB>(min(D - 1, nk - 2 - t));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile t are complete
        __builtin_amdgcn_s_barrier();                       // tile t+1 visible; tile t's stage is free
        if (t + 1 + D < nk) stage(t + 1 + D);               // into the stage tile t was read from
        read_half(smem + ((t + 1) % NS) * C_::kStage, a0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      read_half(smem + ((nk - 1) % NS) * C_::kStage, a1, b1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(a0, b0);
      mfma_half(a1, b1);
    } else {
      // weight-gradient kernels: transposed reads need more address registers, so one fragment
      // set only (no register double buffering) to stay inside 256 registers
      for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t % NS) * C_::kStage;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          read_half(sa, a0, b0, half);
          mfma_half(a0, b0);
        }
        bias_grad_tile(sa);
        if (t + 1 < nk) {
          wait_tiles_in_flight<C_::kDmaA + C_::kDmaB>(min(D - 1, nk - 2 - t));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (t + 1 + D < nk) stage(t + 1 + D);
        }
      }
    }
  }
  __syncthreads();  // all fragment reads done before the epilogue reuses LDS

  tile_epilogue<EPI, ACT>(acc, smem, d, z, WGRAD ? z : z % d.E_local, m0, n0, mext, wave, lane);
  if constexpr (BIAS_GRAD) {
    if (do_bias_grad) bias_grad_store<BM, C_::kThreads>(bsum, smem, d, z, m0);  // block-uniform
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------ ping-pong kernel
// 256 x 256 block tile, 8 waves; the two waves of every SIMD (wave w and w + 4) run half a K step
// apart, so that while one of them issues its 32 MFMAs of a K step ("M" phase, operands already in
// registers) its partner issues its LDS-DMAs and fragment reads for the next one ("L" phase): the
// matrix pipe of a SIMD always has one wave feeding it and the other wave's memory instructions,
// which an in-order wave cannot overlap with its own MFMAs, run beside them.  Workgroup barriers
// separate the phases (intervals I_0, I_1, ...):
//     waves 0-3:  L(0) | M(0) | L(1) | M(1) | ...
//     waves 4-7:   -   | L(0) | M(0) | L(1) | M(1) ...
// Operand tiles live in a ring of five 256 x BKI sub-tiles (160 KiB at BKI = 64, all of the CU's
// LDS): sub-tile 2t is A(t), 2t+1 is B(t).  Waves 0-3 fetch B(t+1) during L(t) (interval 2t) and
// wait for it at the end of M(t); waves 4-7 fetch A(t+2) during their L(t) (interval 2t+1) and wait
// at the end of their M(t).  A slot is rewritten only after the barrier that ends the interval in
// which the later half read its previous tenant (every L phase ends with lgkmcnt(0) + barrier),
// and every fetch has more than a whole interval to land before anyone needs it.
// Per-lane source offset (bytes) of a wave's first DMA piece of a sub-tile; piece i of the wave lies
// sub_dma_stride() bytes * i further on (the chunk swizzle repeats with that period), which keeps the
// per-piece part of the address wave-uniform.
template <bool COL, int BKI>
__device__ __forceinline__ uint32_t sub_dma_offset0(int ld, int w4, int lane) {
  const int P = w4 * 64 + lane;
  int row, c;
  if constexpr (COL) {
    row = P >> 5;                       // 32 chunks per k-row of 256 columns; 8 k-rows per 4 pieces
    c = (P & 31) ^ ((row & 3) << 2);
  } else {
    static_assert(COL || BKI == 64, "row-major sub-tiles are 64 deep");
    row = P >> 3;                       // 8 chunks per row; 32 rows per 4 pieces
    c = (P & 7) ^ ((row >> 1) & 7);
  }
  return ((uint32_t)row * (uint32_t)ld + (uint32_t)c * 8u) * 2u;
}
template <bool COL> __device__ __forceinline__ uint32_t sub_dma_stride(int ld) { return (uint32_t)(COL ? 8 : 32) * (uint32_t)ld * 2u; }

template <bool A_COL, bool B_COL, int EPI, int ACT, bool WGRAD, bool BIAS_GRAD, int BKI, int WR, bool CMP = false>
__device__ __forceinline__ void pp_tile(const MfmaPair& pr, char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Two independent problems of the same kind may share a launch (the two weight-gradient GEMMs of a
  // backward pass): blocks [0, n0) work on the first, the rest on the second, so that the tail of the
  // first one's last wave of tiles is filled by the second instead of leaving CUs idle.
  const bool second = blockIdx.x >= (unsigned)pr.n0;
  const MfmaDesc& d = pr.d[second ? 1 : 0];
  const unsigned bid = second ? blockIdx.x - (unsigned)pr.n0 : blockIdx.x;
  // WR = MFMA row tiles per wave: 4 -> 256-row block tiles, 3 -> 192-row tiles (row-major A only),
  // which pack N = 768 problems on 256 CUs without a half-empty last wave of tiles.
  constexpr int BM = 64 * WR, KI = BKI / kKS, NSLOT = 5;
  constexpr int NTM = WR * 32 / kTR, NTN = 64 / kTR;  // MFMA tiles per wave
  // plain-store epilogues run on transposed accumulators (MFMA operands swapped), see tile_epilogue_t
  constexpr bool kT = !kM16 && (EPI == MEPI_STORE || EPI == MEPI_BIAS);
  static_assert(WR == 4 || (WR == 3 && !A_COL && !WGRAD), "192-row tiles: row-major A only");
  constexpr int kSub = 256 * BKI * 2;       // bytes of one ring slot (a 256 x BKI sub-tile)
  constexpr int NPB = kSub / 1024 / 4;      // DMA instructions per wave per B sub-tile
  constexpr int NPA = BM * BKI * 2 / 1024 / 4;  // ... per A sub-tile
  constexpr int NP = NPB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int h = wave_u >> 2, w4 = wave_u & 3;  // half (= row half of the tile), index inside the half
  const int wm = wave >> 2, wn = wave & 3;
  // block -> tile (XCD-aware order, see tile_of_block) and, for weight gradients, the K range of the block:
  // all segments (groups that share the expert) and all their rows, or -- split-K -- a run of whole segments
  // (ksplit <= nseg) / one slice of one segment's rows (ksplit > nseg), or -- dynamic slices -- one of the
  // ceil(rows[e] / L) equal slices of expert e's single segment
  const int nseg = WGRAD ? d.G / d.E_local : 1;
  int z, zpart, m0, n0, mext;
  int seg_lo = 0, seg_hi = nseg, kbase = 0, kchunk = d.K;
  if (WGRAD && d.kslots > 0) {
    const int tiles_m = (d.Mext + BM - 1) / BM, tiles_per_group = tiles_m * (d.N / BN);
    const int L = dyn_slice_rows(d);
    int nslices = 0;
    for (int e = 0; e < d.E_local; ++e) nslices += dyn_nslices(min(d.K, d.rows[e]), L);
    const int per_xcd = (nslices * tiles_per_group + 7) >> 3;
    const int xcd = bid & 7, local = bid >> 3;
    const int item = xcd * per_xcd + local;
    if (local >= per_xcd || item >= nslices * tiles_per_group) return;
    const int p = item / tiles_per_group, tile = item % tiles_per_group;
    int e = 0, sl = p, ns = 1, rows_e = 0;
    for (; e < d.E_local; ++e) {
      rows_e = min(d.K, d.rows[e]);
      ns = dyn_nslices(rows_e, L);
      if (sl < ns) break;
      sl -= ns;
    }
    z = e; zpart = p;
    if (d.nfast) { const int tiles_n = d.N / BN; n0 = (tile % tiles_n) * BN; m0 = (tile / tiles_n) * BM; }
    else { m0 = (tile % tiles_m) * BM; n0 = (tile / tiles_m) * BN; }
    mext = d.Mext;
    kchunk = ((rows_e + ns - 1) / ns + BKI - 1) / BKI * BKI;
    kbase = sl * kchunk;
    seg_hi = 1;
  } else {
    TileCoord tc;
    if (!tile_of_block<BM, WGRAD, CMP>(d, bid, tc)) return;
    z = tc.z; m0 = tc.m0; n0 = tc.n0; mext = tc.mext;
    zpart = tc.split * d.E_local + z;  // split-K partials are indexed by (split, expert)
    const int nsplit = WGRAD ? max(d.ksplit, 1) : 1;
    if (WGRAD && nsplit > 1) {
      if (nsplit <= nseg) {
        seg_lo = tc.split * (nseg / nsplit);
        seg_hi = seg_lo + nseg / nsplit;
      } else {
        const int rs = nsplit / nseg;  // row slices per segment
        seg_lo = tc.split / rs;
        seg_hi = seg_lo + 1;
        kchunk = ((d.K + rs - 1) / rs + BKI - 1) / BKI * BKI;
        kbase = (tc.split % rs) * kchunk;
      }
    }
  }

  acc_tile_t<kM16> acc[NTM][NTN];
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < (kM16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const bool do_bias_grad = BIAS_GRAD && n0 == 0;
#ifdef EVE_PP_STAMP
  uint32_t stamp_v = 0;
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime(); stamp_v = lane == 62 ? c_ : stamp_v; }
#endif

  // waves 0-3 only ever fetch B tiles, waves 4-7 only A tiles
  const uint32_t voff0 = h == 0 ? sub_dma_offset0<B_COL, BKI>(d.ldb, w4, lane) : sub_dma_offset0<A_COL, BKI>(d.lda, w4, lane);
  const uint32_t vstride = h == 0 ? sub_dma_stride<B_COL>(d.ldb) : sub_dma_stride<A_COL>(d.lda);
  const int np = h == 0 ? NPB : NPA;  // pieces per wave of my operand's sub-tile

  for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int g = WGRAD ? seg * d.E_local + z : z;
    const int e = WGRAD ? z : z % d.E_local;
    int kext = d.K;
    if (WGRAD && d.rows) kext = min(kext, d.rows[g]);
    kext = min(kext, kbase + kchunk);  // (first row past my slice; kbase = 0 and kchunk = K without row slices)
    const int nk = kext > kbase ? (kext - kbase + BKI - 1) / BKI : 0;
    if (nk == 0) continue;
    const bf16_t* Ag = d.A + (size_t)g * d.a_g;
    const bf16_t* Bg = d.B + (size_t)(WGRAD ? g : e) * d.b_g;
    const uint32_t a_bytes = (uint32_t)(A_COL ? (WGRAD ? kext : d.K) * d.lda : d.Mext * d.lda) * 2u;
    const uint32_t b_bytes = (uint32_t)(B_COL ? (WGRAD ? kext : d.K) * d.ldb : d.N * d.ldb) * 2u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, b_bytes, 0x00020000);
    // my operand's K tile t: origin inside the operand, and its ring slot
    auto my_origin = [&](int t) -> uint32_t {
#ifdef EVE_PP_STAMP
      if (d.same_tile) t = 0;
#endif
      if (h == 0) return (uint32_t)(B_COL ? (kbase + t * BKI) * d.ldb + n0 : n0 * d.ldb + kbase + t * BKI) * 2u;
      return (uint32_t)(A_COL ? (kbase + t * BKI) * d.lda + m0 : m0 * d.lda + kbase + t * BKI) * 2u;
    };
    auto my_slot = [&](int t) -> char* { return smem + ((2 * t + (h == 0 ? 1 : 0)) % NSLOT) * kSub; };  // prologue only
    auto dma_piece = [&](char* slot, uint32_t origin, int i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(h == 0 ? b_rsrc : a_rsrc,
                                               (__attribute__((address_space(3))) void*)(slot + (i * 4 + w4) * 1024), 16,
                                               voff0 + (origin + (uint32_t)i * vstride), 0, 0, 0);
    };

    bf16x8 fa[KI][NTM], fb[KI][NTN];
    // L phase: fragment reads of tile t, interleaved with the DMA pieces of the tile fetched now
    auto load_phase = [&](int sl_a, bool fetch, int tf, int t) {
      const int sl_b = sl_a + 1 >= NSLOT ? sl_a + 1 - NSLOT : sl_a + 1;
      // B(t+1) goes 2 slots past B(t); A(t+2) goes 4 slots past A(t)
      const int sl_f = h == 0 ? (sl_b + 2 >= NSLOT ? sl_b + 2 - NSLOT : sl_b + 2) : (sl_a + 4 >= NSLOT ? sl_a + 4 - NSLOT : sl_a + 4);
      const char* sa = smem + sl_a * kSub;
      const char* sb = smem + sl_b * kSub;
      char* slot = smem + sl_f * kSub;
      const uint32_t origin = my_origin(tf);
      // The fetch is interleaved with the fragment reads: the DMA issue is paced by the CU's texture path
      // and the reads by the LDS, so side by side the phase is shorter (not with the bias-gradient reads,
      // plain loads that the compiler would make wait for the fetch).
      constexpr bool kInterleave = !BIAS_GRAD && !kM16;
      // lane addresses of the transposed reads (asm reads, see frag_load_tr)
      uint32_t ta[NTM], tb[NTN];
      if constexpr (A_COL && !kM16) {
#pragma unroll
        for (int i = 0; i < NTM; ++i) ta[i] = lds_addr(sa) + tr_lane_offset<BM>(wm * (WR * 32) + i * 32, lane);
      }
      if constexpr (B_COL && !kM16) {
#pragma unroll
        for (int j = 0; j < NTN; ++j) tb[j] = lds_addr(sb) + tr_lane_offset<BN>(wn * 64 + j * 32, lane);
      }
#pragma unroll
      for (int ks = 0; ks < KI; ++ks) {
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          if constexpr (kM16) fb[ks][j] = frag_load16<B_COL, BN>(sb, wn * 64 + j * 16, ks, lane);
          else if constexpr (B_COL) fb[ks][j] = frag_load_tr<BN>(tb[j], ks);
          else fb[ks][j] = frag_load<false, BN>(sb, wn * 64 + j * 32, ks, lane);
        }
#pragma unroll
        for (int i = 0; i < NTM; ++i) {
          if constexpr (kM16) fa[ks][i] = frag_load16<A_COL, BM>(sa, wm * (WR * 32) + i * 16, ks, lane);
          else if constexpr (A_COL) fa[ks][i] = frag_load_tr<BM>(ta[i], ks);
          else fa[ks][i] = frag_load<false, BM>(sa, wm * (WR * 32) + i * 32, ks, lane);
        }
        if constexpr (kInterleave) {
          if (fetch) {
#pragma unroll
            for (int i = ks * (NP / KI); i < (ks + 1) * (NP / KI); ++i)
              if (i < np) dma_piece(slot, origin, i);
          }
        }
      }
      if constexpr (BIAS_GRAD) {
        if (do_bias_grad) {  // column sums of the k-major A tile, read back from LDS
#pragma unroll
          for (int i = 0; i < kSub / 16 / 512; ++i) {
            const int idx = threadIdx.x + 512 * i;
            const uint4 v = *reinterpret_cast<const uint4*>(sa + col_tile_off<BM>(idx >> 5, idx & 31));
            const uint32_t* w = &v.x;
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += bf16_bits_to_f32(w[j >> 1], j & 1);
          }
        }
      }
      // The fetch is issued after the reads: the compiler puts a vmcnt(0) in front of a transposed LDS
      // read that follows an LDS-DMA in program order, which would make the wave wait for its own fetch.
      EVE_STAMP(5);  // diagnostic build: fragment reads complete (the stamp drains lgkmcnt), fetch not yet issued
      if (!kInterleave && fetch) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (i < np) dma_piece(slot, origin, i);
      }
    };

    // prologue: A(0), A(1) by waves 4-7, B(0) by waves 0-3
    {
      char* s0 = my_slot(0);
      const uint32_t o0 = my_origin(0);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (i < np) dma_piece(s0, o0, i);
      if (h == 1 && nk > 1) {
        char* s1 = my_slot(1);
        const uint32_t o1 = my_origin(1);
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (i < np) dma_piece(s1, o1, i);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (h == 1) __builtin_amdgcn_s_barrier();  // interval 0: waves 4-7 have nothing to do yet
    int sl_a = 0;  // ring slot of A(t); B(t) is the next one
    for (int t = 0; t < nk; ++t) {
      const int tf = t + 1 + h;  // waves 0-3 fetch B(t+1), waves 4-7 fetch A(t+2)
      EVE_STAMP(0);
      load_phase(sl_a, tf < nk, tf, t);
      sl_a = sl_a + 2 >= NSLOT ? sl_a + 2 - NSLOT : sl_a + 2;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_fence(fa, fb);
      EVE_STAMP(1);
      __builtin_amdgcn_s_barrier();
      EVE_STAMP(2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KI; ++ks)
#pragma unroll
        for (int i = 0; i < NTM; ++i)
#pragma unroll
          for (int j = 0; j < NTN; ++j) {
            if constexpr (kM16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
            else if constexpr (kT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      EVE_STAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my fetch has landed (it had the whole M phase)
      EVE_STAMP(4);
      // no barrier after the last M phase of waves 4-7: nobody reads or writes the ring any more, and
      // waves 0-3 (one barrier ahead) go straight on to their epilogue / next segment
      if (!(h == 1 && t == nk - 1)) __builtin_amdgcn_s_barrier();
    }
  }
  // The ring is dead from here on (all reads and fetches completed before the last barrier each wave
  // passed), so each wave may overwrite its own epilogue image without another barrier.
#ifdef EVE_PP_STAMP
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime(); stamp_v = lane == 60 ? c_ : stamp_v; }
#endif
  // (zpart: the group, or the (split, expert) / slice index of a split-K partial)
  if constexpr (kT)
    tile_epilogue_t<EPI, WR>(acc, smem + wave_u * 4352, d, zpart, WGRAD ? z : z % d.E_local, m0, n0, mext, wave, lane);
  else
    tile_epilogue<EPI, ACT, WR, kM16>(acc, smem, d, zpart, WGRAD ? z : z % d.E_local, m0, n0, mext, wave, lane);
  if constexpr (BIAS_GRAD) {
    if (do_bias_grad) bias_grad_store<BM, 512>(bsum, smem, d, zpart, m0);  // block-uniform
  }
#ifdef EVE_PP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime(); stamp_v = lane == 61 ? c_ : stamp_v; }
  if (d.dbg && (bid == 0 || bid == 1000)) d.dbg[((bid ? 1 : 0) * 8 + wave) * 64 + lane] = stamp_v;
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <bool A_COL, bool B_COL, int EPI, int ACT, bool WGRAD, bool BIAS_GRAD, int BKI, int WR, bool CMP = false>
__global__ __launch_bounds__(512) void grouped_gemm_pp(MfmaPair pr) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pp_tile<A_COL, B_COL, EPI, ACT, WGRAD, BIAS_GRAD, BKI, WR, CMP>(pr, smem);
#endif
}

// Weight gradients: both GEMMs of a backward pass in one launch, every tile through the 64-deep body.  (The bias
// gradients are formed outside these GEMMs -- bias_grads_kernel -- since round 2: the column sums of the k-major A
// tile do not fit next to 96 fragment and 128 accumulator registers, and the 32-deep body that had room for them
// reaches about half the MFMA rate.)
template <int EPI>
__global__ __launch_bounds__(512) void grouped_gemm_pp_wgrad(MfmaPair pr) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pp_tile<true, true, EPI, EVE_ACT_IDENTITY, true, false, 64, 4>(pr, smem);
#endif
}

// ------------------------------------------------------------------ persistent ping-pong kernel
// The schedule of grouped_gemm_pp, but a workgroup walks tiles gb = blockIdx.x, + gridDim.x, ... and the
// operand ring never drains: the fetch cursor simply runs on into the next tile, so a tile's first
// sub-tiles are already in LDS when its predecessor's epilogue ends (a fresh workgroup spends ~5 k
// cycles fetching them cold).  The epilogue is one more phase per half, "E":
//     waves 0-3:  ... L(n-1) | M(n-1) |   E    | L'(0) | M'(0) ...
//     waves 4-7:  ... M(n-2) | L(n-1) | M(n-1) |   E   | L'(0) ...
// It stages 16-row half slabs (4 waves x 4.25 KiB) in the one ring slot that is free during both E
// phases -- the slot of the finished tile's last B sub-tile, which waves 4-7 refill only in their L'(0),
// after their own E -- or behind the ring where LDS allows (the 32-deep weight-gradient ring).
// Single K segment only (weight gradients under expert parallelism use grouped_gemm_pp).
struct PTile { int which, z, e, m0, n0, mext, nk, kext; };  // e: the group's expert

// Epilogue of one wave's 32*WR x 64 outputs through a wave-private [16][EPI_LD] fp32 image.
template <int EPI, int ACT, int WR>
__device__ __forceinline__ void tile_epilogue_half(const f32x16 (&acc)[WR][WTN], float* ep, const MfmaDesc& d, int z,
                                                   int e_out, int m0, int n0, int mext, int wave, int lane
#ifdef EVE_PP_STAMP
                                                   , uint32_t* stamp_e = nullptr  // diagnostic: lane 4q+k = stamp k of half slab q
#endif
) {
#ifdef EVE_PP_STAMP
#define EVE_ESTAMP(q_, k_)                                                          \
  do {                                                                             \
    if (stamp_e) {                                                                 \
      const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime();                  \
      *stamp_e = lane == (q_) * 4 + (k_) ? c_ : *stamp_e;                          \
    }                                                                              \
  } while (0)
#else
#define EVE_ESTAMP(q_, k_) do { } while (0)
#endif
  const int wm = wave >> 2, wn = wave & 3;
  bf16_t* Cg = d.C ? d.C + (size_t)z * d.c_g : nullptr;
  bf16_t* C2g = d.C2 ? d.C2 + (size_t)z * d.c_g : nullptr;
  const bf16_t* Zg = d.Zin ? d.Zin + (size_t)z * d.c_g : nullptr;
  float bv[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    bv[j] = 0.f;
    if constexpr (EPI == MEPI_BIAS || EPI == MEPI_BIAS_ACT)
      bv[j] = to_f32<bf16_t>(d.bias[(size_t)e_out * d.N + n0 + wn * 64 + j * 32 + (lane & 31)]);
  }
  const int irow = lane >> 3, ich = lane & 7;
  const int ncol = n0 + wn * 64 + ich * 8;
  const int mbase = m0 + wm * (WR * 32);
  // Global accesses go through buffer descriptors whose extent is the group's first `mext` rows: a lane whose row
  // lies beyond it loads zeros and its store is dropped by the hardware.  No exec-masked branch around a store
  // then -- behind such branches the compiler must assume the stores were NOT issued when it counts vmcnt, and
  // the wait for a half slab's saved-tensor loads (issued one half slab ahead, with that half slab's stores
  // between them and the wait) became a wait for those stores' completion: the write latency, about 2 000
  // cycles, once per half slab (in-kernel stamps, round 3) -- and no 64-bit address arithmetic per access.
  // (mext comes from rows[], a vector load as far as the compiler knows: readfirstlane keeps the descriptors scalar)
  const uint32_t ext = (uint32_t)__builtin_amdgcn_readfirstlane(mext) * (uint32_t)d.ldc * 2u;
  auto group_rsrc = [&](const bf16_t* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, p ? ext : 0u, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t c_rsrc = group_rsrc(Cg), c2_rsrc = group_rsrc(C2g), z_rsrc = group_rsrc(Zg);
  const uint32_t voff0 = ((uint32_t)(mbase + irow) * (uint32_t)d.ldc + (uint32_t)ncol) * 2u;  // row mbase + irow
  const uint32_t step8 = 8u * (uint32_t)d.ldc * 2u;                                           // eight rows on
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  // DACT: the saved factors of the wave's whole 32 WR x 64 block are requested before the first store is issued
  // (16 WR registers, which the K loop's fragments occupy outside the epilogue): they are then ahead of the
  // tile's stores in the CU's memory pipeline and in the wave's in-order vmcnt queue, and their latency is paid
  // once per tile beside the first half slab's staging.  Requested a half slab ahead (round 2), each load queued
  // behind the stores of all eight waves and took 1 500 - 2 500 cycles, once per half slab.
  uint4 zbuf[EPI == MEPI_DACT ? 2 * WR : 1][2];
  auto load_z = [&](int q, uint4 (&dst)[2]) {  // half slab q: rows mbase + 16 q ..
    if constexpr (EPI == MEPI_DACT) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, voff0 + (uint32_t)(2 * q + it) * step8, 0, 0);
        dst[it] = make_uint4(w.x, w.y, w.z, w.w);
      }
    }
  };
  auto store8 = [&](const __amdgpu_buffer_rsrc_t& rsrc, const uint4& v, int q, int it) {
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, voff0 + (uint32_t)(2 * q + it) * step8, 0, 0);
  };
  if constexpr (EPI == MEPI_DACT) {
#pragma unroll
    for (int q = 0; q < 2 * WR; ++q) load_z(q, zbuf[q]);
  }
  const int c = lane & 31, hh = lane >> 5;
  f32x2_t cs[4];  // DACT: this lane's share of the column sums of the stored values
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = f32x2_t{0.f, 0.f};
  // The image is private to the wave and the LDS executes one wave's instructions in issue order, so nothing
  // has to drain between the stores of a half slab and the reads that follow them, nor between those reads and
  // the stores of the next half slab: the next half slab's stores are issued as soon as this one's reads are,
  // and their transfer (4 cycles of the VGPR -> LDS path per ds_write_b32) and the reads' latency run beside the
  // conversions / activation of this half slab instead of in front of them.  (Waiting for lgkmcnt(0) after the
  // stores and again after the reads, as round 2 did, made a tile's epilogue the SUM of its LDS, VALU and
  // store-issue time.)
  auto stage = [&](int q) {
    const int i = q >> 1, hv = q & 1;
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = hv * 8 + rr;
        const int row = (r & 3) + 8 * ((r >> 2) & 1) + 4 * hh;
        ep[row * EPI_LD + j * 32 + c] = add_bias<EPI>(acc[i][j][r], bv[j]);
      }
  };
  EVE_ESTAMP(0, 3);  // (bias loaded)
  stage(0);
#pragma unroll
  for (int q = 0; q < 2 * WR; ++q) {
    __builtin_amdgcn_wave_barrier();
    EVE_ESTAMP(q, 0);
    float4 lo_[2], hi_[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 8 + irow;
      lo_[it] = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ich * 8);
      hi_[it] = *reinterpret_cast<const float4*>(ep + row * EPI_LD + ich * 8 + 4);
    }
    __builtin_amdgcn_wave_barrier();
    if (q + 1 < 2 * WR) stage(q + 1);
    __builtin_amdgcn_wave_barrier();
    EVE_ESTAMP(q, 1);  // (the stamp drains lgkmcnt: reads and the next half slab's stores are complete)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      uint4 o1, o2;
      // (rows beyond the extent: zero accumulators -- the A fetch ends at the extent -- times a zero factor)
      epi_row8<EPI, ACT>(lo_[it], hi_[it], zbuf[EPI == MEPI_DACT ? q : 0][it], Cg != nullptr, true, cs, o1, o2);
#ifdef EVE_EXP_NOSTORE  // experiment (diagnostic builds): 1 = no global stores, 2 = only the activation
      if (EVE_EXP_NOSTORE == 2 && EPI == MEPI_BIAS_ACT) store8(c2_rsrc, o2, q, it);
      else asm volatile("" :: "v"(o1.x), "v"(o1.y), "v"(o1.z), "v"(o1.w), "v"(o2.x), "v"(o2.y), "v"(o2.z), "v"(o2.w));
#else
      if constexpr (EPI == MEPI_BIAS_ACT) {
        if (Cg) store8(c_rsrc, o1, q, it);  // (uniform: no saved tensor under no_grad)
        store8(c2_rsrc, o2, q, it);
      } else {
        store8(c_rsrc, o1, q, it);
      }
#endif
    }
    EVE_ESTAMP(q, 2);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the caller reuses or hands on the image)
  __builtin_amdgcn_wave_barrier();
  if constexpr (EPI == MEPI_DACT) colsum_store(cs, d, z, m0 / (WR * 32) + wm, ncol, lane);
}

template <bool A_COL, bool B_COL, int EPI, int ACT, bool WGRAD, bool BIAS_GRAD, int BKI, int WR, bool CMP = false>
__global__ __launch_bounds__(512) void grouped_gemm_ppp(MfmaPair pr) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64 * WR, KI = BKI / 16, NSLOT = 5;
  static_assert(WR == 4 || (WR == 3 && !A_COL && !WGRAD), "192-row tiles: row-major A only");
  static_assert(!BIAS_GRAD || (A_COL && WR == 4), "bias gradient: k-major A tiles");
  constexpr int kSub = 256 * BKI * 2;
  constexpr int kRing = NSLOT * kSub;
  // plain-store epilogues run on transposed accumulators (MFMA operands swapped), see tile_epilogue_t
  constexpr bool kT = EPI == MEPI_STORE || EPI == MEPI_BIAS;
  constexpr int kStageWave = 16 * EPI_LD * 4;
  static_assert(kStageWave >= 32 * kImgPitch, "tile_epilogue_t's bf16 slab image fits the staging area");
  constexpr bool kStageInRing = kRing + 8 * kStageWave > 160 * 1024;
  // Long (activation) epilogues: the two halves' E phases must overlap each other -- one after the other, as the
  // plain schedule runs them, would cost a whole extra epilogue per tile.  Waves 4-7 therefore go from their last
  // M phase straight into E (no barrier in between) and spend one empty phase after it, which restores their
  // one-phase lag.  Both halves then stage at the same time: waves 0-3 in ring slot sl_c + 4, waves 4-7 in slot
  // sl_c + 3 (free until waves 0-3 fetch into it in the next tile's first L phase, which starts after the barrier
  // that ends both E phases).
  constexpr bool kResync = EPI == MEPI_BIAS_ACT || EPI == MEPI_DACT;
  static_assert(!kResync || !WGRAD, "resynchronised E phases: forward / dgrad launches (every tile has K steps)");
  constexpr int NPB = kSub / 1024 / 4;
  constexpr int NPA = BM * BKI * 2 / 1024 / 4;
  constexpr int NP = NPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int h = wave_u >> 2, w4 = wave_u & 3;
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned ntotal = (unsigned)pr.ntotal;
  const int np = h == 0 ? NPB : NPA;

  // ---- tile table.  The workgroup's tiles are gb = blockIdx.x + i * gridDim.x, i < 64 (launch_pp only takes this
  // kernel when that covers the launch).  Lane i decodes tile i ONCE, here, with vector arithmetic -- gb -> (group,
  // row tile, column tile) in the XCD-aware order costs half a dozen integer divisions -- and keeps the result in
  // two registers; both cursors (the tile being multiplied, the tile being fetched) then move on with two
  // v_readlane each.  Decoding at every move (round 2) put ~1 600 cycles of scalar division into the L phase in
  // which a cursor crossed a tile boundary, once per tile and half (in-kernel stamps: a K step of 4 100 cycles
  // instead of 2 400 at t = nk - 3 and again at t = nk - 2), 6 % of a K = 768 tile.
  //   tt0: bit 0 valid | bit 1 problem | 10 bits column tile | 14 bits group | 6 bits group / E_local
  //   tt1: 12 bits row tile origin / 64 | 20 bits row extent
  uint32_t tt0 = 0, tt1 = 0;
  int ttk = 0;  // weight gradients: K extent (rows of the expert's segment)
  int n_mine;
  {
    const unsigned gb = blockIdx.x + (unsigned)lane * gridDim.x;
    n_mine = __builtin_popcountll(__ballot(gb < ntotal));
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      if (gb < ntotal && (w == 1) == (gb >= (unsigned)pr.n0)) {
        const MfmaDesc& d = pr.d[w];
        TileCoord tc;  // (persistent launches never split K: tc.split = 0)
        if (tile_of_block<BM, WGRAD, CMP>(d, w ? gb - (unsigned)pr.n0 : gb, tc)) {
          const int src = WGRAD ? 0 : tc.z / d.E_local;
          tt0 = 1u | ((uint32_t)w << 1) | ((uint32_t)(tc.n0 / BN) << 2) | ((uint32_t)tc.z << 12) | ((uint32_t)src << 26);
          tt1 = (uint32_t)(tc.m0 / 64) | ((uint32_t)tc.mext << 12);
          if (WGRAD) ttk = d.rows ? min(d.K, d.rows[tc.z]) : d.K;  // single segment: group == expert
        }
      }
    }
  }
  auto decode = [&](int idx, PTile& T) -> bool {
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)tt0, idx);
    if (!(a & 1u)) return false;  // the block id maps to no work
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)tt1, idx);
    T.which = (int)((a >> 1) & 1u);
    const int e_local = T.which ? pr.d[1].E_local : pr.d[0].E_local, kfull = T.which ? pr.d[1].K : pr.d[0].K;
    T.n0 = (int)((a >> 2) & 1023u) * BN;
    T.z = (int)((a >> 12) & 16383u);
    T.e = WGRAD ? T.z : T.z - (int)(a >> 26) * e_local;
    T.m0 = (int)(b & 4095u) * 64;
    T.mext = (int)(b >> 12);
    T.kext = WGRAD ? __builtin_amdgcn_readlane(ttk, idx) : kfull;
    T.nk = (T.kext + BKI - 1) / BKI;
    return true;
  };
  auto advance = [&](int& idx, PTile& T, bool need_k) -> bool {
    for (++idx; idx < n_mine; ++idx)
      if (decode(__builtin_amdgcn_readfirstlane(idx), T) && (!need_k || T.nk > 0)) return true;
    return false;
  };

  int idx_c = -1;
  PTile cur;
  bool have = advance(idx_c, cur, false);
  if (!have) return;

  // ---- fetch cursor: my operand (waves 0-3: B, waves 4-7: A) of K step t_f of tile ft
  int idx_f = idx_c;
  PTile ft = cur;
  int t_f = 0;
  bool f_valid = true;
  if (ft.nk == 0) f_valid = advance(idx_f, ft, true);
  __amdgpu_buffer_rsrc_t f_rsrc;
  uint32_t f_voff0 = 0, f_vstride = 0, f_origin0 = 0, f_ostep = 0;
  auto set_fetch_desc = [&]() {
    const MfmaDesc& d = pr.d[ft.which];
    const int e = ft.e;
    if (h == 0) {
      const bf16_t* Bg = d.B + (size_t)(WGRAD ? ft.z : e) * d.b_g;
      const uint32_t bytes = (uint32_t)(B_COL ? (WGRAD ? ft.kext : d.K) * d.ldb : d.N * d.ldb) * 2u;
      f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, bytes, 0x00020000);
      f_voff0 = sub_dma_offset0<B_COL, BKI>(d.ldb, w4, lane);
      f_vstride = sub_dma_stride<B_COL>(d.ldb);
      f_origin0 = (uint32_t)(B_COL ? ft.n0 : ft.n0 * d.ldb) * 2u;
      f_ostep = (uint32_t)(B_COL ? BKI * d.ldb : BKI) * 2u;
    } else {
      const bf16_t* Ag = d.A + (size_t)ft.z * d.a_g;
      // (row-major A: the extent ends at the group's last occupied row, so rows beyond it are fetched as zeros and
      // their accumulators are exactly zero, whatever the tensor holds there: tile_epilogue_half relies on it)
      const uint32_t bytes = (uint32_t)(A_COL ? (WGRAD ? ft.kext : d.K) * d.lda : ft.mext * d.lda) * 2u;
      f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, bytes, 0x00020000);
      f_voff0 = sub_dma_offset0<A_COL, BKI>(d.lda, w4, lane);
      f_vstride = sub_dma_stride<A_COL>(d.lda);
      f_origin0 = (uint32_t)(A_COL ? ft.m0 : ft.m0 * d.lda) * 2u;
      f_ostep = (uint32_t)(A_COL ? BKI * d.lda : BKI) * 2u;
    }
  };
  if (f_valid) set_fetch_desc();
  // issue my operand's sub-tile of the cursor's K step into ring slot `sl`, then move the cursor on
  auto fetch_next = [&](int sl) {
    char* slot = smem + sl * kSub;
    const uint32_t origin = f_origin0 + (uint32_t)t_f * f_ostep;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (i < np)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(f_rsrc, (__attribute__((address_space(3))) void*)(slot + (i * 4 + w4) * 1024),
                                                 16, f_voff0 + (origin + (uint32_t)i * f_vstride), 0, 0, 0);
    if (++t_f == ft.nk) {
      t_f = 0;
      f_valid = advance(idx_f, ft, true);
      if (f_valid) set_fetch_desc();
    }
  };
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };

  f32x16 acc[WR][WTN];
#pragma unroll
  for (int i = 0; i < WR; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  // prologue: global steps 0 and 1 -> A(0) slot 0, B(0) slot 1, A(1) slot 2
  if (f_valid) {
    if (h == 0) {
      fetch_next(1);
    } else {
      fetch_next(0);
      if (f_valid) fetch_next(2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (h == 1) __builtin_amdgcn_s_barrier();  // interval 0: waves 4-7 have nothing to do yet
  int sl_c = 0;  // ring slot of the A sub-tile of the K step being computed; its B sub-tile is the next slot
  bf16x8 fa[KI][WR], fb[KI][WTN];
#ifdef EVE_PP_STAMP
  // diagnostic build (tools/pp_stamps.hip): per tile i < 12, lane 5i+k of stamp_v: k = 0 tile start, 1 K loop done,
  // 2 epilogue done, 3 barrier(s) behind the epilogue passed; lanes 62 / 63: kernel start (cycles / 100 MHz ticks).
  // K-step stamps of tile number pr.d[0].same_tile: lane 6t+k of stamp_w, t < 10: k = 0 L start, 1 reads done
  // (lgkmcnt 0), 2 barrier passed, 3 MFMAs issued, 4 fetch landed (vmcnt 0), 5 barrier passed
  uint32_t stamp_v = 0, stamp_w = 0, stamp_e = 0;
  int tile_i = 0;
#define EVE_KSTAMP(k)                                                              \
  do {                                                                             \
    if (tile_i == pr.d[0].same_tile && t < 10) {                                   \
      const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime();                  \
      stamp_w = lane == t * 6 + (k) ? c_ : stamp_w;                                \
    }                                                                              \
  } while (0)
#define EVE_PSTAMP(k)                                                              \
  do {                                                                             \
    if (tile_i < 12) {                                                             \
      const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime();                  \
      stamp_v = lane == tile_i * 5 + (k) ? c_ : stamp_v;                           \
    }                                                                              \
  } while (0)
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime(); stamp_v = lane == 62 ? c_ : stamp_v; }
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memrealtime(); stamp_v = lane == 63 ? c_ : stamp_v; }
#else
#define EVE_PSTAMP(k) do { } while (0)
#define EVE_KSTAMP(k) do { } while (0)
#endif
  for (;;) {
    const MfmaDesc& d = pr.d[cur.which];
    const bool do_bias_grad = BIAS_GRAD && cur.n0 == 0;
    EVE_PSTAMP(0);
    for (int t = 0; t < cur.nk; ++t) {
      EVE_KSTAMP(0);
      // ---- L phase
      const char* sa = smem + sl_c * kSub;
      const char* sb = smem + wrap(sl_c + 1) * kSub;
      uint32_t ta[WR], tb[WTN];  // lane addresses of the transposed reads (asm reads, see frag_load_tr)
      if constexpr (A_COL) {
#pragma unroll
        for (int i = 0; i < WR; ++i) ta[i] = lds_addr(sa) + tr_lane_offset<BM>(wm * (WR * 32) + i * 32, lane);
      }
      if constexpr (B_COL) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) tb[j] = lds_addr(sb) + tr_lane_offset<BN>(wn * 64 + j * 32, lane);
      }
      // without bias-gradient reads (plain loads that the compiler would make wait for the fetch) the fetch
      // goes first: its issue and latency then overlap the fragment reads
      if (!BIAS_GRAD && f_valid) fetch_next(wrap(sl_c + 3 + h));  // waves 0-3: B of the next step; waves 4-7: A of the one after
#pragma unroll
      for (int ks = 0; ks < KI; ++ks) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          if constexpr (B_COL) fb[ks][j] = frag_load_tr<BN>(tb[j], ks);
          else fb[ks][j] = frag_load<false, BN>(sb, wn * 64 + j * 32, ks, lane);
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
          if constexpr (A_COL) fa[ks][i] = frag_load_tr<BM>(ta[i], ks);
          else fa[ks][i] = frag_load<false, BM>(sa, wm * (WR * 32) + i * 32, ks, lane);
        }
      }
      if constexpr (BIAS_GRAD) {
        if (do_bias_grad) {
          // column sums of the k-major A tile: wave w owns 16-byte chunks 4w .. 4w+3 (32 columns), lane

          // (lane & 3) one chunk and (lane >> 2) every 16th k row; reduced inside the wave at the end
          const int chunk = 4 * wave + (lane & 3);
#pragma unroll
          for (int kk = 0; kk < BKI; kk += 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(sa + col_tile_off<BM>(kk + (lane >> 2), chunk));
            const uint32_t* w = &v.x;
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += bf16_bits_to_f32(w[j >> 1], j & 1);
          }
        }
      }
      if (BIAS_GRAD && f_valid) fetch_next(wrap(sl_c + 3 + h));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_fence(fa, fb);
      EVE_KSTAMP(1);
      __builtin_amdgcn_s_barrier();
      EVE_KSTAMP(2);
      // ---- M phase
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KI; ++ks)
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            acc[i][j] = kT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      EVE_KSTAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my fetch has landed (it had the whole M phase)
      EVE_KSTAMP(4);
      if (!(kResync && h == 1 && t == cur.nk - 1)) __builtin_amdgcn_s_barrier();
      EVE_KSTAMP(5);
      sl_c = wrap(sl_c + 2);
    }
    EVE_PSTAMP(1);
    // ---- E phase
    // (behind the ring every wave has its own image: after a tile without K steps -- an expert that
    // received no rows -- waves 0-3 are in their next E phase while waves 4-7 are still in this one)
    float* ep = reinterpret_cast<float*>(kStageInRing ? smem + wrap(sl_c + 4 - (kResync ? h : 0)) * kSub + w4 * kStageWave
                                                      : smem + kRing + wave_u * kStageWave);
    if (WGRAD && cur.nk == 0) {
      // An expert that received no rows: its gradient tile is zero.  Stored straight from registers -- after
      // such a tile waves 0-3 reach their next E phase while waves 4-7 are still in this one, and with the
      // staging image inside the ring the two would share it.
      bf16_t* Cg = d.C + (size_t)cur.z * d.c_g;
#pragma unroll 4
      for (int it = 0; it < WR * 4; ++it) {
        const int m = cur.m0 + wm * (WR * 32) + it * 8 + (lane >> 3);
        if (m < cur.mext)
          *reinterpret_cast<uint4*>(Cg + (size_t)m * d.ldc + cur.n0 + wn * 64 + (lane & 7) * 8) = make_uint4(0, 0, 0, 0);
      }
    } else {
      if constexpr (kT)
        tile_epilogue_t<EPI, WR>(acc, reinterpret_cast<char*>(ep), d, cur.z, cur.e, cur.m0, cur.n0, cur.mext, wave, lane);
      else {
        tile_epilogue_half<EPI, ACT, WR>(acc, ep, d, cur.z, cur.e, cur.m0, cur.n0, cur.mext, wave, lane
#ifdef EVE_PP_STAMP
                                         , tile_i == pr.d[0].same_tile ? &stamp_e : nullptr
#endif
        );
      }
    }
    if constexpr (BIAS_GRAD) {
      if (do_bias_grad) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = bsum[j];
          v += __shfl_xor(v, 4, 64);
          v += __shfl_xor(v, 8, 64);
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          bsum[j] = v;
        }
        if (lane < 4)
          *reinterpret_cast<uint4*>(d.dbias + (size_t)cur.z * d.Mext + cur.m0 + (4 * wave + lane) * 8) = pack8(bsum);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    have = advance(idx_c, cur, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my staging image is free again
    EVE_PSTAMP(2);
    // waves 4-7 run one phase behind: their last E phase has no partner phase and no barrier
    if (!have) {
      if (h == 0 && !kResync) __builtin_amdgcn_s_barrier();
      break;
    }
    __builtin_amdgcn_s_barrier();
    if (kResync && h == 1) __builtin_amdgcn_s_barrier();  // the empty phase: waves 0-3 are in the next tile's first L phase
    EVE_PSTAMP(3);
#ifdef EVE_PP_STAMP
    ++tile_i;
#endif
  }
#ifdef EVE_PP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memtime(); stamp_v = lane == 60 ? c_ : stamp_v; }
  { const uint32_t c_ = (uint32_t)__builtin_amdgcn_s_memrealtime(); stamp_v = lane == 61 ? c_ : stamp_v; }
  if (pr.d[0].dbg && (blockIdx.x == 0 || blockIdx.x == 100)) {
    pr.d[0].dbg[((blockIdx.x ? 1 : 0) * 8 + wave) * 64 + lane] = stamp_v;
    pr.d[0].dbg[1024 + ((blockIdx.x ? 1 : 0) * 8 + wave) * 64 + lane] = stamp_w;
    pr.d[0].dbg[2048 + ((blockIdx.x ? 1 : 0) * 8 + wave) * 64 + lane] = stamp_e;
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}


// out[i] = bf16(sum over splits j of part[j * stride + i]), splits in order; 8 elements per thread
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, bf16_t* __restrict__ out,
                                                           size_t n, size_t stride, int nsplit) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.f;
  for (int j = 0; j < nsplit; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(part + (size_t)j * stride + i);
    const float4 b = *reinterpret_cast<const float4*>(part + (size_t)j * stride + i + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  *reinterpret_cast<uint4*>(out + i) = pack8(v);
}

// The same for dynamic slices (MfmaDesc::kslots): expert blockIdx.y's slices are part[first .. first + ns), slices
// numbered expert-major as the GEMM numbered them (dyn_slice_rows / dyn_nslices on the same rows[]); blockIdx.z
// picks the problem (dw2 / dw1: both have n elements per expert).
__global__ __launch_bounds__(256) void wgrad_reduce_dyn_kernel(const float* __restrict__ part0, bf16_t* __restrict__ out0,
                                                               const float* __restrict__ part1, bf16_t* __restrict__ out1,
                                                               const int32_t* __restrict__ rows, int E_local, int K,
                                                               int kslots, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const int e = blockIdx.y;
  int total = 0;
  for (int q = 0; q < E_local; ++q) total += min(K, rows[q]);
  const int L = max(64, ((total + kslots - 1) / kslots + 63) & ~63);
  int first = 0;
  for (int q = 0; q < e; ++q) first += dyn_nslices(min(K, rows[q]), L);
  const int ns = dyn_nslices(min(K, rows[e]), L);
  const float* part = (blockIdx.z ? part1 : part0) + (size_t)first * n + i;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.f;
  for (int j = 0; j < ns; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(part + (size_t)j * n);
    const float4 b = *reinterpret_cast<const float4*>(part + (size_t)j * n + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  *reinterpret_cast<uint4*>((blockIdx.z ? out1 : out0) + (size_t)e * n + i) = pack8(v);
}

// Both bias gradients of the ping-pong path in ONE launch (they sit between the dz and the dx GEMM, where a launch
// of a few microseconds is pure latency): blocks [0, nb2) form db2, the rest db1.
//   db2: out2[e][c] = bf16(sum over the groups g that share expert e and rows r < rows[g] of dy[g][r][c]), dy a
//        row-major [G, R, M] bf16 tensor.  Block = 64 columns (8 chunks of 16 B) x 128 row lanes; each thread walks
//        its rows in order (two loads in flight), the row lanes are added through LDS in order.
//   db1: out1[e][n] = bf16(sum over the groups g = seg * E_local + e and over the wave row blocks ("slots") that the
//        MEPI_DACT launch wrote for g, of part[(g * slots + s) * I + n]); fixed order.  Slots of tiles the launch
//        skipped (first row at or beyond rows[g]) were never written and are not read.  Block = 1024 columns.
__global__ __launch_bounds__(1024) void bias_grads_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ out2,
                                                          const float* __restrict__ part, bf16_t* __restrict__ out1,
                                                          const int32_t* __restrict__ rows, int G, int E_local, int R,
                                                          int M, int I, int slots, int tile_rows, int spt, int nb2) {
  __shared__ float red[128][64 + 1];
  if ((int)blockIdx.x >= nb2) {
    const int b = blockIdx.x - nb2, per_e = (I + 1023) / 1024;
    const int n = (b % per_e) * 1024 + threadIdx.x, e = b / per_e;
    if (n >= I) return;
    float acc = 0.f;
    for (int g = e; g < G; g += E_local) {
      const int mext = rows ? min(R, rows[g]) : R;
      const int valid = (mext + tile_rows - 1) / tile_rows * spt;
      for (int sl = 0; sl < valid; ++sl) acc += part[((size_t)g * slots + sl) * I + n];
    }
    out1[(size_t)e * I + n] = static_cast<bf16_t>(acc);
    return;
  }
  const int per_e = (M + 63) / 64;
  const int bx = blockIdx.x % per_e, e = blockIdx.x / per_e;
  const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = bx * 64 + ch * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (col < M) {
    for (int g = e; g < G; g += E_local) {
      const int mext = rows ? min(R, rows[g]) : R;
      const bf16_t* xg = dy + (size_t)g * R * M + col;
      int r = rl;
      for (; r + 128 < mext; r += 256) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(xg + (size_t)r * M);
        const uint4 v1 = *reinterpret_cast<const uint4*>(xg + (size_t)(r + 128) * M);
        const uint32_t *w0 = &v0.x, *w1 = &v1.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_bits_to_f32(w0[j >> 1], j & 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_bits_to_f32(w1[j >> 1], j & 1);
      }
      if (r < mext) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(xg + (size_t)r * M);
        const uint32_t* w0 = &v0.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_bits_to_f32(w0[j >> 1], j & 1);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][ch * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float sum = 0.f;
    for (int r = 0; r < 128; ++r) sum += red[r][threadIdx.x];
    const int c = bx * 64 + threadIdx.x;
    if (c < M) out2[(size_t)e * M + c] = static_cast<bf16_t>(sum);
  }
}

bool ffn_mfma_supported(int M, int I) { return M % 256 == 0 && I % 256 == 0; }

// Kernel attributes (the dynamic-LDS limit) and the CU count belong to a DEVICE: a process that launches on a
// second GPU must raise the limit there as well.  One bit per device and kernel, set after the attribute call
// succeeded with that device current; racing threads at worst repeat the (idempotent) call.
constexpr int kMaxDevices = 64;
static int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> dev) & 1ull; }
  void set(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};
static int reserve_lds(PerDeviceOnce& once, const void* kern, int bytes) {
  const int dev = current_device();
  if (once.done(dev)) return EVE_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    set_error("ffn_mfma: cannot reserve %d bytes of LDS on device %d", bytes, dev);
    return EVE_ERR_LAUNCH;
  }
  once.set(dev);
  return EVE_OK;
}

template <int BM, int NS, bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BIAS_GRAD>
static int launch_cfg(const MfmaDesc& d, int act, hipStream_t st) {
  using C_ = Cfg<BM, NS>;
  const int ngroups = WGRAD ? d.E_local : d.G;
  dim3 grid(8 * ceil_div(ngroups * (d.N / BN) * ceil_div(d.Mext, BM), 8));
#define EVE_MFMA_LAUNCH(A)                                                                                   \
  do {                                                                                                       \
    auto kern = grouped_gemm_mfma<BM, NS, A_COL, B_COL, EPI, A, WGRAD, BIAS_GRAD>;                           \
    static PerDeviceOnce once;                                                                               \
    if (int rc_ = reserve_lds(once, reinterpret_cast<const void*>(kern), C_::kLds)) return rc_;              \
    hipLaunchKernelGGL(kern, grid, dim3(C_::kThreads), C_::kLds, st, d);                                     \
  } while (0)
  if constexpr (EPI == MEPI_BIAS_ACT || EPI == MEPI_DACT) {
    switch (act) {
      case EVE_ACT_QUICK_GELU: EVE_MFMA_LAUNCH(EVE_ACT_QUICK_GELU); break;
      case EVE_ACT_GELU: EVE_MFMA_LAUNCH(EVE_ACT_GELU); break;
      case EVE_ACT_RELU: EVE_MFMA_LAUNCH(EVE_ACT_RELU); break;
      case EVE_ACT_IDENTITY: EVE_MFMA_LAUNCH(EVE_ACT_IDENTITY); break;
      default:
        set_error("ffn_mfma: unknown activation %d", act);
        return EVE_ERR_ARG;
    }
  } else {
    EVE_MFMA_LAUNCH(EVE_ACT_IDENTITY);
  }
#undef EVE_MFMA_LAUNCH
  EVE_CHECK_LAUNCH("grouped_gemm_mfma");
  return EVE_OK;
}

static int cu_count() {  // of the current device; cached per device
  static std::atomic<int> n[kMaxDevices];
  const int dev = current_device();
  int v = n[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (!(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)) v = 256;
    n[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// Debug knobs for A/B timing.  They exist only in a build with -DEVE_FFN_DEBUG_KNOBS (`python eve_amd/build.py
// --debug-knobs` writes eve_amd/libeve_moe_dbg.so; select it with EVE_MOE_LIB); the shipped library reads no
// environment variable.  Read once per process:
//   EVE_FFN_KERNEL=lockstep    the older lock-step 256-row kernel instead of the ping-pong kernels
//   EVE_FFN_PERSIST=0          one tile per workgroup instead of the persistent kernel
//   EVE_FFN_PERSIST_ACT=0      ... for the activation-epilogue launches only
//   EVE_FFN_TILE_ROWS=128|192|256   force the block-tile height of the forward / dgrad launches
//   EVE_FFN_KSPLIT=n           force the weight-gradient split-K factor
//   EVE_FFN_WGRAD32=1          32-deep persistent weight-gradient kernel for all tiles
//   EVE_FFN_WGRAD_PERSIST=1    persistent 64-deep kernel for the (unsplit) weight-gradient launch (measured in round 3:
//                              eve_ffn_bwd 0.646 ms against 0.595 ms with one tile per workgroup -- it spills)
//   EVE_FFN_NFAST=0            weight gradients: row tile fastest for both GEMMs (MfmaDesc::nfast off)
//   EVE_FFN_COMPACT=0          few unequal groups: the plain G x ceil(R / BM) tile enumeration and equal K splits
struct Knobs {
  bool pingpong, persist, persist_act, wgrad32, wgrad_persist, compact, nfast;
  int tile_rows, ksplit;
};
static const Knobs& knobs() {
#ifdef EVE_FFN_DEBUG_KNOBS
  static const Knobs k = [] {
    auto env = [](const char* n) { return getenv(n); };
    Knobs q;
    q.pingpong = !(env("EVE_FFN_KERNEL") && env("EVE_FFN_KERNEL")[0] == 'l');
    q.persist = !(env("EVE_FFN_PERSIST") && env("EVE_FFN_PERSIST")[0] == '0');
    q.persist_act = !(env("EVE_FFN_PERSIST_ACT") && env("EVE_FFN_PERSIST_ACT")[0] == '0');
    q.wgrad32 = env("EVE_FFN_WGRAD32") != nullptr;
    q.wgrad_persist = env("EVE_FFN_WGRAD_PERSIST") && env("EVE_FFN_WGRAD_PERSIST")[0] == '1';
    q.compact = !(env("EVE_FFN_COMPACT") && env("EVE_FFN_COMPACT")[0] == '0');
    q.nfast = !(env("EVE_FFN_NFAST") && env("EVE_FFN_NFAST")[0] == '0');
    q.tile_rows = env("EVE_FFN_TILE_ROWS") ? atoi(env("EVE_FFN_TILE_ROWS")) : 0;
    q.ksplit = env("EVE_FFN_KSPLIT") ? atoi(env("EVE_FFN_KSPLIT")) : 0;
    return q;
  }();
#else
  static const Knobs k = {true, true, true, false, false, true, true, 0, 0};
#endif
  return k;
}
#ifdef EVE_PP_STAMP
static int g_stamp_grid = 0;
#endif
static bool use_pingpong() { return knobs().pingpong; }
// `shared` (EVE_FFN_SHARED_GPU of the call's flags): other kernels run beside this launch, so no persistent grid
static bool use_persistent(bool shared) { return knobs().persist && !shared; }

// Few groups whose heights differ (rows[] given, at most 8 groups, one segment per expert): the two-group modality
// split of the non-MoE layers (12 608 vision rows beside 2 560 text rows at the cfg-2 shape).  Forward / dgrad tiles
// are then enumerated over the rows that exist (MfmaDesc::compact) and the weight gradients' K ranges are cut into
// equal slices of about (sum of rows) / kslots rows (MfmaDesc::kslots); with the plain enumeration the short
// group's XCDs idle while the long group's run 5 rounds of tiles instead of 3, and equal splits of the padded K
// range leave most of the short group's slices empty.
static bool few_unequal_groups(const int32_t* rows, int G, int E_local) {
  return knobs().compact && rows != nullptr && G == E_local && G >= 2 && G <= 8;
}
// slices aimed at: about two rounds of tiles for the CUs (both GEMMs share the launch)
static int wgrad_dyn_slots(int E_local, int M, int I) {
  const long tiles = 2L * (M / 256) * (I / 256);
  long n = 2L * cu_count() / tiles - (E_local - 1);
  if (n < E_local) n = E_local;
  if (n > 16) n = 16;
  return (int)n;
}

template <bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BIAS_GRAD, int BKI, int WR = 4>
static int launch_pp(const MfmaDesc& d, int act, hipStream_t st, bool shared, const MfmaDesc* d2 = nullptr) {
  constexpr int kEpi = 8 * 32 * EPI_LD * 4;
  constexpr int kLds = 5 * 256 * BKI * 2 > kEpi ? 5 * 256 * BKI * 2 : kEpi;
  auto blocks = [](const MfmaDesc& q) {
    const int ngroups = WGRAD ? q.E_local * (q.ksplit > 1 ? q.ksplit : 1) : q.G;
    return 8 * ceil_div(ngroups * (q.N / BN) * ceil_div(q.Mext, 64 * WR), 8);
  };
  MfmaPair pr;
  pr.d[0] = d;
  pr.d[1] = d2 ? *d2 : d;
  pr.nobias = 0;
  pr.n0 = blocks(d);
  pr.ntotal = pr.n0 + (d2 ? blocks(*d2) : 0);
  // persistent variant: one workgroup per CU walks the tiles (single K segment only).  The activation
  // epilogues run it with resynchronised E phases (kResync; EVE_FFN_PERSIST_ACT=0 selects the one-tile kernel
  // for A/B timing: the persistent one measured 2 % faster on those launches).
  const bool act_epi = EPI == MEPI_BIAS_ACT || EPI == MEPI_DACT;
  // (the persistent kernel keeps its tiles in a 64-entry per-lane table with packed fields, see grouped_gemm_ppp)
  const int pgrid = (cu_count() / 8) * 8;
  auto packs = [](const MfmaDesc& q) {
    return q.G < 16384 && q.G / q.E_local <= 64 && q.N / BN < 1024 && q.Mext <= 262144;
  };
  const bool persist = use_persistent(shared) && (!act_epi || knobs().persist_act) &&
                       EPI != MEPI_STORE_F32 && (!WGRAD || d.G == d.E_local) && pr.ntotal > cu_count() &&
                       pr.ntotal <= 64 * pgrid && packs(pr.d[0]) && packs(pr.d[1]);
  constexpr int kStage = 8 * 16 * EPI_LD * 4;
  constexpr int kLdsP = (5 * 256 * BKI * 2 + kStage > 160 * 1024) ? 5 * 256 * BKI * 2 : 5 * 256 * BKI * 2 + kStage;
  dim3 grid(persist ? (cu_count() / 8) * 8 : pr.ntotal);
#ifdef EVE_PP_STAMP
  if (persist && g_stamp_grid > 0) grid = dim3(g_stamp_grid);  // diagnostic: fewer CUs at work (tools/pp_stamps.hip)
#endif
#define EVE_PP_LAUNCH_C(A, C)                                                                                \
  do {                                                                                                       \
    auto kern = grouped_gemm_pp<A_COL, B_COL, EPI, A, WGRAD, BIAS_GRAD, BKI, WR, C>;                         \
    static PerDeviceOnce once, oncep;                                                                        \
    auto kernp = grouped_gemm_ppp<A_COL, B_COL, EPI, A, WGRAD, BIAS_GRAD, BKI, WR, C>;                       \
    if (persist) {                                                                                           \
      if (int rc_ = reserve_lds(oncep, reinterpret_cast<const void*>(kernp), kLdsP)) return rc_;             \
    } else {                                                                                                 \
      if (int rc_ = reserve_lds(once, reinterpret_cast<const void*>(kern), kLds)) return rc_;                \
    }                                                                                                        \
    if (persist) hipLaunchKernelGGL(kernp, grid, dim3(512), kLdsP, st, pr);                                  \
    else hipLaunchKernelGGL(kern, grid, dim3(512), kLds, st, pr);                                            \
  } while (0)
  // (the compact-enumeration variants exist for the forward / dgrad launches only)
#define EVE_PP_LAUNCH(A)                                                                                     \
  do {                                                                                                       \
    if constexpr (!WGRAD) {                                                                                  \
      if (d.compact) { EVE_PP_LAUNCH_C(A, true); break; }                                                    \
    }                                                                                                        \
    EVE_PP_LAUNCH_C(A, false);                                                                               \
  } while (0)
  if constexpr (EPI == MEPI_BIAS_ACT || EPI == MEPI_DACT) {
    switch (act) {
      case EVE_ACT_QUICK_GELU: EVE_PP_LAUNCH(EVE_ACT_QUICK_GELU); break;
      case EVE_ACT_GELU: EVE_PP_LAUNCH(EVE_ACT_GELU); break;
      case EVE_ACT_RELU: EVE_PP_LAUNCH(EVE_ACT_RELU); break;
      case EVE_ACT_IDENTITY: EVE_PP_LAUNCH(EVE_ACT_IDENTITY); break;
      default:
        set_error("ffn_mfma: unknown activation %d", act);
        return EVE_ERR_ARG;
    }
  } else {
    EVE_PP_LAUNCH(EVE_ACT_IDENTITY);
  }
#undef EVE_PP_LAUNCH_C
#undef EVE_PP_LAUNCH
  EVE_CHECK_LAUNCH("grouped_gemm_pp");
  return EVE_OK;
}

// Tile choice.  One workgroup fills a CU, so a launch costs about ceil(tiles / CUs) * rows-per-tile:
// 256 x 256 tiles (8 waves, ping-pong schedule) by default; 192 x 256 tiles when they need clearly
// fewer row-steps (e.g. 3 column tiles x 948 rows x 32 experts: 480 tiles of 192 rows fill two waves of
// 256 CUs, where 384 tiles of 256 rows leave the second wave half empty); 128 x 256 tiles (4 waves) when
// the row extent is small enough that a larger tile would be mostly padding.
struct TileChoice { int rows, spt; bool pingpong; };  // tile height, wave row blocks per tile, kernel family
template <bool A_COL, bool WGRAD>
static TileChoice choose_tile(const MfmaDesc& d) {
  const int force = knobs().tile_rows;  // debug knob: force 128, 192 or 256
  const bool small = !WGRAD && (force == 128 ||
                                (!force && (d.Mext <= 128 || (d.Mext % 256 != 0 && d.Mext % 256 <= 128 && d.Mext < 1024))));
  if (small) return {128, 1, false};
  if (!use_pingpong()) return {256, 2, false};
  if constexpr (!A_COL && !WGRAD) {
    const long cus = cu_count(), per = (long)d.G * (d.N / BN);
    const long cost256 = ceil_div(per * ceil_div(d.Mext, 256), cus) * 256;
    const long cost192 = ceil_div(per * ceil_div(d.Mext, 192), cus) * 192;
    const bool use192 = force ? force == 192 : cost192 * 10 < cost256 * 9;
    if (use192) return {192, 2, true};
  }
  return {256, 2, true};
}
template <bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BIAS_GRAD>
static int launch(const MfmaDesc& d, int act, hipStream_t st, bool shared) {
  const TileChoice tc = choose_tile<A_COL, WGRAD>(d);
  if (!tc.pingpong) {  // (the lock-step kernels keep the plain tile enumeration: MfmaDesc::compact is not read there)
    if (tc.rows == 128) return launch_cfg<128, 2, A_COL, B_COL, EPI, WGRAD, BIAS_GRAD>(d, act, st);
    return launch_cfg<256, 2, A_COL, B_COL, EPI, WGRAD, BIAS_GRAD>(d, act, st);
  }
  if constexpr (!A_COL && !WGRAD) {
    if (tc.rows == 192) return launch_pp<A_COL, B_COL, EPI, WGRAD, BIAS_GRAD, 64, 3>(d, act, st, shared);
  }
  return launch_pp<A_COL, B_COL, EPI, WGRAD, BIAS_GRAD, (A_COL && B_COL) ? 32 : 64>(d, act, st, shared);
}

int ffn_fwd_mfma(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* z, void* a,
                 void* y, const int32_t* rows, int G, int E_local, int R, int M, int I, int act, bool shared, hipStream_t st) {
  MfmaDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows; d.Mext = R;
  d.compact = few_unequal_groups(rows, G, E_local) && G == 2 ? 1 : 0;
  // z = x @ w1^T + b1 ; a = act(z)
  d.A = (const bf16_t*)x; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)w1; d.b_g = (long)I * M; d.ldb = M;
  d.C = (bf16_t*)z; d.C2 = (bf16_t*)a; d.c_g = (long)R * I; d.ldc = I; d.bias = (const bf16_t*)b1; d.N = I; d.K = M;
  if (int rc = launch<false, false, MEPI_BIAS_ACT, false, false>(d, act, st, shared)) return rc;
  // y = a @ w2^T + b2
  d.A = (const bf16_t*)a; d.a_g = (long)R * I; d.lda = I;
  d.B = (const bf16_t*)w2; d.b_g = (long)M * I; d.ldb = I;
  d.C = (bf16_t*)y; d.C2 = nullptr; d.c_g = (long)R * M; d.ldc = M; d.bias = (const bf16_t*)b2; d.N = M; d.K = I;
  return launch<false, false, MEPI_BIAS, false, false>(d, act, st, shared);
}

// both weight-gradient GEMMs of a backward pass in one launch of grouped_gemm_pp_wgrad
template <int EPI>
static int launch_pp_wgrad(const MfmaDesc& d, const MfmaDesc& d2, hipStream_t st, bool nobias) {
  constexpr int kLds = 5 * 256 * 64 * 2;
  auto blocks = [](const MfmaDesc& q) {  // (dynamic slices: at most kslots + E_local of them)
    const int groups = q.kslots > 0 ? q.kslots + q.E_local : q.E_local * (q.ksplit > 1 ? q.ksplit : 1);
    return 8 * ceil_div(groups * (q.N / BN) * ceil_div(q.Mext, 256), 8);
  };
  MfmaPair pr;
  pr.d[0] = d;
  pr.d[1] = d2;
  pr.nobias = nobias ? 1 : 0;
  pr.n0 = blocks(d);
  pr.ntotal = pr.n0 + blocks(d2);
  auto kern = grouped_gemm_pp_wgrad<EPI>;
  static PerDeviceOnce once;
  if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), kLds)) return rc;
  hipLaunchKernelGGL(kern, dim3(pr.ntotal), dim3(512), kLds, st, pr);
  EVE_CHECK_LAUNCH("grouped_gemm_pp_wgrad");
  return EVE_OK;
}

// Weight-gradient split-K factor: with few local experts (expert parallelism, the two-group modality split) the
// two dW GEMMs have too few 256 x 256 tiles for the CUs, each with a very long K loop; the K range is then cut
// into `ksplit` slices (whole segments first, then row slices) with fp32 partials and a reduction pass.  The
// partials cost HBM traffic (j x the weights' size in fp32, written and read), so the factor is the smallest that
// gives every CU a tile: measured at the cfg-2 dims (tools/ffn_only.py, EVE_FFN_KSPLIT), 1 expert: 0.39 / 0.28 /
// 0.25 / 0.26 ms backward for j = 1 / 2 / 4 / 8; 2 experts: 0.51 / 0.40 / 0.41 / 0.48; 4 experts: 0.69 / 0.71 /
// 0.75 / 0.86.
static int wgrad_ksplit(int G, int E_local, int M, int I) {
  const long tiles = 2L * E_local * (M / 256) * (I / 256);  // both GEMMs share a launch
  const long cus = cu_count();
  if (knobs().ksplit > 0) return knobs().ksplit;  // debug knob
  int j = 1;
  while (j < 8 && tiles * j < cus) j *= 2;
  const int nseg = G / E_local;
  if ((nseg & (nseg - 1)) != 0) return j <= nseg && nseg % j == 0 ? j : 1;  // odd segment counts: whole-segment splits only
  return j;
}

// Bias gradients of the ping-pong path are formed outside the weight-gradient GEMMs (whose bias-gradient tiles had
// to run a 32-deep body at about half the MFMA rate: a third of the dw1 tiles, a twelfth of the dw2 tiles):
//   db1 = column sums of dz: partial sums per wave row block, taken in the MEPI_DACT epilogue while dz is in
//         registers (fp32, before rounding), summed by bias_grads_kernel;
//   db2 = column sums of dy: the same launch, over the (L2 / Infinity-Cache resident) gradient rows.
// Partial slots per group: one per wave row block; 96 rows is the smallest wave row block of a ping-pong tile.
static size_t colsum_bytes(int G, int R, int I) { return align_up((size_t)G * (ceil_div(R, 96) + 1) * I * sizeof(float), 256); }

size_t ffn_bwd_mfma_workspace_bytes(int G, int E_local, int R, int M, int I) {
  size_t n = use_pingpong() ? colsum_bytes(G, R, I) : 0;
  const int j = wgrad_ksplit(G, E_local, M, I);
  size_t part = j > 1 ? (size_t)j * E_local * (2 * (size_t)M * I) * sizeof(float) : 0;
  if (knobs().compact && G == E_local && G >= 2 && G <= 8) {  // rows[] may be given: room for the dynamic slices as well
    const size_t dyn = (size_t)(wgrad_dyn_slots(E_local, M, I) + E_local) * (2 * (size_t)M * I) * sizeof(float);
    if (dyn > part) part = dyn;
  }
  return n + part;
}

int ffn_bwd_mfma(const void* dy, const void* x, const void* z, const void* a, const void* w1, const void* w2, void* dz,
                 void* dx, void* dw1, void* db1, void* dw2, void* db2, const int32_t* rows, int G, int E_local, int R,
                 int M, int I, int act, bool shared, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const bool pp = use_pingpong();
  const size_t need = ffn_bwd_mfma_workspace_bytes(G, E_local, R, M, I);
  if (need && (!workspace || workspace_bytes < need)) {
    set_error("eve_ffn_bwd: workspace %zu < %zu (eve_ffn_bwd_workspace_bytes)", workspace_bytes, need);
    return EVE_ERR_WORKSPACE;
  }
  if (need && ((uintptr_t)workspace & 255) != 0) {
    set_error("eve_ffn_bwd: workspace must be 256-byte aligned");
    return EVE_ERR_ARG;
  }
  float* colsum = pp ? static_cast<float*>(workspace) : nullptr;
  float* ksplit_ws = pp ? reinterpret_cast<float*>(static_cast<char*>(workspace) + colsum_bytes(G, R, I)) : nullptr;
  MfmaDesc d{};
  d.G = G; d.E_local = E_local; d.rows = rows;
  d.compact = few_unequal_groups(rows, G, E_local) && G == 2 ? 1 : 0;
  // dz = (dy @ w2) * act'(z): A = dy [R,M] row-major, B = w2 [M,I] k-major
  d.Mext = R;
  d.A = (const bf16_t*)dy; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)w2; d.b_g = (long)M * I; d.ldb = I;
  d.C = (bf16_t*)dz; d.c_g = (long)R * I; d.ldc = I; d.Zin = (const bf16_t*)z; d.N = I; d.K = M;
  const TileChoice tc = choose_tile<false, false>(d);
  const int wave_rows = tc.rows / tc.spt;
  if (pp) {
    d.colsum = colsum;
    d.colsum_slots = ceil_div(R, tc.rows) * tc.spt;
    if ((size_t)G * d.colsum_slots * I * sizeof(float) > colsum_bytes(G, R, I)) {
      set_error("eve_ffn_bwd: internal: column-sum partials exceed their workspace (tile %d rows)", tc.rows);
      return EVE_ERR_WORKSPACE;
    }
  }
  (void)wave_rows;
  if (int rc = launch<false, true, MEPI_DACT, false, false>(d, act, st, shared)) return rc;
  if (pp) {  // db1 from the partials, db2 from dy
    const int nb2 = ceil_div(M, 64) * E_local, nb1 = ceil_div(I, 1024) * E_local;
    hipLaunchKernelGGL(bias_grads_kernel, dim3(nb2 + nb1), dim3(1024), 0, st, (const bf16_t*)dy, (bf16_t*)db2, colsum,
                       (bf16_t*)db1, rows, G, E_local, R, M, I, d.colsum_slots, tc.rows, tc.spt, nb2);
    EVE_CHECK_LAUNCH("bias_grads_kernel");
  }
  d.colsum = nullptr;
  // dx = dz @ w1: A = dz [R,I] row-major, B = w1 [I,M] k-major
  d.A = (const bf16_t*)dz; d.a_g = (long)R * I; d.lda = I;
  d.B = (const bf16_t*)w1; d.b_g = (long)I * M; d.ldb = M;
  d.C = (bf16_t*)dx; d.c_g = (long)R * M; d.ldc = M; d.Zin = nullptr; d.N = M; d.K = I;
  if (int rc = launch<false, true, MEPI_STORE, false, false>(d, act, st, shared)) return rc;
  // dw2[e] = sum_g dy[g]^T @ a[g]  ([M,I]);  db2[e] = column sums of dy
  d.Mext = M;
  d.A = (const bf16_t*)dy; d.a_g = (long)R * M; d.lda = M;
  d.B = (const bf16_t*)a; d.b_g = (long)R * I; d.ldb = I;
  d.C = (bf16_t*)dw2; d.c_g = (long)M * I; d.ldc = I; d.N = I; d.K = R; d.dbias = (bf16_t*)db2;
  // dw1[e] = sum_g dz[g]^T @ x[g]  ([I,M]);  db1[e] = column sums of dz
  MfmaDesc d1 = d;
  d1.Mext = I;
  d1.A = (const bf16_t*)dz; d1.a_g = (long)R * I; d1.lda = I;
  d1.B = (const bf16_t*)x; d1.b_g = (long)R * M; d1.ldb = M;
  d1.C = (bf16_t*)dw1; d1.c_g = (long)I * M; d1.ldc = M; d1.N = M; d1.K = R; d1.dbias = (bf16_t*)db1;
  d.nfast = knobs().nfast && d.Mext > d.N ? 1 : 0;     // (share the larger operand's panel between consecutive tiles)
  d1.nfast = knobs().nfast && d1.Mext > d1.N ? 1 : 0;
  d.compact = d1.compact = 0;
  if (pp) {
    if (few_unequal_groups(rows, G, E_local)) {  // dynamic slices, see few_unequal_groups
      const int slots = wgrad_dyn_slots(E_local, M, I);
      const size_t nw = (size_t)M * I;
      float* p_dw2 = ksplit_ws;                               // [slots + E_local][M][I]
      float* p_dw1 = p_dw2 + (size_t)(slots + E_local) * nw;  // [slots + E_local][I][M]
      d.kslots = d1.kslots = slots;
      d.Cpart = p_dw2; d1.Cpart = p_dw1;
      if (int rc = launch_pp_wgrad<MEPI_STORE_F32>(d, d1, st, true)) return rc;
      hipLaunchKernelGGL(wgrad_reduce_dyn_kernel, dim3((unsigned)((nw / 8 + 255) / 256), E_local, 2), dim3(256), 0, st,
                         p_dw2, (bf16_t*)dw2, p_dw1, (bf16_t*)dw1, rows, E_local, R, slots, nw);
      EVE_CHECK_LAUNCH("wgrad_reduce_dyn_kernel");
      return EVE_OK;
    }
    const int ks = wgrad_ksplit(G, E_local, M, I);
    if (ks > 1) {
      const size_t nw = (size_t)E_local * M * I;
      float* p_dw2 = ksplit_ws;                // [ks][E_local][M][I]
      float* p_dw1 = p_dw2 + ks * nw;          // [ks][E_local][I][M]
      d.ksplit = ks; d.Cpart = p_dw2; d.dbias_part = nullptr;
      d1.ksplit = ks; d1.Cpart = p_dw1; d1.dbias_part = nullptr;
      if (int rc = launch_pp_wgrad<MEPI_STORE_F32>(d, d1, st, true)) return rc;
      auto reduce = [&](const float* part, void* out, size_t n) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, part, (bf16_t*)out, n, n, ks);
      };
      reduce(p_dw2, dw2, nw);
      reduce(p_dw1, dw1, nw);
      EVE_CHECK_LAUNCH("wgrad_reduce_kernel");
      return EVE_OK;
    }
#ifdef EVE_FFN_DEBUG_KNOBS
    if (!knobs().wgrad32 && knobs().wgrad_persist && G == E_local)  // A/B only: slower than one tile per workgroup
      return launch_pp<true, true, MEPI_STORE, true, false, 64>(d, act, st, shared, &d1);
#endif
    if (!knobs().wgrad32) return launch_pp_wgrad<MEPI_STORE>(d, d1, st, true);  // one launch for both GEMMs
    return launch_pp<true, true, MEPI_STORE, true, true, 32>(d, act, st, shared, &d1);  // debug knob: all tiles 32 deep (persistent)
  }
  if (int rc = launch<true, true, MEPI_STORE, true, true>(d, act, st, shared)) return rc;
  return launch<true, true, MEPI_STORE, true, true>(d1, act, st, shared);
}

}  // namespace eve
