// Experiment (not part of the library), third version of tools/gemm_w4.hip: the forward N = 768 expert GEMM
// (y = a @ w2^T, K = 3072) with ONE wave per SIMD -- 4 waves, each a (32 WR) x 128 output tile -- and an operand ring
// cut along K: a ring slot is a 32-deep slice of the block tile's A and B operands (64-byte rows), five slots.  A slot is
// free as soon as every wave has read its slice's fragments, so the LDS-DMA fetch of slice s + 5 goes out at the start
// of slice s and has four slices (two 64-deep K steps) to land, where the 64-deep sub-tile rings of the first two
// versions gave the operand needed first only one step.  Tiles in the library's XCD-aware order.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ieve_amd/csrc -Wno-unused-const-variable \
//         tools/gemm_w4k.hip eve_amd/csrc/capi.hip -o tools/w4k.bin && tools/w4k.bin
//   -DW4_NOSTORE: K loop alone; -DW4_SAMEK: every slice fetches K slice 0 (all fetches hit in L2)
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../eve_amd/csrc/ffn_mfma.hip"

using namespace eve;

template <int WR>
struct W4K {
  static constexpr int BM = 64 * WR, SA = BM * 64, SB = 256 * 64, SLOT = SA + SB, NS = 5;
  static constexpr int PA = BM / 64, PB = 4, NPW = PA + PB;  // LDS-DMA pieces (1 KiB = 16 rows x 64 B) per wave and slice
  static constexpr int LDS = NS * SLOT;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WR>
__global__ __launch_bounds__(256) void gemm_w4k_kernel(MfmaDesc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C = W4K<WR>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (d.Mext + C::BM - 1) / C::BM, tiles_n = d.N / 256;
  const int per_group = tiles_m * tiles_n;
  const int per_xcd = (d.G * per_group + 7) >> 3;
  const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || item >= d.G * per_group) return;
  const int z = item / per_group, tile = item % per_group;
  const int m0 = (tile % tiles_m) * C::BM, n0 = (tile / tiles_m) * 256;
  const bf16_t* Ag = d.A + (size_t)z * d.a_g;
  const bf16_t* Bg = d.B + (size_t)(z % d.E_local) * d.b_g;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, (uint32_t)(d.Mext * d.lda) * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, (uint32_t)(d.N * d.ldb) * 2u, 0x00020000);
  // LDS image of a slice operand: row r at r * 64, its 16-byte chunk c stored at position c ^ ((r >> 2) & 3): a
  // fragment read (32 rows x 16 B, per 16-lane group 16 consecutive rows) then touches 16 different bank quads.
  // A DMA piece is 16 rows: lane -> (row lane >> 2, stored position lane & 3), fetching source chunk pos ^ swizzle.
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const uint32_t va0 = (uint32_t)(prow * d.lda + pchunk * 8) * 2u, vb0 = (uint32_t)(prow * d.ldb + pchunk * 8) * 2u;
  const int nsl = d.K / 32;
  auto issue = [&](int s) {  // my NPW pieces of slice s
    char* slot = smem + (s % C::NS) * C::SLOT;
#ifdef W4_SAMEK
    const uint32_t kofs = 0;
#else
    const uint32_t kofs = (uint32_t)s * 64u;
#endif
#pragma unroll
    for (int j = 0; j < C::PA; ++j) {
      const int P = j * 4 + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot + P * 1024), 16, va0,
                                               (uint32_t)((m0 + 16 * P) * d.lda) * 2u + kofs, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < C::PB; ++j) {
      const int P = j * 4 + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + C::SA + P * 1024), 16, vb0,
                                               (uint32_t)((n0 + 16 * P) * d.ldb) * 2u + kofs, 0, 0);
    }
  };
  // fragment of MFMA row block at row0 (multiple of 32), 16-deep sub-step q of the slice: lane (i, hh) reads chunk 2q + hh
  const int fi = lane & 31, fh = lane >> 5;
  const uint32_t foff0 = (uint32_t)(fi * 64 + (((0 + fh) ^ ((fi >> 2) & 3)) * 16));
  const uint32_t foff1 = (uint32_t)(fi * 64 + (((2 + fh) ^ ((fi >> 2) & 3)) * 16));
  f32x16 acc[WR][4];
#pragma unroll
  for (int i = 0; i < WR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[2][2][WR], fb[2][2][4];
  auto read_frags = [&](int s, int buf, int q, int g) {  // row block g of A (g < WR) and of B (g < 4), sub-step q
    const char* slot = smem + (s % C::NS) * C::SLOT;
    const uint32_t fo = q ? foff1 : foff0;
    if (g < WR) fa[buf][q][g] = *reinterpret_cast<const bf16x8*>(slot + (wm * WR + g) * 2048 + fo);
    if (g < 4) fb[buf][q][g] = *reinterpret_cast<const bf16x8*>(slot + C::SA + (wn * 4 + g) * 2048 + fo);
  };
  for (int s = 0; s < C::NS && s < nsl; ++s) issue(s);
  wait_vm<4 * C::NPW>();
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) read_frags(0, 0, q, g);
  auto body = [&](int s, auto bufc) {
    constexpr int cur = decltype(bufc)::value, nxt = cur ^ 1;
    const int rem = nsl - 1 - s;  // slices after this one
    // my pieces of slice s + 1 have landed (groups s + 1 .. s + 4 are out, fewer at the end of K)
    if (rem >= 4) wait_vm<3 * C::NPW>();
    else if (rem == 3) wait_vm<2 * C::NPW>();
    else if (rem == 2) wait_vm<C::NPW>();
    else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my fragment reads of slice s are done
    __builtin_amdgcn_s_barrier();                        // everyone's: slice s + 1 is complete, slot s % 5 is free
    __builtin_amdgcn_sched_barrier(0);
    if (s + C::NS < nsl) issue(s + C::NS);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        read_frags(s + 1, nxt, q, g);  // past the last slice: a harmless read of a stale slot
        __builtin_amdgcn_sched_barrier(0);
        if (g < WR) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][q][g], fb[cur][q][j], acc[g][j], 0, 0, 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < nsl; s += 2) {
    body(s, std::integral_constant<int, 0>{});
    body(s + 1, std::integral_constant<int, 1>{});
  }
  // plain epilogue (prototype: scalar bf16 stores straight from the accumulators)
  bf16_t* Cg = d.C + (size_t)z * d.c_g;
  const int c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < WR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WR + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int n = n0 + wn * 128 + j * 32 + c;
#ifdef W4_NOSTORE
        if (m < d.Mext && acc[i][j][r] == 123456.789f) Cg[(size_t)m * d.ldc + n] = static_cast<bf16_t>(acc[i][j][r]);  // timing of the K loop alone
#else
        if (m < d.Mext) Cg[(size_t)m * d.ldc + n] = static_cast<bf16_t>(acc[i][j][r]);
#endif
      }
#endif
}

template <int WR>
static void run(const MfmaDesc& d0, bf16_t* C, const std::vector<unsigned short>& hr, hipEvent_t e0, hipEvent_t e1, double flops) {
  MfmaDesc d = d0;
  d.C = C;
  const size_t nc = (size_t)d.G * d.Mext * d.N;
  hipMemset(C, 0, nc * 2);
  using K = W4K<WR>;
  const int tiles = 8 * ((d.G * ((d.Mext + K::BM - 1) / K::BM) * (d.N / 256) + 7) / 8);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4k_kernel<WR>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS) != hipSuccess) {
    printf("cannot reserve LDS\n");
    return;
  }
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(gemm_w4k_kernel<WR>, dim3(tiles), dim3(256), K::LDS, 0, d);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gemm_w4k_kernel<WR>, dim3(tiles), dim3(256), K::LDS, 0, d);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("4 waves x (%3d x 128), K-sliced ring : %7.1f us  %6.0f TFLOP/s, %d tiles (%s)\n", 32 * WR, ms * 1000 / 50,
         flops / (ms / 50 * 1e-3) / 1e12, tiles, hipGetErrorString(hipGetLastError()));
  std::vector<unsigned short> hc(nc);
  hipMemcpy(hc.data(), C, nc * 2, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < nc; ++i) bad += hc[i] != hr[i];
  printf("  elements that differ from the library kernel's output: %zu of %zu\n", bad, nc);
}

int main() {
  const int G = 32, R = 948, N = 768, K = 3072;
  const size_t na = (size_t)G * R * K, nb = (size_t)G * N * K, nc = (size_t)G * R * N;
  std::vector<unsigned short> ha(na), hb(nb);
  srand(1);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& v : hb) v = 0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  bf16_t *A, *B, *C, *Cref, *bias;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2); hipMalloc(&Cref, nc * 2); hipMalloc(&bias, (size_t)G * N * 2);
  hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)G * N * 2);
  hipMemset(Cref, 0, nc * 2);
  MfmaDesc d{};
  d.G = G; d.E_local = G; d.rows = nullptr; d.Mext = R;
  d.A = A; d.a_g = (long)R * K; d.lda = K;
  d.B = B; d.b_g = (long)N * K; d.ldb = K;
  d.C = Cref; d.C2 = nullptr; d.c_g = (long)R * N; d.ldc = N; d.bias = bias; d.N = N; d.K = K;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 100; ++i) launch_pp<false, false, MEPI_BIAS, false, false, 64, 3>(d, EVE_ACT_IDENTITY, 0, false);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) launch_pp<false, false, MEPI_BIAS, false, false, 64, 3>(d, EVE_ACT_IDENTITY, 0, false);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * G * R * (double)N * K;
  printf("library 192-row ping-pong            : %7.1f us  %6.0f TFLOP/s (%s)\n", ms * 1000 / 50, flops / (ms / 50 * 1e-3) / 1e12,
         hipGetErrorString(hipGetLastError()));
  std::vector<unsigned short> hr(nc);
  hipMemcpy(hr.data(), Cref, nc * 2, hipMemcpyDeviceToHost);
  run<4>(d, C, hr, e0, e1, flops);
  run<3>(d, C, hr, e0, e1, flops);
  return 0;
}
