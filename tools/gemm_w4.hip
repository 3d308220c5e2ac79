// Experiment (not part of the library): the forward N = 768 expert GEMM (y = a @ w2^T, K = 3072) with ONE wave per
// SIMD -- 4 waves per workgroup, each a 128 x 128 output tile (256 accumulator registers of the 512 a lone wave may
// use) -- instead of the library's 8 waves x (128 x 64).  A wave reads 4 + 4 fragments per 16 MFMAs instead of 4 + 2
// per 8: a third less LDS traffic per flop, which is what bounds the 192-row ping-pong tiles.
// There is no partner wave to overlap with: the wave interleaves the fragment reads of K sub-step ks + 1 with the
// MFMAs of sub-step ks itself (double-buffered fragments), and the LDS-DMA fetch of K step t + 1 runs beside step t.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ieve_amd/csrc -Wno-unused-const-variable \
//         tools/gemm_w4.hip eve_amd/csrc/capi.hip -o tools/gemm_w4 && tools/gemm_w4
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../eve_amd/csrc/ffn_mfma.hip"

using namespace eve;

constexpr int W4_BM = 256, W4_BN = 256, W4_STAGE = (W4_BM + W4_BN) * BK * 2;  // 64 KiB per K step

__global__ __launch_bounds__(256) void gemm_w4_kernel(MfmaDesc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (d.Mext + W4_BM - 1) / W4_BM, tiles_n = d.N / W4_BN;
  const int per_group = tiles_m * tiles_n;
  const int z = blockIdx.x / per_group, tile = blockIdx.x % per_group;
  const int m0 = (tile % tiles_m) * W4_BM, n0 = (tile / tiles_m) * W4_BN;
  const bf16_t* Ag = d.A + (size_t)z * d.a_g;
  const bf16_t* Bg = d.B + (size_t)(z % d.E_local) * d.b_g;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, (uint32_t)(d.Mext * d.lda) * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, (uint32_t)(d.N * d.ldb) * 2u, 0x00020000);
  uint32_t va[8], vb[8];
  tile_dma_offsets<false, W4_BM, 4>(d.lda, wave, lane, va);
  tile_dma_offsets<false, W4_BN, 4>(d.ldb, wave, lane, vb);
  const uint32_t oa = (uint32_t)(m0 * d.lda) * 2u, ob = (uint32_t)(n0 * d.ldb) * 2u;
  const int nk = d.K / BK;
  auto fetch = [&](int t) {
    char* st = smem + (t & 1) * W4_STAGE;
    tile_dma<W4_BM, 4>(ra, oa + (uint32_t)t * BK * 2u, va, st, wave);
    tile_dma<W4_BN, 4>(rb, ob + (uint32_t)t * BK * 2u, vb, st + W4_BM * BK * 2, wave);
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fetch(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 fa[2][4], fb[2][4];
  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + (t & 1) * W4_STAGE;
    const char* sb = sa + W4_BM * BK * 2;
#ifdef W4_NOREAD
    if (t == 0) {
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = frag_load<false, W4_BM>(sa, wm * 128 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = frag_load<false, W4_BN>(sb, wn * 128 + j * 32, 0, lane);
#ifdef W4_NOREAD
    }
#endif
#ifndef W4_NODMA
    fetch(t + 1);  // unconditional (past the last step the descriptor's extent makes it a zero fill): no branch in the step
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
#ifdef W4_NOREAD
      if (t == 0 && ks + 1 < 4) {  // diagnostic: the MFMA stream alone (stale fragments after the first step)
#else
      if (ks + 1 < 4) {
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[nxt][i] = frag_load<false, W4_BM>(sa, wm * 128 + i * 32, ks + 1, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nxt][j] = frag_load<false, W4_BN>(sb, wn * 128 + j * 32, ks + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
#ifdef W4_SCHED
      // 16 MFMAs and 8 LDS reads per sub-step: one read behind every second MFMA
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read: one LDS-DMA piece of the next K step
      }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // plain epilogue (prototype: scalar bf16 stores straight from the accumulators)
  bf16_t* Cg = d.C + (size_t)z * d.c_g;
  const int c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int n = n0 + wn * 128 + j * 32 + c;
#ifdef W4_NOSTORE
        if (m < d.Mext && acc[i][j][r] == 123456.789f) Cg[(size_t)m * d.ldc + n] = static_cast<bf16_t>(acc[i][j][r]);  // timing of the K loop alone
#else
        if (m < d.Mext) Cg[(size_t)m * d.ldc + n] = static_cast<bf16_t>(acc[i][j][r]);
#endif
      }
#endif
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
  hipMemset(C, 0, nc * 2); hipMemset(Cref, 0, nc * 2);
  MfmaDesc d{};
  d.G = G; d.E_local = G; d.rows = nullptr; d.Mext = R;
  d.A = A; d.a_g = (long)R * K; d.lda = K;
  d.B = B; d.b_g = (long)N * K; d.ldb = K;
  d.C = Cref; d.C2 = nullptr; d.c_g = (long)R * N; d.ldc = N; d.bias = bias; d.N = N; d.K = K;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // library kernel: 192-row persistent ping-pong
  for (int i = 0; i < 100; ++i) launch_pp<false, false, MEPI_BIAS, false, false, 64, 3>(d, EVE_ACT_IDENTITY, 0, false);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) launch_pp<false, false, MEPI_BIAS, false, false, 64, 3>(d, EVE_ACT_IDENTITY, 0, false);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * G * R * (double)N * K;
  printf("library 192-row ping-pong : %7.1f us  %6.0f TFLOP/s (%s)\n", ms * 1000 / 50, flops / (ms / 50 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
  d.C = C;
  const int tiles = G * ((R + W4_BM - 1) / W4_BM) * (N / W4_BN);
  const int lds = 2 * W4_STAGE;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
    printf("cannot reserve LDS\n");
    return 1;
  }
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(gemm_w4_kernel, dim3(tiles), dim3(256), lds, 0, d);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gemm_w4_kernel, dim3(tiles), dim3(256), lds, 0, d);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  hipEventElapsedTime(&ms, e0, e1);
  printf("4 waves x (128 x 128)     : %7.1f us  %6.0f TFLOP/s, %d tiles (%s)\n", ms * 1000 / 50, flops / (ms / 50 * 1e-3) / 1e12, tiles,
         hipGetErrorString(hipGetLastError()));
  std::vector<unsigned short> hc(nc), hr(nc);
  hipMemcpy(hc.data(), C, nc * 2, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), Cref, nc * 2, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < nc; ++i) bad += hc[i] != hr[i];
  printf("elements that differ from the library kernel's output: %zu of %zu\n", bad, nc);
  return 0;
}
