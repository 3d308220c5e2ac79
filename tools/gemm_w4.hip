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

// v2: a ring of ten 16 KiB sub-tiles (128 rows x 64 k: A top / A bottom / B left / B right of a K step, in that order),
// so that a wave's 16 LDS-DMA pieces per step are spread one per four MFMAs (scheduling fences pin the order) and a
// sub-tile has most of a step to land: during step t the pieces of B(t+1) go out first, then those of A(t+2).
constexpr int W4_SUB = 128 * BK * 2, W4_NSLOT = 10;
__global__ __launch_bounds__(256) void gemm_w4_kernel(MfmaDesc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (d.Mext + W4_BM - 1) / W4_BM, tiles_n = d.N / W4_BN;
  const int per_group = tiles_m * tiles_n;
#ifdef W4_PLAIN_ORDER
  const int item = blockIdx.x;
#else
  // the library's XCD-aware order: block b runs on XCD b % 8, each XCD walks a contiguous run of (expert, tile) items
  const int per_xcd = (d.G * per_group + 7) >> 3;
  const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || item >= d.G * per_group) return;
#endif
  const int z = item / per_group, tile = item % per_group;
  const int m0 = (tile % tiles_m) * W4_BM, n0 = (tile / tiles_m) * W4_BN;
  const bf16_t* Ag = d.A + (size_t)z * d.a_g;
  const bf16_t* Bg = d.B + (size_t)(z % d.E_local) * d.b_g;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ag), 0, (uint32_t)(d.Mext * d.lda) * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bg), 0, (uint32_t)(d.N * d.ldb) * 2u, 0x00020000);
  uint32_t va[4], vb[4];  // per-lane offsets of this wave's 4 pieces of a 128-row sub-tile
  tile_dma_offsets<false, 128, 4>(d.lda, wave, lane, va);
  tile_dma_offsets<false, 128, 4>(d.ldb, wave, lane, vb);
  const int nk = d.K / BK;
  // piece i (0..3) of sub-tile q of the sequence: q = 4 t + part, part 0 / 1 = A rows 0.. / 128.., 2 / 3 = B cols 0.. / 128..
  auto piece = [&](int q, int i) {
    const int t = q >> 2, part = q & 3;
    char* slot = smem + (q % W4_NSLOT) * W4_SUB;
#ifdef W4_SAMEK
    const uint32_t kofs = 0;  // diagnostic: every step fetches K tile 0 (hits in L2): is the fetch latency the cost?
#else
    const uint32_t kofs = (uint32_t)t * BK * 2u;
#endif
    if (part < 2) {
      const uint32_t org = (uint32_t)((m0 + part * 128) * d.lda) * 2u + kofs;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, va[i] + org, 0, 0, 0);
    } else {
      const uint32_t org = (uint32_t)((n0 + (part - 2) * 128) * d.ldb) * 2u + kofs;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, vb[i] + org, 0, 0, 0);
    }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int q = 0; q < 6; ++q)  // steps 0 and the A half of step 1; the steady state keeps 6 sub-tiles ahead
#pragma unroll
    for (int i = 0; i < 4; ++i) piece(q, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 fa[2][4], fb[2][4];
  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + ((4 * t + wm) % W4_NSLOT) * W4_SUB;       // my 128 rows of A(t)
    const char* sb = smem + ((4 * t + 2 + wn) % W4_NSLOT) * W4_SUB;   // my 128 columns of B(t)
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = frag_load<false, 128>(sa, i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = frag_load<false, 128>(sb, j * 32, 0, lane);
    const int qn = 4 * t + 6;  // sub-tiles fetched during this step: B(t+1) left / right, A(t+2) top / bottom
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        piece(qn + ks, g);  // past the end of K the descriptor's extent turns it into a zero fill
        if (ks + 1 < 4) {
          fa[nxt][g] = frag_load<false, 128>(sa, g * 32, ks + 1, lane);
          fb[nxt][g] = frag_load<false, 128>(sb, g * 32, ks + 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][g], fb[cur][j], acc[g][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but this step's last two sub-tiles (A(t+2)) has landed
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
  const int lds = W4_NSLOT * W4_SUB;
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
