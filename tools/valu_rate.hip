// Diagnostic: issue cost of the VALU instructions the GEMM epilogues are made of, at one and at two waves per
// SIMD (the epilogues run at two).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin
// Each test is 64 independent chains x 16 rounds of one instruction between two s_memtime stamps; reported:
// cycles per wave-instruction seen by one wave (so with two waves per SIMD the SIMD's rate is half that figure).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned* cyc, int rounds) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
  __builtin_amdgcn_s_barrier();
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
   for (int u = 0; u < 8; ++u) {
    if constexpr (KIND == 0) {  // v_mul_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, 0x3f7ff000, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 1) {  // v_pk_mul_f32
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 a = {v[i], v[i + 1]};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(a));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(a));
        v[i] = a.x; v[i + 1] = a.y;
      }
    } else if constexpr (KIND == 2) {  // v_exp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 3) {  // v_rcp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 4) {  // v_cvt_pk_bf16_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 5) {  // v_add_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 6) {  // v_pk_fma_f32
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 a = {v[i], v[i + 1]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(a));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(a));
        v[i] = a.x; v[i + 1] = a.y;
      }
    } else if constexpr (KIND == 7) {  // the quick-GELU chain per element: mul, exp, add, rcp, mul (scalar forms)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float t;
        asm volatile("v_mul_f32 %0, 0xc01d265f, %1" : "=v"(t) : "v"(v[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(t));
      }
    } else if constexpr (KIND == 8) {  // v_lshlrev_b32 (bf16 unpack)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i]));
    }
   }
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_round) {
  const int rounds = 16;
  for (int threads : {256, 512}) {
    float* out; unsigned* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, rounds);
    hipDeviceSynchronize();
    std::vector<unsigned> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0; int n = 256 * threads / 64;
    for (int i = 0; i < n; ++i) sum += h[i];
    printf("%-28s %d wave(s)/SIMD: %.2f cycles per wave-instruction\n", name, threads / 256, sum / n / rounds / per_round / 8);
    hipFree(out); hipFree(cyc);
  }
}

int main() {
  run<0>("v_mul_f32", 16);
  run<5>("v_add_f32", 16);
  run<8>("v_lshlrev_b32", 16);
  run<1>("v_pk_mul_f32", 16);
  run<6>("v_pk_fma_f32", 16);
  run<4>("v_cvt_pk_bf16_f32", 16);
  run<2>("v_exp_f32", 16);
  run<3>("v_rcp_f32", 16);
  run<7>("quick-GELU chain (5 instr)", 80);
  return 0;
}
