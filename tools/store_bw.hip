// Per-CU global store throughput by access pattern (diagnostic for the GEMM epilogues).
//   hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o tools/store_bw && tools/store_bw
// A wave store instruction writes 64 lanes x 16 bytes = 1 KiB as SEG-byte runs in 1024/SEG different rows of a
// row-major bf16 matrix with row pitch 6144 bytes (the fc1 output).  Every workgroup (512 threads, one per CU) owns its
// own 256-row x 256-column tile area, as a GEMM epilogue does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SEG>
__global__ __launch_bounds__(512) void store_kernel(uint4* out, int reps, int pitch16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int ROWS = 64 / LPR;           // rows per instruction
  const int r = lane / LPR, c = lane % LPR;
  // tile origin: block b -> rows 256 * (b / 12), cols 256 * (b % 12) elements (512 bytes = 32 uint4)
  const size_t origin = (size_t)(blockIdx.x / 12) * 256 * pitch16 + (size_t)(blockIdx.x % 12) * 32;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int it = 0; it < reps; ++it) {
    // the block's 256 x 256 tile = 128 KiB = 128 instructions of 1 KiB; wave w issues 16 of them
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int instr = wave * 16 + k;                 // 0..127
      int row, col16;
      if (SEG == 128) {  // wave column strip of 64 columns (128 B), 8 rows per instruction (the current epilogue)
        const int wn = instr & 3, slab = instr >> 2;   // 32 slabs of 8 rows
        row = slab * 8 + r; col16 = wn * 8 + c;
      } else if (SEG == 512) {  // whole 256-column rows (512 B), 2 rows per instruction
        row = instr * 2 + r; col16 = c;
      } else {  // SEG == 256: half rows
        row = instr * 2 + (r >> 1); col16 = (r & 1) * 16 + c;
      }
      out[origin + (size_t)row * pitch16 + col16] = v;
    }
  }
}

int main() {
  const int pitch16 = 6144 / 16;
  const int nblk = 256;
  const size_t bytes = (size_t)((nblk + 11) / 12) * 256 * 6144 + (1 << 20);
  uint4* d;
  hipMalloc(&d, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 200;
  auto run = [&](auto kern, const char* name, int blocks) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, reps, pitch16);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, reps, pitch16);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double total = (double)blocks * reps * 128 * 1024;
    printf("%-28s blocks=%3d  %7.1f us  %7.1f GB/s total  %6.1f B/us/CU\n", name, blocks, ms * 1e3, total / ms / 1e6, total / blocks / (ms * 1e3));
  };
  for (int blocks : {256, 64, 8}) {
    run(store_kernel<128>, "8 rows x 128 B per instr", blocks);
    run(store_kernel<256>, "4 x 256 B per instr", blocks);
    run(store_kernel<512>, "2 rows x 512 B per instr", blocks);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
  return 0;
}
