// Experiment (not part of the library): does the store-heavy dz launch run better BESIDE the weight-gradient GEMM of
// dw2 (which needs only dy and a, not dz) than before it?  Both as one-tile-per-workgroup grids on two streams, so
// that the dispatcher mixes their workgroups over the CUs, against the shipped sequence on one stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ieve_amd/csrc -Wno-unused-const-variable \
//         tools/overlap_exp.hip eve_amd/csrc/capi.hip -o tools/overlap_exp.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../eve_amd/csrc/ffn_mfma.hip"

using namespace eve;

int main() {
  const int G = 32, R = 948, M = 768, I = 3072;
  const size_t ndy = (size_t)G * R * M, na = (size_t)G * R * I, nw = (size_t)G * M * I;
  std::vector<unsigned short> h1(ndy), h2(na), h3(nw);
  srand(1);
  for (auto& v : h1) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& v : h2) v = 0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& v : h3) v = 0x3800 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  bf16_t *dy, *x, *a, *g, *dz, *dx, *w1, *w2, *dw1, *dw2;
  float* colsum;
  hipMalloc(&dy, ndy * 2); hipMalloc(&x, ndy * 2); hipMalloc(&dx, ndy * 2); hipMalloc(&a, na * 2); hipMalloc(&g, na * 2); hipMalloc(&dz, na * 2);
  hipMalloc(&w1, nw * 2); hipMalloc(&w2, nw * 2); hipMalloc(&dw1, nw * 2); hipMalloc(&dw2, nw * 2);
  hipMalloc(&colsum, (size_t)G * 16 * I * 4);
  hipMemcpy(dy, h1.data(), ndy * 2, hipMemcpyHostToDevice); hipMemcpy(x, h1.data(), ndy * 2, hipMemcpyHostToDevice);
  hipMemcpy(a, h2.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(g, h2.data(), na * 2, hipMemcpyHostToDevice);
  hipMemcpy(dz, h2.data(), na * 2, hipMemcpyHostToDevice);
  hipMemcpy(w1, h3.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(w2, h3.data(), nw * 2, hipMemcpyHostToDevice);
  // dz = (dy @ w2) * g
  MfmaDesc dzd{};
  dzd.G = G; dzd.E_local = G; dzd.rows = nullptr; dzd.Mext = R;
  dzd.A = dy; dzd.a_g = (long)R * M; dzd.lda = M; dzd.B = w2; dzd.b_g = (long)M * I; dzd.ldb = I;
  dzd.C = dz; dzd.c_g = (long)R * I; dzd.ldc = I; dzd.Zin = g; dzd.N = I; dzd.K = M;
  dzd.colsum = colsum; dzd.colsum_slots = 8;
  // dx = dz @ w1
  MfmaDesc dxd = dzd;
  dxd.colsum = nullptr; dxd.Zin = nullptr;
  dxd.A = dz; dxd.a_g = (long)R * I; dxd.lda = I; dxd.B = w1; dxd.b_g = (long)I * M; dxd.ldb = M;
  dxd.C = dx; dxd.c_g = (long)R * M; dxd.ldc = M; dxd.N = M; dxd.K = I;
  // dw2 = dy^T a, dw1 = dz^T x
  MfmaDesc d2{};
  d2.G = G; d2.E_local = G; d2.rows = nullptr;
  d2.Mext = M; d2.A = dy; d2.a_g = (long)R * M; d2.lda = M; d2.B = a; d2.b_g = (long)R * I; d2.ldb = I;
  d2.C = dw2; d2.c_g = (long)M * I; d2.ldc = I; d2.N = I; d2.K = R;
  MfmaDesc d1 = d2;
  d1.Mext = I; d1.A = dz; d1.a_g = (long)R * I; d1.lda = I; d1.B = x; d1.b_g = (long)R * M; d1.ldb = M;
  d1.C = dw1; d1.c_g = (long)I * M; d1.ldc = M; d1.N = M; d1.K = R;
  d2.nfast = d2.Mext > d2.N; d1.nfast = d1.Mext > d1.N;
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1, f0, f1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&f0, hipEventDisableTiming); hipEventCreateWithFlags(&f1, hipEventDisableTiming);
  auto seq = [&](hipStream_t s) {  // the shipped order: dz (persistent), dx (persistent), both weight gradients
    launch_pp<false, true, MEPI_DACT, false, false, 64>(dzd, EVE_ACT_QUICK_GELU, s, false);
    launch_pp<false, true, MEPI_STORE, false, false, 64, 3>(dxd, EVE_ACT_QUICK_GELU, s, false);
    launch_pp_wgrad<MEPI_STORE>(d2, d1, s, true);
  };
  // single-problem weight-gradient launch: both descriptors the same problem would double the work, so run d2 with an
  // empty second problem by giving launch_pp_wgrad the pair (p, p) and halving?  Simpler: launch_pp handles one problem.
  auto ovl = [&](bool shared_dz) {  // [dz || dw2] then [dx || dw1], one-tile grids so that the workgroups mix
    hipEventRecord(f0, sa);
    hipStreamWaitEvent(sb, f0, 0);
    launch_pp<false, true, MEPI_DACT, false, false, 64>(dzd, EVE_ACT_QUICK_GELU, sa, shared_dz);
    launch_pp<true, true, MEPI_STORE, true, false, 64>(d2, EVE_ACT_IDENTITY, sb, true);
    hipEventRecord(f1, sa);
    hipStreamWaitEvent(sb, f1, 0);  // dw1 needs dz
    launch_pp<false, true, MEPI_STORE, false, false, 64, 3>(dxd, EVE_ACT_QUICK_GELU, sa, shared_dz);
    launch_pp<true, true, MEPI_STORE, true, false, 64>(d1, EVE_ACT_IDENTITY, sb, true);
    hipEventRecord(f0, sb);
    hipStreamWaitEvent(sa, f0, 0);
  };
  auto time_it = [&](const char* name, auto fn) {
    for (int i = 0; i < 20; ++i) fn();
    hipStreamSynchronize(sa); hipStreamSynchronize(sb);
    hipEventRecord(e0, sa);
    for (int i = 0; i < 30; ++i) fn();
    hipEventRecord(e1, sa);
    hipEventSynchronize(e1);
    hipStreamSynchronize(sb);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f us per backward (%s)\n", name, ms * 1000 / 30, hipGetErrorString(hipGetLastError()));
  };
  for (int rep = 0; rep < 2; ++rep) {
    time_it("shipped: dz, dx persistent; dw2 + dw1 one launch; one stream", [&] { seq(sa); });
    time_it("[dz || dw2] then [dx || dw1], all one tile per workgroup", [&] { ovl(true); });
    time_it("[dz || dw2] then [dx || dw1], dz / dx persistent", [&] { ovl(false); });
  }
  return 0;
}
