// Diagnostic build of the ping-pong grouped GEMM with in-kernel cycle stamps (s_memtime) around the
// L / M phases of the first K tiles of two workgroups.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ieve_amd/csrc -DEVE_PP_STAMP -Wno-unused-const-variable \
//         tools/pp_stamps.hip eve_amd/csrc/capi.hip -o tools/pp_stamps
//   tools/pp_stamps x      per-tile / per-K-step timelines of the persistent kernel (what the library launches)
//   tools/pp_stamps p 128  the same with the persistent grid cut to 128 workgroups
//   tools/pp_stamps        the one-tile kernel's K-step stamps: only written when the launch takes that kernel, i.e. in a
//                          build with -DEVE_FFN_DEBUG_KNOBS and EVE_FFN_PERSIST=0 in the environment (zeros otherwise)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../eve_amd/csrc/ffn_mfma.hip"

using namespace eve;

template <bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BG, int BKI, int WR = 4>
static void run(const char* name, int G, int R, int N, int K, bool bcol, int same_tile = 0) {
  const size_t na = (size_t)G * R * K, nb = (size_t)G * N * K, nc = (size_t)G * R * N;
  std::vector<unsigned short> ha(na), hb(nb);
  srand(1);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // random bf16 around +-1
  for (auto& v : hb) v = 0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  bf16_t *A, *B, *C, *bias, *Z;
  unsigned long long* dbg;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2); hipMalloc(&Z, nc * 2); hipMalloc(&bias, (size_t)G * N * 2);
  hipMalloc(&dbg, 2 * 8 * 64 * 8);
  hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)G * N * 2); hipMemset(Z, 0, nc * 2);
  MfmaDesc d{};
  d.G = G; d.E_local = G; d.rows = nullptr; d.Mext = R;
  d.A = A; d.a_g = (long)R * K; d.lda = A_COL ? R : K;
  d.B = B; d.b_g = (long)N * K; d.ldb = bcol ? N : K;
  d.C = C; d.C2 = C; d.c_g = (long)R * N; d.ldc = N; d.bias = bias; d.Zin = Z; d.N = N; d.K = K;
  d.dbg = nullptr; d.same_tile = same_tile;
  for (int i = 0; i < 20; ++i) launch_pp<A_COL, B_COL, EPI, WGRAD, BG, BKI, WR>(d, EVE_ACT_QUICK_GELU, 0, false);
  hipDeviceSynchronize();
  hipMemset(dbg, 0, 2 * 8 * 64 * 8);
  d.dbg = dbg;
  launch_pp<A_COL, B_COL, EPI, WGRAD, BG, BKI, WR>(d, EVE_ACT_QUICK_GELU, 0, false);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * 8 * 64);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  printf("== %s  G=%d R=%d N=%d K=%d (%s)\n", name, G, R, N, K, hipGetErrorString(hipGetLastError()));
  for (int b = 0; b < 1; ++b)
    for (int w = 0; w < 8; w += 4) {
      printf("block %d wave %d: per tile  L(reads+dma+lgkm0) | barrier | M(issue) | vmcnt0 | barrier   [start rel. wave0 tile0]\n", b ? 1000 : 0, w);
      const unsigned long long* base0 = &h[(b * 8 + 0) * 64];
      {
        const unsigned long long* s = &h[(b * 8 + w) * 64];
        auto df = [](unsigned long long x, unsigned long long y) { return (long long)(int)((unsigned)x - (unsigned)y); };
        printf("   kernel start -> first L phase %lld ; epilogue (incl. store drain) %lld ; whole wave %lld\n", df(s[0], s[62]), df(s[61], s[60]), df(s[61], s[62]));
      }
      for (int t = 4; t < 6; ++t) {
        const unsigned long long* s = &h[(b * 8 + w) * 64 + t * 6];
        const unsigned long long* n = s + 6;
        auto df = [](unsigned long long x, unsigned long long y) { return (long long)(int)((unsigned)x - (unsigned)y); };
        printf("   t=%2d  reads %5lld dma %5lld | %6lld %6lld %6lld %6lld   [%lld]\n", t, df(s[5], s[0]), df(s[1], s[5]), df(s[2], s[1]), df(s[3], s[2]),
               df(s[4], s[3]), df(n[0], s[4]), df(s[0], base0[0]));
      }
    }
  hipFree(A); hipFree(B); hipFree(C); hipFree(Z); hipFree(bias); hipFree(dbg);
}

// persistent kernel (grouped_gemm_ppp): per-tile timeline of two workgroups, waves 0 and 4, and the clock held
template <bool A_COL, bool B_COL, int EPI, bool WGRAD, bool BG, int BKI, int WR = 4>
static void run_p(const char* name, int G, int R, int N, int K, bool bcol, int act, int ktile = 2) {
  const size_t na = (size_t)G * R * K, nb = (size_t)G * N * K, nc = (size_t)G * R * N;
  std::vector<unsigned short> ha(na), hb(nb);
  srand(1);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& v : hb) v = 0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  bf16_t *A, *B, *C, *C2, *bias, *Z;
  unsigned long long* dbg;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2); hipMalloc(&C2, nc * 2); hipMalloc(&Z, nc * 2); hipMalloc(&bias, (size_t)G * N * 2);
  hipMalloc(&dbg, 3 * 2 * 8 * 64 * 8);
  hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)G * N * 2); hipMemset(Z, 0, nc * 2);
  MfmaDesc d{};
  d.G = G; d.E_local = G; d.rows = nullptr; d.Mext = R;
  d.A = A; d.a_g = (long)R * K; d.lda = A_COL ? R : K;
  d.B = B; d.b_g = (long)N * K; d.ldb = bcol ? N : K;
  d.C = C; d.C2 = C2; d.c_g = (long)R * N; d.ldc = N; d.bias = bias; d.Zin = Z; d.N = N; d.K = K;
  d.dbg = nullptr; d.same_tile = ktile;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) launch_pp<A_COL, B_COL, EPI, WGRAD, BG, BKI, WR>(d, act, 0, false);  // reach the steady clock
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) launch_pp<A_COL, B_COL, EPI, WGRAD, BG, BKI, WR>(d, act, 0, false);
  hipEventRecord(e1, 0);
  hipMemset(dbg, 0, 3 * 2 * 8 * 64 * 8);
  d.dbg = dbg;
  launch_pp<A_COL, B_COL, EPI, WGRAD, BG, BKI, WR>(d, act, 0, false);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(3 * 2 * 8 * 64);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  auto df = [](unsigned long long x, unsigned long long y) { return (long long)(int)((unsigned)x - (unsigned)y); };
  printf("== persistent %s  G=%d R=%d N=%d K=%d act=%d: %.1f us per launch (%s)\n", name, G, R, N, K, act, ms * 1000 / 50, hipGetErrorString(hipGetLastError()));
  for (int b = 0; b < 2; ++b)
    for (int w = 0; w < 8; w += 4) {
      const unsigned long long* s = &h[(b * 8 + w) * 64];
      const double clk = (double)df(s[60], s[62]) / (double)df(s[61], s[63]) * 100.0;  // MHz
      printf(" block %d wave %d: whole kernel %lld cycles, clock %.0f MHz\n", b ? 100 : 0, w, df(s[60], s[62]), clk);
      for (int i = 0; i < 8; ++i) {
        const unsigned long long* q = s + 5 * i;
        if (!q[0] && !q[1]) break;
        printf("   tile %d: start %7lld | K loop %6lld | epilogue %6lld | barrier %6lld | -> next start %6lld\n", i, df(q[0], s[62]), df(q[1], q[0]),
               df(q[2], q[1]), df(q[3], q[2]), q[5] ? df(q[5], q[3]) : 0LL);
      }
      if (b == 0) {
        const unsigned long long* k = &h[1024 + (b * 8 + w) * 64];
        printf("   K steps of tile %d:  L(reads) | barrier | M(issue) | vmcnt0 | barrier\n", ktile);
        for (int t = 0; t < 10; ++t) {
          const unsigned long long* q = k + 6 * t;
          if (!q[0]) break;
          printf("     t=%2d  %5lld | %5lld | %5lld | %5lld | %5lld\n", t, df(q[1], q[0]), df(q[2], q[1]), df(q[3], q[2]), df(q[4], q[3]), df(q[5], q[4]));
        }
        const unsigned long long* e = &h[2048 + (b * 8 + w) * 64];
        if (e[0]) {
          printf("   epilogue of tile %d, per half slab:  reads+next stores issued and landed | conversions, activation, global stores  (first: bias wait %lld)\n", ktile, df(e[3], k[6 * 9 + 5] ? k[6 * 9 + 5] : e[3]));
          for (int q = 0; q < 8; ++q) {
            if (!e[4 * q]) break;
            printf("     q=%d  %5lld | %5lld   [start %lld after q=0]\n", q, df(e[4 * q + 1], e[4 * q]), df(e[4 * q + 2], e[4 * q + 1]), df(e[4 * q], e[0]));
          }
        }
      }
    }
  hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(Z); hipFree(bias); hipFree(dbg);
}

// the production weight-gradient launch (grouped_gemm_pp_wgrad: one tile per workgroup, both dW GEMMs): K-step
// timeline of two workgroups
static void run_w(int G, int R, int M, int I) {
  const size_t ndy = (size_t)G * R * M, na = (size_t)G * R * I;
  std::vector<unsigned short> h1(ndy), h2(na);
  srand(1);
  for (auto& v : h1) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& v : h2) v = 0x3a00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  bf16_t *dy, *x, *a, *dz, *dw1, *dw2;
  unsigned long long* dbg;
  hipMalloc(&dy, ndy * 2); hipMalloc(&x, ndy * 2); hipMalloc(&a, na * 2); hipMalloc(&dz, na * 2);
  hipMalloc(&dw1, (size_t)G * M * I * 2); hipMalloc(&dw2, (size_t)G * M * I * 2);
  hipMalloc(&dbg, 2 * 8 * 64 * 8);
  hipMemcpy(dy, h1.data(), ndy * 2, hipMemcpyHostToDevice); hipMemcpy(x, h1.data(), ndy * 2, hipMemcpyHostToDevice);
  hipMemcpy(a, h2.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dz, h2.data(), na * 2, hipMemcpyHostToDevice);
  MfmaDesc d{};
  d.G = G; d.E_local = G; d.rows = nullptr;
  d.Mext = M; d.A = dy; d.a_g = (long)R * M; d.lda = M; d.B = a; d.b_g = (long)R * I; d.ldb = I;
  d.C = dw2; d.c_g = (long)M * I; d.ldc = I; d.N = I; d.K = R;
  MfmaDesc d1 = d;
  d1.Mext = I; d1.A = dz; d1.a_g = (long)R * I; d1.lda = I; d1.B = x; d1.b_g = (long)R * M; d1.ldb = M;
  d1.C = dw1; d1.c_g = (long)I * M; d1.ldc = M; d1.N = M; d1.K = R;
  d.nfast = d.Mext > d.N; d1.nfast = d1.Mext > d1.N;
  d.dbg = d1.dbg = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 100; ++i) launch_pp_wgrad<MEPI_STORE>(d, d1, 0, true);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 30; ++i) launch_pp_wgrad<MEPI_STORE>(d, d1, 0, true);
  hipEventRecord(e1, 0);
  hipMemset(dbg, 0, 2 * 8 * 64 * 8);
  d.dbg = d1.dbg = dbg;
  launch_pp_wgrad<MEPI_STORE>(d, d1, 0, true);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * 8 * 64);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  auto df = [](unsigned long long p, unsigned long long q) { return (long long)(int)((unsigned)p - (unsigned)q); };
  printf("== weight gradients (one tile per workgroup) G=%d R=%d M=%d I=%d: %.1f us per launch (%s)\n", G, R, M, I, ms * 1000 / 30, hipGetErrorString(hipGetLastError()));
  for (int b = 0; b < 2; ++b)
    for (int w = 0; w < 8; w += 4) {
      const unsigned long long* q = &h[(b * 8 + w) * 64];
      printf(" block %d wave %d: start -> first L %lld | K loop %lld | epilogue + store drain %lld | whole %lld\n", b ? 1000 : 0, w, df(q[0], q[62]),
             df(q[60], q[0]), df(q[61], q[60]), df(q[61], q[62]));
      for (int t = 0; t < 10; ++t) {
        const unsigned long long* k = q + 6 * t;
        const unsigned long long* n = k + 6;
        if (!k[0]) break;
        printf("     t=%2d  L %5lld | %5lld | M %5lld | vmcnt %5lld | %5lld\n", t, df(k[1], k[0]), df(k[2], k[1]), df(k[3], k[2]), df(k[4], k[3]),
               t < 9 ? df(n[0], k[4]) : 0LL);
      }
    }
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'w') { run_w(32, 948, 768, 3072); return 0; }
  if (argc > 2) {  // "p <nwg>": fc1 with half the groups on <nwg> workgroups -- the same work per CU, fewer CUs storing
    g_stamp_grid = atoi(argv[2]);
    run_p<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1 bias+act, G=32 on the override grid", 32, 948, 3072, 768, false, EVE_ACT_QUICK_GELU);
    run_p<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1 bias+act, G=16 on the override grid", 16, 948, 3072, 768, false, EVE_ACT_QUICK_GELU);
    run_p<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1 bias+act, G=8 on the override grid", 8, 948, 3072, 768, false, EVE_ACT_QUICK_GELU);
    return 0;
  }
  if (argc > 1) {  // persistent-kernel timelines
    run_p<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1 bias+act", 32, 948, 3072, 768, false, EVE_ACT_QUICK_GELU);
    run_p<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1 bias+act, K steps of tile 0", 32, 948, 3072, 768, false, EVE_ACT_QUICK_GELU, 0);
    run_p<false, true, MEPI_DACT, false, false, 64>("dgrad2 act'", 32, 948, 3072, 768, true, EVE_ACT_QUICK_GELU);
    run_p<false, false, MEPI_BIAS, false, false, 64, 3>("fwd2 192-row", 32, 948, 768, 3072, false, EVE_ACT_QUICK_GELU);
    run_p<false, true, MEPI_STORE, false, false, 64, 3>("dgrad1 192-row", 32, 948, 768, 3072, true, EVE_ACT_QUICK_GELU);
    return 0;
  }
  run<false, false, MEPI_BIAS_ACT, false, false, 64>("fwd1  (row,row) N=3072 K=768, bias + quick_gelu epilogue", 32, 948, 3072, 768, false);
  run<false, true, MEPI_DACT, false, false, 64>("dgrad2 (row,col) N=3072 K=768, act' epilogue", 32, 948, 3072, 768, true);
  run<false, false, MEPI_BIAS, false, false, 64>("fwd2  (row,row) N=768 K=3072, 256-row tiles", 32, 948, 768, 3072, false);
  run<false, false, MEPI_BIAS, false, false, 64, 3>("fwd2  (row,row) N=768 K=3072, 192-row tiles", 32, 948, 768, 3072, false);
  run<false, true, MEPI_STORE, false, false, 64, 3>("dgrad1 (row,col) N=768 K=3072, 192-row tiles", 32, 948, 768, 3072, true);
  run<true, true, MEPI_STORE, true, false, 64>("wgrad (col,col) dw1: Mext=3072 N=768 K=948 rows, 64-deep", 32, 3072, 768, 948, true);
  run<true, true, MEPI_STORE, true, false, 32>("wgrad (col,col) dw1: Mext=3072 N=768 K=948 rows, 32-deep", 32, 3072, 768, 948, true);
  return 0;
}
