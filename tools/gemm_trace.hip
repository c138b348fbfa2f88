// Cycle-stamp trace of one K-iteration of the 8-phase GEMM (block 0 and block 300, all 8 waves).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFX_GEMM_TRACE=1|2 -I arcflow_amd/csrc tools/gemm_trace.hip -o gpurun_out/gemm_trace
// Stamps per phase q (4 each): after barrier 1 | after lgkmcnt(0) | after the 16 MFMAs issued | after barrier 2; stamp 16 = loop top.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../arcflow_amd/csrc/afx_gemm.hip"

int main(int argc, char** argv) {
  const int M = 4608, N = argc > 1 ? atoi(argv[1]) : 9216, K = 3072;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);   // bf16 in +-[0.0078, 0.0156): random mantissas
  for (auto& v : hw) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
  uint16_t *a, *w, *c;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&c, (size_t)M * N * 2);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  afx::GemmBatch b{};
  b.nprob = 1;
  afx::GemmProblem& p = b.p[0];
  p.A = a; p.W = w; p.C = c; p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K; p.rows_per_batch = M;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) afx::launch_gemm(b, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) afx::launch_gemm(b, 0);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M=%d N=%d K=%d  %.1f us  %.0f TF (with stamps)\n", M, N, K, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
  unsigned tr[2][8][32];
  hipMemcpyFromSymbol(tr, HIP_SYMBOL(afx::g_gemm_trace), sizeof(tr));
  for (int blk = 0; blk < 2; ++blk) {
    printf("block %d: per phase [load+vmcnt+bar1 | lgkm | mfma issue | bar2]\n", blk ? 300 : 0);
    for (int wv = 0; wv < 8; ++wv) {
      unsigned* t = tr[blk][wv];
      unsigned prev = t[16];
      if (AFX_GEMM_TRACE != 1) goto coarse;
      printf(" w%d:", wv);
      for (int q = 0; q < 4; ++q) {
        printf("  %4u %4u %4u %4u |", t[q * 4] - prev, t[q * 4 + 1] - t[q * 4], t[q * 4 + 2] - t[q * 4 + 1], t[q * 4 + 3] - t[q * 4 + 2]);
        prev = t[q * 4 + 3];
      }
      printf("  iter %u\n", t[15] - t[16]);
    coarse:
      printf("      w%d kernel: prologue %u  loop %u (%u / K-tile)  drain %u  epilogue %u  total %u\n", wv, t[18] - t[17], t[19] - t[18],
             (t[19] - t[18]) / (K / 64), t[20] - t[19], t[21] - t[20], t[21] - t[17]);
    }
  }
  return 0;
}
