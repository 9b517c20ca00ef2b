// Standalone timing of the fp32-MFMA GEMM kernels (kbj_gemm.h) on the shapes of the PPO update / rollout.
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ikbot-joystick_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "kbj_gemm.h"

thread_local kbj_ctx* kbj_prof_ctx = nullptr;
thread_local std::string kbj_global_error;
using namespace kbj;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool A_KC, bool B_KC>
double run(const char* name, int M, int N, int K, int splitk, int force_big, float* A, float* B, float* C, bool check) {
  GemmArgs g{A, B, C, nullptr, M, N, K, A_KC ? K : M, B_KC ? K : N, N, splitk > 1 ? 1 : 0, splitk, nullptr};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (splitk > 1) CK(hipMemset(C, 0, (size_t)M * N * 4));
  gemm_launch<A_KC, B_KC>(0, g, force_big);
  CK(hipDeviceSynchronize());
  double maxerr = 0;
  if (check) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), B, hB.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 200; ++t) {
      int m = rand() % M, n = rand() % N;
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)(A_KC ? hA[(size_t)m * K + k] : hA[(size_t)k * M + m]) * (B_KC ? hB[(size_t)n * K + k] : hB[(size_t)k * N + n]);
      maxerr = std::fmax(maxerr, std::fabs(s - hC[(size_t)m * N + n]) / (1 + std::fabs(s)));
    }
  }
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) gemm_launch<A_KC, B_KC>(0, g, force_big);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("%-28s M=%6d N=%5d K=%6d sk=%3d  %8.1f us  %6.1f TF", name, M, N, K, splitk, us, tf);
  if (check) printf("  maxrelerr %.2e", maxerr);
  printf("\n");
  return us;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : 0;   // 1..4: that shape alone, without the checks (for counter passes)
  size_t big = (size_t)51200 * 1024;
  float *A, *B, *C;
  CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&B, big * 4)); CK(hipMalloc(&C, big * 4));
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
  CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data() + 777, (big - 777) * 4, hipMemcpyHostToDevice));
  if (only) {
    if (only == 1) run<true, true>("fwd ih (R x 4H x H)", 51200, 1024, 256, 1, -1, A, B, C, false);
    if (only == 2) run<true, false>("dx (R x H x 4H)", 51200, 256, 1024, 1, -1, A, B, C, false);
    if (only == 3) run<false, false>("dW pair (4H x 2H x R) sk24", 1024, 512, 51200, 24, 1, A, B, C, false);
    if (only == 4) run<true, true>("square 4096", 4096, 4096, 4096, 1, -1, A, B, C, false);
    return 0;
  }
  run<true, true>("check fwd small", 300, 200, 100, 1, -1, A, B, C, true);
  run<true, false>("check dx small", 300, 200, 100, 1, -1, A, B, C, true);
  run<false, false>("check dW small", 200, 136, 3000, 4, -1, A, B, C, true);
  run<false, false>("check dW big-tile", 256, 256, 3000, 4, 1, A, B, C, true);
  {  // two k sources: C = [A | A2] [B | B2]^T must equal the single-source product over the concatenation
    int M = 500, N = 300, K1 = 64, K2 = 96;
    std::vector<float> hA((size_t)M * (K1 + K2)), hB((size_t)N * (K1 + K2)), hC((size_t)M * N);
    for (auto& v : hA) v = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
    for (auto& v : hB) v = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
    // device copies: A1 [M][K1] ld 96, A2 [M][K2] ld 96 (same ld), B likewise
    std::vector<float> a1((size_t)M * 96), a2((size_t)M * 96), b1((size_t)N * 96), b2((size_t)N * 96);
    for (int m = 0; m < M; ++m) { for (int k = 0; k < K1; ++k) a1[(size_t)m * 96 + k] = hA[(size_t)m * 160 + k]; for (int k = 0; k < K2; ++k) a2[(size_t)m * 96 + k] = hA[(size_t)m * 160 + K1 + k]; }
    for (int n = 0; n < N; ++n) { for (int k = 0; k < K1; ++k) b1[(size_t)n * 96 + k] = hB[(size_t)n * 160 + k]; for (int k = 0; k < K2; ++k) b2[(size_t)n * 96 + k] = hB[(size_t)n * 160 + K1 + k]; }
    float *dA1 = A, *dA2 = A + (size_t)M * 96, *dB1 = B, *dB2 = B + (size_t)N * 96;
    CK(hipMemcpy(dA1, a1.data(), a1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dA2, a2.data(), a2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    GemmArgs g{dA1, dB1, C, nullptr, M, N, K1 + K2, 96, 96, N, 0, 1, nullptr};
    g.A2 = dA2; g.B2 = dB2; g.k1 = K1;
    gemm_launch<true, true>(0, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < M; m += 7) for (int n = 0; n < N; n += 5) {
      double s = 0;
      for (int k = 0; k < K1 + K2; ++k) s += (double)hA[(size_t)m * 160 + k] * hB[(size_t)n * 160 + k];
      maxerr = std::fmax(maxerr, std::fabs(s - hC[(size_t)m * N + n]) / (1 + std::fabs(s)));
    }
    printf("check two k sources         M=%6d N=%5d K=%3d+%3d  maxrelerr %.2e\n", M, N, K1, K2, maxerr);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
    CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data() + 777, (big - 777) * 4, hipMemcpyHostToDevice));
  }
  run<true, true>("fwd ih (R x 4H x H)", 51200, 1024, 256, 1, -1, A, B, C, true);
  run<true, true>("fwd in critic (R x H x 475)", 51200, 256, 475, 1, -1, A, B, C, false);
  run<true, false>("dx (R x H x 4H)", 51200, 256, 1024, 1, -1, A, B, C, true);
  run<false, false>("dW (4H x H x R) sk96", 1024, 256, 51200, 96, 1, A, B, C, true);
  run<true, true>("rollout ih (N x 4H x H)", 8192, 1024, 256, 1, -1, A, B, C, false);
  run<true, true>("rollout half (N/2 x 4H x H)", 4096, 1024, 256, 1, -1, A, B, C, false);
  run<true, true>("square 4096", 4096, 4096, 4096, 1, -1, A, B, C, false);
  // the same product with the other operand layouts: what a row-contiguous operand (4 x ds_read_b32 per fragment) costs by itself
  run<true, false>("square 4096 (B row-contig)", 4096, 4096, 4096, 1, -1, A, B, C, false);
  run<false, false>("square 4096 (A, B row-contig)", 4096, 4096, 4096, 1, -1, A, B, C, false);
  // the in-situ shapes of the update's tail: long k slices (sk 16 / 32)
  run<false, false>("dW critic L0 (4H x 768 x R) sk16", 1024, 768, 51200, 16, 1, A, B, C, false);
  run<false, false>("dW actor L0 (4H x 384 x R) sk32", 1024, 384, 51200, 32, 1, A, B, C, false);
  return 0;
}
