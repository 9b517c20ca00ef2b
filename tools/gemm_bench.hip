// Standalone timing of the fp32-MFMA GEMM kernels (kbj_gemm.h) on the shapes of the PPO update / rollout.
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ikbot-joystick_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include <utility>
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


// ---- PROTOTYPE (this tool only, not in libkbj.so): fp32 GEMM on the bf16 matrix cores through an EXACT three-way operand split ----
// x = hi + mid + lo with hi = x truncated to its top 8 significand bits (a bf16), mid = (x - hi) truncated likewise, lo = x - hi - mid:
// both subtractions are exact in fp32 and lo has at most 8 significant bits left, so the three bf16 pieces carry all 24 bits of x.
// a b = sum of the 9 piece products, each exact in the MFMA's fp32 accumulation; the 6-product form drops mid*lo, lo*mid, lo*lo
// (relative size <= 2^-23). `v_mfma_f32_32x32x16_bf16` runs at 16x the per-instruction rate of `v_mfma_f32_32x32x2_f32` and holds the
// vector issue port for 8 of its 32 cycles, so 6 / 9 of them per 16 k cost 0.375 / 0.56 of the fp32 form's time and leave the port free
// for the split itself. Operands are split ONCE per element, on the way into LDS (and + sub + pack: ~5.5 vector instructions per
// element against 128 multiply-adds it takes part in). 128 x 128 x 32 tiles on 4 wavefronts (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles);
// LDS: [piece][row][40] bf16 per operand (16-byte fragments), one stage + register prefetch, two workgroups per CU.
typedef short pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
constexpr int P3_LD = 40;                       // bf16 per LDS row (32 k + 8 pad: 80-byte rows keep ds_read_b128 of 32 rows conflict-free)
constexpr int P3_PIECE = 128 * P3_LD;           // bf16 per piece of one operand tile
__device__ __forceinline__ void split3(const f32x4& x, pu32x2& hi, pu32x2& mid, pu32x2& lo) {   // four consecutive k of one row -> 3 x (4 bf16)
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned xb = __float_as_uint(x[e]);
    h[e] = xb & 0xFFFF0000u;
    const float r1 = x[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m[e]);
    l[e] = __float_as_uint(r2);                 // <= 8 significant bits: its upper half is exact
  }
  hi = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
  mid = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  lo = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}
// one operand tile 128 rows x 32 k: every thread stages 16 elements as four (row, 4 consecutive k) groups
template <bool KC> struct P3Stage {
  f32x4 v[4];
  // k-contiguous [rows][ld]: thread = (row rr + 32 i, k 4 kq); row-contiguous [k][ld]: thread = (rows 4 rq .. 4 rq + 3, k 4 kg .. 4 kg + 3), transposed in registers
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int k0) {
    const int t = threadIdx.x;
    if (KC) { const int kq = t & 7, rr = t >> 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(P + (size_t)(r0 + rr + 32 * i) * ld + k0 + 4 * kq);
    } else { const int rq = t & 31, kg = t >> 5;
      f32x4 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f32x4*>(P + (size_t)(k0 + 4 * kg + j) * ld + r0 + 4 * rq);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = {w[0][i], w[1][i], w[2][i], w[3][i]};      // row 4 rq + i, k 4 kg .. + 3
    }
  }
  __device__ __forceinline__ void store(short* lds) const {    // lds: [3][128][P3_LD]
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = KC ? (t >> 3) + 32 * i : 4 * (t & 31) + i, k = KC ? 4 * (t & 7) : 4 * (t >> 5);
      pu32x2 hi, mid, lo;
      split3(v[i], hi, mid, lo);
      short* p = lds + row * P3_LD + k;
      *reinterpret_cast<pu32x2*>(p) = hi; *reinterpret_cast<pu32x2*>(p + P3_PIECE) = mid; *reinterpret_cast<pu32x2*>(p + 2 * P3_PIECE) = lo;
    }
  }
};
template <int NPROD, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                             int lda, int ldb, int ldc, int splitk) {
  extern __shared__ __attribute__((aligned(16))) short p3lds[];   // A: 3 pieces, then B: 3 pieces
  short* As = p3lds; short* Bs = p3lds + 3 * P3_PIECE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int tiles_n = N / 128, tiles_m = M / 128;
  const int per = splitk > 1 ? ((K + splitk - 1) / splitk + 31) / 32 * 32 : K;
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);     // XCD-contiguous hand-out as in kbj_gemm.h
  if (item >= tiles_n * tiles_m * (splitk > 1 ? splitk : 1)) return;
  const int tn = item % tiles_n, tm = (item / tiles_n) % tiles_m, ks = item / (tiles_n * tiles_m);
  const int m0 = tm * 128, n0 = tn * 128, kbeg = ks * per, kend = min(K, kbeg + per);
  if (kbeg >= kend) return;
  P3Stage<A_KC> sa; P3Stage<B_KC> sb;
  sa.load(A, lda, m0, kbeg); sb.load(B, ldb, n0, kbeg);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    __syncthreads();                       // every wavefront has read the previous tile
    sa.store(As); sb.store(Bs);
    __syncthreads();
    if (k0 + 32 < kend) { sa.load(A, lda, m0, k0 + 32); sb.load(B, ldb, n0, k0 + 32); }     // in flight behind the MFMAs
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      pbf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = *reinterpret_cast<const pbf16x8*>(As + p * P3_PIECE + (wr * 64 + 32 * i + lr) * P3_LD + 16 * kk + 8 * lh);
          b[p][i] = *reinterpret_cast<const pbf16x8*>(Bs + p * P3_PIECE + (wc * 64 + 32 * i + lr) * P3_LD + 16 * kk + 8 * lh);
        }
      // smallest products first; pa / pb = piece of A / B (0 hi, 1 mid, 2 lo)
      constexpr int PA[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0}, PB[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0};   // lo*lo, mid*lo, lo*mid | mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
#pragma unroll
      for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + 32 * j + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* c = C + (size_t)m * ldc + n;
        if (splitk > 1) atomicAdd(c, acc[i][j][r]); else *c = acc[i][j][r];
      }
    }
}
template <int NPROD, bool A_KC, bool B_KC> void launch_x3(const float* A, const float* B, float* C, int M, int N, int K, int splitk) {
  constexpr size_t bytes = 6 * P3_PIECE * sizeof(short);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_kernel<NPROD, A_KC, B_KC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)attr;
  int wgs = (M / 128) * (N / 128) * (splitk > 1 ? splitk : 1);
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_bf16x3_kernel<NPROD, A_KC, B_KC>), dim3(wgs), dim3(256), bytes, 0, A, B, C, M, N, K, A_KC ? K : M, B_KC ? K : N, N, splitk);
}
// error study: n sampled outputs against an fp64 dot product of the SAME fp32 operands; relative to 1 + |exact| (the existing column) and to
// sum |a||b| (the scale rounding errors of a dot product live on)
struct ErrStat { double max1 = 0, p999_1 = 0, maxs = 0, p999_s = 0, med_s = 0; };
template <bool A_KC, bool B_KC> ErrStat err_study(const std::vector<float>& hA, const std::vector<float>& hB, const float* C, int M, int N, int K, int n) {
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> e1(n), es(n);
  srand(12345);
  for (int t = 0; t < n; ++t) {
    const int m = rand() % M, c = rand() % N;
    double s = 0, sa = 0;
    for (int k = 0; k < K; ++k) {
      const double a = A_KC ? hA[(size_t)m * K + k] : hA[(size_t)k * M + m], b = B_KC ? hB[(size_t)c * K + k] : hB[(size_t)k * N + c];
      s += a * b; sa += std::fabs(a * b);
    }
    const double d = std::fabs(s - hC[(size_t)m * N + c]);
    e1[t] = d / (1 + std::fabs(s)); es[t] = d / sa;
  }
  std::sort(e1.begin(), e1.end()); std::sort(es.begin(), es.end());
  ErrStat r; r.max1 = e1.back(); r.p999_1 = e1[(size_t)(0.999 * (n - 1))]; r.maxs = es.back(); r.p999_s = es[(size_t)(0.999 * (n - 1))]; r.med_s = es[n / 2];
  return r;
}
template <bool A_KC, bool B_KC> void compare_x3(const char* name, int M, int N, int K, int splitk, float* A, float* B, float* C, int samples) {
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), B, hB.size() * 4, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto&& fn) {
    if (splitk > 1) CK(hipMemset(C, 0, (size_t)M * N * 4));
    fn(); CK(hipDeviceSynchronize());
    ErrStat st = err_study<A_KC, B_KC>(hA, hB, C, M, N, K, samples);
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) fn();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return std::make_pair(ms * 1e3 / reps, st);
  };
  GemmArgs g{A, B, C, nullptr, M, N, K, A_KC ? K : M, B_KC ? K : N, N, splitk > 1 ? 1 : 0, splitk, nullptr};
  auto ex = timeit([&] { gemm_launch<A_KC, B_KC>(0, g, 1); });
  GemmArgs gx = g; gx.x3 = 1;     // the product kernel (kbj_gemm.h gemm_x3_kernel, kbj_config.gemm_bf16x3)
  auto x6 = timeit([&] { gemm_launch<A_KC, B_KC>(0, gx, 1); });
  auto x9 = timeit([&] { launch_x3<9, A_KC, B_KC>(A, B, C, M, N, K, splitk); });
  auto row = [&](const char* what, std::pair<double, ErrStat>& r) {
    printf("  %-34s %8.1f us %6.1f TF   err/(1+|c|): max %.2e p99.9 %.2e   err/sum|ab|: max %.2e p99.9 %.2e median %.2e\n", what, r.first,
           2.0 * M * N * K / (r.first * 1e-6) / 1e12, r.second.max1, r.second.p999_1, r.second.maxs, r.second.p999_s, r.second.med_s);
  };
  printf("%s  M=%d N=%d K=%d split-K %d  (%d sampled outputs against fp64)\n", name, M, N, K, splitk, samples);
  row("exact fp32 MFMA (kbj_gemm.h)", ex); row("bf16 x3, 6 products (product kernel)", x6); row("bf16 x3 split, 9 products", x9);
}


// ---- mode 11: the update's TAIL - the four paired weight-gradient launches of a minibatch, together on four streams as kbj_ppo_grad queues them -------
// (critic layer 0: dG0^T [Hm | obs476], actor layer 0: dG0^T [Hm | obs68], critic / actor layer 1: dG1^T [Hm | X]; R = 51200 samples, H = 256.)
// Prints the time of the four together and their joint rate for a tile configuration <MT, NT, WM, WN> and a split-K workgroup target.
struct TailProblem { const float* dG; const float* Hm; const float* X; int nx, ldx; float* dW; float* Z; };
template <int MT, int NT, int WM, int WN>
double tail4(const char* name, const TailProblem (&pr)[4], hipStream_t (&st)[4], int wg_target, int reps, bool report = true) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  const int R = 51200, H = 256, M = 4 * H;
  GemmArgs g[4]; int wgs[4]; double flops = 0;
  for (int p = 0; p < 4; ++p) {
    const int N = H + pr[p].nx;
    const int tiles = ((M + BM - 1) / BM) * (H / BN + (pr[p].nx + BN - 1) / BN);
    int sk = std::max(2, std::min(256, wg_target / std::max(1, tiles)));
    sk = std::max(2, std::min(sk, (R + 255) / 256));
    g[p] = GemmArgs{pr[p].dG, pr[p].Hm, pr[p].dW, nullptr, M, N, R, M, H, H, 1, sk, nullptr};
    g[p].B2 = pr[p].X; g[p].C2 = pr[p].Z; g[p].n1 = H; g[p].ldb2 = pr[p].ldx; g[p].ldc2 = pr[p].ldx;
    wgs[p] = (tiles * sk + 7) / 8 * 8;
    flops += 2.0 * M * N * R;
  }
  auto go = [&]() { for (int p = 0; p < 4; ++p) gemm_launch_tile<MT, NT, false, false, WM, WN>(st[p], g[p], wgs[p]); };
  static_assert(256 % BN == 0, "the column split (n1 = H = 256) must fall on a tile boundary");
  go();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1[4]; CK(hipEventCreate(&e0));
  for (int p = 0; p < 4; ++p) CK(hipEventCreate(&e1[p]));
  CK(hipEventRecord(e0, st[0]));
  for (int p = 1; p < 4; ++p) CK(hipStreamWaitEvent(st[p], e0, 0));
  for (int r = 0; r < reps; ++r) go();
  for (int p = 0; p < 4; ++p) CK(hipEventRecord(e1[p], st[p]));
  CK(hipDeviceSynchronize());
  float ms = 0;
  for (int p = 0; p < 4; ++p) { float m; CK(hipEventElapsedTime(&m, e0, e1[p])); ms = std::max(ms, m); }
  const double us = ms * 1e3 / reps;
  if (report) printf("tail4 %-34s tile %3d x %3d, %d waves, target %4d wgs (sk %d/%d/%d/%d, %d wgs)  %8.1f us  %6.1f TF together\n", name, BM, BN, WM * WN, wg_target,
                     g[0].splitk, g[1].splitk, g[2].splitk, g[3].splitk, wgs[0] + wgs[1] + wgs[2] + wgs[3], us, flops / (us * 1e-6) / 1e12);
  return us;
}


// ---- mode 12: one k-contiguous product (the critic's input projection R x H x 476 at the head of a minibatch's critical chain) by tile configuration ----
template <int MT, int NT, int WM, int WN, bool B_KC = true>
void one_kc(const char* name, int M, int N, int K, float* A, float* B, float* C) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  GemmArgs g{A, B, C, nullptr, M, N, K, K, B_KC ? K : N, N, 0, 1, nullptr};
  const int wgs = (((M + BM - 1) / BM) * ((N + BN - 1) / BN) + 7) / 8 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gemm_launch_tile<MT, NT, true, B_KC, WM, WN>(0, g, wgs);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) gemm_launch_tile<MT, NT, true, B_KC, WM, WN>(0, g, wgs);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-36s tile %3d x %3d, %d waves, %5d wgs  M=%d N=%d K=%d  %8.1f us  %6.1f TF\n", name, BM, BN, WM * WN, wgs, M, N, K, ms * 1e3 / reps, 2.0 * M * N * K / (ms * 1e-3 / reps) / 1e12);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : 0;   // 1..4: that shape alone, without the checks (for counter passes)
  size_t big = (size_t)51200 * 1024;
  float *A, *B, *C;
  CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&B, big * 4)); CK(hipMalloc(&C, big * 4));
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
  CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data() + 777, (big - 777) * 4, hipMemcpyHostToDevice));
  if (only == 9) {   // the bf16-split prototype beside the exact kernel on the update's shapes: TF and error against fp64
    compare_x3<true, true>("forward  [x|h] W^T", 51200, 1024, 256, 1, A, B, C, 20000);
    compare_x3<true, false>("input gradient dG W", 51200, 256, 1024, 1, A, B, C, 20000);
    compare_x3<false, false>("weight gradient dG^T [h|x] (paired)", 1024, 512, 51200, 24, A, B, C, 4000);
    compare_x3<true, true>("square", 4096, 4096, 4096, 1, A, B, C, 8000);
    return 0;
  }
  if (only == 10) {   // operand range of the PRODUCT split kernel (kbj_config.gemm_bf16x3): the three-way split is exact only while hi, mid and lo are
    // normal bf16 numbers. bf16 has fp32's exponent range, mid sits 8 and lo 16 binades below the element: an element below 2^-110 loses its lo
    // piece to the bf16 subnormal range (2^-118: mid as well), whatever the matrix cores do with subnormal inputs. The test scales A by 2^ea and
    // B by 2^eb (elements uniform in [-1, 1)), keeps the products inside fp32's range, and compares err / sum|a b| of sampled outputs - a
    // scale-free figure - with the exact fp32-MFMA kernel on the same operands. PASS: inside 2^-100 .. 2^100 the split path is as accurate as on
    // unit-scale operands (max <= 2 x the exact kernel's, p99.9 <= 1.25 x); at 2^-120 (either operand) the loss is bounded by the dropped pieces
    // (<= 2^-7 of the element if mid goes too; measured on MI355X: 1e-5 = 2^-16.5, i.e. the lo piece is lost, and 2^-110 is still exact) -
    // documented, not silent. Magnitudes that small do not occur on this path: activations, gate derivatives and weights are O(1e-6 .. 1e2).
    const int M = 1024, N = 512, K = 2048, samples = 4000;
    std::vector<float> a0((size_t)M * K), b0((size_t)N * K), ha(a0.size()), hb(b0.size());
    for (auto& v : a0) v = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
    for (auto& v : b0) v = (float)((rand() & 0xFFFF) - 32768) / 32768.0f;
    struct Case { int ea, eb; bool strict; };
    const Case cases[] = {{0, 0, true}, {-60, 60, true}, {60, -60, true}, {-100, 0, true}, {-100, 100, true}, {100, -100, true}, {0, -100, true}, {60, 40, true},
                          {-110, 0, false}, {-120, 0, false}, {-120, 120, false}, {120, -120, false}};
    int bad = 0;
    printf("gemm_x3_kernel operand range: M=%d N=%d K=%d, A k-contiguous, B row-contiguous (input-gradient layout) and both row-contiguous, split-K 4 (weight-gradient layout)\n", M, N, K);
    for (const Case& c : cases) {
      for (size_t i = 0; i < a0.size(); ++i) ha[i] = std::ldexp(a0[i], c.ea);
      for (size_t i = 0; i < b0.size(); ++i) hb[i] = std::ldexp(b0[i], c.eb);
      CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
      for (int layout = 0; layout < 2; ++layout) {
        ErrStat st[2];
        for (int x3 = 0; x3 < 2; ++x3) {
          if (layout == 0) {   // C[M][N] = A[M][K] B[K][N]
            GemmArgs g{A, B, C, nullptr, M, N, K, K, N, N, 0, 1, nullptr};
            g.x3 = x3;
            gemm_launch<true, false>(0, g, 1);
            CK(hipDeviceSynchronize());
            st[x3] = err_study<true, false>(ha, hb, C, M, N, K, samples);
          } else {             // C[M'][N'] = A^T B over k = the long axis: A[K'][M'], B[K'][N'] with K' = M (1024 rows), split over k
            const int M2 = 512, N2 = 512, K2 = 1024;      // reinterpret the buffers: A as [K2][M2 .. ld K], B as [K2][N2 .. ld N]
            GemmArgs g{A, B, C, nullptr, M2, N2, K2, K, N, N2, 1, 4, nullptr};
            g.x3 = x3;
            CK(hipMemset(C, 0, (size_t)M2 * N2 * 4));
            gemm_launch<false, false>(0, g, 1);
            CK(hipDeviceSynchronize());
            // reference on the same view
            std::vector<float> va((size_t)K2 * M2), vb((size_t)K2 * N2);
            for (int k = 0; k < K2; ++k) { for (int m = 0; m < M2; ++m) va[(size_t)k * M2 + m] = ha[(size_t)k * K + m]; for (int n = 0; n < N2; ++n) vb[(size_t)k * N2 + n] = hb[(size_t)k * N + n]; }
            st[x3] = err_study<false, false>(va, vb, C, M2, N2, K2, samples);
          }
        }
        const bool ok = c.strict ? (st[1].maxs <= 2.0 * st[0].maxs + 1e-9 && st[1].p999_s <= 1.25 * st[0].p999_s + 1e-9) : (st[1].maxs <= 1.0 / 128);
        printf("  A x 2^%-4d B x 2^%-4d %-14s exact: max %.2e p99.9 %.2e | split: max %.2e p99.9 %.2e median %.2e  %s%s\n", c.ea, c.eb, layout ? "dG^T x (sk 4)" : "dG W", st[0].maxs,
               st[0].p999_s, st[1].maxs, st[1].p999_s, st[1].med_s, ok ? "ok" : "FAIL", c.strict ? "" : " (below 2^-110: pieces in the bf16 subnormal range, bounded loss)");
        bad += !ok;
      }
    }
    printf(bad ? "X3 RANGE TEST FAILED (%d)\n" : "X3 RANGE TEST PASSED\n", bad);
    return bad ? 1 : 0;
  }
  if (only == 11) {   // the update's tail: four paired weight-gradient launches together, by tile configuration and split-K target
    const size_t R = 51200, H = 256;
    auto dalloc = [&](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), std::min(n, big) * 4, hipMemcpyHostToDevice)); return p; };
    TailProblem pr[4];
    const int nx[4] = {475, 65, 256, 256}, ldx[4] = {476, 68, 256, 256};
    for (int p = 0; p < 4; ++p) {
      pr[p].dG = dalloc(R * 4 * H); pr[p].Hm = dalloc(R * H); pr[p].X = dalloc(R * ldx[p]); pr[p].nx = nx[p]; pr[p].ldx = ldx[p];
      float *w, *z; CK(hipMalloc(&w, 4 * H * H * 4)); CK(hipMalloc(&z, 4 * H * 512 * 4)); CK(hipMemset(w, 0, 4 * H * H * 4)); CK(hipMemset(z, 0, 4 * H * 512 * 4));
      pr[p].dW = w; pr[p].Z = z;
    }
    hipStream_t st[4];
    for (int p = 0; p < 4; ++p) CK(hipStreamCreateWithFlags(&st[p], hipStreamNonBlocking));
    const int reps = 5;
    for (int pass = 0; pass < 2; ++pass) {
      for (int tgt : {512, 768, 1024}) tail4<2, 1, 2, 4>("128x128 on 8 waves (product)", pr, st, tgt, reps);
      for (int tgt : {512, 768, 1024}) tail4<2, 2, 2, 2>("128x128 on 4 waves (64x64 each)", pr, st, tgt, reps);
      for (int tgt : {256, 384, 512, 768}) tail4<2, 2, 4, 2>("256x128 on 8 waves (64x64 each)", pr, st, tgt, reps);
      for (int tgt : {256, 384, 512, 768}) tail4<2, 2, 2, 4>("128x256 on 8 waves (64x64 each)", pr, st, tgt, reps);
      for (int tgt : {256, 384, 512, 768}) tail4<4, 1, 2, 4>("256x128 on 8 waves (128x32 each)", pr, st, tgt, reps);
      for (int tgt : {512, 768, 1024, 1536}) tail4<1, 1, 2, 4>("64x128 on 8 waves (32x32 each)", pr, st, tgt, reps);
    }
    return 0;
  }
  if (only == 12) {
    for (int pass = 0; pass < 2; ++pass) {
      one_kc<2, 1, 2, 4>("in-proj: 128x128 / 8 waves (product)", 51200, 256, 476, A, B, C);
      one_kc<2, 2, 2, 2>("in-proj: 128x128 / 4 waves", 51200, 256, 476, A, B, C);
      one_kc<2, 2, 4, 2>("in-proj: 256x128 / 8 waves", 51200, 256, 476, A, B, C);
      one_kc<2, 2, 2, 4>("in-proj: 128x256 / 8 waves", 51200, 256, 476, A, B, C);
      one_kc<1, 1, 2, 2>("in-proj: 64x64 / 4 waves", 51200, 256, 476, A, B, C);
      one_kc<1, 2, 2, 2>("in-proj: 64x128 / 4 waves", 51200, 256, 476, A, B, C);
      one_kc<1, 1, 2, 4>("in-proj: 64x128 / 8 waves", 51200, 256, 476, A, B, C);
      one_kc<2, 1, 2, 4>("dX-like fwd ih: 128x128 / 8", 51200, 1024, 256, A, B, C);
      one_kc<2, 2, 4, 2>("dX-like fwd ih: 256x128 / 8", 51200, 1024, 256, A, B, C);
      one_kc<2, 1, 2, 4, false>("dX (R x H x 4H): 128x128 / 8 (product)", 51200, 256, 1024, A, B, C);
      one_kc<1, 1, 2, 4, false>("dX (R x H x 4H): 64x128 / 8", 51200, 256, 1024, A, B, C);
      one_kc<1, 1, 2, 2, false>("dX (R x H x 4H): 64x64 / 4", 51200, 256, 1024, A, B, C);
    }
    return 0;
  }
  if (only == 7) {   // tile-count quantisation of the 128 x 128 kernel on the input-gradient shape: 512 slots (2 workgroups per CU)
    // (M <= 51200: the operand buffers hold 51200 x 1024 floats)
    for (int M : {16384, 32768, 40960, 49152, 51200}) run<true, false>("dx (M x H x 4H), M sweep", M, 256, 1024, 1, -1, A, B, C, false);
    for (int M : {32768, 49152, 51200}) run<true, true>("fwd in critic (M x H x 476), M sweep", M, 256, 476, 1, -1, A, B, C, false);
    return 0;
  }
  if (only == 8) {   // the paired weight-gradient product (4H x 2H x R) over its split-K factor: workgroups = 32 tiles x sk
    for (int sk : {8, 12, 16, 20, 24, 25, 28, 32, 40, 48}) run<false, false>("dW pair (4H x 2H x R), sk sweep", 1024, 512, 51200, sk, 1, A, B, C, false);
    return 0;
  }
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
