// tools/wave_test: every wave primitive of kbj_wave.h and the whole arrow solve, GPU against the host emulation of the same source.
// Build + run (GPU box): make -C tools/wave_test && tools/wave_test/wave_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../kbot-joystick_amd/csrc/kbj_env_phys.h"
using namespace kbj;
extern "C" void wave_emu(const float* in, float* out);
extern "C" void solve_emu(const float* in, float* out);
constexpr int NPRIM = 16;
__global__ __launch_bounds__(64) void wave_kernel(const float* in, float* out) {
  const int l = threadIdx.x;
  const float x = in[l], y = in[64 + l], z = in[128 + l];
  int k = 0;
  out[64 * k++ + l] = wbcast<3>(x);
  { float a = z; wfmac_bcast<5>(a, x, y); out[64 * k++ + l] = a; }
  out[64 * k++ + l] = wmul_bcast<12>(x, y);
  { float a = z; wset_rhs<7, 0xF>(a, x); out[64 * k++ + l] = a; }
  { float a = z; wset_rhs<9, 0x1>(a, x); out[64 * k++ + l] = a; }
  out[64 * k++ + l] = wrow_sum16(x);
  out[64 * k++ + l] = wrows_sum1(x);
  { float a = x, b = y; wrows_sum2(a, b); out[64 * k++ + l] = a; out[64 * k++ + l] = b; }
  out[64 * k++ + l] = wsum(x);
  { float s1, s2; wsum2(x, y, s1, s2); out[64 * k++ + l] = s1; out[64 * k++ + l] = s2; }
  out[64 * k++ + l] = wsel<wmask_r(4)>(x, y);
  out[64 * k++ + l] = wsel0<(wmask_r_below(5) | (wmask_r_below(11) & 0xFFFFull))>(x);
  out[64 * k++ + l] = wneg_div_bcast<2>(x);
  { const unsigned long long m = wballot([&](int) { return x > 0.0f; }); out[64 * k++ + l] = (float)((m >> l) & 1); }
}
__global__ __launch_bounds__(64) void solve_kernel(const float* in, float* out, int n) {
  const int l = threadIdx.x;
  for (int s = blockIdx.x; s < n; s += gridDim.x) {
    const float* p = in + (size_t)s * 64 * 29;
    float m[11], h[11], oh[5];
#pragma unroll
    for (int j = 0; j < 11; ++j) { m[j] = p[64 * j + l]; h[j] = p[64 * (11 + j) + l]; }
#pragma unroll
    for (int j = 0; j < 5; ++j) oh[j] = p[64 * (22 + j) + l];
    out[(size_t)s * 64 + l] = arrow_solve_w(m, h, oh, p[64 * 27 + l], p[64 * 28 + l]);
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
int main() {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  int bad = 0;
  {  // primitives: exact equality, except the reciprocal (v_rcp_f32 + one Newton step against a division): 4 ulp
    std::vector<float> in(192), ref(64 * NPRIM), got(64 * NPRIM);
    for (auto& v : in) v = nd(rng);
    for (int l = 0; l < 64; ++l) if (std::fabs(in[l]) < 0.05f) in[l] = 0.7f;      // x is also a divisor
    wave_emu(in.data(), ref.data());
    float *din, *dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, got.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wave_kernel, dim3(1), dim3(64), 0, 0, din, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    const char* names[NPRIM] = {"wbcast", "wfmac_bcast", "wmul_bcast", "wset_rhs<0xF>", "wset_rhs<0x1>", "wrow_sum16", "wrows_sum1", "wrows_sum2.x", "wrows_sum2.y",
                                "wsum", "wsum2.s1", "wsum2.s2", "wsel", "wsel0", "wneg_div_bcast", "wballot"};
    for (int k = 0; k < NPRIM; ++k) {
      int nbad = 0; float worst = 0;
      for (int l = 0; l < 64; ++l) {
        const float a = ref[64 * k + l], b = got[64 * k + l];
        const bool ok = k == 14 ? std::fabs(a - b) <= 4 * 1.2e-7f * std::fabs(a) : std::memcmp(&a, &b, 4) == 0;
        if (!ok) { ++nbad; worst = std::fmax(worst, std::fabs(a - b)); }
      }
      printf("%-16s %s (%d lanes differ, worst %.3g)\n", names[k], nbad ? "FAIL" : "ok", nbad, worst);
      bad += nbad != 0;
    }
  }
  {  // arrow solve on random SPD arrow matrices in the solver layout: GPU vs emulation, and both vs a double-precision dense solve
    const int NS = 256;
    std::vector<float> in((size_t)NS * 64 * 29, 0.0f), ref((size_t)NS * 64), got((size_t)NS * 64);
    double worst_emu = 0, worst_dense = 0;
    std::vector<std::vector<double>> dense_x(NS);
    for (int s = 0; s < NS; ++s) {
      // dense 26 x 26 SPD with arrow sparsity: A = sum of a few rank-one terms per (chain, base) block + diagonal
      double A[26][26] = {}; double g[26];
      for (int i = 0; i < 26; ++i) { A[i][i] = 0.5 + std::fabs(nd(rng)); g[i] = nd(rng); }
      for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 6; ++t) {
          double v[11];                       // local: 0..4 chain dofs (ankle..hip), 5..10 base
          for (int k = 0; k < 11; ++k) v[k] = nd(rng);
          auto dof = [&](int k) { return k < 5 ? 10 + 5 * c - k : k - 5; };
          for (int a = 0; a < 11; ++a) for (int b = 0; b < 11; ++b) A[dof(a)][dof(b)] += 0.3 * v[a] * v[b];
        }
      float* p = in.data() + (size_t)s * 64 * 29;
      for (int l = 0; l < 64; ++l) {
        const int c = l >> 4, r = l & 15;
        auto dof = [&](int k) { return k < 5 ? 10 + 5 * c - k : k - 5; };
        if (r <= 10) {
          for (int j = 0; j < 11; ++j) {
            const bool basebase = r >= 5 && j >= 5;
            const double val = A[dof(r)][dof(j)];
            // split M / h arbitrarily: M gets 70 %, h the rest; the base-base block lives in DPP row 0 only
            p[64 * j + l] = basebase && c != 0 ? 0.0f : (float)(0.7 * val);
            p[64 * (11 + j) + l] = basebase && c != 0 ? 0.0f : (float)(0.3 * val);
          }
          p[64 * 28 + l] = (float)g[dof(r)];
        }
        for (int j = 0; j < 5; ++j) p[64 * (22 + j) + l] = r == j ? 1.0f : 0.0f;
        p[64 * 27 + l] = r < 5 ? 0.25f : 0.0f;     // dnow on the chain diagonals
      }
      for (int c = 0; c < 4; ++c) for (int k = 0; k < 5; ++k) A[10 + 5 * c - k][10 + 5 * c - k] += 0.25;
      // dense solve in double (Gaussian elimination, SPD: no pivoting needed)
      std::vector<double> x(26);
      { double M[26][27]; for (int i = 0; i < 26; ++i) { for (int j = 0; j < 26; ++j) M[i][j] = A[i][j]; M[i][26] = g[i]; }
        for (int p2 = 0; p2 < 26; ++p2) for (int i = p2 + 1; i < 26; ++i) { const double f = M[i][p2] / M[p2][p2]; for (int j = p2; j < 27; ++j) M[i][j] -= f * M[p2][j]; }
        for (int i = 25; i >= 0; --i) { double sum = M[i][26]; for (int j = i + 1; j < 26; ++j) sum -= M[i][j] * x[j]; x[i] = sum / M[i][i]; } }
      dense_x[s] = x;
      solve_emu(p, ref.data() + (size_t)s * 64);
    }
    float *din, *dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, got.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(solve_kernel, dim3(64), dim3(64), 0, 0, din, dout, NS);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < NS; ++s) {
      double scale = 0; for (double v : dense_x[s]) scale = std::fmax(scale, std::fabs(v));
      for (int l = 0; l < 64; ++l) {
        const int c = l >> 4, r = l & 15;
        if (r > 10) continue;
        const int d = r < 5 ? 10 + 5 * c - r : r - 5;
        worst_emu = std::fmax(worst_emu, std::fabs((double)got[s * 64 + l] - ref[s * 64 + l]) / scale);
        worst_dense = std::fmax(worst_dense, std::fabs((double)got[s * 64 + l] - dense_x[s][d]) / scale);
      }
    }
    printf("arrow_solve_w    GPU vs emulation %.3g, GPU vs double dense solve %.3g (relative to max |x|, %d systems)\n", worst_emu, worst_dense, NS);
    if (!(worst_emu < 2e-5) || !(worst_dense < 2e-4)) { printf("arrow_solve_w    FAIL\n"); ++bad; }
  }
  printf(bad ? "WAVE TEST FAILED (%d)\n" : "WAVE TEST PASSED\n", bad);
  return bad ? 1 : 0;
}
