// kbj_gemm.h — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 157 TFLOP/s peak).
//
// C[M][N] (+)= A(m,k) * B(n,k) (+ bias[n]); both operands may be stored k-contiguous ([rows][k]) or
// row-contiguous ([k][rows]) so that one kernel serves
//   y  = x W^T        (A: x [M][K] k-contig,  B: W [N][K] k-contig)      forward projections
//   dx = dy W         (A: dy [M][K] k-contig, B: W [K][N] row-contig)    input gradients
//   dW = dy^T x       (A: dy [K][M] row-contig, B: x [K][N] row-contig)  weight gradients (split-K + atomics)
// Tiling: workgroup = 4 wavefronts (2x2), each wavefront owns MT x NT 32x32 accumulator tiles (64 VGPRs at 2x2);
// K is consumed in 32-wide LDS tiles. k-contiguous operands sit in LDS as [row][36] (ds_read_b128 fragments,
// conflict-free), row-contiguous ones as [k][row] (4 x ds_read_b32).
#pragma once
#include <hip/hip_runtime.h>

namespace kbj {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;          // C is M x N, contraction length K
  int lda, ldb, ldc;    // leading dimensions (elements) of the stored arrays
  int beta;             // 1: C += result, 0: C = result
  int splitk;           // >1: grid.z slices of K, results atomically added into C (C must be pre-zeroed or beta semantics handled by caller)
  const int* a_rows;    // optional gather: logical row m of A is stored row a_rows[m] (k-contiguous A only)
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = 36;  // padded k stride of a k-contiguous LDS tile

template <int MT, int NT, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr int A_ELEMS = A_KC ? BM * GEMM_LDK : GEMM_BK * (BM + 4);
  constexpr int B_ELEMS = B_KC ? BN * GEMM_LDK : GEMM_BK * (BN + 4);
  __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];
  float* As = lds;
  float* Bs = lds + A_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  int kbeg = 0, kend = g.K;
  if (g.splitk > 1) {
    int per = ((g.K + g.splitk - 1) / g.splitk + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    kbeg = blockIdx.z * per;
    kend = min(g.K, kbeg + per);
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const bool a_vec = A_KC ? ((g.lda & 3) == 0 && (((size_t)g.A & 15) == 0)) : ((g.lda & 3) == 0 && (((size_t)g.A & 15) == 0));
  const bool b_vec = B_KC ? ((g.ldb & 3) == 0 && (((size_t)g.B & 15) == 0)) : ((g.ldb & 3) == 0 && (((size_t)g.B & 15) == 0));

  for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
    // ---- stage A tile ----
    if (A_KC) {
      // BM rows x 32 k: 8 float4 per row; thread t -> k-quad t%8, rows t/8 + 32 i
      const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
      for (int i = 0; i < BM / 32; ++i) {
        int row = r0 + 32 * i, m = m0 + row, k = k0 + 4 * kq;
        f32x4 v = {0, 0, 0, 0};
        if (m < g.M) {
          size_t base = (size_t)(g.a_rows ? g.a_rows[m] : m) * g.lda;
          if (a_vec && k + 3 < kend) v = *reinterpret_cast<const f32x4*>(g.A + base + k);
          else { for (int e = 0; e < 4; ++e) if (k + e < kend) v[e] = g.A[base + k + e]; }
        }
        *reinterpret_cast<f32x4*>(As + row * GEMM_LDK + 4 * kq) = v;
      }
    } else {
      // 32 k-rows x BM m: BM/4 float4 per k-row
      constexpr int QPR = BM / 4;           // float4 per k-row
      constexpr int KROWS = 256 / QPR;      // k-rows covered per pass
      const int mq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < GEMM_BK / KROWS; ++i) {
        int kr = kr0 + KROWS * i, k = k0 + kr, m = m0 + 4 * mq;
        f32x4 v = {0, 0, 0, 0};
        if (k < kend) {
          size_t base = (size_t)k * g.lda;
          if (a_vec && m + 3 < g.M) v = *reinterpret_cast<const f32x4*>(g.A + base + m);
          else { for (int e = 0; e < 4; ++e) if (m + e < g.M) v[e] = g.A[base + m + e]; }
        }
        *reinterpret_cast<f32x4*>(As + kr * (BM + 4) + 4 * mq) = v;
      }
    }
    // ---- stage B tile ----
    if (B_KC) {
      const int kq = tid & 7, r0 = tid >> 3;
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {
        int row = r0 + 32 * i, n = n0 + row, k = k0 + 4 * kq;
        f32x4 v = {0, 0, 0, 0};
        if (n < g.N) {
          size_t base = (size_t)n * g.ldb;
          if (b_vec && k + 3 < kend) v = *reinterpret_cast<const f32x4*>(g.B + base + k);
          else { for (int e = 0; e < 4; ++e) if (k + e < kend) v[e] = g.B[base + k + e]; }
        }
        *reinterpret_cast<f32x4*>(Bs + row * GEMM_LDK + 4 * kq) = v;
      }
    } else {
      constexpr int QPR = BN / 4;
      constexpr int KROWS = 256 / QPR;
      const int nq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < GEMM_BK / KROWS; ++i) {
        int kr = kr0 + KROWS * i, k = k0 + kr, n = n0 + 4 * nq;
        f32x4 v = {0, 0, 0, 0};
        if (k < kend) {
          size_t base = (size_t)k * g.ldb;
          if (b_vec && n + 3 < g.N) v = *reinterpret_cast<const f32x4*>(g.B + base + n);
          else { for (int e = 0; e < 4; ++e) if (n + e < g.N) v[e] = g.B[base + n + e]; }
        }
        *reinterpret_cast<f32x4*>(Bs + kr * (BN + 4) + 4 * nq) = v;
      }
    }
    __syncthreads();
    // ---- MFMA over the 32-wide k tile: 4 groups of 8 k; lane half h = lane>>5 owns k = 8 kk + 4 h + j ----
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wr * 32 * MT + 32 * i + lr;
        if (A_KC) a[i] = *reinterpret_cast<const f32x4*>(As + row * GEMM_LDK + 8 * kk + 4 * lh);
        else { for (int e = 0; e < 4; ++e) a[i][e] = As[(8 * kk + 4 * lh + e) * (BM + 4) + row]; }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int row = wc * 32 * NT + 32 * j + lr;
        if (B_KC) b[j] = *reinterpret_cast<const f32x4*>(Bs + row * GEMM_LDK + 8 * kk + 4 * lh);
        else { for (int e = 0; e < 4; ++e) b[j][e] = Bs[(8 * kk + 4 * lh + e) * (BN + 4) + row]; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: accumulator (col = lane&31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)) -> C ----
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int n = n0 + wc * 32 * NT + 32 * j + lr;
      if (n >= g.N) continue;
      float bv = (g.bias && (g.splitk <= 1 || blockIdx.z == 0)) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + wr * 32 * MT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* c = g.C + (size_t)m * g.ldc + n;
        float v = acc[i][j][r] + bv;
        if (g.splitk > 1) atomicAdd(c, v);
        else *c = g.beta ? *c + v : v;
      }
    }
}

// host-side launcher; picks the 128x128 tile for large outputs and 64x64 when that leaves the chip underfilled
template <bool A_KC, bool B_KC>
inline void gemm_launch(hipStream_t s, const GemmArgs& g, int force_big = -1) {
  int sk = g.splitk > 1 ? g.splitk : 1;
  long big_blocks = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * sk;
  bool big = force_big >= 0 ? force_big != 0 : big_blocks >= 192;
  if (big) {
    dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, sk);
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2, A_KC, B_KC>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, sk);
    hipLaunchKernelGGL((gemm_f32_kernel<1, 1, A_KC, B_KC>), grid, dim3(256), 0, s, g);
  }
}

}  // namespace kbj
