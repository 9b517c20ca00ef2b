// kbj_gemm.h — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 157 TFLOP/s peak).
//
// C[M][N] (+)= A(m,k) * B(n,k) (+ bias[n]); both operands may be stored k-contiguous ([rows][k]) or
// row-contiguous ([k][rows]) so that one kernel serves
//   y  = x W^T        (A: x [M][K] k-contig,  B: W [N][K] k-contig)      forward projections
//   dx = dy W         (A: dy [M][K] k-contig, B: W [K][N] row-contig)    input gradients
//   dW = dy^T x       (A: dy [K][M] row-contig, B: x [K][N] row-contig)  weight gradients (split-K + atomics)
// Tiling: workgroup = 4 wavefronts (2x2), each wavefront owns MT x NT 32x32 accumulator tiles (64 VGPRs at 2x2);
// K is consumed in 32-wide LDS tiles. k-contiguous operands sit in LDS as [row][36] (ds_read_b128 fragments,
// conflict-free), row-contiguous ones as [k][row] (4 x ds_read_b32). The global loads of tile k+1 are issued into
// registers before the MFMAs of tile k and written to LDS after them (software pipelining: HBM/L2 latency behind the
// matrix pipe).
#pragma once
#include <hip/hip_runtime.h>
#include "kbj_ctx.h"

namespace kbj {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;          // C is M x N, contraction length K
  int lda, ldb, ldc;    // leading dimensions (elements) of the stored arrays
  int beta;             // 1: C += result, 0: C = result
  int splitk;           // >1: grid.z slices of K, results atomically added into C
  const int* a_rows;    // unused (reserved)
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = 36;  // padded k stride of a k-contiguous LDS tile

// one operand tile (ROWS rows x 32 k) staged through registers: 256 threads, ROWS/32 float4 each
template <int ROWS, bool KC>
struct GemmStage {
  static constexpr int NV = ROWS / 32;
  f32x4 v[NV];
  // P: operand base, ld, R: number of valid rows, r0: first row of the tile, k0/kend: k range
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend, bool vec) {
    const int tid = threadIdx.x;
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int r = r0 + rr + 32 * i, k = k0 + 4 * kq;
        f32x4 x = {0, 0, 0, 0};
        if (r < R) {
          size_t base = (size_t)r * ld;
          if (vec && k + 3 < kend) x = *reinterpret_cast<const f32x4*>(P + base + k);
          else { for (int e = 0; e < 4; ++e) if (k + e < kend) x[e] = P[base + k + e]; }
        }
        v[i] = x;
      }
    } else {
      constexpr int QPR = ROWS / 4, KROWS = 256 / QPR;
      const int rq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int k = k0 + kr0 + KROWS * i, r = r0 + 4 * rq;
        f32x4 x = {0, 0, 0, 0};
        if (k < kend) {
          size_t base = (size_t)k * ld;
          if (vec && r + 3 < R) x = *reinterpret_cast<const f32x4*>(P + base + r);
          else { for (int e = 0; e < 4; ++e) if (r + e < R) x[e] = P[base + r + e]; }
        }
        v[i] = x;
      }
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int tid = threadIdx.x;
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(lds + (rr + 32 * i) * GEMM_LDK + 4 * kq) = v[i];
    } else {
      constexpr int QPR = ROWS / 4, KROWS = 256 / QPR;
      const int rq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(lds + (kr0 + KROWS * i) * (ROWS + 4) + 4 * rq) = v[i];
    }
  }
};

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

  const bool a_vec = (g.lda & 3) == 0 && (((size_t)g.A & 15) == 0);
  const bool b_vec = (g.ldb & 3) == 0 && (((size_t)g.B & 15) == 0);
  GemmStage<BM, A_KC> sa;
  GemmStage<BN, B_KC> sb;
  const int lr = lane & 31, lh = lane >> 5;
  if (kbeg < kend) {
    sa.load(g.A, g.lda, g.M, m0, kbeg, kend, a_vec);
    sb.load(g.B, g.ldb, g.N, n0, kbeg, kend, b_vec);
    sa.store(As); sb.store(Bs);
    __syncthreads();
  }
  for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
    const bool more = k0 + GEMM_BK < kend;
    if (more) {  // next tile in flight during this tile's MFMAs
      sa.load(g.A, g.lda, g.M, m0, k0 + GEMM_BK, kend, a_vec);
      sb.load(g.B, g.ldb, g.N, n0, k0 + GEMM_BK, kend, b_vec);
    }
    // MFMA over the 32-wide k tile: 4 groups of 8 k; lane half lh owns k = 8 kk + 4 lh + e
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
    if (more) { sa.store(As); sb.store(Bs); __syncthreads(); }
  }
  // ---- epilogue: accumulator (col = lane&31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)) -> C ----
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
  KbjKernelTimer timer(s, KBJ_KIND_GEMM + (big ? 0 : 4) + (A_KC ? 2 : 0) + (B_KC ? 1 : 0), 2.0 * g.M * g.N * g.K);
  if (big) {
    dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, sk);
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2, A_KC, B_KC>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, sk);
    hipLaunchKernelGGL((gemm_f32_kernel<1, 1, A_KC, B_KC>), grid, dim3(256), 0, s, g);
  }
}

}  // namespace kbj
