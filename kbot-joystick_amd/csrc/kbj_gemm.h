// kbj_gemm.h — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 157 TFLOP/s peak).
//
// C[M][N] (+)= A(m,k) * B(n,k) (+ bias[n]); both operands may be stored k-contiguous ([rows][k]) or
// row-contiguous ([k][rows]) so that one kernel serves
//   y  = x W^T        (A: x [M][K] k-contig,  B: W [N][K] k-contig)      forward projections
//   dx = dy W         (A: dy [M][K] k-contig, B: W [K][N] row-contig)    input gradients
//   dW = dy^T x       (A: dy [K][M] row-contig, B: x [K][N] row-contig)  weight gradients (split-K + atomics)
// Tiling: workgroup = WM x WN wavefronts, each owning MT x NT 32x32 accumulator tiles. The 128x128 tile runs on 8 wavefronts
// (2 x 4, 64x32 each, 99 VGPRs: four waves per SIMD at two workgroups per CU), the 64x64 tile on 4 (2 x 2, 32x32 each);
// K is consumed in 32-wide LDS tiles. k-contiguous operands sit in LDS as [row][36] (ds_read_b128 fragments,
// conflict-free), row-contiguous ones as [k][row] (4 x ds_read_b32).
// Pipeline: two LDS stages and one barrier per k tile — the global loads of tile k+1 are issued into registers before
// the MFMAs of tile k and written to the other stage after them. One work item (output tile x k slice) per workgroup: with two
// workgroups per CU the neighbour's k loop covers a workgroup's first-tile latency and C write-back. Interior tiles are fetched with
// buffer_load_b128 from per-thread byte offsets computed once per item plus a SCALAR k offset, and the k loop is unrolled over the two
// LDS stages, so a k tile costs no vector address arithmetic: fp32 MFMA and the vector ALU share a SIMD's issue slots (DESIGN.md
// section 5), every VALU instruction in this loop is matrix time lost.
// Work items are ordered n-tile fastest and handed out so that the workgroups of one XCD (blockIdx % 8) own a contiguous
// range: the operand panel shared by neighbouring tiles is fetched into one L2, not eight.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include "kbj_ctx.h"

namespace kbj {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;          // C is M x N, contraction length K
  int lda, ldb, ldc;    // leading dimensions (elements) of the stored arrays
  int beta;             // 1: C += result, 0: C = result
  int splitk;           // >1: slices of K, results atomically added into C
  const int* a_rows;    // unused (reserved)
  // optional second source along k (k-contiguous operands only, k1 a multiple of 32, no split-K): for k >= k1 the operands are
  // A2[m][k - k1], B2[n][k - k1] with the same leading dimensions, i.e. C = [A | A2] [B | B2]^T without materialising the concatenation
  const float* A2 = nullptr; const float* B2 = nullptr; int k1 = 0;
  // optional second problem along n (n1 a multiple of the tile width): output columns n >= n1 use B2 (leading dimension ldb2,
  // 0 = ldb) and go to C2[m][n - n1] (ldc2, 0 = ldc) — two products that share A in one launch, so neighbouring workgroups share
  // the A panel in L2
  float* C2 = nullptr; int n1 = 0, ldb2 = 0, ldc2 = 0;
  // optional row gather of a k-contiguous A (A_KC kernels, no second k source): logical row m = t * a_B + b is read from stored row
  // t * a_N + a_idx[b] - the minibatch view [T][B] of a trajectory array [T][N] through the minibatch's env indices, without a
  // gathered copy in front of the GEMM (the per-thread row offsets are computed once per work item either way)
  const int* a_idx = nullptr; int a_B = 0, a_N = 0;
  // deterministic split-K (kbj_config.deterministic): instead of fp32 atomics into C, k slice ks stores its partial tile into
  // skws[ks][M][N] (N = all columns of the launch, both problems) and splitk_reduce_kernel adds the slices to C in slice order
  float* skws = nullptr;
  // kbj_config.gemm_bf16x3: eligible 128x128-tile launches run on the bf16 matrix cores through the exact three-way operand split
  // (gemm_x3_kernel below); set by the callers from the context's schedule, ignored where the launch is not eligible
  int x3 = 0;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = 36;  // padded k stride of a k-contiguous LDS tile

// one operand tile (ROWS rows x 32 k) staged through registers: NTH threads, ROWS * 8 / NTH float4 each
template <int ROWS, bool KC, int NTH>
struct GemmStage {
  static constexpr int NV = ROWS * 8 / NTH;
  static constexpr int RPP = NTH / 8;        // rows per pass of a k-contiguous tile (8 threads x float4 per 32-k row)
  f32x4 v[NV];
  // interior tile (all ROWS rows and 32 k in range, 16-byte aligned rows): straight-line vector loads
  __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int r0, int k0) {
    const int tid = threadIdx.x;
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;
      const float* p = P + (size_t)(r0 + rr) * ld + k0 + 4 * kq;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(RPP * i) * ld);
    } else {
      constexpr int QPR = ROWS / 4, KROWS = NTH / QPR;
      const int rq = tid % QPR, kr0 = tid / QPR;
      const float* p = P + (size_t)(k0 + kr0) * ld + r0 + 4 * rq;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(KROWS * i) * ld);
    }
  }
  // P: operand base, ld, R: number of valid rows, r0: first row of the tile, k0/kend: k range
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int R, int r0, int k0, int kend, bool vec, const int* idx = nullptr, int iB = 0,
                                       int iN = 0) {
    const int tid = threadIdx.x;
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int r = r0 + rr + RPP * i, k = k0 + 4 * kq;
        f32x4 x = {0, 0, 0, 0};
        if (r < R) {
          size_t base = (size_t)r * ld;
          if (idx) { const int t = r / iB; base = ((size_t)t * iN + idx[r - t * iB]) * ld; }   // GemmArgs::a_idx
          if (vec && k + 3 < kend) x = *reinterpret_cast<const f32x4*>(P + base + k);
          else { for (int e = 0; e < 4; ++e) if (k + e < kend) x[e] = P[base + k + e]; }
        }
        v[i] = x;
      }
    } else {
      constexpr int QPR = ROWS / 4, KROWS = NTH / QPR;
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
      for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(lds + (rr + RPP * i) * GEMM_LDK + 4 * kq) = v[i];
    } else {
      constexpr int QPR = ROWS / 4, KROWS = NTH / QPR;
      const int rq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(lds + (kr0 + KROWS * i) * (ROWS + 4) + 4 * rq) = v[i];
    }
  }
};

// work item -> output tile and k range (n tile fastest, then m tile, then k slice)
struct GemmItem { int m0, n0, kbeg, kend, ks; const float* B; float* C; int ncols, ldb, ldc, ncol0; };   // n0, ncols: local to the (B, C) problem; ncol0: its first column in the launch
template <int BM, int BN>
__device__ __forceinline__ GemmItem gemm_item(const GemmArgs& g, int item, int tiles_n, int tiles_m, int per) {
  GemmItem it;
  int tn = item % tiles_n, r = item / tiles_n;
  int tm = r % tiles_m;
  it.ks = r / tiles_m;
  it.m0 = tm * BM; it.n0 = tn * BN;
  it.B = g.B; it.C = g.C; it.ncols = g.N; it.ldb = g.ldb; it.ldc = g.ldc; it.ncol0 = 0;
  if (g.n1 > 0) {
    if (it.n0 >= g.n1) {
      it.B = g.B2; it.C = g.C2; it.n0 -= g.n1; it.ncols = g.N - g.n1; it.ncol0 = g.n1;
      if (g.ldb2 > 0) it.ldb = g.ldb2;
      if (g.ldc2 > 0) it.ldc = g.ldc2;
    }
    else it.ncols = g.n1;
  }
  it.kbeg = it.ks * per;
  it.kend = min(g.K, it.kbeg + per);
  return it;
}

// WM x WN wavefronts per workgroup, each owning MT x NT 32x32 accumulator tiles. One work item (output tile x k slice) per workgroup.
// On this chip fp32 MFMA and vector instructions share a SIMD's issue, so the k loop is written to need almost no
// vector instructions besides the MFMAs: interior tiles are fetched with buffer loads whose per-thread byte offsets are computed once
// and whose k advance is a scalar offset; the loop is unrolled over the two LDS stages so that every LDS address is a per-thread base
// plus an immediate.
template <int MT, int NT, bool A_KC, bool B_KC, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN, NTH = 64 * WM * WN;
  constexpr int A_ELEMS = A_KC ? BM * GEMM_LDK : GEMM_BK * (BM + 4);
  constexpr int B_ELEMS = B_KC ? BN * GEMM_LDK : GEMM_BK * (BN + 4);
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * STAGE floats (72 KB at 128x128: above the 64 KB static limit)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  const int per = sk > 1 ? ((g.K + sk - 1) / sk + GEMM_BK - 1) / GEMM_BK * GEMM_BK : g.K;
  const int items = tiles_n * tiles_m * sk;
  // XCD-contiguous hand-out: workgroup b runs on XCD b % 8 and takes logical slot (b % 8) * (G / 8) + b / 8
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (item >= items) return;
  const GemmItem cur = gemm_item<BM, BN>(g, item, tiles_n, tiles_m, per);
  if (cur.kbeg >= cur.kend) return;   // empty k slice (K not a multiple of the slice length): nothing to add

  const bool a_vec = (g.lda & 3) == 0 && (((size_t)g.A & 15) == 0) && (((size_t)g.A2 & 15) == 0);
  const bool b_vec = (g.ldb & 3) == 0 && (g.ldb2 & 3) == 0 && (((size_t)g.B & 15) == 0) && (((size_t)g.B2 & 15) == 0);   // B2 null or aligned
  // interior tiles (all rows in range, vector-aligned) take the buffer-load path, decided per operand
  const bool rows_a = a_vec && cur.m0 + BM <= g.M, rows_b = b_vec && cur.n0 + BN <= cur.ncols;
  GemmStage<BM, A_KC, NTH> sa;
  GemmStage<BN, B_KC, NTH> sb;
  // per-thread byte offsets of the staged pieces at k = 0 (buffer-load path): k-contiguous operand: (row * ld + 4 kq) * 4, k advances by
  // 4 bytes per k; row-contiguous operand: (k row * ld + r0 + 4 rq) * 4, k advances by ld * 4 bytes per k
  unsigned voa[GemmStage<BM, A_KC, NTH>::NV], vob[GemmStage<BN, B_KC, NTH>::NV];
  {
    if (A_KC) { const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int i = 0; i < sa.NV; ++i) {
        size_t row = (size_t)(cur.m0 + rr + sa.RPP * i);
        if (g.a_idx && rows_a) {   // gathered rows, relative to the tile's first time step (a_base below): offsets stay small
          const int m = (int)row, t = m / g.a_B;
          row = (size_t)(t - cur.m0 / g.a_B) * g.a_N + g.a_idx[m - t * g.a_B];
        }
        voa[i] = (unsigned)((row * g.lda + 4 * kq) * 4);
      } }
    else { constexpr int QPR = BM / 4, KROWS = NTH / QPR; const int rq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < sa.NV; ++i) voa[i] = (unsigned)(((size_t)(kr0 + KROWS * i) * g.lda + cur.m0 + 4 * rq) * 4); }
    if (B_KC) { const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int i = 0; i < sb.NV; ++i) vob[i] = (unsigned)(((size_t)(cur.n0 + rr + sb.RPP * i) * cur.ldb + 4 * kq) * 4); }
    else { constexpr int QPR = BN / 4, KROWS = NTH / QPR; const int rq = tid % QPR, kr0 = tid / QPR;
#pragma unroll
      for (int i = 0; i < sb.NV; ++i) vob[i] = (unsigned)(((size_t)(kr0 + KROWS * i) * cur.ldb + cur.n0 + 4 * rq) * 4); }
  }
  // one k tile of both operands into the staging registers
  auto load_tile = [&](int k) {
    const float *pa = g.A, *pb = cur.B;
    int kk = k, ke = cur.kend;
    if (g.k1 > 0) {
      if (k >= g.k1) { pa = g.A2; pb = g.B2; kk = k - g.k1; ke = g.K - g.k1; }
      else ke = g.k1;
    }
    const bool kfull = kk + GEMM_BK <= ke;
    if (rows_a && kfull) {
      if (A_KC && g.a_idx) pa += (size_t)(cur.m0 / g.a_B) * g.a_N * g.lda;   // first time step of the tile
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa), 0, 0x7FFFFFFF, 0x00020000);
      const unsigned so = (unsigned)(A_KC ? (size_t)kk * 4 : (size_t)kk * g.lda * 4);      // scalar: the k advance costs no vector instruction
#pragma unroll
      for (int i = 0; i < sa.NV; ++i) sa.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[i], so, 0));
    } else sa.load(pa, g.lda, g.M, cur.m0, kk, ke, a_vec, A_KC ? g.a_idx : nullptr, g.a_B, g.a_N);
    if (rows_b && kfull) {
      __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pb), 0, 0x7FFFFFFF, 0x00020000);
      const unsigned so = (unsigned)(B_KC ? (size_t)kk * 4 : (size_t)kk * cur.ldb * 4);
#pragma unroll
      for (int i = 0; i < sb.NV; ++i) sb.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vob[i], so, 0));
    } else sb.load(pb, cur.ldb, cur.ncols, cur.n0, kk, ke, b_vec);
  };
  load_tile(cur.kbeg);
  sa.store(lds); sb.store(lds + A_ELEMS);
  __syncthreads();

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // one k tile out of LDS stage S (a compile-time constant: every LDS address below is a per-thread base plus an immediate). The next
  // tile's global loads are issued first and written to the other stage half way through the MFMAs (they have landed behind the first
  // 16 x MT x NT of them; that stage was last read one k tile ago, before the barrier every wavefront has passed since).
  auto k_tile = [&](auto stage_tag, int k0) {
    constexpr int S = decltype(stage_tag)::value;
    const bool more = k0 + GEMM_BK < cur.kend;
    if (more) load_tile(k0 + GEMM_BK);
    const float* As = lds + S * STAGE;
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int kk = 2 * half; kk < 2 * half + 2; ++kk) {
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
      if (half == 0 && more) {
        float* dst = lds + (S ^ 1) * STAGE;
        sa.store(dst); sb.store(dst + A_ELEMS);
      }
    }
    __syncthreads();
  };
  for (int k0 = cur.kbeg; k0 < cur.kend; k0 += 2 * GEMM_BK) {
    k_tile(std::integral_constant<int, 0>{}, k0);
    if (k0 + GEMM_BK < cur.kend) k_tile(std::integral_constant<int, 1>{}, k0 + GEMM_BK);
  }
  // ---- epilogue: accumulator (col = lane&31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)) -> C ----
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int n = cur.n0 + wc * 32 * NT + 32 * j + lr;
      if (n >= cur.ncols) continue;
      float bv = (g.bias && cur.ks == 0) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = cur.m0 + wr * 32 * MT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* c = cur.C + (size_t)m * cur.ldc + n;
        float v = acc[i][j][r] + bv;
#ifdef KBJ_GEMM_NOSTORE   // timing experiment only (tools/gemm_bench): keep the MFMAs live, drop the write-back
        if (v == 1.2345e-30f) *c = v;
#else
        if (sk > 1) {
          if (g.skws) g.skws[((size_t)cur.ks * g.M + m) * g.N + cur.ncol0 + n] = v;
          else atomicAdd(c, v);
        } else *c = g.beta ? *c + v : v;
#endif
      }
    }
}


// ---- fp32 GEMM on the bf16 matrix cores through an EXACT three-way operand split (kbj_config.gemm_bf16x3) ----
// x = hi + mid + lo with hi = x truncated to its top 8 significand bits (a bf16), mid = (x - hi) truncated likewise, lo = x - hi - mid: both
// subtractions are exact in fp32 and lo has at most 8 significant bits left, so the three bf16 pieces carry all 24 bits of x. The products
// hi hi, hi mid, mid hi, hi lo, lo hi, mid mid are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (each exact; smallest first); the three
// dropped ones (mid lo, lo mid, lo lo) are <= 2^-23 relative, below the accumulation's own rounding: measured against fp64 the result is
// MORE accurate than the v_mfma_f32_32x32x2_f32 chain at the median and at p99.9 (one rounding per 16 k instead of one per 2). The bf16
// instruction has 16x the per-instruction rate of the fp32 one and holds the vector issue port 8 of its 32 cycles, which leaves the port to
// the split itself (~5.5 vector instructions per element, once per element on the way into LDS). NOT the default: the headline path of this
// library is the plain fp32-MFMA kernel above.
// Tiling: 128 x 128 x 32 on 4 wavefronts (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles); LDS [piece][row][40] bf16 per operand (80-byte rows:
// conflict-free ds_read_b128 fragments), one stage + register prefetch, 60 KB: two workgroups per CU cover each other's barriers.
// Serves what the update's large launches need: both operand layouts, split-K with atomics, the paired problem along n, ragged M / N
// (guarded edge tiles). Not: a second k source, row gathers, bias, deterministic split-K slabs - those launches stay on the exact kernel.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int X3_LD = 40;
constexpr int X3_PIECE = 128 * X3_LD;
__device__ __forceinline__ void x3_split(const f32x4& x, u32x2& hi, u32x2& mid, u32x2& lo) {   // four consecutive k of one row -> 3 x (4 bf16)
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
    const float r1 = x[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
  }
  hi = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
  mid = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  lo = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}
// one operand tile ROWS rows x 32 k staged as ROWS / 32 (row, 4 consecutive k) groups per thread (256 threads)
template <bool KC, int ROWS> struct X3Stage {
  static constexpr int NG = ROWS / 32;
  static_assert(KC || ROWS == 128, "row-contiguous operands are transposed in 4 x 4 blocks: 128-row tiles only");
  f32x4 v[NG];
  // P: operand base, ld; R: valid rows; r0: first row of the tile; k0: first k (a full 32-k tile, k-range checked by the caller)
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int R, int r0, int k0) {
    const int t = threadIdx.x;
    if (KC) { const int kq = t & 7, rr = t >> 3;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int r = r0 + rr + 32 * i;
        v[i] = r < R ? *reinterpret_cast<const f32x4*>(P + (size_t)r * ld + k0 + 4 * kq) : f32x4{0, 0, 0, 0};
      }
    } else { const int kg = t & 7, rq = t >> 3, r = r0 + 4 * rq;   // 8 lanes walk the k groups of one row group: their LDS writes below are 64 contiguous bytes of one row
      f32x4 w[4];                                                   // (row groups on consecutive lanes would be 320 bytes apart: 8-way bank conflicts)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* p = P + (size_t)(k0 + 4 * kg + j) * ld + r;
        if (r + 3 < R) w[j] = *reinterpret_cast<const f32x4*>(p);
        else { for (int e = 0; e < 4; ++e) w[j][e] = r + e < R ? p[e] : 0.0f; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = {w[0][i], w[1][i], w[2][i], w[3][i]};      // row 4 rq + i, k 4 kg .. + 3
    }
  }
  // general form of a k-contiguous tile (GEN kernels): per-group element offsets of the rows (row gathers resolved by the caller, -1 = no
  // such row), a k range that may end inside the tile (whole float4s: the callers' K are multiples of 4)
  __device__ __forceinline__ void load_gen(const float* __restrict__ P, const long (&roff)[NG], int k0, int kend) {
    const int k = k0 + 4 * (threadIdx.x & 7);
#pragma unroll
    for (int i = 0; i < NG; ++i) v[i] = (roff[i] >= 0 && k + 3 < kend) ? *reinterpret_cast<const f32x4*>(P + roff[i] + k) : f32x4{0, 0, 0, 0};
  }
  __device__ __forceinline__ void store(short* lds) const {    // lds: [3][ROWS][X3_LD]
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int row = KC ? (t >> 3) + 32 * i : 4 * (t >> 3) + i, k = 4 * (t & 7);
      u32x2 hi, mid, lo;
      x3_split(v[i], hi, mid, lo);
      short* p = lds + row * X3_LD + k;
      *reinterpret_cast<u32x2*>(p) = hi; *reinterpret_cast<u32x2*>(p + ROWS * X3_LD) = mid; *reinterpret_cast<u32x2*>(p + 2 * ROWS * X3_LD) = lo;
    }
  }
};
// TM: MFMA tiles per wavefront and dimension: 2 = 128 x 128 workgroup tiles (4 wavefronts of 64 x 64), 1 = 64 x 64 (4 wavefronts of 32 x 32:
// 30 KB of LDS, ~70 VGPRs - it fits beside the env kernel's partial last round like the exact small-tile kernel, DESIGN.md "Rollout schedule").
// GEN (k-contiguous operands only): bias, a second k source (k1 a multiple of 32), row gathers of A, K that is only a multiple of 4.
template <int TM, bool A_KC, bool B_KC, bool GEN>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmArgs g) {
  constexpr int BM = 64 * TM, PIECE = BM * X3_LD;
  static_assert(!GEN || (A_KC && B_KC), "the general loads are written for k-contiguous operands");
  extern __shared__ __attribute__((aligned(16))) short x3lds[];   // A: 3 pieces, then B: 3 pieces
  short* As = x3lds; short* Bs = x3lds + 3 * PIECE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int tiles_n = (g.N + BM - 1) / BM, tiles_m = (g.M + BM - 1) / BM;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  const int per = sk > 1 ? ((g.K + sk - 1) / sk + GEMM_BK - 1) / GEMM_BK * GEMM_BK : g.K;
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (item >= tiles_n * tiles_m * sk) return;
  const GemmItem cur = gemm_item<BM, BM>(g, item, tiles_n, tiles_m, per);
  if (cur.kbeg >= cur.kend) return;
  X3Stage<A_KC, BM> sa; X3Stage<B_KC, BM> sb;
  long roa[X3Stage<A_KC, BM>::NG], rob[X3Stage<B_KC, BM>::NG];
  if (GEN) {
#pragma unroll
    for (int i = 0; i < sa.NG; ++i) {
      const int m = cur.m0 + (threadIdx.x >> 3) + 32 * i, n = cur.n0 + (threadIdx.x >> 3) + 32 * i;
      long row = m;
      if (g.a_idx && m < g.M) { const int tt = m / g.a_B; row = (long)tt * g.a_N + g.a_idx[m - tt * g.a_B]; }    // minibatch view of a trajectory array (GemmArgs::a_idx)
      roa[i] = m < g.M ? row * g.lda : -1;
      rob[i] = n < cur.ncols ? (long)n * cur.ldb : -1;
    }
  }
  auto load_tile = [&](int k) {
    if (GEN) {
      const float *pa = g.A, *pb = cur.B;
      int kk = k, ke = cur.kend;
      if (g.k1 > 0) { if (k >= g.k1) { pa = g.A2; pb = g.B2; kk = k - g.k1; ke = g.K - g.k1; } else ke = g.k1; }
      sa.load_gen(pa, roa, kk, ke); sb.load_gen(pb, rob, kk, ke);
    } else { sa.load(g.A, g.lda, g.M, cur.m0, k); sb.load(cur.B, cur.ldb, cur.ncols, cur.n0, k); }
  };
  load_tile(cur.kbeg);
  f32x16 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  for (int k0 = cur.kbeg; k0 < cur.kend; k0 += 32) {
    __syncthreads();                       // every wavefront has read the previous tile
    sa.store(As); sb.store(Bs);
    __syncthreads();
    if (k0 + 32 < cur.kend) load_tile(k0 + 32);     // in flight behind the MFMAs
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[3][TM], b[3][TM];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          a[p][i] = *reinterpret_cast<const bf16x8*>(As + p * PIECE + (wr * 32 * TM + 32 * i + lr) * X3_LD + 16 * kk + 8 * lh);
          b[p][i] = *reinterpret_cast<const bf16x8*>(Bs + p * PIECE + (wc * 32 * TM + 32 * i + lr) * X3_LD + 16 * kk + 8 * lh);
        }
      constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};   // mid mid, lo hi, hi lo, mid hi, hi mid, hi hi (0 hi, 1 mid, 2 lo)
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int n = cur.n0 + wc * 32 * TM + 32 * j + lr;
      if (n >= cur.ncols) continue;
      const float bv = (GEN && g.bias && cur.ks == 0) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = cur.m0 + wr * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* c = cur.C + (size_t)m * cur.ldc + n;
        const float v = acc[i][j][r] + bv;
        if (sk > 1) atomicAdd(c, v);
        else *c = g.beta ? *c + v : v;
      }
    }
}
// the launches the split kernel serves (everything else stays on the exact kernel, silently): 0 = none, 1 = the plain form, 2 = the general
// form (k-contiguous operands: bias / second k source / row gather / K a multiple of 4)
inline int gemm_x3_form(const GemmArgs& g, bool a_kc, bool b_kc, bool big) {
  if (!g.x3 || g.skws || (g.B2 && g.n1 <= 0 && g.k1 <= 0)) return 0;
  auto al = [](const void* p, int ld) { return ((size_t)p & 15) == 0 && (ld & 3) == 0; };
  if (!al(g.A, g.lda) || !al(g.B, g.ldb) || g.M < 64 || g.N < 64) return 0;
  const bool gen = g.bias || g.k1 > 0 || g.a_idx || g.K % GEMM_BK != 0;
  if (gen) {
    if (!a_kc || !b_kc || g.n1 > 0 || g.splitk > 1 || g.K % 4 != 0 || g.k1 % GEMM_BK != 0) return 0;
    if (g.k1 > 0 && (!al(g.A2, g.lda) || !al(g.B2, g.ldb))) return 0;
    return 2;
  }
  if (!big && !(a_kc && b_kc)) return 0;                    // the 64 x 64 form exists for k-contiguous operands
  if (g.n1 > 0 && (!al(g.B2, g.ldb2 > 0 ? g.ldb2 : g.ldb) || g.n1 % (big ? 128 : 64) != 0)) return 0;
  return 1;
}
template <int TM, bool A_KC, bool B_KC, bool GEN> inline void gemm_x3_launch_tile(hipStream_t s, const GemmArgs& g, int sk) {
  constexpr int BM = 64 * TM;
  constexpr size_t bytes = 6 * BM * X3_LD * sizeof(short);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<TM, A_KC, B_KC, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)attr;
  const long it = (long)((g.M + BM - 1) / BM) * ((g.N + BM - 1) / BM) * sk;
  KbjKernelTimer timer(s, kbj_kind_gemm_x3(TM, A_KC, B_KC, GEN), 2.0 * g.M * g.N * g.K);
  hipLaunchKernelGGL((gemm_x3_kernel<TM, A_KC, B_KC, GEN>), dim3((unsigned)((it + 7) / 8 * 8)), dim3(256), bytes, s, g);
}

// second stage of the deterministic split-K: C (+)= sum over the k slices, in slice order, of the partial tiles (one thread per output
// element; slices whose k range is empty wrote nothing and are skipped exactly as the GEMM skipped them)
__global__ void splitk_reduce_kernel(GemmArgs g, int sk, int per) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)g.M * g.N) return;
  const int m = (int)(i / g.N), n = (int)(i % g.N);
  float s = 0.0f;
  for (int ks = 0; ks < sk && (long)ks * per < g.K; ++ks) s += g.skws[((size_t)ks * g.M + m) * g.N + n];
  float* c = (g.n1 > 0 && n >= g.n1) ? g.C2 + (size_t)m * (g.ldc2 > 0 ? g.ldc2 : g.ldc) + (n - g.n1) : g.C + (size_t)m * g.ldc + n;
  *c += s;
}

template <int MT, int NT, bool A_KC, bool B_KC, int WM = 2, int WN = 2>
inline void gemm_launch_tile(hipStream_t s, const GemmArgs& g, int wgs) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr size_t bytes = 2 * sizeof(float) * ((A_KC ? BM * GEMM_LDK : GEMM_BK * (BM + 4)) + (B_KC ? BN * GEMM_LDK : GEMM_BK * (BN + 4)));
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<MT, NT, A_KC, B_KC, WM, WN>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)attr;
  hipLaunchKernelGGL((gemm_f32_kernel<MT, NT, A_KC, B_KC, WM, WN>), dim3(wgs), dim3(64 * WM * WN), bytes, s, g);
}

// host-side launcher; picks the 128x128 tile for large outputs and 64x64 when that leaves the chip underfilled, and
// lets a workgroup walk several work items when there are many more items than workgroup slots
template <bool A_KC, bool B_KC>
inline void gemm_launch(hipStream_t s, const GemmArgs& g_in, int force_big = -1) {
  GemmArgs g = g_in;
  int sk = g.splitk > 1 ? g.splitk : 1;
  long big_items = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * sk;
  // a narrow output (the actor's 40-column head) wastes 2/3 of a 128-wide tile's MFMAs and operand traffic: 64-wide tiles there
  bool big = force_big >= 0 ? force_big != 0 : (big_items >= 192 && g.N > 64);
  long items = big ? big_items : (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * sk;
  // one work item per workgroup (walking several items per workgroup was slower at every setting for this path's shapes)
  int wgs = (int)items;
  wgs = (wgs + 7) / 8 * 8;
  if (const int form = gemm_x3_form(g, A_KC, B_KC, big)) {
    if constexpr (A_KC && B_KC) {
      if (form == 2) { if (big) gemm_x3_launch_tile<2, true, true, true>(s, g, sk); else gemm_x3_launch_tile<1, true, true, true>(s, g, sk); }
      else { if (big) gemm_x3_launch_tile<2, true, true, false>(s, g, sk); else gemm_x3_launch_tile<1, true, true, false>(s, g, sk); }
    } else gemm_x3_launch_tile<2, A_KC, B_KC, false>(s, g, sk);
    return;
  }
  // force_big == 2 (A k-contiguous, no split-K): 64 x 128 tiles on 8 wavefronts (2 x 4, each 32 x 32; 55 KB of LDS, two workgroups per CU).
  // For a product with FEW 128 x 128 tiles - the critic's input projection R x 256 x 476 and the input gradients R x 256 x 1024 are 800 tiles on 512
  // workgroup slots, 1.56 rounds paid as 2 - the 1600 half-height tiles fill 3.1 rounds: 118 instead of 132-146 us and 225 instead of 254 us alone
  // (tools/gemm_bench 12, profiles/r06t_*), on the chain a minibatch waits for.
  if constexpr (A_KC) {
    if (force_big == 2 && sk == 1) {
      KbjKernelTimer timer(s, KBJ_KIND_GEMM_64x128 + (B_KC ? 0 : 1), 2.0 * g.M * g.N * g.K);
      const long it = (long)((g.M + 63) / 64) * ((g.N + 127) / 128);
      gemm_launch_tile<1, 1, true, B_KC, 2, 4>(s, g, (int)((it + 7) / 8 * 8));
      return;
    }
  }
  KbjKernelTimer timer(s, KBJ_KIND_GEMM + (big ? 0 : 4) + (A_KC ? 2 : 0) + (B_KC ? 1 : 0), 2.0 * g.M * g.N * g.K);
#ifdef KBJ_GEMM_W4   // 128x128 tile on 4 wavefronts (2 x 2, each 64x64, 207 registers): 2-4 % faster only at K >= 4096
  if (big) gemm_launch_tile<2, 2, A_KC, B_KC>(s, g, wgs);
#else                // 128x128 tile on 8 wavefronts (2 x 4, each 64x32, 99 registers): 4 waves per SIMD at two workgroups per CU
  if (big) gemm_launch_tile<2, 1, A_KC, B_KC, 2, 4>(s, g, wgs);
#endif
  else gemm_launch_tile<1, 1, A_KC, B_KC>(s, g, wgs);
  if (sk > 1 && g.skws) {
    const int per = ((g.K + sk - 1) / sk + GEMM_BK - 1) / GEMM_BK * GEMM_BK;   // as gemm_f32_kernel slices k
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)g.M * g.N + 255) / 256)), dim3(256), 0, s, g, sk, per);
  }
}

}  // namespace kbj
