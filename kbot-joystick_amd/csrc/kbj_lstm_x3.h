// kbj_lstm_x3.h — the persistent LSTM recurrences on the bf16 matrix cores through the exact three-way split (kbj_config.gemm_bf16x3 only:
// NOT the default, never the headline).
//
// fp32 MFMA runs at the vector rate and holds the SIMD's vector issue; v_mfma_f32_16x16x32_bf16 moves 8x the FLOPs per
// instruction in half the cycles. An fp32 number is EXACTLY hi + mid + lo with three bf16 pieces (kbj_gemm.h x3_split); a product of two such
// numbers is the sum of six piece products down to 2^-23 relative, each exact in the instruction's fp32 accumulation. Here:
//   * the recurrent weight slice of a wavefront lives in registers as three bf16 pieces (96 registers for H = 256, K = H per wavefront; the
//     fp32 form of kbj_lstm_seq.h holds 64 for W_hh + 64 for the fused W_ih - which is why the input projection is NOT fused in this form: it
//     runs as a gemm_x3_kernel launch in front of the recurrence, where it costs 0.375 of its fp32 time);
//   * the h tile of a step (32 rows x H, fp32 hand-off payload as before) is split once per element on its way into LDS ([3][32][H + 8] bf16,
//     row stride 4 (mod 64) words: conflict-free ds_read_b128 fragments);
//   * per step and wavefront: H / 32 k blocks x 6 products x 2 row tiles = 96 MFMAs of 16 cycles (H = 256) instead of 128 of 32.
// Hand-off protocol, cell, stash and failure behaviour are those of lstm_seq_fwd_kernel / lstm_seq_bwd16_kernel.
#pragma once
#include "kbj_gemm.h"
#include "kbj_lstm_seq.h"

namespace kbj {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// 8 consecutive k of one row / column -> three bf16x8 fragments (element j of a fragment = k j)
__device__ __forceinline__ void x3_split8(const float* x, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x2 h0, m0, l0, h1, m1, l1;
  x3_split(f32x4{x[0], x[1], x[2], x[3]}, h0, m0, l0);
  x3_split(f32x4{x[4], x[5], x[6], x[7]}, h1, m1, l1);
  hi = __builtin_bit_cast(bf16x8, u32x4s{h0[0], h0[1], h1[0], h1[1]});
  mid = __builtin_bit_cast(bf16x8, u32x4s{m0[0], m0[1], m1[0], m1[1]});
  lo = __builtin_bit_cast(bf16x8, u32x4s{l0[0], l0[1], l1[0], l1[1]});
}
// acc += (a_hi + a_mid + a_lo)(b_hi + b_mid + b_lo) without mid lo, lo mid, lo lo; smallest products first
__device__ __forceinline__ void x3_mma(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4m& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}

// ---- forward: gates_t = G_t (= x_t W_ih^T + b, precomputed) + h_{t-1} W_hh^T ------------------------------------------------------------
// Workgroup = 32 rows x 32 units on 8 wavefronts (wavefront = one gate's 16 units, as lstm_seq_fwd_kernel<H, 2, false>).
template <int H>
__global__ __launch_bounds__(512) void lstm_seq_fwd_x3_kernel(SeqFwdArgs a) {
  static_assert(H % 64 == 0 && H <= 256, "hidden sizes 64, 128, 192, 256");
  constexpr int NTH = 512, UNITS = 32, NUG = H / UNITS, KB = H / 32;
  constexpr int LDA = H + 8;        // bf16 elements per row of a piece: (H + 8) / 2 words = 4 (mod 32) words
  __shared__ __attribute__((aligned(16))) short as[3][SEQ_ROWS * LDA];
  __shared__ float gbuf[4][SEQ_ROWS][UNITS + 4];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, gate = (tid >> 6) & 3, uh = tid >> 8;
  const int nblk = gridDim.x;
  const int lid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ug = lid % NUG, rg = lid / NUG;
  const int r0 = rg * SEQ_ROWS, u0 = ug * UNITS;
  const int B = a.B, T = a.T;
  // B operands: lane (col = lane & 15, k group g = lane >> 4) holds W_hh[gate H + u0 + 16 uh + col][32 kb + 8 g .. + 7] as three bf16x8
  bf16x8 wq[KB][3];
  {
    const float* wrow = a.Whh + (size_t)(gate * H + u0 + SEQ_UNITS * uh + (lane & 15)) * H + 8 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4m q0 = *reinterpret_cast<const f32x4m*>(wrow + 32 * kb), q1 = *reinterpret_cast<const f32x4m*>(wrow + 32 * kb + 4);
      const float x[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      x3_split8(x, wq[kb][0], wq[kb][1], wq[kb][2]);
    }
  }
  int erow[2], eunit[2];
  float cm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + NTH * i;
    erow[i] = e / UNITS; eunit[i] = e % UNITS;
    const int r = r0 + erow[i];
    cm[i] = r < B ? a.Cm[(size_t)r * H + u0 + eunit[i]] : 0.0f;
  }
  float ig[2], fg[2], gg[2], og[2], tc[2], hh[2];  // results of the previous step, stored lazily
  float gxn[2][4], kpn[2];
  auto fetch_inputs = [&](int tt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + erow[i];
      const bool ok = r < B && tt < T;
      const float* g = a.G + ((size_t)(ok ? tt : 0) * B + (ok ? r : 0)) * 4 * H + u0 + eunit[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) gxn[i][k] = ok ? g[k * H] : 0.0f;
      kpn[i] = ok ? a.keep[(size_t)tt * B + r] : 0.0f;
    }
  };
  fetch_inputs(0);
  const bool full = r0 + SEQ_ROWS <= B;
  for (int t = 0; t < T; ++t) {
    float gx[2][4], kp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { kp[i] = kpn[i]; for (int k = 0; k < 4; ++k) gx[i][k] = gxn[i][k]; }
    if (t > 0) { if (!seq_wait(a.counters + rg * NUG, NUG, (unsigned)t, a.err, &flag, a.spin_limit)) return; }
    SeqTile<H, NTH> tile;
    tile.load(a.Hm + (size_t)t * B * H, H, r0, B);
    // split on the way into LDS: piece p of (row, 4 consecutive k) = 8 bytes
#pragma unroll
    for (int i = 0; i < SeqTile<H, NTH>::NV; ++i) {
      const int q = tid + NTH * i, row = q / (H / 4), c4 = q % (H / 4);
      if (!SeqTile<H, NTH>::EXACT && q >= SeqTile<H, NTH>::NQ) continue;
      const f32x4m v = (full || r0 + row < B) ? tile.v[i] : f32x4m{0, 0, 0, 0};
      u32x2 hi, mid, lo;
      x3_split(f32x4{v[0], v[1], v[2], v[3]}, hi, mid, lo);
      short* p = as[0] + row * LDA + 4 * c4;
      *reinterpret_cast<u32x2*>(p) = hi; *reinterpret_cast<u32x2*>(p + SEQ_ROWS * LDA) = mid; *reinterpret_cast<u32x2*>(p + 2 * SEQ_ROWS * LDA) = lo;
    }
    fetch_inputs(t + 1);
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r0 + erow[i];
        if (r >= B) continue;
        const size_t o1 = ((size_t)(t - 1) * B + r) * H + u0 + eunit[i];
        float* g = a.G + ((size_t)(t - 1) * B + r) * 4 * H + u0 + eunit[i];
        g[0] = ig[i]; g[H] = fg[i]; g[2 * H] = gg[i]; g[3 * H] = og[i];
        a.Hout[o1] = hh[i]; a.TanhC[o1] = tc[i];
        a.Cm[o1 + (size_t)B * H] = cm[i];
      }
    }
    __syncthreads();
    f32x4m acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    {
      const short* p0 = as[0] + (lane & 15) * LDA + 8 * (lane >> 4);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 a0[3], a1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a0[p] = *reinterpret_cast<const bf16x8*>(p0 + p * SEQ_ROWS * LDA + 32 * kb);
          a1[p] = *reinterpret_cast<const bf16x8*>(p0 + p * SEQ_ROWS * LDA + 16 * LDA + 32 * kb);
        }
        x3_mma(a0, wq[kb], acc0);
        x3_mma(a1, wq[kb], acc1);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gbuf[gate][(lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc0[r];
      gbuf[gate][16 + (lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc1[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = erow[i], u = eunit[i];
      ig[i] = seq_sigmoid(gbuf[0][row][u] + gx[i][0]); fg[i] = seq_sigmoid(gbuf[1][row][u] + gx[i][1]);
      gg[i] = seq_tanh(gbuf[2][row][u] + gx[i][2]); og[i] = seq_sigmoid(gbuf[3][row][u] + gx[i][3]);
      const float c = fg[i] * cm[i] + ig[i] * gg[i];
      tc[i] = seq_tanh(c); hh[i] = og[i] * tc[i];
      cm[i] = c * kp[i];
      if (r0 + row < B) seq_store(a.Hm + ((size_t)(t + 1) * B + r0 + row) * H + u0 + u, hh[i] * kp[i]);  // the hand-off payload
    }
    seq_publish(a.counters + rg * NUG + ug, (unsigned)(t + 1));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + erow[i];
    if (r >= B) continue;
    const size_t o1 = ((size_t)(T - 1) * B + r) * H + u0 + eunit[i];
    float* g = a.G + ((size_t)(T - 1) * B + r) * 4 * H + u0 + eunit[i];
    g[0] = ig[i]; g[H] = fg[i]; g[2 * H] = gg[i]; g[3 * H] = og[i];
    a.Hout[o1] = hh[i]; a.TanhC[o1] = tc[i];
    a.Cm[o1 + (size_t)B * H] = cm[i];
  }
}

// (The backward recurrence stays on the fp32 form, lstm_seq_bwd16_kernel: with the matrix time of a step cut to 0.375 its hand-off - 48 KB gathered
// and, here, split per step and workgroup, or 1.5x the write-through bytes if the producers published the pieces - is what a step costs, and a
// 16 x 64 tile's weight slice as three bf16 pieces (192 registers) does not fit beside the rest; the 16 x 32 tile that does needs 2 x 256
// resident workgroups. Estimated gain: none. DESIGN.md section 10b.)

}  // namespace kbj
