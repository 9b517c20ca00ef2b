// kbj_lstm_bwd16.h — backward LSTM recurrence (BPTT), 16-row x 64-unit workgroup tiles, partner-major contraction (round 5).
//
// Same contract as lstm_seq_bwd_kernel (kbj_lstm_seq.h): one persistent launch runs all T steps of one (net, layer) downwards in t; per step
//   dh_t = dHabove_t + keep_t * (dG_{t+1} W_hh),  cell derivative -> dG_t [4H] (hand-off payload AND the input of the batched dX / dW GEMMs).
// What changed is the tiling, chosen from the stamps of the 32 x 32 form (8.2 k of 17.3 k cycles per step were MFMA, the rest
// the hand-off: publish 0.9 k + flag wait 1.0 k + first-chunk latency 1.5 k + 4 x staging, with a 128 KB tile gathered per workgroup and step):
//   * a workgroup owns 16 rows x 64 hidden units: the K = 4H contraction's tile is 16 rows x 4H = 64 KB, not 128 KB, from H / 64 = 4 partners
//     instead of 8; the weight slice is [4H][64] = 128 registers per lane (8 wavefronts = 4 unit tiles x 2 k halves);
//   * the contraction is ordered BY PARTNER, not by gate: chunk p = the 4 x 64 dG columns partner p produced (k = gate * 64 + unit inside the
//     chunk). The workgroup's OWN chunk - a quarter of K - never leaves the CU: the cell writes it into LDS beside the write-through stores,
//     and its MFMAs run while the flags and the partners' chunks are in flight - 2 k of the 8.2 k MFMA cycles of a step sit inside the
//     hand-off latency instead of behind it;
//   * all partner chunks (3 x 16 KB) are requested at once, right behind the flag poll; they land in LDS buffers that alternate, one barrier
//     per chunk.
// Hidden sizes 64 / 128 / 192 / 256 (1..4 partners). The bytes WRITTEN through per step in front of the flag are those of the 32 x 32 form
// (the 16 KB of dG the GEMMs need anyway): the lesson of the reduce-scatter experiment.
#pragma once
#include "kbj_lstm_seq.h"

namespace kbj {

constexpr int BWD16_ROWS = 16, BWD16_UNITS = 64, BWD16_NTH = 512;
#ifndef KBJ_BWD16_OWN_SPLIT
#define KBJ_BWD16_OWN_SPLIT 4   // 16-k blocks (of 8) of the own chunk's contraction issued BEFORE the flag poll; the rest hides the partner chunks' flight
#endif

template <int H>
__global__ __launch_bounds__(BWD16_NTH) void lstm_seq_bwd16_kernel(SeqBwdArgs a) {
  static_assert(H % 64 == 0 && H >= 64 && H <= 256, "16 x 64 tiles: hidden sizes 64, 128, 192, 256");
  constexpr int NTH = BWD16_NTH, ROWS = BWD16_ROWS, UNITS = BWD16_UNITS;
  constexpr int NUG = H / UNITS;            // partners of a row group (this workgroup included)
  constexpr int KC = 4 * UNITS;             // contraction length of one partner chunk: 4 gates x 64 units
  constexpr int LDC = KC + 8;               // row stride = 8 (mod 16) words: conflict-free ds_read_b128 fragments (SeqK)
  constexpr int KHALF = KC / 2, KS = KHALF / 4;   // a wavefront contracts half a chunk: 32 k-steps
  typedef SeqK<KHALF> KK;
  __shared__ __attribute__((aligned(16))) float cbuf[2][ROWS * LDC];      // chunk buffers: own chunk in [0], partner chunks alternate [1], [0], [1]
  __shared__ float pbuf[2][ROWS][UNITS + 4];                               // the two k halves' partial sums of dh
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = wave & 3, kh = wave >> 2;
  // XCD-aware mapping (speed only): workgroups b, b + 8, ... share an XCD, so give each XCD whole row groups
  const int nblk = gridDim.x;
  const int lid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ug = lid % NUG, rg = lid / NUG;
  const int r0 = rg * ROWS, u0 = ug * UNITS;
  const int B = a.B, T = a.T;
  if (seq_aborted(a.err, &flag)) return;
  // B operands. Chunk j of this workgroup = partner (ug + j) % NUG (own chunk first); k-step s of the wave's half: local k = KHALF kh + KK::kidx(s, g),
  // i.e. gate = local k / 64, unit = local k % 64 of that partner; column = this wave's unit tile.
  float wreg[NUG][KS];
#pragma unroll
  for (int j = 0; j < NUG; ++j) {
    const int p = (ug + j) % NUG;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kl = KHALF * kh + KK::kidx(s, lane >> 4);
      wreg[j][s] = a.Whh[(size_t)((kl / UNITS) * H + UNITS * p + (kl % UNITS)) * H + u0 + 16 * nt + (lane & 15)];
    }
  }
  // the two (row, unit) elements of this thread in the cell derivative
  int erow[2], eunit[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int e = tid + NTH * i; erow[i] = e / UNITS; eunit[i] = e % UNITS; }
  float dcm[2] = {0.0f, 0.0f};
  float bsum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  // everything the cell derivative of a step needs (produced by earlier kernels) is fetched ONE STEP AHEAD
  float actn[2][4], tcn[2], cprevn[2], dhan[2], kpn[2];
  // NO SELECTS (round 6): a row beyond B or a step before 0 is fetched from a clamped address and its values are never used - rows are independent in
  // the contraction, every global store and bias sum below is guarded by r < B. A select per fetched value (and per staged chunk piece) was 63 of the
  // loop's 169 vector instructions per step, and fp32 MFMA shares the issue port with them: -0.55 % per iteration in A/B (profiles/r06q_*). The same
  // loads as buffer instructions on a scalar per-step offset (67 instructions) LOST that gain again: plain global loads / stores it stays.
  int rcl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int r = r0 + erow[i]; rcl[i] = r < B ? r : 0; }
  auto fetch_inputs = [&](int tt) {
    const int tq = tt < 0 ? 0 : tt;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t o1 = ((size_t)tq * B + rcl[i]) * H + u0 + eunit[i];
      const float* g = a.Gact + ((size_t)tq * B + rcl[i]) * 4 * H + u0 + eunit[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) actn[i][k] = g[k * H];
      tcn[i] = a.TanhC[o1];
      cprevn[i] = a.Cm[o1];
      dhan[i] = a.dHabove[o1];
      kpn[i] = a.keep[(size_t)tq * B + rcl[i]];
    }
  };
  fetch_inputs(T - 1);
  // one partner chunk = 16 rows x (4 gates x 64 floats): 1024 16-byte pieces, two per thread (sc1 buffer loads: hand-off payload)
  struct Chunk { f32x4m v[2]; };
  auto chunk_load = [&](Chunk& c, const float* src, int p) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + NTH * i, row = q >> 6, seg = q & 63, r = r0 + row;
      const unsigned off = (unsigned)(((size_t)(r < B ? r : 0) * 4 * H + (size_t)(seg >> 4) * H + UNITS * p + 4 * (seg & 15)) * sizeof(float));
      c.v[i] = __builtin_bit_cast(f32x4m, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
    }
  };
  auto chunk_to_lds = [&](const Chunk& c, float* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + NTH * i, row = q >> 6, seg = q & 63;
      *reinterpret_cast<f32x4m*>(buf + row * LDC + UNITS * (seg >> 4) + 4 * (seg & 15)) = c.v[i];   // (rows beyond B hold a clamped row's values: harmless, see fetch_inputs)
    }
  };
  // acc += A[16 rows][k of blocks jb..je) of this wave's half] * w  (two accumulators: two independent MFMA chains)
  auto mma = [&](const float* buf, const float* w, f32x4m& acc0, f32x4m& acc1, int jb, int je) {
    const float* p0 = buf + (lane & 15) * LDC + KHALF * kh + 4 * (lane >> 4);
#pragma unroll
    for (int j = jb; j < je; ++j) {
      const f32x4m f = *reinterpret_cast<const f32x4m*>(p0 + 16 * j);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0], w[4 * j + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1], w[4 * j + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[2], w[4 * j + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[3], w[4 * j + 3], acc1, 0, 0, 0);
    }
  };
  constexpr int OWN_SPLIT = NUG > 1 ? KBJ_BWD16_OWN_SPLIT : KK::NB;
  for (int t = T - 1; t >= 0; --t) {
    float dhm[2] = {0.0f, 0.0f};
    float act[2][4], tc[2], cprev[2], dha[2], kp[2];
    auto prefetch = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) { tc[i] = tcn[i]; cprev[i] = cprevn[i]; dha[i] = dhan[i]; kp[i] = kpn[i]; for (int k = 0; k < 4; ++k) act[i][k] = actn[i][k]; }
      fetch_inputs(t - 1);
    };
    if (t == T - 1) prefetch();
    else {
      // the own chunk of dG_{t+1} sits in cbuf[0] (written by the cell of the previous iteration, made visible by its publish barrier)
      f32x4m acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      mma(cbuf[0], wreg[0], acc0, acc1, 0, OWN_SPLIT);
      Chunk ch[NUG > 1 ? NUG - 1 : 1];
      if (NUG > 1) {
        if (!seq_wait(a.counters + rg * NUG, NUG, (unsigned)(T - 1 - t), a.err, &flag, a.timeout_ticks)) return;
        const float* src = a.dG + (size_t)(t + 1) * B * 4 * H;
#pragma unroll
        for (int j = 1; j < NUG; ++j) chunk_load(ch[j - 1], src, (ug + j) % NUG);    // every partner chunk in flight at once
        __builtin_amdgcn_sched_barrier(0);   // keep the loads' issue in front of the MFMAs below
      }
      mma(cbuf[0], wreg[0], acc0, acc1, OWN_SPLIT, KK::NB);
      if (NUG == 1) prefetch();
#pragma unroll
      for (int j = 1; j < NUG; ++j) {
        float* buf = cbuf[j & 1];
        chunk_to_lds(ch[j - 1], buf);      // (vector-memory operations retire in order: this waits for chunk j only)
        if (j == NUG - 1) prefetch();      // own inputs of the next step: issued behind the LAST payload wait, they land during the cell and the hand-off
        __syncthreads();                   // chunk j staged; every wavefront is past its reads of the buffer chunk j + 1 will overwrite
        mma(buf, wreg[j], acc0, acc1, 0, KK::NB);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pbuf[kh][(lane >> 4) * 4 + r][16 * nt + (lane & 15)] = acc0[r] + acc1[r];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) dhm[i] = pbuf[0][erow[i]][eunit[i]] + pbuf[1][erow[i]][eunit[i]];
    }
    // cell derivative; dG_t goes out write-through (hand-off payload + GEMM input) and, as this workgroup's own chunk of the next step, into cbuf[0]
    // (every wavefront is past the contraction: the barrier above, or no contraction at all in the first step)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + erow[i];
      const float ig = act[i][0], fg = act[i][1], gg = act[i][2], og = act[i][3];
      const float dh = dha[i] + kp[i] * dhm[i];
      const float dc = kp[i] * dcm[i] + dh * og * (1 - tc[i] * tc[i]);
      const float d0 = dc * gg * ig * (1 - ig), d1 = dc * cprev[i] * fg * (1 - fg), d2 = dc * ig * (1 - gg * gg), d3 = dh * tc[i] * og * (1 - og);
      float* own = cbuf[0] + erow[i] * LDC + eunit[i];      // (rows beyond B carry a clamped row's values: harmless, rows are independent)
      own[0] = d0; own[UNITS] = d1; own[2 * UNITS] = d2; own[3 * UNITS] = d3;
      if (r < B) {
        float* dg = a.dG + ((size_t)t * B + r) * 4 * H + u0 + eunit[i];
        seq_store(dg, d0); seq_store(dg + H, d1); seq_store(dg + 2 * H, d2); seq_store(dg + 3 * H, d3);
        bsum[i][0] += d0; bsum[i][1] += d1; bsum[i][2] += d2; bsum[i][3] += d3;
      }
      dcm[i] = dc * fg;
    }
    seq_publish(a.counters + rg * NUG + ug, (unsigned)(T - t));    // drain, barrier (also publishes cbuf[0] inside the workgroup), flag
  }
  // bias gradient = column sums of dG over all rows and steps: this workgroup's 16 rows through cbuf[0] (the own-chunk layout), one atomic per column
  if (a.db || a.db_part) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) cbuf[0][erow[i] * LDC + k * UNITS + eunit[i]] = bsum[i][k];
    __syncthreads();
    if (tid < KC) {
      const int k = tid / UNITS, u = tid % UNITS;
      float s = 0;
      for (int r = 0; r < ROWS; ++r) s += cbuf[0][r * LDC + tid];
      if (a.db_part) a.db_part[(size_t)rg * 4 * H + k * H + u0 + u] = s;
      else atomicAdd(a.db + k * H + u0 + u, s);
    }
  }
}

}  // namespace kbj
