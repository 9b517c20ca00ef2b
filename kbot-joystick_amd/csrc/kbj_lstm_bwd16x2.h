// kbj_lstm_bwd16x2.h — backward LSTM recurrence with TWO independent 16-row groups per workgroup, software-interleaved (round 6 experiment, KBJ_BWD16X2=1).
//
// Same contract as lstm_seq_bwd16_kernel (kbj_lstm_bwd16.h). A workgroup owns 2 x 16 rows x 32 hidden units: its W_hh slice is [4H][32] = 64 registers per
// lane, the two row groups A and B are independent chains that share it. While one group contracts (64 MFMAs per wavefront) the OTHER group's hand-off
// is in flight in the same instruction stream: its stores drain, its flag goes out, its partners' flags are read and its seven partner chunks are
// requested - so the publish / flag / chunk latencies of a step sit under the other group's matrix instructions instead of in front of its own.
// The price is the gather volume of the 32-unit form (16 rows x 4H per group and step = 2 x 64 KB per workgroup).
// H = 256 only (8 partners); anything else stays on lstm_seq_bwd16_kernel.
#pragma once
#include "kbj_lstm_seq.h"

namespace kbj {

constexpr int BWDX2_ROWS = 16, BWDX2_UNITS = 32, BWDX2_NTH = 512;
template <int H> constexpr size_t bwdx2_lds_bytes() {
  return ((size_t)(H / BWDX2_UNITS - 1 + 2) * BWDX2_ROWS * (4 * BWDX2_UNITS + 8) + 4 * BWDX2_ROWS * (BWDX2_UNITS + 4)) * sizeof(float);
}

template <int H>
__global__ __launch_bounds__(BWDX2_NTH) void lstm_seq_bwd16x2_kernel(SeqBwdArgs a) {
  static_assert(H == 256, "two-group form: written for 8 partners");
  constexpr int NTH = BWDX2_NTH, ROWS = BWDX2_ROWS, UNITS = BWDX2_UNITS;
  constexpr int NUG = H / UNITS, NP = NUG - 1;   // partners of a row group (this workgroup included), partner chunks
  constexpr int KC = 4 * UNITS;                  // contraction length of one partner chunk: 4 gates x 32 units
  constexpr int LDC = KC + 8;                    // row stride = 8 (mod 16) words: conflict-free ds_read_b128 fragments (SeqK)
  constexpr int KQ = KC / 4, KS = KQ / 4;        // a wavefront contracts a quarter of a chunk: 8 k-steps
  constexpr int TILE = ROWS * LDC;
  typedef SeqK<KQ> KK;
  extern __shared__ __attribute__((aligned(16))) float x2lds[];
  float* tile = x2lds;                           // [NP][TILE] partner chunks of the group being contracted
  float* ownb = tile + NP * TILE;                // [2][TILE]  own chunk of each group
  float* pbuf = ownb + 2 * TILE;                 // [4][ROWS][UNITS + 4] partial sums of the four k quarters
  __shared__ int flag;
  __shared__ int lds_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = wave & 1, kq = wave >> 1;
  const int nblk = gridDim.x;
  const int lid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ug = lid % NUG, rp = lid / NUG, u0 = ug * UNITS;
  const int B = a.B, T = a.T;
  if (seq_aborted(a.err, &flag)) return;
  if (tid == 0) lds_abort = 0;
  // B operands: chunk j = partner (ug + j) % NUG (own chunk first); k-step s of this wave's quarter: local k = KQ kq + KK::kidx(s, g), gate = k / 32, unit = k % 32
  float wreg[NUG][KS];
#pragma unroll
  for (int j = 0; j < NUG; ++j) {
    const int p = (ug + j) % NUG;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kl = KQ * kq + KK::kidx(s, lane >> 4);
      wreg[j][s] = a.Whh[(size_t)((kl / UNITS) * H + UNITS * p + (kl % UNITS)) * H + u0 + 16 * nt + (lane & 15)];
    }
  }
  const int erow = tid / UNITS, eunit = tid % UNITS;     // this thread's (row, unit) element of each group's cell derivative
  int r0[2], rcl[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) { r0[g] = (2 * rp + g) * ROWS; const int r = r0[g] + erow; rcl[g] = r < B ? r : 0; }
  float dcm[2] = {0.0f, 0.0f}, bsum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float actn[2][4], tcn[2], cprevn[2], dhan[2], kpn[2];
  auto fetch_inputs = [&](int g, int tt) {       // one step ahead; clamped addresses, never-used values (kbj_lstm_bwd16.h)
    const int tq = tt < 0 ? 0 : tt;
    const size_t o1 = ((size_t)tq * B + rcl[g]) * H + u0 + eunit;
    const float* gp = a.Gact + ((size_t)tq * B + rcl[g]) * 4 * H + u0 + eunit;
#pragma unroll
    for (int k = 0; k < 4; ++k) actn[g][k] = gp[k * H];
    tcn[g] = a.TanhC[o1]; cprevn[g] = a.Cm[o1]; dhan[g] = a.dHabove[o1]; kpn[g] = a.keep[(size_t)tq * B + rcl[g]];
  };
  fetch_inputs(0, T - 1); fetch_inputs(1, T - 1);
  // one partner chunk = 16 rows x (4 gates x 32 floats) = 512 16-byte pieces: one per thread (sc1 buffer loads: hand-off payload)
  f32x4m ch[NP];
  const int crow = tid >> 5, cseg = tid & 31;
  auto chunks_load = [&](int g, int tsrc) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dG + (size_t)tsrc * B * 4 * H), 0, 0x7FFFFFFF, 0x00020000);
    const int r = r0[g] + crow;
    const size_t base = (size_t)(r < B ? r : 0) * 4 * H + (size_t)(cseg >> 3) * H + 4 * (cseg & 7);
#pragma unroll
    for (int j = 1; j < NUG; ++j) {
      const unsigned off = (unsigned)((base + (size_t)UNITS * ((ug + j) % NUG)) * sizeof(float));
      ch[j - 1] = __builtin_bit_cast(f32x4m, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
    }
  };
  auto chunks_to_lds = [&]() {
#pragma unroll
    for (int j = 0; j < NP; ++j) *reinterpret_cast<f32x4m*>(tile + j * TILE + crow * LDC + UNITS * (cseg >> 3) + 4 * (cseg & 7)) = ch[j];
  };
  auto mma = [&](const float* buf, const float* w, f32x4m& acc0, f32x4m& acc1) {   // one chunk: this wave's k quarter, 8 matrix instructions
    const float* p0 = buf + (lane & 15) * LDC + KQ * kq + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < KK::NB; ++j) {
      const f32x4m f = *reinterpret_cast<const f32x4m*>(p0 + 16 * j);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0], w[4 * j + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1], w[4 * j + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[2], w[4 * j + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[3], w[4 * j + 3], acc1, 0, 0, 0);
    }
  };
  // per-wavefront flag check of a row group (lanes 0..NUG-1 hold one partner's counter each): non-blocking read first, polled only if a partner is late
  auto flags_ready = [&](const unsigned* flags, unsigned first, unsigned target) -> bool {
    unsigned v = first;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
      if (__all(lane >= NUG || v >= target)) return true;
      __builtin_amdgcn_s_sleep(1);
      if (lane < NUG) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((++spins & 255u) == 0) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        const bool aborted = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (aborted || now - t0 > a.timeout_ticks) {
          if (lane == 0) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); lds_abort = 1; }
          return false;
        }
      }
    }
  };
  unsigned pend[2] = {0u, 0u};     // steps-done value of a group whose dG stores are issued but not yet published (0 = nothing pending)
  __syncthreads();
  // one step of group G at time t; the hand-off of the other group O rides inside it
  auto phase = [&](auto g_tag, int t) -> bool {
    constexpr int G = decltype(g_tag)::value, O = 1 - G;
    const bool has = t < T - 1;                           // first step: dh comes from above only
    const int t_o = G == 0 ? t : t - 1;                   // the step group O runs next
    const bool fetch_o = t_o >= 0 && t_o < T - 1;         // ... needs its partners' dG_{t_o + 1}
    unsigned* oflags = a.counters + (size_t)(2 * rp + O) * NUG;
    float act[4], tc, cprev, dha, kp;
#pragma unroll
    for (int k = 0; k < 4; ++k) act[k] = actn[G][k];
    tc = tcn[G]; cprev = cprevn[G]; dha = dhan[G]; kp = kpn[G];
    if (G == 0) SEQ_BSTAMP(0);
    if (has) chunks_to_lds();                             // (vector-memory operations retire in order: waits for the chunk loads only)
    __syncthreads();                                      // tile staged; every wavefront is past the previous phase's reads of it; own[G] long visible
    if (lds_abort) return false;
    if (G == 0) SEQ_BSTAMP(1);
    f32x4m acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float* own = ownb + G * TILE;
    // the contraction in four quarters of 16 matrix instructions per wavefront (~1 k cycles of the SIMD each); group O's hand-off rides between them:
    // publish behind the first (its stores have been draining), partner flags read behind the second (they publish at the same point of THEIR
    // schedule, one quarter earlier), checked - and the partner chunks requested - behind the third, so the chunks land under the fourth and the cell
    if (has) { mma(own, wreg[0], acc0, acc1); mma(tile, wreg[1], acc0, acc1); }
    if (G == 0) SEQ_BSTAMP(2);
    if (pend[O]) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.counters + (size_t)(2 * rp + O) * NUG + ug, pend[O], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pend[O] = 0;
    }
    if (G == 0) SEQ_BSTAMP(3);
    if (has) { mma(tile + 1 * TILE, wreg[2], acc0, acc1); mma(tile + 2 * TILE, wreg[3], acc0, acc1); }
    unsigned fv = 0;
    if (fetch_o && lane < NUG) fv = __hip_atomic_load(oflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    if (G == 0) SEQ_BSTAMP(4);
    if (has) { mma(tile + 3 * TILE, wreg[4], acc0, acc1); mma(tile + 4 * TILE, wreg[5], acc0, acc1); }
    if (G == 0) SEQ_BSTAMP(5);
    if (fetch_o) {
#ifndef KBJ_X2_NOWAIT
      if (flags_ready(oflags, fv, (unsigned)(T - 1 - t_o)))
#endif
      {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        chunks_load(O, t_o + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (G == 0) SEQ_BSTAMP(6);
    float dhm = 0.0f;
    if (has) {
      mma(tile + 5 * TILE, wreg[6], acc0, acc1); mma(tile + 6 * TILE, wreg[7], acc0, acc1);
#pragma unroll
      for (int r = 0; r < 4; ++r) pbuf[(kq * ROWS + (lane >> 4) * 4 + r) * (UNITS + 4) + 16 * nt + (lane & 15)] = acc0[r] + acc1[r];
    }
    fetch_inputs(G, t - 1);                               // own inputs of this group's next step
    __syncthreads();
    if (lds_abort) return false;
    if (G == 0) SEQ_BSTAMP(7);
    if (has) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dhm += pbuf[(q * ROWS + erow) * (UNITS + 4) + eunit];
    }
    // cell derivative; dG_t goes out write-through (hand-off payload + GEMM input) and, as this group's own chunk of its next step, into own[G]
    {
      const int r = r0[G] + erow;
      const float ig = act[0], fg = act[1], gg = act[2], og = act[3];
      const float dh = dha + kp * dhm;
      const float dc = kp * dcm[G] + dh * og * (1 - tc * tc);
      const float d0 = dc * gg * ig * (1 - ig), d1 = dc * cprev * fg * (1 - fg), d2 = dc * ig * (1 - gg * gg), d3 = dh * tc * og * (1 - og);
      float* o = own + erow * LDC + eunit;
      o[0] = d0; o[UNITS] = d1; o[2 * UNITS] = d2; o[3 * UNITS] = d3;
      if (r < B) {
        float* dg = a.dG + ((size_t)t * B + r) * 4 * H + u0 + eunit;
        seq_store(dg, d0); seq_store(dg + H, d1); seq_store(dg + 2 * H, d2); seq_store(dg + 3 * H, d3);
        bsum[G][0] += d0; bsum[G][1] += d1; bsum[G][2] += d2; bsum[G][3] += d3;
      }
      dcm[G] = dc * fg;
    }
    pend[G] = (unsigned)(T - t);
    if (G == 0) SEQ_BSTAMP(8);
    return true;
  };
  for (int t = T - 1; t >= 0; --t) {
    if (!phase(std::integral_constant<int, 0>{}, t)) return;
    if (!phase(std::integral_constant<int, 1>{}, t)) return;
  }
  // (the last publishes - step 0 of either group - have no reader)
  if (a.db || a.db_part) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      __syncthreads();
      float* own = ownb + g * TILE;
#pragma unroll
      for (int k = 0; k < 4; ++k) own[erow * LDC + k * UNITS + eunit] = bsum[g][k];
      __syncthreads();
      if (tid < KC) {
        const int k = tid / UNITS, u = tid % UNITS;
        float s = 0;
        for (int r = 0; r < ROWS; ++r) s += own[r * LDC + tid];
        if (a.db_part) a.db_part[(size_t)(2 * rp + g) * 4 * H + k * H + u0 + u] = s;
        else atomicAdd(a.db + k * H + u0 + u, s);
      }
    }
  }
}

}  // namespace kbj
