// kbj_lstm_seq.h — persistent fused LSTM recurrence over a whole trajectory (forward and BPTT), fp32 MFMA.
//
// The recurrent products h_{t-1} W_hh^T (forward) and dG_{t+1} W_hh (backward) are sequential in t and tiny
// (M = minibatch 512, 268 MFLOP): as separate launches they are latency bound (profiles/r01). Here ONE launch runs all
// T steps of one (net, layer):
//   * workgroup (rg, ug) owns rows [32 rg, 32 rg + 32) x hidden units [16 ug, 16 ug + 16) of the minibatch for the
//     whole sequence; its slice of W_hh (64 x H, resp. 4H x 16 floats) stays in VGPRs as MFMA B-operands
//     (16x16x4 f32 MFMA: A = one f32 per lane [row = lane&15][k = lane>>4], B = [k = lane>>4][col = lane&15]);
//   * the LSTM cell (forward) / its derivative (backward) is fused; the cell state / its gradient lives in registers;
//   * per time step the only exchange is the h tile (forward) or dG tile (backward) of the SAME row group, produced by the
//     H/16 workgroups of that row group. Hand-off protocol (cdna_hip_programming.md Guideline 16, recipe R1): the
//     payload is stored write-through (sc1: relaxed agent-scope atomic stores), every storing wave drains, the
//     workgroup meets, one lane bumps a monotonic per-row-group arrival counter; consumers poll that one word and read
//     the payload with 16-byte sc1 buffer loads (L1 bypass), so neither a release nor an acquire cache operation is
//     needed. The poll is issued with an empty vector-memory queue: stores nobody reads inside the launch (activations
//     kept for BPTT) and prefetches of the next step's own inputs are issued AFTER the payload loads, behind the MFMAs.
//     Grid = (B/32)(H/16) <= 256 workgroups, all resident; every spin is bounded and reports through an error word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kbj {

typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr int SEQ_ROWS = 32;   // rows per workgroup
constexpr int SEQ_UNITS = 16;  // hidden units per workgroup
// Backward-recurrence stamps (KBJ_SEQ_BSTAMPS) exist in diagnostics builds only (-DKBJ_SEQ_BSTAMPS_BUILD, `make bstamps`): the stamp code
// costs 12 VGPRs (186 instead of 174), and at 2 x 192 per SIMD a weight-gradient GEMM workgroup (2 x 80) no longer fits beside a
// backward-recurrence workgroup - 6.46 instead of 6.34 ms per minibatch.
#ifdef KBJ_SEQ_BSTAMPS_BUILD
#define SEQ_BSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[t * 10 + (k)] = clock64(); } while (0)
#define SEQ_BSTAMP_ON 1
#else
#define SEQ_BSTAMP(k) do { } while (0)
#define SEQ_BSTAMP_ON 0
#endif
constexpr unsigned SEQ_TIMEOUT_MS = 2000;   // default wall-clock bound of every inter-workgroup wait (args.timeout_ticks on the 100 MHz constant-rate clock; KBJ_SEQ_TIMEOUT_MS; 20 ms under fault injection)

// Register budget of the backward recurrence: at <= 176 allocated VGPRs two of its waves AND two waves of a weight-gradient GEMM
// workgroup (80) share a SIMD (2 x 176 + 2 x 80 = 512), which is what lets those GEMMs run beside the layer-0 recurrences (DESIGN.md
// section 10); amdgpu_num_vgpr(N) makes hipcc allocate 2 N for a wave64 kernel.
#ifndef KBJ_SEQ_BWD_NUM_VGPR
#define KBJ_SEQ_BWD_NUM_VGPR 88
#endif

struct SeqFwdArgs {
  float* G;            // [T][B][4H] in: x W_ih^T + b, out: gate activations (i,f,g,o)
  const float* Whh;    // [4H][H]
  float* Hm;           // [T+1][B][H] masked h consumed by step t (slot 0 = initial carry)
  float* Cm;           // [T+1][B][H]
  float* Hout;         // [T][B][H]
  float* TanhC;        // [T][B][H]
  const float* keep;   // [T][B]
  unsigned* counters;  // [ceil(B/32)] zeroed before launch
  unsigned* err;       // set to 1 on a spin timeout
  int T, B;
  long long* stamps;   // optional [T][6] shader-clock stamps of workgroup 0 (diagnostics), else null
  unsigned timeout_ticks = SEQ_TIMEOUT_MS * 100000u;   // wall_clock64 ticks (kbj_nn.hip sets it from the device's wall-clock rate)
  // fused input projection (FUSE kernels): gates = x_t W_ih^T + bias + h_{t-1} W_hh^T computed here, G is then output only
  const float* X = nullptr;     // [T][B][H] layer input
  const float* Wih = nullptr;   // [4H][H]
  const float* bias = nullptr;  // [4H]
  int ldx = 0, ldw = 0;         // row strides of X and Wih (0: H)
  int kx = 0;                   // valid input columns (0: all KX); columns beyond are taken as zero whatever the buffers hold
};

struct SeqBwdArgs {
  const float* Gact;     // [T][B][4H]
  const float* TanhC;    // [T][B][H]
  const float* Cm;       // [T+1][B][H]
  const float* dHabove;  // [T][B][H]
  const float* keep;     // [T][B]
  const float* Whh;      // [4H][H]
  float* dG;             // [T][B][4H] out
  unsigned* counters;
  unsigned* err;
  int T, B;
  float* db;             // optional [4H]: bias gradient accumulated with atomics (column sums of dG)
  float* db_part = nullptr;   // deterministic mode: [row groups][4H] per-row-group column sums instead (added to db in order by reduce_rows_kernel)
  unsigned timeout_ticks = SEQ_TIMEOUT_MS * 100000u;   // wall_clock64 ticks (kbj_nn.hip sets it from the device's wall-clock rate)
  long long* stamps = nullptr;   // optional [T][10] shader-clock stamps of workgroup 0 (diagnostics)
};

// gate non-linearities on the hardware exp/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each; shared with the rollout cell kernel)
// (__builtin_amdgcn_rcpf = one v_rcp_f32; __frcp_rn compiles to the 10-instruction correctly rounded division, and on this chip fp32 MFMA and
// the vector ALU share the SIMD's issue: every vector instruction in a recurrence step is matrix time lost - DESIGN.md section 10)
__device__ __forceinline__ float seq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float seq_tanh(float x) { float xc = fminf(fmaxf(x, -15.0f), 15.0f); return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * xc)); }

// wait until every flag >= target; returns false on timeout or abort
// flags: one word per producer workgroup of the row group (nflags consecutive words = one cache line), each holding the
// number of steps that producer has published; lanes 0..nflags-1 of wave 0 poll them with sc1 loads until all reach target.
// The bound is WALL-CLOCK time (round 6; it was 2^24 polls, i.e. tens of seconds per hand-off and minutes per call when a grid could not be
// made resident - gpurun_out/r05r): every 256th poll reads the constant-rate counter (s_memrealtime) and the error word, so the fast path is
// unchanged. A wait gives up when `timeout_ticks` have passed since its first check, or as soon as ANY workgroup of the context has raised the
// error word (a partner that timed out or a launch that was aborted will never publish): a failing call drains in one bound, not in
// steps x bound.
__device__ __forceinline__ bool seq_wait(unsigned* flags, int nflags, unsigned target, unsigned* err, int* lds_flag, unsigned timeout_ticks) {
#ifdef KBJ_EXPERIMENT_NOWAIT   // TIMING EXPERIMENT ONLY (wrong results): what a step costs when the partners' flags never have to be waited for
  __syncthreads();
  return true;
#endif
  if (threadIdx.x < 64) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    int ok = 1;
    const int l = threadIdx.x;
    for (;;) {
      bool mine = l >= nflags || __hip_atomic_load(flags + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
      if (__all(mine)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        const bool aborted = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (aborted || now - t0 > timeout_ticks) { ok = 0; if (l == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    if (l == 0) *lds_flag = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the payload loads below the poll
  return *lds_flag != 0;
}
// a launch of a call whose earlier launch already failed does not start: its partners would only run into their bounds one after the other
// (uniform per workgroup only if every thread reads the same value: thread 0 decides)
__device__ __forceinline__ bool seq_aborted(unsigned* err, int* lds_flag) {
  if (threadIdx.x == 0) *lds_flag = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  __syncthreads();
  const bool r = *lds_flag != 0;
  __syncthreads();
  return r;
}
// payload store: write-through (sc1)
__device__ __forceinline__ void seq_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every storing wave drains its (payload) stores, the workgroup meets, one lane signals
__device__ __forceinline__ void seq_publish(unsigned* my_flag, unsigned steps_done) {
#ifndef KBJ_EXPERIMENT_NODRAIN   // TIMING EXPERIMENT ONLY (wrong results): the flag without the payload's drain in front of it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(my_flag, steps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store, never a plain one
}
// k order of the register-resident products. The A operand of v_mfma_f32_16x16x4_f32 is one float per lane (row = lane & 15,
// k slot g = lane >> 4); WHICH k a slot holds in a given step is free as long as the B registers (the weight slices) use the same
// order. Fragments are fetched as 16-byte ds_read_b128 (k = 16 j + 4 g + p feeds step 4 j + p): a quarter of the LDS instructions of
// the one-float form, and with a row stride = 8 (mod 16) words the 16 lanes of every b128 lane group fall on 64 distinct banks
// (MI355X_MICROARCH.md, LDS: the former stride H + 4 with ds_read_b32 was 2-way conflicted on 32 banks, SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE 0.39-0.46 in profiles/r02b_pmc_mfma.json). A width that is not a multiple of 16 (the 68-float observation
// row) ends in single-float steps (k = 16 NB + 4 q + g).
template <int K> struct SeqK {
  static_assert(K % 4 == 0, "row width must be a multiple of 4");
  static constexpr int NB = K / 16, REM = (K % 16) / 4, STEPS = 4 * NB + REM;
  static constexpr int LD = (K + 7) / 16 * 16 + 8;   // smallest row stride >= K that is 8 (mod 16)
  __host__ __device__ static constexpr int kidx(int s, int g) { return s < 4 * NB ? 16 * (s / 4) + 4 * g + (s % 4) : 16 * NB + 4 * (s - 4 * NB) + g; }
  // acc0 / acc1 += A[rows 0-15 / 16-31][k of blocks jb..je) (and the tail steps when `tail`)] * w; lds: tile [32][LD], w: this lane's STEPS registers
  template <bool TAIL>
  __device__ static __forceinline__ void mma(const float* lds, int lane, const float* w, f32x4m& acc0, f32x4m& acc1, int jb, int je) {
    const float* p0 = lds + (lane & 15) * LD + 4 * (lane >> 4);
    const float* p1 = p0 + 16 * LD;
#pragma unroll
    for (int j = jb; j < je; ++j) {
      const f32x4m a0 = *reinterpret_cast<const f32x4m*>(p0 + 16 * j), a1 = *reinterpret_cast<const f32x4m*>(p1 + 16 * j);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], w[4 * j + q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], w[4 * j + q], acc1, 0, 0, 0);
      }
    }
    if (TAIL) {
      const float* r0 = lds + (lane & 15) * LD + 16 * NB + (lane >> 4);
      const float* r1 = r0 + 16 * LD;
#pragma unroll
      for (int q = 0; q < REM; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r0[4 * q], w[4 * NB + q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r1[4 * q], w[4 * NB + q], acc1, 0, 0, 0);
      }
    }
  }
};

// payload tile: rows [r0, r0+32) x H floats of a handed-off [B][ld] array, 16-byte buffer loads with the sc1 bit
// (aux = 16: bypass this CU's L1; counted by the compiler's s_waitcnt), then written to LDS as [32][SeqK<H>::LD]
template <int H, int NTH> struct SeqTile {
  static constexpr int NQ = SEQ_ROWS * (H / 4);          // 16-byte elements of the tile
  static constexpr int NV = (NQ + NTH - 1) / NTH;
  static constexpr bool EXACT = NQ % NTH == 0;
  f32x4m v[NV];
  __device__ __forceinline__ void load(const float* src, int ld, int r0, int B) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int q = threadIdx.x + NTH * i, row = q / (H / 4), c4 = q % (H / 4), r = r0 + row;
      unsigned off = (unsigned)(((size_t)(r < B ? r : 0) * ld + 4 * c4) * sizeof(float));
      u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
      v[i] = __builtin_bit_cast(f32x4m, u);   // rows beyond B are zeroed in to_lds: touching the value here would put the vmcnt wait here
    }
  }
  // same tile from memory written by an EARLIER kernel (no hand-off): plain cached 16-byte loads
  __device__ __forceinline__ void load_plain(const float* src, int ld, int r0, int B) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int q = threadIdx.x + NTH * i;
      if (!EXACT && q >= NQ) q = 0;
      int row = q / (H / 4), c4 = q % (H / 4), r = r0 + row;
      v[i] = *reinterpret_cast<const f32x4m*>(src + (size_t)(r < B ? r : 0) * ld + 4 * c4);
    }
  }
  __device__ __forceinline__ void to_lds(float* lds, int r0, int B) const {
    if (r0 + SEQ_ROWS <= B) {   // a full row group (every one but a ragged last): no per-element selects - vector instructions are matrix time lost
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int q = threadIdx.x + NTH * i, row = q / (H / 4), c4 = q % (H / 4);
        if (!EXACT && q >= NQ) continue;
        *reinterpret_cast<f32x4m*>(lds + row * SeqK<H>::LD + 4 * c4) = v[i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int q = threadIdx.x + NTH * i, row = q / (H / 4), c4 = q % (H / 4);
      if (!EXACT && q >= NQ) continue;
      *reinterpret_cast<f32x4m*>(lds + row * SeqK<H>::LD + 4 * c4) = r0 + row < B ? v[i] : f32x4m{0, 0, 0, 0};
    }
  }
  // the same with columns >= ncol forced to zero (padded input rows whose padding is not ours to trust)
  __device__ __forceinline__ void to_lds_cols(float* lds, int ldl, int r0, int B, int ncol) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int q = threadIdx.x + NTH * i, row = q / (H / 4), c4 = q % (H / 4);
      if (!EXACT && q >= NQ) continue;
      f32x4m x = v[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) if (r0 + row >= B || 4 * c4 + j >= ncol) x[j] = 0.0f;
      *reinterpret_cast<f32x4m*>(lds + row * ldl + 4 * c4) = x;
    }
  }
};

// UW = 1: 4 wavefronts own 16 hidden units; UW = 2: 8 wavefronts own 32 units (two 16-unit halves that share the staged h tile and
// advance in lock step on the same SIMDs: their matrix phases queue behind each other deterministically instead of colliding at
// random with another workgroup's, and the row group has half as many partners to wait for)
// FUSE: the layer's input projection x_t W_ih^T + b (no dependence on the recurrence) is computed inside the step instead of by a
// GEMM launch in front of it: its MFMAs are split in two halves placed before the flag poll and behind the issue of the h-tile loads,
// where the matrix pipe otherwise idles through the hand-off latency; W_ih's slice sits in a second set of 64 B-operand registers.
// KX: width of the fused input (H for a hidden layer; the padded observation row when the actor's layer-0 gates come straight from the
// observations through the folded weight W_ih0 W_in - 17 k-steps instead of a GEMM launch and a 210 MB round trip of G)
template <int H, int UW, bool FUSE = false, int KX = H>
__global__ __launch_bounds__(256 * UW) void lstm_seq_fwd_kernel(SeqFwdArgs a) {
  constexpr int NTH = 256 * UW, UNITS = SEQ_UNITS * UW;
  typedef SeqK<H> KH;
  typedef SeqK<KX> KXK;
  constexpr int LDH = KH::LD, KXS = KXK::STEPS, LDX = KXK::LD;   // row strides = 8 (mod 16) words: conflict-free ds_read_b128 fragments (SeqK)
  constexpr int NUG = H / UNITS;
#ifndef KBJ_SEQ_XSPLIT_NUM
#define KBJ_SEQ_XSPLIT_NUM 4   // eighths of the input projection's k-steps issued before the flag poll (the rest hides the tile fetch)
#endif
  constexpr int XSPLIT = KXK::NB * KBJ_SEQ_XSPLIT_NUM / 8;   // in 16-k blocks
  __shared__ __attribute__((aligned(16))) float hs[SEQ_ROWS * LDH];
  __shared__ __attribute__((aligned(16))) float xs[FUSE ? SEQ_ROWS * LDX : 4];
  __shared__ float gbuf[4][SEQ_ROWS][UNITS + 4];   // row stride = 4 (mod 8): the accumulator rows of lanes 0-15 / 16-31 (4 rows apart) fall on disjoint banks
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, gate = (tid >> 6) & 3, uh = tid >> 8;
  // XCD-aware mapping (speed only, the protocol is placement independent): workgroups b, b+8, b+16, ... share an XCD,
  // so give each XCD whole row groups and the tile hand-off stays inside one L2
  const int nblk = gridDim.x;
  const int lid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ug = lid % NUG, rg = lid / NUG;
  const int r0 = rg * SEQ_ROWS, u0 = ug * UNITS;
  const int B = a.B, T = a.T;
  if (seq_aborted(a.err, &flag)) return;
  // W_hh rows of this wave's gate for the 16 units, as B operands: B[k slot g][col] = Whh[gate H + u0 + col][KH::kidx(step, g)]
  float wreg[H / 4];
  {
    const float* wrow = a.Whh + (size_t)(gate * H + u0 + SEQ_UNITS * uh + (lane & 15)) * H;
#pragma unroll
    for (int s = 0; s < H / 4; ++s) wreg[s] = wrow[KH::kidx(s, lane >> 4)];
  }
  float wxreg[FUSE ? KXS : 1];
  float bias_col = 0.0f;
  const int ldx = a.ldx ? a.ldx : H, kxv = a.kx ? a.kx : KX;
  if (FUSE) {
    const int grow = gate * H + u0 + SEQ_UNITS * uh + (lane & 15);
    const float* wrow = a.Wih + (size_t)grow * (a.ldw ? a.ldw : H);
#pragma unroll
    for (int s = 0; s < KXS; ++s) { const int k = KXK::kidx(s, lane >> 4); wxreg[s] = k < kxv ? wrow[k] : 0.0f; }
    bias_col = a.bias[grow];
  }
  // the two (row, unit) elements of this thread in the cell epilogue and their cell state
  int erow[2], eunit[2];
  float cm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int e = tid + NTH * i;
    erow[i] = e / UNITS; eunit[i] = e % UNITS;
    int r = r0 + erow[i];
    cm[i] = r < B ? a.Cm[(size_t)r * H + u0 + eunit[i]] : 0.0f;
  }
  float ig[2], fg[2], gg[2], og[2], tc[2], hh[2];  // results of the previous step, stored lazily
#ifdef KBJ_SEQ_BSTAMPS_BUILD   // forward stamps (KBJ_SEQ_STAMPS): diagnostics build as well
#define SEQ_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[t * 6 + (k)] = clock64(); } while (0)
#else
#define SEQ_STAMP(k) do { } while (0)
#endif
  // own inputs (input projection pre-activations, keep flags) are fetched ONE STEP AHEAD: their HBM latency hides behind
  // a whole step instead of stalling the cell
  float gxn[2][4], kpn[2];
  // (no selects: a row beyond B or the step behind the last is fetched from a clamped address and never used - rows are independent, every
  // store below is guarded by r < B; kbj_lstm_bwd16.h has the measurement)
  int rcl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int r = r0 + erow[i]; rcl[i] = r < B ? r : 0; }
  auto fetch_inputs = [&](int tt) {
    const int tq = tt < T ? tt : T - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!FUSE) {
        const float* g = a.G + ((size_t)tq * B + rcl[i]) * 4 * H + u0 + eunit[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) gxn[i][k] = g[k * H];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) gxn[i][k] = 0.0f;
      }
      kpn[i] = a.keep[(size_t)tq * B + rcl[i]];
    }
  };
  fetch_inputs(0);
  if (FUSE) {   // x_0 tile
    SeqTile<KX, NTH> xt;
    xt.load_plain(a.X, ldx, r0, B);
    if (KX == H) xt.to_lds(xs, r0, B); else xt.to_lds_cols(xs, LDX, r0, B, kxv);
    __syncthreads();
  }
  for (int t = 0; t < T; ++t) {
    SEQ_STAMP(0);
    float gx[2][4], kp[2];   // this step's own inputs, fetched one step ago (taken over before any load of this step is in flight)
#pragma unroll
    for (int i = 0; i < 2; ++i) { kp[i] = kpn[i]; for (int k = 0; k < 4; ++k) gx[i][k] = gxn[i][k]; }
    f32x4m acc0 = {bias_col, bias_col, bias_col, bias_col}, acc1 = acc0;
    if (FUSE) {   // first half of the input projection: runs while the partners' flags travel
      KXK::template mma<false>(xs, lane, wxreg, acc0, acc1, 0, XSPLIT);
    }
    if (t > 0) { if (!seq_wait(a.counters + rg * NUG, NUG, (unsigned)t, a.err, &flag, a.timeout_ticks)) return; }
    SEQ_STAMP(1);
    SeqTile<H, NTH> tile;
    tile.load(a.Hm + (size_t)t * B * H, H, r0, B);
    if (FUSE) {   // second half of the input projection: hides the h-tile fetch (the scheduling fences keep the compiler from hoisting the
                  // tile's LDS stores - and the vmcnt wait in front of them - above these MFMAs)
      __builtin_amdgcn_sched_barrier(0);
      KXK::template mma<true>(xs, lane, wxreg, acc0, acc1, XSPLIT, KXK::NB);
      __builtin_amdgcn_sched_barrier(0);
    }
    tile.to_lds(hs, r0, B);
    // AFTER the tile has gone to LDS (vector-memory operations retire in order: anything issued between the tile loads and their wait
    // would be waited for as well): the next step's own inputs, then the previous step's BPTT stash - both have a whole step to land
    fetch_inputs(t + 1);
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int r = r0 + erow[i];
        if (r >= B) continue;
        size_t o1 = ((size_t)(t - 1) * B + r) * H + u0 + eunit[i];
        float* g = a.G + ((size_t)(t - 1) * B + r) * 4 * H + u0 + eunit[i];
        g[0] = ig[i]; g[H] = fg[i]; g[2 * H] = gg[i]; g[3 * H] = og[i];
        a.Hout[o1] = hh[i]; a.TanhC[o1] = tc[i];
        a.Cm[o1 + (size_t)B * H] = cm[i];
      }
    }
    SeqTile<KX, NTH> xt;
    if (FUSE && t + 1 < T) xt.load_plain(a.X + (size_t)(t + 1) * B * ldx, ldx, r0, B);   // next step's input tile rides behind the recurrent MFMAs
    __syncthreads();
    SEQ_STAMP(2);
    KH::template mma<false>(hs, lane, wreg, acc0, acc1, 0, KH::NB);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gbuf[gate][(lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc0[r];
      gbuf[gate][16 + (lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc1[r];
    }
    if (FUSE && t + 1 < T) { if (KX == H) xt.to_lds(xs, r0, B); else xt.to_lds_cols(xs, LDX, r0, B, kxv); }   // every wave is past its reads of xs (they precede this step's barrier above)
    __syncthreads();
    SEQ_STAMP(3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = erow[i], u = eunit[i];
      ig[i] = seq_sigmoid(gbuf[0][row][u] + gx[i][0]); fg[i] = seq_sigmoid(gbuf[1][row][u] + gx[i][1]);
      gg[i] = seq_tanh(gbuf[2][row][u] + gx[i][2]); og[i] = seq_sigmoid(gbuf[3][row][u] + gx[i][3]);
      float c = fg[i] * cm[i] + ig[i] * gg[i];
      tc[i] = seq_tanh(c); hh[i] = og[i] * tc[i];
      cm[i] = c * kp[i];
      if (r0 + row < B) seq_store(a.Hm + ((size_t)(t + 1) * B + r0 + row) * H + u0 + u, hh[i] * kp[i]);  // the hand-off payload
    }
    SEQ_STAMP(4);
    seq_publish(a.counters + rg * NUG + ug, (unsigned)(t + 1));
    SEQ_STAMP(5);
  }
  // BPTT stash of the last step
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = r0 + erow[i];
    if (r >= B) continue;
    size_t o1 = ((size_t)(T - 1) * B + r) * H + u0 + eunit[i];
    float* g = a.G + ((size_t)(T - 1) * B + r) * 4 * H + u0 + eunit[i];
    g[0] = ig[i]; g[H] = fg[i]; g[2 * H] = gg[i]; g[3 * H] = og[i];
    a.Hout[o1] = hh[i]; a.TanhC[o1] = tc[i];
    a.Cm[o1 + (size_t)B * H] = cm[i];
  }
}

template <int H, int UW>
__device__ __forceinline__ void lstm_seq_bwd_body(const SeqBwdArgs a) {
  constexpr int NTH = 256 * UW, UNITS = SEQ_UNITS * UW;
  typedef SeqK<H> KH;
  constexpr int LDH = KH::LD;
  constexpr int NUG = H / UNITS;
  // one gate chunk of dG_{t+1} at a time: a single 34 KB buffer (plus pbuf) keeps the workgroup at 52 KB of LDS so that two
  // recurrences (actor + critic) AND two GEMM workgroups fit on a CU together (profiles/r01: double buffering starved the GEMMs)
  __shared__ __attribute__((aligned(16))) float ds[SEQ_ROWS * LDH];
  __shared__ float pbuf[4][SEQ_ROWS][UNITS + 4];                     // per-wave partial sums of dh (row stride = 4 mod 8, as gbuf above)
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, uh = tid >> 8;   // wave: k quarter, uh: 16-unit half
  // XCD-aware mapping (speed only, the protocol is placement independent): workgroups b, b+8, b+16, ... share an XCD,
  // so give each XCD whole row groups and the tile hand-off stays inside one L2
  const int nblk = gridDim.x;
  const int lid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ug = lid % NUG, rg = lid / NUG;
  const int r0 = rg * SEQ_ROWS, u0 = ug * UNITS;
  const int B = a.B, T = a.T;
  if (seq_aborted(a.err, &flag)) return;
  if (SEQ_BSTAMP_ON && a.stamps && tid == 0 && blockIdx.x < 256) a.stamps[T * 10 + 3 * blockIdx.x] = wall_clock64();   // per-workgroup entry / loop start / exit
  constexpr int KW = H / 4;        // k range of one wave inside a gate chunk
  constexpr int KS = KW / 4;       // k-steps per (wave, gate chunk)
  typedef SeqK<KW> KWK;            // k order inside the wave's range (KW is a multiple of 16 for every served H)
  static_assert(KW % 16 == 0, "the backward recurrence fetches its fragments in 16-k blocks");
  // B operands: for gate chunk c and k-step s: B[k slot g][col] = Whh[c H + wave KW + KWK::kidx(s, g)][u0 + col]
  float wreg[4][KS];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[c][s] = a.Whh[(size_t)(c * H + wave * KW + KWK::kidx(s, lane >> 4)) * H + u0 + SEQ_UNITS * uh + (lane & 15)];
  int erow[2], eunit[2];
  float dcm[2] = {0.0f, 0.0f};
  float bsum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // running column sums of dG (bias gradient)
#pragma unroll
  for (int i = 0; i < 2; ++i) { int e = tid + NTH * i; erow[i] = e / UNITS; eunit[i] = e % UNITS; }
  // everything the cell derivative of a step needs (produced by earlier kernels) is fetched ONE STEP AHEAD
  float actn[2][4], tcn[2], cprevn[2], dhan[2], kpn[2];
  auto fetch_inputs = [&](int tt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = r0 + erow[i];
      bool ok = r < B && tt >= 0;
      size_t o1 = ((size_t)(ok ? tt : 0) * B + (ok ? r : 0)) * H + u0 + eunit[i];
      const float* g = a.Gact + ((size_t)(ok ? tt : 0) * B + (ok ? r : 0)) * 4 * H + u0 + eunit[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) actn[i][k] = ok ? g[k * H] : 0.0f;
      tcn[i] = ok ? a.TanhC[o1] : 0.0f;
      cprevn[i] = ok ? a.Cm[o1] : 0.0f;
      dhan[i] = ok ? a.dHabove[o1] : 0.0f;
      kpn[i] = ok ? a.keep[(size_t)tt * B + r] : 0.0f;
    }
  };
  fetch_inputs(T - 1);
  if (SEQ_BSTAMP_ON && a.stamps && tid == 0 && blockIdx.x < 256) a.stamps[T * 10 + 3 * blockIdx.x + 1] = wall_clock64();
  for (int t = T - 1; t >= 0; --t) {
    const bool last = t == T - 1;
    SEQ_BSTAMP(0);
    if (SEQ_BSTAMP_ON && a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[t * 10 + 9] = wall_clock64();   // constant-rate counter: gives the shader clock the stamps tick at
    float dhm[2] = {0.0f, 0.0f};
    float act[2][4], tc[2], cprev[2], dha[2], kp[2];
    auto prefetch = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) { tc[i] = tcn[i]; cprev[i] = cprevn[i]; dha[i] = dhan[i]; kp[i] = kpn[i]; for (int k = 0; k < 4; ++k) act[i][k] = actn[i][k]; }
      fetch_inputs(t - 1);
    };
    if (last) prefetch();
    else {
      if (!seq_wait(a.counters + rg * NUG, NUG, (unsigned)(T - 1 - t), a.err, &flag, a.timeout_ticks)) return;
      SEQ_BSTAMP(1);
      f32x4m acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      const float* src = a.dG + (size_t)(t + 1) * B * 4 * H;
      SeqTile<H, NTH> tile;
      tile.load(src, 4 * H, r0, B);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float* buf = ds;
        if (c) __syncthreads();                                   // the previous chunk's fragment reads are done
        tile.to_lds(buf, r0, B);
        if (c < 3) tile.load(src + (c + 1) * H, 4 * H, r0, B);   // next chunk in flight during this chunk's MFMAs
        else prefetch();   // own inputs of the next step: issued behind the LAST payload wait (vector-memory operations retire in order, so
                           // anything issued between a chunk's loads and its wait is waited for too); they land during the cell and the hand-off
        __syncthreads();
        SEQ_BSTAMP(2 + c);
        const float* a0p = buf + (lane & 15) * LDH + wave * KW + 4 * (lane >> 4);
        const float* a1p = a0p + 16 * LDH;
#pragma unroll
        for (int j = 0; j < KW / 16; ++j) {
          const f32x4m f0 = *reinterpret_cast<const f32x4m*>(a0p + 16 * j), f1 = *reinterpret_cast<const f32x4m*>(a1p + 16 * j);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0[q], wreg[c][4 * j + q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[q], wreg[c][4 * j + q], acc1, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pbuf[wave][(lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc0[r];
        pbuf[wave][16 + (lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc1[r];
      }
      __syncthreads();
      SEQ_BSTAMP(6);
#pragma unroll
      for (int i = 0; i < 2; ++i) dhm[i] = pbuf[0][erow[i]][eunit[i]] + pbuf[1][erow[i]][eunit[i]] + pbuf[2][erow[i]][eunit[i]] + pbuf[3][erow[i]][eunit[i]];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = r0 + erow[i];
      if (r >= B) continue;
      float ig = act[i][0], fg = act[i][1], gg = act[i][2], og = act[i][3];
      float dh = dha[i] + kp[i] * dhm[i];
      float dc = kp[i] * dcm[i] + dh * og * (1 - tc[i] * tc[i]);
      float* dg = a.dG + ((size_t)t * B + r) * 4 * H + u0 + eunit[i];
      float d0 = dc * gg * ig * (1 - ig), d1 = dc * cprev[i] * fg * (1 - fg), d2 = dc * ig * (1 - gg * gg), d3 = dh * tc[i] * og * (1 - og);
      seq_store(dg, d0);            // dG is the hand-off payload (and the input of the batched dW GEMMs)
      seq_store(dg + H, d1);
      seq_store(dg + 2 * H, d2);
      seq_store(dg + 3 * H, d3);
      bsum[i][0] += d0; bsum[i][1] += d1; bsum[i][2] += d2; bsum[i][3] += d3;
      dcm[i] = dc * fg;
    }
    SEQ_BSTAMP(7);
    seq_publish(a.counters + rg * NUG + ug, (unsigned)(T - t));
    SEQ_BSTAMP(8);
  }
  // bias gradient = column sums of dG over all rows and steps: reduce this workgroup's 32 rows in LDS, one atomic per column
  if (a.db || a.db_part) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pbuf[k][erow[0]][eunit[0]] = bsum[0][k];
      pbuf[k][erow[1]][eunit[1]] = bsum[1][k];
    }
    __syncthreads();
    if (tid < 4 * UNITS) {
      int k = tid / UNITS, u = tid % UNITS;
      float s = 0;
      for (int r = 0; r < SEQ_ROWS; ++r) s += pbuf[k][r][u];
      if (a.db_part) a.db_part[(size_t)rg * 4 * H + k * H + u0 + u] = s;
      else atomicAdd(a.db + k * H + u0 + u, s);
    }
  }
  if (SEQ_BSTAMP_ON && a.stamps && tid == 0 && blockIdx.x < 256) a.stamps[T * 10 + 3 * blockIdx.x + 2] = wall_clock64();
}
// hidden sizes up to 256: the capped register budget above. Wider layers (the weight slice alone is H / 4 registers per gate chunk pair)
// run one recurrence at a time and without GEMM workgroups beside them (kbj_nn.hip: one_stream), so their kernel takes what it needs.
template <int H, int UW>
__global__ __launch_bounds__(256 * UW) __attribute__((amdgpu_num_vgpr(KBJ_SEQ_BWD_NUM_VGPR))) void lstm_seq_bwd_kernel(SeqBwdArgs a) { lstm_seq_bwd_body<H, UW>(a); }
template <int H, int UW>
__global__ __launch_bounds__(256 * UW) void lstm_seq_bwd_wide_kernel(SeqBwdArgs a) { lstm_seq_bwd_body<H, UW>(a); }

// ---- one LSTM layer step for MANY independent rows (rollout: 8192 envs, no recurrence inside the launch) ------------------------------
// Same register-resident weight slices and fused cell as the forward recurrence above, but the loop runs over row groups instead of
// time: workgroup (chunk, ug) keeps the 128 gate columns of its 32 hidden units as MFMA B operands (W_ih and W_hh slices, 128 VGPRs)
// and walks the row groups chunk, chunk + nchunk, ...: stage the x and h tiles of 32 rows through LDS, 2 x (KX + H) / 4 MFMAs per
// wavefront, exchange the four gates through LDS, apply the cell, store h and c. One launch replaces the [x | h] GEMM, its 33 MB gate
// round trip and the cell kernel of a rollout layer; the weights are read once per workgroup instead of once per output tile.
// Hout must not alias Hin (the other unit groups of a row group still read the full h rows); C is updated in place.
struct StepArgs {
  const float* X; int ldx, kx;   // layer input [M][ldx]; columns >= kx count as zero (kx = 0: all KX)
  const float* Wih; int ldw;     // [4H][ldw]
  const float* Whh;              // [4H][H]
  const float* bias;             // [4H]
  const float* Hin;              // [M][H]
  float* Hout;                   // [M][H]
  float* C;                      // [M][H], in place
  int M;
};
template <int H, int UW, int KX = H>
__global__ __launch_bounds__(256 * UW) void lstm_step_kernel(StepArgs a) {
  constexpr int NTH = 256 * UW, UNITS = SEQ_UNITS * UW;
  typedef SeqK<H> KH;
  typedef SeqK<KX> KXK;
  constexpr int LDH = KH::LD, KXS = KXK::STEPS, LDX = KXK::LD, NUG = H / UNITS;
  __shared__ __attribute__((aligned(16))) float hs[SEQ_ROWS * LDH];
  __shared__ __attribute__((aligned(16))) float xs[SEQ_ROWS * LDX];
  __shared__ float gbuf[4][SEQ_ROWS][UNITS + 4];
  const int tid = threadIdx.x, lane = tid & 63, gate = (tid >> 6) & 3, uh = tid >> 8;
  const int ug = blockIdx.x % NUG, chunk = blockIdx.x / NUG, nchunk = gridDim.x / NUG;
  const int u0 = ug * UNITS, M = a.M, nrg = (M + SEQ_ROWS - 1) / SEQ_ROWS;
  const int kxv = a.kx ? a.kx : KX;
  const int grow = gate * H + u0 + SEQ_UNITS * uh + (lane & 15);
  float wreg[H / 4], wxreg[KXS];
  {
    // registers 4j..4j+3 of a lane are 4 CONSECUTIVE k (SeqK::kidx): 16-byte loads, 16 half cache lines per instruction instead of 64
    // per 4-byte load - this kernel reads its weight slices once per launch, not once per 100 steps like the recurrences
    const float* wrow = a.Whh + (size_t)grow * H + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < KH::NB; ++j) {
      const f32x4m q = *reinterpret_cast<const f32x4m*>(wrow + 16 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) wreg[4 * j + e] = q[e];
    }
    const float* xrow = a.Wih + (size_t)grow * a.ldw;
    if (a.kx == 0 && (a.ldw & 3) == 0) {
#pragma unroll
      for (int j = 0; j < KXK::NB; ++j) {
        const f32x4m q = *reinterpret_cast<const f32x4m*>(xrow + 4 * (lane >> 4) + 16 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) wxreg[4 * j + e] = q[e];
      }
#pragma unroll
      for (int s = 4 * KXK::NB; s < KXS; ++s) wxreg[s] = xrow[KXK::kidx(s, lane >> 4)];
    } else {
#pragma unroll
      for (int s = 0; s < KXS; ++s) { const int k = KXK::kidx(s, lane >> 4); wxreg[s] = k < kxv ? xrow[k] : 0.0f; }
    }
  }
  const float bias_col = a.bias[grow];
  int erow[2], eunit[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { int e = tid + NTH * i; erow[i] = e / UNITS; eunit[i] = e % UNITS; }
  SeqTile<KX, NTH> xt;
  SeqTile<H, NTH> ht;
  float cn[2] = {0.0f, 0.0f};
  auto fetch = [&](int rg) {   // tiles and cell state of row group rg into registers (they land behind the previous group's MFMAs)
    const int r0 = rg * SEQ_ROWS;
    xt.load_plain(a.X, a.ldx, r0, M);
    ht.load_plain(a.Hin, H, r0, M);
#pragma unroll
    for (int i = 0; i < 2; ++i) { int r = r0 + erow[i]; cn[i] = r < M ? a.C[(size_t)r * H + u0 + eunit[i]] : 0.0f; }
  };
  int rg = chunk;
  if (rg < nrg) fetch(rg);
  for (; rg < nrg; rg += nchunk) {
    const int r0 = rg * SEQ_ROWS;
    if (KX == H) xt.to_lds(xs, r0, M); else xt.to_lds_cols(xs, LDX, r0, M, kxv);
    ht.to_lds(hs, r0, M);
    const float cp[2] = {cn[0], cn[1]};
    __syncthreads();
    if (rg + nchunk < nrg) fetch(rg + nchunk);
    f32x4m acc0 = {bias_col, bias_col, bias_col, bias_col}, acc1 = acc0;
    KXK::template mma<true>(xs, lane, wxreg, acc0, acc1, 0, KXK::NB);
    KH::template mma<false>(hs, lane, wreg, acc0, acc1, 0, KH::NB);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gbuf[gate][(lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc0[r];
      gbuf[gate][16 + (lane >> 4) * 4 + r][SEQ_UNITS * uh + (lane & 15)] = acc1[r];
    }
    __syncthreads();   // every wavefront is past its fragment reads of xs / hs: the next group's tiles may overwrite them
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = erow[i], u = eunit[i], r = r0 + row;
      float ig = seq_sigmoid(gbuf[0][row][u]), fg = seq_sigmoid(gbuf[1][row][u]), gg = seq_tanh(gbuf[2][row][u]), og = seq_sigmoid(gbuf[3][row][u]);
      float c = fg * cp[i] + ig * gg;
      if (r < M) { a.Hout[(size_t)r * H + u0 + u] = og * seq_tanh(c); a.C[(size_t)r * H + u0 + u] = c; }
    }
    // gbuf is rewritten only behind the next group's first barrier, which every wavefront reaches after these reads
  }
}

}  // namespace kbj
