// kbj_nn_kernels.h — element-wise / scan kernels of the LSTM actor-critic and the PPO update (device code).
// GEMMs live in kbj_gemm.h. Reference semantics: Actor/Critic forward train.py:913-941, 993-1004; equinox
// LSTMCell (gate order i,f,g,o, single bias); distrax diagonal Gaussian; carry reset on done train.py:1502-1506.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kbj_model.h"

namespace kbj {

constexpr float kLog2Pi = 1.8378770664093453f;

// gate non-linearities on the hardware exp/rcp units (same definitions as kbj_lstm_seq.h so rollout and update agree)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }      // same forms as kbj_lstm_seq.h seq_sigmoid / seq_tanh
__device__ __forceinline__ float tanhf_(float x) { float xc = fminf(fmaxf(x, -15.0f), 15.0f); return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * xc)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// ---- LSTM cell, forward ----------------------------------------------------------------------------------------
// G [M][4H] pre-activations (bias included) -> overwritten with the gate activations (i,f,g,o);
// c_prev [M][H]; writes h_out, c_out (may alias c_prev) and, when given, the masked copies consumed by the next
// time step (h*keep, c*keep with keep[m] = 1 - done) and tanh(c) for the backward pass.
struct CellFwdArgs {
  float* G; const float* c_prev; float* h_out; float* c_out; float* hm_next; float* cm_next; float* tanhc; const float* keep;
  int M, H;
};
struct CellFwdArgs2 { CellFwdArgs a[2]; };
__global__ void lstm_cell_fwd_kernel(CellFwdArgs2 args) {
  const CellFwdArgs& a = args.a[blockIdx.y];
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.M * a.H) return;
  int m = idx / a.H, u = idx - m * a.H;
  float* g = a.G + (size_t)m * 4 * a.H;
  float i = sigmoidf_(g[u]), f = sigmoidf_(g[a.H + u]), gg = tanhf_(g[2 * a.H + u]), o = sigmoidf_(g[3 * a.H + u]);
  float c = f * a.c_prev[idx] + i * gg;
  float tc = tanhf_(c), h = o * tc;
  g[u] = i; g[a.H + u] = f; g[2 * a.H + u] = gg; g[3 * a.H + u] = o;
  a.h_out[idx] = h;
  a.c_out[idx] = c;
  if (a.hm_next) { float k = a.keep[m]; a.hm_next[idx] = h * k; a.cm_next[idx] = c * k; a.tanhc[idx] = tc; }
}

// ---- threefry (same generator as the env kernels; RNG stream KBJ_RNG_ACTION / KBJ_RNG_INIT) ---------------------
__device__ __forceinline__ uint32_t nn_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void nn_threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const int rot[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = nn_rotl32(x1, rot[(g & 1) * 4 + r]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  o0 = x0; o1 = x1;
}

// ---- actor head at rollout time (train.py:924-941, 1545-1572): 32 lanes per env, one per joint (20 active) --------------
struct HeadParams { float min_std, max_std, var_scale, alpha; int ld_obs; };   // ld_obs: stride of the actor observation rows (68 + user columns)
// The rollout's actor head as ONE launch on the env -> actor -> env chain: output projection (H -> 40), low-pass, Gaussian sample and
// log-prob. (As a 64x64-tile GEMM the 40-column projection was 45 us of mostly latency at 8192 envs, the sampling kernel 22 us more.)
// It runs beside the critic's GEMMs of the same step, which hold most of every CU's LDS and registers, so it is built to fit in the
// gaps: one wavefront per 16 envs, the projection as 3 x H / 4 v_mfma_f32_16x16x4_f32 with both operands fetched straight from global
// memory in fragment layout (16-byte loads: 4 consecutive k per lane; W_out is 40 KB and stays in L2 / L1), ~5 KB of LDS per
// workgroup for the hand-over to the per-joint stage.
constexpr int HEAD_ENVS = 16;    // envs per wavefront (one MFMA row block)
constexpr int HEAD_WAVES = 4;    // wavefronts per workgroup
typedef float head_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64 * HEAD_WAVES) void actor_head_fused_kernel(const float* __restrict__ hin /*[N][H]*/, int H, const float* __restrict__ Wout /*[40][H]*/,
                                                                          const float* __restrict__ bout, const float* __restrict__ obs /*[N][68]*/,
                                                                          float* __restrict__ lpf, const float* __restrict__ joint_bias, HeadParams hp, uint32_t seed,
                                                                          uint32_t env_off, uint32_t step, int argmax, int N, float* __restrict__ action,
                                                                          float* __restrict__ logp) {
  __shared__ float outs[HEAD_WAVES][HEAD_ENVS][49];          // [env][output], 48 columns used (40 real)
  __shared__ float lps[HEAD_WAVES][HEAD_ENVS][KBJ_NU + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int n0 = (blockIdx.x * HEAD_WAVES + wv) * HEAD_ENVS;
  if (n0 < N) {   // wavefront-uniform
    const float* arow = hin + (size_t)min(n0 + c, N - 1) * H + 4 * g;
    const float* brow[3] = {Wout + (size_t)c * H + 4 * g, Wout + (size_t)(16 + c) * H + 4 * g, Wout + (size_t)min(32 + c, 2 * KBJ_NU - 1) * H + 4 * g};
    head_f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int k = 0; k < H; k += 16) {   // lane (g, c) feeds k + 4 g + q to MFMA q of the block: the same k for A and B
      const head_f32x4 a = *reinterpret_cast<const head_f32x4*>(arow + k);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const head_f32x4 b = *reinterpret_cast<const head_f32x4*>(brow[t] + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) outs[wv][4 * g + r][16 * t + c] = acc[t][r];   // C[row = 4 g + r][col = c]
  }
  __syncthreads();
  if (n0 < N) {
    for (int item = lane; item < HEAD_ENVS * KBJ_NU; item += 64) {   // one (env, joint) per lane and pass
      const int e = item / KBJ_NU, j = item - e * KBJ_NU, n = n0 + e;
      float lp = 0;
      if (n < N) {
        const float om = outs[wv][e][j] + bout[j], os = outs[wv][e][KBJ_NU + j] + bout[KBJ_NU + j];
        float mean = om + joint_bias[j] + (j >= 10 ? obs[(size_t)n * hp.ld_obs + KBJ_OBS_CMD + 6 + (j - 10)] : 0.0f);
        float sd = fminf((softplusf_(os) + hp.min_std) * hp.var_scale, hp.max_std);
        float y0 = lpf[n * KBJ_NU + j];
        float y = y0 + hp.alpha * (mean - y0);
        lpf[n * KBJ_NU + j] = y;
        float a = y;
        if (!argmax) {
          uint32_t b0, b1;
          nn_threefry(seed ^ ((uint32_t)KBJ_RNG_ACTION * 0x9E3779B9u), env_off + (uint32_t)n, step, (uint32_t)j, b0, b1);
          float u1 = (float)((b0 >> 8) + 1u) * (1.0f / 16777216.0f), u2 = (float)(b1 >> 8) * (1.0f / 16777216.0f);
          a = y + sd * (sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2));
        }
        action[n * KBJ_NU + j] = a;
        float z = (a - y) / sd;
        lp = -0.5f * z * z - logf(sd) - 0.5f * kLog2Pi;
      }
      lps[wv][e][j] = lp;
    }
  }
  __syncthreads();
  if (n0 < N && lane < HEAD_ENVS && n0 + lane < N) {
    float lp = 0;
#pragma unroll
    for (int j = 0; j < KBJ_NU; ++j) lp += lps[wv][lane][j];   // fixed order: the log-prob of an env does not depend on the launch shape
    logp[n0 + lane] = lp;
  }
}

// the rollout's critic head as one launch: value[n] = h[n] . w_out + b_out (one output; 32 lanes per env)
__global__ __launch_bounds__(256) void critic_value_fused_kernel(const float* __restrict__ hin /*[N][H]*/, int H, const float* __restrict__ wout /*[H]*/,
                                                                 const float* __restrict__ bout, int N, float* __restrict__ value) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, n = tid >> 5, j = tid & 31;
  float acc = 0;
  if (n < N)
    for (int c = j; c < H / 4; c += 32) {
      const float4 h = *reinterpret_cast<const float4*>(hin + (size_t)n * H + 4 * c), w = *reinterpret_cast<const float4*>(wout + 4 * c);
      acc = fmaf(h.x, w.x, acc); acc = fmaf(h.y, w.y, acc); acc = fmaf(h.z, w.z, acc); acc = fmaf(h.w, w.w, acc);
    }
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (n < N && j == 0) value[n] = acc + bout[0];
}

// value_d[n] = out[n][0]
__global__ void critic_value_kernel(const float* __restrict__ out, int ld, int N, float* __restrict__ value) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) value[n] = out[(size_t)n * ld];
}

// carry <- carry * (done == 0) (train.py:1502-1506): 2 x depth [cnt][H] planes (h0, c0, h1, c1, ... - the h planes may live in the
// rollout's ping-pong scratch), lpf [cnt][20]
struct CarryPlanes { float* p[2 * KBJ_MAX_DEPTH]; int n; };
// one wavefront per env row: a single flag load decides, and only the (few) finished envs touch their planes
__global__ void carry_reset_kernel(CarryPlanes hc, int cnt, int H, float* __restrict__ lpf, const float* __restrict__ done, int stride) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= cnt || done[(size_t)row * stride] == 0.0f) return;
  for (int p = 0; p < hc.n; ++p)
    for (int k = lane; k < H; k += 64) hc.p[p][(size_t)row * H + k] = 0.0f;
  if (lpf && lane < KBJ_NU) lpf[(size_t)row * KBJ_NU + lane] = 0.0f;
}

// ---- minibatch gathers ---------------------------------------------------------------------------------------------
// dst[t][b][0:w] = src[t][idx[b]][0:w]
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int T, int N, int B, int w, int ld_src, int ld_dst, float* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)T * B * w;
  if (i >= total) return;
  int k = i % w;
  size_t r = i / w;
  int b = r % B, t = r / B;
  dst[r * ld_dst + k] = src[((size_t)t * N + idx[b]) * ld_src + k];
}
// the same with 16-byte accesses (w, ld_src, ld_dst multiples of 4 and 16-byte aligned bases: the padded observation rows): one thread
// moves a float4, a row of the critic observations is 119 consecutive lanes
__global__ void gather_rows4_kernel(const float4* __restrict__ src, const int* __restrict__ idx, int T, int N, int B, int w4, int ld_src4, int ld_dst4, float4* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)T * B * w4;
  if (i >= total) return;
  int k = i % w4;
  size_t r = i / w4;
  int b = r % B, t = r / B;
  dst[r * ld_dst4 + k] = src[((size_t)t * N + idx[b]) * ld_src4 + k];
}
// keep[t][b] = 1 - (aux[t][idx[b]][DONE] != 0)
__global__ void gather_keep_kernel(const float* __restrict__ aux, const int* __restrict__ idx, int T, int N, int B, float* __restrict__ keep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * B) return;
  int b = i % B, t = i / B;
  keep[i] = aux[((size_t)t * N + idx[b]) * KBJ_AUX_SIZE + KBJ_AUX_DONE] != 0 ? 0.0f : 1.0f;
}

// all per-sample scalars of the minibatch in one launch: action (20 columns), old log-prob, old value, advantage, target, keep
struct GatherSmallArgs {
  const float *action, *logp, *value, *adv, *target, *aux;
  float *action_o, *logp_o, *value_o, *adv_o, *target_o, *keep_o;
};
// columns [c0, c1) of the 25 per sample: 0..19 action, 20 old log-prob, 21 old value, 22 advantage, 23 target, 24 keep. The forward recurrences
// need the keep flags only (one launch of column 24 on their lane); the rest is gathered on a side lane, off the chain that starts them.
__global__ void gather_small_kernel(GatherSmallArgs a, const int* __restrict__ idx, int T, int N, int B, int c0, int c1) {
  const int W = c1 - c0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)T * B * W) return;
  size_t r = i / W;
  int c = c0 + (int)(i - r * W), b = (int)(r % B);
  size_t src = (r / B) * N + idx[b];
  if (c < KBJ_NU) a.action_o[r * KBJ_NU + c] = a.action[src * KBJ_NU + c];
  else if (c == KBJ_NU) { if (a.logp) a.logp_o[r] = a.logp[src]; }          // the forward-only pass (kbj_ppo_forward) gathers actions and keep flags only
  else if (c == KBJ_NU + 1) { if (a.value) a.value_o[r] = a.value[src]; }
  else if (c == KBJ_NU + 2) { if (a.adv) a.adv_o[r] = a.adv[src]; }
  else if (c == KBJ_NU + 3) { if (a.target) a.target_o[r] = a.target[src]; }
  else a.keep_o[r] = a.aux[src * KBJ_AUX_SIZE + KBJ_AUX_DONE] != 0 ? 0.0f : 1.0f;
}
// the carries at the start of the trajectory: up to 16 [N][H] planes and 2 [N][20] low-pass states, one launch (blockIdx.y = plane)
struct GatherCarryArgs { const float* src[8 * KBJ_MAX_DEPTH + 2]; float* dst[8 * KBJ_MAX_DEPTH + 2]; int nplanes, nlpf; };   // 4 nets x depth x (h, c) + 2 low-pass states
__global__ void gather_carry_kernel(GatherCarryArgs a, const int* __restrict__ idx, int B, int H) {
  int p = blockIdx.y;
  int w = p < a.nplanes ? H : KBJ_NU;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * w) return;
  int b = i / w, k = i - b * w;
  a.dst[p][i] = a.src[p][(size_t)idx[b] * w + k];
}

// ---- actor head over a minibatch trajectory: per (b, j) thread scans time (low-pass filter recursion) --------------
// out [T][B][40], obs [T][B][68], act [T][B][20], keep [T][B], lpf0 [B][20] -> y [T][B][20] (filtered mean), std [T][B][20]
// stage 1 (parallel over all T*B*20 elements): unfiltered mean -> y, std -> sd
__global__ void actor_head_pre_kernel(const float* __restrict__ out, const float* __restrict__ obs, const float* __restrict__ joint_bias, HeadParams hp,
                                      int R, float* __restrict__ y, float* __restrict__ sd) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * KBJ_NU) return;
  size_t r = i / KBJ_NU;
  int j = (int)(i - r * KBJ_NU);
  y[i] = out[r * 40 + j] + joint_bias[j] + (j >= 10 ? obs[r * hp.ld_obs + KBJ_OBS_CMD + 6 + (j - 10)] : 0.0f);
  sd[i] = fminf((softplusf_(out[r * 40 + KBJ_NU + j]) + hp.min_std) * hp.var_scale, hp.max_std);
}
// stage 2 (one thread per (b, j), serial in t): one-pole low-pass over the means in place, state reset where done
__global__ void actor_head_train_fwd_kernel(const float* __restrict__ keep, const float* __restrict__ lpf0, HeadParams hp, int T, int B, float* __restrict__ y) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * KBJ_NU) return;
  int b = i / KBJ_NU, j = i % KBJ_NU;
  float state = lpf0[i];
  constexpr int U = 10;   // the recursion is serial in t, its inputs are not: fetch U steps, then run the U dependent updates
  for (int t0 = 0; t0 < T; t0 += U) {
    float mean[U], kp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = t0 + u;
      if (t < T) { size_t r = (size_t)t * B + b; mean[u] = y[r * KBJ_NU + j]; kp[u] = keep[r]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = t0 + u;
      if (t < T) {
        float yy = state + hp.alpha * (mean[u] - state);
        y[((size_t)t * B + b) * KBJ_NU + j] = yy;
        state = yy * kp[u];
      }
    }
  }
}
// logp[r] / entropy[r] from y, sd, act (one thread per (t,b))
__global__ void gaussian_logp_kernel(const float* __restrict__ y, const float* __restrict__ sd, const float* __restrict__ act, int R,
                                     float* __restrict__ logp, float* __restrict__ ent) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float lp = 0, en = 0;
  for (int j = 0; j < KBJ_NU; ++j) {
    float s = sd[(size_t)r * KBJ_NU + j], z = (act[(size_t)r * KBJ_NU + j] - y[(size_t)r * KBJ_NU + j]) / s;
    lp += -0.5f * z * z - logf(s) - 0.5f * kLog2Pi;
    en += 0.5f + 0.5f * kLog2Pi + logf(s);
  }
  logp[r] = lp; ent[r] = en;
}

// ---- PPO loss (restated ksim defaults): statistics pass then per-sample gradient coefficients ------------
// stats[0..1] += sum(adv), sum(adv^2) over the minibatch (double accumulation, one atomic pair per block)
// part != null (deterministic mode): the block's pair goes to part[2 * blockIdx.x ..] and reduce_double_kernel adds the blocks in order
__global__ void adv_stats_kernel(const float* __restrict__ adv, int R, double* __restrict__ stats, double* __restrict__ part) {
  __shared__ double s1[256], s2[256];
  double a = 0, b = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += 256 * gridDim.x) { double v = adv[i]; a += v; b += v * v; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) {
    if (part) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
    else { atomicAdd(&stats[0], s1[0]); atomicAdd(&stats[1], s2[0]); }   // stats zeroed by the caller
  }
}
// out[j] += sum over blocks b (in order) of part[b * w + j], j < w (w <= 64): the fixed-order second stage of the double-precision sums
__global__ void reduce_double_kernel(const double* __restrict__ part, int nblocks, int w, double* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= w) return;
  double s = 0;
  for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * w + j];
  out[j] += s;
}
struct PpoParams { float clip, vclip, vcoef, ecoef, lrclip, adv_eps; };
// per sample: coefficients dL/dlogp, dL/dvalue, dL/dentropy(const) and metric partial sums (atomics into metrics_acc[8] doubles)
__global__ void ppo_loss_kernel(const float* __restrict__ logp, const float* __restrict__ value, const float* __restrict__ ent,
                                const float* __restrict__ logp_old, const float* __restrict__ value_old, const float* __restrict__ adv,
                                const float* __restrict__ target, const double* __restrict__ stats, PpoParams pp, int R,
                                float* __restrict__ dlogp, float* __restrict__ dvalue, double* __restrict__ macc, int part) {
  // part: 1 = policy terms (surrogate, entropy, clip fraction, KL: inputs logp / ent / adv), 2 = value terms (inputs value / target), 3 = both.
  // The two halves share no data, so each runs on its own net's lane and the actor and critic chains never wait for each other.
  __shared__ double red[6][256];
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  double m[6] = {0, 0, 0, 0, 0, 0};
  const float inv = 1.0f / R;
  if (r < R && (part & 1)) {
    // stats[11]: the number of samples behind the two sums when the caller supplied them (kbj_set_advantage_sums: the global minibatch of a
    // data-parallel job), else 0 = this minibatch's own R samples
    const double cnt = stats[11] > 0 ? stats[11] : (double)R;
    float mean = (float)(stats[0] / cnt);
    float var = fmaxf((float)(stats[1] / cnt) - mean * mean, 0.0f);
    float a = (adv[r] - mean) / (sqrtf(var) + pp.adv_eps);
    float d = logp[r] - logp_old[r];
    float dcl = fminf(fmaxf(d, -pp.lrclip), pp.lrclip);
    float ratio = expf(dcl);
    float rc = fminf(fmaxf(ratio, 1 - pp.clip), 1 + pp.clip);
    float s1 = ratio * a, s2 = rc * a;
    bool unclipped = s1 <= s2;
    float surr = unclipped ? s1 : s2;
    dlogp[r] = (unclipped && fabsf(d) < pp.lrclip) ? -inv * a * ratio : 0.0f;
    m[0] = -surr; m[2] = ent[r]; m[3] = fabsf(ratio - 1) > pp.clip ? 1.0 : 0.0; m[4] = -d;
  }
  if (r < R && (part & 2)) {
    float v = value[r], vo = value_old[r], tg = target[r];
    float dv = v - vo, dvc = fminf(fmaxf(dv, -pp.vclip), pp.vclip), vcl = vo + dvc;
    float e1 = (v - tg) * (v - tg), e2 = (vcl - tg) * (vcl - tg);
    float gv = e1 >= e2 ? (v - tg) : ((fabsf(dv) < pp.vclip) ? (vcl - tg) : 0.0f);
    dvalue[r] = pp.vcoef * inv * gv;
    m[1] = 0.5f * fmaxf(e1, e2);
  }
  for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = m[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) for (int k = 0; k < 5; ++k) if (k == 1 ? (part & 2) : (part & 1)) atomicAdd(&macc[k], red[k][0]);
}
// critic head, fused (one output, train.py:993-1004 + the value terms of the PPO loss): value = h w + b -> clipped value loss ->
// dL/dvalue -> dL/dh = dL/dvalue * w, all in one pass over the top layer's output rows (one wavefront per row, VPL = H / 64 floats per
// lane). Replaces a [R x H] x [H x 1] GEMM, the value gather, the value half of ppo_loss_kernel, a strided copy and a K = 1 GEMM on
// the critic's critical chain. dout (leading dimension 40) receives dL/dvalue in column 0 for the output layer's weight / bias gradient.
template <int VPL>
__global__ void critic_head_kernel(const float* __restrict__ h, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                   const float* __restrict__ value_old, const float* __restrict__ target, PpoParams pp, int R,
                                   float* __restrict__ value, float* __restrict__ dvalue, float* __restrict__ dout, float* __restrict__ dh,
                                   double* __restrict__ macc) {
  constexpr int H = 64 * VPL;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float wreg[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) wreg[j] = w_out[lane * VPL + j];
  const float bias = b_out[0], inv = 1.0f / R;
  double m1 = 0;
  for (int r = blockIdx.x * 4 + wv; r < R; r += gridDim.x * 4) {
    float x[VPL], s = 0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { x[j] = h[(size_t)r * H + lane * VPL + j]; s += x[j] * wreg[j]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float v = s + bias, vo = value_old[r], tg = target[r];
    const float dv = v - vo, dvc = fminf(fmaxf(dv, -pp.vclip), pp.vclip), vcl = vo + dvc;
    const float e1 = (v - tg) * (v - tg), e2 = (vcl - tg) * (vcl - tg);
    const float gv = e1 >= e2 ? (v - tg) : ((fabsf(dv) < pp.vclip) ? (vcl - tg) : 0.0f);
    const float dval = pp.vcoef * inv * gv;
#pragma unroll
    for (int j = 0; j < VPL; ++j) dh[(size_t)r * H + lane * VPL + j] = dval * wreg[j];
    if (lane == 0) { value[r] = v; dvalue[r] = dval; dout[(size_t)r * 40] = dval; m1 += 0.5f * fmaxf(e1, e2); }
  }
  if (lane == 0) red[wv] = m1;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&macc[1], red[0] + red[1] + red[2] + red[3]);
}
// metrics[10] = loss, policy, value, entropy, clipfrac, kl, adv_mean, adv_std, action_mirror_loss, value_mirror_loss
__global__ void ppo_metrics_kernel(const double* __restrict__ macc, const double* __restrict__ stats, PpoParams pp, int R, float* __restrict__ metrics) {
  double pol = macc[0] / R, vl = macc[1] / R, en = macc[2] / R, ma = macc[5] / R, mc = macc[6] / R;
  const double cnt = stats[11] > 0 ? stats[11] : (double)R;
  double mean = stats[0] / cnt, var = stats[1] / cnt - mean * mean;
  metrics[0] = (float)(pol + pp.vcoef * vl - pp.ecoef * en + ma + mc);
  metrics[8] = (float)ma; metrics[9] = (float)mc;
  metrics[1] = (float)pol; metrics[2] = (float)vl; metrics[3] = (float)en; metrics[4] = (float)(macc[3] / R); metrics[5] = (float)(macc[4] / R);
  metrics[6] = (float)mean; metrics[7] = (float)sqrt(var > 0 ? var : 0);
}

// ---- mirror aux losses (train.py:1463-1481, 1574-1756) ---------------------------------------------------------------
// The sagittal mirror of a PACKED observation row is a signed permutation plus an affine fix-up of the normalised joint
// positions (source and destination joints have different biases/ranges): out[k] = mul[k] * in[src[k]] + add[k].
struct MirrorEntry { int src; float mul, add; };
__global__ void mirror_rows_kernel(const float* __restrict__ in, float* __restrict__ out, size_t rows, int ld, const MirrorEntry* __restrict__ tab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  size_t r = i / ld;
  int k = (int)(i - r * ld);
  MirrorEntry e = tab[k];
  out[i] = e.mul * in[r * ld + e.src] + e.add;
}
// mirror branch of the actor head at rollout time: only the low-pass state advances (no sampling)
__global__ void actor_head_lpf_kernel(const float* __restrict__ out, const float* __restrict__ obs, float* __restrict__ lpf,
                                      const float* __restrict__ joint_bias, float alpha, int N, int ld_obs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * KBJ_NU) return;
  int n = i / KBJ_NU, j = i % KBJ_NU;
  float mean = out[n * 40 + j] + joint_bias[j] + (j >= 10 ? obs[(size_t)n * ld_obs + KBJ_OBS_CMD + 6 + (j - 10)] : 0.0f);
  float y0 = lpf[i];
  lpf[i] = y0 + alpha * (mean - y0);
}
// per sample: action_mirror_loss = mean_j (y_j - mirror_joints(y_m)_j)^2 * sa, value_mirror_loss = (v - v_m)^2 * sc; both are
// averaged over the minibatch and added to the loss. Writes the direct gradients on y, y_m (dy, dym [R][20]) and adds the
// value terms to dvalue / writes dvalue_m. macc[5], macc[6] accumulate the two loss sums.
__global__ void mirror_loss_kernel(const float* __restrict__ y, const float* __restrict__ ym, const float* __restrict__ v, const float* __restrict__ vm,
                                   float sa, float sc, int R, float* __restrict__ dy, float* __restrict__ dym, float* __restrict__ dvalue,
                                   float* __restrict__ dvalue_m, double* __restrict__ macc, int part) {   // part: 1 = actor term, 2 = critic term (as ppo_loss_kernel)
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float ca = sa / (20.0f * R), la = 0;
  if (part & 1) for (int i = 0; i < KBJ_NU; ++i) {
    int s = i < 5 ? i + 5 : (i < 10 ? i - 5 : i);
    float e = y[(size_t)r * KBJ_NU + i] + ym[(size_t)r * KBJ_NU + s];     // y_i - (-y_m[s(i)])
    la += e * e;
    dy[(size_t)r * KBJ_NU + i] = 2 * ca * e;
    dym[(size_t)r * KBJ_NU + s] = 2 * ca * e;
  }
  if (part & 1) atomicAdd(&macc[5], (double)(sa * la / 20.0f));
  if (part & 2) {
    float ev = v[r] - vm[r];
    float gv = 2 * sc / R * ev;
    dvalue[r] += gv;
    dvalue_m[r] = -gv;
    atomicAdd(&macc[6], (double)(sc * ev * ev));
  }
}

// actor head backward: per (b, j) thread, reverse scan through the low-pass recursion.
// dL/dlogp [T][B] and the constant entropy coefficient (+ an optional direct gradient on y) -> dOut [T][B][40]
// stage 1 (parallel over all T*B*20 elements): gradient wrt the pre-activation std -> dout[.., 20 + j]; direct gradient wrt the
// filtered mean y (log-prob term + optional aux term) -> dout[.., j] (turned into the pre-filter gradient by stage 2)
__global__ void actor_head_bwd_pre_kernel(const float* __restrict__ out, const float* __restrict__ y, const float* __restrict__ sd,
                                          const float* __restrict__ act, const float* __restrict__ dlogp, const float* __restrict__ dy_extra, float dent,
                                          HeadParams hp, int R, float* __restrict__ dout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * KBJ_NU) return;
  size_t r = i / KBJ_NU;
  int j = (int)(i - r * KBJ_NU);
  float s = sd[i], z = (act[i] - y[i]) / s;
  float gl = dlogp[r];
  dout[r * 40 + j] = gl * (z / s) + (dy_extra ? dy_extra[i] : 0.0f);
  float gs = gl * ((z * z - 1.0f) / s) + dent / s;
  float raw = out[r * 40 + KBJ_NU + j];
  float pre = (softplusf_(raw) + hp.min_std) * hp.var_scale;
  dout[r * 40 + KBJ_NU + j] = pre < hp.max_std ? gs * hp.var_scale * sigmoidf_(raw) : 0.0f;
}
// stage 2 (one thread per (b, j), reverse scan through the low-pass recursion), in place on dout[.., j]
__global__ void actor_head_train_bwd_kernel(const float* __restrict__ keep, HeadParams hp, int T, int B, float* __restrict__ dout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * KBJ_NU) return;
  int b = i / KBJ_NU, j = i % KBJ_NU;
  float gcarry = 0;  // gradient wrt the filter state entering step t+1 (before the keep mask of step t)
  constexpr int U = 10;
  for (int t1 = T - 1; t1 >= 0; t1 -= U) {
    float gdir[U], kp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = t1 - u;
      if (t >= 0) { size_t r = (size_t)t * B + b; gdir[u] = dout[r * 40 + j]; kp[u] = keep[r]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = t1 - u;
      if (t >= 0) {
        float gy = gdir[u] + kp[u] * gcarry;
        dout[((size_t)t * B + b) * 40 + j] = hp.alpha * gy;
        gcarry = (1 - hp.alpha) * gy;
      }
    }
  }
}

// y[m] = sum_k W[m][k] x[k] + add[m]   (W [M][K] row-major; one thread per row)
// one wavefront per output row (launch: M / 4 blocks of 256 threads): coalesced reads along k, DPP-free shuffle reduction
__global__ void matvec_kernel(const float* __restrict__ W, const float* __restrict__ x, const float* __restrict__ add, int M, int K, float* __restrict__ y) {
  int m = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (m >= M) return;
  float s = 0;
  for (int k = l; k < K; k += 64) s += W[(size_t)m * K + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (l == 0) y[m] = s + (add ? add[m] : 0.0f);
}
// y[n] += sum_k W[k][n] x[k]   (W [K][N] row-major): block = 64 columns x 4 row phases, grid.y slices of k, one atomic per column and
// block (a single thread per column walking all K rows serially took 270 us on the critical path of every minibatch)
// part != null (deterministic mode): block row blockIdx.y stores its sums to part[blockIdx.y][N] and reduce_rows_kernel adds the rows in order
__global__ void matvec_t_acc_kernel(const float* __restrict__ W, const float* __restrict__ x, int K, int N, float* __restrict__ y, float* __restrict__ part) {
  __shared__ float red[4][64];
  int n = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0;
  if (n < N) for (int k = ph + 4 * blockIdx.y; k < K; k += 4 * gridDim.y) s += W[(size_t)k * N + n] * x[k];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && n < N) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (part) part[(size_t)blockIdx.y * N + n] = v; else atomicAdd(&y[n], v);
  }
}

// dst[r][0:ld_dst] = src[r][0:cols] followed by zeros: a weight matrix whose row length is not a multiple of 4 (the critic's 475-wide
// input projection) re-pitched to 16-byte aligned rows, so that the GEMM takes its vector-free buffer-load path for it
__global__ void repitch_rows_kernel(const float* __restrict__ src, int rows, int cols, int ld_dst, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld_dst) return;
  const int r = i / ld_dst, c = i - r * ld_dst;
  dst[i] = c < cols ? src[(size_t)r * cols + c] : 0.0f;
}

// C[m][n] += u[m] v[n]   (rank-1 update, C [M][N] row-major)
__global__ void outer_acc_kernel(float* __restrict__ C, const float* __restrict__ u, const float* __restrict__ v, int M, int N) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  C[i] += u[i / N] * v[i % N];
}

// column sums: out[n] (+)= sum_m X[m][n]  (bias gradients); one block per 64 columns, 256 threads = 4 row phases
// out[c] += sum over rows p (in order) of part[p][c]: the fixed-order second stage of the fp32 column sums (deterministic mode)
__global__ void reduce_rows_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.0f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + c];
  out[c] += s;
}
__global__ void colsum_kernel(const float* __restrict__ X, int M, int N, int ld, float* __restrict__ out, float* __restrict__ part) {
  __shared__ float red[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0;
  if (c < N) for (int m = ph + 4 * blockIdx.y; m < M; m += 4 * gridDim.y) s += X[(size_t)m * ld + c];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < N) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (part) part[(size_t)blockIdx.y * N + c] = v; else atomicAdd(&out[c], v);
  }
}

// ---- GAE (gamma, lambda train.py:1769-1770): one thread per env, reverse scan ------------------------------------------
__global__ void gae_kernel(const float* __restrict__ value, const float* __restrict__ reward, const float* __restrict__ aux, int T, int N,
                           float gamma, float lam, float* __restrict__ adv, float* __restrict__ target) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float last = 0;
  for (int t = T - 1; t >= 0; --t) {
    size_t r = (size_t)t * N + n;
    float keep = aux[r * KBJ_AUX_SIZE + KBJ_AUX_DONE] != 0 ? 0.0f : 1.0f;
    float v = value[r], vn = t + 1 < T ? value[r + N] : v;
    float delta = reward[r] + gamma * vn * keep - v;
    last = delta + gamma * lam * keep * last;
    adv[r] = last; target[r] = last + v;
  }
}

// ---- AdamW with global-norm clipping (optax.adamw, train.py:1059-1077) -------------------------------------------------
__global__ void sumsq_kernel(const float* __restrict__ g, size_t n, float scale, double* __restrict__ out, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double v = (double)g[i] * scale; s += v * v; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { if (part) part[blockIdx.x] = red[0]; else atomicAdd(out, red[0]); }
}
struct AdamParams { float lr, b1, b2, eps, wd, max_norm, bc1, bc2, gscale; };
// Fail-stop: a gradient whose global norm is not finite leaves parameters and moments untouched and raises err[1]; err[0] is the
// hand-off timeout flag of the persistent recurrences (kbj_lstm_seq.h), which also poisons the gradient (lane_tail_kernel) so that
// after the data-parallel all-reduce EVERY rank skips the step instead of applying a truncated gradient. Both are reported by
// kbj_synchronize.
// The last kernel of a lane of kbj_ppo_grad: clear the hand-off counter blocks [phase][net][words] of the lane's nets (net & 1 == lane; lane < 0:
// all) and, when a recurrence timed out (err[0]), poison the lane's slice(s) of the gradient with a NaN marker (fail-stop, see adamw_kernel)
__global__ void lane_tail_kernel(unsigned* __restrict__ counters, int lane, const unsigned* __restrict__ err, float* __restrict__ g0, float* __restrict__ g1) {
  const int net = blockIdx.x & 3;
  if (blockIdx.x == 0 && threadIdx.x == 0 && err[0]) { g0[0] = __int_as_float(0x7FC00000); if (g1) g1[0] = __int_as_float(0x7FC00000); }
  if (lane >= 0 && (net & 1) != lane) return;
  counters[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = 0u;
}
__global__ void poison_vars_kernel(const unsigned* __restrict__ err, float* __restrict__ logp, float* __restrict__ value) {
  if (err[0]) { logp[0] = __int_as_float(0x7FC00000); value[0] = __int_as_float(0x7FC00000); }
}
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g, size_t n,
                             const double* __restrict__ sumsq, AdamParams ap, unsigned* __restrict__ err) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double ss = *sumsq;
  if (!(ss <= 1.0e300) || err[0]) {   // NaN / inf norm, or a recurrence timed out on this rank
    if (i == 0) err[1] = 1u;
    return;
  }
  float norm = (float)sqrt(ss);
  float clip = fminf(ap.max_norm / (norm + 1e-6f), 1.0f);
  float gi = g[i] * ap.gscale * clip;
  float mi = ap.b1 * m[i] + (1 - ap.b1) * gi, vi = ap.b2 * v[i] + (1 - ap.b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float mh = mi / ap.bc1, vh = vi / ap.bc2;
  p[i] -= ap.lr * (mh / (sqrtf(vh) + ap.eps) + ap.wd * p[i]);
}

// ---- free hidden_size: the caller's layout (hidden size Hu) <-> the kernels' (H = Hu rounded up to 64 units, zero padded) --------------
// one parameter tensor as [nblk][rows][cols] in both layouts (LSTM weights: nblk = 4 gate blocks of Hu / H rows)
struct PadDesc { unsigned long long uoff, ioff; int nblk, rows_u, rows_i, cols_u, cols_i; };

// one thread per element of the INTERNAL vector; to_user == 0: internal <- caller's value or 0; != 0: caller's <- internal (padding dropped)
__global__ void pad_params_kernel(const PadDesc* __restrict__ desc, int nd, float* user, float* internal, size_t n_internal, int to_user) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_internal) return;
  int d = 0;
  while (d + 1 < nd && desc[d + 1].ioff <= i) ++d;
  const PadDesc t = desc[d];
  const size_t local = i - t.ioff;
  const int c = (int)(local % t.cols_i), r = (int)((local / t.cols_i) % t.rows_i), b = (int)(local / ((size_t)t.cols_i * t.rows_i));
  const bool real = r < t.rows_u && c < t.cols_u;
  const size_t u = t.uoff + ((size_t)b * t.rows_u + r) * t.cols_u + c;
  if (to_user) { if (real) user[u] = internal[i]; }
  else internal[i] = real ? user[u] : 0.0f;
}

// rows of width ws -> rows of width wd: the common columns copied, the rest of a wider destination zeroed
__global__ void repitch_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int ws, int wd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * wd) return;
  const size_t r = i / wd; const int c = (int)(i % wd);
  dst[i] = c < ws ? src[r * ws + c] : 0.0f;
}

// ---- parameter init: U(+-1/sqrt(fan_in)) per leaf (equinox default), threefry stream KBJ_RNG_INIT ------------------------
__global__ void init_uniform_kernel(float* __restrict__ p, size_t n, float bound, uint32_t seed, uint32_t leaf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t b0, b1;
  nn_threefry(seed ^ ((uint32_t)KBJ_RNG_INIT * 0x9E3779B9u), leaf, (uint32_t)(i >> 32), (uint32_t)i, b0, b1);
  float u = (float)(b0 >> 8) * (1.0f / 16777216.0f);
  p[i] = fmaf(2 * bound, u, -bound);
}

}  // namespace kbj
