// kbj_nn.hip — LSTM actor-critic + PPO update on the GPU (C ABI: kbj_policy_step, kbj_rollout, kbj_gae, kbj_ppo_grad,
// kbj_adamw_step, kbj_init_params). SURVEY.md §8 rows a4-a13. All matrix products run on the fp32 matrix cores
// (kbj_gemm.h); activations needed by back-propagation-through-time are stashed in HBM (sized for 288 GB).
#include <hip/hip_runtime.h>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include "kbj_ctx.h"
#include "kbj_gemm.h"
#include "kbj_nn_kernels.h"
#include "kbj_lstm_seq.h"
#include "kbj_lstm_bwd16.h"

using namespace kbj;

namespace {

constexpr int MAXD = KBJ_MAX_DEPTH;   // LSTM layers per net (config.depth = 1..MAXD; train.py:82-85 default 2)

struct NetOff {  // float offsets into the flat parameter vector (kbj.h layout = equinox leaf order)
  size_t w_in, b_in, w_ih[MAXD], w_hh[MAXD], b[MAXD], w_out, b_out;
  int nin, nout, ld_obs;
};

struct TrainBufs {  // per net, minibatch-sized
  float *obs, *X0, *G[MAXD], *Hm[MAXD], *Hout[MAXD], *Cm[MAXD], *TanhC[MAXD], *Out, *dOut, *dHa, *dHb, *dGl[MAXD];
};

// Formulation switches of the schedule. Read from the environment ONCE PER CONTEXT (kbj_create), so that a test process can build contexts
// under different settings; every non-default value below is exercised by a parity test (tests/test_gpu_switches.py) and README.md lists
// exactly these. `deterministic` comes from kbj_config (KBJ_DETERMINISTIC=1 forces it on).
struct Sched {
  bool fold_actor = true;          // KBJ_FOLD_ACTOR=0: actor input projection as its own GEMM (65 -> H -> 4H) instead of folded into layer 0
  bool fold_critic = true;         // KBJ_FOLD_CRITIC=0: critic layer-0 backward through dX0 instead of Z = dG0^T obs
  bool fuse_ih = true;             // KBJ_SEQ_FUSE=0: input products x W_ih^T as GEMM launches in front of the forward recurrences
  bool fuse_obs = true;            // KBJ_SEQ_FUSE_OBS=0: the folded actor layer 0 as a GEMM launch instead of inside its recurrence
  bool fused_critic_head = true;   // KBJ_FUSED_CRITIC_HEAD=0: critic head as output GEMM + value kernel + loss kernel + K = 1 GEMM
  bool rollout_step = true;        // KBJ_ROLLOUT_STEP=0: rollout layers as [x | h] gate GEMM + cell kernel instead of lstm_step_kernel
  bool one_stream = false;         // KBJ_ONE_STREAM=1: the whole update on the caller's stream (no lanes)
  bool debug_sync = false;         // KBJ_DEBUG=1: kbj_ppo_grad synchronises and reports device-side errors at the call that caused them
  bool deterministic = false;      // fixed-order reductions instead of fp32 / fp64 atomics (bit-reproducible update)
  bool bwd16 = true;               // KBJ_BWD16=0: backward recurrences on the 32-row x 32-unit form of rounds 1-4 (lstm_seq_bwd_kernel) instead of 16-row x 64-unit
                                   // tiles with the partner-major contraction (kbj_lstm_bwd16.h: 620 instead of 907 us per launch in situ)
  bool critic_on_caller = true;    // KBJ_CRITIC_LANE=2nd: the critic's chain on the context's SECOND stream (rounds 1-5). Default (round 6): the critic - the longer
                                   // chain, the one a minibatch waits for - runs on the caller's stream, so that nothing between the optimizer step and the
                                   // critic's first kernel, nor between its last kernel and the next optimizer step, crosses a queue (a cross-queue event wait
                                   // costs 10-25 us on this runtime); the actor's chain, which has ~0.3 ms of slack, takes the second stream and the hops
  bool gemm_x3 = false;            // kbj_config.gemm_bf16x3 / KBJ_GEMM_X3=1: the GEMM launches that are eligible (the update's input gradients, weight-gradient
                                   // pairs and critic input projection, the rollout's [x | h] gate GEMMs) on the bf16 matrix cores through the exact three-way
                                   // operand split (kbj_gemm.h gemm_x3_kernel); not the default, not the headline
};
bool env_flag(const char* name, bool dflt) { const char* v = getenv(name); return v ? atoi(v) != 0 : dflt; }

constexpr int MAX_PAD_DESC = 2 * (4 + 3 * MAXD);

struct NnWs {
  Sched sched;
  int H = 0, N = 0, B = 0, T = 0, D = 2;
  // hidden_size is free (train.py:78-81); the kernels tile hidden units in groups of 64. Hu = the caller's value, H = the kernels'.
  // Hu != H: every entry point that takes parameters, carries or a gradient converts at the boundary (zero padding is exact for this
  // network: a padded unit has zero weights and bias, so its gates are sigma(0), tanh(0), its cell stays 0, its output 0, and every
  // gradient into or out of it is 0) and runs the H-wide schedule on the internal copies below.
  int Hu = 0;
  bool inner = false;                 // set while an entry point runs on the internal copies
  size_t unparams = 0, unactor = 0;   // parameter counts in the caller's layout
  PadDesc* pad_desc = nullptr; int npad = 0;
  float *pparams = nullptr, *pgrad = nullptr;
  float *phc[4] = {nullptr, nullptr, nullptr, nullptr}, *pc0[4] = {nullptr, nullptr, nullptr, nullptr};   // carries / trajectory start carries [D][2][N][H]
  bool padded() const { return Hu != H; }
  // deterministic mode: per-lane workspaces (lane = stream: caller's, second, side 0, side 1) for split-K slabs and reduction partials
  float* skws[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t skws_cap[4] = {0, 0, 0, 0};
  float* detp[4] = {nullptr, nullptr, nullptr, nullptr};   // [512][1024] floats each
  double* detd = nullptr;                                    // [512][2] doubles (advantage statistics, gradient norm)
  NetOff net[2];
  size_t nparams = 0, nactor = 0;
  // rollout scratch
  float *rX[4] = {nullptr, nullptr, nullptr, nullptr}, *rG[4] = {nullptr, nullptr, nullptr, nullptr}, *rOut[4] = {nullptr, nullptr, nullptr, nullptr};
  float* joint_bias_d = nullptr;
  // mirror aux losses (nets 2, 3 = actor, critic evaluated on mirrored observations with the SAME weights)
  bool mirror = false;
  int nnets = 2;
  MirrorEntry* mtab[2] = {nullptr, nullptr};
  float *rObsM[2] = {nullptr, nullptr};   // rollout: mirrored observation rows [N][ld]
  float* rH[4][MAXD] = {};                  // rollout: the h planes' ping-pong partners [N][H] per (net, layer)
  float *y_m = nullptr, *sd_m = nullptr, *value_m = nullptr, *lpf0_m = nullptr, *dy = nullptr, *dy_m = nullptr, *dvalue_m = nullptr, *zeroR = nullptr;
  // training
  TrainBufs tb[4];
  float *keep = nullptr, *act = nullptr, *logp_old = nullptr, *val_old = nullptr, *adv = nullptr, *target = nullptr;
  float *y = nullptr, *sd = nullptr, *logp = nullptr, *ent = nullptr, *value = nullptr, *dlogp = nullptr, *dvalue = nullptr, *lpf0 = nullptr;
  // actor layer 0 with the input projection folded in (the projection has no activation and 65 < H inputs):
  // W_eff = W_ih0 W_in [4H][68 (65 used)], b_eff = W_ih0 b_in + b_0; Zeff[n] = dG0^T obs of actor-type net n (n = 0, 2)
  float *Weff = nullptr, *beff = nullptr, *Zeff[4] = {nullptr, nullptr, nullptr, nullptr};
  float* WinP[2] = {nullptr, nullptr};   // input-projection weights re-pitched to the observation rows' 16-byte aligned stride (per call: the parameters change)
  double* stats = nullptr;  // [0..1] adv stats, [2..9] metric accumulators, [10] grad sumsq, [11] sample count of caller-supplied adv sums (0: own minibatch)
  const int32_t* prefetched_idx = nullptr;   // kbj_ppo_prefetch: the gathers of the next kbj_ppo_grad / kbj_ppo_forward with these indices are already queued
  const kbj_traj* prefetched_traj = nullptr;
  const double* ext_adv_sums = nullptr;   // kbj_set_advantage_sums: (sum adv, sum adv^2, count) on the device, used instead of the minibatch's own statistics
  unsigned* seq_counters = nullptr;  // per row-group arrival counters of the persistent LSTM kernels
  unsigned* seq_err = nullptr;       // spin-timeout flag
  bool counters_clean = false;       // the hand-off counters were cleared at the tail of the last kbj_ppo_grad (per lane, behind its last recurrence): the next call skips its own clear
  bool sumsq_clean = false;          // stats[10] (the gradient's sum of squares) has been zeroed by kbj_ppo_grad's own clear and not used since: kbj_adamw_step skips its memset
  int seq_grid = 0, seq_slots = 0;   // workgroups of one recurrence launch / resident workgroups of the worst-fitting recurrence kernel (kbj_recurrence_residency)
  long long* seq_stamps = nullptr;   // diagnostics (KBJ_SEQ_STAMPS=1): per-step clock stamps of one workgroup
  long long* seq_bstamps = nullptr;  // same for a backward recurrence (KBJ_SEQ_BSTAMPS = 1 + net + 2 * layer)
  std::vector<void*> allocs;
};

NnWs* ws_of(kbj_ctx* ctx) { return reinterpret_cast<NnWs*>(ctx->nn_ws); }

void layout_params(NnWs& w, int H, int D, int extra_actor = 0, int extra_critic = 0) {
  w.D = D;
  size_t off = 0;
  for (int n = 0; n < 2; ++n) {
    NetOff& o = w.net[n];
    o.nin = n == 0 ? KBJ_NOBS_ACTOR + extra_actor : KBJ_NOBS_CRITIC + extra_critic;     // user observation columns behind the reference's (kbj_model.h)
    o.ld_obs = KBJ_LD_OF(o.nin);
    o.nout = n == 0 ? 2 * KBJ_NU : 1;
    o.w_in = off; off += (size_t)H * o.nin;
    o.b_in = off; off += H;
    for (int l = 0; l < D; ++l) {
      o.w_ih[l] = off; off += (size_t)4 * H * H;
      o.w_hh[l] = off; off += (size_t)4 * H * H;
      o.b[l] = off; off += (size_t)4 * H;
    }
    o.w_out = off; off += (size_t)o.nout * H;
    o.b_out = off; off += o.nout;
    if (n == 0) w.nactor = off;
  }
  w.nparams = off;
}

// Mirror of the packed observation rows as (source index, multiplier, offset) per element (mirror_rows_kernel).
// Follows the index/sign lists of the reference's mirror_obs functions (train.py:1574-1756); element order = the obs packing
// of kbj_env_task.h write_obs (kbj_model.h KBJ_NOBS_*).
// extra_actor / extra_critic user columns behind the reference's are carried over unchanged (identity entries): a user term that is not
// mirror-invariant has to be mirrored by the user's own mirror-loss code, as in the reference (train.py:1574-1756 names every key).
void build_mirror_tables(const kbj_model& m, std::vector<MirrorEntry>& ta, std::vector<MirrorEntry>& tc, int extra_actor = 0, int extra_critic = 0) {
  auto swp = [](int i) { return i < 5 ? i + 5 : (i < 10 ? i - 5 : i); };  // left leg <-> right leg, arms stay (train.py:1574-1582)
  tc.assign(KBJ_LD_CRITIC, MirrorEntry{0, 0.0f, 0.0f});
  for (int k = 0; k < KBJ_LD_CRITIC; ++k) tc[k].src = k;
  auto keep = [&](int k, float sgn) { tc[k] = MirrorEntry{k, sgn, 0.0f}; };
  for (int i = 0; i < KBJ_NU; ++i) {
    int s = swp(i);
    auto rng = [&](int j) { return std::fmax(m.joint_bias[j] - m.joint_lo[j], m.joint_hi[j] - m.joint_bias[j]); };
    tc[KBJ_OBS_JPOS + i] = MirrorEntry{KBJ_OBS_JPOS + s, -rng(s) / rng(i), (-m.joint_bias[s] - m.joint_bias[i]) / rng(i)};   // normalised joint positions
    tc[KBJ_OBS_JVEL + i] = MirrorEntry{KBJ_OBS_JVEL + s, -1.0f, 0.0f};                                                      // joint velocities / 10
    tc[KBJ_OBS_ACTFRC + i] = MirrorEntry{KBJ_OBS_ACTFRC + s, -1.0f, 0.0f};                                                  // actuator force / 4 (critic)
  }
  // roll, pitch, unit projected gravity: the reference mirrors the raw vector (g0, -g1, g2) and encodes it again (train.py:1596-1603,
  // 1338-1349): roll = atan2(g1, -g2) changes sign, pitch and the norm do not
  keep(KBJ_OBS_PG, -1); keep(KBJ_OBS_PG + 1, 1); keep(KBJ_OBS_PG + 2, 1); keep(KBJ_OBS_PG + 3, -1); keep(KBJ_OBS_PG + 4, 1);
  keep(KBJ_OBS_GYRO, -1); keep(KBJ_OBS_GYRO + 1, 1); keep(KBJ_OBS_GYRO + 2, -1);   // gyro
  keep(KBJ_OBS_ZEROCMD, 1);                                                          // zero-command flag (the norm of cmd[0:3] is mirror invariant)
  const float cs[16] = {1, -1, -1, 1, -1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // vx, vy, wz, height, roll, pitch, 10 arm targets
  for (int k = 0; k < 16; ++k) keep(KBJ_OBS_CMD + k, cs[k]);
  ta.assign(tc.begin(), tc.begin() + KBJ_LD_ACTOR);
  for (int k = KBJ_NOBS_ACTOR; k < KBJ_LD_ACTOR; ++k) ta[k] = MirrorEntry{k, 0.0f, 0.0f};
  tc[KBJ_OBS_TOUCH] = MirrorEntry{KBJ_OBS_TOUCH + 1, 1, 0}; tc[KBJ_OBS_TOUCH + 1] = MirrorEntry{KBJ_OBS_TOUCH, 1, 0};           // foot touch L <-> R
  const float fs[3] = {1, -1, 1};
  for (int k = 0; k < 3; ++k) { tc[KBJ_OBS_FEETPOS + k] = MirrorEntry{KBJ_OBS_FEETPOS + 3 + k, fs[k], 0}; tc[KBJ_OBS_FEETPOS + 3 + k] = MirrorEntry{KBJ_OBS_FEETPOS + k, fs[k], 0}; }   // feet positions
  keep(KBJ_OBS_BASEPOS, 1); keep(KBJ_OBS_BASEPOS + 1, 1); keep(KBJ_OBS_BASEPOS + 2, 1);                                      // base position
  keep(KBJ_OBS_BASEQUAT, 1); keep(KBJ_OBS_BASEQUAT + 1, -1); keep(KBJ_OBS_BASEQUAT + 2, -1); keep(KBJ_OBS_BASEQUAT + 3, 1);  // base quaternion
  const float ci[10] = {1, 1, -1, 1, 1, 1, 1, -1, 1, -1}, cv[6] = {1, -1, 1, -1, 1, -1};
  for (int b = 0; b < 23; ++b) {
    for (int k = 0; k < 10; ++k) keep(KBJ_OBS_CINERT + 10 * b + k, ci[k]);      // cinert
    for (int k = 0; k < 6; ++k) keep(KBJ_OBS_CVEL + 6 * b + k, cv[k]);          // cvel
  }
  keep(KBJ_OBS_LINVEL, 1); keep(KBJ_OBS_LINVEL + 1, -1); keep(KBJ_OBS_LINVEL + 2, 1);     // base linear velocity
  keep(KBJ_OBS_ANGVEL, -1); keep(KBJ_OBS_ANGVEL + 1, 1); keep(KBJ_OBS_ANGVEL + 2, -1);    // base angular velocity
  keep(KBJ_OBS_HEIGHT, 1);                                                                 // base height
  for (int k = KBJ_NOBS_CRITIC; k < KBJ_LD_CRITIC; ++k) tc[k] = MirrorEntry{k, 0.0f, 0.0f};
  auto widen = [](std::vector<MirrorEntry>& t, int nobs, int extra) {
    t.resize(KBJ_LD_OF(nobs + extra));
    for (int k = nobs; k < (int)t.size(); ++k) t[k] = MirrorEntry{k, k < nobs + extra ? 1.0f : 0.0f, 0.0f};
  };
  if (extra_actor > 0) widen(ta, KBJ_NOBS_ACTOR, extra_actor);
  if (extra_critic > 0) widen(tc, KBJ_NOBS_CRITIC, extra_critic);
}

template <class T> int dalloc(kbj_ctx* ctx, NnWs& w, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) return kbj_fail(ctx, std::string("hipMalloc (nn workspace): ") + hipGetErrorString(e));
  w.allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

thread_local int g_gemm_x3 = 0;     // set by kbj_ppo_grad from the context's schedule for the duration of the call (the wrappers below have no context)
constexpr int g_fold_sk = 8;        // k slices of the small W_ih0^T Z product of the folded input projections
#ifndef KBJ_SPLITK_WGS
#define KBJ_SPLITK_WGS 768   // re-swept in round 5 under the four-launches-together schedule: 256 / 384 / 512 / 768 / 1024 -> 6.10 / 6.0 / 5.96 / 5.86 / 5.95 ms per minibatch
#endif
constexpr int g_splitk_wgs = KBJ_SPLITK_WGS;   // target number of workgroups of a split-K weight-gradient GEMM (512 ... 1536 measured flat)
constexpr int DETP_ROWS = 512, DETP_COLS = 4 * 512;   // (columns: one gate row of the widest layer, 4 SEQ_MAX_H)

// lane index of a stream of this context (deterministic-mode workspaces are per lane: launches on different lanes overlap)
int lane_of(kbj_ctx* ctx, hipStream_t s) { return s == ctx->stream ? 0 : (s == ctx->stream2 ? 1 : (s == ctx->side[0] ? 2 : 3)); }
// split-K slab workspace of the lane of stream s, grown on demand (deterministic mode only; null otherwise = atomics)
float* sk_workspace(kbj_ctx* ctx, hipStream_t s, size_t floats);
float* det_partials(kbj_ctx* ctx, hipStream_t s);

inline dim3 g1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// y = x W^T + b  (x [M][K] lda, W [N][K])
void linear_fwd(hipStream_t s, const float* x, int lda, const float* W, int ldw, const float* bias, float* y, int ldy, int M, int N, int K, int beta) {
  GemmArgs g{x, W, y, bias, M, N, K, lda, ldw, ldy, beta, 1, nullptr};
  gemm_launch<true, true>(s, g);
}
// dx = dy W   (dy [M][K=nout] , W [K][N])
void linear_bwd_input(hipStream_t s, const float* dy, int lddy, const float* W, int ldw, float* dx, int lddx, int M, int N, int K, int beta, int tile = -1) {
  GemmArgs g{dy, W, dx, nullptr, M, N, K, lddy, ldw, lddx, beta, 1, nullptr};
  g.x3 = g_gemm_x3;
  gemm_launch<true, false>(s, g, g_gemm_x3 ? -1 : tile);   // tile = 2: 64 x 128 tiles (kbj_gemm.h: few 128 x 128 tiles quantise badly)
}
// dW[Nout][Nin] += dy^T x  (dy [R][Nout], x [R][Nin]); split-K over the R samples with atomics (dW pre-zeroed by the caller)
void linear_bwd_weight(kbj_ctx* ctx, hipStream_t s, const float* dy, int lddy, const float* x, int ldx, float* dW, int lddw, int Nout, int Nin, int R) {
  // the long contraction (R = T*B samples) is split over the grid: 128x128 tiles when the output allows it, ~768 workgroups
  bool big = Nout >= 128 && Nin >= 128;
  int ts = big ? 128 : 64;
  int tiles = ((Nout + ts - 1) / ts) * ((Nin + ts - 1) / ts);
  int sk = std::max(2, std::min(256, g_splitk_wgs / std::max(1, tiles)));
  sk = std::max(2, std::min(sk, (R + 255) / 256));
  GemmArgs g{dy, x, dW, nullptr, Nout, Nin, R, lddy, ldx, lddw, 1, sk, nullptr};  // always the split-K path: accumulates into dW
  g.skws = sk_workspace(ctx, s, (size_t)sk * Nout * Nin);
  g.x3 = g_gemm_x3;
  gemm_launch<false, false>(s, g, big ? 1 : 0);
}

// two weight gradients that share dy in one launch: dW1 += dy^T x1, dW2 += dy^T x2 (x1, x2 [R][Nin] with the same ld)
void linear_bwd_weight2(kbj_ctx* ctx, hipStream_t s, const float* dy, int lddy, const float* x1, const float* x2, int ldx, float* dW1, float* dW2, int lddw, int Nout, int Nin, int R) {
  bool big = Nout >= 128 && Nin >= 128;
  int ts = big ? 128 : 64;
  if (Nin % ts != 0) {  // the column split must fall on a tile boundary
    linear_bwd_weight(ctx, s, dy, lddy, x1, ldx, dW1, lddw, Nout, Nin, R);
    linear_bwd_weight(ctx, s, dy, lddy, x2, ldx, dW2, lddw, Nout, Nin, R);
    return;
  }
  int tiles = ((Nout + ts - 1) / ts) * (2 * Nin / ts);
  int sk = std::max(2, std::min(256, g_splitk_wgs / std::max(1, tiles)));
  sk = std::max(2, std::min(sk, (R + 255) / 256));
  GemmArgs g{dy, x1, dW1, nullptr, Nout, 2 * Nin, R, lddy, ldx, lddw, 1, sk, nullptr};
  g.B2 = x2; g.C2 = dW2; g.n1 = Nin;
  g.skws = sk_workspace(ctx, s, (size_t)sk * Nout * 2 * Nin);
  g.x3 = g_gemm_x3;
  gemm_launch<false, false>(s, g, big ? 1 : 0);
}
// column sums of X [M][N] (ld) added to out[N]: atomics over 512 row slices, or (deterministic) per-slice partials + ordered reduce
void colsum_acc(kbj_ctx* ctx, hipStream_t s, const float* X, int M, int N, int ld, float* out) {
  float* part = det_partials(ctx, s);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, DETP_ROWS), dim3(256), 0, s, X, M, N, ld, out, part);
  if (part) hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, DETP_ROWS, N, out);
}

// recurrence workgroups are 8 wavefronts owning 32 hidden units (kbj_lstm_seq.h; the 4-wavefront / 16-unit form of round 1 is gone:
// slower in situ at every size)
constexpr int SEQ_UW = 2;
// Hidden sizes above this run "wide": the forward recurrence keeps only its W_hh slice in registers (two slices of H / 4 registers each do
// not fit beyond 256), so the input products are GEMM launches in front of it; the backward recurrence takes the registers it needs
// (lstm_seq_bwd_wide_kernel); the rollout's layers are [x | h] gate GEMM + cell kernel; and the update runs on one stream when two
// recurrence launches would not be resident together. Served, not tuned: the launch configuration is 256.
constexpr int SEQ_FUSED_MAX_H = 256;
constexpr int SEQ_MAX_H = 512;
constexpr int SEQ_COUNTER_WORDS = 256;   // hand-off words per recurrence launch (one per workgroup): the grid of a launch may not exceed it
constexpr int SEQ_COUNTER_TOTAL = 2 * MAXD * 4 * SEQ_COUNTER_WORDS;   // all launches of one call (forward + backward, MAXD layers, 4 nets): ONE clear
// fault injection for the tests (KBJ_DEBUG_DROP_SEQ_WG = n at kbj_create): the next n forward-recurrence launches run with one
// workgroup missing, so its partners' bounded spins expire and the timeout / fail-stop path is exercised on real hardware
int g_seq_drop = 0;
int g_seq_drop_bwd = 0;   // the same for the backward recurrences (KBJ_DEBUG_DROP_SEQ_BWD_WG = n)
unsigned g_seq_timeout_ticks = SEQ_TIMEOUT_MS * 100000u;   // wall_clock64 ticks; set per context in kbj_nn_create
template <int H, int UW> void seq_fwd_launch(hipStream_t s, const SeqFwdArgs& a0) {
  SeqFwdArgs a = a0;
  a.timeout_ticks = g_seq_timeout_ticks;
  int grid = (H / (SEQ_UNITS * UW)) * ((a.B + SEQ_ROWS - 1) / SEQ_ROWS);
  if (g_seq_drop > 0 && grid > 1) { --g_seq_drop; --grid; }
  if constexpr (H <= SEQ_FUSED_MAX_H) {
    if (a.X && a.ldx == KBJ_LD_ACTOR) { hipLaunchKernelGGL((lstm_seq_fwd_kernel<H, UW, true, KBJ_LD_ACTOR>), dim3(grid), dim3(256 * UW), 0, s, a); return; }   // gates from the observation rows
    if (a.X) { hipLaunchKernelGGL((lstm_seq_fwd_kernel<H, UW, true>), dim3(grid), dim3(256 * UW), 0, s, a); return; }   // input projection fused
  }
  hipLaunchKernelGGL((lstm_seq_fwd_kernel<H, UW, false>), dim3(grid), dim3(256 * UW), 0, s, a);   // (wide layers: the schedule never passes X, kbj_nn_create)
}
template <int H, int UW> void seq_bwd_launch(hipStream_t s, const SeqBwdArgs& a0) {
  SeqBwdArgs a = a0;
  a.timeout_ticks = g_seq_timeout_ticks;
  int grid = (H / (SEQ_UNITS * UW)) * ((a.B + SEQ_ROWS - 1) / SEQ_ROWS);
  if constexpr (H <= SEQ_FUSED_MAX_H) hipLaunchKernelGGL((lstm_seq_bwd_kernel<H, UW>), dim3(grid), dim3(256 * UW), 0, s, a);
  else hipLaunchKernelGGL((lstm_seq_bwd_wide_kernel<H, UW>), dim3(grid), dim3(256 * UW), 0, s, a);
}
int seq_fwd(kbj_ctx* ctx, hipStream_t st, int H, const SeqFwdArgs& a) {   // a.counters: zeroed by the caller
  KbjKernelTimer timer(st, a.X ? (a.ldx == KBJ_LD_ACTOR ? KBJ_KIND_SEQ_FWD_OBS : KBJ_KIND_SEQ_FWD_FUSED) : KBJ_KIND_SEQ_FWD, 2.0 * a.T * a.B * 4.0 * H * (H + (a.X ? (a.kx ? a.kx : H) : 0)));
  switch (H) {
    case 64: seq_fwd_launch<64, SEQ_UW>(st, a); break;
    case 128: seq_fwd_launch<128, SEQ_UW>(st, a); break;
    case 192: seq_fwd_launch<192, SEQ_UW>(st, a); break;
    case 256: seq_fwd_launch<256, SEQ_UW>(st, a); break;
    case 320: seq_fwd_launch<320, SEQ_UW>(st, a); break;
    case 384: seq_fwd_launch<384, SEQ_UW>(st, a); break;
    case 448: seq_fwd_launch<448, SEQ_UW>(st, a); break;
    case 512: seq_fwd_launch<512, SEQ_UW>(st, a); break;
    default: return kbj_fail(ctx, "persistent LSTM kernels are built for hidden sizes 64, 128, ..., 512");
  }
  return 0;
}
template <int H> void seq_bwd16_launch(hipStream_t s, const SeqBwdArgs& a0) {
  SeqBwdArgs a = a0;
  a.timeout_ticks = g_seq_timeout_ticks;
  int grid = (H / BWD16_UNITS) * ((a.B + BWD16_ROWS - 1) / BWD16_ROWS);
  if (g_seq_drop_bwd > 0 && grid > 1 && H > BWD16_UNITS) { --g_seq_drop_bwd; --grid; }   // fault injection (a launch without partners has nobody to time out)
  hipLaunchKernelGGL((lstm_seq_bwd16_kernel<H>), dim3(grid), dim3(BWD16_NTH), 0, s, a);
}
// row groups of a backward-recurrence launch (deterministic mode: rows of its per-row-group bias partials)
int seq_bwd_row_groups(int B, bool tiles16) { return tiles16 ? (B + BWD16_ROWS - 1) / BWD16_ROWS : (B + SEQ_ROWS - 1) / SEQ_ROWS; }
int seq_bwd(kbj_ctx* ctx, hipStream_t st, int H, const SeqBwdArgs& a, bool tiles16 = false) {   // a.counters: zeroed by the caller
  KbjKernelTimer timer(st, tiles16 ? KBJ_KIND_SEQ_BWD16 : KBJ_KIND_SEQ_BWD, 2.0 * a.T * a.B * 4.0 * H * H);
  if (tiles16) {
    switch (H) {
      case 64: seq_bwd16_launch<64>(st, a); break;
      case 128: seq_bwd16_launch<128>(st, a); break;
      case 192: seq_bwd16_launch<192>(st, a); break;
      case 256: seq_bwd16_launch<256>(st, a); break;
      default: return kbj_fail(ctx, "lstm_seq_bwd16_kernel is built for hidden sizes 64, 128, 192, 256");
    }
    return 0;
  }
  switch (H) {
    case 64: seq_bwd_launch<64, SEQ_UW>(st, a); break;
    case 128: seq_bwd_launch<128, SEQ_UW>(st, a); break;
    case 192: seq_bwd_launch<192, SEQ_UW>(st, a); break;
    case 256: seq_bwd_launch<256, SEQ_UW>(st, a); break;
    case 320: seq_bwd_launch<320, SEQ_UW>(st, a); break;
    case 384: seq_bwd_launch<384, SEQ_UW>(st, a); break;
    case 448: seq_bwd_launch<448, SEQ_UW>(st, a); break;
    case 512: seq_bwd_launch<512, SEQ_UW>(st, a); break;
    default: return kbj_fail(ctx, "persistent LSTM kernels are built for hidden sizes 64, 128, ..., 512");
  }
  return 0;
}

// resident workgroups per CU of the recurrence kernel that fits worst, over EVERY recurrence kernel the schedule may launch for this
// hidden size (fused forward with the hidden-layer input, fused forward from the observation rows, plain forward, backward): in the
// gfx950 code object the fused forward is the largest (252 VGPRs / 86 KB of LDS at H = 256 against 176 / 52 KB for the backward)
template <int H> hipError_t seq_min_blocks_per_cu(int* out) {
  constexpr int threads = 256 * SEQ_UW;
  int n[4] = {0, 0, 0, 0};
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[2], lstm_seq_fwd_kernel<H, SEQ_UW, false, H>, threads, 0);
  if constexpr (H <= SEQ_FUSED_MAX_H) {
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[0], lstm_seq_fwd_kernel<H, SEQ_UW, true, H>, threads, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[1], lstm_seq_fwd_kernel<H, SEQ_UW, true, KBJ_LD_ACTOR>, threads, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[3], lstm_seq_bwd_kernel<H, SEQ_UW>, threads, 0);
    int n16 = 0;   // the 16 x 64-tile backward form: same grid size ((B / 16) x (H / 64) = (B / 32) x (H / 32)), same 512 threads
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, lstm_seq_bwd16_kernel<H>, BWD16_NTH, 0);
    n[3] = std::min(n[3], n16);
  } else {
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[3], lstm_seq_bwd_wide_kernel<H, SEQ_UW>, threads, 0);
    n[0] = n[1] = n[2];
  }
  *out = std::min(std::min(n[0], n[1]), std::min(n[2], n[3]));
  return e;
}

// one LSTM layer step over many rows (kbj_lstm_seq.h lstm_step_kernel): grid = unit groups x row chunks, about one workgroup per CU
int lstm_step(kbj_ctx* ctx, hipStream_t st, int H, const StepArgs& a) {
  const bool obs = a.ldx == KBJ_LD_ACTOR && a.ldw == KBJ_LD_ACTOR;
  KbjKernelTimer timer(st, obs ? KBJ_KIND_LSTM_STEP_OBS : KBJ_KIND_LSTM_STEP, 2.0 * a.M * 4.0 * H * (H + (a.kx ? a.kx : H)));
  const int nug = H / (SEQ_UNITS * 2), nrg = (a.M + SEQ_ROWS - 1) / SEQ_ROWS;
  const int nch = std::max(1, std::min(nrg, 256 / nug));
  dim3 grid(nug * nch), block(512);
  switch (H * 2 + (obs ? 1 : 0)) {
    case 128: hipLaunchKernelGGL((lstm_step_kernel<64, 2>), grid, block, 0, st, a); break;
    case 129: hipLaunchKernelGGL((lstm_step_kernel<64, 2, KBJ_LD_ACTOR>), grid, block, 0, st, a); break;
    case 256: hipLaunchKernelGGL((lstm_step_kernel<128, 2>), grid, block, 0, st, a); break;
    case 257: hipLaunchKernelGGL((lstm_step_kernel<128, 2, KBJ_LD_ACTOR>), grid, block, 0, st, a); break;
    case 384: hipLaunchKernelGGL((lstm_step_kernel<192, 2>), grid, block, 0, st, a); break;
    case 385: hipLaunchKernelGGL((lstm_step_kernel<192, 2, KBJ_LD_ACTOR>), grid, block, 0, st, a); break;
    case 512: hipLaunchKernelGGL((lstm_step_kernel<256, 2>), grid, block, 0, st, a); break;
    case 513: hipLaunchKernelGGL((lstm_step_kernel<256, 2, KBJ_LD_ACTOR>), grid, block, 0, st, a); break;
    default: return kbj_fail(ctx, "LSTM step kernels are built for hidden_size 64, 128, 192, 256");
  }
  return 0;
}

}  // namespace

namespace {
float* sk_workspace(kbj_ctx* ctx, hipStream_t s, size_t floats) {
  NnWs& w = *ws_of(ctx);
  if (!w.sched.deterministic) return nullptr;
  const int l = lane_of(ctx, s);
  if (w.skws_cap[l] < floats) {   // grows during the first calls only; nothing may still be reading the old slab
    hipDeviceSynchronize();
    if (w.skws[l]) hipFree(w.skws[l]);
    w.skws[l] = nullptr; w.skws_cap[l] = 0;
    void* q = nullptr;
    if (hipMalloc(&q, floats * sizeof(float)) != hipSuccess) { kbj_fail(ctx, "hipMalloc (deterministic split-K workspace)"); return nullptr; }
    w.skws[l] = reinterpret_cast<float*>(q); w.skws_cap[l] = floats;
  }
  return w.skws[l];
}
float* det_partials(kbj_ctx* ctx, hipStream_t s) {
  NnWs& w = *ws_of(ctx);
  return w.sched.deterministic ? w.detp[lane_of(ctx, s)] : nullptr;
}
}  // namespace

int kbj_nn_check_errors(kbj_ctx* ctx) {
  NnWs* w = ws_of(ctx);
  if (!w) return 0;
  if (w->seq_stamps) {  // diagnostics: average shader-clock cycles per phase of the actor layer-0 forward recurrence
    std::vector<long long> st((size_t)w->T * 6);
    if (hipMemcpy(st.data(), w->seq_stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess && w->T > 2) {
      double d[6] = {0, 0, 0, 0, 0, 0};
      for (int t = 1; t < w->T - 1; ++t) { for (int k = 0; k < 5; ++k) d[k] += (double)(st[t * 6 + k + 1] - st[t * 6 + k]); d[5] += (double)(st[(t + 1) * 6] - st[t * 6 + 5]); }
      fprintf(stderr, "[kbj seq_fwd stamps, cycles/step] prefetch+wait %.0f stage %.0f mfma %.0f cell %.0f publish %.0f bulk-store %.0f\n",
              d[0] / (w->T - 2), d[1] / (w->T - 2), d[2] / (w->T - 2), d[3] / (w->T - 2), d[4] / (w->T - 2), d[5] / (w->T - 2));
    }
  }
  if (w->seq_bstamps) {
    std::vector<long long> st((size_t)w->T * 10 + 768);
    if (hipMemcpy(st.data(), w->seq_bstamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess && w->T > 3) {
      double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      int cnt = 0;
      for (int t = w->T - 3; t >= 1; --t, ++cnt) { for (int k = 0; k < 8; ++k) d[k] += (double)(st[t * 10 + k + 1] - st[t * 10 + k]); d[8] += (double)(st[(t - 1) * 10] - st[t * 10 + 8]); }
      fprintf(stderr, "[kbj seq_bwd stamps, cycles/step] wait %.0f chunk0-staged %.0f chunk1 %.0f chunk2 %.0f chunk3 %.0f last-mfma+reduce %.0f cell %.0f publish %.0f loop %.0f\n",
              d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt, d[6] / cnt, d[7] / cnt, d[8] / cnt);
      double ticks = (double)(st[1 * 10] - st[(w->T - 3) * 10]), wall = (double)(st[1 * 10 + 9] - st[(w->T - 3) * 10 + 9]);
      int wall_khz = 0;
      hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device);
      if (wall_khz > 0) {   // per-workgroup entry / loop start / exit on the constant-rate clock
        const long long* g = st.data() + (size_t)w->T * 10;
        int nwg = ((w->B + SEQ_ROWS - 1) / SEQ_ROWS) * (w->H / (SEQ_UNITS * SEQ_UW));
        if (nwg > 256) nwg = 256;
        long long e0 = g[0], e1 = g[0], l1 = g[1], x0 = g[2], x1 = g[2];
        for (int i = 1; i < nwg; ++i) { e0 = std::min(e0, g[3 * i]); e1 = std::max(e1, g[3 * i]); l1 = std::max(l1, g[3 * i + 1]); x0 = std::min(x0, g[3 * i + 2]); x1 = std::max(x1, g[3 * i + 2]); }
        const double us = 1e3 / wall_khz;
        fprintf(stderr, "[kbj seq_bwd stamps] %d workgroups: last entry +%.1f us, last loop start +%.1f us, first exit +%.1f us, last exit +%.1f us (from the first entry)\n", nwg,
                (e1 - e0) * us, (l1 - e0) * us, (x0 - e0) * us, (x1 - e0) * us);
      }
      if (wall > 0 && wall_khz > 0) fprintf(stderr, "[kbj seq_bwd stamps] %.2f us/step, stamp counter at %.0f MHz\n", wall / wall_khz * 1e3 / cnt, ticks / wall * wall_khz * 1e-3);
    }
  }
  unsigned e[2] = {0, 0};
  if (hipMemcpy(e, w->seq_err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return kbj_fail(ctx, "hipMemcpy seq_err");
  if (e[0] || e[1]) {   // sticky until acknowledged here: clear, so the context stays usable (and destroyable) after the report
    hipMemset(w->seq_err, 0, sizeof(e));
    if (e[0]) return kbj_fail(ctx, "persistent LSTM kernel: inter-workgroup wait timed out (grid not fully resident?); the optimizer steps fed by it were skipped");
    return kbj_fail(ctx, "non-finite gradient norm: the optimizer step was skipped (parameters and moments unchanged)");
  }
  return 0;
}

// A pending kbj_ppo_prefetch is tied to the CONTENTS of the trajectory and of the index array it was given: every entry point that writes
// trajectory arrays (rollout, policy / env steps, GAE, rewards) calls this first - the stale gathers are ordered in front of the writer (they
// only read the trajectory, but their workspace rows must not be consumed) and the marker is cleared, so the next kbj_ppo_grad gathers afresh.
void kbj_nn_drop_prefetch(kbj_ctx* ctx) {
  NnWs* w = ws_of(ctx);
  if (!w || !w->prefetched_idx) return;
  hipStreamWaitEvent(ctx->stream, ctx->ev_prefetch, 0);
  w->prefetched_idx = nullptr; w->prefetched_traj = nullptr;
}

int kbj_nn_create(kbj_ctx* ctx) {
  NnWs* w = new NnWs();
  ctx->nn_ws = w;
  const kbj_config& c = ctx->cfg_h;
  w->Hu = c.hidden_size; w->H = (c.hidden_size + 63) / 64 * 64; w->N = c.num_envs; w->B = c.batch_size; w->T = c.rollout_len;
  if (w->B <= 0 || w->B > w->N) return kbj_fail(ctx, "kbj_create: batch_size must be in [1, num_envs]");
  layout_params(*w, w->H, ctx->cfg_h.depth, ctx->cfg_h.extra_obs_actor, ctx->cfg_h.extra_obs_critic);
  size_t N = w->N, H = w->H, B = w->B, T = w->T;
  w->unparams = w->nparams; w->unactor = w->nactor;
  if (w->padded()) {
    NnWs u;
    layout_params(u, w->Hu, c.depth, c.extra_obs_actor, c.extra_obs_critic);
    w->unparams = u.nparams; w->unactor = u.nactor;
    std::vector<PadDesc> pd;
    const int Hu = w->Hu, Hi = w->H;
    for (int n = 0; n < 2; ++n) {
      const NetOff &a = u.net[n], &b = w->net[n];
      pd.push_back(PadDesc{a.w_in, b.w_in, 1, Hu, Hi, a.nin, a.nin});
      pd.push_back(PadDesc{a.b_in, b.b_in, 1, Hu, Hi, 1, 1});
      for (int l = 0; l < w->D; ++l) {
        pd.push_back(PadDesc{a.w_ih[l], b.w_ih[l], 4, Hu, Hi, Hu, Hi});
        pd.push_back(PadDesc{a.w_hh[l], b.w_hh[l], 4, Hu, Hi, Hu, Hi});
        pd.push_back(PadDesc{a.b[l], b.b[l], 4, Hu, Hi, 1, 1});
      }
      pd.push_back(PadDesc{a.w_out, b.w_out, 1, a.nout, a.nout, Hu, Hi});
      pd.push_back(PadDesc{a.b_out, b.b_out, 1, a.nout, a.nout, 1, 1});
    }
    w->npad = (int)pd.size();
    if (dalloc(ctx, *w, &w->pad_desc, pd.size())) return -1;
    if (hipMemcpy(w->pad_desc, pd.data(), pd.size() * sizeof(PadDesc), hipMemcpyHostToDevice) != hipSuccess) return kbj_fail(ctx, "hipMemcpy pad descriptors");
    if (dalloc(ctx, *w, &w->pparams, w->nparams) || dalloc(ctx, *w, &w->pgrad, w->nparams)) return -1;
    const bool mir = c.actor_mirror_loss_scale != 0.0f || c.critic_mirror_loss_scale != 0.0f;
    for (int k = 0; k < (mir ? 4 : 2); ++k)
      if (dalloc(ctx, *w, &w->phc[k], (size_t)2 * w->D * N * H) || dalloc(ctx, *w, &w->pc0[k], (size_t)2 * w->D * N * H)) return -1;
  }
  w->mirror = c.actor_mirror_loss_scale != 0.0f || c.critic_mirror_loss_scale != 0.0f;
  w->nnets = w->mirror ? 4 : 2;
  for (int n = 0; n < w->nnets; ++n) {
    if (dalloc(ctx, *w, &w->rX[n], N * H)) return -1;
    if (dalloc(ctx, *w, &w->rG[n], N * 4 * H)) return -1;
    if (dalloc(ctx, *w, &w->rOut[n], N * 40)) return -1;
    for (int l = 0; l < w->D; ++l) if (dalloc(ctx, *w, &w->rH[n][l], N * H)) return -1;
  }
  if (dalloc(ctx, *w, &w->joint_bias_d, KBJ_NU)) return -1;
  if (hipMemcpy(w->joint_bias_d, ctx->model_h.joint_bias, KBJ_NU * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return kbj_fail(ctx, "hipMemcpy joint_bias");
  size_t R = T * B;
  if (w->mirror) {
    std::vector<MirrorEntry> ta, tc;
    build_mirror_tables(ctx->model_h, ta, tc, ctx->cfg_h.extra_obs_actor, ctx->cfg_h.extra_obs_critic);
    if (dalloc(ctx, *w, &w->mtab[0], ta.size()) || dalloc(ctx, *w, &w->mtab[1], tc.size())) return -1;
    if (hipMemcpy(w->mtab[0], ta.data(), ta.size() * sizeof(MirrorEntry), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(w->mtab[1], tc.data(), tc.size() * sizeof(MirrorEntry), hipMemcpyHostToDevice) != hipSuccess) return kbj_fail(ctx, "hipMemcpy mirror tables");
    if (dalloc(ctx, *w, &w->rObsM[0], N * w->net[0].ld_obs) || dalloc(ctx, *w, &w->rObsM[1], N * w->net[1].ld_obs)) return -1;
    float** rs[] = {&w->value_m, &w->dvalue_m, &w->zeroR};
    for (float** p : rs) if (dalloc(ctx, *w, p, R)) return -1;
    float** r20[] = {&w->y_m, &w->sd_m, &w->dy, &w->dy_m};
    for (float** p : r20) if (dalloc(ctx, *w, p, R * KBJ_NU)) return -1;
    if (dalloc(ctx, *w, &w->lpf0_m, B * KBJ_NU)) return -1;
    if (hipMemset(w->zeroR, 0, R * sizeof(float)) != hipSuccess) return kbj_fail(ctx, "hipMemset zeroR");
  }
  for (int n = 0; n < w->nnets; ++n) {
    TrainBufs& t = w->tb[n];
    if (dalloc(ctx, *w, &t.obs, R * w->net[n & 1].ld_obs)) return -1;
    if (dalloc(ctx, *w, &t.X0, R * H)) return -1;
    for (int l = 0; l < w->D; ++l) {
      if (dalloc(ctx, *w, &t.dGl[l], R * 4 * H)) return -1;
      if (dalloc(ctx, *w, &t.G[l], R * 4 * H)) return -1;
      if (dalloc(ctx, *w, &t.Hm[l], (T + 1) * B * H)) return -1;
      if (dalloc(ctx, *w, &t.Hout[l], R * H)) return -1;
      if (dalloc(ctx, *w, &t.Cm[l], (T + 1) * B * H)) return -1;
      if (dalloc(ctx, *w, &t.TanhC[l], R * H)) return -1;
    }
    if (dalloc(ctx, *w, &t.Out, R * 40)) return -1;
    if (dalloc(ctx, *w, &t.dOut, R * 40)) return -1;
    if (dalloc(ctx, *w, &t.dHa, R * H)) return -1;
    if (dalloc(ctx, *w, &t.dHb, R * H)) return -1;
  }
  float** small[] = {&w->keep, &w->logp_old, &w->val_old, &w->adv, &w->target, &w->logp, &w->ent, &w->value, &w->dlogp, &w->dvalue};
  for (float** p : small) if (dalloc(ctx, *w, p, R)) return -1;
  if (dalloc(ctx, *w, &w->act, R * KBJ_NU)) return -1;
  if (dalloc(ctx, *w, &w->y, R * KBJ_NU)) return -1;
  if (dalloc(ctx, *w, &w->sd, R * KBJ_NU)) return -1;
  if (dalloc(ctx, *w, &w->lpf0, B * KBJ_NU)) return -1;
  if (dalloc(ctx, *w, &w->stats, 16)) return -1;
  if (dalloc(ctx, *w, &w->Weff, (size_t)4 * H * w->net[0].ld_obs) || dalloc(ctx, *w, &w->beff, 4 * H)) return -1;
  for (int n = 0; n < w->nnets; ++n) if (dalloc(ctx, *w, &w->Zeff[n], 4 * H * w->net[n & 1].ld_obs)) return -1;
  for (int k = 0; k < 2; ++k) if (dalloc(ctx, *w, &w->WinP[k], H * w->net[k].ld_obs)) return -1;
  if (hipMemset(w->Weff, 0, (size_t)4 * H * w->net[0].ld_obs * sizeof(float)) != hipSuccess) return kbj_fail(ctx, "hipMemset Weff");
  if (dalloc(ctx, *w, &w->seq_counters, SEQ_COUNTER_TOTAL)) return -1;   // [phase: forward layer l = l, backward layer l = D + l][net][row group x unit group]
  if (dalloc(ctx, *w, &w->seq_err, 4)) return -1;
  if (getenv("KBJ_SEQ_STAMPS")) { if (dalloc(ctx, *w, &w->seq_stamps, (size_t)T * 6)) return -1; }
  if (getenv("KBJ_SEQ_BSTAMPS")) { if (dalloc(ctx, *w, &w->seq_bstamps, (size_t)T * 10 + 768)) return -1; }
  if (hipMemset(w->seq_err, 0, 4 * sizeof(unsigned)) != hipSuccess) return kbj_fail(ctx, "hipMemset seq_err");
  {
    Sched& sc = w->sched;
    sc.fold_actor = env_flag("KBJ_FOLD_ACTOR", true); sc.fold_critic = env_flag("KBJ_FOLD_CRITIC", true);
    sc.fuse_ih = env_flag("KBJ_SEQ_FUSE", true); sc.fuse_obs = sc.fuse_ih && env_flag("KBJ_SEQ_FUSE_OBS", true);
    sc.fused_critic_head = env_flag("KBJ_FUSED_CRITIC_HEAD", true); sc.rollout_step = env_flag("KBJ_ROLLOUT_STEP", true);
    sc.one_stream = env_flag("KBJ_ONE_STREAM", false); sc.debug_sync = env_flag("KBJ_DEBUG", false);
    sc.deterministic = c.deterministic != 0 || env_flag("KBJ_DETERMINISTIC", false);
    { const char* cl = getenv("KBJ_CRITIC_LANE"); sc.critic_on_caller = !(cl && std::string(cl) == "2nd"); }
    sc.bwd16 = env_flag("KBJ_BWD16", true) && H <= (size_t)SEQ_FUSED_MAX_H;   // wide layers keep lstm_seq_bwd_wide_kernel
    sc.gemm_x3 = (c.gemm_bf16x3 != 0 || env_flag("KBJ_GEMM_X3", false)) && !sc.deterministic;   // (the deterministic split-K slabs stay on the exact kernel)
    if (H > (size_t)SEQ_FUSED_MAX_H) sc.fuse_ih = sc.fuse_obs = sc.rollout_step = false;   // wide layers (SEQ_FUSED_MAX_H above)
    if (sc.deterministic) {
      for (int l = 0; l < 4; ++l) if (dalloc(ctx, *w, &w->detp[l], (size_t)DETP_ROWS * DETP_COLS)) return -1;
      if (dalloc(ctx, *w, &w->detd, 2 * 512)) return -1;
    }
  }
  g_seq_drop = getenv("KBJ_DEBUG_DROP_SEQ_WG") ? atoi(getenv("KBJ_DEBUG_DROP_SEQ_WG")) : 0;
  g_seq_drop_bwd = getenv("KBJ_DEBUG_DROP_SEQ_BWD_WG") ? atoi(getenv("KBJ_DEBUG_DROP_SEQ_BWD_WG")) : 0;
  {   // wall-clock bound of the recurrences' inter-workgroup waits (kbj_lstm_seq.h seq_wait): 2 s by default, KBJ_SEQ_TIMEOUT_MS=n overrides,
      // 20 ms under fault injection
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    long ms = (g_seq_drop > 0 || g_seq_drop_bwd > 0) ? 20 : (long)SEQ_TIMEOUT_MS;
    if (getenv("KBJ_SEQ_TIMEOUT_MS")) ms = std::max(1, std::min(30000, atoi(getenv("KBJ_SEQ_TIMEOUT_MS"))));
    g_seq_timeout_ticks = (unsigned)std::min<long long>(0xFFFFFFFFll, (long long)ms * wall_khz);
  }
  // Residency of the persistent recurrences: the workgroups of one launch spin on each other, and kbj_ppo_grad keeps TWO launches
  // (actor-type and critic-type net, one per stream; the mirror branches queue behind them on the same two streams) in flight, so
  // 2 x grid workgroups must be resident at once. Every other kernel of the schedule (GEMMs, heads) terminates on its own, so it can
  // only delay a recurrence workgroup, never starve it. Slots = the occupancy query's answer for the WORST-fitting recurrence kernel
  // of this hidden size. Independently, a launch owns SEQ_COUNTER_WORDS hand-off words (one per workgroup): a larger grid would write
  // into the next launch's block, so it is refused whatever the occupancy says.
  {
    const int grid = (int)((B + SEQ_ROWS - 1) / SEQ_ROWS) * (int)(H / (SEQ_UNITS * SEQ_UW));
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return kbj_fail(ctx, "hipGetDeviceProperties");
    cus = prop.multiProcessorCount;
    hipError_t oe = hipSuccess;
    switch ((int)H) {
      case 64: oe = seq_min_blocks_per_cu<64>(&per_cu); break;
      case 128: oe = seq_min_blocks_per_cu<128>(&per_cu); break;
      case 192: oe = seq_min_blocks_per_cu<192>(&per_cu); break;
      case 256: oe = seq_min_blocks_per_cu<256>(&per_cu); break;
      case 320: oe = seq_min_blocks_per_cu<320>(&per_cu); break;
      case 384: oe = seq_min_blocks_per_cu<384>(&per_cu); break;
      case 448: oe = seq_min_blocks_per_cu<448>(&per_cu); break;
      case 512: oe = seq_min_blocks_per_cu<512>(&per_cu); break;
      default: return kbj_fail(ctx, "kbj_create: hidden_size above 512 (persistent LSTM kernels)");
    }
    if (oe != hipSuccess || per_cu < 1) return kbj_fail(ctx, "kbj_create: occupancy query of the persistent LSTM kernels failed");
    const long slots = (long)per_cu * cus;
    w->seq_grid = grid; w->seq_slots = (int)slots;
    char msg[320];
    if (grid > SEQ_COUNTER_WORDS) {
      snprintf(msg, sizeof(msg), "kbj_create: a persistent LSTM launch would need %d workgroups (batch_size / 32 x hidden_size / 32), the hand-off "
               "counters hold %d per launch: lower batch_size", grid, SEQ_COUNTER_WORDS);
      return kbj_fail(ctx, msg);
    }
    if (H > SEQ_FUSED_MAX_H && 2L * grid > slots) w->sched.one_stream = true;   // wide layers: one recurrence at a time
    if ((w->sched.one_stream ? 1L : 2L) * grid > slots) {
      snprintf(msg, sizeof(msg), "kbj_create: %s persistent LSTM launches need %ld resident workgroups, the device holds %ld "
               "(%d per CU x %d CUs): lower batch_size", w->sched.one_stream ? "the" : "two concurrent", (w->sched.one_stream ? 1L : 2L) * grid, slots, per_cu, cus);
      return kbj_fail(ctx, msg);
    }
  }
  return 0;
}

void kbj_nn_destroy(kbj_ctx* ctx) {
  NnWs* w = ws_of(ctx);
  if (!w) return;
  for (void* p : w->allocs) hipFree(p);
  for (int l = 0; l < 4; ++l) if (w->skws[l]) hipFree(w->skws[l]);
  delete w;
  ctx->nn_ws = nullptr;
}


int kbj_env_step_range(kbj_ctx* ctx, hipStream_t s, int env0, int count, const float* action_d, float* aux_t_d, float* actor_next_d, float* critic_next_d,
                       float* aux_next_d, float* qstate_t_d);   // kbj_env.hip

namespace {

// One control step of the nets [net_lo, net_hi) (0 actor, 1 critic, 2/3 their mirror branches) for the env rows [n0, n0 + cnt) on
// stream s. All pointers are those of env row 0.
// Rollout-time layer steps run in lstm_step_kernel (fused [x | h] product + cell). Its output h cannot alias its input, so every h
// plane of the carry has a ping-pong partner in the workspace: a call with parity 0 reads the caller's planes and writes the partners,
// parity 1 the other way round. kbj_rollout alternates per control step; the caller's arrays hold the state again when it returns.
// KBJ_ROLLOUT_STEP=0 (Sched::rollout_step): the previous form, one GEMM over [x | h] plus a cell kernel per layer, in place.
// Only the actor proper (net 0) sits on the rollout's critical chain (env -> actor -> env). The critic and the mirror branches run on side
// lanes under the env kernel, where the only free resources are those of its last, partial round of wavefronts (8192 envs = 2.67 rounds
// of 12 per CU: 168 VGPRs per SIMD and ~58 KB of LDS per CU for the last third of the kernel): a 64x64-tile GEMM workgroup (4 waves x 84
// VGPRs, 37 KB) fits there, a layer-step workgroup (8 waves x 220 VGPRs) does not and would run into the next actor chain instead. So the
// side-lane nets keep the GEMM + cell form on small tiles (414 vs 418 ms per iteration).
bool net_uses_step_kernel(const NnWs& w, int n) { return w.sched.rollout_step && n == 0; }
float* h_plane(NnWs& w, float* hc, int n, int l, int n0, bool partner) {
  if (!net_uses_step_kernel(w, n)) partner = false;
  return (partner ? w.rH[n][l] : hc + (size_t)(2 * l) * w.N * w.H) + (size_t)n0 * w.H;
}

// The actor's input projection (65 -> H, no activation) feeds only layer 0's input product, so gates_0 = obs (W_ih0 W_in)^T +
// (W_ih0 b_in + b_0), as in kbj_ppo_grad: prepares w.Weff / w.beff for these parameters and returns w.Weff (null: not folded)
const float* fold_actor_weights(kbj_ctx* ctx, hipStream_t s, const float* params_d) {
  NnWs& w = *ws_of(ctx);
  const int H = w.H;
  if (!w.sched.rollout_step || w.net[0].ld_obs != KBJ_LD_ACTOR || !w.sched.fold_actor) return nullptr;
  const NetOff& oa = w.net[0];
  GemmArgs g{params_d + oa.w_ih[0], params_d + oa.w_in, w.Weff, nullptr, 4 * H, oa.nin, H, H, oa.nin, oa.ld_obs, 0, 1, nullptr};
  gemm_launch<true, false>(s, g);
  hipLaunchKernelGGL(matvec_kernel, dim3((4 * H + 3) / 4), dim3(256), 0, s, params_d + oa.w_ih[0], params_d + oa.b_in, params_d + oa.b[0], 4 * H, H, w.beff);
  return w.Weff;
}

// Rollout-time input projections read W_in [H][nin] with nin = 65 / 475 floats per row: no 16-byte alignment, which puts the GEMM on its element-wise
// load path for that operand (the critic's 8192 x 256 x 475 product ran at 27 TFLOP/s, with four times the load instructions - beside an env kernel
// that is bound by vector issue). As in the update (ppo_forward_nets): a re-pitched copy with the observation rows' stride, zeros behind column nin,
// made once per rollout / policy step; the contraction then runs over the padded width (the rows' padding columns hold zeros: host/buffers.py).
void repitch_input_weights(kbj_ctx* ctx, hipStream_t s, const float* params_d) {
  NnWs& w = *ws_of(ctx);
  for (int k = 0; k < 2; ++k) {
    const NetOff& o = w.net[k];
    if (o.nin == o.ld_obs) continue;
    hipLaunchKernelGGL(repitch_rows_kernel, g1((size_t)w.H * o.ld_obs), dim3(256), 0, s, params_d + o.w_in, w.H, o.nin, o.ld_obs, w.WinP[k]);
  }
}

// weff: the folded actor input weights (W_ih0 W_in, bias in w.beff) when the caller has prepared them for these parameters, else null
// (the caller has also run repitch_input_weights for these parameters)
int policy_nets(kbj_ctx* ctx, hipStream_t s, const float* params_d, int net_lo, int net_hi, int n0, int cnt, const float* actor_obs_d, const float* critic_obs_d,
                kbj_carry* carry, uint32_t seed, uint32_t step_index, int argmax, float* action_d, float* logp_d, float* value_d, int parity, const float* weff) {
  NnWs& w = *ws_of(ctx);
  const kbj_config& c = ctx->cfg_h;
  const int N = w.N, H = w.H;
  const bool fused_any = w.sched.rollout_step;
  const float* obs_base[2] = {actor_obs_d, critic_obs_d};
  float* hc[4] = {carry->actor_hc_d, carry->critic_hc_d, carry->actor_mirror_hc_d, carry->critic_mirror_hc_d};
  HeadParams hp{c.min_std, c.max_std, c.var_scale, c.lpf_alpha, w.net[0].ld_obs};
  for (int n = net_lo; n < net_hi; ++n) {
    const int k = n & 1;
    const NetOff& o = w.net[k];
    const float* obs = obs_base[k] + (size_t)n0 * o.ld_obs;
    float* obs_m = n >= 2 ? w.rObsM[k] + (size_t)n0 * o.ld_obs : nullptr;
    // scratch rows are private to (net, env row): lanes and side lanes never share them
    float* X = w.rX[n] + (size_t)n0 * H;
    float* G = w.rG[n] + (size_t)n0 * 4 * H;
    float* Out = w.rOut[n] + (size_t)n0 * 40;
    if (n >= 2) {  // mirror branches advance their own carries on the mirrored observations (train.py:1463-1481, 1555-1560)
      hipLaunchKernelGGL(mirror_rows_kernel, g1((size_t)cnt * o.ld_obs), dim3(256), 0, s, obs, obs_m, (size_t)cnt, o.ld_obs, w.mtab[k]);
      obs = obs_m;
    }
    const bool fused = fused_any && net_uses_step_kernel(w, n);
    const bool folded = fused && weff && k == 0 && o.ld_obs == KBJ_LD_ACTOR;   // actor-type net: layer-0 gates straight from the observation row
    if (!folded) {
      if (o.nin != o.ld_obs) linear_fwd(s, obs, o.ld_obs, w.WinP[k], o.ld_obs, params_d + o.b_in, X, H, cnt, H, o.ld_obs, 0);   // aligned copy of W_in (repitch_input_weights)
      else linear_fwd(s, obs, o.ld_obs, params_d + o.w_in, o.nin, params_d + o.b_in, X, H, cnt, H, o.nin, 0);
    }
    const float* x = X;
    for (int l = 0; l < w.D; ++l) {
      float* cc = hc[n] + (size_t)(2 * l + 1) * N * H + (size_t)n0 * H;
      if (fused) {
        const float* h_in = h_plane(w, hc[n], n, l, n0, parity != 0);
        float* h_out = h_plane(w, hc[n], n, l, n0, parity == 0);
        StepArgs sa{x, H, 0, params_d + o.w_ih[l], H, params_d + o.w_hh[l], params_d + o.b[l], h_in, h_out, cc, cnt};
        if (l == 0 && folded) { sa.X = obs; sa.ldx = o.ld_obs; sa.kx = o.nin; sa.Wih = weff; sa.ldw = o.ld_obs; sa.bias = w.beff; }
        if (lstm_step(ctx, s, H, sa)) return -1;
        x = h_out;
        continue;
      }
      float* h = hc[n] + (size_t)(2 * l) * N * H + (size_t)n0 * H;
      {  // gates = [x | h] [W_ih | W_hh]^T + b as ONE launch over the concatenated contraction (no read-modify-write of G)
        GemmArgs g{x, params_d + o.w_ih[l], G, params_d + o.b[l], cnt, 4 * H, 2 * H, H, H, 4 * H, 0, 1, nullptr};
        g.A2 = h; g.B2 = params_d + o.w_hh[l]; g.k1 = H;
        g.x3 = w.sched.gemm_x3 ? 1 : 0;
        gemm_launch<true, true>(s, g, (n > 0 && w.sched.rollout_step) ? 0 : -1);   // side-lane nets: 64x64 tiles (see net_uses_step_kernel)
      }
      CellFwdArgs2 ca;
      ca.a[0] = CellFwdArgs{G, cc, h, cc, nullptr, nullptr, nullptr, nullptr, cnt, H};
      hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((cnt * H + 255) / 256, 1), dim3(256), 0, s, ca);
      x = h;
    }
    if (n == 3) continue;  // the mirrored critic's value is only needed under the gradient; at rollout time only its carry advances
    if (n == 0) {   // output projection + low-pass + sample + log-prob as one launch (the chain env -> actor -> env waits for it)
      hipLaunchKernelGGL(actor_head_fused_kernel, dim3((cnt + HEAD_ENVS * HEAD_WAVES - 1) / (HEAD_ENVS * HEAD_WAVES)), dim3(64 * HEAD_WAVES), 0, s, x, H,
                         params_d + o.w_out, params_d + o.b_out, obs, carry->lpf_d + (size_t)n0 * KBJ_NU, w.joint_bias_d, hp, seed, (uint32_t)(c.env_id_offset + n0),
                         step_index, argmax, cnt, action_d + (size_t)n0 * KBJ_NU, logp_d + n0);
      continue;
    }
    if (n == 1 && o.nout == 1) {
      hipLaunchKernelGGL(critic_value_fused_kernel, g1((size_t)cnt * 32), dim3(256), 0, s, x, H, params_d + o.w_out, params_d + o.b_out, cnt, value_d + n0);
      continue;
    }
    // n == 2: the mirrored actor only advances its low-pass state (no sample)
    linear_fwd(s, x, H, params_d + o.w_out, H, params_d + o.b_out, Out, 40, cnt, o.nout, H, 0);
    hipLaunchKernelGGL(actor_head_lpf_kernel, g1((size_t)cnt * KBJ_NU), dim3(256), 0, s, Out, obs, carry->lpf_mirror_d + (size_t)n0 * KBJ_NU, w.joint_bias_d, c.lpf_alpha, cnt, w.net[0].ld_obs);
  }
  return 0;
}

// carry <- 0 where done, for the nets [net_lo, net_hi) and env rows [n0, n0 + cnt); parity as policy_nets: which copy of the h planes is live
void carry_reset_nets(kbj_ctx* ctx, hipStream_t s, int net_lo, int net_hi, int n0, int cnt, kbj_carry* carry, const float* done_d, int done_stride, int parity) {
  NnWs& w = *ws_of(ctx);
  float* hc[4] = {carry->actor_hc_d, carry->critic_hc_d, carry->actor_mirror_hc_d, carry->critic_mirror_hc_d};
  float* lpf[4] = {carry->lpf_d, nullptr, carry->lpf_mirror_d, nullptr};
  const bool partner = parity != 0 && w.sched.rollout_step;
  for (int k = net_lo; k < net_hi; ++k) {
    CarryPlanes cp;
    cp.n = 2 * w.D;
    for (int l = 0; l < w.D; ++l) {
      cp.p[2 * l] = h_plane(w, hc[k], k, l, n0, partner);
      cp.p[2 * l + 1] = hc[k] + (size_t)(2 * l + 1) * w.N * w.H + (size_t)n0 * w.H;
    }
    hipLaunchKernelGGL(carry_reset_kernel, dim3((cnt + 3) / 4), dim3(256), 0, s, cp, cnt, w.H, lpf[k] ? lpf[k] + (size_t)n0 * KBJ_NU : nullptr, done_d + (size_t)n0 * done_stride, done_stride);
  }
}

// the h planes of nets [net_lo, net_hi) back from their ping-pong partners into the caller's arrays
int carry_h_home(kbj_ctx* ctx, hipStream_t s, int net_lo, int net_hi, kbj_carry* carry) {
  NnWs& w = *ws_of(ctx);
  float* hc[4] = {carry->actor_hc_d, carry->critic_hc_d, carry->actor_mirror_hc_d, carry->critic_mirror_hc_d};
  for (int k = net_lo; k < net_hi; ++k)
    for (int l = 0; l < w.D && net_uses_step_kernel(w, k); ++l)
      KBJ_HIP(ctx, hipMemcpyAsync(hc[k] + (size_t)(2 * l) * w.N * w.H, w.rH[k][l], (size_t)w.N * w.H * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- free hidden_size: conversions at the ABI boundary of a padded context (NnWs::Hu != NnWs::H) ----
void pad_params(kbj_ctx* ctx, hipStream_t s, const float* user, float* internal) {
  NnWs& w = *ws_of(ctx);
  hipLaunchKernelGGL(pad_params_kernel, g1(w.nparams), dim3(256), 0, s, w.pad_desc, w.npad, const_cast<float*>(user), internal, w.nparams, 0);
}
void unpad_params(kbj_ctx* ctx, hipStream_t s, const float* internal, float* user) {
  NnWs& w = *ws_of(ctx);
  hipLaunchKernelGGL(pad_params_kernel, g1(w.nparams), dim3(256), 0, s, w.pad_desc, w.npad, user, const_cast<float*>(internal), w.nparams, 1);
}
// carries [D][2][N][width]: ws -> wd columns per row
void repitch_hc(kbj_ctx* ctx, hipStream_t s, const float* src, float* dst, int ws, int wd) {
  NnWs& w = *ws_of(ctx);
  const size_t rows = (size_t)2 * w.D * w.N;
  hipLaunchKernelGGL(repitch_pad_kernel, g1(rows * wd), dim3(256), 0, s, src, dst, rows, ws, wd);
}
// the caller's carry as an H-wide internal one (lpf state is not hidden-size dependent: shared)
kbj_carry pad_carry(kbj_ctx* ctx, hipStream_t s, const kbj_carry& c) {
  NnWs& w = *ws_of(ctx);
  kbj_carry p = c;
  float* const* src[4] = {&c.actor_hc_d, &c.critic_hc_d, &c.actor_mirror_hc_d, &c.critic_mirror_hc_d};
  float** dst[4] = {&p.actor_hc_d, &p.critic_hc_d, &p.actor_mirror_hc_d, &p.critic_mirror_hc_d};
  for (int k = 0; k < 4; ++k)
    if (*src[k] && w.phc[k]) { repitch_hc(ctx, s, *src[k], w.phc[k], w.Hu, w.H); *dst[k] = w.phc[k]; }
  return p;
}
void unpad_carry(kbj_ctx* ctx, hipStream_t s, const kbj_carry& c) {
  NnWs& w = *ws_of(ctx);
  float* user[4] = {c.actor_hc_d, c.critic_hc_d, c.actor_mirror_hc_d, c.critic_mirror_hc_d};
  for (int k = 0; k < 4; ++k)
    if (user[k] && w.phc[k]) repitch_hc(ctx, s, w.phc[k], user[k], w.H, w.Hu);
}
// the trajectory with its start carries as H-wide internal copies
kbj_traj pad_traj_carry0(kbj_ctx* ctx, hipStream_t s, const kbj_traj& tr) {
  NnWs& w = *ws_of(ctx);
  kbj_traj p = tr;
  float* const* src[4] = {&tr.carry0_actor_hc_d, &tr.carry0_critic_hc_d, &tr.carry0_actor_mirror_hc_d, &tr.carry0_critic_mirror_hc_d};
  float** dst[4] = {&p.carry0_actor_hc_d, &p.carry0_critic_hc_d, &p.carry0_actor_mirror_hc_d, &p.carry0_critic_mirror_hc_d};
  for (int k = 0; k < 4; ++k)
    if (*src[k] && w.pc0[k]) { repitch_hc(ctx, s, *src[k], w.pc0[k], w.Hu, w.H); *dst[k] = w.pc0[k]; }
  return p;
}
struct InnerScope { NnWs& w; explicit InnerScope(NnWs& ws) : w(ws) { w.inner = true; } ~InnerScope() { w.inner = false; } };

}  // namespace

extern "C" {

int kbj_mirror_table(const void* model_blob, size_t model_bytes, int critic, int32_t* src_h, float* mul_h, float* add_h) {
  if (!model_blob || !src_h || !mul_h || !add_h || model_bytes != sizeof(kbj_model)) return kbj_fail(nullptr, "kbj_mirror_table: bad argument");
  std::vector<MirrorEntry> ta, tc;
  build_mirror_tables(*reinterpret_cast<const kbj_model*>(model_blob), ta, tc);
  const std::vector<MirrorEntry>& t = critic ? tc : ta;
  for (size_t k = 0; k < t.size(); ++k) { src_h[k] = t[k].src; mul_h[k] = t[k].mul; add_h[k] = t[k].add; }
  return (int)t.size();
}

size_t kbj_param_count(const kbj_config* cfg) { NnWs w; layout_params(w, cfg->hidden_size, cfg->depth, cfg->extra_obs_actor, cfg->extra_obs_critic); return w.nparams; }
size_t kbj_actor_param_count(const kbj_config* cfg) { NnWs w; layout_params(w, cfg->hidden_size, cfg->depth, cfg->extra_obs_actor, cfg->extra_obs_critic); return w.nactor; }

int kbj_init_params(kbj_ctx* ctx, uint32_t seed, float* params_d) {
  if (!ctx || !params_d) return kbj_fail(ctx, "kbj_init_params: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  NnWs u;                                // the caller's layout: fan-ins and offsets follow its hidden_size, not the kernels' padded one
  layout_params(u, w.Hu, w.D, ctx->cfg_h.extra_obs_actor, ctx->cfg_h.extra_obs_critic);
  int H = w.Hu;
  uint32_t leaf = 0;
  auto fill = [&](size_t off, size_t n, int fan_in) {
    hipLaunchKernelGGL(init_uniform_kernel, g1(n), dim3(256), 0, ctx->stream, params_d + off, n, 1.0f / std::sqrt((float)fan_in), seed, leaf++);
  };
  for (int n = 0; n < 2; ++n) {
    const NetOff& o = u.net[n];
    fill(o.w_in, (size_t)H * o.nin, o.nin); fill(o.b_in, H, o.nin);
    for (int l = 0; l < w.D; ++l) { fill(o.w_ih[l], (size_t)4 * H * H, H); fill(o.w_hh[l], (size_t)4 * H * H, H); fill(o.b[l], (size_t)4 * H, H); }
    fill(o.w_out, (size_t)o.nout * H, H); fill(o.b_out, o.nout, H);
  }
  KBJ_CHECK_LAUNCH(ctx, "init_uniform_kernel");
  return 0;
}

int kbj_policy_step(kbj_ctx* ctx, const float* params_d, const float* actor_obs_d, const float* critic_obs_d, kbj_carry* carry, uint32_t seed,
                    uint32_t step_index, int argmax, float* action_d, float* logp_d, float* value_d) {
  if (!ctx || !params_d || !actor_obs_d || !critic_obs_d || !carry || !action_d || !logp_d || !value_d) return kbj_fail(ctx, "kbj_policy_step: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  if (w.mirror && (!carry->actor_mirror_hc_d || !carry->critic_mirror_hc_d || !carry->lpf_mirror_d))
    return kbj_fail(ctx, "kbj_policy_step: the mirror losses are enabled, the carry needs the mirror-branch arrays");
  if (w.padded() && !w.inner) {
    InnerScope in(w);
    pad_params(ctx, ctx->stream, params_d, w.pparams);
    kbj_carry pc = pad_carry(ctx, ctx->stream, *carry);
    const int rc = kbj_policy_step(ctx, w.pparams, actor_obs_d, critic_obs_d, &pc, seed, step_index, argmax, action_d, logp_d, value_d);
    unpad_carry(ctx, ctx->stream, *carry);
    return rc;
  }
  kbj_nn_drop_prefetch(ctx);   // action / logp / value may be trajectory rows
  KbjTimed timed(ctx, true);
  repitch_input_weights(ctx, ctx->stream, params_d);
  if (policy_nets(ctx, ctx->stream, params_d, 0, w.nnets, 0, w.N, actor_obs_d, critic_obs_d, carry, seed, step_index, argmax, action_d, logp_d, value_d, 0,
                  fold_actor_weights(ctx, ctx->stream, params_d))) return -1;
  if (w.sched.rollout_step && carry_h_home(ctx, ctx->stream, 0, w.nnets, carry)) return -1;   // the new h sits in the partners: bring it home
  KBJ_CHECK_LAUNCH(ctx, "kbj_policy_step");
  return 0;
}

int kbj_carry_reset(kbj_ctx* ctx, kbj_carry* carry, const float* done_d, int done_stride) {
  if (!ctx || !carry || !done_d) return kbj_fail(ctx, "kbj_carry_reset: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  if (w.mirror && (!carry->actor_mirror_hc_d || !carry->critic_mirror_hc_d || !carry->lpf_mirror_d)) return kbj_fail(ctx, "kbj_carry_reset: mirror-branch carry arrays are NULL");
  if (w.padded() && !w.inner) {
    InnerScope in(w);
    kbj_carry pc = pad_carry(ctx, ctx->stream, *carry);
    const int rc = kbj_carry_reset(ctx, &pc, done_d, done_stride);
    unpad_carry(ctx, ctx->stream, *carry);
    return rc;
  }
  carry_reset_nets(ctx, ctx->stream, 0, w.nnets, 0, w.N, carry, done_d, done_stride, 0);
  KBJ_CHECK_LAUNCH(ctx, "carry_reset_kernel");
  return 0;
}

int kbj_rollout(kbj_ctx* ctx, const float* params_d, kbj_carry* carry, uint32_t seed, uint32_t first_step_index, kbj_traj* tr) {
  if (!ctx || !params_d || !carry || !tr) return kbj_fail(ctx, "kbj_rollout: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  int N = w.N, H = w.H, T = tr->T;
  if (tr->N != N || T <= 0) return kbj_fail(ctx, "kbj_rollout: trajectory shape does not match the context");
  if (w.padded() && !w.inner) {   // the H-wide schedule on internal copies; the caller's start-carry snapshot is taken here, at its own width
    InnerScope in(w);
    const size_t ub = (size_t)2 * w.D * N * w.Hu * sizeof(float);
    float* c0[4] = {tr->carry0_actor_hc_d, tr->carry0_critic_hc_d, tr->carry0_actor_mirror_hc_d, tr->carry0_critic_mirror_hc_d};
    float* cu[4] = {carry->actor_hc_d, carry->critic_hc_d, carry->actor_mirror_hc_d, carry->critic_mirror_hc_d};
    for (int k = 0; k < w.nnets; ++k) {
      if (!c0[k] || !cu[k]) return kbj_fail(ctx, "kbj_rollout: carry / trajectory start-carry array is NULL");
      KBJ_HIP(ctx, hipMemcpyAsync(c0[k], cu[k], ub, hipMemcpyDeviceToDevice, ctx->stream));
    }
    pad_params(ctx, ctx->stream, params_d, w.pparams);
    kbj_carry pc = pad_carry(ctx, ctx->stream, *carry);
    kbj_traj pt = *tr;
    pt.carry0_actor_hc_d = w.pc0[0]; pt.carry0_critic_hc_d = w.pc0[1]; pt.carry0_actor_mirror_hc_d = w.pc0[2]; pt.carry0_critic_mirror_hc_d = w.pc0[3];
    const int rc = kbj_rollout(ctx, w.pparams, &pc, seed, first_step_index, &pt);
    unpad_carry(ctx, ctx->stream, *carry);
    return rc;
  }
  hipStream_t s = ctx->stream;
  kbj_nn_drop_prefetch(ctx);
  size_t la = w.net[0].ld_obs, lc = w.net[1].ld_obs, lx = KBJ_AUX_SIZE;
  // observation row T of the previous rollout is row 0 of this one
  KBJ_HIP(ctx, hipMemcpyAsync(tr->actor_obs_d, tr->actor_obs_d + (size_t)T * N * la, N * la * sizeof(float), hipMemcpyDeviceToDevice, s));
  KBJ_HIP(ctx, hipMemcpyAsync(tr->critic_obs_d, tr->critic_obs_d + (size_t)T * N * lc, N * lc * sizeof(float), hipMemcpyDeviceToDevice, s));
  KBJ_HIP(ctx, hipMemcpyAsync(tr->aux_d, tr->aux_d + (size_t)T * N * lx, N * lx * sizeof(float), hipMemcpyDeviceToDevice, s));
  size_t hcb = (size_t)2 * w.D * N * H * sizeof(float);
  KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_actor_hc_d, carry->actor_hc_d, hcb, hipMemcpyDeviceToDevice, s));
  KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_critic_hc_d, carry->critic_hc_d, hcb, hipMemcpyDeviceToDevice, s));
  KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_lpf_d, carry->lpf_d, (size_t)N * KBJ_NU * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (w.mirror) {
    if (!tr->carry0_actor_mirror_hc_d || !tr->carry0_critic_mirror_hc_d || !tr->carry0_lpf_mirror_d || !carry->actor_mirror_hc_d || !carry->critic_mirror_hc_d || !carry->lpf_mirror_d)
      return kbj_fail(ctx, "kbj_rollout: the mirror losses are enabled, carry and trajectory need the mirror-branch arrays");
    KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_actor_mirror_hc_d, carry->actor_mirror_hc_d, hcb, hipMemcpyDeviceToDevice, s));
    KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_critic_mirror_hc_d, carry->critic_mirror_hc_d, hcb, hipMemcpyDeviceToDevice, s));
    KBJ_HIP(ctx, hipMemcpyAsync(tr->carry0_lpf_mirror_d, carry->lpf_mirror_d, (size_t)N * KBJ_NU * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  // The actor -> env step -> carry reset chain runs on the caller's stream; the critic (and the mirror branches), which the env never waits
  // for, on a side lane: its small kernels and GEMMs slip into the env kernel's ramp-up / tail (-8.5 ms per iteration against the serial
  // order). KBJ_ROLLOUT_PIPELINE=0 (read per call, diagnostics): strictly serial on the caller's stream. (A two-lane software pipeline over
  // env halves existed until round 4: with 12 env wavefronts per CU no GEMM workgroup can be co-resident, the lanes only alternated - gone.)
  const char* pipe_env = getenv("KBJ_ROLLOUT_PIPELINE");
  const bool serial = pipe_env && atoi(pipe_env) == 0;
  hipStream_t cs = serial ? s : ctx->side[0];                 // critic + mirror branches
  const float* weff = fold_actor_weights(ctx, s, params_d);   // once: the parameters are fixed for the whole rollout
  repitch_input_weights(ctx, s, params_d);                    // (ahead of the fork: the side lane's critic reads the copy)
  if (!serial) {
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
    KBJ_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_fork, 0));
  }
  for (int t = 0; t < T; ++t) {
    const float* ao = tr->actor_obs_d + (size_t)t * N * la;
    const float* co = tr->critic_obs_d + (size_t)t * N * lc;
    float* aux_t = tr->aux_d + (size_t)t * N * lx;
    float* act = tr->action_d + (size_t)t * N * KBJ_NU;
    if (policy_nets(ctx, s, params_d, 0, 1, 0, N, ao, co, carry, seed, first_step_index + (uint32_t)t, ctx->rollout_argmax, act, tr->logp_d + (size_t)t * N, tr->value_d + (size_t)t * N, t & 1, weff)) return -1;
    if (policy_nets(ctx, cs, params_d, 1, w.nnets, 0, N, ao, co, carry, seed, first_step_index + (uint32_t)t, 0, act, tr->logp_d + (size_t)t * N, tr->value_d + (size_t)t * N, t & 1, weff)) return -1;
    int rc = kbj_env_step_range(ctx, s, 0, N, act, aux_t, tr->actor_obs_d + (size_t)(t + 1) * N * la, tr->critic_obs_d + (size_t)(t + 1) * N * lc,
                                tr->aux_d + (size_t)(t + 1) * N * lx, tr->qstate_d ? tr->qstate_d + (size_t)t * N * KBJ_QSTATE_SIZE : nullptr);
    if (rc) return rc;
    carry_reset_nets(ctx, s, 0, 1, 0, N, carry, aux_t + KBJ_AUX_DONE, KBJ_AUX_SIZE, (t + 1) & 1);   // the planes step t + 1 reads
    if (!serial) {   // the side lane needs this step's done flags and the next critic observation
      KBJ_HIP(ctx, hipEventRecord(ctx->ev_side[0], s));
      KBJ_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_side[0], 0));
    }
    carry_reset_nets(ctx, cs, 1, w.nnets, 0, N, carry, aux_t + KBJ_AUX_DONE, KBJ_AUX_SIZE, (t + 1) & 1);
  }
  KBJ_CHECK_LAUNCH(ctx, "kbj_rollout");
  if (!serial) {   // join the side lane back into the caller's stream
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_side[0], cs));
    KBJ_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_side[0], 0));
  }
  if (w.sched.rollout_step && (T & 1) && carry_h_home(ctx, s, 0, w.nnets, carry)) return -1;   // an odd number of steps leaves the live h planes in the partners
  return kbj_rewards(ctx, tr->aux_d, T, tr->reward_d, tr->reward_comps_d);
}

int kbj_gae(kbj_ctx* ctx, const kbj_traj* tr, float* adv_d, float* target_d) {
  if (!ctx || !tr || !adv_d || !target_d) return kbj_fail(ctx, "kbj_gae: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  const kbj_config& c = ctx->cfg_h;
  kbj_nn_drop_prefetch(ctx);
  hipLaunchKernelGGL(gae_kernel, g1(tr->N, 64), dim3(64), 0, ctx->stream, tr->value_d, tr->reward_d, tr->aux_d, tr->T, tr->N, c.gamma, c.lam, adv_d, target_d);
  KBJ_CHECK_LAUNCH(ctx, "gae_kernel");
  return 0;
}

}  // extern "C"

namespace {

// the parameter-independent gathers in front of the first recurrences: actor observation rows, keep flags, start carries (and low-pass state)
int head_gathers(kbj_ctx* ctx, hipStream_t st, const kbj_traj* tr, const int32_t* idx) {
  NnWs& w = *ws_of(ctx);
  const int T = tr->T, N = tr->N, H = w.H, B = w.B, R = T * B, D = w.D;
  const int lda = w.net[0].ld_obs;
  if (lda % 4 == 0 && ((size_t)tr->actor_obs_d & 15) == 0)
    hipLaunchKernelGGL(gather_rows4_kernel, g1((size_t)R * (lda / 4)), dim3(256), 0, st, reinterpret_cast<const float4*>(tr->actor_obs_d), idx, T, N, B, lda / 4, lda / 4, lda / 4,
                       reinterpret_cast<float4*>(w.tb[0].obs));
  else hipLaunchKernelGGL(gather_rows_kernel, g1((size_t)R * lda), dim3(256), 0, st, tr->actor_obs_d, idx, T, N, B, lda, lda, lda, w.tb[0].obs);
  GatherSmallArgs gs{tr->action_d, nullptr, nullptr, nullptr, nullptr, tr->aux_d, w.act, w.logp_old, w.val_old, w.adv, w.target, w.keep};
  hipLaunchKernelGGL(gather_small_kernel, g1((size_t)R), dim3(256), 0, st, gs, idx, T, N, B, KBJ_NU + 4, KBJ_NU + 5);   // keep flags: all the recurrences need of these
  const float* carry0[4] = {tr->carry0_actor_hc_d, tr->carry0_critic_hc_d, tr->carry0_actor_mirror_hc_d, tr->carry0_critic_mirror_hc_d};
  if (w.mirror && (!carry0[2] || !carry0[3] || !tr->carry0_lpf_mirror_d))
    return kbj_fail(ctx, "PPO pass: the mirror losses are enabled, the trajectory needs the mirror-branch carries");
  GatherCarryArgs gc;
  gc.nplanes = 0;
  for (int n = 0; n < w.nnets; ++n)
    for (int l = 0; l < D; ++l) {  // carry at the start of the trajectory: [N][H] planes h, c of every layer
      gc.src[gc.nplanes] = carry0[n] + (size_t)(2 * l) * N * H; gc.dst[gc.nplanes++] = w.tb[n].Hm[l];
      gc.src[gc.nplanes] = carry0[n] + (size_t)(2 * l + 1) * N * H; gc.dst[gc.nplanes++] = w.tb[n].Cm[l];
    }
  gc.nlpf = 0;
  gc.src[gc.nplanes] = tr->carry0_lpf_d; gc.dst[gc.nplanes] = w.lpf0; gc.nlpf++;
  if (w.mirror) { gc.src[gc.nplanes + 1] = tr->carry0_lpf_mirror_d; gc.dst[gc.nplanes + 1] = w.lpf0_m; gc.nlpf++; }
  hipLaunchKernelGGL(gather_carry_kernel, dim3((B * H + 255) / 256, gc.nplanes + gc.nlpf), dim3(256), 0, st, gc, idx, B, H);
  return 0;
}

// ---- the first half of a PPO minibatch pass, shared by kbj_ppo_grad and kbj_ppo_forward -----------------------------------------------
// gathers the minibatch (observations, actions, keep flags, start-of-trajectory carries; with `grad` also the old log-probs / values,
// advantages and targets), prepares the folded actor weights, and runs both nets forward through time: afterwards w.tb[n].Hout[D-1]
// holds the top layer's outputs and the BPTT stash is in place. Lanes: actor-type nets on ns[0] (the caller's stream), critic-type nets
// on ns[1]; on return the two lanes are still forked (the caller joins them).
int ppo_forward_nets(kbj_ctx* ctx, const float* params_d, const kbj_traj* tr, const int32_t* idx, const float* adv_d, const float* target_d, float* grad_d,
                     bool grad, hipStream_t ns[2]) {
  NnWs& w = *ws_of(ctx);
  const Sched& sc = w.sched;
  const int T = tr->T, N = tr->N, H = w.H, B = w.B, R = T * B, D = w.D;
  // lanes: ns[0] = actor-type nets, ns[1] = critic-type nets; one of them is the caller's stream (Sched::critic_on_caller), the other the context's second
  ns[0] = (sc.one_stream || !sc.critic_on_caller) ? ctx->stream : ctx->stream2;
  ns[1] = (sc.one_stream || sc.critic_on_caller) ? ctx->stream : ctx->stream2;
  hipStream_t s = ns[0];   // "s" below = the actor's lane
  // hand-off counters of all eight (sixteen with the mirror branches) recurrence launches of this call: one clear, ahead of both lanes
  // (kbj_ppo_grad leaves them cleared: each lane clears its nets' blocks behind its last recurrence, off the head of the next minibatch's chain)
  if (!w.counters_clean) KBJ_HIP(ctx, hipMemsetAsync(w.seq_counters, 0, SEQ_COUNTER_TOTAL * sizeof(unsigned), ctx->stream));
  w.counters_clean = false;
  KBJ_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  KBJ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  // (first on the actor's lane, ahead of the gathers: the two small launches depend on the parameters only, and behind the gathers they
  // queue for CU slots behind the 800 workgroups of the critic's input projection - 87 us on the actor's chain instead of ~25)
  if (sc.fold_actor) {
    // The actor's input projection (65 -> H, no activation) feeds only layer 0's input GEMM, so gates_0 = obs (W_ih0 W_in)^T +
    // (W_ih0 b_in + b_0): a 65-deep contraction instead of 65 -> H -> 4H (6.8 instead of 28.5 GFLOP per minibatch forward, and
    // 6.8 instead of 55 GFLOP backward). Same function, different rounding order (inside the parity tolerances).
    const NetOff& oa = w.net[0];
    GemmArgs g{params_d + oa.w_ih[0], params_d + oa.w_in, w.Weff, nullptr, 4 * H, oa.nin, H, H, oa.nin, oa.ld_obs, 0, 1, nullptr};
    gemm_launch<true, false>(s, g);
    hipLaunchKernelGGL(matvec_kernel, dim3((4 * H + 3) / 4), dim3(256), 0, s, params_d + oa.w_ih[0], params_d + oa.b_in, params_d + oa.b[0], 4 * H, H, w.beff);
  }
  auto gather = [&](hipStream_t st, const float* src, int wdt, int lds, float* dst, int ldd) {
    if (wdt % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((size_t)src & 15) == 0 && ((size_t)dst & 15) == 0)
      hipLaunchKernelGGL(gather_rows4_kernel, g1((size_t)R * (wdt / 4)), dim3(256), 0, st, reinterpret_cast<const float4*>(src), idx, T, N, B, wdt / 4, lds / 4, ldd / 4,
                         reinterpret_cast<float4*>(dst));
    else hipLaunchKernelGGL(gather_rows_kernel, g1((size_t)R * wdt), dim3(256), 0, st, src, idx, T, N, B, wdt, lds, ldd, dst);
  };
  // The large critic observation block (97 MB per minibatch, 45 us at HBM speed): the forward pass reads it once, through the critic's
  // input projection - which fetches its rows straight from the trajectory through the minibatch's indices (GemmArgs::a_idx) - so the
  // gathered copy, which the backward pass and the mirror branch want, is made on the critic's side lane, off the chain that starts the
  // critic's first recurrence. (One stream, mirror branches or an unfolded actor: gathered in front of the projection as before.)
  const bool gather_late = !sc.one_stream && !w.mirror && sc.fold_actor;
  if (!gather_late) gather(ns[1], tr->critic_obs_d, w.net[1].ld_obs, w.net[1].ld_obs, w.tb[1].obs, w.net[1].ld_obs);
  // What the FIRST recurrences read besides the parameters - the actor's observation rows, the keep flags, the start carries - depends on
  // the trajectory and the indices only. kbj_ppo_prefetch has queued these gathers behind the end of the previous kbj_ppo_grad, on a
  // side lane, where they run under the optimizer step and the folded-weight preparation instead of between them and the first recurrence.
  GatherSmallArgs gs{tr->action_d, grad ? tr->logp_d : nullptr, grad ? tr->value_d : nullptr, adv_d, target_d, tr->aux_d, w.act, w.logp_old, w.val_old, w.adv, w.target, w.keep};
  // A pending hint is ALWAYS waited for, matching or not: its gathers write the same workspace rows (observation copy, keep flags, start
  // carries) from a side lane, so a call with other arguments must order its own gathers behind them before it overwrites them.
  const bool pending = w.prefetched_idx != nullptr;
  const bool prefetched = pending && w.prefetched_idx == idx && w.prefetched_traj == tr && !sc.one_stream;
  w.prefetched_idx = nullptr; w.prefetched_traj = nullptr;
  if (pending) KBJ_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_prefetch, 0));
  if (!prefetched && head_gathers(ctx, s, tr, idx)) return -1;
  {
    // Nothing on the forward path needs the rest: actions, old log-probs / values, advantages, targets, the advantage statistics and the
    // cleared accumulators (gradient, folded layer-0 products) are wanted at the loss, two recurrences later. They run on the actor's
    // side lane, idle until the backward pass, instead of in front of the first recurrence (where they were ~170 us of a 6 ms
    // minibatch with the matrix cores idle); both net lanes wait for them behind their last forward recurrence, and the side lanes
    // fork from the net lanes after that point, so every later reader and accumulator is ordered behind the clears.
    hipStream_t sm = sc.one_stream ? s : ctx->side[0];
    if (!sc.one_stream) KBJ_HIP(ctx, hipStreamWaitEvent(sm, ctx->ev_fork, 0));
    hipLaunchKernelGGL(gather_small_kernel, g1((size_t)R * (KBJ_NU + 4)), dim3(256), 0, sm, gs, idx, T, N, B, 0, KBJ_NU + 4);
    if (grad) {
      KBJ_HIP(ctx, hipMemsetAsync(w.stats, 0, 16 * sizeof(double), sm));
#ifndef KBJ_NO_TAIL_CLEAN
      w.sumsq_clean = true;
#endif
      // (ordered in front of the end of this call on every lane: both net lanes wait for ev_small, the caller's stream joins them)
      if (w.ext_adv_sums) {   // global-batch statistics from the caller (all-reduced over the data-parallel ranks)
        KBJ_HIP(ctx, hipMemcpyAsync(w.stats, w.ext_adv_sums, 2 * sizeof(double), hipMemcpyDeviceToDevice, sm));
        KBJ_HIP(ctx, hipMemcpyAsync(w.stats + 11, w.ext_adv_sums + 2, sizeof(double), hipMemcpyDeviceToDevice, sm));
      } else {
        hipLaunchKernelGGL(adv_stats_kernel, dim3(32), dim3(256), 0, sm, w.adv, R, w.stats, sc.deterministic ? w.detd : (double*)nullptr);
        if (sc.deterministic) hipLaunchKernelGGL(reduce_double_kernel, dim3(1), dim3(64), 0, sm, w.detd, 32, 2, w.stats);
      }
      KBJ_HIP(ctx, hipMemsetAsync(grad_d, 0, w.nparams * sizeof(float), sm));
      if (sc.fold_actor)
        for (int n = 0; n < w.nnets; ++n)
          if ((n & 1) == 0 || sc.fold_critic) KBJ_HIP(ctx, hipMemsetAsync(w.Zeff[n], 0, (size_t)4 * H * w.net[n & 1].ld_obs * sizeof(float), sm));
    }
    if (!sc.one_stream) KBJ_HIP(ctx, hipEventRecord(ctx->ev_small, sm));
  }
  // the critic's lane needs keep / carries (gathered above on the actor's lane) before its first recurrence - not before its input
  // projection, which only reads its own gather: the wait sits in front of the recurrences below
  KBJ_HIP(ctx, hipEventRecord(ctx->ev_join, ns[0]));
  if (w.mirror)   // mirrored observation rows: actor's on the caller's stream, critic's behind its gather
    for (int k = 0; k < 2; ++k)
      hipLaunchKernelGGL(mirror_rows_kernel, g1((size_t)R * w.net[k].ld_obs), dim3(256), 0, ns[k], w.tb[k].obs, w.tb[2 + k].obs, (size_t)R, w.net[k].ld_obs, w.mtab[k]);
  // ---- forward through time: actor on the caller's stream, critic on the context's second stream (the recurrences are
  // latency bound, so the two nets overlap) ----
  static const int stamp_sel = getenv("KBJ_SEQ_STAMPS") ? atoi(getenv("KBJ_SEQ_STAMPS")) : 1;   // diagnostics build: 1 + net + 2 * layer picks the stamped forward launch
  const int stamp_net = (stamp_sel - 1) & 1, stamp_layer = ((stamp_sel - 1) >> 1) & 1;
  // forward: the K = H input projections (x W_ih^T + b) run inside the persistent recurrence, their MFMAs placed around the flag poll and
  // the h-tile fetch where the matrix pipe idles (kbj_lstm_seq.h FUSE): a fused launch takes 1.11 instead of 0.88 ms, the 0.41-0.48 ms
  // GEMM in front of it and its 210 MB round trip of G disappear: ppo_grad 7.72 -> 7.47 ms. Sched::fuse_ih = false: separate GEMMs.
  const bool fuse_ih = sc.fuse_ih;
  // the folded actor layer 0 the same way (observation row x Weff inside the recurrence, 17 k-steps): Sched::fuse_obs = false keeps the GEMM
  const bool fuse_obs = sc.fuse_obs && w.net[0].ld_obs == KBJ_LD_ACTOR;
  // Launch order is layer-major over the nets (nets 2, 3 = mirror branches, same weights, queued behind nets 0, 1 on the same two streams).
  for (int n = 0; n < w.nnets; ++n) {
    const NetOff& o = w.net[n & 1];
    if (sc.fold_actor && (n & 1) == 0) continue;   // actor-type nets: layer-0 gates come straight from the observations
    // X0 = obs W_in^T + b_in. The stored rows of W_in are nin (65 / 475) floats long - no 16-byte alignment, which would put the GEMM on
    // its element-wise load path for this operand (55 instead of ~90 TF for the critic's 12 GFLOP, at the head of the longer chain): a
    // re-pitched copy with the observation rows' stride (zeros behind column nin, as in the observation rows) is made first; the
    // contraction then runs over the padded width
    if (n < 2) hipLaunchKernelGGL(repitch_rows_kernel, g1((size_t)H * o.ld_obs), dim3(256), 0, ns[n & 1], params_d + o.w_in, H, o.nin, o.ld_obs, w.WinP[n & 1]);
    if (n == 1 && gather_late) {
      GemmArgs g{tr->critic_obs_d, w.WinP[1], w.tb[1].X0, params_d + o.b_in, R, H, o.ld_obs, o.ld_obs, o.ld_obs, H, 0, 1, nullptr};
      g.a_idx = idx; g.a_B = B; g.a_N = N;
      g.x3 = sc.gemm_x3 ? 1 : 0;
      gemm_launch<true, true>(ns[1], g, sc.gemm_x3 ? -1 : 2);   // 64 x 128 tiles (kbj_gemm.h: 800 tiles of 128 x 128 are 1.56 rounds paid as 2)
      // the copy under the recurrences, on the critic's side lane. It starts behind the ACTOR lane's head (ev_join: the folded-weight preparation
      // and, without a prefetch hint, the head gathers - about when the projection ends; beside it the two would share HBM: 175 instead of 144 us
      // for the GEMM) and behind the side-lane gathers (ev_small), so that ev_obs implies both and the critic's lane - the chain a minibatch
      // waits for - carries ONE event wait in front of its loss instead of a record and two waits (each costs the queue ~7 us, round 6)
      KBJ_HIP(ctx, hipStreamWaitEvent(ctx->side[1], ctx->ev_join, 0));
      KBJ_HIP(ctx, hipStreamWaitEvent(ctx->side[1], ctx->ev_small, 0));
      if (grad) gather(ctx->side[1], tr->critic_obs_d, w.net[1].ld_obs, w.net[1].ld_obs, w.tb[1].obs, w.net[1].ld_obs);   // the forward-only pass never reads the copy
      KBJ_HIP(ctx, hipEventRecord(ctx->ev_obs, ctx->side[1]));
      continue;
    }
    linear_fwd(ns[n & 1], w.tb[n].obs, o.ld_obs, w.WinP[n & 1], o.ld_obs, params_d + o.b_in, w.tb[n].X0, H, R, H, o.ld_obs, 0);
  }
  if (!sc.one_stream) KBJ_HIP(ctx, hipStreamWaitEvent(ns[1], ctx->ev_join, 0));   // keep / carries gathered on the actor's lane
  for (int l = 0; l < D; ++l) {
    for (int n = 0; n < w.nnets; ++n) {
      const NetOff& o = w.net[n & 1];
      TrainBufs& t = w.tb[n];
      if (sc.fold_actor && (n & 1) == 0 && l == 0) { if (!fuse_obs) linear_fwd(ns[0], t.obs, o.ld_obs, w.Weff, o.ld_obs, w.beff, t.G[0], 4 * H, R, 4 * H, o.nin, 0); }
      else if (!fuse_ih) linear_fwd(ns[n & 1], l == 0 ? t.X0 : t.Hout[l - 1], H, params_d + o.w_ih[l], H, params_d + o.b[l], t.G[l], 4 * H, R, 4 * H, H, 0);
    }
    for (int n = 0; n < w.nnets; ++n) {
      const NetOff& o = w.net[n & 1];
      TrainBufs& t = w.tb[n];
      SeqFwdArgs fa{t.G[l], params_d + o.w_hh[l], t.Hm[l], t.Cm[l], t.Hout[l], t.TanhC[l], w.keep, w.seq_counters + SEQ_COUNTER_WORDS * (4 * l + n), w.seq_err, T, B, (n == stamp_net && l == stamp_layer) ? w.seq_stamps : nullptr};
      if (sc.fold_actor && (n & 1) == 0 && l == 0) {
        if (fuse_obs) { fa.X = t.obs; fa.ldx = o.ld_obs; fa.Wih = w.Weff; fa.ldw = o.ld_obs; fa.bias = w.beff; fa.kx = o.nin; }   // gates_0 = obs Weff^T + beff inside the recurrence
      } else if (fuse_ih) {   // K = H input projections ride inside the recurrence (kbj_lstm_seq.h FUSE)
        fa.X = l == 0 ? t.X0 : t.Hout[l - 1]; fa.Wih = params_d + o.w_ih[l]; fa.bias = params_d + o.b[l];
      }
      if (seq_fwd(ctx, ns[n & 1], H, fa)) return -1;
    }
  }
  if (!sc.one_stream) {
    KBJ_HIP(ctx, hipStreamWaitEvent(ns[0], ctx->ev_small, 0));   // the side-lane gathers / clears above
    // the critic's lane: its gathered rows (read by its backward pass) - an event that implies ev_small (above) - or ev_small itself
    KBJ_HIP(ctx, hipStreamWaitEvent(ns[1], gather_late ? ctx->ev_obs : ctx->ev_small, 0));
  }
  return 0;
}

}  // namespace

extern "C" {

int kbj_ppo_forward(kbj_ctx* ctx, const float* params_d, const kbj_traj* tr, const int32_t* env_idx_d, int B, kbj_ppo_vars* out) {
  if (!ctx || !params_d || !tr || !env_idx_d || !out || !out->logp_d || !out->value_d) return kbj_fail(ctx, "kbj_ppo_forward: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  const kbj_config& c = ctx->cfg_h;
  if (B != w.B) return kbj_fail(ctx, "kbj_ppo_forward: B must equal config.batch_size");
  if (tr->T != w.T || tr->N != w.N) return kbj_fail(ctx, "kbj_ppo_forward: trajectory shape does not match the context");
  if (w.padded() && !w.inner) {
    InnerScope in(w);
    pad_params(ctx, ctx->stream, params_d, w.pparams);
    kbj_traj pt = pad_traj_carry0(ctx, ctx->stream, *tr);
    return kbj_ppo_forward(ctx, w.pparams, &pt, env_idx_d, B, out);
  }
  const int T = tr->T, H = w.H, R = T * B, D = w.D;
  hipStream_t ns[2];
  if (ppo_forward_nets(ctx, params_d, tr, env_idx_d, nullptr, nullptr, nullptr, false, ns)) return -1;
  hipStream_t s = ns[0];   // the actor's lane
  for (int n = 0; n < 2; ++n) {
    const NetOff& o = w.net[n];
    linear_fwd(ns[n], w.tb[n].Hout[D - 1], H, params_d + o.w_out, H, params_d + o.b_out, w.tb[n].Out, 40, R, o.nout, H, 0);
  }
  HeadParams hp{c.min_std, c.max_std, c.var_scale, c.lpf_alpha, w.net[0].ld_obs};
  hipLaunchKernelGGL(actor_head_pre_kernel, g1((size_t)R * KBJ_NU), dim3(256), 0, s, w.tb[0].Out, w.tb[0].obs, w.joint_bias_d, hp, R, w.y, w.sd);
  hipLaunchKernelGGL(actor_head_train_fwd_kernel, g1((size_t)B * KBJ_NU, 64), dim3(64), 0, s, w.keep, w.lpf0, hp, T, B, w.y);
  hipLaunchKernelGGL(gaussian_logp_kernel, g1(R), dim3(256), 0, s, w.y, w.sd, w.act, R, out->logp_d, out->entropy_d ? out->entropy_d : w.ent);
  if (out->action_std_d) KBJ_HIP(ctx, hipMemcpyAsync(out->action_std_d, w.sd, (size_t)R * KBJ_NU * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (out->action_mean_d) KBJ_HIP(ctx, hipMemcpyAsync(out->action_mean_d, w.y, (size_t)R * KBJ_NU * sizeof(float), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(critic_value_kernel, g1(R), dim3(256), 0, ns[1], w.tb[1].Out, 40, R, out->value_d);
  if (!w.sched.one_stream) {
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
    KBJ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  }
  hipLaunchKernelGGL(poison_vars_kernel, dim3(1), dim3(1), 0, ctx->stream, w.seq_err, out->logp_d, out->value_d);   // a recurrence timed out: no silent garbage
  KBJ_CHECK_LAUNCH(ctx, "kbj_ppo_forward");
  if (w.sched.debug_sync) return kbj_synchronize(ctx);
  return 0;
}

int kbj_ppo_grad(kbj_ctx* ctx, const float* params_d, const kbj_traj* tr, const int32_t* env_idx_d, int B, const float* adv_d, const float* target_d,
                 float* grad_d, float* metrics_d) {
  if (!ctx || !params_d || !tr || !env_idx_d || !adv_d || !target_d || !grad_d || !metrics_d) return kbj_fail(ctx, "kbj_ppo_grad: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  const kbj_config& c = ctx->cfg_h;
  const Sched& sc = w.sched;
  if (B != w.B) return kbj_fail(ctx, "kbj_ppo_grad: B must equal config.batch_size");
  int T = tr->T, N = tr->N, H = w.H;
  if (T != w.T || N != w.N) return kbj_fail(ctx, "kbj_ppo_grad: trajectory shape does not match the context");
  hipStream_t s = ctx->stream;
  if (w.padded() && !w.inner) {
    InnerScope in(w);
    pad_params(ctx, s, params_d, w.pparams);
    kbj_traj pt = pad_traj_carry0(ctx, s, *tr);
    const int rc = kbj_ppo_grad(ctx, w.pparams, &pt, env_idx_d, B, adv_d, target_d, w.pgrad, metrics_d);
    if (rc) return rc;
    unpad_params(ctx, s, w.pgrad, grad_d);          // the poison markers of a timed-out recurrence sit on real elements: they travel
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_actor_grad, s));   // the caller's actor slice is final only now
    KBJ_CHECK_LAUNCH(ctx, "pad_params_kernel");
    return 0;
  }
  const int R = T * B, D = w.D;
  KbjTimed timed(ctx, true);
  const bool one_stream = sc.one_stream, fold_actor = sc.fold_actor, fold_critic = sc.fold_critic;
  hipStream_t ns[2];
  struct X3Scope { X3Scope(int v) { g_gemm_x3 = v; } ~X3Scope() { g_gemm_x3 = 0; } } x3scope(sc.gemm_x3 ? 1 : 0);   // backward-pass GEMMs only
  if (ppo_forward_nets(ctx, params_d, tr, env_idx_d, adv_d, target_d, grad_d, true, ns)) return -1;
  s = ns[0];   // from here on "s" = the actor's lane (the caller's stream or the context's second one, Sched::critic_on_caller)
  static const int bstamp_sel = getenv("KBJ_SEQ_BSTAMPS") ? atoi(getenv("KBJ_SEQ_BSTAMPS")) : 0;   // diagnostics build only
  // Without the mirror branches the critic's one-output head, the value half of the loss and the head's backward are ONE kernel on the
  // critic's lane (critic_head_kernel: ~30 us instead of two degenerate GEMMs and three small kernels, ~220 us, on the longer chain).
  const bool fused_critic_head = sc.fused_critic_head && !w.mirror && w.net[1].nout == 1;
  for (int n = 0; n < w.nnets; ++n) {
    const NetOff& o = w.net[n & 1];
    if (n == 1 && fused_critic_head) continue;
    linear_fwd(ns[n & 1], w.tb[n].Hout[D - 1], H, params_d + o.w_out, H, params_d + o.b_out, w.tb[n].Out, 40, R, o.nout, H, 0);
  }
  // heads and losses: the policy terms need the actor only, the value terms the critic only (the mirror terms likewise), so each lane
  // computes its own and the two chains stay independent through the whole call: the shorter actor chain runs ahead, and its
  // recurrences meet the critic's GEMM phases instead of the critic's recurrences
  HeadParams hp{c.min_std, c.max_std, c.var_scale, c.lpf_alpha, w.net[0].ld_obs};
  PpoParams pp{c.clip_param, c.value_clip, c.value_loss_coef, c.entropy_coef, c.log_ratio_clip, c.adv_eps};
  hipLaunchKernelGGL(actor_head_pre_kernel, g1((size_t)R * KBJ_NU), dim3(256), 0, s, w.tb[0].Out, w.tb[0].obs, w.joint_bias_d, hp, R, w.y, w.sd);
  hipLaunchKernelGGL(actor_head_train_fwd_kernel, g1((size_t)B * KBJ_NU, 64), dim3(64), 0, s, w.keep, w.lpf0, hp, T, B, w.y);
  hipLaunchKernelGGL(gaussian_logp_kernel, g1(R), dim3(256), 0, s, w.y, w.sd, w.act, R, w.logp, w.ent);
  hipLaunchKernelGGL(ppo_loss_kernel, g1(R), dim3(256), 0, s, w.logp, w.value, w.ent, w.logp_old, w.val_old, w.adv, w.target, w.stats, pp, R, w.dlogp, w.dvalue,
                     w.stats + 2, one_stream ? 0 : 1);
  if (fused_critic_head) {
    const NetOff& oc = w.net[1];
    TrainBufs& tc = w.tb[1];
    const dim3 grid(2048), block(256);
#define KBJ_CRITIC_HEAD(V) hipLaunchKernelGGL((critic_head_kernel<V>), grid, block, 0, ns[1], tc.Hout[D - 1], params_d + oc.w_out, params_d + oc.b_out, w.val_old, w.target, \
                                              pp, R, w.value, w.dvalue, tc.dOut, tc.dHa, w.stats + 2)
    switch (H / 64) {
      case 1: KBJ_CRITIC_HEAD(1); break; case 2: KBJ_CRITIC_HEAD(2); break; case 3: KBJ_CRITIC_HEAD(3); break; case 4: KBJ_CRITIC_HEAD(4); break;
      case 5: KBJ_CRITIC_HEAD(5); break; case 6: KBJ_CRITIC_HEAD(6); break; case 7: KBJ_CRITIC_HEAD(7); break; case 8: KBJ_CRITIC_HEAD(8); break;
      default: return kbj_fail(ctx, "critic_head_kernel: hidden size above 512");
    }
#undef KBJ_CRITIC_HEAD
    if (one_stream) hipLaunchKernelGGL(ppo_loss_kernel, g1(R), dim3(256), 0, s, w.logp, w.value, w.ent, w.logp_old, w.val_old, w.adv, w.target, w.stats, pp, R, w.dlogp, w.dvalue,
                                       w.stats + 2, 1);   // (the actor's launch above was a no-op in this diagnostic mode)
  } else {
    hipLaunchKernelGGL(critic_value_kernel, g1(R), dim3(256), 0, ns[1], w.tb[1].Out, 40, R, w.value);
    hipLaunchKernelGGL(ppo_loss_kernel, g1(R), dim3(256), 0, ns[1], w.logp, w.value, w.ent, w.logp_old, w.val_old, w.adv, w.target, w.stats, pp, R, w.dlogp, w.dvalue,
                       w.stats + 2, one_stream ? 3 : 2);
  }
  if (w.mirror) {   // aux losses between each net and its mirror branch (train.py:1463-1481)
    hipLaunchKernelGGL(actor_head_pre_kernel, g1((size_t)R * KBJ_NU), dim3(256), 0, s, w.tb[2].Out, w.tb[2].obs, w.joint_bias_d, hp, R, w.y_m, w.sd_m);
    hipLaunchKernelGGL(actor_head_train_fwd_kernel, g1((size_t)B * KBJ_NU, 64), dim3(64), 0, s, w.keep, w.lpf0_m, hp, T, B, w.y_m);
    hipLaunchKernelGGL(mirror_loss_kernel, g1(R), dim3(256), 0, s, w.y, w.y_m, w.value, w.value_m, c.actor_mirror_loss_scale, c.critic_mirror_loss_scale, R, w.dy, w.dy_m,
                       w.dvalue, w.dvalue_m, w.stats + 2, one_stream ? 0 : 1);
    hipLaunchKernelGGL(critic_value_kernel, g1(R), dim3(256), 0, ns[1], w.tb[3].Out, 40, R, w.value_m);
    hipLaunchKernelGGL(mirror_loss_kernel, g1(R), dim3(256), 0, ns[1], w.y, w.y_m, w.value, w.value_m, c.actor_mirror_loss_scale, c.critic_mirror_loss_scale, R, w.dy, w.dy_m,
                       w.dvalue, w.dvalue_m, w.stats + 2, one_stream ? 3 : 2);
  }
  // The metrics line is nobody's input. It used to be a one-thread launch on a side lane right here - where every CU is about to be taken by the
  // backward recurrences (250 registers x 2 wavefronts per SIMD), so in most minibatches it sat dispatched-but-unplaced for ~450 us (2.9 % of the
  // summed kernel time in a rocprofv3 summary, and the side lane's next launch queued behind it): the same "waiting for a wave slot" artefact as the
  // pause kernel of rounds 4-5. It now runs at the END of the actor's lane (below), behind an event of the critic's loss half.
  // (the critic lane's record doubles as the fork of its side lane below when nothing is launched on it in between - the fused head: one packet)
  const bool critic_fork_recorded = !one_stream && fused_critic_head;
  hipEvent_t ev_critic_loss = critic_fork_recorded ? ctx->ev_side[1] : ctx->ev_pool[ctx->ev_next++ & 31];
  if (!one_stream) hipEventRecord(ev_critic_loss, ns[1]);
  // ---- backward ---- (dOut needs no clearing: the actor head writes all 40 columns, the critic's GEMMs read column 0 only)
  hipLaunchKernelGGL(actor_head_bwd_pre_kernel, g1((size_t)R * KBJ_NU), dim3(256), 0, s, w.tb[0].Out, w.y, w.sd, w.act, w.dlogp,
                     w.mirror ? w.dy : (const float*)nullptr, -c.entropy_coef / (float)R, hp, R, w.tb[0].dOut);
  hipLaunchKernelGGL(actor_head_train_bwd_kernel, g1((size_t)B * KBJ_NU, 64), dim3(64), 0, s, w.keep, hp, T, B, w.tb[0].dOut);
  if (!fused_critic_head) KBJ_HIP(ctx, hipMemcpy2DAsync(w.tb[1].dOut, 40 * sizeof(float), w.dvalue, sizeof(float), sizeof(float), R, hipMemcpyDeviceToDevice, ns[1]));
  if (w.mirror) {   // the mirror actor only sees the aux gradient on its filtered mean (no log-prob, no entropy term)
    hipLaunchKernelGGL(actor_head_bwd_pre_kernel, g1((size_t)R * KBJ_NU), dim3(256), 0, s, w.tb[2].Out, w.y_m, w.sd_m, w.y_m, w.zeroR, w.dy_m, 0.0f, hp, R, w.tb[2].dOut);
    hipLaunchKernelGGL(actor_head_train_bwd_kernel, g1((size_t)B * KBJ_NU, 64), dim3(64), 0, s, w.keep, hp, T, B, w.tb[2].dOut);
    KBJ_HIP(ctx, hipMemcpy2DAsync(w.tb[3].dOut, 40 * sizeof(float), w.dvalue_m, sizeof(float), sizeof(float), R, hipMemcpyDeviceToDevice, ns[1]));
  }
  // The critical path of a net is dOut -> dH -> (recurrence, dX) per layer. Weight/bias gradients hang off it: they go to the
  // net's side stream so the throughput-bound split-K GEMMs run beside the dX GEMMs in the GEMM phases.
  auto side_of = [&](int n) { return one_stream ? ns[n & 1] : ctx->side[n & 1]; };
  auto fork_side = [&](int n) { if (!one_stream) { hipEventRecord(ctx->ev_side[n & 1], ns[n & 1]); hipStreamWaitEvent(ctx->side[n & 1], ctx->ev_side[n & 1], 0); } };
  float *dh_above[4], *dx_out[4];
  for (int n = 0; n < w.nnets; ++n) {
    const NetOff& o = w.net[n & 1];
    TrainBufs& t = w.tb[n];
    if (!(n == 1 && fused_critic_head)) linear_bwd_input(ns[n & 1], t.dOut, 40, params_d + o.w_out, H, t.dHa, H, R, H, o.nout, 0);
    if (n == 1 && critic_fork_recorded) hipStreamWaitEvent(ctx->side[1], ctx->ev_side[1], 0);   // recorded with the metrics fork above
    else fork_side(n);
    linear_bwd_weight(ctx, side_of(n), t.dOut, 40, t.Hout[D - 1], H, grad_d + o.w_out, H, o.nout, H, R);
    colsum_acc(ctx, side_of(n), t.dOut, R, o.nout, 40, grad_d + o.b_out);
    dh_above[n] = t.dHa; dx_out[n] = t.dHb;
  }
  // INVARIANT of the schedule: no kernel waits for another LAUNCH. The only inter-workgroup waits are those of the persistent recurrences,
  // inside one launch whose grid is resident by construction (kbj_nn_create). Everything else is ordered by stream events, so any
  // serialisation of kernels (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL=3) runs the same schedule to the same results. (Round 3 / 4 carried
  // chunk-gated variants - a one-wavefront gate kernel polling a recurrence's progress in front of the GEMMs that consume it, KBJ_BWD_CHUNKS /
  // KBJ_DW_GATE: measured flat to 0.3 %, and dispatched alone under a serialising profiler the gate spins to its bound. They are gone,
  // DESIGN.md section 10.)
  // the bias terms of layer 0 (db_0 is complete with the net's layer-0 recurrence; the read-modify-write of dW_ih0 must follow the folded product into it, on the same lane): db_in = W_ih0^T db_0, dW_ih0 += db_0 b_in^T
  // (X0 = obs W_in^T + b_in)
  auto fold_bias_terms = [&](int k, hipStream_t st) {
    const NetOff& oa = w.net[k];
    float* part = det_partials(ctx, st);
    hipLaunchKernelGGL(matvec_t_acc_kernel, dim3((H + 63) / 64, 16), dim3(256), 0, st, params_d + oa.w_ih[0], grad_d + oa.b[0], 4 * H, H, grad_d + oa.b_in, part);
    if (part) hipLaunchKernelGGL(reduce_rows_kernel, dim3((H + 255) / 256), dim3(256), 0, st, part, 16, H, grad_d + oa.b_in);
    hipLaunchKernelGGL(outer_acc_kernel, g1((size_t)4 * H * H), dim3(256), 0, st, grad_d + oa.w_ih[0], grad_d + oa.b[0], params_d + oa.b_in, 4 * H, H);
  };
  bool bias_done[2] = {false, false};
  // WHERE A LAYER'S WEIGHT-GRADIENT PAIR RUNS (round 6: explicit stream order, no pause kernel). The input gradient of layer l is on the
  // net's critical chain (the recurrence of layer l - 1 reads it) and the recurrences take whole CUs (250 registers x 2 wavefronts per SIMD):
  // a weight-gradient GEMM that is eligible at the same moment as a recurrence races it for the CUs, and a recurrence advances at the pace of
  // its last workgroup to be placed (+4 % per iteration when that race is lost). So the pair of layer l > 0 is ORDERED BEHIND THE END of the
  // net's layer l - 1 recurrence: on the net's own lane behind its layer-0 work when l - 1 is the last layer (nothing else is left for that
  // lane to do, and a cross-lane hop costs 15-20 us each way), else on the net's side lane behind an event recorded after the recurrence
  // launch. Rounds 4-5 reached the same order through a 30 us sleep kernel that in practice waited for a wave slot until the recurrences
  // retired (KBJ_DW_DELAY_US, seq_delay_kernel: deleted) - an accident of occupancy, not an order. No launch's start depends on another
  // launch's occupancy any more.
  struct PendingDW { int n, l; };
  std::vector<PendingDW> pending_dw;
  auto launch_dw_pair = [&](hipStream_t st, int n, int l) {
    const NetOff& o = w.net[n & 1];
    TrainBufs& t = w.tb[n];
    linear_bwd_weight2(ctx, st, t.dGl[l], 4 * H, t.Hm[l], l == 0 ? t.X0 : t.Hout[l - 1], H, grad_d + o.w_hh[l], grad_d + o.w_ih[l], H, 4 * H, H, R);
  };
  for (int l = D - 1; l >= 0; --l) {
    for (int n = 0; n < w.nnets; ++n) {
      const NetOff& o = w.net[n & 1];
      TrainBufs& t = w.tb[n];
      SeqBwdArgs ba{t.G[l], t.TanhC[l], t.Cm[l], dh_above[n], w.keep, params_d + o.w_hh[l], t.dGl[l], w.seq_counters + SEQ_COUNTER_WORDS * (4 * (D + l) + n), w.seq_err, T, B, grad_d + o.b[l]};
      ba.db_part = det_partials(ctx, ns[n & 1]);   // deterministic mode: per-row-group bias sums, added in order below
      if (w.seq_bstamps && bstamp_sel == 1 + n + 2 * l) ba.stamps = w.seq_bstamps;
      const bool tiles16 = sc.bwd16;
      if (seq_bwd(ctx, ns[n & 1], H, ba, tiles16)) return -1;
      if (ba.db_part) hipLaunchKernelGGL(reduce_rows_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, ns[n & 1], ba.db_part, seq_bwd_row_groups(B, tiles16), 4 * H, grad_d + o.b[l]);
    }
    // the layer above's weight gradients, behind THIS layer's recurrence of the same net
    const bool tail_on_own_lane = l == 0 && !w.mirror && fold_actor;   // (the own_lane case below: the pair follows the net's layer-0 work on its lane)
    for (const PendingDW& p : pending_dw) {
      if (one_stream) { launch_dw_pair(ns[p.n & 1], p.n, p.l); continue; }   // one lane: stream order is the order
      if (tail_on_own_lane && ((p.n & 1) == 0 || fold_critic)) continue;      // issued in the net loop below
      hipStream_t ws = side_of(p.n);
      hipEventRecord(ctx->ev_dx[p.n & 1], ns[p.n & 1]);   // behind the recurrence launch(es) above on that lane
      hipStreamWaitEvent(ws, ctx->ev_dx[p.n & 1], 0);
      launch_dw_pair(ws, p.n, p.l);
    }
    for (int n = 0; n < w.nnets; ++n) {
      const NetOff& o = w.net[n & 1];
      TrainBufs& t = w.tb[n];
      hipStream_t s = ns[n & 1], ws = side_of(n);
      const bool folded = fold_actor && l == 0 && ((n & 1) == 0 || fold_critic);
      // The last thing a lane does - layer 0 of its net - needs no side lane: nothing is left on the net's lane for the weight
      // gradients to run beside, and a cross-lane hop costs 15-20 us each way (fork + join) in the minibatch's tail. (Not with mirror
      // branches: two nets per lane then read-modify-write the same dW_ih0 through their small products, which only the shared side
      // lane orders.)
      const bool own_lane = l == 0 && !w.mirror && folded;
      if (folded) {
        if (own_lane) ws = s;
        else fork_side(n);     // the side lane starts behind the whole recurrence
        // layer 0 backwards through the (activation-free) input projection without dX0: dW_hh0 += dG0^T Hm and Z = dG0^T obs in
        // one launch; two small products then carry Z back to the stored parameters
        float* Z = w.Zeff[n];
        const int ts = H % 128 == 0 ? 128 : 64;   // the column split (n1 = H) must fall on a tile boundary
        int sk = std::max(2, std::min(g_splitk_wgs / ((4 * H / ts) * (H / ts + (o.nin + ts - 1) / ts)), (R + 255) / 256));
        GemmArgs g{t.dGl[0], t.Hm[0], grad_d + o.w_hh[0], nullptr, 4 * H, H + o.nin, R, 4 * H, H, H, 1, sk, nullptr};
        g.B2 = t.obs; g.C2 = Z; g.n1 = H; g.ldb2 = o.ld_obs; g.ldc2 = o.ld_obs;
        g.skws = sk_workspace(ctx, ws, (size_t)sk * 4 * H * (H + o.nin));
        g.x3 = g_gemm_x3;
        gemm_launch<false, false>(ws, g, ts == 128 ? 1 : 0);
        // dW_in += W_ih0^T Z, dW_ih0 += Z W_in^T (the bias terms follow from db_0 at the end)
        // (a 4H-deep contraction on a handful of output tiles: split over k so that it is a short kernel, not a 130 us tail on 32 workgroups)
        GemmArgs g1a{params_d + o.w_ih[0], Z, grad_d + o.w_in, nullptr, H, o.nin, 4 * H, H, o.ld_obs, o.nin, 1, g_fold_sk, nullptr};
        g1a.skws = sk_workspace(ctx, ws, (size_t)g_fold_sk * H * o.nin);   // (same lane, stream-ordered behind the launch above: the slab is free again)
        gemm_launch<false, false>(ws, g1a);
        GemmArgs g2a{Z, params_d + o.w_in, grad_d + o.w_ih[0], nullptr, 4 * H, H, o.nin, o.ld_obs, o.nin, H, 1, 1, nullptr};
        gemm_launch<true, true>(ws, g2a);
        // (365.3 / 364.5 / 363.6 -> 364.8 / 363.8 / 362.9 ms per iteration: the critic's pair no longer waits for the last low-priority GEMM)
        if (own_lane && n < 2) { fold_bias_terms(n, ws); bias_done[n] = true; }   // right here, on the net's own lane: not behind the side lane's join at the end
        if (own_lane && !one_stream)   // the upper layers' pairs of this net: last on its own lane (two lanes in the tail instead of four: measured equal)
          for (const PendingDW& p : pending_dw) if (p.n == n && p.l > l) launch_dw_pair(s, p.n, p.l);
        continue;
      }
      // the input gradient as ONE product on the net's own lane, behind the recurrence (it is the next layer's input)
      linear_bwd_input(s, t.dGl[l], 4 * H, params_d + o.w_ih[l], H, dx_out[n], H, R, H, 4 * H, 0, H <= 256 ? 2 : -1);
      if (l > 0) pending_dw.push_back(PendingDW{n, l});   // issued in the next pass of the layer loop, behind layer l - 1's recurrence
      else {
        fork_side(n);   // the weight gradients start behind the input gradient
        launch_dw_pair(ws, n, l);
      }
      std::swap(dh_above[n], dx_out[n]);
    }
    // what was pending when this pass began has been issued; what this pass pushed stays for the next
    pending_dw.erase(std::remove_if(pending_dw.begin(), pending_dw.end(), [&](const PendingDW& p) { return p.l > l; }), pending_dw.end());
  }
  for (int n = 0; n < w.nnets; ++n) {
    const NetOff& o = w.net[n & 1];
    TrainBufs& t = w.tb[n];
    hipStream_t s = ns[n & 1];
    if (!(fold_actor && ((n & 1) == 0 || fold_critic))) {   // input projection (dh_above now holds dX0)
      linear_bwd_weight(ctx, s, dh_above[n], H, t.obs, o.ld_obs, grad_d + o.w_in, o.nin, H, o.nin, R);
      colsum_acc(ctx, s, dh_above[n], R, H, H, grad_d + o.b_in);
    }
    if (!one_stream) {
      // join of the net's side lane (output-projection gradients, bias sums). The critic's side lane joins the ACTOR's lane when nothing on the
      // critic's own lane reads what it wrote any more (bias terms done): the actor's lane is joined into the caller's stream below, so the
      // critic's chain - on the caller's stream by default - ends with ONE event wait, for an event that fired long before
      hipStream_t joiner = ((n & 1) == 1 && sc.critic_on_caller && bias_done[1]) ? ns[0] : s;
      hipEventRecord(ctx->ev_side[n & 1], ctx->side[n & 1]); hipStreamWaitEvent(joiner, ctx->ev_side[n & 1], 0);
    }
  }
  // Each lane ends with ONE small kernel behind its last gradient GEMM: it clears the hand-off counter blocks of the lane's nets (every recurrence
  // of the call has run: the next kbj_ppo_grad / kbj_ppo_forward starts without a clear on its chain) and poisons the lane's slice of the gradient
  // when a recurrence timed out (a truncated gradient: every data-parallel rank must skip the optimizer step; one NaN marker per slice is
  // enough, the norm covers the whole vector).
  // The actor's slice grad[0, nactor) is final here, ~0.4 ms before the critic's (shorter chain: folded layer 0, no 475-wide projection).
  // A data-parallel host may start its all-reduce now, under the critic's tail (kbj_stream_wait_actor_grad).
  if (fold_actor && !bias_done[0]) fold_bias_terms(0, ns[0]);
  if (!one_stream) hipStreamWaitEvent(ns[0], ev_critic_loss, 0);   // (fired ~3 ms ago)
  hipLaunchKernelGGL(ppo_metrics_kernel, dim3(1), dim3(1), 0, ns[0], w.stats + 2, w.stats, pp, R, metrics_d);
  if (one_stream) {
    if (fold_actor && fold_critic && !bias_done[1]) fold_bias_terms(1, ns[1]);
    hipLaunchKernelGGL(lane_tail_kernel, dim3(2 * MAXD * 4), dim3(SEQ_COUNTER_WORDS), 0, ctx->stream, w.seq_counters, -1, w.seq_err, grad_d, grad_d + w.nactor);
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_actor_grad, ctx->stream));
  } else {
    hipLaunchKernelGGL(lane_tail_kernel, dim3(2 * MAXD * 4), dim3(SEQ_COUNTER_WORDS), 0, ns[0], w.seq_counters, 0, w.seq_err, grad_d, (float*)nullptr);
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_actor_grad, ns[0]));
    if (fold_actor && fold_critic && !bias_done[1]) fold_bias_terms(1, ns[1]);   // on the critic's own lane (its side lane has joined it above), beside the actor's
    hipLaunchKernelGGL(lane_tail_kernel, dim3(2 * MAXD * 4), dim3(SEQ_COUNTER_WORDS), 0, ns[1], w.seq_counters, 1, w.seq_err, grad_d + w.nactor, (float*)nullptr);
    // join: the lane that is NOT the caller's stream into the caller's stream. With the critic on the caller's stream (default) the wait is for an
    // event that was recorded long ago - the actor's lane finishes first - so the tail of a minibatch crosses no queue either.
    KBJ_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
    KBJ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  }
#ifndef KBJ_NO_TAIL_CLEAN   // A/B
  w.counters_clean = true;
#endif
  KBJ_CHECK_LAUNCH(ctx, "kbj_ppo_grad");
  if (sc.debug_sync) return kbj_synchronize(ctx);   // KBJ_DEBUG=1: surface a hand-off timeout at the call that caused it
  return 0;
}

int kbj_ppo_prefetch(kbj_ctx* ctx, const kbj_traj* traj, const int32_t* env_idx_d) {
  if (!ctx || !ctx->nn_ws || !traj || !env_idx_d) return kbj_fail(ctx, "kbj_ppo_prefetch: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  if (traj->T != w.T || traj->N != w.N) return kbj_fail(ctx, "kbj_ppo_prefetch: trajectory shape does not match the context");
  if (w.sched.one_stream || w.padded()) return 0;      // nothing to overlap with on one lane; padded sizes re-pitch the carries per call: served without the hint
  hipStream_t lane = ctx->side[1];
  KBJ_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));      // behind everything queued so far (the previous minibatch's last reader of these buffers)
  KBJ_HIP(ctx, hipStreamWaitEvent(lane, ctx->ev_fork, 0));
  if (head_gathers(ctx, lane, traj, env_idx_d)) return -1;
  KBJ_HIP(ctx, hipEventRecord(ctx->ev_prefetch, lane));
  w.prefetched_idx = env_idx_d; w.prefetched_traj = traj;
  KBJ_CHECK_LAUNCH(ctx, "kbj_ppo_prefetch");
  return 0;
}

int kbj_recurrence_residency(kbj_ctx* ctx, int* grid_wgs, int* concurrent, int* slots) {
  if (!ctx || !ctx->nn_ws || !grid_wgs || !concurrent || !slots) return kbj_fail(ctx, "kbj_recurrence_residency: null argument");
  const NnWs& w = *ws_of(ctx);
  *grid_wgs = w.seq_grid; *concurrent = w.sched.one_stream ? 1 : 2; *slots = w.seq_slots;
  return 0;
}

int kbj_set_rollout_argmax(kbj_ctx* ctx, int argmax) {
  if (!ctx) return kbj_fail(nullptr, "kbj_set_rollout_argmax: null ctx");
  ctx->rollout_argmax = argmax != 0;
  return 0;
}

int kbj_set_advantage_sums(kbj_ctx* ctx, const double* sums_d) {
  if (!ctx || !ctx->nn_ws) return kbj_fail(ctx, "kbj_set_advantage_sums: null ctx");
  ws_of(ctx)->ext_adv_sums = sums_d;
  return 0;
}

int kbj_stream_wait_actor_grad(kbj_ctx* ctx, void* hip_stream) {
  if (!ctx) return kbj_fail(nullptr, "kbj_stream_wait_actor_grad: null ctx");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamWaitEvent((hipStream_t)hip_stream, ctx->ev_actor_grad, 0));
  return 0;
}

int kbj_set_learning_rate(kbj_ctx* ctx, float learning_rate) {
  // a negative rate is legal: the reference's scale_by_adam + scale_by_schedule chain (train.py:1074-1075) has no sign flip and is served
  // as written by passing minus the schedule's value (host/task.py update())
  if (!ctx || !std::isfinite(learning_rate)) return kbj_fail(ctx, "kbj_set_learning_rate: bad argument");
  ctx->cfg_h.learning_rate = learning_rate;
  return 0;
}

int kbj_adamw_step(kbj_ctx* ctx, float* params_d, float* m_d, float* v_d, const float* grad_d, int64_t step, float grad_scale) {
  if (!ctx || !params_d || !m_d || !v_d || !grad_d || step < 1) return kbj_fail(ctx, "kbj_adamw_step: bad argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  NnWs& w = *ws_of(ctx);
  const kbj_config& c = ctx->cfg_h;
  hipStream_t s = ctx->stream;
  double* sumsq = w.stats + 10;
  // one launch less on the chain between the last gradient GEMM and the next minibatch's first kernel: the accumulator is already zero when this
  // step follows a kbj_ppo_grad (whose clear of the statistics block covers it); any other caller (several steps in a row, per-pass accumulation) clears it here
  if (!w.sumsq_clean) KBJ_HIP(ctx, hipMemsetAsync(sumsq, 0, sizeof(double), s));
  w.sumsq_clean = false;
  double* part = w.sched.deterministic ? w.detd : nullptr;
  hipLaunchKernelGGL(sumsq_kernel, dim3(512), dim3(256), 0, s, grad_d, w.unparams, grad_scale, sumsq, part);
  if (part) hipLaunchKernelGGL(reduce_double_kernel, dim3(1), dim3(64), 0, s, part, 512, 1, sumsq);
  AdamParams ap{c.learning_rate, c.adam_b1, c.adam_b2, c.adam_eps, c.weight_decay, c.max_grad_norm,
                (float)(1.0 - std::pow((double)c.adam_b1, (double)step)), (float)(1.0 - std::pow((double)c.adam_b2, (double)step)), grad_scale};
  hipLaunchKernelGGL(adamw_kernel, g1(w.unparams), dim3(256), 0, s, params_d, m_d, v_d, grad_d, w.unparams, sumsq, ap, w.seq_err);
  KBJ_CHECK_LAUNCH(ctx, "adamw_kernel");
  return 0;
}

}  // extern "C"
