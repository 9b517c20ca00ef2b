// kbj_api.hip — context lifetime and the rollout driver of libkbj.so (C ABI in include/kbj.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "kbj_ctx.h"
#include "kbj_env_core.h"

thread_local std::string kbj_global_error;
thread_local kbj_ctx* kbj_prof_ctx = nullptr;

namespace {

// the HIP kernels are specialised on the kbot tree (kbj_env_core.h body_parent / dof_parent / dof_body)
bool topology_ok(const kbj_model& m, std::string& why) {
  if (m.magic != KBJ_MAGIC || m.version != KBJ_VERSION) { why = "bad magic/version"; return false; }
  if (m.nbody != KBJ_NBODY || m.nq != KBJ_NQ || m.nv != KBJ_NV || m.nu != KBJ_NU || m.ncap != KBJ_NCAP) { why = "unexpected sizes"; return false; }
  for (int b = 0; b < KBJ_NBODY; ++b) {
    int par = b <= 1 ? 0 : (b == 2 ? 1 : ((b - 3) % 5 == 0 ? 2 : b - 1));
    int num = b == 1 ? 6 : ((b >= 3 && b <= 22) ? 1 : 0);
    int adr = b == 1 ? 0 : ((b >= 3 && b <= 22) ? b + 3 : -1);
    if (m.body_parent[b] != par || m.body_dofnum[b] != num || m.body_dofadr[b] != adr) { why = "body tree differs from the kbot topology"; return false; }
  }
  for (int d = 0; d < KBJ_NV; ++d) {
    int body = d < 6 ? 1 : d - 3;
    int par = d == 0 ? -1 : (d < 6 ? d - 1 : ((d - 6) % 5 == 0 ? 5 : d - 1));
    if (m.dof_body[d] != body || m.dof_parent[d] != par) { why = "dof tree differs from the kbot topology"; return false; }
  }
  const int cap_body[4] = {7, 7, 12, 12};
  for (int c = 0; c < 4; ++c) if (m.cap_body[c] != cap_body[c]) { why = "collision capsules must sit on the foot bodies 7 and 12"; return false; }
  if (m.base_body != 1 || m.torso_body != 2 || m.lfoot_body != 7 || m.rfoot_body != 12 || m.imu_body != 23) { why = "unexpected named body ids"; return false; }
  return true;
}

}  // namespace

int kbj_nn_create(kbj_ctx* ctx);   // kbj_nn.hip
void kbj_nn_destroy(kbj_ctx* ctx);
int kbj_nn_check_errors(kbj_ctx* ctx);

extern "C" {

int kbj_sizeof_model(void) { return (int)sizeof(kbj_model); }
int kbj_sizeof_config(void) { return (int)sizeof(kbj_config); }
int kbj_sizeof_traj(void) { return (int)sizeof(kbj_traj); }
int kbj_sizeof_carry(void) { return (int)sizeof(kbj_carry); }

// Host-only validation of the sizes the kernels are built for (no device needed; kbj_create runs it first). Besides the plain range
// checks: the GEMM's interior tiles and the recurrences' hand-off tiles are fetched with buffer loads whose per-thread and scalar byte
// offsets are 32-bit against a descriptor of 2^31 - 1 bytes (kbj_gemm.h load_tile, kbj_lstm_seq.h SeqTile::load) - an operand that
// reaches 2 GiB would be read as zeros beyond that offset, silently. The largest operands are the BPTT stash of one minibatch
// [T * B][4 H] (and the gathered critic observations [T * B][476]) and, per control step, the [N][476] observation rows and [N][4 H] gates.
int kbj_check_config(const kbj_config* cfg, char* why, size_t why_bytes) {
  auto fail = [&](const std::string& msg) {
    if (why && why_bytes) { std::snprintf(why, why_bytes, "%s", msg.c_str()); }
    return -1;
  };
  if (!cfg) return fail("null config");
  if (cfg->num_envs <= 0 || cfg->substeps <= 0 || cfg->rollout_len <= 0) return fail("bad config sizes");
  if (cfg->command_mode < 0 || cfg->command_mode > 2) return fail("command_mode must be 0 (UnifiedCommand sampler), 1 (fixed command) or 2 (the sampler with jax.random's key handling)");
  if (cfg->solver_newton != 1) return fail("only the Newton solver is implemented on the GPU (solver_newton = 1)");
  if (cfg->hidden_size < 1 || cfg->hidden_size > 512 || cfg->depth < 1 || cfg->depth > KBJ_MAX_DEPTH)
    return fail("hidden_size must be in 1..512 (multiples of 64 run unpadded; above 256 on the wide, untuned schedule) and depth in 1..4 (train.py:78-85 defaults 128 / 2, launch 256 / 2)");
  if (cfg->extra_obs_actor < 0 || cfg->extra_obs_actor > KBJ_MAX_EXTRA_OBS || cfg->extra_obs_critic < 0 || cfg->extra_obs_critic > KBJ_MAX_EXTRA_OBS)
    return fail("extra_obs_actor / extra_obs_critic must be in 0.." + std::to_string(KBJ_MAX_EXTRA_OBS));
  const unsigned long long lim = 1ull << 31;
  const unsigned long long Hp = (unsigned long long)((cfg->hidden_size + 63) / 64 * 64), T = (unsigned long long)cfg->rollout_len;
  const unsigned long long B = (unsigned long long)(cfg->batch_size > 0 ? cfg->batch_size : cfg->num_envs), N = (unsigned long long)cfg->num_envs;
  const unsigned long long ldc = (unsigned long long)KBJ_LD_OF(KBJ_NOBS_CRITIC + cfg->extra_obs_critic);
  const unsigned long long wide = 4 * Hp > ldc ? 4 * Hp : ldc;
  if (T * B * wide * sizeof(float) >= lim)
    return fail("rollout_len x batch_size x max(4 hidden_size, 476) x 4 bytes reaches 2 GiB: the minibatch's stash arrays are addressed with 32-bit "
                "byte offsets (buffer loads) - use a smaller batch_size (" + std::to_string(T * B * wide * sizeof(float)) + " bytes)");
  if (N * wide * sizeof(float) >= lim)
    return fail("num_envs x max(4 hidden_size, 476) x 4 bytes reaches 2 GiB: one control step's observation / gate rows are addressed with 32-bit byte "
                "offsets - shard the envs over more GPUs (" + std::to_string(N * wide * sizeof(float)) + " bytes)");
  return 0;
}

const char* kbj_last_error(const kbj_ctx* ctx) { return ctx ? ctx->error.c_str() : kbj_global_error.c_str(); }

int kbj_create(kbj_ctx** out, const void* model_blob, size_t model_bytes, const kbj_config* cfg, int device, void* hip_stream) {
  if (!out || !model_blob || !cfg) return kbj_fail(nullptr, "kbj_create: null argument");
  *out = nullptr;
  if (model_bytes != sizeof(kbj_model)) return kbj_fail(nullptr, "kbj_create: model blob has the wrong size");
  kbj_ctx* ctx = new kbj_ctx();
  std::memcpy(&ctx->model_h, model_blob, sizeof(kbj_model));
  ctx->cfg_h = *cfg;
  std::string why;
  if (!topology_ok(ctx->model_h, why)) { delete ctx; return kbj_fail(nullptr, "kbj_create: " + why); }
  {
    char msg[512];
    if (kbj_check_config(cfg, msg, sizeof(msg)) != 0) { delete ctx; return kbj_fail(nullptr, std::string("kbj_create: ") + msg); }
  }
  ctx->device = device;
  ctx->stream = (hipStream_t)hip_stream;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) { delete ctx; return kbj_fail(nullptr, "kbj_create: no HIP device (this library has no CPU fallback)"); }
  if (device < 0 || device >= ndev) { delete ctx; return kbj_fail(nullptr, "kbj_create: bad device index"); }
  auto cleanup = [&](const std::string& msg) { kbj_destroy(ctx); return kbj_fail(nullptr, msg); };
#define KBJ_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return cleanup(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
  KBJ_TRY(hipSetDevice(device));
  size_t N = cfg->num_envs;
  KBJ_TRY(hipMalloc(&ctx->model_d, sizeof(kbj_model)));
  KBJ_TRY(hipMalloc(&ctx->cfg_d, sizeof(kbj_config)));
  KBJ_TRY(hipMemcpy(ctx->model_d, &ctx->model_h, sizeof(kbj_model), hipMemcpyHostToDevice));
  KBJ_TRY(hipMemcpy(ctx->cfg_d, &ctx->cfg_h, sizeof(kbj_config), hipMemcpyHostToDevice));
  {
    kbj::KbjModelLds mc;
    kbj::model_lds_fill(mc, ctx->model_h);
    KBJ_TRY(hipMalloc(&ctx->mc_d, sizeof(mc)));
    KBJ_TRY(hipMemcpy(ctx->mc_d, &mc, sizeof(mc), hipMemcpyHostToDevice));
    const kbj::PhysConst pc = kbj::phys_const(ctx->cfg_h, ctx->model_h);
    KBJ_TRY(hipMalloc(&ctx->pc_d, sizeof(pc)));
    KBJ_TRY(hipMemcpy(ctx->pc_d, &pc, sizeof(pc), hipMemcpyHostToDevice));
  }
  KBJ_TRY(hipMalloc(&ctx->ep_d, N * KBJ_EP_SIZE * sizeof(float)));
  KBJ_TRY(hipMalloc(&ctx->es_d, N * KBJ_ES_SIZE * sizeof(float)));
  KBJ_TRY(hipMalloc(&ctx->rcarry_d, N * KBJ_RC_SIZE * sizeof(float)));
  KBJ_TRY(hipMemset(ctx->ep_d, 0, N * KBJ_EP_SIZE * sizeof(float)));
  KBJ_TRY(hipMemset(ctx->es_d, 0, N * KBJ_ES_SIZE * sizeof(float)));
  KBJ_TRY(hipMemset(ctx->rcarry_d, 0, N * KBJ_RC_SIZE * sizeof(float)));
  KBJ_TRY(hipEventCreate(&ctx->ev0));
  KBJ_TRY(hipEventCreate(&ctx->ev1));
  // stream2 = the SECOND net lane of the update. Rounds 1-5 (KBJ_CRITIC_LANE=2nd): it carries the critic-type nets - the longer chain (a 475-wide
  // input projection in front of layer 0 that the actor folds away), which decides the length of a minibatch - at the highest queue priority
  // (6.66 -> 6.61 ms per minibatch then). Round 6 (default): the critic's chain runs on the CALLER's stream (kbj_nn.hip Sched::critic_on_caller: no
  // queue hop between the optimizer step and the chain's two ends), stream2 carries the actor's chain and gets the normal priority - it has
  // ~0.3 ms of slack and should not take CUs from the critic where they compete (measured: normal 354.7 / high 355.3 ms, then 350.7 / 350.9).
  {
    int lo = 0, hi = 0;
    KBJ_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* cl = getenv("KBJ_CRITIC_LANE");
    const bool critic_on_stream2 = cl && std::string(cl) == "2nd";
    KBJ_TRY(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, critic_on_stream2 ? hi : (lo + hi) / 2));
  }
  // The side lanes carry work that hangs off the critical chain of the update (weight-gradient GEMMs, bias sums) and the critic of the
  // rollout: lowest queue priority, so that when a dX GEMM of the chain and a dW GEMM compete for CUs the chain's workgroups go first.
  // Nearly zero-sum (the dX GEMMs finish in 520 instead of 756 us, but the displaced dW GEMMs then run beside the backward recurrences,
  // which slow from 895 to 1209 us): 6.91 -> 6.86 ms per minibatch, 444.2 -> 441.8 ms per iteration.
  int prio_least = 0, prio_greatest = 0;
  KBJ_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  for (int n = 0; n < 2; ++n) {
    KBJ_TRY(hipStreamCreateWithPriority(&ctx->side[n], hipStreamNonBlocking, prio_least));
    KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_side[n], hipEventDisableTiming));
  }
  for (int n = 0; n < 2; ++n) KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_dx[n], hipEventDisableTiming));
  for (int k = 0; k < 32; ++k) KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_pool[k], hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_actor_grad, hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_small, hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_prefetch, hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_obs, hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  KBJ_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
#undef KBJ_TRY
  if (kbj_nn_create(ctx) != 0) { std::string msg = ctx->error; kbj_destroy(ctx); return kbj_fail(nullptr, msg); }
  *out = ctx;
  return 0;
}

int kbj_destroy(kbj_ctx* ctx) {
  if (!ctx) return 0;
  hipSetDevice(ctx->device);
  kbj_nn_destroy(ctx);
  if (ctx->model_d) hipFree(ctx->model_d);
  if (ctx->cfg_d) hipFree(ctx->cfg_d);
  if (ctx->mc_d) hipFree(ctx->mc_d);
  if (ctx->pc_d) hipFree(ctx->pc_d);
  if (ctx->ep_d) hipFree(ctx->ep_d);
  if (ctx->es_d) hipFree(ctx->es_d);
  if (ctx->rcarry_d) hipFree(ctx->rcarry_d);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  for (int k = 0; k < 32; ++k) if (ctx->ev_pool[k]) hipEventDestroy(ctx->ev_pool[k]);
  if (ctx->ev_actor_grad) hipEventDestroy(ctx->ev_actor_grad);
  if (ctx->ev_small) hipEventDestroy(ctx->ev_small);
  if (ctx->ev_prefetch) hipEventDestroy(ctx->ev_prefetch);
  if (ctx->ev_obs) hipEventDestroy(ctx->ev_obs);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->stream2) hipStreamDestroy(ctx->stream2);
  for (int n = 0; n < 2; ++n) {
    if (ctx->side[n]) hipStreamDestroy(ctx->side[n]);
    if (ctx->ev_side[n]) hipEventDestroy(ctx->ev_side[n]);
    if (ctx->ev_dx[n]) hipEventDestroy(ctx->ev_dx[n]);
  }
  delete ctx;
  return 0;
}

int kbj_synchronize(kbj_ctx* ctx) {
  if (!ctx) return kbj_fail(nullptr, "kbj_synchronize: null ctx");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return kbj_nn_check_errors(ctx);
}

int kbj_profile_begin(kbj_ctx* ctx) {
  if (!ctx) return kbj_fail(nullptr, "kbj_profile_begin: null ctx");
  ctx->profiling = true; ctx->env_ms = ctx->nn_ms = 0; ctx->env_launches = ctx->nn_launches = 0;
  ctx->krecs.clear();
  kbj_prof_ctx = ctx;
  return 0;
}
int kbj_profile_end(kbj_ctx* ctx, float* env_step_ms, int* env_step_launches, float* nn_ms, int* nn_launches) {
  if (!ctx) return kbj_fail(nullptr, "kbj_profile_end: null ctx");
  ctx->profiling = false;
  kbj_prof_ctx = nullptr;
  KBJ_HIP(ctx, hipDeviceSynchronize());
  static const char* const names[KBJ_KIND_COUNT] = {
      "kbj::gemm_f32_kernel<2, 1, false, false, 2, 4>", "kbj::gemm_f32_kernel<2, 1, false, true, 2, 4>", "kbj::gemm_f32_kernel<2, 1, true, false, 2, 4>",
      "kbj::gemm_f32_kernel<2, 1, true, true, 2, 4>",   "kbj::gemm_f32_kernel<1, 1, false, false, 2, 2>", "kbj::gemm_f32_kernel<1, 1, false, true, 2, 2>",
      "kbj::gemm_f32_kernel<1, 1, true, false, 2, 2>",  "kbj::gemm_f32_kernel<1, 1, true, true, 2, 2>",   "kbj::lstm_seq_fwd_kernel", "kbj::lstm_seq_bwd_kernel", "env_step_kernel", "kbj::lstm_seq_fwd_kernel",
      "kbj::lstm_seq_fwd_kernel", "kbj::lstm_step_kernel", "kbj::lstm_step_kernel",
      // gemm_x3_kernel<TM, A_KC, B_KC, GEN> as rocprofv3 prints the instantiations the launcher uses (kbj_ctx.h kbj_kind_gemm_x3)
      "kbj::gemm_x3_kernel<2, false, false, false>", "kbj::gemm_x3_kernel<2, false, true, false>", "kbj::gemm_x3_kernel<2, true, false, false>",
      "kbj::gemm_x3_kernel<2, true, true, false>", "kbj::gemm_x3_kernel<2, true, true, true>", "kbj::gemm_x3_kernel<1, true, true, false>",
      "kbj::gemm_x3_kernel<1, true, true, true>", "kbj::lstm_seq_bwd16_kernel", "kbj::gemm_f32_kernel<1, 1, true, true, 2, 4>", "kbj::gemm_f32_kernel<1, 1, true, false, 2, 4>"};
  for (int k = 0; k < KBJ_KIND_COUNT; ++k) {
    kbj_kernel_stat& st = ctx->kstats[k];
    const int uw = 2;   // wavefront pairs per recurrence workgroup (kbj_nn.hip SEQ_UW), as rocprofv3 prints the template argument
    const int hk = (ctx->cfg_h.hidden_size + 63) / 64 * 64;   // the kernels' hidden size (kbj_nn.hip NnWs::H)
    if (k == KBJ_KIND_SEQ_BWD) snprintf(st.name, sizeof(st.name), "%s<%d, %d>", hk > 256 ? "kbj::lstm_seq_bwd_wide_kernel" : names[k], hk, uw);   // kbj_nn.hip SEQ_FUSED_MAX_H
    else if (k == KBJ_KIND_SEQ_FWD || k == KBJ_KIND_SEQ_FWD_FUSED || k == KBJ_KIND_SEQ_FWD_OBS)   // as rocprofv3 prints the template arguments
      snprintf(st.name, sizeof(st.name), "%s<%d, %d, %s, %d>", names[k], hk, uw, k == KBJ_KIND_SEQ_FWD ? "false" : "true",
               k == KBJ_KIND_SEQ_FWD_OBS ? KBJ_LD_ACTOR : hk);
    else if (k == KBJ_KIND_SEQ_BWD16) snprintf(st.name, sizeof(st.name), "%s<%d>", names[k], hk);
    else if (k == KBJ_KIND_LSTM_STEP || k == KBJ_KIND_LSTM_STEP_OBS)
      snprintf(st.name, sizeof(st.name), "%s<%d, 2, %d>", names[k], hk, k == KBJ_KIND_LSTM_STEP_OBS ? KBJ_LD_ACTOR : hk);
    else snprintf(st.name, sizeof(st.name), "%s", names[k]);
    st.launches = 0; st.total_ms = 0; st.flops = 0;
  }
  for (KbjKernelRec& r : ctx->krecs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { kbj_kernel_stat& st = ctx->kstats[r.kind]; st.launches++; st.total_ms += ms; st.flops += r.flops; }
    hipEventDestroy(r.a); hipEventDestroy(r.b);
  }
  ctx->krecs.clear();
  ctx->env_ms = ctx->kstats[KBJ_KIND_ENV_STEP].total_ms;
  ctx->env_launches = ctx->kstats[KBJ_KIND_ENV_STEP].launches;
  if (env_step_ms) *env_step_ms = ctx->env_ms;
  if (env_step_launches) *env_step_launches = ctx->env_launches;
  if (nn_ms) *nn_ms = ctx->nn_ms;
  if (nn_launches) *nn_launches = ctx->nn_launches;
  return 0;
}

int kbj_profile_kernel_stats(kbj_ctx* ctx, kbj_kernel_stat* out, int capacity, int* count) {
  if (!ctx || !out || !count) return kbj_fail(ctx, "kbj_profile_kernel_stats: null argument");
  int n = 0;
  for (int k = 0; k < KBJ_KIND_COUNT && n < capacity; ++k)
    if (ctx->kstats[k].launches > 0) out[n++] = ctx->kstats[k];
  *count = n;
  return 0;
}

}  // extern "C"
