// kbj_ctx.h — library context shared by the translation units of libkbj.so (host side only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstdio>
#include "kbj.h"

// per-launch HIP-event records of the MFMA kernels while kbj_profile_begin/end is active (bench.py roofline)
struct KbjKernelRec { int kind; double flops; hipEvent_t a, b; };
enum { KBJ_KIND_GEMM = 0 /* +4 small tile, +2 A k-contiguous, +1 B k-contiguous */, KBJ_KIND_SEQ_FWD = 8, KBJ_KIND_SEQ_BWD = 9, KBJ_KIND_ENV_STEP = 10, KBJ_KIND_SEQ_FWD_FUSED = 11, KBJ_KIND_SEQ_FWD_OBS = 12, KBJ_KIND_LSTM_STEP = 13, KBJ_KIND_LSTM_STEP_OBS = 14,
       // gemm_x3_kernel<TM, A_KC, B_KC, GEN> (kbj_config.gemm_bf16x3), one kind per instantiation the launcher uses: 15..18 = <2, A_KC, B_KC, false>
       // (+2 A k-contiguous, +1 B k-contiguous), 19 = <2, true, true, true>, 20 = <1, true, true, false>, 21 = <1, true, true, true>
       KBJ_KIND_GEMM_X3 = 15, KBJ_KIND_GEMM_X3_GEN = 19, KBJ_KIND_GEMM_X3_SMALL = 20, KBJ_KIND_GEMM_X3_SMALL_GEN = 21,
       KBJ_KIND_SEQ_BWD16 = 22 /* lstm_seq_bwd16_kernel<H> */, KBJ_KIND_GEMM_64x128 = 23 /* gemm_f32_kernel<1, 1, true, true, 2, 4>: the critic's input projection; 24 = <1, 1, true, false, 2, 4>: the input gradients */,
       KBJ_KIND_COUNT = 25 };
inline int kbj_kind_gemm_x3(int tm, bool a_kc, bool b_kc, bool gen) {
  if (tm == 2) return gen ? KBJ_KIND_GEMM_X3_GEN : KBJ_KIND_GEMM_X3 + (a_kc ? 2 : 0) + (b_kc ? 1 : 0);
  return gen ? KBJ_KIND_GEMM_X3_SMALL_GEN : KBJ_KIND_GEMM_X3_SMALL;
}

struct kbj_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second lane for the critic network inside kbj_ppo_grad
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t side[2] = {nullptr, nullptr};   // per-net side lanes for weight-gradient GEMMs
  hipEvent_t ev_dx[2] = {nullptr, nullptr};   // per net lane: the lane's recurrence launches of a layer (the upper layer's weight-gradient pair on the side lane starts behind their END)
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  hipEvent_t ev_obs = nullptr;                // the critic's gathered observation rows are in place (side lane, ppo_forward_nets)
  hipEvent_t ev_prefetch = nullptr;           // kbj_ppo_prefetch: the next minibatch's head gathers are done
  hipEvent_t ev_small = nullptr;              // the minibatch's small gathers / clears on the actor's side lane are done (ppo_forward_nets)
  hipEvent_t ev_actor_grad = nullptr;         // recorded by kbj_ppo_grad once the ACTOR's slice of the gradient is final (kbj_stream_wait_actor_grad)
  hipEvent_t ev_pool[32] = {};                // lane-alignment events of kbj_ppo_grad
  int ev_next = 0;
  kbj_model model_h;
  kbj_config cfg_h;
  kbj_model* model_d = nullptr;
  kbj_config* cfg_d = nullptr;
  void* pc_d = nullptr;         // kbj::PhysConst (kbj_env_core.h): solver / impedance / terrain constants derived from the config, computed at create
  float* mc_d = nullptr;        // KbjModelLds image (kbj_env_core.h): the model constants every env workgroup copies into LDS
  float* ep_d = nullptr;        // [N][KBJ_EP_SIZE]
  float* es_d = nullptr;        // [N][KBJ_ES_SIZE]
  float* rcarry_d = nullptr;    // [N][KBJ_RC_SIZE] reward carries
  uint32_t seed = 0;
  int rollout_argmax = 0;       // kbj_set_rollout_argmax: kbj_rollout acts with the distribution's mode (validation rollouts, train.py:1564)
  float* qstate_next = nullptr; // kbj_env_record_state: where the next kbj_env_step writes its state record (one-shot)
  // NN workspace (kbj_nn.hip)
  void* nn_ws = nullptr;
  size_t nn_ws_bytes = 0;
  // profiling
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float env_ms = 0, nn_ms = 0;
  int env_launches = 0, nn_launches = 0;
  std::vector<KbjKernelRec> krecs;
  kbj_kernel_stat kstats[KBJ_KIND_COUNT];
  std::string error;
};

extern thread_local std::string kbj_global_error;

inline int kbj_fail(kbj_ctx* ctx, const std::string& msg) {
  if (ctx) ctx->error = msg;
  kbj_global_error = msg;
  return -1;
}

#define KBJ_HIP(ctx, call)                                                                                   \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return kbj_fail(ctx, std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

#define KBJ_CHECK_LAUNCH(ctx, name)                                                                          \
  do {                                                                                                       \
    hipError_t e_ = hipGetLastError();                                                                       \
    if (e_ != hipSuccess) return kbj_fail(ctx, std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
  } while (0)

// the context being profiled on this host thread (nullptr outside kbj_profile_begin/end)
extern thread_local kbj_ctx* kbj_prof_ctx;

// brackets ONE kernel launch on stream s with two timing events
struct KbjKernelTimer {
  kbj_ctx* c; hipStream_t s; hipEvent_t b = nullptr;
  KbjKernelTimer(hipStream_t st, int kind, double flops) : c(kbj_prof_ctx), s(st) {
    if (!c) return;
    hipEvent_t a;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { c = nullptr; return; }
    c->krecs.push_back(KbjKernelRec{kind, flops, a, b});
    hipEventRecord(a, s);
  }
  ~KbjKernelTimer() { if (c) hipEventRecord(b, s); }
};

// timed section helpers for bench.py's roofline (HIP events on the context's stream)
struct KbjTimed {
  kbj_ctx* ctx; bool nn;
  KbjTimed(kbj_ctx* c, bool is_nn) : ctx(c), nn(is_nn) { if (ctx->profiling) hipEventRecord(ctx->ev0, ctx->stream); }
  ~KbjTimed() {
    if (!ctx->profiling) return;
    hipEventRecord(ctx->ev1, ctx->stream);
    hipEventSynchronize(ctx->ev1);
    float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (nn) { ctx->nn_ms += ms; ctx->nn_launches++; } else { ctx->env_ms += ms; ctx->env_launches++; }
  }
};
