// kbj_ctx.h — library context shared by the translation units of libkbj.so (host side only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>
#include "kbj.h"

struct kbj_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second lane for the critic network inside kbj_ppo_grad
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t side[2] = {nullptr, nullptr};   // per-net side lanes for weight-gradient GEMMs
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  kbj_model model_h;
  kbj_config cfg_h;
  kbj_model* model_d = nullptr;
  kbj_config* cfg_d = nullptr;
  float* ep_d = nullptr;        // [N][KBJ_EP_SIZE]
  float* es_d = nullptr;        // [N][KBJ_ES_SIZE]
  float* rcarry_d = nullptr;    // [N][KBJ_RC_SIZE] reward carries
  uint32_t seed = 0;
  // NN workspace (kbj_nn.hip)
  void* nn_ws = nullptr;
  size_t nn_ws_bytes = 0;
  // profiling
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float env_ms = 0, nn_ms = 0;
  int env_launches = 0, nn_launches = 0;
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
