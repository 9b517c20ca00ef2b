// kbj_nn.hip — actor-critic kernels (entry points stubbed until the MFMA path lands later this round;
// every stub fails loudly, nothing falls back to the CPU)
#include <hip/hip_runtime.h>
#include "kbj_ctx.h"
int kbj_nn_create(kbj_ctx*) { return 0; }
void kbj_nn_destroy(kbj_ctx*) {}
#define NI(ctx, name) return kbj_fail(ctx, name ": not implemented yet")
extern "C" {
size_t kbj_param_count(const kbj_config*) { return 0; }
size_t kbj_actor_param_count(const kbj_config*) { return 0; }
int kbj_init_params(kbj_ctx* c, uint32_t, float*) { NI(c, "kbj_init_params"); }
int kbj_policy_step(kbj_ctx* c, const float*, const float*, const float*, kbj_carry*, uint32_t, uint32_t, int, float*, float*, float*) { NI(c, "kbj_policy_step"); }
int kbj_carry_reset(kbj_ctx* c, kbj_carry*, const float*, int) { NI(c, "kbj_carry_reset"); }
int kbj_rollout(kbj_ctx* c, const float*, kbj_carry*, uint32_t, uint32_t, kbj_traj*) { NI(c, "kbj_rollout"); }
int kbj_gae(kbj_ctx* c, const kbj_traj*, float*, float*) { NI(c, "kbj_gae"); }
int kbj_ppo_grad(kbj_ctx* c, const float*, const kbj_traj*, const int32_t*, int, const float*, const float*, float*, float*) { NI(c, "kbj_ppo_grad"); }
int kbj_adamw_step(kbj_ctx* c, float*, float*, float*, const float*, int64_t, float) { NI(c, "kbj_adamw_step"); }
}
