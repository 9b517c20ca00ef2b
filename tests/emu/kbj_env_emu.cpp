// Host emulation build of the HIP env-step kernel body (kbot-joystick_amd/csrc/kbj_env_*.h with -DKBJ_EMU).
// TEST INFRASTRUCTURE ONLY: lets tests/ run the exact lane-parallel algorithm of the GPU kernel on the CPU
// (one "lane", butterfly reductions replayed in GPU order) and diff it against the oracle without a GPU.
// The product library never contains this code path.
#define KBJ_EMU 1
#include "../../kbot-joystick_amd/csrc/kbj_env_task.h"
#include <cstring>
#include <memory>

using namespace kbj;

static PhysConst make_pc(const kbj_config* c, const kbj_model* m) { return phys_const(*c, *m); }

extern "C" {

void kbj_emu_reset_all(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, float* a0, float* c0, float* x0) {
  PhysConst pc = make_pc(c, m);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < c->num_envs; ++i) {
    std::unique_ptr<KbjShared> S(new KbjShared());
    std::memset(S.get(), 0, sizeof(KbjShared));
    model_lds_fill(S->mc, *m);
    S->pc = pc;
    Rng rng{seed, (uint32_t)(c->env_id_offset + i)};
    task_reset(*S, *m, *c, S->pc, rng);
    task_write_obs(*S, *m, *c, rng, a0 + (size_t)i * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor), c0 + (size_t)i * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), x0 + (size_t)i * KBJ_AUX_SIZE);
    std::memcpy(ep + (size_t)i * KBJ_EP_SIZE, S->ep, sizeof(S->ep));
    std::memcpy(es + (size_t)i * KBJ_ES_SIZE, S->es, sizeof(S->es));
  }
}

void kbj_emu_env_step(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, const float* action, float* aux_t,
                      float* an, float* cn, float* xn) {
  PhysConst pc = make_pc(c, m);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < c->num_envs; ++i) {
    std::unique_ptr<KbjShared> S(new KbjShared());
    std::memset(S.get(), 0, sizeof(KbjShared));
    model_lds_fill(S->mc, *m);
    S->pc = pc;
    std::memcpy(S->ep, ep + (size_t)i * KBJ_EP_SIZE, sizeof(S->ep));
    std::memcpy(S->es, es + (size_t)i * KBJ_ES_SIZE, sizeof(S->es));
    Rng rng{seed, (uint32_t)(c->env_id_offset + i)};
    task_step(*S, *m, *c, S->pc, rng, action + (size_t)i * KBJ_NU, aux_t + (size_t)i * KBJ_AUX_SIZE, an + (size_t)i * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor),
              cn + (size_t)i * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), xn + (size_t)i * KBJ_AUX_SIZE);
    std::memcpy(ep + (size_t)i * KBJ_EP_SIZE, S->ep, sizeof(S->ep));
    std::memcpy(es + (size_t)i * KBJ_ES_SIZE, S->es, sizeof(S->es));
  }
}

int kbj_emu_shared_bytes(void) { return (int)sizeof(KbjShared); }
}
