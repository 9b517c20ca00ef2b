// kbj_oracle.cpp — TEST INFRASTRUCTURE ONLY. C entry points of the CPU oracle (fp32: kbj_cpu_*, fp64: kbj_cpu64_*).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (kbot-joystick_amd/csrc -> libkbj.so) never links or calls it. Parity with the JAX reference is UNPINNED
// (no reference tests/golden vectors; dependencies not installable offline — SURVEY.md §8c); the oracle is
// pinned by the analytic known-answer tests in tests/test_oracle_*.py.
#include "kbj_oracle_task.h"
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace kbjo;

namespace {

template <class R>
void reset_all(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, float* actor0, float* critic0, float* aux0) {
  int N = c->num_envs;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    Env<R>* e = new Env<R>();
    float* esi = es + (size_t)i * KBJ_ES_SIZE;
    for (int k = 0; k < KBJ_ES_SIZE; ++k) esi[k] = 0;
    e->bind(m, c, seed, c->env_id_offset + i, ep + (size_t)i * KBJ_EP_SIZE, esi);
    e->reset();
    e->write_obs(actor0 + (size_t)i * KBJ_LD_ACTOR, critic0 + (size_t)i * KBJ_LD_CRITIC, aux0 + (size_t)i * KBJ_AUX_SIZE);
    delete e;
  }
}

template <class R>
void env_step(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, const float* action, float* aux_t,
              float* actor_next, float* critic_next, float* aux_next, int* diag = nullptr) {
  int N = c->num_envs;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    Env<R>* e = new Env<R>();
    e->bind(m, c, seed, c->env_id_offset + i, ep + (size_t)i * KBJ_EP_SIZE, es + (size_t)i * KBJ_ES_SIZE);
    if (diag) { e->diag = diag + 4 * (size_t)i; for (int k = 0; k < 4; ++k) e->diag[k] = 0; }
    e->step(action + (size_t)i * KBJ_NU, aux_t + (size_t)i * KBJ_AUX_SIZE, actor_next + (size_t)i * KBJ_LD_ACTOR,
            critic_next + (size_t)i * KBJ_LD_CRITIC, aux_next + (size_t)i * KBJ_AUX_SIZE);
    delete e;
  }
}

template <class R>
void rewards(const kbj_model* m, const kbj_config* c, const float* aux, int T, int N, float* carry, float* reward, float* comps) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i)
    rewards_scan<R>(m, c, aux + (size_t)i * KBJ_AUX_SIZE, (size_t)N * KBJ_AUX_SIZE, T, carry + (size_t)i * KBJ_RC_SIZE, reward + i, (size_t)N,
                    comps ? comps + (size_t)i * KBJ_NREW : nullptr, (size_t)N * KBJ_NREW);
}

// layout of the debug dump of one forward pass (doubles)
enum { DBG_XPOS = 0, DBG_XQUAT = 72, DBG_M = 168, DBG_QACC = 844, DBG_QACC_SMOOTH = 870, DBG_BIAS = 896, DBG_EFC_FORCE = 922,
       DBG_TOUCH = 994, DBG_GYRO = 996, DBG_SUBCOM = 999, DBG_CINERT = 1071, DBG_CVEL = 1311, DBG_CONPOS = 1455, DBG_CONDIST = 1479,
       DBG_ENERGY = 1487, DBG_ITERS = 1489, DBG_IMUQUAT = 1490, DBG_QFRC_CON = 1494, DBG_ACTFRC = 1520, DBG_EFC_ACTIVE = 1546,
       DBG_EFC_AREF = 1618, DBG_EFC_D = 1690, DBG_SIZE = 1762 };

template <class R>
void forward_dump(const kbj_model* m, const kbj_config* c, const float* ep, const double* qpos, const double* qvel, const double* ctrl,
                  const double* push, const double* warm, double* out, double* qpos_next, double* qvel_next) {
  Physics<R> phy;
  phy.m = m; phy.dt = c->dt; phy.opt.iterations = c->solver_iterations; phy.opt.ls_iterations = c->ls_iterations; phy.opt.tolerance = c->solver_tolerance; phy.opt.newton = c->solver_newton;
    phy.terrain_amp = c->terrain_amp; phy.terrain_kw = c->terrain_amp != 0 ? (float)(6.283185307179586 / c->terrain_wavelength) : 0;
  phy.p.load(ep);
  R q[NQ], v[NV], u[NU], w[NV], ps[6];
  for (int i = 0; i < NQ; ++i) q[i] = (R)qpos[i];
  for (int i = 0; i < NV; ++i) { v[i] = (R)qvel[i]; w[i] = (R)warm[i]; }
  for (int i = 0; i < NU; ++i) u[i] = (R)ctrl[i];
  if (push) for (int i = 0; i < 6; ++i) ps[i] = (R)push[i];
  Derived<R>* d = new Derived<R>();
  phy.forward(q, v, u, push ? ps : nullptr, w, *d);
  for (int b = 0; b < NB; ++b) {
    for (int k = 0; k < 3; ++k) { out[DBG_XPOS + 3 * b + k] = d->xpos[b][k]; out[DBG_SUBCOM + 3 * b + k] = d->subtree_com[b][k]; }
    for (int k = 0; k < 4; ++k) out[DBG_XQUAT + 4 * b + k] = d->xquat[b][k];
    for (int k = 0; k < 10; ++k) out[DBG_CINERT + 10 * b + k] = d->cinert[b][k];
    for (int k = 0; k < 6; ++k) out[DBG_CVEL + 6 * b + k] = d->cvel[b][k];
  }
  for (int i = 0; i < NV; ++i) {
    for (int j = 0; j < NV; ++j) out[DBG_M + NV * i + j] = d->M[i][j];
    out[DBG_QACC + i] = d->qacc[i]; out[DBG_QACC_SMOOTH + i] = d->qacc_smooth[i]; out[DBG_BIAS + i] = d->qfrc_bias[i];
    out[DBG_QFRC_CON + i] = d->qfrc_constraint[i]; out[DBG_ACTFRC + i] = d->qfrc_actuator[i];
  }
  for (int r = 0; r < NEFC; ++r) {
    out[DBG_EFC_FORCE + r] = d->efc_force[r]; out[DBG_EFC_ACTIVE + r] = d->efc_active[r];
    out[DBG_EFC_AREF + r] = d->efc_aref[r]; out[DBG_EFC_D + r] = d->efc_D[r];
  }
  out[DBG_TOUCH] = d->touch[0]; out[DBG_TOUCH + 1] = d->touch[1];
  for (int k = 0; k < 3; ++k) out[DBG_GYRO + k] = d->gyro[k];
  for (int k = 0; k < 4; ++k) out[DBG_IMUQUAT + k] = d->imu_quat[k];
  for (int ci = 0; ci < NCON; ++ci) { for (int k = 0; k < 3; ++k) out[DBG_CONPOS + 3 * ci + k] = d->con_pos[ci][k]; out[DBG_CONDIST + ci] = d->con_dist[ci]; }
  out[DBG_ENERGY] = phy.kinetic_energy(*d, v); out[DBG_ENERGY + 1] = phy.potential_energy(*d);
  out[DBG_ITERS] = d->solver_iters;
  if (qpos_next) {
    phy.integrate(q, v, *d);
    for (int i = 0; i < NQ; ++i) qpos_next[i] = q[i];
    for (int i = 0; i < NV; ++i) qvel_next[i] = v[i];
  }
  delete d;
}

}  // namespace

extern "C" {

int kbj_cpu_sizeof_model(void) { return (int)sizeof(kbj_model); }
int kbj_cpu_sizeof_config(void) { return (int)sizeof(kbj_config); }
int kbj_cpu_dbg_size(void) { return DBG_SIZE; }
int kbj_cpu_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void kbj_cpu_set_num_threads(int n) {   // bench.py pins the thread count of the timed CPU baseline
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}
void kbj_cpu_threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* out) { threefry2x32(k0, k1, c0, c1, out[0], out[1]); }

#define KBJ_EXPORTS(SUF, R)                                                                                                            \
  void kbj_cpu##SUF##_reset_all(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, float* a0, float* c0,    \
                                float* x0) { reset_all<R>(m, c, seed, ep, es, a0, c0, x0); }                                          \
  void kbj_cpu##SUF##_env_step(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, const float* action,      \
                               float* aux_t, float* an, float* cn, float* xn) { env_step<R>(m, c, seed, ep, es, action, aux_t, an, cn, xn); } \
  /* same step + per-env int[4] diagnostics: max / total Newton iterations over the substeps, hashes of the active-contact history */ \
  /* and of the force-carrying-row history (tools/parity_quantiles.py: which env-steps sit on a discrete switch) */                   \
  void kbj_cpu##SUF##_env_step_diag(const kbj_model* m, const kbj_config* c, uint32_t seed, float* ep, float* es, const float* action, \
                                    float* aux_t, float* an, float* cn, float* xn, int* diag) {                                      \
    env_step<R>(m, c, seed, ep, es, action, aux_t, an, cn, xn, diag);                                                                 \
  }                                                                                                                                    \
  void kbj_cpu##SUF##_rewards(const kbj_model* m, const kbj_config* c, const float* aux, int T, int N, float* carry, float* reward,    \
                              float* comps) { rewards<R>(m, c, aux, T, N, carry, reward, comps); }                                     \
  void kbj_cpu##SUF##_forward(const kbj_model* m, const kbj_config* c, const float* ep, const double* qpos, const double* qvel,        \
                              const double* ctrl, const double* push, const double* warm, double* out, double* qn, double* vn) {        \
    forward_dump<R>(m, c, ep, qpos, qvel, ctrl, push, warm, out, qn, vn);                                                              \
  }                                                                                                                                    \
  double kbj_cpu##SUF##_com_distance(const double* pts, const double* com) {                                                           \
    R p[NCON][3], cm[2] = {(R)com[0], (R)com[1]};                                                                                      \
    for (int i = 0; i < NCON; ++i) for (int k = 0; k < 3; ++k) p[i][k] = (R)pts[3 * i + k];                                             \
    return (double)com_distance<R>(p, cm);                                                                                             \
  }

KBJ_EXPORTS(, float)
KBJ_EXPORTS(64, double)

// default per-env parameter row without randomisation (used by KATs)
void kbj_cpu_default_params(const kbj_model* m, const kbj_config* c, float* ep) {
  kbj_config cc = *c;
  cc.enable_randomizers = 0; cc.enable_noise = 0;
  float es[KBJ_ES_SIZE] = {0};
  Env<double> e;
  e.bind(m, &cc, 0, 0, ep, es);
  e.randomize();
}

// diagnostics: copy (and optionally clear) the histogram of solver iterations per forward pass
void kbj_cpu_solver_hist(long long* out16, int clear) {
  for (int k = 0; k < 16; ++k) { out16[k] = g_solver_hist[k]; if (clear) g_solver_hist[k] = 0; }
}

}  // extern "C"
