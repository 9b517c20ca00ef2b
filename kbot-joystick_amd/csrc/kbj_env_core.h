// kbj_env_core.h — the per-env control step of the K-Bot joystick task, one wavefront = one env.
//
// Hot-path rows a1-a3, a14-a22 of SURVEY.md §8: rigid-body step (restating what the reference gets from
// mujoco-mjx via ksim's engine; configured at train.py:1775-1781), PD actuators (train.py:1091-1105),
// observations (train.py:1155-1204, 509-707, 1329-1433), command (train.py:710-785), terminations
// (train.py:788-823, 1258-1269), resets (train.py:826-844, 1146-1153), randomisers (train.py:1107-1132)
// and push events (train.py:1134-1144).
//
// Execution model: a workgroup is ONE 64-lane wavefront and owns one environment. All per-body / per-dof /
// per-constraint-row state of that env is staged in LDS (struct KbjShared, 13.4 KB); the code is a sequence
// of "phases": `PFOR(i, n)` spreads n independent work items over the 64 lanes, `KBJ_SYNC()` separates
// phases that communicate through LDS, `wsum*()` are butterfly wave reductions. HBM traffic is only the
// env's parameter/state rows at entry/exit and the observation rows, all lane-contiguous (coalesced).
//
// The same source compiles as a scalar host emulation (KBJ_EMU: 1 "lane", butterfly reductions replayed in
// the same order) used by tests/ to debug the lane-parallel algorithm without a GPU. The emulation is test
// infrastructure: the product library only ever launches the HIP kernel.
#pragma once
#include <stdint.h>
#include <math.h>
#include "kbj_model.h"

#ifdef KBJ_EMU
#define KBJ_DEV static inline
#define KBJ_LANE 0
#define KBJ_NLANE 1
#define KBJ_SYNC() ((void)0)
#else
#define KBJ_DEV __device__ __forceinline__
#define KBJ_LANE ((int)threadIdx.x)
#define KBJ_NLANE 64
#define KBJ_SYNC() __syncthreads()
#endif
#define PFOR(i, n) for (int i = KBJ_LANE; i < (n); i += KBJ_NLANE)
#ifdef KBJ_EMU
#define KBJ_HD static inline
#else
#define KBJ_HD __host__ __device__ inline
#endif

namespace kbj {

constexpr int NB = KBJ_NBODY, NQ = KBJ_NQ, NV = KBJ_NV, NU = KBJ_NU, NCON = KBJ_NCON;
constexpr int NROW = 72, ROW_LIM = 20, ROW_CON = 40;  // frictionloss | limits | pyramidal contacts

// ---- fixed kbot topology (validated against the model blob in kbj_create) ----
KBJ_DEV int body_parent(int b) { return b <= 1 ? 0 : (b == 2 ? 1 : ((b - 3) % 5 == 0 ? 2 : b - 1)); }
KBJ_DEV int dof_body(int d) { return d < 6 ? 1 : d - 3; }
KBJ_DEV int dof_parent(int d) { return d == 0 ? -1 : (d < 6 ? d - 1 : ((d - 6) % 5 == 0 ? 5 : d - 1)); }

// The model constants the per-substep phases read (same field names as kbj_model, so the phases are written against either): a
// per-workgroup LDS copy, loaded once per launch with coalesced reads. In global memory every one of these reads was a dependent
// vector load (~200-500 cycles) in the middle of a serial chain (e.g. 13 per body in the kinematics walk).
struct KbjModelLds {
  float body_pos[NB][3], body_quat[NB][4], jnt_axis[NB][3];
  float cap_axis[KBJ_NCAP][3];
  float act_range[NU][2], dof_range[NV][2], dof_invweight0[NV], body_invweight0[NB][2];
  float site_pos[2][3], site_size[2][3], imu_quat[4];
  float fric_solref[2], fric_solimp[5], limit_solref[2], limit_solimp[5], contact_solref[2], contact_solimp[5];
  float gravity[3], meaninertia;
};
static inline void model_lds_fill(KbjModelLds& o, const kbj_model& m) {   // host side (kbj_create, emulation)
  for (int b = 0; b < NB; ++b) {
    for (int k = 0; k < 3; ++k) { o.body_pos[b][k] = m.body_pos[b][k]; o.jnt_axis[b][k] = m.jnt_axis[b][k]; }
    for (int k = 0; k < 4; ++k) o.body_quat[b][k] = m.body_quat[b][k];
    o.body_invweight0[b][0] = m.body_invweight0[b][0]; o.body_invweight0[b][1] = m.body_invweight0[b][1];
  }
  for (int c = 0; c < KBJ_NCAP; ++c) for (int k = 0; k < 3; ++k) o.cap_axis[c][k] = m.cap_axis[c][k];
  for (int u = 0; u < NU; ++u) { o.act_range[u][0] = m.act_range[u][0]; o.act_range[u][1] = m.act_range[u][1]; }
  for (int d = 0; d < NV; ++d) { o.dof_range[d][0] = m.dof_range[d][0]; o.dof_range[d][1] = m.dof_range[d][1]; o.dof_invweight0[d] = m.dof_invweight0[d]; }
  for (int f = 0; f < 2; ++f) for (int k = 0; k < 3; ++k) { o.site_pos[f][k] = m.site_pos[f][k]; o.site_size[f][k] = m.site_size[f][k]; }
  for (int k = 0; k < 4; ++k) o.imu_quat[k] = m.imu_quat[k];
  for (int k = 0; k < 2; ++k) { o.fric_solref[k] = m.fric_solref[k]; o.limit_solref[k] = m.limit_solref[k]; o.contact_solref[k] = m.contact_solref[k]; }
  for (int k = 0; k < 5; ++k) { o.fric_solimp[k] = m.fric_solimp[k]; o.limit_solimp[k] = m.limit_solimp[k]; o.contact_solimp[k] = m.contact_solimp[k]; }
  for (int k = 0; k < 3; ++k) o.gravity[k] = m.gravity[k];
  o.meaninertia = m.meaninertia;
}

// solref / solimp of one constraint family, reduced ONCE PER CONTEXT (on the host, kbj_create) to what the rows need: stiffness k and damping b of the
// reference acceleration, and the clamped impedance parameters (MuJoCo's mj_makeImpedance)
struct ImpConst { float k, b, dmin, dmax, width, mid, power, iwidth, imid, i1mid; };   // + reciprocals of width, mid, 1 - mid (once per launch)
KBJ_HD ImpConst imp_const(const float* solref, const float* solimp, float dt) {
  ImpConst ic;
  ic.dmin = fminf(fmaxf(solimp[0], 0.0001f), 0.9999f); ic.dmax = fminf(fmaxf(solimp[1], 0.0001f), 0.9999f);
  ic.width = fmaxf(solimp[2], 1e-15f); ic.mid = fminf(fmaxf(solimp[3], 0.0001f), 0.9999f); ic.power = fmaxf(solimp[4], 1.0f);
  const float tc = fmaxf(solref[0], 2 * dt), dr = solref[1];
  ic.k = 1 / (ic.dmax * ic.dmax * tc * tc * dr * dr);
  ic.b = 2 / (ic.dmax * tc);
  ic.iwidth = 1 / ic.width; ic.imid = 1 / ic.mid; ic.i1mid = 1 / (1 - ic.mid);
  return ic;
}
struct PhysConst {  // constants derived from kbj_config and the model: computed on the host at kbj_create, copied into LDS by every env workgroup
                     // (inside the kernel this was ~20 IEEE divisions and a cosf per launch, i.e. per control step)
  float dt, tolerance;
  float tol2;        // Newton exit: scale sqrt(|grad|^2) < tolerance with scale = 1 / (meaninertia nv)  <=>  |grad|^2 < (tolerance meaninertia nv)^2
  float fric_ratio;  // (1 - imp) / imp of the friction-loss rows (impedance at distance 0)
  float cos_max_tilt;
  int iterations, ls_iterations;
  float tamp, tkw;   // terrain z = tamp sin(tkw x) sin(tkw y); tamp = 0: the plane z = 0
  ImpConst fric, lim, con;
};
KBJ_HD PhysConst phys_const(const kbj_config& c, const kbj_model& m) {
  PhysConst pc;
  pc.dt = c.dt; pc.tolerance = c.solver_tolerance; pc.iterations = c.solver_iterations; pc.ls_iterations = c.ls_iterations;
  pc.tamp = c.terrain_amp;
  pc.tkw = c.terrain_amp != 0 ? (float)(6.283185307179586 / c.terrain_wavelength) : 0.0f;
  pc.fric = imp_const(m.fric_solref, m.fric_solimp, c.dt);
  pc.lim = imp_const(m.limit_solref, m.limit_solimp, c.dt);
  pc.con = imp_const(m.contact_solref, m.contact_solimp, c.dt);
  pc.tol2 = (c.solver_tolerance * m.meaninertia * NV) * (c.solver_tolerance * m.meaninertia * NV);
  pc.fric_ratio = (1 - pc.fric.dmin) / pc.fric.dmin;
  pc.cos_max_tilt = cosf(c.max_tilt_rad);
  return pc;
}
// Everything one env needs between phases, 12.4 KB so that 12 single-wavefront workgroups (3 waves per SIMD) share a CU.
// Buffers whose lifetimes do not overlap share storage (union `u`): composite inertias (until the mass matrix exists),
// RNE body forces (until the bias force exists), then the arrow-matrix blocks of the Newton solves. Rotation matrices
// are recomputed from xquat where needed instead of being stored; the mass matrix is stored in its tree sparsity.
struct KbjShared {
  KbjModelLds mc;
  PhysConst pc;         // per-workgroup LDS copy like mc: in SGPRs these ~45 scalars were spilled to vector lanes (v_readlane per use)
  float ep[KBJ_EP_SIZE];
  float es[KBJ_ES_SIZE];
  float xpos[NB][3], xquat[NB][4], xipos[NB][3];
  float com[4];   // tree centre of mass (subtree_com[1]) and total mass
  float com2[3];  // subtree_com[2] (everything but the base body)
  float cinert[NB][10];
  float cdof[NV][6], cvel[NB][6];
  union {
    struct { float crb[NB][10]; };
    struct { float cfrc[NB][6], cfrc_acc[NB][6]; };
    struct { float jp[NCON][11][3]; };   // contact-frame point Jacobians (normal, tangent 1, tangent 2) while the constraint rows are built
#if defined(KBJ_ARROW_LDS)
    struct {
      float A[4][12][12];  // chain-local blocks [ankle..hip | base 6], row 11 = right-hand side
      float B[7][8];       // base block, row 6 = right-hand side
    };                     // (LDS formulation of arrow_solve only; the product kernel keeps the blocks in registers)
#endif
  } u;
  float Mb[6][6];       // mass matrix, base block
  float Mc[4][5][11];   // limb c, dof a (hip..ankle): columns 0..5 base dofs, 6..10 the limb's own dofs
  float zrow[12];       // zeros (directly behind Mc: the solver's per-lane offset table points here for the entries a lane does not have)
  float qfrc_act[NV], qfrc_smooth[NV], qacc[NV];
#if defined(KBJ_ARROW_LDS)   // the LDS formulation of the solver keeps its vectors here; the product kernel keeps them in registers
  float qacc_smooth[NV], Ma[NV], grad[NV], search[NV], mv[NV], vec[NV];
  float jar[NROW], jv[NROW];
#endif
  float conpos[NCON][3], condist[NCON], connrm[NCON][3];
  int conact[NCON];
  float Jc[32][11];  // contact rows: columns 0..5 base dofs, 6..10 the leg's dofs hip..ankle
  float D[NROW], aref[NROW], force[NROW];  // a row is active iff D != 0
  float Rf[NU], floss[NU], lsign[NU];                           // Huber half-width data of the frictionloss rows, limit signs
  int quad[NROW];
  float ctrl[NU], push[6], act_eff[NU];
  float gyro[3], imuquat[4], touch[2], pg[3];
  int iters, pushing, done;
};

// symmetric access to the tree-sparse mass matrix (i, j must be related: same limb, or one of them a base dof)
KBJ_DEV float& M_at(KbjShared& S, int i, int j) {
  if (i < j) { int t = i; i = j; j = t; }
  if (i < 6) return S.Mb[i][j];
  int c = (i - 6) / 5, a = (i - 6) % 5;
  return S.Mc[c][a][j < 6 ? j : 6 + (j - 6) % 5];
}
KBJ_DEV float M_get(const KbjShared& S, int i, int j) { return M_at(const_cast<KbjShared&>(S), i, j); }

// ---- small vector helpers ----
KBJ_DEV void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
KBJ_DEV void quat_mul(const float* a, const float* b, float* o) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
KBJ_DEV void quat_norm(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}
KBJ_DEV void quat_to_mat(const float* q, float* m) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
KBJ_DEV void mat_vec(const float* m, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = m[3 * i] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
}
KBJ_DEV void matT_vec(const float* m, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = m[i] * v[0] + m[3 + i] * v[1] + m[6 + i] * v[2];
}
KBJ_DEV void inert_mul(const float* I, const float* v, float* o) {
  o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  o[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  o[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  o[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
KBJ_DEV void cross_motion(const float* vel, const float* v, float* o) {
  float t1[3], t2[3];
  cross3(vel, v, o); cross3(vel, v + 3, t1); cross3(vel + 3, v, t2);
  for (int k = 0; k < 3; ++k) o[3 + k] = t1[k] + t2[k];
}
KBJ_DEV void cross_force(const float* vel, const float* f, float* o) {
  float t1[3], t2[3];
  cross3(vel, f, t1); cross3(vel + 3, f + 3, t2);
  for (int k = 0; k < 3; ++k) o[k] = t1[k] + t2[k];
  cross3(vel, f + 3, o + 3);
}
KBJ_DEV void quat_to_euler(const float* q, float* e) {  // xax.quat_to_euler
  float w = q[0], x = q[1], y = q[2], z = q[3];
  e[0] = atan2f(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  float sp = fminf(1.0f, fmaxf(-1.0f, 2 * (w * y - z * x)));
  e[1] = asinf(sp);
  e[2] = atan2f(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}
KBJ_DEV void euler_to_quat(const float* e, float* q) {
  float cr = cosf(e[0] / 2), sr = sinf(e[0] / 2), cp = cosf(e[1] / 2), sp = sinf(e[1] / 2), cy = cosf(e[2] / 2), sy = sinf(e[2] / 2);
  q[0] = cr * cp * cy + sr * sp * sy; q[1] = sr * cp * cy - cr * sp * sy; q[2] = cr * sp * cy + sr * cp * sy; q[3] = cr * cp * sy - sr * sp * cy;
}
KBJ_DEV void rotate_by_quat(const float* v, const float* q_in, bool inverse, float* o) {
  float q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]}, mat[9];
  quat_norm(q);
  if (inverse) { q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  quat_to_mat(q, mat); mat_vec(mat, v, o);
}

// ---- wave reductions ------------------------------------------------------------------------------------------
// Sum of one value per lane. Order (identical on the GPU and in the emulation, so both give the same bits):
// within each row of 16 lanes: v += v[l^1]; v += v[l^2]; v += v[mirror in 8]; v += v[mirror in 16]  (DPP quad_perm /
// row_half_mirror / row_mirror: register-to-register, no LDS), then ((row0 + row1) + row2) + row3 via v_readlane.
#ifdef KBJ_EMU
KBJ_DEV float wsum_lanes(float* v) {
  float t[64];
  for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ 1];
  for (int l = 0; l < 64; ++l) v[l] = t[l] + t[l ^ 2];
  for (int l = 0; l < 64; ++l) t[l] = v[l] + v[(l & ~7) | (7 - (l & 7))];
  for (int l = 0; l < 64; ++l) v[l] = t[l] + t[(l & ~15) | (15 - (l & 15))];
  return ((v[0] + v[16]) + v[32]) + v[48];
}
template <class F> KBJ_DEV float wsum(int n, F f) {
  float v[64];
  for (int l = 0; l < 64; ++l) v[l] = l < n ? f(l) : 0.0f;
  return wsum_lanes(v);
}
template <class F> KBJ_DEV void wsum2(int n, F f, float& a, float& b) {
  float va[64], vb[64];
  for (int l = 0; l < 64; ++l) { va[l] = 0; vb[l] = 0; if (l < n) f(l, va[l], vb[l]); }
  a = wsum_lanes(va); b = wsum_lanes(vb);
}
#else
template <int CTRL> KBJ_DEV float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(t);
}
KBJ_DEV float wsum_lanes(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return ((r0 + r1) + r2) + r3;
}
template <class F> KBJ_DEV float wsum(int n, F f) {
  int l = KBJ_LANE;
  return wsum_lanes(l < n ? f(l) : 0.0f);
}
template <class F> KBJ_DEV void wsum2(int n, F f, float& a, float& b) {
  int l = KBJ_LANE;
  float va = 0, vb = 0;
  if (l < n) f(l, va, vb);
  a = wsum_lanes(va); b = wsum_lanes(vb);
}
#endif

// ---- threefry2x32-20 counter RNG (a25: the generator behind jax.random; Random123 KATs in tests) ----
KBJ_DEV uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
KBJ_DEV void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const int rot[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = rotl32(x1, rot[(g & 1) * 4 + r]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  o0 = x0; o1 = x1;
}
struct Rng { uint32_t seed, env; };
KBJ_DEV void rng_bits(const Rng& r, int stream, uint32_t a, uint32_t b, uint32_t& o0, uint32_t& o1) {
  threefry2x32(r.seed ^ ((uint32_t)stream * 0x9E3779B9u), r.env, a, b, o0, o1);
}
// uniform in [0,1) on 24 bits (exact in fp32)
KBJ_DEV float rng_u01(const Rng& r, int stream, uint32_t a, uint32_t b) {
  uint32_t x, y; rng_bits(r, stream, a, b, x, y);
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// lo + (hi - lo) u with one rounding (explicit fma so the oracle reproduces it bit for bit)
KBJ_DEV float rng_uniform(const Rng& r, int stream, uint32_t a, uint32_t b, float lo, float hi) {
  return fmaf(hi - lo, rng_u01(r, stream, a, b), lo);
}
KBJ_DEV float rng_normal(const Rng& r, int stream, uint32_t a, uint32_t b) {  // Box-Muller
  uint32_t x, y; rng_bits(r, stream, a, b, x, y);
  float u1 = (float)((x >> 8) + 1u) * (1.0f / 16777216.0f), u2 = (float)(y >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}


// ---- jax.random key handling (a25; kbj_config.command_mode == 2) -------------------------------------------------------------------
// The in-tree samplers of the reference (UnifiedCommand train.py:725-752, 782-785; PlaneXYPositionReset train.py:834-836) are written against
// jax.random: `split`, `uniform`, `bernoulli`, `randint` on threefry2x32 keys. In this mode the kernels derive those draws FROM THE KEY THE
// SAMPLER IS CALLED WITH exactly as jax 0.6.0 does by default (requirements.lock:72; jax_threefry_partitionable = True since jax 0.5):
//   split(key, n)[i]              = threefry2x32(key, counter (0, i))                                   (both output words = the new key)
//   random_bits(key, 32, shape)[i] = x0 ^ x1 of threefry2x32(key, counter (0, i))   (shape () = index 0)
//   uniform(key, shape, lo, hi)   = max(lo, u * (hi - lo) + lo), u = bitcast((bits >> 9) | 0x3F800000) - 1    (mantissa fill; mul and add round separately)
//   bernoulli(key, p, shape)      = uniform(key, shape) < p
//   randint(key, (), 0, span)     = ((hi % span) * (2^32 % span) + lo % span) % span with hi, lo = random_bits of split(key)[0], [1]
// What stays this build's own is the tree ABOVE the call (how ksim's engine derives the per-env, per-step key it passes in: un-vendored): the
// call key is threefry(seed ^ stream, env; counters) as everywhere else. Pinned offline by the public known answers of jax.random (split and
// uniform of PRNGKey(0), tests/test_oracle_task.py) on the Python restatement oracle/jax_random.py, which the C++ oracle and these kernels
// are compared with; against a live JAX the mode is UNVERIFIED (no JAX in the image).
struct JaxKey { uint32_t k0, k1; };
#ifdef KBJ_EMU
KBJ_DEV float mul_add_2r(float a, float b, float c) { volatile float p = a * b; return p + c; }
#else
KBJ_DEV float mul_add_2r(float a, float b, float c) {   // never contracted into one fma (HIP's __fmul_rn / __fadd_rn are plain operators: they would be)
#pragma clang fp contract(off)
  const float p = a * b;
  return p + c;
}
#endif
KBJ_DEV JaxKey jax_split(const JaxKey& k, uint32_t i) { JaxKey o; threefry2x32(k.k0, k.k1, 0u, i, o.k0, o.k1); return o; }
KBJ_DEV uint32_t jax_bits(const JaxKey& k, uint32_t i) { uint32_t a, b; threefry2x32(k.k0, k.k1, 0u, i, a, b); return a ^ b; }
KBJ_DEV float jax_u01(const JaxKey& k, uint32_t i) { union { float f; uint32_t u; } x; x.u = (jax_bits(k, i) >> 9) | 0x3F800000u; return x.f - 1.0f; }
KBJ_DEV float jax_uniform(const JaxKey& k, uint32_t i, float lo, float hi) { return fmaxf(lo, mul_add_2r(jax_u01(k, i), hi - lo, lo)); }
KBJ_DEV uint32_t jax_randint(const JaxKey& k, uint32_t span) {
  const JaxKey k1 = jax_split(k, 0), k2 = jax_split(k, 1);
  const uint32_t hb = jax_bits(k1, 0), lb = jax_bits(k2, 0);
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  return ((hb % span) * mult + lb % span) % span;
}
KBJ_DEV JaxKey jax_call_key(const Rng& r, int stream, uint32_t a, uint32_t b) { JaxKey k; rng_bits(r, stream, a, b, k.k0, k.k1); return k; }

}  // namespace kbj
