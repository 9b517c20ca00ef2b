// kbj_wave.h — wave-level primitives of the register-resident constraint solver (kbj_env_phys.h phys_solve).
//
// The solver keeps everything in vector registers and talks across lanes with DPP (`row_newbcast`, butterflies, `row_bcast:15/31`) and
// the gfx950 lane swaps (`v_permlane16_swap`, `v_permlane32_swap`). It is written ONCE against the small vocabulary below:
//   * a wave value `WF` is one float per lane. On the GPU that is a plain `float` (the code is the lane's code); in the host emulation
//     (KBJ_EMU) it is an array of 64 floats and `WLANES(l) { ... WL(x, l) ... }` runs the per-lane statements for every lane;
//   * every cross-lane operation is a function here with a GPU body (DPP builtin or one inline-asm instruction) and an emulation body
//     that moves the same lanes and adds in the same order, so the emulation reproduces the kernel's arithmetic up to `v_rcp_f32`.
// The env kernel is bound by vector-instruction issue (a wave64 instruction holds its SIMD for 4 cycles), so the primitives are chosen
// by instruction count: a broadcast feeding an FMA is ONE `v_fmac_f32_dpp` (the compiler cannot fuse it: its DPP combiner runs while the
// FMA is still the three-address VOP3 form), a select on a lane role is ONE `v_cndmask` on a constant lane mask (s_mov + inverse ballot),
// two wave sums share one reduction tree, two cross-row sums share their lane swaps.
//
// Hazards: the hazard recogniser does not look inside inline asm, and a DPP read of a VGPR written by one of the two preceding vector
// instructions is one it would otherwise pad. Every asm statement that reads through DPP therefore starts with `s_nop 1` (no vector
// issue slot). All of these run with every lane enabled (the solver's control flow is wave-uniform).
#pragma once
#include "kbj_env_core.h"
#include <type_traits>

namespace kbj {

template <int I, int N, class F> KBJ_DEV void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

#ifdef KBJ_EMU
struct WF { float v[64]; };
#define WLANES(l) for (int l = 0; l < 64; ++l)
#define WL(x, l) ((x).v[l])
#else
typedef float WF;
#define WLANES(l) for (int l = KBJ_LANE, kbj_once_ = 1; kbj_once_; kbj_once_ = 0)
#define WL(x, l) (x)
#endif

// lane-role masks (bit l = lane l): lane 16 c + r of DPP row c
constexpr unsigned long long wmask_r(int r) { return 0x0001000100010001ull << r; }                                   // lanes with r == this
constexpr unsigned long long wmask_r_below(int n) { return n >= 16 ? ~0ull : 0x0001000100010001ull * ((1ull << n) - 1); }  // lanes with r < n

#ifdef KBJ_EMU
// ---- host emulation: same lanes, same order of additions ----
template <unsigned long long MASK> KBJ_DEV WF wsel(const WF& a, const WF& b) { WF o; for (int l = 0; l < 64; ++l) o.v[l] = ((MASK >> l) & 1) ? a.v[l] : b.v[l]; return o; }
template <unsigned long long MASK> KBJ_DEV WF wsel0(const WF& a) { WF o; for (int l = 0; l < 64; ++l) o.v[l] = ((MASK >> l) & 1) ? a.v[l] : 0.0f; return o; }
template <int J> KBJ_DEV WF wbcast(const WF& v) { WF o; for (int l = 0; l < 64; ++l) o.v[l] = v.v[(l & ~15) | J]; return o; }
template <int J> KBJ_DEV void wfmac_bcast(WF& acc, const WF& src, const WF& mul) { for (int l = 0; l < 64; ++l) acc.v[l] = fmaf(src.v[(l & ~15) | J], mul.v[l], acc.v[l]); }
template <int J> KBJ_DEV WF wmul_bcast(const WF& src, const WF& mul) { WF o; for (int l = 0; l < 64; ++l) o.v[l] = src.v[(l & ~15) | J] * mul.v[l]; return o; }
template <int J, int ROWMASK> KBJ_DEV void wset_rhs(WF& dst, const WF& g) {   // lanes 12..15 of the rows in ROWMASK <- lane J of their row
  for (int l = 0; l < 64; ++l) if ((l & 15) >= 12 && ((ROWMASK >> (l >> 4)) & 1)) dst.v[l] = g.v[(l & ~15) | J];
}
#ifdef KBJ_EMU_RCP_1ULP   // model of v_rcp_f32 (1 ulp): the rounded reciprocal moved by -1 / 0 / +1 ulp, pseudo-randomly from the argument's bits
KBJ_DEV float wrcp_scalar(float x) {
  union { float f; unsigned u; } a, r; a.f = x; r.f = 1.0f / x;
  r.u += ((a.u >> 3) & 1u) - ((a.u >> 5) & 1u);
  return r.f;
}
#else
KBJ_DEV float wrcp_scalar(float x) { return 1.0f / x; }
#endif
KBJ_DEV float wclamp(float x, float t) { return fminf(fmaxf(x, -t), t); }
KBJ_DEV float wmin0(float x) { return fminf(x, 0.0f); }
KBJ_DEV void wopaque(WF&) {}
KBJ_DEV WF wrow_sum16(const WF& x) {
  WF v = x, t;
  for (int l = 0; l < 64; ++l) t.v[l] = v.v[l] + v.v[l ^ 1];
  for (int l = 0; l < 64; ++l) v.v[l] = t.v[l] + t.v[l ^ 2];
  for (int l = 0; l < 64; ++l) t.v[l] = v.v[l] + v.v[(l & ~7) | (7 - (l & 7))];
  for (int l = 0; l < 64; ++l) v.v[l] = t.v[l] + t.v[(l & ~15) | (15 - (l & 15))];
  return v;
}
KBJ_DEV WF wrows_sum1(const WF& x) {   // (row0 + row1) + (row2 + row3), same lane-in-row, in every row
  WF o;
  for (int l = 0; l < 64; ++l) { const int i = l & 15; o.v[l] = (x.v[i] + x.v[16 + i]) + (x.v[32 + i] + x.v[48 + i]); }
  return o;
}
KBJ_DEV void wrows_sum2(WF& x, WF& y) { x = wrows_sum1(x); y = wrows_sum1(y); }
KBJ_DEV float wsum(const WF& x) {      // (s2 + s3) + (s0 + s1) of the four row sums
  const WF v = wrow_sum16(x);
  return (v.v[48] + v.v[32]) + (v.v[16] + v.v[0]);
}
KBJ_DEV void wsum2(const WF& x1, const WF& x2, float& s1, float& s2) {   // lanes l and l + 32 first, then the rows of 16, then the two rows
  WF p;
  for (int l = 0; l < 32; ++l) { p.v[l] = x1.v[l] + x1.v[l + 32]; p.v[l + 32] = x2.v[l] + x2.v[l + 32]; }
  const WF v = wrow_sum16(p);
  s1 = v.v[16] + v.v[0]; s2 = v.v[48] + v.v[32];
}
template <class F> KBJ_DEV unsigned long long wballot(F f) { unsigned long long m = 0; for (int l = 0; l < 64; ++l) if (f(l)) m |= 1ull << l; return m; }
#else
// ---- gfx950 ----
template <unsigned long long MASK> KBJ_DEV float wsel(float a, float b) { return __builtin_amdgcn_inverse_ballot_w64(MASK) ? a : b; }
template <unsigned long long MASK> KBJ_DEV float wsel0(float a) { return __builtin_amdgcn_inverse_ballot_w64(MASK) ? a : 0.0f; }
template <int J> KBJ_DEV float wbcast(float v) {   // value of lane J of each 16-lane row, in every lane of that row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xF, 0xF, true));
}
template <int J> KBJ_DEV void wfmac_bcast(float& acc, float src, float mul) {   // acc += (lane J of the row of src) * mul
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
}
template <int J> KBJ_DEV float wmul_bcast(float src, float mul) {
  float o;
  asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "v"(mul), "n"(J));
  return o;
}
template <int J, int ROWMASK> KBJ_DEV void wset_rhs(float& dst, float g) {   // bank 3 (lanes 12..15) of the rows in ROWMASK <- lane J of their row
  dst = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(dst), __float_as_int(g), 0x150 + J, ROWMASK, 0x8, false));
}
KBJ_DEV float wrcp_scalar(float x) { return __builtin_amdgcn_rcpf(x); }
KBJ_DEV float wclamp(float x, float t) { return __builtin_amdgcn_fmed3f(x, -t, t); }   // t >= 0: one v_med3_f32
KBJ_DEV float wmin0(float x) { float o; asm("v_min_f32 %0, 0, %1" : "=v"(o) : "v"(x)); return o; }   // fminf() costs a canonicalising v_max first
// keeps the compiler from re-associating a product into the first butterfly step (it turns mul + add_dpp into mul + mov_dpp + fmac)
KBJ_DEV void wopaque(float& x) { asm("" : "+v"(x)); }
KBJ_DEV float wrow_sum16(float v) { v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v); return v; }
KBJ_DEV float wrows_sum1(float v) {
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(p[0]) + __uint_as_float(p[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
// x and y both summed over the four DPP rows (same lane-in-row) and replicated in every row: 3 lane swaps + 2 adds + 2 copies for the
// pair. swap16(x, y) leaves [x0 y0 x2 y2] / [x1 y1 x3 y3] (rows), their sum [x01 y01 x23 y23]; swap32 of that with itself and an add
// give [xt yt xt yt]; a last swap16 with itself un-interleaves into [xt xt xt xt] and [yt yt yt yt]. xt = (x0 + x1) + (x2 + x3).
KBJ_DEV void wrows_sum2(float& x, float& y) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  const float p = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
  const float s = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  x = __uint_as_float(c[0]); y = __uint_as_float(c[1]);
}
// v += dpp(v) in the rows of ROWMASK only, other rows keep v (one instruction; through the builtin it is a zero, a masked move and an add)
KBJ_DEV float dpp_add_bcast15(float v) { asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v)); return v; }
KBJ_DEV float dpp_add_bcast31(float v) { asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v)); return v; }
KBJ_DEV float wsum(float v) {   // butterflies inside the rows, row_bcast:15 into rows 1 and 3, row_bcast:31 into row 3, lane 63
  v = wrow_sum16(v);
  v = dpp_add_bcast15(v);
  v = dpp_add_bcast31(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
KBJ_DEV void wsum2(float x1, float x2, float& s1, float& s2) {   // two wave sums on one tree: x1 in the lower, x2 in the upper 32 lanes
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x1), __float_as_uint(x2), false, false);
  float v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  v = wrow_sum16(v);
  v = dpp_add_bcast15(v);
  s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <class F> KBJ_DEV unsigned long long wballot(F f) { return __builtin_amdgcn_ballot_w64(f(KBJ_LANE)); }
#endif

// ---- scalar arithmetic that replaces IEEE division / square-root sequences (10-14 vector instructions each) -----------------------
// hardware estimate + one Newton step: within an ulp of the rounded result. The emulation uses the exact operation.
KBJ_DEV float kbj_frcp(float x) {    // 1 / x
#ifdef KBJ_EMU
  return 1.0f / x;
#else
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
#endif
}
KBJ_DEV float kbj_rsqrt(float x) {   // 1 / sqrt(x): y (1 + (1 - x y^2) / 2)
#ifdef KBJ_EMU
  return 1.0f / sqrtf(x);
#else
  const float y = __builtin_amdgcn_rsqf(x);
  return fmaf(0.5f * y, fmaf(-x * y, y, 1.0f), y);
#endif
}
// sin and cos of one argument (|x| up to a few thousand; the callers pass joint half-angles below 2): quadrant by round-to-nearest of
// 2 x / pi, two-term Cody-Waite reduction, the single-precision minimax polynomials of the Cephes library on [-pi/4, pi/4]; ~1 ulp.
// Plain arithmetic, identical on the GPU and in the emulation; ~23 instructions against ~80 for sinf() + cosf().
KBJ_DEV void kbj_sincos(float x, float& s, float& c) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.57079637f, x);
  r = fmaf(-k, -4.37113883e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// -x / (lane P of the row of x): the multiplier column of a pivot. ONE hardware reciprocal read through DPP (v_rcp_f32: 1 ulp) and one
// multiply. The emulation with a reciprocal that is off by -1 / 0 / +1 ulp (KBJ_EMU_RCP_1ULP) gives the same error quantiles against the
// fp64 oracle as the exact division (tools/emu_parity.py, 49 k env-steps: qpos p99.9 1.71e-6 vs 1.83e-6, qacc 5.1e-5 vs 4.7e-5, the oracle's
// own fp32 1.6e-6 / 4.7e-5): an LDL^T pivot's reciprocal is one rounding among the eleven a row already takes. KBJ_SOLVER_NR adds a
// Newton step on the quotient (A/B builds; 3 more instructions per pivot, 33 per solve).
template <int P> KBJ_DEV WF wneg_div_bcast(const WF& x) {
  WF q;
#if defined(KBJ_EMU) || defined(KBJ_SOLVER_NR)
  const WF d = wbcast<P>(x);
  WLANES(l) {
    const float r = wrcp_scalar(WL(d, l)), q0 = -WL(x, l) * r;
#ifdef KBJ_SOLVER_NR
    const float e = fmaf(q0, WL(d, l), WL(x, l));
    WL(q, l) = fmaf(-e, r, q0);
#else
    WL(q, l) = q0;
#endif
  }
#else
  float r;
  asm("s_nop 1\n\tv_rcp_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(P));
  q = -x * r;
#endif
  return q;
}

}  // namespace kbj
