// Host emulation side of tools/wave_test: the same primitives and the same arrow solve as the kernel, from the KBJ_EMU bodies of kbj_wave.h.
#define KBJ_EMU 1
#include "../../kbot-joystick_amd/csrc/kbj_env_phys.h"
using namespace kbj;
static WF ld(const float* p) { WF w; for (int l = 0; l < 64; ++l) w.v[l] = p[l]; return w; }
static void st(float* p, const WF& w) { for (int l = 0; l < 64; ++l) p[l] = w.v[l]; }
extern "C" void wave_emu(const float* in, float* out) {
  const WF x = ld(in), y = ld(in + 64), z = ld(in + 128);
  int k = 0;
  st(out + 64 * k++, wbcast<3>(x));
  { WF a = z; wfmac_bcast<5>(a, x, y); st(out + 64 * k++, a); }
  st(out + 64 * k++, wmul_bcast<12>(x, y));
  { WF a = z; wset_rhs<7, 0xF>(a, x); st(out + 64 * k++, a); }
  { WF a = z; wset_rhs<9, 0x1>(a, x); st(out + 64 * k++, a); }
  st(out + 64 * k++, wrow_sum16(x));
  st(out + 64 * k++, wrows_sum1(x));
  { WF a = x, b = y; wrows_sum2(a, b); st(out + 64 * k++, a); st(out + 64 * k++, b); }
  { WF a; const float s = wsum(x); for (int l = 0; l < 64; ++l) a.v[l] = s; st(out + 64 * k++, a); }
  { WF a, b; float s1, s2; wsum2(x, y, s1, s2); for (int l = 0; l < 64; ++l) { a.v[l] = s1; b.v[l] = s2; } st(out + 64 * k++, a); st(out + 64 * k++, b); }
  st(out + 64 * k++, wsel<wmask_r(4)>(x, y));
  st(out + 64 * k++, wsel0<(wmask_r_below(5) | (wmask_r_below(11) & 0xFFFFull))>(x));
  st(out + 64 * k++, wneg_div_bcast<2>(x));
  { WF a; const unsigned long long m = wballot([&](int l) { return x.v[l] > 0.0f; }); for (int l = 0; l < 64; ++l) a.v[l] = (float)((m >> l) & 1); st(out + 64 * k++, a); }
}
// arrow solve: m[11], h[11], oh[5], dnow, g as 29 wave values; x out
extern "C" void solve_emu(const float* in, float* out) {
  WF m[11], h[11], oh[5];
  for (int j = 0; j < 11; ++j) { m[j] = ld(in + 64 * j); h[j] = ld(in + 64 * (11 + j)); }
  for (int j = 0; j < 5; ++j) oh[j] = ld(in + 64 * (22 + j));
  const WF dnow = ld(in + 64 * 27), g = ld(in + 64 * 28);
  st(out, arrow_solve_w(m, h, oh, dnow, g));
}
