// mcx_numerics.hpp -- "MCX arithmetic v2"  for gfx950 and for the engine's host
// control code.  fp32 only, round-to-nearest-even, explicit fma; this translation unit must be
// built with -ffp-contract=off and without fast-math so that the device result of every function
// is a pure function of its input bits.
//
// Replaces the Intel MKL VSL calls of the reference hot path:
//   vslNewStream(MT2203+rank, 8675309)         src/mcpar.cc:270-271  -> Philox4x32-10 counters
//   vsRngUniform                               src/mcpar.cc:63,146,163,401 -> u24()
//   viRngUniform                               src/mcpar.cc:337      -> mulhi(word, N)
//   vsRngGaussianMV(BOXMULLER2)                src/mcpar.cc:306,348  -> normal4()
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // hiprtc (MCX_VL_SOURCE: a user's likelihood compiled into the step kernels at run time) has no system headers
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
typedef int int32_t;
typedef long int64_t;
#endif

#define MCX_HD __host__ __device__ __forceinline__

namespace mcx {

// RNG streams: Philox key = (seed, stream); counter = (t, g, a, b).  DESIGN.md §3.2
enum : uint32_t { ST_LOCAL = 0, ST_ACCEPT = 1, ST_COIN = 2, ST_RSEL = 3, ST_RNORM = 4 };

constexpr float FPEPS = 1.0e-14f;  // src/mcpar.cc:15

struct u32x4 {
  uint32_t x, y, z, w;
};

MCX_HD float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
MCX_HD uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }

// a ^ b ^ c: gfx950 has a three-input bit operation (v_bitop3_b32, truth table 0x96) that the compiler does not pick for
// two xors by itself -- a Philox round is two multiplies and two of these instead of two multiplies and four xors
MCX_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

MCX_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1)
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

MCX_HD float u24(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }
MCX_HD float uopen(uint32_t w) { return __builtin_fmaf((float)w, 0x1p-32f, 0x1p-33f); }

// Range reduction x = 2^e (1 + m), 1 + m in [sqrt(1/2), sqrt(2)): as the oracle writes it (oracle/mcx_oracle.c: mantissa to
// [0.5, 1), compare with 0.70710678f = 0x3f3504f3, double it and lower e if below), and as it is computed here for
// 0 <= x < inf -- the integer form of the same thing, no compare and no select (v_cndmask and v_cmp cost 4.4 and 3 cycles,
// an integer add 2.6: tools/ubench.hip).  With b = E << 23 | M: ix = b - 0x3f3504f3 has floor(ix / 2^23) = E - 127 when
// M < 0x3504f3 and E - 126 otherwise, and its low 23 bits + 0x3f3504f3 are 0x3f800000 | M resp. 0x3f000000 | M: the bits of
// 2m resp. m of the text-book form.  Same e, same mantissa, and the subtraction of 1 is exact in both: same bits.
MCX_HD void log_reduce(float x, int &e, float &m)
{
  const uint32_t ix = as_u32(x) - 0x3f3504f3u;
  e = (int)ix >> 23;  // (arithmetic shift)
  m = as_f32((ix & 0x007fffffu) + 0x3f3504f3u) - 1.0f;
}

MCX_HD float logf_v1(float x)
{
  int e;
  float m;
  log_reduce(x, e, m);
  const float fe = (float)e;
  const float z = m * m;
  float p = 7.0376836292e-2f;
  p = __builtin_fmaf(p, m, -1.1514610310e-1f);
  p = __builtin_fmaf(p, m, 1.1676998740e-1f);
  p = __builtin_fmaf(p, m, -1.2420140846e-1f);
  p = __builtin_fmaf(p, m, 1.4249322787e-1f);
  p = __builtin_fmaf(p, m, -1.6668057665e-1f);
  p = __builtin_fmaf(p, m, 2.0000714765e-1f);
  p = __builtin_fmaf(p, m, -2.4999993993e-1f);
  p = __builtin_fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  y = __builtin_fmaf(-2.12194440e-4f, fe, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = m + y;
  r = __builtin_fmaf(0.693359375f, fe, r);
  return r;
}

// exp = 2^n e^r, n = floor(x log2(e) + 1/2).  n > 127 -> +inf, n < -125 -> 0: results are normal or zero,
// never denormal, so the scaling is an exact integer add to the exponent field.  NaN in, NaN out (the
// acceptance test then fails: src/mcpar.cc:66-69).
MCX_HD float expf_v2(float x)
{
  const float fn = __builtin_floorf(__builtin_fmaf(x, 1.44269504f, 0.5f));
  float r = __builtin_fmaf(fn, -0.693359375f, x);
  r = __builtin_fmaf(fn, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float z = r * r;
  float y = __builtin_fmaf(p, z, r);
  y = y + 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  const int n = (int)fn;  // v_cvt_i32_f32: saturates, NaN -> 0 (a NaN y stays NaN)
#else
  if (!(x == x)) return y;
  const int n = (int)(fn > 200.0f ? 200.0f : (fn < -200.0f ? -200.0f : fn));
#endif
  float s = as_f32(as_u32(y) + ((uint32_t)n << 23));
  s = fn < -125.0f ? 0.0f : s;
  s = fn > 127.0f ? __builtin_inff() : s;
  return s;
}

// log of the acceptance draw u24(w) in [0, 1): -inf at 0.  Local steps (cfac = 1) test u < exp(ly' - ly)
// (src/mcpar.cc:66-69,166-169) in the log domain, log u < ly' - ly: the transcendental then depends on the
// random draw alone and comes off the chain-state critical path.
MCX_HD float accept_lu(uint32_t w)
{
  const float l = logf_v1(u24(w));
  return (w >> 8) == 0u ? -__builtin_inff() : l;
}

MCX_HD void sincos2pi_v1(uint32_t w, float &s, float &c)
{
  const uint32_t k = ((w + 0x20000000u) >> 30) & 3u;
  const int32_t rem = (int32_t)(w - (k << 30));
  const float phi = (float)rem * 1.4629180792671596e-9f;
  const float z = phi * phi;
  float ps = -1.9515295891e-4f;
  ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float sp = __builtin_fmaf(phi * z, ps, phi);
  float pc = 2.443315711809948e-5f;
  pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float cp = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  const bool swap = (k & 1u) != 0u;
  const float sv = swap ? cp : sp;
  const float cv = swap ? sp : cp;
  // k: 0 (s,c)  1 (c,-s)  2 (-s,-c)  3 (-c,s)
  s = (k >= 2u) ? -sv : sv;
  c = (k == 1u || k == 2u) ? -cv : cv;
}

// Box-Muller on one Philox block: (w.x,w.y) -> z0,z1 ; (w.z,w.w) -> z2,z3
MCX_HD void normal4_from_words(const u32x4 &w, float z[4])
{
  float s, c;
  float r = __builtin_sqrtf(-2.0f * logf_v1(uopen(w.x)));
  sincos2pi_v1(w.y, s, c);
  z[0] = r * c;
  z[1] = r * s;
  r = __builtin_sqrtf(-2.0f * logf_v1(uopen(w.z)));
  sincos2pi_v1(w.w, s, c);
  z[2] = r * c;
  z[3] = r * s;
}

// ---- packed (2-wide) forms for gfx950's v_pk_*_f32: the same operations in the same order on two
// independent values, hence the same bits as the scalar forms above -------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

__device__ __forceinline__ f32x2 logf_v1x2(f32x2 x)
{
  const uint32_t ix = as_u32(x.x) - 0x3f3504f3u, iy = as_u32(x.y) - 0x3f3504f3u;  // (log_reduce, twice)
  const int ex = (int)ix >> 23, ey = (int)iy >> 23;
  const f32x2 t = {as_f32((ix & 0x007fffffu) + 0x3f3504f3u), as_f32((iy & 0x007fffffu) + 0x3f3504f3u)};
  const f32x2 m = t - splat2(1.0f);
  const f32x2 fe = {(float)ex, (float)ey};
  const f32x2 z = m * m;
  f32x2 p = splat2(7.0376836292e-2f);
  p = fma2(p, m, splat2(-1.1514610310e-1f));
  p = fma2(p, m, splat2(1.1676998740e-1f));
  p = fma2(p, m, splat2(-1.2420140846e-1f));
  p = fma2(p, m, splat2(1.4249322787e-1f));
  p = fma2(p, m, splat2(-1.6668057665e-1f));
  p = fma2(p, m, splat2(2.0000714765e-1f));
  p = fma2(p, m, splat2(-2.4999993993e-1f));
  p = fma2(p, m, splat2(3.3333331174e-1f));
  f32x2 y = (p * m) * z;
  y = fma2(splat2(-2.12194440e-4f), fe, y);
  y = fma2(splat2(-0.5f), z, y);
  f32x2 r = m + y;
  r = fma2(splat2(0.693359375f), fe, r);
  return r;
}

// a ^ (b & c) in one instruction (v_bitop3_b32, truth table 0x78): flips the sign of a where bit 31 of b is set, for c = 1 << 31
__device__ __forceinline__ float flip_sign_by(float a, uint32_t b)
{
  return as_f32((uint32_t)__builtin_amdgcn_bitop3_b32(as_u32(a), b, 0x80000000u, 0x78));
}

// sincos2pi_v1 (further up, = the oracle's) on two angles.  The quadrant k = (w + 2^29) >> 30 is never formed: with
// t = w + 2^29, the reduced angle is w - (t & 0xc0000000), "sine and cosine change places" is bit 30 of t, "the sine is
// negated" (k >= 2) is bit 31 of t, "the cosine is negated" (k = 1 or 2) is bit 31 ^ bit 30 = bit 31 of t ^ (t << 1), and a
// negation is an xor of the sign bit: 9 cheap integer operations, one compare and two selects per angle where the text-book
// form has 8, four compares and four selects (a select costs 4.4 cycles per wavefront, an integer operation 2.6:
// tools/ubench.hip).  Same values, same bits.
__device__ __forceinline__ void sincos2pi_v1x2(uint32_t wa, uint32_t wb, f32x2 &s, f32x2 &c)
{
  const uint32_t ta = wa + 0x20000000u, tb = wb + 0x20000000u;
  const int32_t ra = (int32_t)(wa - (ta & 0xc0000000u)), rb = (int32_t)(wb - (tb & 0xc0000000u));
  const f32x2 phi = f32x2{(float)ra, (float)rb} * splat2(1.4629180792671596e-9f);
  const f32x2 z = phi * phi;
  f32x2 ps = splat2(-1.9515295891e-4f);
  ps = fma2(ps, z, splat2(8.3321608736e-3f));
  ps = fma2(ps, z, splat2(-1.6666654611e-1f));
  const f32x2 sp = fma2(phi * z, ps, phi);
  f32x2 pc = splat2(2.443315711809948e-5f);
  pc = fma2(pc, z, splat2(-1.388731625493765e-3f));
  pc = fma2(pc, z, splat2(4.166664568298827e-2f));
  const f32x2 cp = fma2(z * z, pc, fma2(splat2(-0.5f), z, splat2(1.0f)));
  const uint32_t ua = ta << 1, ub = tb << 1;
  const bool swa = (int32_t)ua < 0, swb = (int32_t)ub < 0;  // bit 30 of t
  const f32x2 sv = {swa ? cp.x : sp.x, swb ? cp.y : sp.y};
  const f32x2 cv = {swa ? sp.x : cp.x, swb ? sp.y : cp.y};
  s = f32x2{flip_sign_by(sv.x, ta), flip_sign_by(sv.y, tb)};
  c = f32x2{flip_sign_by(cv.x, ta ^ ua), flip_sign_by(cv.y, tb ^ ub)};
}

// expf_v2 on two values: same operations in the same order
__device__ __forceinline__ f32x2 expf_v2x2(f32x2 x)
{
  const f32x2 fn = {__builtin_floorf(__builtin_fmaf(x.x, 1.44269504f, 0.5f)),
                    __builtin_floorf(__builtin_fmaf(x.y, 1.44269504f, 0.5f))};
  f32x2 r = fma2(fn, splat2(-0.693359375f), x);
  r = fma2(fn, splat2(2.12194440e-4f), r);
  f32x2 p = splat2(1.9875691500e-4f);
  p = fma2(p, r, splat2(1.3981999507e-3f));
  p = fma2(p, r, splat2(8.3334519073e-3f));
  p = fma2(p, r, splat2(4.1665795894e-2f));
  p = fma2(p, r, splat2(1.6666665459e-1f));
  p = fma2(p, r, splat2(5.0000001201e-1f));
  const f32x2 z = r * r;
  f32x2 y = fma2(p, z, r);
  y = y + splat2(1.0f);
  const int na = (int)fn.x, nb = (int)fn.y;
  f32x2 o;
  o.x = as_f32(as_u32(y.x) + ((uint32_t)na << 23));
  o.y = as_f32(as_u32(y.y) + ((uint32_t)nb << 23));
  o.x = fn.x < -125.0f ? 0.0f : o.x;
  o.y = fn.y < -125.0f ? 0.0f : o.y;
  o.x = fn.x > 127.0f ? __builtin_inff() : o.x;
  o.y = fn.y > 127.0f ? __builtin_inff() : o.y;
  return o;
}

// expf_v2x2 for arguments that are never positive (the Murray sweeps' -arg/2: arg is a sum of squares times 1/sigma^2): n <= 0,
// so the overflow select can never fire -- left out, same bits (a NaN argument compares false there as well)
__device__ __forceinline__ f32x2 expf_v2x2_nonpos(f32x2 x)
{
  const f32x2 fn = {__builtin_floorf(__builtin_fmaf(x.x, 1.44269504f, 0.5f)),
                    __builtin_floorf(__builtin_fmaf(x.y, 1.44269504f, 0.5f))};
  f32x2 r = fma2(fn, splat2(-0.693359375f), x);
  r = fma2(fn, splat2(2.12194440e-4f), r);
  f32x2 p = splat2(1.9875691500e-4f);
  p = fma2(p, r, splat2(1.3981999507e-3f));
  p = fma2(p, r, splat2(8.3334519073e-3f));
  p = fma2(p, r, splat2(4.1665795894e-2f));
  p = fma2(p, r, splat2(1.6666665459e-1f));
  p = fma2(p, r, splat2(5.0000001201e-1f));
  const f32x2 z = r * r;
  f32x2 y = fma2(p, z, r);
  y = y + splat2(1.0f);
  const int na = (int)fn.x, nb = (int)fn.y;
  f32x2 o;
  o.x = as_f32(as_u32(y.x) + ((uint32_t)na << 23));
  o.y = as_f32(as_u32(y.y) + ((uint32_t)nb << 23));
  o.x = fn.x < -125.0f ? 0.0f : o.x;
  o.y = fn.y < -125.0f ? 0.0f : o.y;
  return o;
}

// accept_lu of two draws at once (same bits)
__device__ __forceinline__ f32x2 accept_lu_x2(uint32_t wa, uint32_t wb)
{
  const f32x2 l = logf_v1x2(f32x2{u24(wa), u24(wb)});
  return f32x2{(wa >> 8) == 0u ? -__builtin_inff() : l.x, (wb >> 8) == 0u ? -__builtin_inff() : l.y};
}

// Correctly rounded sqrt for x = +-0 or x >= 2^-96: the raw v_sqrt_f32 (<= 1 ulp) plus
// the one-ulp-down / one-ulp-up residual test -- the sequence hipcc emits for IEEE sqrtf, minus its
// denormal pre-scaling and its zero/inf class check, neither of which the Box-Muller argument
// -2 ln(u), u in (0,1], can need (it is 0 or >= 1.19e-7).  Bit-identical to sqrtf on that domain (exhaustively checked:
// tests/test_gpu_numerics.py::test_sqrt_rn_exhaustive).
__device__ __forceinline__ float sqrt_rn_pos(float x)
{
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = as_f32(as_u32(s) - 1u), su = as_f32(as_u32(s) + 1u);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float r = (0.0f >= rd) ? sd : s;
  r = (0.0f < ru) ? su : r;
  return r;
}

// Box-Muller on one Philox block, both pairs at once.  ze = (z0, z2), zo = (z1, z3)
__device__ __forceinline__ void normal4_packed(const u32x4 &w, f32x2 &ze, f32x2 &zo)
{
  const f32x2 u = fma2(f32x2{(float)w.x, (float)w.z}, splat2(0x1p-32f), splat2(0x1p-33f));
  const f32x2 a = splat2(-2.0f) * logf_v1x2(u);
  const f32x2 r = {sqrt_rn_pos(a.x), sqrt_rn_pos(a.y)};
  f32x2 s, c;
  sincos2pi_v1x2(w.y, w.w, s, c);
  ze = r * c;
  zo = r * s;
}

MCX_HD uint32_t pick_word(const u32x4 &w, uint32_t i)
{
  return i == 0u ? w.x : (i == 1u ? w.y : (i == 2u ? w.z : w.w));
}

}  // namespace mcx
