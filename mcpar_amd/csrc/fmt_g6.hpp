// fmt_g6.hpp -- the characters `std::ostream << float` produces with the stream's defaults (precision 6, no
// floatfield flag), i.e. printf("%g", (double)f): the text format of the reference's sample dump
// (src/mcout.cc:41-45: every field followed by two blanks, a newline after the last column).
//
// That conversion is 65-87 % of the reference's wall time and, with the chain steps on the GPU, all of
// the time a driver spends that prints its samples.  It is done here exactly -- correctly rounded, ties to even, like
// glibc -- in integer arithmetic only, by one routine that compiles for the host (the facade's MCout::output) and for
// gfx950 (the engine's text sink): same bytes everywhere, no libm, no locale.
//
//   value = m 2^e  (m < 2^24).  With k = floor(log10 value) the six significant digits are N = round(value 10^(5-k)):
//     p = 5 - k >= 0 (value < 1e6, hence e < 0):  N = (m 10^p) >> -e, rounded on the bits shifted out -- m 10^p < 2^191;
//     p < 0:  N = round(m 2^e / 10^-p), quotient < 2^20: estimated in double, corrected and rounded on the exact remainder.
//   k is estimated from the bit length (off by at most one) and settled by N itself: N > 10^6 -> k + 1; N == 10^6 ->
//   digits 100000, k + 1; N < 10^5 -> k - 1.  %g: exponent X = k; X < -4 or X >= 6 -> d.ddddde+XX, else fixed with
//   5 - X decimals; trailing zeros (and a bare point) removed.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FMTG6_HD __host__ __device__ inline
#else
#define FMTG6_HD inline
#endif

namespace fmtg6 {

struct Big {  // 256-bit unsigned, little-endian 32-bit limbs
  uint32_t w[8];
};

FMTG6_HD void big_set(Big &a, uint32_t v)
{
  a.w[0] = v;
  for (int i = 1; i < 8; ++i) a.w[i] = 0u;
}

FMTG6_HD void big_mul_small(Big &a, uint32_t m)
{
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.w[i] * m;
    a.w[i] = (uint32_t)c;
    c >>= 32;
  }
}

FMTG6_HD void big_mul_pow10(Big &a, int q)  // a *= 10^q, q >= 0
{
  for (; q >= 9; q -= 9) big_mul_small(a, 1000000000u);
  uint32_t t = 1u;
  for (; q > 0; --q) t *= 10u;
  big_mul_small(a, t);
}

FMTG6_HD void big_shl(Big &a, int s)  // a <<= s, 0 <= s < 256 (bits shifted out are lost: callers keep them in range)
{
  const int ws = s >> 5, bs = s & 31;
  for (int i = 7; i >= 0; --i) {
    uint32_t v = 0u;
    if (i - ws >= 0) {
      v = a.w[i - ws] << bs;
      if (bs && i - ws - 1 >= 0) v |= a.w[i - ws - 1] >> (32 - bs);
    }
    a.w[i] = v;
  }
}

FMTG6_HD int big_cmp(const Big &a, const Big &b)
{
  for (int i = 7; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}

FMTG6_HD void big_sub(Big &a, const Big &b)  // a -= b (a >= b)
{
  uint64_t br = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a.w[i] - b.w[i] - br;
    a.w[i] = (uint32_t)d;
    br = (d >> 32) & 1u;
  }
}

FMTG6_HD uint32_t big_bit(const Big &a, int b) { return (a.w[b >> 5] >> (b & 31)) & 1u; }

// bits [s, s + 32) of a
FMTG6_HD uint32_t big_extract(const Big &a, int s)
{
  const int ws = s >> 5, bs = s & 31;
  uint32_t v = ws < 8 ? a.w[ws] >> bs : 0u;
  if (bs && ws + 1 < 8) v |= a.w[ws + 1] << (32 - bs);
  return v;
}

FMTG6_HD bool big_low_bits_nonzero(const Big &a, int nbits)  // any of the bits [0, nbits) set
{
  const int ws = nbits >> 5, bs = nbits & 31;
  for (int i = 0; i < ws && i < 8; ++i)
    if (a.w[i]) return true;
  return bs && ws < 8 && (a.w[ws] & ((1u << bs) - 1u));
}

// round(m 2^e 10^p) to an integer, ties to even -- exact.  The result is known to be small (callers choose p so).
FMTG6_HD uint32_t scaled_round(uint32_t m, int e, int p)
{
  if (p >= 0 && p <= 11 && e < 0 && e >= -63) {  // 1e-6 <= value < 1e6, i.e. nearly every sample: m 10^p < 2^61
    uint64_t num = m;
    for (int i = 0; i < p; ++i) num *= 10u;
    const int s = -e;
    uint64_t n = num >> s;
    const uint64_t half = (num >> (s - 1)) & 1u, sticky = num & ((1ull << (s - 1)) - 1ull);
    if (half && (sticky || (n & 1u))) ++n;
    return n > 0xffffffffull ? 0xffffffffu : (uint32_t)n;
  }
  if (p >= 0) {
    Big num;
    big_set(num, m);
    big_mul_pow10(num, p);
    if (e >= 0) {  // (an integer; happens only while k settles, with e == 0)
      big_shl(num, e);
      for (int i = 1; i < 8; ++i)
        if (num.w[i]) return 0xffffffffu;
      return num.w[0];
    }
    const int s = -e;  // 1 .. 149
    if (s > 224) return 0u;
    uint32_t n = big_extract(num, s);
    const uint32_t half = big_bit(num, s - 1);
    if (half && (big_low_bits_nonzero(num, s - 1) || (n & 1u))) ++n;
    return n;
  }
  const int q = -p;  // 1 .. 33
  Big num, den;
  big_set(num, m);
  big_set(den, 1u);
  big_mul_pow10(den, q);
  if (e >= 0) big_shl(num, e);
  else big_shl(den, -e);
  // estimate of the quotient (< 2^21): the top bits of num over the top bits of den, then settled exactly
  double dn = 0.0, dd = 0.0;
  for (int i = 7; i >= 0; --i) {
    dn = dn * 4294967296.0 + (double)num.w[i];
    dd = dd * 4294967296.0 + (double)den.w[i];
  }
  uint32_t n = (uint32_t)(dn / dd);
  Big prod = den;
  big_mul_small(prod, n);
  while (big_cmp(prod, num) > 0) {  // n too large
    --n;
    big_sub(prod, den);
  }
  Big rem = num;
  big_sub(rem, prod);
  while (big_cmp(rem, den) >= 0) {  // n too small
    ++n;
    big_sub(rem, den);
  }
  big_shl(rem, 1);
  const int c = big_cmp(rem, den);
  if (c > 0 || (c == 0 && (n & 1u))) ++n;
  return n;
}

// Appends to two 8-byte words (little endian: byte i of the text is byte i of lo, then of hi); at most 15 characters.
struct Text {
  uint64_t lo, hi;
  int len;
};

FMTG6_HD void put(Text &t, char c)
{
  if (t.len < 8) t.lo |= (uint64_t)(uint8_t)c << (8 * t.len);
  else t.hi |= (uint64_t)(uint8_t)c << (8 * (t.len - 8));
  ++t.len;
}

// the six significant digits of a finite non-zero float (100000 <= n <= 999999) and its decimal exponent k
FMTG6_HD uint32_t digits6(uint32_t ex, uint32_t mant, int *kout)
{
  const uint32_t m = ex ? (mant | 0x800000u) : mant;
  const int e = ex ? (int)ex - 150 : -149;
  int blen = 0;
  for (uint32_t v = m; v; v >>= 1) ++blen;
  const int l2 = e + blen - 1;                 // floor(log2 value)
  int k = (l2 * 78913) >> 18;                  // floor(l2 log10 2): floor(log10 value) or one less
  uint32_t n;
  for (;;) {
    n = scaled_round(m, e, 5 - k);
    if (n > 1000000u) { ++k; continue; }
    if (n == 1000000u) { n = 100000u; ++k; break; }
    if (n < 100000u) { --k; continue; }
    break;
  }
  *kout = k;
  return n;
}

// the text of the float whose bits are `bits`
FMTG6_HD Text format(uint32_t bits)
{
  Text t;
  t.lo = t.hi = 0;
  t.len = 0;
  const uint32_t ex = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
  if (bits >> 31) put(t, '-');
  if (ex == 0xffu) {  // glibc: "inf" / "nan" behind the sign
    if (mant) { put(t, 'n'); put(t, 'a'); put(t, 'n'); }
    else { put(t, 'i'); put(t, 'n'); put(t, 'f'); }
    return t;
  }
  if (ex == 0u && mant == 0u) {
    put(t, '0');
    return t;
  }
  int k;
  uint32_t n = digits6(ex, mant, &k);
  char d[6];
  for (int i = 5; i >= 0; --i) {
    d[i] = (char)('0' + n % 10u);
    n /= 10u;
  }
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0') --nd;
  if (k < -4 || k >= 6) {  // d.ddddde+XX
    put(t, d[0]);
    if (nd > 1) {
      put(t, '.');
      for (int i = 1; i < nd; ++i) put(t, d[i]);
    }
    put(t, 'e');
    int x = k;
    if (x < 0) { put(t, '-'); x = -x; }
    else put(t, '+');
    put(t, (char)('0' + x / 10));
    put(t, (char)('0' + x % 10));
  } else if (k >= 0) {  // k + 1 integer digits, the rest behind the point
    for (int i = 0; i <= k; ++i) put(t, i < nd ? d[i] : '0');
    if (nd > k + 1) {
      put(t, '.');
      for (int i = k + 1; i < nd; ++i) put(t, d[i]);
    }
  } else {  // 0.000ddd
    put(t, '0');
    put(t, '.');
    for (int i = 0; i < -k - 1; ++i) put(t, '0');
    for (int i = 0; i < nd; ++i) put(t, d[i]);
  }
  return t;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// host: append the text of f at dst (room for 16 bytes), return the new end.  The same digits (digits6) written
// straight into the buffer -- format() assembles two words for the GPU's lanes, which costs the host a third of its time.
inline char *append(char *dst, float f)
{
  uint32_t bits;
  memcpy(&bits, &f, 4);
  const uint32_t ex = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
  char *p = dst;
  if (bits >> 31) *p++ = '-';
  if (ex == 0xffu) {
    memcpy(p, mant ? "nan" : "inf", 3);
    return p + 3;
  }
  if (ex == 0u && mant == 0u) {
    *p++ = '0';
    return p;
  }
  int k;
  uint32_t n = digits6(ex, mant, &k);
  char d[6];
  const uint32_t hi = n / 1000u, lo = n - hi * 1000u;  // two three-digit halves
  d[0] = (char)('0' + hi / 100u); d[1] = (char)('0' + hi / 10u % 10u); d[2] = (char)('0' + hi % 10u);
  d[3] = (char)('0' + lo / 100u); d[4] = (char)('0' + lo / 10u % 10u); d[5] = (char)('0' + lo % 10u);
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0') --nd;
  if (k < -4 || k >= 6) {
    *p++ = d[0];
    if (nd > 1) {
      *p++ = '.';
      for (int i = 1; i < nd; ++i) *p++ = d[i];
    }
    *p++ = 'e';
    int x = k;
    if (x < 0) { *p++ = '-'; x = -x; }
    else *p++ = '+';
    *p++ = (char)('0' + x / 10);
    *p++ = (char)('0' + x % 10);
  } else if (k >= 0) {
    for (int i = 0; i <= k; ++i) *p++ = i < nd ? d[i] : '0';
    if (nd > k + 1) {
      *p++ = '.';
      for (int i = k + 1; i < nd; ++i) *p++ = d[i];
    }
  } else {
    *p++ = '0';
    *p++ = '.';
    for (int i = 0; i < -k - 1; ++i) *p++ = '0';
    for (int i = 0; i < nd; ++i) *p++ = d[i];
  }
  return p;
}

// the same through format(): what the GPU kernels place (tests compare the two)
inline char *append_words(char *dst, float f)
{
  uint32_t b;
  memcpy(&b, &f, 4);
  const Text t = format(b);
  memcpy(dst, &t.lo, 8);
  memcpy(dst + 8, &t.hi, 8);
  return dst + t.len;
}
#endif

}  // namespace fmtg6
