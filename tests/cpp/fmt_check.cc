// fmt_check -- mcpar_amd/csrc/fmt_g6.hpp against the C library: the text of a float as `ostream << float` / printf("%g")
// gives it (src/mcout.cc:41-45 prints every sample through it).
//   fmt_check quick        every 1009th bit pattern + neighbourhoods of the powers of ten and of 6-digit values and ties
//   fmt_check more         the same, denser (every 211th; ~80 M values)
//   fmt_check all          all 2^32 bit patterns (minutes; OpenMP)
//   fmt_check text         a few rows in MCout's layout through std::ostream, for the eye
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include "../../mcpar_amd/csrc/fmt_g6.hpp"

static bool same(uint32_t bits, long *bad)
{
  float f;
  memcpy(&f, &bits, 4);
  char want[64], got[32];
  snprintf(want, sizeof want, "%g", (double)f);
  char *end = fmtg6::append(got, f);
  *end = 0;
  char got2[32];
  *fmtg6::append_words(got2, f) = 0;  // (the form the GPU's lanes assemble)
  if (strcmp(want, got) != 0 || strcmp(want, got2) != 0) {
    if (++*bad <= 10) fprintf(stderr, "bits %08x: libc '%s' fmtg6 '%s'\n", bits, want, got);
    return false;
  }
  return true;
}

int main(int argc, char **argv)
{
  const std::string mode = argc > 1 ? argv[1] : "quick";
  long bad = 0, n = 0;
  if (mode == "text") {
    std::ostringstream os;
    const float row[] = {1e-05f, 123456.0f, 1e+10f, -0.0f, 0.1f, 1234567.0f, 3.14159274f, -1.5e-7f, INFINITY, -NAN};
    for (float v : row) os << v << "  ";
    printf("%s\n", os.str().c_str());
    char buf[512], *p = buf;
    for (float v : row) { p = fmtg6::append(p, v); *p++ = ' '; *p++ = ' '; }
    *p = 0;
    printf("%s\n", buf);
    return strcmp(os.str().c_str(), buf) != 0;
  }
  if (mode == "all") {
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad, n)
    for (long hi = 0; hi < 65536; ++hi)
      for (uint32_t lo = 0; lo < 65536u; ++lo) {
        long b = 0;
        same(((uint32_t)hi << 16) | lo, &b);
        bad += b;
        ++n;
      }
  } else {
    const int stride = mode == "more" ? 211 : 1009, dstep = mode == "more" ? 37 : 499;
    for (uint64_t b = 0; b < (1ull << 32); b += stride) { same((uint32_t)b, &bad); ++n; }
    // around every power of ten and every value whose seventh digit is a 5 followed by zeros (the ties)
    for (int k = -45; k <= 38; ++k)
      for (int d = 100000; d <= 1000000; d += dstep) {
        for (int half = 0; half < 2; ++half) {
          const double v = ((double)d + 0.5 * half) * std::pow(10.0, k - 5);
          float f = (float)v;
          uint32_t bits;
          memcpy(&bits, &f, 4);
          for (int o = -3; o <= 3; ++o) { same(bits + (uint32_t)o, &bad); same((bits + (uint32_t)o) | 0x80000000u, &bad); n += 2; }
        }
      }
    for (uint32_t b = 0; b < 4096; ++b) { same(b, &bad); same(0x7f7ff000u + b, &bad); same(0x7f800000u + b * 2048u, &bad); n += 3; }
  }
  printf("%ld values, %ld differences\n", n, bad);
  return bad != 0;
}
