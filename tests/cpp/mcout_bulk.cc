// mcout_bulk -- MCout::output on many rows (the threaded fmtg6 path of the facade) against the stream itself, which is
// what the reference prints through (src/mcout.cc:41-45); and a stream that is NOT in its default state, which must
// be honoured (the facade falls back to the stream).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <vector>

#include "mcout.hh"

static float value(uint64_t &s)
{
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  const uint32_t r = (uint32_t)(s >> 32);
  switch (r & 7u) {
  case 0: { float f; uint32_t b = (uint32_t)(s >> 16); memcpy(&f, &b, 4); return f; }  // any bit pattern (nan, inf, denormals)
  case 1: return (float)((int)(r >> 8) % 2000001 - 1000000);
  case 2: return (float)std::ldexp((double)(r >> 8), -40 + (int)(r & 63u));
  default: return (float)(((double)(r >> 8) / 16777216.0 - 0.5) * 20.0);
  }
}

int main()
{
  const int np = 4, nrows = 300000;  // 1.5 M numbers: several pieces, every thread
  MPI_Comm comm = MPI_COMM_WORLD;
  std::ostringstream got, want;
  MCout out(np, &got, comm);
  out.newsamps(nrows);
  uint64_t s = 12345;
  std::vector<float> row(np + 1);
  for (int r = 0; r < nrows; ++r) {
    for (int c = 0; c <= np; ++c) row[c] = value(s);
    out.add(row.data(), row[np]);
    for (int c = 0; c <= np; ++c) want << row[c] << "  ";
    want << "\n";
  }
  out.output();
  if (got.str() != want.str()) { std::cout << "bulk text differs\n"; return 1; }
  // a stream with its own precision: the reference would print through it as it is
  std::ostringstream got2, want2;
  got2 << std::setprecision(9);
  want2 << std::setprecision(9);
  MCout out2(np, &got2, comm);
  out2.newsamps(3);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c <= np; ++c) row[c] = value(s);
    out2.add(row.data(), row[np]);
    for (int c = 0; c <= np; ++c) want2 << row[c] << "  ";
    want2 << "\n";
  }
  out2.output();
  if (got2.str() != want2.str()) { std::cout << "non-default stream differs\n"; return 1; }
  std::cout << "ok " << got.str().size() << "\n";
  return 0;
}
