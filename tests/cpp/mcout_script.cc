// mcout_script.cc -- drives an MCout through a script and prints a transcript of everything observable.
//
// ONE source, two builds (it only uses MCout's public interface, src/mcout.hh:32-50):
//   * oracle/Makefile compiles it against the REFERENCE's own mcout.hh / mcout.cc, in place, with the real MPI
//     headers (-> oracle/_ref/ref_mcout_script); oracle/gen_golden.py runs that on 1 rank and under
//     `mpiexec -n 2` and commits scripts + transcripts as tests/golden/mcout_reference.json;
//   * tests/test_mcout_cpu.py compiles it against include/mcpar/mcout.hh (the facade) and requires the same
//     transcripts, byte for byte.
// Build with -I<directory that holds mcout.hh>.
//
// usage: mcout_script <np> <script.rank0> [<script.rank1> ...]     (one script per rank)
// script lines:  new N | add v0 .. v{np-1} l | output | collect | rewind | maxlike | stat | row I
// floats are written as 8 hex digits of their bit pattern; `add` reads hex bit patterns too.
// transcript: one line per observation, prefixed "r<rank> "; rank 0 prints all of them in rank order.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mcout.hh"

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv)
{
  MPI_Init(&argc, &argv);
  int rank = 0, size = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (argc < 3 || argc < 2 + size) {
    std::fprintf(stderr, "usage: mcout_script <np> <script per rank>...\n");
    MPI_Finalize();
    return 2;
  }
  const int np = std::atoi(argv[1]);
  std::ifstream in(argv[2 + rank]);
  std::ostringstream text;  // what MCout::output writes (rank 0 only, like the reference)
  std::ostringstream tr;    // the transcript
  char hex[16];
  auto H = [&](float f) { std::snprintf(hex, sizeof hex, "%08x", (unsigned)bits(f)); return std::string(hex); };
  {
    MCout o(np, &text, MPI_COMM_WORLD);
    std::string line;
    while (std::getline(in, line)) {
      std::istringstream ls(line);
      std::string cmd;
      if (!(ls >> cmd) || cmd[0] == '#') continue;
      if (cmd == "new") {
        int n = 0;
        ls >> n;
        o.newsamps(n);
      } else if (cmd == "add") {
        std::vector<float> v;
        std::string w;
        while (ls >> w) v.push_back(from_bits((uint32_t)std::stoul(w, nullptr, 16)));
        o.add(v.data(), v[(size_t)np]);
      } else if (cmd == "output") {
        text.str("");
        o.output();
        const std::string s = text.str();
        tr << "r" << rank << " output " << s.size() << "\n";
        std::istringstream rows(s);
        std::string r;
        while (std::getline(rows, r)) tr << "r" << rank << " |" << r << "|\n";
      } else if (cmd == "collect") {
        size_t nt = 12345;
        float *buf = o.collect(&nt);
        tr << "r" << rank << " collect " << (buf ? "buf" : "null");
        if (buf || nt != 12345) tr << " " << nt;  // (ranks > 0 with new rows leave *ntot untouched in the reference)
        if (buf)
          for (size_t i = 0; i < nt; ++i) tr << " " << H(buf[i]);
        tr << "\n";
        delete[] buf;
      } else if (cmd == "rewind") {
        o.rewind();
      } else if (cmd == "maxlike") {
        float lmax = 0.0f;
        const std::vector<float> &p = o.maxlike(&lmax);
        tr << "r" << rank << " maxlike " << H(lmax);
        for (size_t i = 0; i < p.size(); ++i) tr << " " << H(p[i]);
        tr << "\n";
      } else if (cmd == "stat") {
        tr << "r" << rank << " stat " << o.size() << " " << o.maxsize() << " " << o.ncol() << " " << o.vsize() << "\n";
      } else if (cmd == "row") {
        int i = 0;
        ls >> i;
        const float *p = o.getpset(i);
        tr << "r" << rank << " row " << i;
        for (int k = 0; k < np; ++k) tr << " " << H(p[k]);
        tr << " l " << H(o.getlval(i)) << "\n";
      }
    }
  }
  // rank 0 prints every rank's transcript, in rank order (stdout of several ranks would interleave)
  const std::string mine = tr.str();
#ifdef MPI_VERSION
  if (rank == 0) {
    std::fputs(mine.c_str(), stdout);
    for (int r = 1; r < size; ++r) {
      int len = 0;
      MPI_Recv(&len, 1, MPI_INT, r, 7, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
      std::string other((size_t)len, ' ');
      if (len > 0) MPI_Recv(&other[0], len, MPI_CHAR, r, 8, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
      std::fputs(other.c_str(), stdout);
    }
  } else {
    int len = (int)mine.size();
    MPI_Send(&len, 1, MPI_INT, 0, 7, MPI_COMM_WORLD);
    if (len > 0) MPI_Send(const_cast<char *>(mine.data()), len, MPI_CHAR, 0, 8, MPI_COMM_WORLD);
  }
#else
  std::fputs(mine.c_str(), stdout);
#endif
  std::fflush(stdout);
  MPI_Finalize();
  return 0;
}
