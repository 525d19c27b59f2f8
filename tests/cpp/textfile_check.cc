// textfile_check.cc -- MCout::write_text on the CPU under MPI (tests/test_mcout_cpu.py): every rank hands several blocks
// of text of different sizes (one of them empty, one large enough for several pwrite pieces to matter little but the
// offsets a lot) to write_text -- once through rank 0's stream (the funnel), once into a shared file where every rank
// writes its own share at an MPI_Exscan'd offset (MCout::text_file).  Rank 0 then prints "same" if the file holds
// exactly the bytes the funnel wrote, in dump order: block by block, rank by rank.
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "mcout.hh"

static std::string block_text(int rank, int block)
{
  std::string s;
  const int lines = block == 2 ? (rank == 1 ? 0 : 3) : (block == 4 ? 20000 + 777 * rank : 5 + 3 * rank + block);
  for (int i = 0; i < lines; ++i) {
    s += "r" + std::to_string(rank) + " b" + std::to_string(block) + " line " + std::to_string(i) + "  ";
    s.append((size_t)((i * 7 + rank) % 23), 'x');
    s += "\n";
  }
  return s;
}

int main(int argc, char **argv)
{
  MPI_Init(&argc, &argv);
  int rank = 0, size = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const char *path = argc > 1 ? argv[1] : "textfile_check.out";
  std::ostringstream funnel;
  {
    MCout a(2, &funnel, MPI_COMM_WORLD);
    for (int b = 0; b < 6; ++b) {
      const std::string t = block_text(rank, b);
      a.write_text(t.data(), t.size());
    }
  }
  {
    std::ostringstream unused;
    MCout b(2, &unused, MPI_COMM_WORLD);
    if (!b.text_file(path)) {
      if (rank == 0) std::cout << "cannot open " << path << "\n";
      MPI_Finalize();
      return 1;
    }
    for (int k = 0; k < 6; ++k) {
      const std::string t = block_text(rank, k);
      b.write_text(t.data(), t.size());
    }
    b.text_file(0);
    if (rank == 0 && !unused.str().empty()) std::cout << "text leaked to the stream\n";
  }
  MPI_Barrier(MPI_COMM_WORLD);
  if (rank == 0) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream got;
    got << f.rdbuf();
    std::string want;
    for (int b = 0; b < 6; ++b)
      for (int r = 0; r < size; ++r) want += block_text(r, b);
    std::cout << (funnel.str() == want ? "funnel in dump order" : "FUNNEL OUT OF ORDER") << "\n"
              << (got.str() == funnel.str() ? "same" : "DIFFERENT") << " " << want.size() << " bytes, " << size << " ranks\n";
  }
  MPI_Finalize();
  return 0;
}
