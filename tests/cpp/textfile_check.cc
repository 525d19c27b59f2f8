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
  // output() under text_file(): the ROW path (MCPar::run falls back to it when a block's GPU text is not to be had; any
  // driver may call it) must land in the file too, after the text already there, and nothing may reach the stream --
  // first as text, interleaved with write_text blocks, then as raw rows under binary()
  std::string rows_want, bin_want;
  {
    const std::string p2 = std::string(path) + ".rows", p3 = std::string(path) + ".bin";
    std::ostringstream unused, viastream;
    MCout c(2, &unused, MPI_COMM_WORLD), ref(2, &viastream, MPI_COMM_WORLD), cb(2, &unused, MPI_COMM_WORLD);
    const bool ok2 = c.text_file(p2.c_str()), ok3 = cb.text_file(p3.c_str());
    if (!ok2 || !ok3) {
      if (rank == 0) std::cout << "cannot open " << p2 << "\n";
      MPI_Finalize();
      return 1;
    }
    c.newsamps(40); ref.newsamps(40); cb.newsamps(40);
    cb.binary(true);
    for (int blk = 0; blk < 3; ++blk) {
      for (int i = 0; i < 7 + blk; ++i) {
        const float pv[2] = {rank + 0.125f * i, -1.5e-7f * (blk + 1) * (i + 1)};
        c.add(pv, 100.0f * rank + i);
        ref.add(pv, 100.0f * rank + i);
        cb.add(pv, 100.0f * rank + i);
      }
      const std::string t = block_text(rank, blk);
      c.write_text(t.data(), t.size());  // (a block whose text came from the GPU ...)
      c.output();                        // (... then one that took the row path)
      ref.write_text(t.data(), t.size());
      ref.output();
      cb.output();
    }
    c.text_file(0);
    cb.text_file(0);
    MPI_Barrier(MPI_COMM_WORLD);  // (closing is local: every rank's share is in the file only after all have got here)
    if (rank == 0 && !unused.str().empty()) std::cout << "rows leaked to the stream\n";
    if (rank == 0) {
      std::ifstream f2(p2.c_str(), std::ios::binary), f3(p3.c_str(), std::ios::binary);
      std::stringstream g2, g3;
      g2 << f2.rdbuf();
      g3 << f3.rdbuf();
      // binary: per dump, rank-major, (np + 1) floats per row -- what a gathered binary dump holds
      std::string want3;
      for (int blk = 0; blk < 3; ++blk)
        for (int r = 0; r < size; ++r)
          for (int i = 0; i < 7 + blk; ++i) {
            const float row[3] = {r + 0.125f * i, -1.5e-7f * (blk + 1) * (i + 1), 100.0f * r + i};
            want3.append(reinterpret_cast<const char *>(row), sizeof row);
          }
      std::cout << (g2.str() == viastream.str() && !g2.str().empty() ? "rows same" : "ROWS DIFFERENT") << " " << g2.str().size() << " bytes\n"
                << (g3.str() == want3 ? "binary same" : "BINARY DIFFERENT") << " " << g3.str().size() << " bytes\n";
    }
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
