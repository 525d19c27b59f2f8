// mcout_check.cc -- MCout (include/mcpar/mcout.hh) without a GPU: storage, row format, incremental
// output, collect(), rewind(), maxlike -- the behaviour of src/mcout.cc on one rank.
#include <iostream>
#include <sstream>

#include "mcpar/mcout.hh"

int main()
{
  std::ostringstream os;
  MCout o(2, &os, MPI_COMM_WORLD);
  o.newsamps(3);
  const float a[2] = {1.5f, -2.0f}, b[2] = {0.1f, 1e-5f}, c[2] = {123456.0f, 1e10f};
  o.add(a, -3.25f);
  o.add(b, 0.5f);
  std::cout << "size " << o.size() << " maxsize " << o.maxsize() << " ncol " << o.ncol() << " vsize " << o.vsize() << "\n";
  o.output();  // two rows
  o.output();  // nothing new
  const float rows[3] = {123456.0f, 1e10f, 0.25f};
  o.add_rows(rows, 1);
  o.output();  // one row
  std::cout << os.str() << "--\n";
  os.str("");
  o.rewind();
  size_t nt = 0;
  float *buf = o.collect(&nt);
  std::cout << "collect " << nt << " " << buf[2] << " " << buf[5] << " " << buf[8] << "\n";
  delete[] buf;
  buf = o.collect(&nt);
  std::cout << "collect again " << nt << " " << (buf == 0) << "\n";
  float lmax;
  const std::vector<float> &pm = o.maxlike(&lmax);
  std::cout << "maxlike " << lmax << " " << pm[0] << " " << pm[1] << "\n";
  std::cout << "getpset " << o.getpset(1)[0] << " " << o.getlval(1) << " " << o.getlval(2) << "\n";
  (void)c;
  return 0;
}
