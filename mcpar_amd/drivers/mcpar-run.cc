// mcpar-run -- the BASELINE configurations as a first-class driver (the reference reaches them
// only through its library API: SURVEY fact 3).
//   mcpar-run [--func rosen1|rosen2|rosen2fixed|gauss|dgauss|mix | --func-source FILE.hip [--par a,b,...]] [--np D]
//             [--nc CHAINS] [--nsamp N] [--nburn B] [--pl P] [--sync S] [--ncomp K] [--quiet] [--iter] [--binary]
//             [--stream-text] [--out FILE]
// --func-source: the user's own likelihood as HIP source of device functions (SourceVLFunc, MCX_VL_SOURCE: compiled into
// the engine's fused step kernels at run time; mcpar_amd/examples/ has three), --par its parameter block.
// Output: the reference's row format (src/mcout.cc:41-45); --iter prepends the iteration index
// that src/anly/mcpar-analysis.R:80-120 reconstructs; --quiet prints only the summary (stderr); --out FILE: the sample
// text goes to FILE, every rank writing its own share at its place (MCout::text_file) instead of through rank 0.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mcpar/mcout.hh"
#include "mcpar/mcpar.hh"
#include "mcpar/rosenbrock.hh"

#include "../csrc/fmt_g6.hpp"

int main(int argc, char *argv[])
{
  std::string func = "rosen1";
  int np = 16, nc = 4096, nsamp = 100, nburn = 500, sync = 10, ncomp = 8;
  float pl = 1.0f;
  bool quiet = false, iter = false, binary = false, stream_text = false;
  std::string out_file, func_source;
  std::vector<float> user_par;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : "0"; };
    if (a == "--func") func = val();
    else if (a == "--func-source") func_source = val();
    else if (a == "--par") {
      std::stringstream ss(val());
      for (std::string tok; std::getline(ss, tok, ',');) user_par.push_back((float)atof(tok.c_str()));
    }
    else if (a == "--np") np = atoi(val());
    else if (a == "--nc") nc = atoi(val());
    else if (a == "--nsamp") nsamp = atoi(val());
    else if (a == "--nburn") nburn = atoi(val());
    else if (a == "--pl") pl = (float)atof(val());
    else if (a == "--sync") sync = atoi(val());
    else if (a == "--ncomp") ncomp = atoi(val());
    else if (a == "--quiet") quiet = true;
    else if (a == "--iter") iter = true;
    else if (a == "--binary") binary = true;  // rows as raw float32 (np+1 per row) instead of text
    else if (a == "--stream-text") stream_text = true;  // the same text, formatted on the GPU, nothing kept on the host
    else if (a == "--out") out_file = val();
    else { std::cerr << "unknown option " << a << "\n"; return 2; }
  }
  MPI_Init(&argc, &argv);
  int size, rank;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);

  VLFunc *L = 0;
  std::vector<float> means, w;
  try {
    if (!func_source.empty()) {
      FILE *f = fopen(func_source.c_str(), "rb");
      if (!f) { std::cerr << "cannot read " << func_source << "\n"; return 2; }
      std::string text;
      char buf[4096];
      for (size_t k; (k = fread(buf, 1, sizeof buf, f)) > 0;) text.append(buf, k);
      fclose(f);
      func = "source:" + func_source;
      L = new SourceVLFunc(np, text.c_str(), user_par.empty() ? 0 : user_par.data(), (int)user_par.size());
    } else
    if (func == "rosen1") L = new Rosenbrock1(np);
    else if (func == "rosen2") L = new Rosenbrock2(np);
    else if (func == "rosen2fixed") L = new Rosenbrock2Fixed(np);  // flagged variant, not reference behaviour
    else if (func == "gauss") L = new Gaussian(np);
    else if (func == "dgauss") { np = 2; L = new DualGaussian(5.0f); }
    else if (func == "mix") {  // means 5k/(K-1) * 1, weights (5,1,...,1)
      means.resize((size_t)ncomp * np);
      w.assign(ncomp, 1.0f);
      w[0] = 5.0f;
      for (int k = 0; k < ncomp; ++k)
        for (int i = 0; i < np; ++i) means[(size_t)k * np + i] = 5.0f * k / (ncomp > 1 ? ncomp - 1 : 1);
      L = new GaussianMixture(np, ncomp, means.data(), w.data());
    } else { std::cerr << "unknown --func " << func << "\n"; return 2; }
  } catch (const char *msg) {
    std::cerr << msg << "\n";
    return 2;
  }

  std::ostringstream sink;
  MCout rslts(np, (quiet || iter) ? static_cast<std::ostream *>(&sink) : &std::cout, MPI_COMM_WORLD);
  rslts.binary(binary);
  rslts.text_only(stream_text && !binary && !iter);
  if (!out_file.empty() && !rslts.text_file(out_file.c_str())) {
    std::cerr << "cannot open " << out_file << "\n";
    MPI_Finalize();
    return 2;
  }
  std::vector<float> pinit((size_t)nc * np);
  for (int j = 0; j < nc; ++j)
    for (int i = 0; i < np; ++i)
      pinit[(size_t)j * np + i] = (float)(0.5 * std::sin(0.37 * ((double)(rank * nc + j) * np + i)));
  try {
    MCPar mcpar(np, nc, size, rank, pl, 0.2f, 0.5f, 0.2f, 1.5f, sync);
    auto t0 = std::chrono::steady_clock::now();
    mcpar.run(nsamp, nburn, pinit.data(), *L, rslts);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (iter && !quiet) {  // (the numbers through fmtg6 like MCout::output: the same characters as `cout << float`)
      std::vector<char> line((size_t)(np + 1) * 18 + 32);
      for (int r = 0; r < rslts.size(); ++r) {
        char *q = line.data() + snprintf(line.data(), 16, "%d  ", r / nc);
        const float *p = rslts.getpset(r);
        for (int j = 0; j < np + 1; ++j) {
          q = fmtg6::append(q, p[j]);
          *q++ = ' ';
          *q++ = ' ';
        }
        *q++ = '\n';
        std::cout.write(line.data(), q - line.data());
      }
    }
    if (rank == 0)
      std::cerr << "chains " << nc << " x np " << np << "  burn " << nburn << " + samples " << nsamp
                << ": accept rate (main) " << (double)mcpar.naccept_main() / ((double)nc * nsamp)
                << ", remote passes " << mcpar.remote_passes() << ", " << (double)nc * (nburn + nsamp) / dt
                << " chain-steps/s incl. output, exchange: " << mcpar.exchange_backend() << "\n";
  } catch (const char *msg) {
    std::cerr << msg << "\n";
    return 2;
  }
  rslts.text_file(0);
  float lmax;
  const std::vector<float> &pmax = rslts.maxlike(&lmax);
  if (rank == 0) {
    std::cerr << "max likelihood value: " << lmax << "\n";
    for (size_t i = 0; i < pmax.size() && i < 8; ++i) std::cerr << pmax[i] << "  ";
    std::cerr << "\n";
  }
  delete L;
  MPI_Finalize();
  return 0;
}
