// mcout.hh -- the reference's sample sink (src/mcout.hh:1-54), same public interface.  MCPar::run
// fills it in bulk from the engine's HBM-resident sample store at the points where the reference
// dumps output; rows, row order and the text format are the reference's (src/mcout.cc:30-48,129-145).
#ifndef MCPAR_AMD_MCOUT_HH_
#define MCPAR_AMD_MCOUT_HH_

#include <assert.h>
#include <iostream>
#include <vector>

#include "mpi_compat.hh"

class MCout {
  std::vector<float> pvals;
  std::vector<float> maxlparams;
  float maxlval;
  const int mnparam;  // number of model parameters
  const int mncol;    // number of data columns = # of parameters + 1
  size_t next;
  int npset;     // number of parameter sets stored
  int maxsamps;  // maximum number of parameter sets that can be stored
  size_t nextout;  // offset (in elements) of the next parameter set to be output
  std::ostream *outstream;
  MPI_Comm mComm;
  int mrank;
  int msize;

public:
  MCout(int np, std::ostream *aoutstream, MPI_Comm acomm);
  void newsamps(int nsamp)
  {
    maxsamps += nsamp;
    pvals.resize(pvals.size() + (size_t)nsamp * mncol);
  }
  void add(const float *pv, float lval);
  // bulk form of add(): nrows rows of (np+1) floats, already in MCout layout
  void add_rows(const float *rows, size_t nrows);
  int size(void) const { return npset; }
  int maxsize(void) const { return maxsamps; }
  int ncol(void) { return mncol; }
  int vsize(void) const { return (int)pvals.size(); }
  const float *getpset(int i) const { return &pvals[(size_t)i * mncol]; }
  float getlval(int i) const { return pvals[(size_t)(i + 1) * mncol - 1]; }
  void output();
  float *collect(size_t *ntot);
  void rewind(void) { nextout = 0; }
  // Warning: maxlike is a COLLECTIVE call.  All processes in the group must call it at the same time.
  const std::vector<float> &maxlike(float *lmax);
};

#endif
