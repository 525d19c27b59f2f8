// mcutil.hh -- utility class of the reference (src/mcutil.hh:10-31): quasi-random initial guesses.
// The reference draws them from MKL's Sobol generator (VSL_BRNG_SOBOL, src/mcutil.cc:16) with a
// skip-ahead of rank*npset*nparam values; here the same role is played by a self-contained Sobol
// sequence (Gray-code construction, Joe & Kuo direction numbers, first 21 dimensions).  Host-only:
// the guesses are the pinit argument of MCPar::run.
#ifndef MCPAR_AMD_MCUTIL_HH_
#define MCPAR_AMD_MCUTIL_HH_

#ifndef restrict
#define restrict __restrict__
#endif

class mcutil {
public:
  enum { MAXDIM = 21 };
  mcutil() {}
  ~mcutil() {}

  /*!
   * \brief quasi-random initial guess
   * \param[in] rank   MPI rank of this process (=0 for serial mcmc): this process takes points
   *                   rank*npset .. (rank+1)*npset-1 of the sequence
   * \param[in] npset  Number of parameter sets in this process
   * \param[in] nparam Number of parameters (<= MAXDIM; throws a string literal otherwise)
   * \param[in] plo    Lower bound for initial parameter guesses (float plo[nparam])
   * \param[in] phi    Upper bound for initial parameter guesses (float phi[nparam])
   * \param[out] pout  Output parameter sets (float pout[nparam*npset])
   */
  void qriguess(int rank, int npset, int nparam, const float plo[], const float phi[], float *restrict pout);
};

#endif
