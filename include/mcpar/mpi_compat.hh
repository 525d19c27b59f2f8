// mpi_compat.hh -- lets code written against the reference (which passes MPI_COMM_WORLD to MCout,
// src/mcout.hh:32) build with or without MPI.  With -DMCX_WITH_MPI the real <mpi.h> is used and
// shards exchange through MPI; without it only the single-rank path (mpisiz == 1, which the
// reference also runs without touching MPI inside MCPar: src/mcpar.cc:226) is available.
#ifndef MCPAR_AMD_MPI_COMPAT_HH_
#define MCPAR_AMD_MPI_COMPAT_HH_

#ifdef MCX_WITH_MPI
#include <mpi.h>
#else
#ifndef MPI_VERSION
typedef int MPI_Comm;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
inline int MPI_Init(int *, char ***) { return MPI_SUCCESS; }
inline int MPI_Finalize() { return MPI_SUCCESS; }
inline int MPI_Comm_size(MPI_Comm, int *size) { *size = 1; return MPI_SUCCESS; }
inline int MPI_Comm_rank(MPI_Comm, int *rank) { *rank = 0; return MPI_SUCCESS; }
#endif
#endif

// MCPar and MCout live in an inline namespace named after the build mode: user code is source-compatible
// with the reference (the names resolve unqualified), but objects built with -DMCX_WITH_MPI cannot be
// linked against the single-rank libmcpar.so or the reverse -- the symbols differ, so a mismatch is a link
// error instead of a silent difference in class layout.
#ifdef MCX_WITH_MPI
#define MCPAR_ABI_NAMESPACE_BEGIN inline namespace mcpar_mpi {
#else
#define MCPAR_ABI_NAMESPACE_BEGIN inline namespace mcpar_single {
#endif
#define MCPAR_ABI_NAMESPACE_END }

#endif
