/*
 * mcx_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the chain-step hot path of rplzzz/mcpar
 * (reference: src/mcpar.cc:17-214 run(), :302-312 genLocal, :315-451 genRemote,
 * :454-484 covar_setup; src/rosenbrock.cc likelihoods), with the MKL VSL RNG
 * replaced by the counter-based Philox4x32-10 streams that the north star asks
 * for, and every transcendental replaced by a fixed fp32 polynomial so that a
 * CPU run and a gfx950 run are bit-identical ("MCX arithmetic v2").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libmcx.so) never links or calls it.
 *
 * PARITY STATUS:
 *   - likelihood functors (a11-a15): pinned against the reference's own
 *     rosenbrock.cc compiled in place (oracle/_ref/libref_vlfunc.so).
 *   - step loop (a1-a10): the reference has no tests or golden vectors and
 *     mcpar.cc cannot be built here without stand-in MKL headers, so the step
 *     loop is "parity unpinned" beyond the accept rates / moments recorded in
 *     BASELINE.md, which tests/test_oracle_reference_stats.py checks statistically.
 */
#ifndef MCX_ORACLE_H_
#define MCX_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* likelihood kinds (same numbering as include/mcx.h, declared independently) */
enum {
  MCXO_VL_ROSENBROCK1 = 1, /* src/rosenbrock.cc:4-21   */
  MCXO_VL_ROSENBROCK2 = 2, /* src/rosenbrock.cc:25-41, as written (sign + boundary quirks) */
  MCXO_VL_GAUSSIAN = 3,    /* src/rosenbrock.cc:44-61, generalised to any d */
  MCXO_VL_DUALGAUSS = 4,   /* src/rosenbrock.cc:63-78  */
  MCXO_VL_GAUSSMIX = 5,    /* N-D K-component unit-variance mixture (BASELINE C5) */
  MCXO_VL_ROSENBROCK2_FIXED = 6, /* the well-posed overlapping Rosenbrock: src/rosenbrock.cc:25-41 with '+' at :38 and
                              the loop kept inside each set -- a flagged variant, not reference behaviour */
  MCXO_VL_HOST = 100       /* user callback, VLFunc::operator() src/vlfunc.hh:9-12 */
};

typedef int (*mcxo_host_fn)(void *ctx, int npset, const float *x, float *y);

typedef struct {
  int kind;
  int d;               /* parameters per set */
  int ncomp;           /* GAUSSMIX: K */
  const float *params; /* GAUSSIAN: mu[d], sig2[d]; DUALGAUSS: w; GAUSSMIX: means[K*d], weights[K] */
  mcxo_host_fn fn;     /* HOST */
  void *ctx;
} mcxo_vlfunc;

/* ---- numerics primitives (exposed so tests can pin them) ---- */
void mcxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float mcxo_logf(float x);
float mcxo_expf(float x);
float mcxo_accept_lu(uint32_t w); /* log of the acceptance draw, -inf at 0 */
void mcxo_sincos2pi(uint32_t w, float *s, float *c);
float mcxo_u24(uint32_t w);
float mcxo_uopen(uint32_t w);
void mcxo_normal4(uint32_t seed, uint32_t stream, uint32_t t, uint32_t g, uint32_t a, uint32_t q,
                  float z[4]);
int mcxo_cholesky(int d, float *a); /* in place, row-major lower; 0 ok */

/* batched likelihood: x[npset][d] row-major -> y[npset] */
int mcxo_vlfunc_eval(const mcxo_vlfunc *f, int npset, const float *x, float *y);

/* ---- engine ---- */
typedef struct mcxo_engine mcxo_engine;

/* exchange hook: replaces MPI_Allgather (src/mcpar.cc:127-140).  musigall has
 * nshards*slot floats; the hook must fill every slot but [shard] from its peers. */
typedef int (*mcxo_exchange_fn)(void *ctx, float *musigall, size_t slot_floats, int shard,
                                int nshards);

/* mirrors MCPar::MCPar (src/mcpar.hh:32-33) + seed */
mcxo_engine *mcxo_create(int np, int nc, int nshards, int shard, float pl, float armin,
                         float armax, float dfac, float ifac, int sync, uint32_t seed);
void mcxo_destroy(mcxo_engine *e);
void mcxo_set_exchange(mcxo_engine *e, mcxo_exchange_fn fn, void *ctx);
void mcxo_set_threads(mcxo_engine *e, int nthreads);
/* 1: the Murray sweep runs as plain scalar loops; 0 (default): eight Gaussians at a time with AVX2 when
 * the CPU has it -- the same IEEE operations per element, hence the same bits (checked by the tests) */
void mcxo_set_scalar_sweep(int on);
/* keep_samples: 1 = store every (chain, step) row like MCout (src/mcpar.cc:177-182) */
void mcxo_set_record(mcxo_engine *e, int keep_samples, int keep_accept_mask);
/* keep only the rows of main-loop steps with isamp % k == 0 (k = 1: every step, the reference's behaviour);
 * lets a full-size job be compared on a strided sample of its rows without holding all of them */
void mcxo_set_sample_stride(mcxo_engine *e, int k);

/* mirrors MCPar::run (src/mcpar.hh:36-37).  pinit[nc*np]; incov[np*np] or NULL */
int mcxo_run(mcxo_engine *e, int nsamp, int nburn, const float *pinit, const mcxo_vlfunc *L,
             const float *incov);
/* run all shards of a job in this process, exchanging by memcpy (engines[s].shard == s) */
int mcxo_run_all(mcxo_engine **engines, int nshards, int nsamp, int nburn,
                 const float *const *pinit, const mcxo_vlfunc *L, const float *incov);

/* results */
const float *mcxo_state(const mcxo_engine *e);     /* pvals[nc*np] */
const float *mcxo_loglike(const mcxo_engine *e);   /* lylast[nc] */
const float *mcxo_mean(const mcxo_engine *e);      /* mu[nc*np] */
const float *mcxo_var(const mcxo_engine *e);       /* sig[nc*np] (population variance) */
const float *mcxo_musigall(const mcxo_engine *e);  /* [nshards*nc*np*2] */
const float *mcxo_chol(const mcxo_engine *e);      /* T[np*np] after tuning */
const uint32_t *mcxo_accept_counts(const mcxo_engine *e); /* per chain, burn + main */
uint64_t mcxo_naccept_burn(const mcxo_engine *e);
uint64_t mcxo_naccept_main(const mcxo_engine *e);
uint64_t mcxo_remote_steps(const mcxo_engine *e);
uint64_t mcxo_remote_passes(const mcxo_engine *e);
size_t mcxo_nsample_rows(const mcxo_engine *e);
const float *mcxo_samples(const mcxo_engine *e);   /* rows of (np+1): step-major, chain, cols */
const uint8_t *mcxo_accept_mask(const mcxo_engine *e); /* [(nburn+nsamp)][nc] of last run */
int mcxo_tuner_trace(const mcxo_engine *e, float *scales, int maxn); /* T[0] after each check */

/* standalone proposal generators on caller buffers (MCPar::genLocal / genRemote, public in
 * the reference: src/mcpar.hh:40-42).  t = step index used for the RNG counters. */
int mcxo_gen_local(const mcxo_engine *e, uint32_t t, const float *pvals, float *ptrial,
                   float *cfac);
int mcxo_gen_remote(mcxo_engine *e, uint32_t t, const float *pvals, const float *musigall,
                    float *ptrial, float *cfac, float *mutrial, float *sigtrial, int *npass);

#ifdef __cplusplus
}
#endif
#endif
