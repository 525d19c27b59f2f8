/*
 * mcx_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See mcx_oracle.h.
 *
 * Every function cites the reference lines it restates.  "MCX arithmetic v3" 
 * is restated here independently of mcpar_amd/csrc/: nothing is shared with the product
 * except the written specification.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off + explicit fmaf() is what makes the result bit-reproducible on gfx950.
 */
#include "mcx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FPEPS 1.0e-14f /* src/mcpar.cc:15 */
#define QBLOCK 256     /* block length of the qisum summation order */

static int g_threads = 1; /* OpenMP team size of every parallel loop (mcxo_set_threads) */

/* RNG stream ids (key[1]); key[0] = seed.  DESIGN.md §3.2 */
enum { ST_LOCAL = 0, ST_ACCEPT = 1, ST_COIN = 2, ST_RSEL = 3, ST_RNORM = 4 };

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11; Random123).  Replaces VSL_BRNG_MT2203
 * (src/mcpar.cc:270-271).
 * ---------------------------------------------------------------------------------------- */
void mcxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* [0,1) on a 2^-24 grid: the role of vsRngUniform(…,0,1) (src/mcpar.cc:63,163,401) */
float mcxo_u24(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }
/* (0,1] for the Box-Muller radius */
float mcxo_uopen(uint32_t w) { return fmaf((float)w, 0x1p-32f, 0x1p-33f); }

/* natural log, x > 0 normal.  Cephes-style: frexp to [sqrt(1/2), sqrt(2)), degree-9 polynomial. */
float mcxo_logf(float x)
{
  uint32_t b = f2bits(x);
  int e = (int)((b >> 23) & 0xffu) - 126;
  float m = bits2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
  if (m < 0.70710678f) { e -= 1; m = (m + m) - 1.0f; }
  else { m = m - 1.0f; }
  float fe = (float)e;
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  y = fmaf(-2.12194440e-4f, fe, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  r = fmaf(0.693359375f, fe, r);
  return r;
}

/* exp = 2^n * e^r, n = floor(x log2(e) + 1/2).  n > 127 -> +inf; n < -125 -> 0 (results are normal or
 * zero, never denormal, so the scaling is an exact add to the exponent field); NaN propagates. */
float mcxo_expf(float x)
{
  float fn = floorf(fmaf(x, 1.44269504f, 0.5f));
  float r = fmaf(fn, -0.693359375f, x);
  r = fmaf(fn, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float z = r * r;
  float y = fmaf(p, z, r);
  y = y + 1.0f;
  if (!(x == x)) return y; /* NaN: (int)fn is undefined in C, skip the scaling */
  if (fn > 127.0f) return INFINITY;
  if (fn < -125.0f) return 0.0f;
  return bits2f(f2bits(y) + ((uint32_t)(int)fn << 23));
}

/* log of the acceptance draw u24(w) in [0,1): -inf at 0 (local steps test log u < ly' - ly) */
float mcxo_accept_lu(uint32_t w) { return (w >> 8) == 0u ? -INFINITY : mcxo_logf(mcxo_u24(w)); }

/* sin/cos of 2*pi*w/2^32 by quadrant reduction on the integer + Cephes sinf/cosf kernels */
void mcxo_sincos2pi(uint32_t w, float *s, float *c)
{
  uint32_t k = ((w + 0x20000000u) >> 30) & 3u;
  int32_t rem = (int32_t)(w - (k << 30));
  float phi = (float)rem * 1.4629180792671596e-9f; /* 2*pi/2^32 */
  float z = phi * phi;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float sp = fmaf(phi * z, ps, phi);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  switch (k) {
  case 0: *s = sp; *c = cp; break;
  case 1: *s = cp; *c = -sp; break;
  case 2: *s = -sp; *c = -cp; break;
  default: *s = -cp; *c = sp; break;
  }
}

/* Box-Muller, both branches used (the role of VSL_RNG_METHOD_GAUSSIAN_BOXMULLER2,
 * src/mcpar.cc:306,348): words (0,1) -> z0,z1 ; words (2,3) -> z2,z3 */
static void normal4_from_words(const uint32_t w[4], float z[4])
{
  for (int h = 0; h < 2; ++h) {
    float u = mcxo_uopen(w[2 * h]);
    float r = sqrtf(-2.0f * mcxo_logf(u));
    float s, c;
    mcxo_sincos2pi(w[2 * h + 1], &s, &c);
    z[2 * h] = r * c;
    z[2 * h + 1] = r * s;
  }
}

void mcxo_normal4(uint32_t seed, uint32_t stream, uint32_t t, uint32_t g, uint32_t a, uint32_t q,
                  float z[4])
{
  uint32_t ctr[4] = {t, g, a, q}, key[2] = {seed, stream}, w[4];
  mcxo_philox4x32_10(ctr, key, w);
  normal4_from_words(w, z);
}

/* Cholesky, lower, row-major, in place; strict upper triangle zeroed.  The role of
 * spotrf('U') on the column-major view (src/mcpar.cc:470-480). */
int mcxo_cholesky(int d, float *a)
{
  for (int i = 0; i < d; ++i) {
    for (int j = 0; j <= i; ++j) {
      float s = a[i * d + j];
      for (int k = 0; k < j; ++k) s = fmaf(-a[i * d + k], a[j * d + k], s);
      if (i == j) {
        if (!(s > 0.0f)) return i + 1;
        a[i * d + i] = sqrtf(s);
      } else {
        a[i * d + j] = s / a[j * d + j];
      }
    }
    for (int j = i + 1; j < d; ++j) a[i * d + j] = 0.0f;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Likelihoods.  Summation order: parameters are grouped in blocks of 4;
 * inside a block terms are added left to right starting from 0; block partials are combined
 * by an xor-butterfly over the block index padded with zeros to a power of two.
 * ---------------------------------------------------------------------------------------- */
static float butterfly_sum(float *p, int nb)
{
  int p2 = 1;
  while (p2 < nb) p2 <<= 1;
  float tmp[64], nxt[64];
  for (int q = 0; q < p2; ++q) tmp[q] = q < nb ? p[q] : 0.0f;
  for (int s = 1; s < p2; s <<= 1) {
    for (int q = 0; q < p2; ++q) nxt[q] = tmp[q] + tmp[q ^ s];
    memcpy(tmp, nxt, sizeof(float) * (size_t)p2);
  }
  return tmp[0];
}

/* src/rosenbrock.cc:4-21 */
static float rosen1_one(int d, const float *x)
{
  float part[64];
  int nb = (d + 3) / 4;
  for (int q = 0; q < nb; ++q) {
    float acc = 0.0f;
    for (int k = 4 * q; k + 1 < d && k < 4 * q + 4; k += 2) {
      float t1 = 1.0f - x[k];
      float t2 = fmaf(-x[k], x[k], x[k + 1]);
      float term = fmaf(100.0f * t2, t2, t1 * t1);
      acc = acc + term;
    }
    part[q] = acc;
  }
  return 0.0f - butterfly_sum(part, nb); /* 0 - s like the reference's fx = 0; fx -= ...: never -0 */
}

/* The well-posed overlapping N-D Rosenbrock function: the reference's Rosenbrock2 (src/rosenbrock.cc:25-41) with the
 * sign of its second term corrected ('+' like :18 instead of the '-' at :38, which makes log L unbounded above)
 * and the loop kept inside each parameter set (the reference's flat loop reads x[i+1] across the set boundary):
 * fx = - sum_{k=0}^{d-2} (1 - x_k)^2 + 100 (x_{k+1} - x_k^2)^2.  NOT in the reference: a flagged variant
 * (SURVEY §7 step 1, §8d), written here, not a patched copy.  Term k belongs to block k / 4. */
static float rosen2f_one(int d, const float *x)
{
  float part[64];
  int nb = (d + 3) / 4;
  for (int q = 0; q < nb; ++q) {
    float acc = 0.0f;
    for (int k = 4 * q; k + 1 < d && k < 4 * q + 4; ++k) {
      float t1 = 1.0f - x[k];
      float t2 = fmaf(-x[k], x[k], x[k + 1]);
      float term = fmaf(100.0f * t2, t2, t1 * t1);
      acc = acc + term;
    }
    part[q] = acc;
  }
  return 0.0f - butterfly_sum(part, nb);
}

/* src/rosenbrock.cc:44-61, any d (reference throws unless d == 2: src/rosenbrock.hh:43) */
static float gauss_one(int d, const float *x, const float *mu, const float *s2inv)
{
  float part[64];
  int nb = (d + 3) / 4;
  for (int q = 0; q < nb; ++q) {
    float acc = 0.0f;
    for (int k = 4 * q; k < d && k < 4 * q + 4; ++k) {
      float a = x[k] - mu[k];
      acc = fmaf((0.5f * a) * a, s2inv[k], acc);
    }
    part[q] = acc;
  }
  return 0.0f - butterfly_sum(part, nb); /* 0 - s like the reference's fx = 0; fx -= ...: never -0 */
}

/* log sum_k w_k exp(-|x-m_k|^2/2), evaluated as a log-sum-exp.  d=2,K=2,m={0,5},w={w,1}
 * is DualGaussian (src/rosenbrock.cc:63-78). */
static float mix_one(int d, int K, const float *x, const float *means, const float *logw)
{
  float e[64];
  int nb = (d + 3) / 4;
  for (int c = 0; c < K; ++c) {
    float part[64];
    for (int q = 0; q < nb; ++q) {
      float acc = 0.0f;
      for (int k = 4 * q; k < d && k < 4 * q + 4; ++k) {
        float a = x[k] - means[c * d + k];
        acc = fmaf(a, a, acc);
      }
      part[q] = acc;
    }
    e[c] = fmaf(-0.5f, butterfly_sum(part, nb), logw[c]);
  }
  float emax = e[0];
  for (int c = 1; c < K; ++c) emax = e[c] > emax ? e[c] : emax;
  float s = 0.0f;
  for (int c = 0; c < K; ++c) s = s + mcxo_expf(e[c] - emax);
  return emax + mcxo_logf(s);
}

int mcxo_vlfunc_eval(const mcxo_vlfunc *f, int npset, const float *x, float *y)
{
  const int d = f->d;
  if (d < 1 || d > 256) return -1;
  switch (f->kind) {
  case MCXO_VL_ROSENBROCK1:
    if (d < 2 || (d & 1)) return -1; /* src/rosenbrock.hh:13-16 */
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) y[j] = rosen1_one(d, x + (size_t)j * d);
    return 0;
  case MCXO_VL_ROSENBROCK2: {
    /* as written: flat loop over i < ntot-1, x[i+1] crosses the set boundary,
     * sign of the second term is '-' (src/rosenbrock.cc:32-38) */
    if (d < 2) return -1;
    const long ntot = (long)npset * d;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) {
      float acc = 0.0f;
      for (long i = (long)j * d; i < (long)(j + 1) * d; ++i) {
        if (i < ntot - 1) {
          float t1 = 1.0f - x[i];
          float t2 = fmaf(-x[i], x[i], x[i + 1]);
          float term = fmaf(-(100.0f * t2), t2, t1 * t1);
          acc = acc + term;
        }
      }
      y[j] = 0.0f - acc;
    }
    return 0;
  }
  case MCXO_VL_ROSENBROCK2_FIXED:
    if (d < 2) return -1;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) y[j] = rosen2f_one(d, x + (size_t)j * d);
    return 0;
  case MCXO_VL_GAUSSIAN: {
    float mu[256], s2inv[256];
    for (int k = 0; k < d; ++k) {
      mu[k] = f->params ? f->params[k] : 0.0f;
      s2inv[k] = f->params ? 1.0f / f->params[d + k] : 1.0f; /* src/rosenbrock.hh:44-47 */
    }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) y[j] = gauss_one(d, x + (size_t)j * d, mu, s2inv);
    return 0;
  }
  case MCXO_VL_DUALGAUSS: {
    if (d != 2) return -1;
    const float means[4] = {0.0f, 0.0f, 5.0f, 5.0f}; /* src/rosenbrock.cc:71-72 */
    const float logw[2] = {mcxo_logf(f->params[0]), 0.0f};
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) y[j] = mix_one(2, 2, x + (size_t)j * 2, means, logw);
    return 0;
  }
  case MCXO_VL_GAUSSMIX: {
    const int K = f->ncomp;
    if (K < 1 || K > 64) return -1;
    float logw[64];
    for (int c = 0; c < K; ++c) logw[c] = mcxo_logf(f->params[(size_t)K * d + c]);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < npset; ++j) y[j] = mix_one(d, K, x + (size_t)j * d, f->params, logw);
    return 0;
  }
  case MCXO_VL_HOST:
    return f->fn(f->ctx, npset, x, y); /* src/vlfunc.hh:9-12 */
  default:
    return -1;
  }
}

/* ------------------------------------------------------------------------------------------
 * Engine: field names follow src/mcpar.hh:44-90
 * ---------------------------------------------------------------------------------------- */
struct mcxo_engine {
  int nparam, nchain, ntot, ncov;
  int size, rank, tchains;
  float PLOCAL, TGT_ARATE_MIN, TGT_ARATE_MAX, SCALE_DEC, SCALE_INC;
  int SYNCSTEP;
  uint32_t seed;
  uint32_t tbase; /* RNG step counter carried across run() calls */
  float *pvals, *ptrial, *mu, *sig, *mutrial, *sigtrial, *musigall, *winvall;
  float *lylast, *lytrial, *cfac, *pacpt, *acpt, *qisum, *qimax, *cmax;
  int *rjct, *chnsel;
  float *cov, *psum2;
  uint8_t *take;
  /* run state */
  const mcxo_vlfunc *L;
  int nsamp, nburn, irate;
  uint64_t tun_ntrial, tun_naccept;
  float pwgt;
  /* records */
  int keep_samples, keep_mask, nthreads;
  int sample_stride; /* keep the rows of main-loop steps with isamp % sample_stride == 0 (1 = every step, like the reference) */
  float *samples; size_t nrows, caprows;
  uint8_t *amask;
  uint32_t *acounts;
  uint64_t nacc_burn, nacc_main, nremote_steps, nremote_passes;
  float trace[256]; int ntrace;
  mcxo_exchange_fn xfn; void *xctx;
};

static void *zalloc(size_t n) { return calloc(n ? n : 1, 1); }

/* src/mcpar.cc:216-272 */
mcxo_engine *mcxo_create(int np, int nc, int nshards, int shard, float pl, float armin,
                         float armax, float dfac, float ifac, int sync, uint32_t seed)
{
  if (np < 1 || np > 256 || nc < 1 || nshards < 1 || shard < 0 || shard >= nshards || sync < 1)
    return NULL;
  mcxo_engine *e = (mcxo_engine *)zalloc(sizeof(*e));
  e->nparam = np; e->nchain = nc; e->ntot = np * nc; e->ncov = np * np;
  e->size = nshards; e->rank = shard; e->tchains = nshards * nc;
  e->PLOCAL = pl; e->TGT_ARATE_MIN = armin; e->TGT_ARATE_MAX = armax;
  e->SCALE_DEC = dfac; e->SCALE_INC = ifac; e->SYNCSTEP = sync; e->seed = seed;
  size_t nt = (size_t)e->ntot, n = (size_t)nc;
  e->pvals = zalloc(4 * nt); e->ptrial = zalloc(4 * nt); e->mu = zalloc(4 * nt);
  e->sig = zalloc(4 * nt); e->mutrial = zalloc(4 * nt); e->sigtrial = zalloc(4 * nt);
  e->psum2 = zalloc(4 * nt);
  e->musigall = zalloc(8 * (size_t)e->tchains * np);
  e->winvall = zalloc(8 * (size_t)e->tchains * np); /* (mu, 1/sig2) pairs of the Murray sweep */
  e->lylast = zalloc(4 * n); e->lytrial = zalloc(4 * n); e->cfac = zalloc(4 * n);
  e->pacpt = zalloc(4 * n); e->acpt = zalloc(4 * n); e->qisum = zalloc(4 * n);
  e->qimax = zalloc(4 * n); e->cmax = zalloc(4 * n);
  e->rjct = zalloc(sizeof(int) * n); e->chnsel = zalloc(sizeof(int) * n);
  e->take = zalloc(n);
  e->cov = zalloc(4 * (size_t)e->ncov);
  e->acounts = zalloc(4 * n);
  for (int i = 0; i < np; ++i) e->cov[i * (np + 1)] = 1.0f; /* identity until covar_setup */
  e->keep_samples = 1; e->keep_mask = 1; e->nthreads = 1; e->sample_stride = 1;
  return e;
}

void mcxo_destroy(mcxo_engine *e)
{
  if (!e) return;
  free(e->pvals); free(e->ptrial); free(e->mu); free(e->sig); free(e->mutrial);
  free(e->sigtrial); free(e->psum2); free(e->musigall); free(e->winvall); free(e->lylast);
  free(e->lytrial); free(e->cfac); free(e->pacpt); free(e->acpt); free(e->qisum);
  free(e->qimax); free(e->cmax); free(e->rjct); free(e->chnsel); free(e->take); free(e->cov);
  free(e->acounts); free(e->samples); free(e->amask);
  free(e);
}

void mcxo_set_exchange(mcxo_engine *e, mcxo_exchange_fn fn, void *ctx) { e->xfn = fn; e->xctx = ctx; }
void mcxo_set_threads(mcxo_engine *e, int nthreads)
{
  e->nthreads = nthreads > 0 ? nthreads : 1;
  g_threads = e->nthreads;
}
void mcxo_set_record(mcxo_engine *e, int ks, int km) { e->keep_samples = ks; e->keep_mask = km; }
void mcxo_set_sample_stride(mcxo_engine *e, int k) { e->sample_stride = k > 0 ? k : 1; }

/* src/mcpar.cc:454-484 */
static int covar_setup(mcxo_engine *e, const float *incov)
{
  const int d = e->nparam;
  if (incov) memcpy(e->cov, incov, sizeof(float) * (size_t)e->ncov);
  else {
    memset(e->cov, 0, sizeof(float) * (size_t)e->ncov);
    for (int i = 0; i < d; ++i) e->cov[i * (d + 1)] = 1.0f;
  }
  return mcxo_cholesky(d, e->cov);
}

static inline uint32_t gchain(const mcxo_engine *e, int j) { return (uint32_t)(e->rank * e->nchain + j); }

/* src/mcpar.cc:302-312: ptrial_j = pvals_j + T z_j ; cfac_j = 1 */
int mcxo_gen_local(const mcxo_engine *e, uint32_t t, const float *pvals, float *ptrial,
                   float *cfac)
{
  const int d = e->nparam, nb = (d + 3) / 4;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int j = 0; j < e->nchain; ++j) {
    float z[256];
    for (int q = 0; q < nb; ++q) mcxo_normal4(e->seed, ST_LOCAL, t, gchain(e, j), (uint32_t)q, 0, z + 4 * q);
    for (int i = 0; i < d; ++i) {
      float acc = pvals[(size_t)j * d + i];
      for (int k = 0; k <= i; ++k) acc = fmaf(e->cov[i * d + k], z[k], acc);
      ptrial[(size_t)j * d + i] = acc;
    }
    cfac[j] = 1.0f;
  }
  return 0;
}

/* sum_k (mu_qk - x_k)^2 / sig2_qk exactly as src/mcpar.cc:369-383 forms it -- xm = mu - x first (so a chain
 * sitting on a Gaussian's mean gives exactly 0 however narrow the Gaussian), then xm*xm -- with the division
 * replaced by a multiplication with w = 1/sig2, precomputed once per Gaussian ("MCX arithmetic v3"; v2's
 * fma(-x, s, mu s) lost the cancellation when |mu| s was large).  qp = the Gaussian's (mu, 1/sig2) pairs. */
static inline float qarg(int d, const float *qp, const float *x)
{
  float arg = 0.0f;
  for (int k = 0; k < d; ++k) {
    float xm = qp[2 * k] - x[k];
    arg = fmaf(xm * xm, qp[2 * k + 1], arg);
  }
  return arg;
}

/* The all-pairs sweep of one chain vector x over the N per-chain Gaussians (src/mcpar.cc:367-395 for
 * ptrial): Q_i = exp(-arg_i/2), running maximum from FPEPS, and the sum in the order of DESIGN.md §3.5: the
 * N terms in blocks of QBLOCK consecutive Q_i, each block summed left to right from 0, block sums added left
 * to right onto FPEPS (so that a GPU can sweep the blocks in parallel and still be bit-exact).
 * qs_out == NULL (src/mcpar.cc:421-437, the numerator of cfac): only max_i Q_i is wanted, which is
 * exp(-min_i arg_i / 2) -- one exp per chain instead of one per pair.  Plain scalar statement. */
static void sweep_scalar(int d, int N, const float *qpar, const float *x, float *qs_out, float *qm_out)
{
  if (!qs_out) {
    float amin = INFINITY;
    for (int qi = 0; qi < N; ++qi) {
      float a = qarg(d, qpar + 2 * (size_t)qi * d, x);
      amin = a < amin ? a : amin;
    }
    *qm_out = mcxo_expf(-0.5f * amin);
    return;
  }
  float qs = FPEPS, qm = FPEPS;
  for (int b0 = 0; b0 < N; b0 += QBLOCK) {
    float part = 0.0f;
    for (int qi = b0; qi < N && qi < b0 + QBLOCK; ++qi) {
      float gv = mcxo_expf(-0.5f * qarg(d, qpar + 2 * (size_t)qi * d, x));
      part = part + gv;
      qm = gv > qm ? gv : qm;
    }
    qs = qs + part;
  }
  *qs_out = qs;
  *qm_out = qm;
}

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define MCXO_HAVE_AVX2 1
/* The same sweep, eight Q_i at a time: every lane performs exactly the scalar sequence of IEEE
 * operations (sub, mul, fma per dimension, the mcxo_expf polynomial), the eight results are then added / compared
 * one by one in index order -- same bits as sweep_scalar (tests/test_oracle_numerics.py checks it),
 * ~8x faster, which is what makes oracle comparisons at 32 768 - 65 536 chains affordable.
 * qt = (mu, 1/sig2) of the Gaussians regrouped in blocks of 8: qt[(b*d + k)*16 + {0..7 | 8..15}]. */
__attribute__((target("avx2,fma"))) static inline __m256 expf8(__m256 x)
{
  const __m256 fn = _mm256_floor_ps(_mm256_fmadd_ps(x, _mm256_set1_ps(1.44269504f), _mm256_set1_ps(0.5f)));
  __m256 r = _mm256_fmadd_ps(fn, _mm256_set1_ps(-0.693359375f), x);
  r = _mm256_fmadd_ps(fn, _mm256_set1_ps(2.12194440e-4f), r);
  __m256 p = _mm256_set1_ps(1.9875691500e-4f);
  p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(1.3981999507e-3f));
  p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(8.3334519073e-3f));
  p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(4.1665795894e-2f));
  p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(1.6666665459e-1f));
  p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(5.0000001201e-1f));
  const __m256 z = _mm256_mul_ps(r, r);
  __m256 y = _mm256_fmadd_ps(p, z, r);
  y = _mm256_add_ps(y, _mm256_set1_ps(1.0f));
  const __m256i n = _mm256_cvttps_epi32(fn);
  __m256 ys = _mm256_castsi256_ps(_mm256_add_epi32(_mm256_castps_si256(y), _mm256_slli_epi32(n, 23)));
  const __m256 isnan = _mm256_cmp_ps(x, x, _CMP_UNORD_Q);
  const __m256 big = _mm256_cmp_ps(fn, _mm256_set1_ps(127.0f), _CMP_GT_OQ);
  const __m256 small = _mm256_cmp_ps(fn, _mm256_set1_ps(-125.0f), _CMP_LT_OQ);
  ys = _mm256_blendv_ps(ys, y, isnan); /* NaN: unscaled, like the scalar early return */
  ys = _mm256_blendv_ps(ys, _mm256_setzero_ps(), small);
  ys = _mm256_blendv_ps(ys, _mm256_set1_ps(INFINITY), big);
  return ys;
}

__attribute__((target("avx2,fma"))) static void sweep_avx2(int d, int N, const float *qpar, const float *qt,
                                                            const float *x, float *qs_out, float *qm_out)
{
  const __m256 mhalf = _mm256_set1_ps(-0.5f);
  if (!qs_out) { /* min_i arg_i: exact in any order */
    __m256 vmin = _mm256_set1_ps(INFINITY);
    int qi = 0;
    for (; qi + 8 <= N; qi += 8) {
      const float *q = qt + (size_t)(qi >> 3) * d * 16;
      __m256 arg = _mm256_setzero_ps();
      for (int k = 0; k < d; ++k) {
        const __m256 xm = _mm256_sub_ps(_mm256_loadu_ps(q + 16 * k), _mm256_set1_ps(x[k]));
        arg = _mm256_fmadd_ps(_mm256_mul_ps(xm, xm), _mm256_loadu_ps(q + 16 * k + 8), arg);
      }
      vmin = _mm256_min_ps(arg, vmin); /* a NaN arg leaves vmin as it is, like `a < amin ? a : amin` */
    }
    float lanes[8], amin = INFINITY;
    _mm256_storeu_ps(lanes, vmin);
    for (int l = 0; l < 8; ++l) amin = lanes[l] < amin ? lanes[l] : amin;
    for (; qi < N; ++qi) {
      float a = qarg(d, qpar + 2 * (size_t)qi * d, x);
      amin = a < amin ? a : amin;
    }
    *qm_out = mcxo_expf(-0.5f * amin);
    return;
  }
  float qs = FPEPS, qm = FPEPS;
  for (int b0 = 0; b0 < N; b0 += QBLOCK) {
    float part = 0.0f;
    const int b1 = b0 + QBLOCK < N ? b0 + QBLOCK : N;
    int qi = b0;
    for (; qi + 8 <= b1; qi += 8) {
      const float *q = qt + (size_t)(qi >> 3) * d * 16;
      __m256 arg = _mm256_setzero_ps();
      for (int k = 0; k < d; ++k) {
        const __m256 xm = _mm256_sub_ps(_mm256_loadu_ps(q + 16 * k), _mm256_set1_ps(x[k]));
        arg = _mm256_fmadd_ps(_mm256_mul_ps(xm, xm), _mm256_loadu_ps(q + 16 * k + 8), arg);
      }
      float gv[8];
      _mm256_storeu_ps(gv, expf8(_mm256_mul_ps(mhalf, arg)));
      for (int l = 0; l < 8; ++l) {
        part = part + gv[l];
        qm = gv[l] > qm ? gv[l] : qm;
      }
    }
    for (; qi < b1; ++qi) {
      float gv = mcxo_expf(-0.5f * qarg(d, qpar + 2 * (size_t)qi * d, x));
      part = part + gv;
      qm = gv > qm ? gv : qm;
    }
    qs = qs + part;
  }
  *qs_out = qs;
  *qm_out = qm;
}
#endif

static int g_force_scalar = -1; /* MCXO_SCALAR_SWEEP=1 or mcxo_set_scalar_sweep(1): plain loops only */
void mcxo_set_scalar_sweep(int on) { g_force_scalar = on ? 1 : 0; }

static int use_avx2(void)
{
#ifdef MCXO_HAVE_AVX2
  if (g_force_scalar < 0) {
    const char *s = getenv("MCXO_SCALAR_SWEEP");
    g_force_scalar = (s && s[0] == '1') ? 1 : 0;
  }
  return !g_force_scalar && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
#else
  return 0;
#endif
}

/* src/mcpar.cc:315-451 */
int mcxo_gen_remote(mcxo_engine *e, uint32_t t, const float *pvals, const float *musigall,
                    float *ptrial, float *cfac, float *mutrial, float *sigtrial, int *npass_out)
{
  const int d = e->nparam, n = e->nchain, N = e->tchains, nb = (d + 3) / 4;
  float *qpar = e->winvall; /* (mu, w) pairs, w = 1/sig2 */
  for (size_t i = 0; i < (size_t)N * d; ++i) {
    qpar[2 * i] = musigall[2 * i];
    qpar[2 * i + 1] = 1.0f / musigall[2 * i + 1];
  }
  const int vec = use_avx2();
  float *qt = NULL;
#ifdef MCXO_HAVE_AVX2
  if (vec && N >= 8) {
    qt = (float *)malloc(sizeof(float) * (size_t)(N / 8) * d * 16);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int b = 0; b < N / 8; ++b)
      for (int k = 0; k < d; ++k)
        for (int l = 0; l < 8; ++l) {
          qt[((size_t)b * d + k) * 16 + l] = qpar[2 * ((size_t)(8 * b + l) * d + k)];
          qt[((size_t)b * d + k) * 16 + 8 + l] = qpar[2 * ((size_t)(8 * b + l) * d + k) + 1];
        }
  }
#define SWEEP(x, qs, qm)                                       \
  do {                                                         \
    if (qt) sweep_avx2(d, N, qpar, qt, (x), (qs), (qm));       \
    else sweep_scalar(d, N, qpar, (x), (qs), (qm));            \
  } while (0)
#else
  (void)vec;
#define SWEEP(x, qs, qm) sweep_scalar(d, N, qpar, (x), (qs), (qm))
#endif
  /* numerator of cfac, max_i Q_i(pvals): independent of the pass (src/mcpar.cc:421-437) */
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
  for (int j = 0; j < n; ++j) SWEEP(pvals + (size_t)j * d, NULL, &e->cmax[j]);
  for (int j = 0; j < n; ++j) e->rjct[j] = 1; /* :329-331 */
  int anyrjct, pass = 0;
  do {
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
    for (int j = 0; j < n; ++j) {
      if (!e->rjct[j]) continue;
      uint32_t ctr[4] = {t, gchain(e, j), (uint32_t)pass, 0}, key[2] = {e->seed, ST_RSEL}, w[4];
      mcxo_philox4x32_10(ctr, key, w);
      int sel = (int)(((uint64_t)w[0] * (uint64_t)N) >> 32); /* :337 */
      e->chnsel[j] = sel;
      float z[256];
      for (int q = 0; q < nb; ++q)
        mcxo_normal4(e->seed, ST_RNORM, t, gchain(e, j), (uint32_t)pass, (uint32_t)q, z + 4 * q);
      for (int i = 0; i < d; ++i) { /* :339-352 */
        size_t ci = (size_t)j * d + i;
        mutrial[ci] = musigall[2 * ((size_t)sel * d + i)];
        sigtrial[ci] = sqrtf(musigall[2 * ((size_t)sel * d + i) + 1]);
        ptrial[ci] = fmaf(sigtrial[ci], z[i], mutrial[ci]);
      }
      /* :355-395, qisum / qimax seeded with FPEPS */
      float qs, qm;
      SWEEP(ptrial + (size_t)j * d, &qs, &qm);
      e->qisum[j] = qs; e->qimax[j] = qm;
      e->pacpt[j] = qm / qs;        /* :397-398 */
      e->acpt[j] = mcxo_u24(w[1]);  /* :401 */
      if (e->acpt[j] < e->pacpt[j]) { /* :405-441 */
        e->rjct[j] = 0;
        cfac[j] = e->cmax[j] / qm;
      }
    }
    anyrjct = 0;
    for (int j = 0; j < n; ++j) anyrjct += e->rjct[j];
    ++pass;
  } while (anyrjct); /* :443 */
#undef SWEEP
  free(qt);
  for (size_t i = 0; i < (size_t)n * d; ++i) sigtrial[i] = sigtrial[i] * sigtrial[i]; /* :447-448 */
  if (npass_out) *npass_out = pass;
  return 0;
}

static void ensure_rows(mcxo_engine *e, size_t add)
{
  if (e->nrows + add <= e->caprows) return;
  e->caprows = e->nrows + add;
  e->samples = (float *)realloc(e->samples, sizeof(float) * e->caprows * (size_t)(e->nparam + 1));
}

/* accept / reject (src/mcpar.cc:62-75, 162-175); returns number accepted */
/* Local steps (cfac = 1) test the same inequality in the log domain, log u < ly' - ly, so that the only
 * transcendental of the test depends on the random draw alone; Murray steps keep u < exp(ly' - ly) cfac. */
static uint64_t accept_all(mcxo_engine *e, uint32_t t, size_t maskrow, int remotep)
{
  const int d = e->nparam, n = e->nchain;
  uint64_t nacc = 0;
#pragma omp parallel for schedule(static) reduction(+ : nacc) num_threads(g_threads)
  for (int j = 0; j < n; ++j) {
    uint32_t ctr[4] = {t >> 2, gchain(e, j), 0, 0}, key[2] = {e->seed, ST_ACCEPT}, w[4];
    mcxo_philox4x32_10(ctr, key, w);
    int take;
    if (remotep) {
      e->acpt[j] = mcxo_u24(w[t & 3u]);
      e->pacpt[j] = mcxo_expf(e->lytrial[j] - e->lylast[j]) * e->cfac[j];
      take = e->acpt[j] < e->pacpt[j];
    } else {
      take = mcxo_accept_lu(w[t & 3u]) < e->lytrial[j] - e->lylast[j];
    }
    e->take[j] = (uint8_t)take;
    if (take) {
      e->lylast[j] = e->lytrial[j];
      memcpy(e->pvals + (size_t)j * d, e->ptrial + (size_t)j * d, sizeof(float) * (size_t)d);
      e->acounts[j] += 1;
      nacc += 1;
    }
    if (e->amask) e->amask[maskrow * (size_t)n + j] = (uint8_t)take;
  }
  return nacc;
}

static int run_begin(mcxo_engine *e, int nsamp, int nburn, const float *pinit, const mcxo_vlfunc *L,
                     const float *incov)
{
  if (L->d != e->nparam) return -2;
  int st = covar_setup(e, incov); /* :20 */
  if (st) return -3;
  e->L = L; e->nsamp = nsamp; e->nburn = nburn;
  e->tun_ntrial = e->tun_naccept = 0; e->irate = 50; e->ntrace = 0;
  e->nacc_burn = e->nacc_main = 0; e->nremote_steps = e->nremote_passes = 0;
  memset(e->acounts, 0, 4 * (size_t)e->nchain);
  if (e->keep_samples) ensure_rows(e, (size_t)((nsamp + e->sample_stride - 1) / e->sample_stride) * e->nchain); /* :31 */
  free(e->amask); e->amask = NULL;
  if (e->keep_mask) e->amask = zalloc((size_t)(nburn + nsamp) * e->nchain);
  memcpy(e->pvals, pinit, sizeof(float) * (size_t)e->ntot); /* :47-50 */
  return mcxo_vlfunc_eval(L, e->nchain, e->pvals, e->lylast); /* :53 */
}

/* one burn-in step (src/mcpar.cc:58-97) */
static void burn_step(mcxo_engine *e, int isamp)
{
  const uint32_t t = e->tbase + (uint32_t)isamp;
  mcxo_gen_local(e, t, e->pvals, e->ptrial, e->cfac);
  mcxo_vlfunc_eval(e->L, e->nchain, e->ptrial, e->lytrial);
  e->tun_ntrial += (uint64_t)e->nchain;
  uint64_t na = accept_all(e, t, (size_t)isamp, 0);
  e->tun_naccept += na; e->nacc_burn += na;
  if (isamp > e->irate) { /* :78 */
    float arate = (float)e->tun_naccept / (float)e->tun_ntrial;
    if (arate < e->TGT_ARATE_MIN) {
      e->tun_naccept = e->tun_ntrial = 0;
      for (int i = 0; i < e->ncov; ++i) e->cov[i] *= e->SCALE_DEC;
    } else if (arate > e->TGT_ARATE_MAX) {
      e->tun_naccept = e->tun_ntrial = 0;
      for (int i = 0; i < e->ncov; ++i) e->cov[i] *= e->SCALE_INC;
    }
    e->irate += 50;
    if (e->ntrace < 256) e->trace[e->ntrace++] = e->cov[0];
  }
}

/* src/mcpar.cc:99-104 */
static void burn_end(mcxo_engine *e)
{
  for (int i = 0; i < e->ntot; ++i) { e->mu[i] = 0.0f; e->psum2[i] = FPEPS; }
  e->pwgt = 0.0f;
}

/* one main-loop step without the exchange (src/mcpar.cc:142-209) */
static void main_step(mcxo_engine *e, int isamp)
{
  const int d = e->nparam, n = e->nchain;
  const uint32_t t = e->tbase + (uint32_t)e->nburn + (uint32_t)isamp;
  float rndlocal;
  if (isamp < e->SYNCSTEP) rndlocal = 0.0f; /* :143-144 */
  else {
    /* one coin per step for the whole job (the reference draws one per rank, :146) */
    uint32_t ctr[4] = {t, 0, 0, 0}, key[2] = {e->seed, ST_COIN}, w[4];
    mcxo_philox4x32_10(ctr, key, w);
    rndlocal = mcxo_u24(w[0]);
  }
  int remotep;
  if (rndlocal <= e->PLOCAL) { /* :152 */
    mcxo_gen_local(e, t, e->pvals, e->ptrial, e->cfac);
    remotep = 0;
  } else {
    int npass = 0;
    mcxo_gen_remote(e, t, e->pvals, e->musigall, e->ptrial, e->cfac, e->mutrial, e->sigtrial, &npass);
    remotep = 1;
    e->nremote_steps += 1; e->nremote_passes += (uint64_t)npass;
  }
  mcxo_vlfunc_eval(e->L, n, e->ptrial, e->lytrial); /* :160 */
  e->nacc_main += accept_all(e, t, (size_t)e->nburn + (size_t)isamp, remotep);
  /* :177-182 (the discarded single-set L call at :180 has no effect and is not made) */
  if (e->keep_samples && isamp % e->sample_stride == 0) {
    float *row = e->samples + e->nrows * (size_t)(d + 1);
    for (int j = 0; j < n; ++j, row += d + 1) {
      memcpy(row, e->pvals + (size_t)j * d, sizeof(float) * (size_t)d);
      row[d] = e->lylast[j];
    }
    e->nrows += (size_t)n;
  }
  /* :186-209 */
  e->pwgt += 1.0f;
  const float pwgt = e->pwgt, winv = 1.0f / pwgt;
  float *slot = e->musigall + 2 * (size_t)e->rank * e->ntot;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < d; ++k) {
      size_t i = (size_t)j * d + k;
      if (remotep && e->take[j]) {
        e->mu[i] = e->mutrial[i];
        e->sig[i] = e->sigtrial[i];
        e->psum2[i] = e->sig[i] * (pwgt - 1.0f);
      }
      float delta = e->pvals[i] - e->mu[i];
      e->mu[i] = fmaf(delta, winv, e->mu[i]);
      e->psum2[i] = fmaf(delta, e->pvals[i] - e->mu[i], e->psum2[i]);
      e->sig[i] = e->psum2[i] * winv;
      slot[2 * i] = e->mu[i];
      slot[2 * i + 1] = e->sig[i];
    }
  }
}

int mcxo_run(mcxo_engine *e, int nsamp, int nburn, const float *pinit, const mcxo_vlfunc *L,
             const float *incov)
{
  if (e->size > 1 && !e->xfn) return -4;
  int st = run_begin(e, nsamp, nburn, pinit, L, incov);
  if (st) return st;
  for (int isamp = 0; isamp < nburn; ++isamp) burn_step(e, isamp);
  burn_end(e);
  for (int isamp = 0; isamp < nsamp; ++isamp) {
    if (isamp % e->SYNCSTEP == 0 && e->size > 1) { /* :127-140 */
      st = e->xfn(e->xctx, e->musigall, 2 * (size_t)e->ntot, e->rank, e->size);
      if (st) return st;
    }
    main_step(e, isamp);
  }
  e->tbase += (uint32_t)(nburn + nsamp);
  return 0;
}

int mcxo_run_all(mcxo_engine **eng, int nshards, int nsamp, int nburn, const float *const *pinit,
                 const mcxo_vlfunc *L, const float *incov)
{
  for (int s = 0; s < nshards; ++s) {
    if (eng[s]->size != nshards || eng[s]->rank != s) return -5;
    int st = run_begin(eng[s], nsamp, nburn, pinit[s], L, incov);
    if (st) return st;
  }
  for (int isamp = 0; isamp < nburn; ++isamp)
    for (int s = 0; s < nshards; ++s) burn_step(eng[s], isamp);
  for (int s = 0; s < nshards; ++s) burn_end(eng[s]);
  const size_t slot = 2 * (size_t)eng[0]->ntot;
  for (int isamp = 0; isamp < nsamp; ++isamp) {
    if (isamp % eng[0]->SYNCSTEP == 0 && nshards > 1)
      for (int s = 0; s < nshards; ++s)
        for (int r = 0; r < nshards; ++r)
          if (r != s)
            memcpy(eng[s]->musigall + slot * r, eng[r]->musigall + slot * r, sizeof(float) * slot);
    for (int s = 0; s < nshards; ++s) main_step(eng[s], isamp);
  }
  for (int s = 0; s < nshards; ++s) eng[s]->tbase += (uint32_t)(nburn + nsamp);
  return 0;
}

const float *mcxo_state(const mcxo_engine *e) { return e->pvals; }
const float *mcxo_loglike(const mcxo_engine *e) { return e->lylast; }
const float *mcxo_mean(const mcxo_engine *e) { return e->mu; }
const float *mcxo_var(const mcxo_engine *e) { return e->sig; }
const float *mcxo_musigall(const mcxo_engine *e) { return e->musigall; }
const float *mcxo_chol(const mcxo_engine *e) { return e->cov; }
const uint32_t *mcxo_accept_counts(const mcxo_engine *e) { return e->acounts; }
uint64_t mcxo_naccept_burn(const mcxo_engine *e) { return e->nacc_burn; }
uint64_t mcxo_naccept_main(const mcxo_engine *e) { return e->nacc_main; }
uint64_t mcxo_remote_steps(const mcxo_engine *e) { return e->nremote_steps; }
uint64_t mcxo_remote_passes(const mcxo_engine *e) { return e->nremote_passes; }
size_t mcxo_nsample_rows(const mcxo_engine *e) { return e->nrows; }
const float *mcxo_samples(const mcxo_engine *e) { return e->samples; }
const uint8_t *mcxo_accept_mask(const mcxo_engine *e) { return e->amask; }
int mcxo_tuner_trace(const mcxo_engine *e, float *scales, int maxn)
{
  int n = e->ntrace < maxn ? e->ntrace : maxn;
  for (int i = 0; i < n; ++i) scales[i] = e->trace[i];
  return e->ntrace;
}
