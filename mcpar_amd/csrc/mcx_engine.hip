// mcx_engine.hip -- host side of libmcx.so: the C ABI of include/mcx.h and the step-loop control
// code that replaces MCPar::run (src/mcpar.cc:17-214) by a schedule of gfx950 kernel launches.
//
// There is no CPU compute path in this file: every entry point that computes anything needs a
// HIP device and returns MCX_ERR_NO_DEVICE without one.
#include "mcx_engine_internal.hpp"


// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" const char *mcx_last_error(void) { return g_err.c_str(); }
extern "C" int mcx_abi_version(void) { return MCX_ABI_VERSION; }


int need_device()
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1)
    return fail(MCX_ERR_NO_DEVICE, "no HIP device visible (%s); libmcx has no CPU fallback",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  return MCX_OK;
}

extern "C" int mcx_set_device(int device)
{
  MCXCHK(need_device());
  HIPCHK(hipSetDevice(device));
  return MCX_OK;
}

extern "C" int mcx_device_info(char *name, size_t namelen, int *cu_count, size_t *hbm_bytes)
{
  MCXCHK(need_device());
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, dev));
  if (name && namelen) snprintf(name, namelen, "%s (%s)", p.name, p.gcnArchName);
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return MCX_OK;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
// Cholesky factor, lower, row-major, strict upper triangle zeroed: the role of spotrf('U') on the
// column-major view in MCPar::covar_setup (src/mcpar.cc:470-480).  Host side, np <= 32, once per run.
static int cholesky_lower(int d, float *a)
{
  for (int i = 0; i < d; ++i) {
    for (int j = 0; j <= i; ++j) {
      float s = a[i * d + j];
      for (int k = 0; k < j; ++k) s = std::fmaf(-a[i * d + k], a[j * d + k], s);
      if (i == j) {
        if (!(s > 0.0f)) return i + 1;
        a[i * d + i] = std::sqrt(s);
      } else {
        a[i * d + j] = s / a[j * d + j];
      }
    }
    for (int j = i + 1; j < d; ++j) a[i * d + j] = 0.0f;
  }
  return 0;
}

// uploads asynchronously on st; the caller synchronises before L.host is touched again
static int lik_setup(LikDev &L, const mcx_vlfunc *f, int np, hipStream_t st)
{
  if (!f) return fail(MCX_ERR_INVALID, "vlfunc is NULL");
  if (f->d != np) return fail(MCX_ERR_INVALID, "vlfunc.d = %d but engine np = %d", f->d, np);
  const int d = f->d;
  std::vector<float> h;
  L.fn = nullptr;
  L.ctx = nullptr;
  L.ncomp = 0;
  switch (f->kind) {
  case MCX_VL_ROSENBROCK1:
    if (d < 2 || (d & 1))  // src/rosenbrock.hh:13-16
      return fail(MCX_ERR_INVALID, "N for Rosenbrock1 must be even and >= 2");
    L.kind = LIK_ROSEN1;
    break;
  case MCX_VL_ROSENBROCK2:
    if (d < 2) return fail(MCX_ERR_INVALID, "N for Rosenbrock2 must be >= 2");  // src/rosenbrock.hh:27-30
    L.kind = LIK_ROSEN2;
    break;
  case MCX_VL_ROSENBROCK2_FIXED:
    if (d < 2) return fail(MCX_ERR_INVALID, "N for Rosenbrock2 must be >= 2");
    L.kind = LIK_ROSEN2F;
    break;
  case MCX_VL_GAUSSIAN:
    L.kind = LIK_GAUSS;
    h.resize(2 * (size_t)d);
    for (int k = 0; k < d; ++k) {  // src/rosenbrock.hh:44-47
      h[k] = f->params ? f->params[k] : 0.0f;
      h[d + k] = f->params ? 1.0f / f->params[d + k] : 1.0f;
    }
    break;
  case MCX_VL_DUALGAUSS: {
    if (d != 2) return fail(MCX_ERR_INVALID, "DualGaussian is two-dimensional");
    if (!f->params) return fail(MCX_ERR_INVALID, "DualGaussian needs params[0] = w");
    L.kind = LIK_MIX;
    L.ncomp = 2;
    const float m[4] = {0.0f, 0.0f, 5.0f, 5.0f};  // src/rosenbrock.cc:71-72
    h.assign(m, m + 4);
    h.push_back(logf_v1(f->params[0]));
    h.push_back(0.0f);
    break;
  }
  case MCX_VL_GAUSSMIX: {
    const int K = f->ncomp;
    if (K < 1 || K > 64 || !f->params) return fail(MCX_ERR_INVALID, "GAUSSMIX needs 1 <= K <= 64 and params");
    L.kind = LIK_MIX;
    L.ncomp = K;
    h.assign(f->params, f->params + (size_t)K * d);
    for (int c = 0; c < K; ++c) h.push_back(logf_v1(f->params[(size_t)K * d + c]));
    break;
  }
  case MCX_VL_HOST:
    if (!f->fn) return fail(MCX_ERR_VLFUNC, "MCX_VL_HOST without a callback");
    L.kind = MCX_VL_HOST;
    L.fn = f->fn;
    L.ctx = f->ctx;
    break;
  case MCX_VL_DEVICE:
    if (!f->ctx) return fail(MCX_ERR_VLFUNC, "MCX_VL_DEVICE without a kernel (hipFunction_t in ctx)");
    L.kind = MCX_VL_DEVICE;
    L.ctx = f->ctx;
    break;
  case MCX_VL_SOURCE: {
    if (f->ncomp < 0 || (f->ncomp > 0 && !f->params)) return fail(MCX_ERR_INVALID, "MCX_VL_SOURCE: ncomp floats of params expected");
    MCXCHK(user_lik_get(static_cast<const char *>(f->ctx), d, &L.user));  // (cached: compiled on first use)
    L.kind = LIK_USER;
    L.ncomp = f->ncomp;
    h.assign(f->params, f->params + (f->params ? f->ncomp : 0));
    if (h.empty()) h.push_back(0.0f);  // `par` is never a null pointer on the device
    break;
  }
  default:
    return fail(MCX_ERR_INVALID, "unknown vlfunc kind %d", f->kind);
  }
  if (L.params.p && h.size() == L.host.size() && (h.empty() || std::memcmp(h.data(), L.host.data(), h.size() * sizeof(float)) == 0))
    return MCX_OK;  // same parameters as the last call: they are on the device already
  HIPCHK(hipStreamSynchronize(st));  // (the last upload, or a run still in flight, may be reading L.host / L.params)
  L.host.swap(h);
  MCXCHK(L.params.alloc(L.host.size()));
  if (!L.host.empty())
    HIPCHK(hipMemcpyAsync(L.params.p, L.host.data(), L.host.size() * sizeof(float), hipMemcpyHostToDevice, st));
  return MCX_OK;
}

// The fused-kernel families are compiled in their own translation units (mcx_k_fast.hip,
// mcx_k_pregen.hip, mcx_k_generic_*.hip) so that the library builds in parallel; see mcx_launch.hpp.
static int launch_fused_plain(int lpc, int lik, bool main, const SegArgs &a, hipStream_t st, bool fast, int bpl = 1, int full_bpl = 0)
{
  hipError_t err;
  const bool fast_lik = lik == LIK_ROSEN1 || lik == LIK_GAUSS || (lik == LIK_MIX && a.ncomp <= 8);
  if (fast && fast_lik && bpl > 1 && bpl <= lpc) err = mcxk_launch_fastb(lpc, bpl, lik, main, a, st);  // hot path, several blocks per lane
  else if (fast) err = mcxk_launch_fast(lpc, lik, main, a, st);  // hot path
  else if (lpc <= 8 && fast_lik && !a.diag && a.vec4 && !a.mask) {
    // full covariance: one block per lane, or two mirrored ones (mcx_fastb.hpp) -- bpl as MCX_OPT_BLOCKS_PER_LANE says,
    // else what tools/fullcov_ab.sh measured best per size (16-D: since the generator got cheaper the mirrored kernel wins
    // at 65 536 chains, 2.49 against 2.55 ms; it has half the wavefronts, so not below that)
    const bool mirrored = (lpc == 4 || lpc == 8) && (full_bpl == 2 || (full_bpl == 0 && (lpc == 8 || a.n >= 65536)));
    err = mirrored ? mcxk_launch_fastb_full(lpc, lik, main, a, st) : mcxk_launch_fast_full(lpc, lik, main, a, st);
  }
  else err = main ? mcxk_launch_generic_main(lpc, lik, a, st) : mcxk_launch_generic_burn(lpc, lik, a, st);
  if (err == hipErrorInvalidValue) return fail(MCX_ERR_UNSUPPORTED, "no fused kernel for lanes/chain = %d, likelihood %d", lpc, lik);
  HIPCHK(err);
  return MCX_OK;
}

template <int LPC>
static int launch_eval(int lik, const float *x, float *y, int n, int d, const float *params,
                       int ncomp, int vec4, hipStream_t st)
{
  const dim3 grid(nblocks((size_t)n * LPC)), block(BLOCK);
  switch (lik) {
  case LIK_ROSEN1:
    hipLaunchKernelGGL((k_eval<LPC, LIK_ROSEN1>), grid, block, 0, st, x, y, n, d, params, ncomp, vec4);
    break;
  case LIK_GAUSS:
    hipLaunchKernelGGL((k_eval<LPC, LIK_GAUSS>), grid, block, 0, st, x, y, n, d, params, ncomp, vec4);
    break;
  case LIK_MIX:
    hipLaunchKernelGGL((k_eval<LPC, LIK_MIX>), grid, block, 0, st, x, y, n, d, params, ncomp, vec4);
    break;
  case LIK_ROSEN2F:
    hipLaunchKernelGGL((k_eval<LPC, LIK_ROSEN2F>), grid, block, 0, st, x, y, n, d, params, ncomp, vec4);
    break;
  case LIK_ROSEN2:
    hipLaunchKernelGGL(k_eval_rosen2, dim3(nblocks((size_t)n)), block, 0, st, x, y, n, d);
    break;
  default:
    return fail(MCX_ERR_INVALID, "likelihood %d has no device kernel", lik);
  }
  HIPCHK(hipGetLastError());
  return MCX_OK;
}

static int eval_device(const LikDev &L, const float *x, float *y, int n, int d, hipStream_t st)
{
  if (L.kind == LIK_USER) return user_lik_launch_eval(*L.user, x, y, n, d, L.params.p, L.ncomp, st);
  const int lpc = lpc_for(d), vec4 = (d % 4 == 0);
  DISPATCH_LPC(lpc, MCXCHK((launch_eval<LPC_>(L.kind, x, y, n, d, L.params.p, L.ncomp, vec4, st))));
  return MCX_OK;
}

// Launch one segment of consecutive local steps.  Small-n mode (MCX_OPT_SPLIT_RNG): with few chains the
// fused kernel is bound by the latency of a single wave's instruction stream, two thirds of it random
// numbers that do not depend on the chain state; they are then generated for 32..256 steps at a time
// by a fully parallel kernel on the otherwise idle SIMDs and streamed into the step kernel.
constexpr int SPLIT_CHUNK_MAX = 256;
constexpr size_t SPLIT_Z_BYTES = (size_t)32 << 20;  // keep a chunk's normals L2-resident (4 MiB per XCD)
constexpr size_t SPLIT_AUTO_MAX_WAVES = 640;

// Does a segment run on one of the hot-path kernels that take the burn-in tuner and the start of the moments into
// the launch (SegArgs::tun, SegArgs::init_moments: k_fused_fast plain / full covariance, k_fused_fastb)?  The same
// tests as launch_fused / launch_fused_plain below.
static bool fused_takes_epilogue(const mcx_engine *e, const SegArgs &a)
{
  const int lik = e->lik.kind, lpc = e->lpc;
  const bool fast_lik = lik == LIK_ROSEN1 || lik == LIK_GAUSS || (lik == LIK_MIX && a.ncomp <= 8);
  const bool fast = lpc <= 8 && (fast_lik || lik == LIK_ROSEN2F) && a.diag && a.vec4 && !a.mask;
  const size_t waves = ((size_t)a.n * lpc + 63) / 64;
  const bool split = fast && fast_lik && (e->opt_split > 0 || (e->opt_split < 0 && waves < SPLIT_AUTO_MAX_WAVES));
  if (split) return false;
  if ((unsigned long long)a.n * (unsigned long long)a.nsteps >= (1ull << 40)) return false;  // tuner_epilogue's 40-bit sums
  if (lik == LIK_USER) return user_lik_variant(lpc, a) < 2;  // the user's hot-path kernels are k_fused_fast's body + epilogue
  return fast || (lpc <= 8 && fast_lik && !a.diag && a.vec4 && !a.mask);
}

static int launch_fused(mcx_engine *e, bool main, const SegArgs &a, hipStream_t st)
{
  const int lik = e->lik.kind, lpc = e->lpc;
  if (lik == LIK_USER) return user_lik_launch_fused(*e->lik.user, main, a, st);  // MCX_VL_SOURCE: mcx_user.hip
  const bool fast_lik = lik == LIK_ROSEN1 || lik == LIK_GAUSS || (lik == LIK_MIX && a.ncomp <= 8);
  // (the overlapping Rosenbrock has the plain hot-path kernel only: no small-n modes, no several blocks per lane)
  const bool fast = lpc <= 8 && (fast_lik || lik == LIK_ROSEN2F) && a.diag && a.vec4 && !a.mask;
  const size_t waves = ((size_t)a.n * lpc + 63) / 64;
  const bool split = fast && fast_lik && (e->opt_split > 0 || (e->opt_split < 0 && waves < SPLIT_AUTO_MAX_WAVES));
  if (!split) {
    // blocks per lane of the hot-path kernel: MCX_OPT_BLOCKS_PER_LANE, or what was measured best (mcx_fastb.hpp)
    // Measured (tools/bpl_sweep.py, 65 536 chains): Rosenbrock1 / Gaussian 16-D 2.51 ms per job with one block per
    // lane, 2.67 with two, 2.81 with four -- the step is bound by the Philox / Box-Muller issue slots, which do not
    // care how the lanes are cut; the 32-D mixture 2.63 -> 1.97 ms with two -- its eight per-component reductions
    // over 8 lanes (DPP + row operations each) become reductions over 4.
    int bpl = e->opt_bpl;
    if (bpl == 0) bpl = (lik == LIK_MIX && lpc == 8) ? 2 : 1;
    while (bpl > lpc) bpl >>= 1;
    return launch_fused_plain(lpc, lik, main, a, st, fast, bpl, e->opt_bpl);
  }
  // generator and step kernel alternate on the engine's stream (overlapping them on two streams was
  // measured slower: the cross-stream event waits cost more than the generator, which is ~10 % of a chunk)
  const size_t per_step = (size_t)a.n * a.d * sizeof(float);
  const int SPLIT_CHUNK = (int)std::max<size_t>(32, std::min<size_t>(SPLIT_CHUNK_MAX, (SPLIT_Z_BYTES / per_step) & ~(size_t)7));
  constexpr int PAD = 16;  // the step kernel prefetches two 8-step batches ahead without bounds checks
  MCXCHK(e->zpre.alloc((size_t)(SPLIT_CHUNK + PAD) * a.n * a.d));
  MCXCHK(e->upre.alloc((size_t)(SPLIT_CHUNK + PAD) * a.n));
  MCXCHK(e->trash.alloc(4 * (size_t)a.n * lpc));
  for (int c0 = 0; c0 < a.nsteps; c0 += SPLIT_CHUNK) {
    const int ns = std::min(SPLIT_CHUNK, a.nsteps - c0);
    {
      ProfScope pg(e, MCX_K_GEN_NORMALS, (uint64_t)ns * (uint64_t)a.n);
      HIPCHK(mcxk_launch_gen(lpc, e->zpre.p, e->upre.p, a.n, a.d, ns, a.t0 + (uint32_t)c0, a.g0, a.seed, st));
    }
    SegArgs b = a;
    b.nsteps = ns;
    b.t0 = a.t0 + (uint32_t)c0;
    b.isamp0 = a.isamp0 + c0;
    b.snap_after = (a.snap_after >= c0 && a.snap_after < c0 + ns) ? a.snap_after - c0 : -1;
    if (b.samp_x && a.samp_stride <= 1) {
      b.samp_x += (size_t)c0 * a.n * a.d;
      b.samp_ly += (size_t)c0 * a.n;
    }
    b.zpre = e->zpre.p;
    b.upre = e->upre.p;
    b.trash = e->trash.p;
    HIPCHK(mcxk_launch_fast_pregen(lpc, lik, main, b, st));
    e->cnt.kernel_launches += 1;  // (the generator's scope counted itself)
  }
  return MCX_OK;
}

static void prof_collect(mcx_engine *e)
{
  for (auto &p : e->evs) {
    float ms = 0.0f;
    (void)hipEventSynchronize(p.b);
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      e->prof.ms[p.kind] += ms;
      e->prof.launches[p.kind] += 1;
      e->prof.chain_steps[p.kind] += p.chain_steps;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  e->evs.clear();
}

extern "C" int mcx_create(mcx_engine **out, int np, int nc, int nshards, int shard, float pl,
                          float armin, float armax, float dfac, float ifac, int sync, uint32_t seed)
{
  if (!out) return fail(MCX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (np < 1 || nc < 1 || nshards < 1 || shard < 0 || shard >= nshards || sync < 1)
    return fail(MCX_ERR_INVALID, "bad problem size np=%d nc=%d nshards=%d shard=%d sync=%d", np, nc,
                nshards, shard, sync);
  if (np > MAXD) return fail(MCX_ERR_UNSUPPORTED, "np = %d > %d is not supported", np, MAXD);
  if ((long long)nshards * nc > 0x7fffffffLL / (2LL * np))
    return fail(MCX_ERR_INVALID, "tchains*np*2 overflows int32 (the reference indexes musigall with int)");
  MCXCHK(need_device());
  mcx_engine *e = new mcx_engine();
  if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
  if (hipDeviceGetAttribute(&e->ncu, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) e->ncu = 0;
  e->nparam = np; e->nchain = nc; e->ntot = np * nc; e->ncov = np * np;
  e->size = nshards; e->rank = shard; e->tchains = nshards * nc;
  e->PLOCAL = pl; e->TGT_ARATE_MIN = armin; e->TGT_ARATE_MAX = armax;
  e->SCALE_DEC = dfac; e->SCALE_INC = ifac; e->SYNCSTEP = sync; e->seed = seed;
  e->lpc = lpc_for(np);
  e->vec4 = (np % 4 == 0);
  const size_t nt = (size_t)e->ntot, n = (size_t)nc;
  int st = MCX_OK;
  auto A = [&](int s) { if (st == MCX_OK) st = s; };
  A(e->pvals.alloc(nt)); A(e->ptrial.alloc(nt)); A(e->mu.alloc(nt)); A(e->sig.alloc(nt));
  A(e->psum2.alloc(nt)); A(e->mutrial.alloc(nt)); A(e->sigtrial.alloc(nt));
  A(e->musigall.alloc(2 * (size_t)e->tchains * np)); A(e->winvall.alloc(2 * (size_t)e->tchains * np));
  A(e->lylast.alloc(n)); A(e->lytrial.alloc(n)); A(e->cfac.alloc(n)); A(e->cmax.alloc(n));
  A(e->cov.alloc((size_t)e->ncov)); A(e->cov0.alloc((size_t)e->ncov)); A(e->trace.alloc(256)); A(e->acc_cnt.alloc(n));
  A(e->ctr.alloc((size_t)CTR_WORDS * CTR_RING + 2));  // (+ RunArgs::report_done, behind the ring)
  e->nslots = (int)(((size_t)nc * e->lpc + 63) / 64);
  A(e->acc_slots.alloc((size_t)e->nslots));
  A(e->tun_cells.alloc(TUN_CELLS + 1));
  if (st == MCX_OK && hipMemset(e->tun_cells.p, 0, (TUN_CELLS + 1) * sizeof(unsigned long long)) != hipSuccess)
    st = fail(MCX_ERR_HIP, "hipMemset failed");
  A(e->active0.alloc(n)); A(e->active1.alloc(n)); A(e->ntrace.alloc(1));
  A(e->nact.alloc(2 * (1 + 2 * NACT_CULL_CELLS) + 2 + 8));  // (+ the word k_remote_decide counts its workgroups in, + its tried[])
  if (st == MCX_OK && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    st = fail(MCX_ERR_HIP, "hipStreamCreate failed");
  if (st != MCX_OK) { mcx_destroy(e); return st; }
  e->own_stream = true;
  (void)hipMemsetAsync(e->musigall.p, 0, 2 * (size_t)e->tchains * np * sizeof(float), e->stream);
  (void)hipMemsetAsync(e->mu.p, 0, nt * sizeof(float), e->stream);
  (void)hipMemsetAsync(e->sig.p, 0, nt * sizeof(float), e->stream);
  (void)hipMemsetAsync(e->psum2.p, 0, nt * sizeof(float), e->stream);
  (void)hipMemsetAsync(e->acc_cnt.p, 0, n * sizeof(uint32_t), e->stream);
  (void)hipMemsetAsync(e->acc_slots.p, 0, (size_t)e->nslots * sizeof(uint32_t), e->stream);  // (every run leaves them zero)
  (void)hipMemsetAsync(e->ntrace.p, 0, sizeof(int), e->stream);
  // identity factor until covar_setup / run installs one (src/mcpar.cc:460-467)
  std::vector<float> eye((size_t)e->ncov, 0.0f);
  for (int i = 0; i < np; ++i) eye[(size_t)i * (np + 1)] = 1.0f;
  (void)hipMemcpyAsync(e->cov.p, eye.data(), eye.size() * sizeof(float), hipMemcpyHostToDevice, e->stream);
  (void)hipMemcpyAsync(e->cov0.p, eye.data(), eye.size() * sizeof(float), hipMemcpyHostToDevice, e->stream);
  e->h_cov_dev = eye;
  (void)hipMemsetAsync(e->ctr.p, 0, ((size_t)CTR_WORDS * CTR_RING + 2) * sizeof(unsigned long long), e->stream);
  if (hipStreamSynchronize(e->stream) != hipSuccess) {
    mcx_destroy(e);
    return fail(MCX_ERR_HIP, "engine initialisation failed");
  }
  *out = e;
  return MCX_OK;
}

extern "C" int mcx_destroy(mcx_engine *e)
{
  if (!e) return MCX_OK;
  (void)hipSetDevice(e->device);
  e->tail_publish = 0;  // (nobody will look at the slot; the gather itself is waited for by mcx_exchange_rccl_destroy)
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  e->pend.active = false;  // (a run nobody waited for: over now; its results go with the engine)
  if (e->meet_held) { (void)flock(e->meet_fd, LOCK_UN); e->meet_held = false; }
  if (e->astream) { (void)hipStreamSynchronize(e->astream); (void)hipStreamDestroy(e->astream); e->astream = nullptr; }
  for (hipEvent_t &ev : e->run_ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  for (hipEvent_t &ev : e->copy_ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  prof_collect(e);
  (void)mcx_exchange_rccl_destroy(e);
  e->pvals.release(); e->ptrial.release(); e->mu.release(); e->sig.release(); e->psum2.release();
  e->mutrial.release(); e->sigtrial.release(); e->musigall.release(); e->winvall.release();
  e->lylast.release(); e->lytrial.release(); e->cfac.release(); e->cmax.release(); e->cov.release(); e->cov0.release();
  e->trace.release(); e->acc_cnt.release(); e->acc_slots.release(); e->ctr.release(); e->active0.release();
  e->active1.release(); e->nact.release(); e->ntrace.release(); e->samp_x.release();
  e->cull_keys.release(); e->cull_hist.release(); e->cull_sorted.release(); e->cull_stats.release(); e->cull_box.release();
  e->cull_lim.release(); e->cull_excl.release(); e->scr_a.release(); e->scr_b.release(); e->scr_centre.release(); e->cand.release(); e->proj_acc.release(); e->proj_p.release(); e->proj_lohi.release(); e->tun_cells.release(); e->text_wg.release(); e->text_dev.release();
  e->samp_ly.release(); e->mask.release(); e->lik.params.release(); e->winv_tab.release(); e->psum.release(); e->pmax.release(); e->racpt.release(); e->pinit_dev.release();
  e->h_ptrial.release(); e->h_lytrial.release(); e->h_ctr.release(); e->h_nact.release(); e->zpre.release(); e->upre.release(); e->trash.release(); e->deal_tab.release(); e->trace_clk.release();
  for (int b = 0; b < 2; ++b) {
    e->sink_stage[b].release();
    e->sink_pin[b].release();
    e->sink_text_wg[b].release();
    e->sink_text_total[b].release();
    if (e->ev_steps[b]) (void)hipEventDestroy(e->ev_steps[b]);
    if (e->ev_copy[b]) (void)hipEventDestroy(e->ev_copy[b]);
    if (e->ev_write[b]) (void)hipEventDestroy(e->ev_write[b]);
    e->sink_text_dev[b].release();
  }
  e->best_row.release(); e->best_key.release(); e->cov0.release();
  for (auto &pr : e->xw_pool) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (e->meet_fd >= 0) (void)close(e->meet_fd);
  if (e->cstream) (void)hipStreamDestroy(e->cstream);
  if (e->mstream) (void)hipStreamDestroy(e->mstream);
  for (hipEvent_t ev : e->mev) (void)hipEventDestroy(ev);
  if (e->tstream) (void)hipStreamDestroy(e->tstream);
  if (e->ev_text) (void)hipEventDestroy(e->ev_text);
  e->sink_text_pin.release();
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return MCX_OK;
}


extern "C" int mcx_set_exchange(mcx_engine *e, mcx_exchange_fn fn, void *ctx)
{
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  if (e->pend.active) MCXCHK(enter(e));
  MCXCHK(finish_tail(e));  // (a gather of the exchange being replaced may still be in flight)
  e->xfn = fn;
  e->xctx = ctx;
  return MCX_OK;
}

extern "C" int mcx_set_output_hook(mcx_engine *e, mcx_output_fn fn, void *ctx)
{
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  e->ofn = fn;
  e->octx = ctx;
  return MCX_OK;
}

extern "C" int mcx_set_option(mcx_engine *e, int opt, int64_t value)
{
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  if (e->pend.active) MCXCHK(enter(e));  // (options apply to whole runs: the one in flight is finished first)
  switch (opt) {
  case MCX_OPT_SAMPLES: e->opt_samples = value ? 1 : 0; break;
  case MCX_OPT_SAMPLE_STRIDE:
    if (value < 1) return fail(MCX_ERR_INVALID, "SAMPLE_STRIDE must be >= 1");
    e->opt_stride = (int)std::min<int64_t>(value, 1 << 30);
    break;
  case MCX_OPT_ACCEPT_MASK: e->opt_mask = value ? 1 : 0; break;
  case MCX_OPT_FUSE: e->opt_fuse = value ? 1 : 0; break;
  case MCX_OPT_MAX_SEGMENT:
    if (value < 1) return fail(MCX_ERR_INVALID, "MAX_SEGMENT must be >= 1");
    e->opt_maxseg = (int)std::min<int64_t>(value, 1 << 20);
    break;
  case MCX_OPT_PROFILE: e->opt_profile = value ? 1 : 0; break;
  case MCX_OPT_EAGER_EXCHANGE: e->opt_eager = value ? 1 : 0; break;
  case MCX_OPT_SINK_TEXT: e->opt_sink_text = value ? 1 : 0; break;
  case MCX_OPT_ASYNC_TAIL: e->opt_async_tail = value == 2 ? 2 : (value ? 1 : 0); break;
  case MCX_OPT_SPLIT_RNG: e->opt_split = value < 0 ? -1 : (value ? 1 : 0); break;
  case MCX_OPT_PERSIST: e->opt_persist = value < 0 ? -1 : (value ? 1 : 0); break;
  case MCX_OPT_CULL: e->opt_cull = value < 0 ? -1 : ((value == 2 || value == 3) ? value : (value ? 1 : 0)); break;
  case MCX_OPT_BLOCKS_PER_LANE:
    if (value != 0 && value != 1 && value != 2 && value != 4) return fail(MCX_ERR_INVALID, "BLOCKS_PER_LANE must be 0 (auto), 1, 2 or 4");
    e->opt_bpl = (int)value;
    break;
  case MCX_OPT_MEET_TIMEOUT_MS:
    if (value < 1) return fail(MCX_ERR_INVALID, "MEET_TIMEOUT_MS must be >= 1");
    e->opt_meet_timeout_ms = (int)std::min<int64_t>(value, 600000);
    break;
  case MCX_OPT_ASYNC_RUN: e->opt_async_run = value ? 1 : 0; break;
  case MCX_OPT_REFERENCE_CALLS: e->opt_reference_calls = value ? 1 : 0; break;
  case MCX_OPT_SELF_REPORT: e->opt_self_report = value ? 1 : 0; break;
  case MCX_OPT_MURRAY_OVERLAP:
    if (value < 0 || value > 64) return fail(MCX_ERR_INVALID, "MURRAY_OVERLAP: 0 (off) or the number of column chunks, <= 64");
    e->opt_murray_overlap = (int)value;
    break;
  case MCX_OPT_MEET_UNDER_GATHER: e->opt_meet_under_gather = value < 0 ? -1 : (value ? 1 : 0); break;
  case MCX_OPT_DEBUG_MEET:
    e->opt_debug_meet = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
    e->persist_broken = false;  // (a test switching the hook off again gets the one-launch kernel back)
    break;
  case MCX_OPT_STREAM:
    if (e->own_stream && e->stream) {
      (void)hipStreamSynchronize(e->stream);
      (void)hipStreamDestroy(e->stream);
    }
    e->stream = (hipStream_t)(uintptr_t)value;
    e->own_stream = false;
    break;
  default: return fail(MCX_ERR_INVALID, "unknown option %d", opt);
  }
  return MCX_OK;
}

// MCPar::covar_setup (src/mcpar.cc:454-484)
static int covar_install(mcx_engine *e, const float *incov, float *cov_out, bool sync = true)
{
  const int d = e->nparam;
  std::vector<float> &c = e->h_cov;
  c.assign((size_t)e->ncov, 0.0f);
  if (incov) std::copy(incov, incov + e->ncov, c.begin());
  else
    for (int i = 0; i < d; ++i) c[(size_t)i * (d + 1)] = 1.0f;
  if (cholesky_lower(d, c.data()) != 0)
    return fail(MCX_ERR_INVALID, "covariance matrix is not positive definite");
  e->diag = true;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < i; ++j)
      if (c[(size_t)i * d + j] != 0.0f) e->diag = false;
  // cov0 keeps the factor as installed; cov (which the tuner rescales) is reset from it, on the device
  if (e->h_cov_dev != c) {
    if (!e->h_cov_dev.empty()) HIPCHK(hipStreamSynchronize(e->stream));  // the last upload may still read h_cov_dev
    e->h_cov_dev = c;
    HIPCHK(hipMemcpyAsync(e->cov0.p, e->h_cov_dev.data(), c.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
  }
  if (sync) {
    HIPCHK(hipMemcpyAsync(e->cov.p, e->cov0.p, c.size() * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->cov_pending = false;
    e->cov_offdiag = !e->diag;
  } else {
    e->cov_pending = true;  // the run resets it: k_run_small reads cov0 itself, every other path copies first
  }
  if (cov_out) std::copy(c.begin(), c.end(), cov_out);
  return MCX_OK;
}

static int cov_reset(mcx_engine *e)
{
  if (!e->cov_pending) return MCX_OK;
  HIPCHK(hipMemcpyAsync(e->cov.p, e->cov0.p, (size_t)e->ncov * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  e->cov_pending = false;
  e->cov_offdiag = !e->diag;
  return MCX_OK;
}

extern "C" int mcx_covar_setup(mcx_engine *e, const float *incov, float *cov)
{
  MCXCHK(enter(e));
  if (!e) return fail(MCX_ERR_INVALID, "engine is NULL");
  return covar_install(e, incov, cov);
}

// VLFunc call on device-resident proposals; HOST kind goes device -> host -> device
static int eval_trials(mcx_engine *e, const float *x_dev, float *y_dev, uint64_t cs)
{
  const int n = e->nchain, d = e->nparam;
  if (e->lik.kind == MCX_VL_HOST) {
    MCXCHK(e->h_ptrial.alloc((size_t)e->ntot));
    MCXCHK(e->h_lytrial.alloc((size_t)n));
    HIPCHK(hipMemcpyAsync(e->h_ptrial.p, x_dev, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    (void)e->lik.fn(e->lik.ctx, n, e->h_ptrial.p, e->h_lytrial.p);  // return code ignored like the reference
    // no second synchronisation: h_lytrial is pinned and is next written by the callback of the NEXT step, which
    // runs only after that step's D2H -- queued behind this copy on the same stream -- has been waited for
    HIPCHK(hipMemcpyAsync(y_dev, e->h_lytrial.p, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return MCX_OK;
  }
  ProfScope ps(e, MCX_K_EVAL, cs);
  if (e->lik.kind == MCX_VL_DEVICE) {  // the user's own kernel, VLFunc contract on device memory
    int npset = n;
    const float *xa = x_dev;
    float *ya = y_dev;
    void *args[] = {&npset, &xa, &ya};
    HIPCHK(hipModuleLaunchKernel((hipFunction_t)e->lik.ctx, nblocks((size_t)n), 1, 1, BLOCK, 1, 1, 0, e->stream, args, nullptr));
    return MCX_OK;
  }
  return eval_device(e->lik, x_dev, y_dev, n, d, e->stream);
}

// MCX_OPT_REFERENCE_CALLS: the reference evaluates L(1, pvals_j, &y) once per chain after every main-loop step and throws
// the result away (src/mcpar.cc:177-182).  A host functor with side effects -- a call counter, a cache, a log -- sees those
// calls there; here they are made only on request, for host functors (a device likelihood has no side effects to keep).
static int discarded_calls(mcx_engine *e)
{
  if (!e->opt_reference_calls || e->lik.kind != MCX_VL_HOST) return MCX_OK;
  const int n = e->nchain, d = e->nparam;
  MCXCHK(e->h_ptrial.alloc((size_t)e->ntot));
  HIPCHK(hipMemcpyAsync(e->h_ptrial.p, e->pvals.p, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  float y = 0.0f;
  for (int j = 0; j < n; ++j) (void)e->lik.fn(e->lik.ctx, 1, e->h_ptrial.p + (size_t)j * d, &y);
  return MCX_OK;
}

// Base pointers such that kept step r = isamp / stride of the run lives at base + r * rowsize: the whole-run
// store itself, or -- in sink mode -- the ring slot of isamp's block shifted back by the block's first row
// (kernels index rows of the run; a launch never straddles a block).
void samp_vbase(const mcx_engine *e, int isamp, float **px, float **pl)
{
  if (!e->run_sink) { *px = e->samp_x.p; *pl = e->samp_ly.p; return; }
  const long long b = isamp / e->run_sblock, slot = b % SINK_RING, shift = (slot - b) * (long long)e->run_kb;
  *px = e->samp_x.p + shift * (long long)e->ntot;
  *pl = e->samp_ly.p + shift * (long long)e->nchain;
}

static void fill_step(mcx_engine *e, StepArgs &a, uint32_t t, int isamp, bool main, size_t maskrow,
                      int samprow, int remote)
{
  a.x = e->pvals.p; a.ly = e->lylast.p; a.mu = e->mu.p; a.psum2 = e->psum2.p;
  a.ptrial = e->ptrial.p; a.lytrial = e->lytrial.p; a.cfac = e->cfac.p;
  a.mutrial = e->mutrial.p; a.sigtrial = e->sigtrial.p;
  a.acc_cnt = e->acc_cnt.p;
  a.acc_slots = e->acc_slots.p;
  a.T = e->cov.p;
  const bool keep = main && e->opt_samples && samprow % e->opt_stride == 0;
  float *vx = nullptr, *vl = nullptr;
  if (keep) samp_vbase(e, samprow, &vx, &vl);
  a.samp_x = keep ? vx + (size_t)(samprow / e->opt_stride) * e->ntot : nullptr;
  a.samp_ly = keep ? vl + (size_t)(samprow / e->opt_stride) * e->nchain : nullptr;
  a.mask = e->opt_mask ? e->mask.p + maskrow * (size_t)e->nchain : nullptr;
  a.lik = e->lik.params.p; a.ncomp = e->lik.ncomp;
  a.n = e->nchain; a.d = e->nparam;
  a.g0 = (uint32_t)(e->rank * e->nchain); a.t = t; a.seed = e->seed;
  a.isamp = isamp; a.diag = e->diag ? 1 : 0; a.vec4 = e->vec4; a.remote = remote;
}

static int launch_propose(mcx_engine *e, const StepArgs &a)
{
  ProfScope ps(e, MCX_K_PROPOSE, (uint64_t)e->nchain);
  const dim3 grid(nblocks((size_t)a.n * e->lpc)), block(BLOCK);
  DISPATCH_LPC(e->lpc, hipLaunchKernelGGL((k_propose_local<LPC_>), grid, block, 0, e->stream, a));
  HIPCHK(hipGetLastError());
  return MCX_OK;
}

static int launch_accept(mcx_engine *e, const StepArgs &a, bool main)
{
  ProfScope ps(e, MCX_K_ACCEPT, (uint64_t)e->nchain);
  const dim3 grid(nblocks((size_t)a.n * e->lpc)), block(BLOCK);
  if (main) { DISPATCH_LPC(e->lpc, hipLaunchKernelGGL((k_accept<LPC_, true>), grid, block, 0, e->stream, a)); }
  else { DISPATCH_LPC(e->lpc, hipLaunchKernelGGL((k_accept<LPC_, false>), grid, block, 0, e->stream, a)); }
  HIPCHK(hipGetLastError());
  return MCX_OK;
}

// the one-launch small-n kernel's generator deal (mcxk_persist_deal) and steps per phase, for tests: host logic only
extern "C" int mcx_debug_persist_deal(int lpc2, int bpl, int own, int *rec, int *ksteps, uint32_t *tab, int max_words)
{
  if (lpc2 < 1 || lpc2 > 8 || (bpl != 1 && bpl != 2 && bpl != 4) || own < 1 || own > POWN_MAX || !rec || !ksteps || !tab ||
      max_words < MCXK_PERSIST_DEAL_WORDS)
    return fail(MCX_ERR_INVALID, "bad arguments");
  *rec = mcxk_persist_recorders(own, bpl) ? 1 : 0;
  *ksteps = mcxk_persist_ksteps(lpc2, bpl, own);
  mcxk_persist_deal(lpc2, bpl, own, *rec, *ksteps, tab);
  return MCX_OK;
}

extern "C" int mcx_device_count(int *n)
{
  if (!n) return fail(MCX_ERR_INVALID, "n is NULL");
  *n = 0;
  MCXCHK(need_device());
  HIPCHK(hipGetDeviceCount(n));
  return MCX_OK;
}

extern "C" int mcx_device_pci_bus_id(char *buf, size_t len)
{
  if (!buf || len < 16) return fail(MCX_ERR_INVALID, "buffer too small");
  MCXCHK(need_device());
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetPCIBusId(buf, (int)len, dev));
  return MCX_OK;
}

// ---------------------------------------------------------------------------------------------
// k_run_small's tuner events are meetings of ALL its workgroups at a device counter: every workgroup of the
// grid must be resident.  The grid is sized to fit the GPU on its own (one workgroup per CU), but two such
// kernels dispatched at the same time -- two engines of one process, or two processes sharing a GPU -- could
// each get part of the CUs and wait for the rest forever.  So a launch that contains meetings (burn-in steps)
// holds an exclusive advisory lock on a per-GPU lock file until it has completed; launches without meetings
// (main-loop steps only: workgroups are independent) need none.  flock() excludes both other processes and
// other engines of this process (each engine has its own open file description).
// ---------------------------------------------------------------------------------------------
// internal status of run_once(): a tuner meeting of k_run_small was abandoned (or its grid cannot be resident):
// mcx_run repeats the run on the per-segment kernels.  Never leaves this file.


// the launch that took the lock has completed (or is waited for here): let the next one in, and look at the
// launch's "abandoned" word -- before any of its results is used or shown to a hook
int meet_release(mcx_engine *e, bool stream_is_idle)
{
  if (!e->meet_held && !e->meet_check) return MCX_OK;
  hipError_t se = hipSuccess;
  if (!stream_is_idle) se = hipStreamSynchronize(e->stream);
  if (e->meet_held) {
    (void)flock(e->meet_fd, LOCK_UN);
    e->meet_held = false;
  }
  HIPCHK(se);
  if (e->meet_check) {
    e->meet_check = false;
    unsigned long long w = 0;
    HIPCHK(hipMemcpyAsync(&w, e->meet_word, sizeof w, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (w) return MCX_INTERNAL_MEET_ABANDONED;
  }
  return MCX_OK;
}

// One fixed path per GPU (PCI bus id), the same for every process and user whatever their TMPDIR: /dev/shm
// first (always local, never a per-job directory), /tmp second.  O_NOFOLLOW: a symbolic link planted under the
// name is not followed; the mode is widened only on the file this call created.
static bool meet_lock_open(mcx_engine *e)
{
  if (e->meet_fd >= 0) return true;
  char bus[64] = "gpu";
  (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, e->device);
  for (char *c = bus; *c; ++c)
    if (*c == ':' || *c == '/') *c = '_';
  const char *dirs[] = {"/dev/shm", "/tmp"};
  for (const char *d : dirs) {
    const std::string path = std::string(d) + "/mcx_meet_" + bus + ".lock";
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0666);
    if (fd >= 0) (void)fchmod(fd, 0666);  // created here: shared by every user of the GPU
    else fd = open(path.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);  // (another user's file: flock needs no write access)
    if (fd >= 0) {
      e->meet_fd = fd;
      return true;
    }
  }
  return false;
}

static int meet_lock_take(mcx_engine *e)
{
  int rc;
  do rc = flock(e->meet_fd, LOCK_EX);
  while (rc != 0 && errno == EINTR);
  if (rc != 0) return fail(MCX_ERR_HIP, "cannot lock the GPU's meeting lock file: %s", strerror(errno));
  e->meet_held = true;
  return MCX_OK;
}

// rerun: repeat the run that was just abandoned -- same likelihood and factor as installed (L and incov are not looked at),
// from the staged state or, when the run started from caller memory that may be gone by now, from the copy kept of it
static int run_once(mcx_engine *e, int nsamp, int nburn, const float *pinit, const mcx_vlfunc *L, const float *incov, bool rerun = false);
constexpr int PERSIST_RETRY_RUNS = 16;

// A tuner meeting of the one-launch small-n kernel was abandoned: some workgroup of its grid was not resident (CU mask,
// partitioned device, a foreign kernel on the CUs).  The launch wrote no state back: the run is repeated on the
// per-segment kernels (same bits), which the engine then keeps to for PERSIST_RETRY_RUNS runs.
static int repeat_abandoned_run(mcx_engine *e, int nsamp, int nburn, const float *pinit, const mcx_vlfunc *L, const float *incov, bool rerun)
{
  (void)hipStreamSynchronize(e->stream);
  e->persist_broken = true;
  e->runs_since_broken = 0;
  e->meet_total++;
  if (getenv("MCX_VERBOSE"))
    fprintf(stderr, "mcx: a tuner meeting of the one-launch small-n kernel was abandoned after %d ms (a workgroup of its grid "
                    "was not resident); the run is repeated on the per-segment kernels\n", e->opt_meet_timeout_ms);
  const int async = e->opt_async_run;
  e->opt_async_run = 0;  // (the repeat is waited for: whoever asked is about to look at its results)
  int rc = run_once(e, nsamp, nburn, pinit, L, incov, rerun);
  e->opt_async_run = async;
  if (rc == MCX_INTERNAL_MEET_ABANDONED) rc = fail(MCX_ERR_HIP, "internal: meeting abandoned without the one-launch kernel");
  e->cnt.meet_timeouts = 1;
  return rc;
}

static void never_leave_the_lock_behind(mcx_engine *e, int rc)
{
  if (rc == MCX_OK) return;
  if (e->meet_held) {
    (void)hipStreamSynchronize(e->stream);
    (void)flock(e->meet_fd, LOCK_UN);
    e->meet_held = false;
  }
  e->meet_check = false;
}

// A run whose last launch reports to its counter slot itself (RunArgs::report): has the serial number arrived?  Spins for
// at most `spin_us` -- jobs this is about take 0.3-0.5 ms; whoever waits for a longer one loses nothing by sleeping in
// hipStreamSynchronize instead -- and says whether it saw it.
static bool report_arrived(const unsigned long long *slot, unsigned long long serial, int spin_us)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0;; ++it) {
    if (__atomic_load_n(slot + 7, __ATOMIC_ACQUIRE) == serial) return true;
    if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) return false;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
}
constexpr int REPORT_SPIN_US = 1500;

// the books of runs that were queued asynchronously and never looked at (the next run was queued behind them): was one of
// their meetings abandoned?  Nobody saw their results -- nothing to repeat -- but the engine keeps to the per-segment kernels.
// all = every such run's counters have arrived (the caller waited for a later copy on the same stream); otherwise only those
// whose copy is over are looked at, the others next time
static void note_superseded(mcx_engine *e, bool all)
{
  for (int sl = 0; sl < mcx_engine::HSLOTS && e->superseded_mask; ++sl) {
    if (!(e->superseded_mask & (1u << sl))) continue;
    if (!all && e->slot_serial[sl]) {  // (it reports itself)
      if (__atomic_load_n(e->h_ctr.p + 8 * sl + 7, __ATOMIC_ACQUIRE) != e->slot_serial[sl]) continue;
    } else if (!all && e->copy_pending[sl]) {
      if (hipEventQuery(e->copy_ev[sl]) != hipSuccess) { (void)hipGetLastError(); continue; }
      e->copy_pending[sl] = false;
    }
    if (e->h_ctr.p[8 * sl + 5] != 0) {
      e->persist_broken = true;
      e->runs_since_broken = 0;
      e->meet_total++;
    }
    e->superseded_mask &= ~(1u << sl);
  }
}

// MCX_OPT_ASYNC_RUN: the end of the run that mcx_run queued and returned from.  Called by every entry point but mcx_run
// (enter(), mcx_engine_internal.hpp).
int finish_pending(mcx_engine *e)
{
  if (!e->pend.active) return MCX_OK;
  const mcx_engine::PendingRun p = e->pend;
  e->pend.active = false;
  hipError_t se = hipSuccess;
  if (!(p.serial && report_arrived(p.hctr, p.serial, REPORT_SPIN_US))) se = hipStreamSynchronize(e->stream);
  if (se == hipSuccess && p.serial && __atomic_load_n(p.hctr + 7, __ATOMIC_ACQUIRE) != p.serial)
    return fail(MCX_ERR_HIP, "internal: the run's last launch is over and has not reported");
  if (se == hipSuccess && e->copy_pending[p.slot]) se = hipEventSynchronize(e->copy_ev[p.slot]);  // (its counters: a stream of their own)
  if (se == hipSuccess)  // (the copies leave in order: every earlier run's counters are in as well)
    for (bool &cp : e->copy_pending) cp = false;
  for (bool &rq : e->run_queued) rq = false;
  const bool abandoned = p.meet_check && se == hipSuccess && p.hctr[5] != 0;
  e->meet_check = false;
  (void)meet_release(e, true);
  if (se == hipSuccess) note_superseded(e, true);
  HIPCHK(se);
  int rc = MCX_OK;
  if (abandoned) {
    e->tbase = p.tbase0;  // (the abandoned launch moved nothing but the step counter)
    rc = repeat_abandoned_run(e, p.nsamp, p.nburn, nullptr, nullptr, nullptr, true);
  } else {
    e->cnt.naccept_burn = p.hctr[3];
    e->cnt.naccept_main = p.hctr[4];
    xwait_collect(e);
    if (e->persist_broken && ++e->runs_since_broken >= PERSIST_RETRY_RUNS) e->persist_broken = false;
  }
  never_leave_the_lock_behind(e, rc);
  return rc;
}

extern "C" int mcx_run(mcx_engine *e, int nsamp, int nburn, const float *pinit, const mcx_vlfunc *L,
                       const float *incov)
{
  MCXCHK(enter_raw(e));
  if (e->pend.active) {
    // A run is still in flight.  With MCX_OPT_ASYNC_RUN the next one is queued right behind it: nobody has looked at its
    // results, and nobody will -- this run overwrites them -- so it needs no waiting for (launch and completion latency
    // of back-to-back small jobs overlap the jobs themselves); its counters are looked at later, for the books only.
    if (e->opt_async_run && hipStreamQuery(e->stream) == hipErrorNotReady) {
      (void)hipGetLastError();
      // at most TWO runs in flight: the one before the pending one must be over before this call queues another
      // (the host queues a small job in 15 us, the GPU takes 400: without a bound the queue would only grow).  Its KERNELS,
      // not its counters: their copy -- a kernel of the runtime's on the other stream -- gets no room beside the pending
      // run's grid and ends with it; waiting for it let the queue run dry after every second job (24 us between two jobs
      // where 7 is the dispatch alone, tools/queued_jobs_probe.py)
      const int before = (e->hctr_slot + mcx_engine::HSLOTS - 1) % mcx_engine::HSLOTS;
      if (e->run_queued[before]) {
        if (e->slot_serial[before]) {  // (it reports itself: no event behind it)
          if (!report_arrived(e->h_ctr.p + 8 * before, e->slot_serial[before], REPORT_SPIN_US)) {
            HIPCHK(hipStreamSynchronize(e->stream));  // a long job: no hurry then
            (void)hipGetLastError();
          }
        } else {
          HIPCHK(hipEventSynchronize(e->run_ev[before]));
        }
        e->run_queued[before] = false;
      }
      note_superseded(e, false);
      if (e->pend.meet_check) e->superseded_mask |= 1u << e->pend.slot;
      e->pend.active = false;
    } else {
      (void)hipGetLastError();
      MCXCHK(finish_pending(e));
    }
  }
  static const int verbose = getenv("MCX_VERBOSE") ? atoi(getenv("MCX_VERBOSE")) : 0;
  const auto ht0 = std::chrono::steady_clock::now();
  e->ht_mark[0] = e->ht_mark[1] = e->ht_mark[2] = ht0;
  int rc = run_once(e, nsamp, nburn, pinit, L, incov);
  if (verbose >= 2) {  // where the host's share of a run goes: set-up / queued everything / stream idle / done
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto ht3 = std::chrono::steady_clock::now();
    fprintf(stderr, "mcx: run host timing: to first launch %.1f us, queueing %.1f us, waiting for the stream %.1f us, after %.1f us\n",
            us(ht0, e->ht_mark[0]), us(e->ht_mark[0], e->ht_mark[1]), us(e->ht_mark[1], e->ht_mark[2]), us(e->ht_mark[2], ht3));
  }
  uint64_t repeated = 0;
  if (rc == MCX_INTERNAL_MEET_ABANDONED) {
    repeated = 1;
    rc = repeat_abandoned_run(e, nsamp, nburn, pinit, L, incov, false);
  }
  if (!e->pend.active) {
    e->cnt.meet_timeouts = repeated;
    // whatever kept a workgroup out may be gone: the one-launch kernel is tried again after PERSIST_RETRY_RUNS runs
    if (e->persist_broken && !repeated && rc == MCX_OK && ++e->runs_since_broken >= PERSIST_RETRY_RUNS) e->persist_broken = false;
  }
  never_leave_the_lock_behind(e, rc);
  return rc;
}

static int run_once(mcx_engine *e, int nsamp, int nburn, const float *pinit, const mcx_vlfunc *L, const float *incov, bool rerun)
{
  if (nsamp < 0 || nburn < 0) return fail(MCX_ERR_INVALID, "bad run arguments");
  // the state a run without `pinit` starts from: what mcx_stage_pinit put there, or (the repeat of an asynchronous run
  // that started from caller memory) the copy kept of that memory
  const float *const staged = (rerun && e->pend.host_pinit) ? e->pinit_async.p : e->pinit_dev.p;
  if (rerun) pinit = nullptr;
  if (!pinit && !rerun && !e->pinit_staged) return fail(MCX_ERR_INVALID, "pinit is NULL and no state was staged (mcx_stage_pinit)");
  if (e->size > 1 && !e->xfn) return fail(MCX_ERR_EXCHANGE, "nshards > 1 needs mcx_set_exchange()");
  const int n = e->nchain, d = e->nparam;
  hipStream_t st = e->stream;
  if (!rerun) {
    MCXCHK(lik_setup(e->lik, L, d, st));
    MCXCHK(covar_install(e, incov, nullptr, false));  // src/mcpar.cc:20
  } else {
    e->cov_pending = true;  // the factor as installed (cov0), rescaled by nothing yet
  }
  e->cull_skip[0] = e->cull_skip[1] = 0;  // a new job starts from pinit: what the last one's Murray sweeps found useless is no guide
  // sample store: every chain, every main-loop step (src/mcpar.cc:31-40, 177-182), kept in HBM
  e->samp_steps = 0;
  const int nkeep = (nsamp + e->opt_stride - 1) / e->opt_stride;  // kept steps: isamp % stride == 0
  // sink mode: a ring of SINK_RING blocks instead of the whole run (block length a multiple of the stride)
  const bool sink = (e->sfn != nullptr || e->tfn != nullptr) && e->opt_samples && nsamp > 0;
  const int sblock = sink ? ((std::max(e->sink_block, 1) + e->opt_stride - 1) / e->opt_stride) * e->opt_stride : 0;
  const int kb = sink ? sblock / e->opt_stride : 0;
  e->run_sink = sink; e->run_sblock = sblock; e->run_kb = kb;
  e->run_sink_text = sink && e->sfn != nullptr && e->opt_sink_text != 0;
  if (e->opt_samples && nsamp > 0) {
    const size_t rows = sink ? (size_t)std::min<long long>((long long)SINK_RING * kb, nkeep + kb) : (size_t)nkeep;
    int s1 = e->samp_x.alloc(rows * e->ntot), s2 = e->samp_ly.alloc(rows * n);
    if (s1 != MCX_OK || s2 != MCX_OK)
      return fail(MCX_ERR_ALLOC, "Unable to allocate space for output samples (%zu bytes)",
                  rows * n * (d + 1) * sizeof(float));
  }
  if (sink) {
    for (int b = 0; b < 2; ++b) {
      MCXCHK(e->sink_stage[b].alloc((size_t)kb * n * (d + 1)));
      if (e->tfn || e->opt_sink_text) {
        MCXCHK(e->sink_text_wg[b].alloc(((size_t)kb * n * (d + 1) + BLOCK - 1) / BLOCK + 1));
        MCXCHK(e->sink_text_total[b].alloc(1));
      }
      if (!e->tfn) MCXCHK(e->sink_pin[b].alloc((size_t)kb * n * (d + 1)));
      if (!e->ev_steps[b]) HIPCHK(hipEventCreateWithFlags(&e->ev_steps[b], hipEventDisableTiming));
      if (!e->ev_copy[b]) HIPCHK(hipEventCreateWithFlags(&e->ev_copy[b], hipEventDisableTiming));
      if (!e->ev_write[b]) HIPCHK(hipEventCreateWithFlags(&e->ev_write[b], hipEventDisableTiming));
    }
    if (!e->cstream) HIPCHK(hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking));
    if (!e->tstream) HIPCHK(hipStreamCreateWithFlags(&e->tstream, hipStreamNonBlocking));
    if (!e->ev_text) HIPCHK(hipEventCreateWithFlags(&e->ev_text, hipEventDisableTiming));
  }
  MCXCHK(e->best_row.alloc((size_t)d + 1));
  MCXCHK(e->best_key.alloc(1));
  if (sink) {
    hipLaunchKernelGGL(k_best_reset, dim3(nblocks((size_t)d + 1)), dim3(BLOCK), 0, st, e->best_row.p, d, e->best_key.p);
    HIPCHK(hipGetLastError());
  }
  if (e->opt_mask) {
    MCXCHK(e->mask.alloc((size_t)(nburn + nsamp) * n));
    HIPCHK(hipMemsetAsync(e->mask.p, 0, (size_t)(nburn + nsamp) * n, st));
  }
  if ((size_t)nsamp > e->h_winv.size()) {  // 1/pwgt for every main-loop step (src/mcpar.cc:186-187),
    HIPCHK(hipStreamSynchronize(st));      // correctly rounded on the host; rebuilt only when it grows
    e->h_winv.resize((size_t)nsamp);
    for (int i = 0; i < nsamp; ++i) e->h_winv[(size_t)i] = 1.0f / (float)(i + 1);
    MCXCHK(e->winv_tab.alloc((size_t)nsamp + 16));  // + the small-n kernel's unchecked prefetch distance
    HIPCHK(hipMemcpyAsync(e->winv_tab.p, e->h_winv.data(), e->h_winv.size() * sizeof(float), hipMemcpyHostToDevice, st));
  }
  e->cnt = mcx_counters{};
  if (e->tail_publish) {  // the last run's final gather may still be in flight (finish_tail)
    if (nsamp > 0) e->tail_publish = 0;  // this run rewrites the slot -- after waiting for that gather -- before anything reads it
    else MCXCHK(finish_tail(e));
  }
  e->published_steps = 0;
  const bool fused = e->opt_fuse && e->lik.fusable();
  const uint32_t g0 = (uint32_t)(e->rank * n);
  // Small-n mode, one launch per stretch of local steps (mcx_persist.hpp): the whole burn-in with its tuner
  // events, the start of the main loop and every run of consecutive local main-loop segments go to k_run_small
  // when the chains fill at most POWN_MAX wavefronts per CU and the hot-path kernel applies.
  // (with two 4-parameter blocks per lane where that takes a workgroup from two or more owner wavefronts towards one)
  const int pbpl = mcxk_persist_bpl(e->lpc, d, n, e->ncu, e->opt_bpl), plpc2 = e->lpc / pbpl;
  const int nown = (int)(((size_t)n * plpc2 + 63) / 64);
  // (the mode is for chains that fill at most POWN_MAX wavefronts per CU at ONE block per lane: beyond that the per-segment
  // kernels are as fast or faster -- 65 536 x 16-D would fit the grid with two blocks per lane and run 35 % slower)
  const size_t nown_one_block = ((size_t)n * e->lpc + 63) / 64;
  const bool fast_lik = e->lik.kind == LIK_ROSEN1 || e->lik.kind == LIK_GAUSS || (e->lik.kind == LIK_MIX && e->lik.ncomp <= 8) ||
                        (e->lik.kind == LIK_USER && user_lik_small_ok(*e->lik.user));  // (a user's source in block form)
  const bool persist = fused && e->lpc <= 8 && fast_lik && e->diag && e->vec4 && !e->opt_mask && e->ncu > 0 &&
                       nown_one_block <= (size_t)POWN_MAX * (size_t)e->ncu && nown <= POWN_MAX * e->ncu && nburn / 50 + 2 <= PEVENTS &&
                       mcxk_persist_lds_bytes(plpc2, pbpl, (nown + std::min(nown, e->ncu) - 1) / std::max(std::min(nown, e->ncu), 1)) <= MCXK_PERSIST_LDS_LIMIT &&
                       (e->opt_persist > 0 || (e->opt_persist < 0 && e->opt_split != 0)) && !e->persist_broken &&
                       (nburn == 0 || meet_lock_open(e));
  // When the run opens with such a launch, the launch itself takes the initial state (and its likelihood,
  // src/mcpar.cc:47-53) and the factor as installed, and starts its counters afresh: no reset kernels at all.
  bool lead = persist && nburn + nsamp > 0;
  // this run's counter block: a ring, zeroed as a whole when it wraps
  e->ctr_set = (e->ctr_set + 1) % CTR_RING;
  if (e->ctr_set == 0) {
    for (int sl = 0; sl < mcx_engine::HSLOTS; ++sl)  // (an asynchronous run's counters may still be on their way out of the ring)
      if (e->copy_pending[sl]) HIPCHK(hipStreamWaitEvent(st, e->copy_ev[sl], 0));
    HIPCHK(hipMemsetAsync(e->ctr.p, 0, (size_t)CTR_WORDS * CTR_RING * sizeof(unsigned long long), st));
  }
  unsigned long long *const ctrp = e->ctr.p + (size_t)e->ctr_set * CTR_WORDS;
  // ... and its slot of the pinned ring the counters end in (slots in turn: with MCX_OPT_ASYNC_RUN earlier runs' may not
  // have been read yet)
  if (!e->h_ctr.p) {
    MCXCHK(e->h_ctr.alloc(8 * mcx_engine::HSLOTS));  // pinned: the copy queues behind the last kernel instead of staging through the runtime
    memset(e->h_ctr.p, 0, 8 * mcx_engine::HSLOTS * sizeof(unsigned long long));
  }
  e->hctr_slot = (e->hctr_slot + 1) % mcx_engine::HSLOTS;
  if (e->copy_pending[e->hctr_slot]) {  // four runs back: long over
    HIPCHK(hipEventSynchronize(e->copy_ev[e->hctr_slot]));
    e->copy_pending[e->hctr_slot] = false;
  }
  if (e->superseded_mask & (1u << e->hctr_slot)) note_superseded(e, false);  // (its word, before the slot is written again)
  e->superseded_mask &= ~(1u << e->hctr_slot);
  e->run_queued[e->hctr_slot] = false;
  e->slot_serial[e->hctr_slot] = 0;
  unsigned long long *const hctr = e->h_ctr.p + 8 * e->hctr_slot;
  unsigned long long reported = 0;  // serial of the launch that reports the run's end itself (RunArgs::report), if one does
  // pinit is pageable caller memory: the runtime stages it before hipMemcpyAsync returns
  if (pinit) HIPCHK(hipMemcpyAsync(e->pvals.p, pinit, (size_t)e->ntot * sizeof(float), hipMemcpyHostToDevice, st));  // :47-50
  if (!lead) {
    hipLaunchKernelGGL(k_run_reset, dim3(nblocks((size_t)e->ntot)), dim3(BLOCK), 0, st, e->acc_slots.p, (size_t)e->nslots,
                       e->acc_cnt.p, (size_t)n, e->ntrace.p, e->pvals.p, pinit ? (const float *)nullptr : staged,
                       (size_t)e->ntot, e->cov.p, e->cov_pending ? (const float *)e->cov0.p : (const float *)nullptr,
                       (size_t)e->ncov);
    HIPCHK(hipGetLastError());
    if (e->cov_pending) {  // (cov_reset's copy went with the reset kernel)
      e->cov_pending = false;
      e->cov_offdiag = !e->diag;
    }
    MCXCHK(eval_trials(e, e->pvals.p, e->lylast.p, 0));  // :53
  }

  e->ht_mark[0] = std::chrono::steady_clock::now();
  SegArgs sa;
  sa.x = e->pvals.p; sa.ly = e->lylast.p; sa.mu = e->mu.p; sa.psum2 = e->psum2.p;
  sa.acc_cnt = e->acc_cnt.p; sa.acc_slots = e->acc_slots.p; sa.T = e->cov.p; sa.lik = e->lik.params.p; sa.ncomp = e->lik.ncomp;
  sa.n = n; sa.d = d; sa.g0 = g0; sa.seed = e->seed; sa.diag = e->diag ? 1 : 0; sa.vec4 = e->vec4;
  sa.winv = e->winv_tab.p;
  sa.samp_stride = e->opt_stride;
  sa.musig_own = e->musigall.p + 2 * (size_t)e->rank * e->ntot;
  sa.snap_after = -1;
  sa.zpre = sa.upre = nullptr;
  sa.trash = nullptr;
  sa.init_moments = 0;
  sa.sig_out = nullptr;
  sa.tun = SegArgs::Tuner{};
  sa.tun.ncov = e->ncov; sa.tun.nslots = e->nslots; sa.tun.ctr = ctrp; sa.tun.T = e->cov.p; sa.tun.trace = e->trace.p;
  sa.tun.ntrace = e->ntrace.p; sa.tun.cells = e->tun_cells.p;
  sa.tun.armin = e->TGT_ARATE_MIN; sa.tun.armax = e->TGT_ARATE_MAX; sa.tun.dfac = e->SCALE_DEC; sa.tun.ifac = e->SCALE_INC;
  bool init_pending = false;  // MCX_PLAN_INIT_MOMENTS handed to the next main segment's launch

  const PlanCfg cfg = {nsamp, nburn, e->SYNCSTEP, e->PLOCAL, e->seed, e->tbase, e->size > 1, e->opt_eager != 0,
                       fused, e->ofn != nullptr, e->opt_maxseg, sink ? sblock : 0};
  const std::vector<mcx_plan_item> plan = build_plan(cfg);
  bool sig_done = false, slots_used = false;
  int sink_seq = 0;
  for (size_t pi = 0; pi < plan.size(); ++pi) {
    const mcx_plan_item &it = plan[pi];
    if (persist && (it.kind == MCX_PLAN_BURN_SEGMENT || it.kind == MCX_PLAN_INIT_MOMENTS || it.kind == MCX_PLAN_MAIN_SEGMENT)) {
      size_t pj = pi;
      int pb = 0, pm = 0, init = 0, is0 = 0, snap = -1;
      while (pj < plan.size() && (plan[pj].kind == MCX_PLAN_BURN_SEGMENT || plan[pj].kind == MCX_PLAN_TUNER)) {
        if (plan[pj].kind == MCX_PLAN_BURN_SEGMENT) pb += plan[pj].nsteps;
        ++pj;
      }
      const size_t pj_burn_end = pj;
      if (pj + 1 < plan.size() && plan[pj].kind == MCX_PLAN_INIT_MOMENTS) {
        // (a sharded run's first sync point is step 0: its slot publish -- of zero steps, i.e. nothing -- sits between
        // the start of the moments and the first segment and must not cost the run a second launch)
        size_t pk = pj + 1;
        while (pk < plan.size() && plan[pk].kind == MCX_PLAN_PUBLISH && plan[pk].first == 0) ++pk;
        if (pk < plan.size() && plan[pk].kind == MCX_PLAN_MAIN_SEGMENT) {
          init = 1;
          pj = pk;
        }
      }
      if (pj < plan.size() && plan[pj].kind == MCX_PLAN_MAIN_SEGMENT) {
        is0 = plan[pj].first;
        while (pj < plan.size() && plan[pj].kind == MCX_PLAN_MAIN_SEGMENT && plan[pj].first == is0 + pm) {
          if (plan[pj].aux >= 0) snap = pm + plan[pj].aux;  // the last sync point inside the stretch
          pm += plan[pj].nsteps;
          ++pj;
        }
      }
      if (e->xchg_pending && pb > 0 && e->opt_meet_under_gather <= 0) {
        // The last run's final gather is still in flight (finish_tail) and this launch has tuner meetings: every one
        // of its workgroups must become resident while the gather's kernel holds whatever it holds -- and that kernel
        // may itself be waiting for a peer GPU whose gather kernel cannot start beside the peer's small-n launch.  No
        // such cycle can form if the launch with meetings starts behind the gather: the step stream waits for it here
        // (MCX_OPT_MEET_UNDER_GATHER = 1 lets the burn-in run under the gather instead; the meetings' timeout is then
        // the net).  Launches without meetings have no workgroup waiting for another and need no such care.
        MCXCHK(exchange_wait(e));
      }
      if (e->xchg_pending && pb > 0 && pm > 0 && snap >= 0) {
        // The last run's final gather is still in flight (finish_tail) and this stretch will rewrite the slot it
        // reads: the burn-in, which does not touch the slot, goes first in a launch of its own and runs under the
        // gather; the main-loop stretch follows in a second launch, behind the wait.
        pj = pj_burn_end;
        pm = 0; init = 0; is0 = 0; snap = -1;
      }
      if (pb + pm > 0) {
        RunArgs ra;
        ra.x = e->pvals.p; ra.ly = e->lylast.p; ra.mu = e->mu.p; ra.psum2 = e->psum2.p; ra.sig = e->sig.p;
        ra.acc_cnt = e->acc_cnt.p; ra.T = e->cov.p;
        ra.samp_x = ra.samp_ly = nullptr;
        if (e->opt_samples && pm > 0) samp_vbase(e, is0, &ra.samp_x, &ra.samp_ly);
        ra.samp_stride = e->opt_stride;
        MCXCHK(e->trash.alloc((size_t)PTRASH * (size_t)PBLOCK * (size_t)std::max(e->ncu, 1)));
        ra.trash = e->trash.p;
        ra.lik = e->lik.params.p; ra.ncomp = e->lik.ncomp; ra.n = n; ra.d = d; ra.g0 = g0; ra.seed = e->seed;
        ra.t0 = pb > 0 ? e->tbase : e->tbase + (uint32_t)nburn + (uint32_t)is0;
        ra.nburn = pb; ra.nmain = pm; ra.isamp0 = is0; ra.init_moments = init;
        ra.winv = e->winv_tab.p; ra.musig_own = sa.musig_own; ra.snap_after = snap;
        ra.final_publish = (e->size == 1 && pm > 0 && is0 + pm == nsamp) ? 1 : 0;
        ra.armin = e->TGT_ARATE_MIN; ra.armax = e->TGT_ARATE_MAX; ra.dfac = e->SCALE_DEC; ra.ifac = e->SCALE_INC;
        ra.ctr = ctrp; ra.bar = ctrp + 8; ra.trace = e->trace.p; ra.ntrace = e->ntrace.p;
        // the run's first launch takes the state where it lies, evaluates it, and starts the counters afresh
        ra.x0 = lead ? (pinit ? e->pvals.p : staged) : nullptr;
        ra.T0 = e->cov_pending ? e->cov0.p : nullptr;
        ra.fresh = lead ? 1 : 0;
        if (e->cov_pending && e->cov_offdiag) {
          // the kernel writes back the diagonal only: a full factor left in cov by an earlier run must not
          // survive next to it (mcx_get_chol would return a mixture)
          HIPCHK(hipMemcpyAsync(e->cov.p, e->cov0.p, (size_t)e->ncov * sizeof(float), hipMemcpyDeviceToDevice, st));
          e->cov_offdiag = false;
        }
        e->cov_pending = false;
        lead = false;
        ra.nown = nown;
        const int nwg = std::min(nown, e->ncu);
        ra.own = (nown + nwg - 1) / nwg;
        ra.ksteps = mcxk_persist_ksteps(plpc2, pbpl, ra.own);
        {  // who generates what: rebuilt and uploaded only when the launch configuration changes
          const int prec = mcxk_persist_recorders(ra.own, pbpl) ? 1 : 0;
          const long long key = (((long long)plpc2 * 8 + pbpl) * 16 + ra.own) * 64 + ra.ksteps + 4096ll * 1024 * prec;
          if (key != e->deal_key) {
            HIPCHK(hipStreamSynchronize(st));  // (an earlier launch may still read the table)
            e->h_deal.assign((size_t)MCXK_PERSIST_DEAL_WORDS, 0u);
            mcxk_persist_deal(plpc2, pbpl, ra.own, prec, ra.ksteps, e->h_deal.data());
            MCXCHK(e->deal_tab.alloc((size_t)MCXK_PERSIST_DEAL_WORDS));
            HIPCHK(hipMemcpyAsync(e->deal_tab.p, e->h_deal.data(), e->h_deal.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            e->deal_key = key;
          }
          ra.deal = e->deal_tab.p;
        }
        ra.trace_clk = nullptr;
        if (const char *tf = getenv("MCX_PERSIST_TRACE")) {  // (a -DMCX_PERSIST_TRACE build writes it: tools/persist_trace.py)
          if (*tf) {
            MCXCHK(e->trace_clk.alloc((size_t)PTRACE_WG * PWAVES * PTRACE_PH * 2));
            HIPCHK(hipMemsetAsync(e->trace_clk.p, 0, (size_t)PTRACE_WG * PWAVES * PTRACE_PH * 2 * sizeof(unsigned long long), st));
            ra.trace_clk = e->trace_clk.p;
          }
        }
        ra.meet_timeout = (unsigned long long)e->opt_meet_timeout_ms * 100000ull;  // s_memrealtime: 100 MHz
        ra.meet_expect_extra = e->opt_debug_meet;
        // the launch that ends the run -- nothing of the plan left, variances and slot written by itself, accept counts its
        // own -- also tells the host: no counters' copy behind it
        ra.report = nullptr; ra.report_done = nullptr; ra.report_serial = 0;
        bool plan_over = true;  // (what is left of the plan: the slot's final publish, which the launch does itself / of no step)
        for (size_t pk = pj; pk < plan.size(); ++pk) plan_over = plan_over && plan[pk].kind == MCX_PLAN_PUBLISH && plan[pk].first == nsamp;
        if (e->opt_self_report && plan_over && (ra.final_publish || nsamp == 0) && !slots_used && !sink && !e->ofn &&
            !e->opt_profile && e->size == 1) {
          reported = ++e->report_serial;
          ra.report = hctr;
          ra.report_done = reinterpret_cast<unsigned *>(e->ctr.p + (size_t)CTR_WORDS * CTR_RING);
          ra.report_serial = reported;
        }
        if (snap >= 0) {  // the kernel rewrites this shard's slot: no gather may still be reading it
          MCXCHK(exchange_wait(e));
          e->published_steps = is0 + snap + 1;
        }
        if (ra.final_publish) {
          MCXCHK(exchange_wait(e));
          e->published_steps = nsamp;
          sig_done = true;
        }
        // tuner events inside: exclusive on this GPU until the kernel has completed.  The lock is given back at
        // the run's next synchronisation with the stream -- before any user hook may block this thread, at the latest
        // at the end of the run
        if (pb > 0) MCXCHK(meet_lock_take(e));
        {
          ProfScope ps(e, MCX_K_RUN_SMALL, (uint64_t)(pb + pm) * n);
          const hipError_t le = e->lik.kind == LIK_USER ? user_lik_launch_small(*e->lik.user, pbpl, ra, st)
                                                        : mcxk_launch_persist(e->lpc, pbpl, e->lik.kind, ra, st);
          if (le != hipSuccess) (void)meet_release(e, true);
          if (le == hipErrorCooperativeLaunchTooLarge) {  // the grid cannot be resident at once on this device
            (void)hipGetLastError();
            return MCX_INTERNAL_MEET_ABANDONED;
          }
          HIPCHK(le);
          if (pb > 0) {  // looked at by meet_release, at the latest at the end of the run
            e->meet_check = true;
            e->meet_word = ctrp + 5;
          }
          e->cnt.small_n_launches++;
          e->cnt.small_n_blocks_per_lane = (uint64_t)pbpl;
        }
        pi = pj - 1;
        continue;
      }
    }
    const int isamp = it.first, steps = it.nsteps;
    switch (it.kind) {
    case MCX_PLAN_BURN_SEGMENT: {  // src/mcpar.cc:58-75
      const uint32_t t0 = e->tbase + (uint32_t)isamp;
      if (fused) {
        sa.samp_x = sa.samp_ly = nullptr;
        sa.mask = e->opt_mask ? e->mask.p + (size_t)isamp * n : nullptr;
        sa.nsteps = steps; sa.t0 = t0; sa.isamp0 = 0; sa.snap_after = -1;
        // the tuner event that follows the segment: inside the launch where the kernel can (its last workgroup)
        const bool fold = pi + 1 < plan.size() && plan[pi + 1].kind == MCX_PLAN_TUNER && fused_takes_epilogue(e, sa);
        sa.tun.on = fold ? 1 : 0;
        if (fold) {
          sa.tun.check = plan[pi + 1].aux;
          sa.tun.add_trials = (unsigned long long)plan[pi + 1].nsteps * (unsigned long long)n;
        }
        {
          ProfScope ps(e, MCX_K_FUSED_BURN, (uint64_t)steps * n);
          MCXCHK(launch_fused(e, false, sa, st));
        }
        sa.tun.on = 0;
        if (fold) ++pi;
        else slots_used = true;
      } else {
        slots_used = true;
        for (int s = 0; s < steps; ++s) {
          StepArgs a;
          fill_step(e, a, t0 + (uint32_t)s, 0, false, (size_t)(isamp + s), 0, 0);
          MCXCHK(launch_propose(e, a));
          MCXCHK(eval_trials(e, e->ptrial.p, e->lytrial.p, (uint64_t)n));
          MCXCHK(launch_accept(e, a, false));
        }
      }
      break;
    }
    case MCX_PLAN_TUNER: {  // src/mcpar.cc:77-96
      ProfScope ps(e, MCX_K_TUNER, 0);
      hipLaunchKernelGGL(k_tuner, dim3(1), dim3(BLOCK), 0, st, ctrp, e->cov.p, e->ncov,
                         (unsigned long long)steps * (unsigned long long)n, it.aux, e->TGT_ARATE_MIN,
                         e->TGT_ARATE_MAX, e->SCALE_DEC, e->SCALE_INC, e->trace.p, e->ntrace.p, e->acc_slots.p,
                         e->nslots);
      HIPCHK(hipGetLastError());
      break;
    }
    case MCX_PLAN_INIT_MOMENTS:  // src/mcpar.cc:99-104
      if (fused && pi + 1 < plan.size() && plan[pi + 1].kind == MCX_PLAN_MAIN_SEGMENT) {
        SegArgs probe = sa;
        probe.mask = e->opt_mask ? e->mask.p : nullptr;
        probe.nsteps = plan[pi + 1].nsteps;
        if (fused_takes_epilogue(e, probe)) {  // the segment's kernel starts from (0, FPEPS) instead of loading them
          init_pending = true;
          break;
        }
      }
      hipLaunchKernelGGL(k_init_moments, dim3(nblocks((size_t)e->ntot)), dim3(BLOCK), 0, st, e->mu.p,
                         e->psum2.p, (size_t)e->ntot);
      HIPCHK(hipGetLastError());
      break;
    case MCX_PLAN_OUTPUT:  // src/mcpar.cc:115-119
      HIPCHK(hipStreamSynchronize(st));
      MCXCHK(meet_release(e, true));
      e->samp_steps = e->opt_samples ? (isamp + e->opt_stride - 1) / e->opt_stride : 0;
      if (e->ofn(e->octx, isamp) != 0) return fail(MCX_ERR_INVALID, "output hook failed");
      break;
    case MCX_PLAN_SINK: MCXCHK(sink_block_done(e, isamp, steps, sink_seq++)); break;
    case MCX_PLAN_PUBLISH: MCXCHK(publish(e, isamp)); break;
    case MCX_PLAN_GATHER_BEGIN: MCXCHK(exchange_begin(e)); break;  // src/mcpar.cc:127-140
    case MCX_PLAN_GATHER_WAIT:
      if (isamp == nsamp && e->xchg_pending && exchange_tail_may_stay_in_flight(e) &&
          pi + 2 == plan.size() && plan[pi + 1].kind == MCX_PLAN_PUBLISH) {
        e->tail_publish = nsamp;  // finish_tail: the run's last gather stays in flight
        ++pi;
        break;
      }
      MCXCHK(exchange_wait(e));
      break;
    case MCX_PLAN_REMOTE_STEP: {  // src/mcpar.cc:152-175 with genRemote
      MCXCHK(meet_release(e, false));  // (genRemote synchronises with the stream after every pass anyway)
      const uint32_t t = e->tbase + (uint32_t)nburn + (uint32_t)isamp;
      int npass = 0;
      MCXCHK(remote_device(e, t, e->pvals.p, e->musigall.p, e->ptrial.p, e->cfac.p, e->mutrial.p,
                           e->sigtrial.p, &npass));
      e->cnt.remote_steps++;
      e->cnt.remote_passes += (uint64_t)npass;
      MCXCHK(eval_trials(e, e->ptrial.p, e->lytrial.p, (uint64_t)n));  // :160
      StepArgs a;
      fill_step(e, a, t, isamp, true, (size_t)(nburn + isamp), isamp, 1);
      MCXCHK(launch_accept(e, a, true));
      MCXCHK(discarded_calls(e));  // :177-182, on request
      slots_used = true;
      break;
    }
    case MCX_PLAN_MAIN_SEGMENT: {  // src/mcpar.cc:152-209 with genLocal
      const uint32_t t = e->tbase + (uint32_t)nburn + (uint32_t)isamp;
      if (fused) {
        const size_t row0 = e->opt_stride == 1 ? (size_t)isamp : 0;  // thinned: the kernel indexes from step 0
        float *vx = nullptr, *vl = nullptr;
        if (e->opt_samples) samp_vbase(e, isamp, &vx, &vl);
        sa.samp_x = e->opt_samples ? vx + row0 * e->ntot : nullptr;
        sa.samp_ly = e->opt_samples ? vl + row0 * n : nullptr;
        sa.mask = e->opt_mask ? e->mask.p + (size_t)(nburn + isamp) * n : nullptr;
        sa.nsteps = steps; sa.t0 = t; sa.isamp0 = isamp;
        sa.snap_after = it.aux;
        if (it.aux >= 0) {  // the kernel rewrites this shard's slot: no gather may still be reading it
          MCXCHK(exchange_wait(e));
          e->published_steps = isamp + it.aux + 1;
        }
        sa.init_moments = init_pending ? 1 : 0;
        init_pending = false;
        // hot-path kernels count their accepted proposals themselves (SegArgs::tun.on = 2), and the one that holds the
        // run's last step leaves the variances and this shard's slot behind (one shard: no gather may want the slot as
        // of the last sync point): no k_reduce_slots / k_variance / k_publish at the end of the run
        const bool self = fused_takes_epilogue(e, sa);
        sa.tun.on = self ? 2 : 0;
        if (self && e->size == 1 && isamp + steps == nsamp && it.aux < 0) {
          sa.snap_after = steps - 1;
          sa.sig_out = e->sig.p;
          e->published_steps = nsamp;
          sig_done = true;
        }
        {
          ProfScope ps(e, MCX_K_FUSED_MAIN, (uint64_t)steps * n);
          MCXCHK(launch_fused(e, true, sa, st));
        }
        sa.init_moments = 0;
        sa.tun.on = 0;
        sa.sig_out = nullptr;
        if (!self) slots_used = true;
      } else {
        slots_used = true;
        for (int s = 0; s < steps; ++s) {
          StepArgs a;
          fill_step(e, a, t + (uint32_t)s, isamp + s, true, (size_t)(nburn + isamp + s), isamp + s, 0);
          MCXCHK(launch_propose(e, a));
          MCXCHK(eval_trials(e, e->ptrial.p, e->lytrial.p, (uint64_t)n));
          MCXCHK(launch_accept(e, a, true));
          MCXCHK(discarded_calls(e));  // :177-182, on request
        }
      }
      break;
    }
    default: return fail(MCX_ERR_INVALID, "internal: unknown plan item %d", it.kind);
    }
  }
  e->cnt.nsteps_burn = (uint64_t)nburn;
  e->cnt.nsteps_main = (uint64_t)nsamp;
  if (nsamp > 0 && !sig_done) {
    hipLaunchKernelGGL(k_variance, dim3(nblocks((size_t)e->ntot)), dim3(BLOCK), 0, st, e->psum2.p,
                       e->sig.p, (size_t)e->ntot, 1.0f / (float)nsamp);
    HIPCHK(hipGetLastError());
  }
  if (slots_used) {
    hipLaunchKernelGGL(k_reduce_slots, dim3(1), dim3(BLOCK), 0, st, e->acc_slots.p, e->nslots, ctrp + 4);
    HIPCHK(hipGetLastError());
  }
  MCXCHK(cov_reset(e));  // (a run without any step)
  // MCX_OPT_ASYNC_RUN: everything is queued -- return.  Only runs whose end needs nothing from the host: one shard, no
  // sink / output hook / host likelihood, no Murray step (a pass waits for its survivors' count), no profiling.
  bool go_async = e->opt_async_run && !rerun && e->size == 1 && !sink && !e->ofn && nsamp > 0 && e->lik.kind != MCX_VL_HOST &&
                  !e->opt_profile && !e->trace_clk.p;
  for (const mcx_plan_item &it : plan) go_async = go_async && it.kind != MCX_PLAN_REMOTE_STEP && it.kind != MCX_PLAN_OUTPUT;
  if (go_async) {
    if (pinit) {  // a repeat (abandoned meeting) must not need the caller's memory again
      MCXCHK(e->pinit_async.alloc((size_t)e->ntot));
      HIPCHK(hipMemcpyAsync(e->pinit_async.p, pinit, (size_t)e->ntot * sizeof(float), hipMemcpyHostToDevice, st));
    }
    // the counters' way to the host on a stream of its own, behind an event: on the step stream the copy would sit between
    // this run's last kernel and the next run's first one (5-8 us of every queued job)
    const int sl = e->hctr_slot;
    if (reported && !pinit) {  // (the run's last launch is the last thing on the stream and reports itself: nothing to add)
      e->slot_serial[sl] = reported;
      e->run_queued[sl] = true;
    } else {
      reported = 0;
      if (!e->astream) HIPCHK(hipStreamCreateWithFlags(&e->astream, hipStreamNonBlocking));
      if (!e->run_ev[sl]) HIPCHK(hipEventCreateWithFlags(&e->run_ev[sl], hipEventDisableTiming));
      if (!e->copy_ev[sl]) HIPCHK(hipEventCreateWithFlags(&e->copy_ev[sl], hipEventDisableTiming));
      HIPCHK(hipEventRecord(e->run_ev[sl], st));
      e->run_queued[sl] = true;
      HIPCHK(hipStreamWaitEvent(e->astream, e->run_ev[sl], 0));
      HIPCHK(hipMemcpyAsync(hctr, ctrp, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->astream));
      HIPCHK(hipEventRecord(e->copy_ev[sl], e->astream));
      e->copy_pending[sl] = true;
    }
    e->ht_mark[1] = std::chrono::steady_clock::now();
    e->pend.active = true;
    e->pend.slot = sl;
    e->pend.nsamp = nsamp; e->pend.nburn = nburn; e->pend.tbase0 = e->tbase;
    e->pend.meet_check = e->meet_check; e->pend.hctr = hctr; e->pend.host_pinit = pinit != nullptr;
    e->pend.serial = reported;
    e->meet_check = false;  // (finish_pending looks at the word itself)
    e->samp_steps = e->opt_samples ? nkeep : 0;
    e->last_nsamp = nsamp; e->last_nburn = nburn; e->have_run = true;
    e->tbase += (uint32_t)(nburn + nsamp);
    return MCX_OK;
  }
  if (!reported) HIPCHK(hipMemcpyAsync(hctr, ctrp, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  e->ht_mark[1] = std::chrono::steady_clock::now();
  {
    hipError_t se = hipSuccess;
    // (spinning only for runs of the length the last one had: whoever runs 10 ms jobs sleeps as ever)
    const auto w0 = std::chrono::steady_clock::now();
    if (!(reported && e->report_wait_us < (double)REPORT_SPIN_US && report_arrived(hctr, reported, REPORT_SPIN_US))) se = hipStreamSynchronize(st);
    if (reported) e->report_wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    if (se == hipSuccess && reported && __atomic_load_n(hctr + 7, __ATOMIC_ACQUIRE) != reported) {
      (void)meet_release(e, true);
      return fail(MCX_ERR_HIP, "internal: the run's last launch is over and has not reported");
    }
    e->ht_mark[2] = std::chrono::steady_clock::now();
    const bool abandoned = e->meet_check && se == hipSuccess && hctr[5] != 0;  // (the word came with the counters)
    e->meet_check = false;
    (void)meet_release(e, true);
    HIPCHK(se);
    if (abandoned) return MCX_INTERNAL_MEET_ABANDONED;
  }
  e->cnt.naccept_burn = hctr[3];
  e->cnt.naccept_main = hctr[4];
  xwait_collect(e);
  if (e->trace_clk.p) {  // debug: the phase clocks of the run's last small-n launch, as raw u64 words
    const char *tf = getenv("MCX_PERSIST_TRACE");
    if (tf && *tf) {
      std::vector<unsigned long long> h((size_t)PTRACE_WG * PWAVES * PTRACE_PH * 2);
      HIPCHK(hipMemcpy(h.data(), e->trace_clk.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE *f = fopen(tf, "wb")) {
        (void)fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        fclose(f);
      }
    }
  }
  if (sink) MCXCHK(sink_drain(e, sink_seq));
  e->samp_steps = (e->opt_samples && !sink) ? nkeep : 0;
  e->last_nsamp = nsamp;
  e->last_nburn = nburn;
  e->have_run = true;
  e->tbase += (uint32_t)(nburn + nsamp);
  prof_collect(e);
  if (e->ofn && e->ofn(e->octx, nsamp) != 0) return fail(MCX_ERR_INVALID, "output hook failed");  // :212
  return MCX_OK;  // :213
}

extern "C" int mcx_stage_pinit(mcx_engine *e, const float *pinit)
{
  MCXCHK(enter(e));
  if (!e || !pinit) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(e->pinit_dev.alloc((size_t)e->ntot));
  HIPCHK(hipMemcpyAsync(e->pinit_dev.p, pinit, (size_t)e->ntot * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->pinit_staged = true;
  return MCX_OK;
}

// ---------------------------------------------------------------------------------------------
// standalone operators on host buffers
// ---------------------------------------------------------------------------------------------
extern "C" int mcx_vlfunc_eval(const mcx_vlfunc *f, int npset, const float *x, float *y)
{
  if (!f || npset < 0 || (npset > 0 && (!x || !y))) return fail(MCX_ERR_INVALID, "bad arguments");
  if (f->d < 1 || f->d > MAXD) return fail(MCX_ERR_UNSUPPORTED, "d = %d outside 1..%d", f->d, MAXD);
  if (f->kind == MCX_VL_HOST) {
    if (!f->fn) return fail(MCX_ERR_VLFUNC, "MCX_VL_HOST without a callback");
    return f->fn(f->ctx, npset, x, y);
  }
  MCXCHK(need_device());
  LikDev L;
  hipStream_t st = nullptr;
  int rc = lik_setup(L, f, f->d, st);
  if (rc == MCX_OK && L.kind == MCX_VL_DEVICE) {
    DevBuf<float> ux, uy;
    rc = ux.alloc((size_t)std::max(npset, 1) * f->d);
    if (rc == MCX_OK) rc = uy.alloc((size_t)std::max(npset, 1));
    if (rc == MCX_OK && npset > 0) {
      auto run = [&]() -> int {
        HIPCHK(hipMemcpy(ux.p, x, (size_t)npset * f->d * sizeof(float), hipMemcpyHostToDevice));
        const float *xa = ux.p;
        float *ya = uy.p;
        void *args[] = {&npset, &xa, &ya};
        HIPCHK(hipModuleLaunchKernel((hipFunction_t)L.ctx, nblocks((size_t)npset), 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(y, uy.p, (size_t)npset * sizeof(float), hipMemcpyDeviceToHost));
        return MCX_OK;
      };
      rc = run();
    }
    ux.release(); uy.release();
    return rc;
  }
  if (rc != MCX_OK || npset == 0) {
    (void)hipDeviceSynchronize();  // the parameter upload reads L.host
    L.params.release();
    return rc;
  }
  DevBuf<float> dx, dy;
  rc = dx.alloc((size_t)npset * f->d);
  if (rc == MCX_OK) rc = dy.alloc((size_t)npset);
  if (rc == MCX_OK) {
    auto run = [&]() -> int {
      HIPCHK(hipMemcpy(dx.p, x, (size_t)npset * f->d * sizeof(float), hipMemcpyHostToDevice));
      MCXCHK(eval_device(L, dx.p, dy.p, npset, f->d, st));
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(y, dy.p, (size_t)npset * sizeof(float), hipMemcpyDeviceToHost));
      return MCX_OK;
    };
    rc = run();
  }
  dx.release(); dy.release(); L.params.release();
  return rc;
}

extern "C" int mcx_gen_local(mcx_engine *e, uint32_t t, const float *pvals, float *ptrial, float *cfac)
{
  MCXCHK(enter(e));
  if (!e || !pvals || !ptrial || !cfac) return fail(MCX_ERR_INVALID, "bad arguments");
  hipStream_t st = e->stream;
  DevBuf<float> x;
  MCXCHK(x.alloc((size_t)e->ntot));
  auto run = [&]() -> int {
    HIPCHK(hipMemcpyAsync(x.p, pvals, (size_t)e->ntot * sizeof(float), hipMemcpyHostToDevice, st));
    StepArgs a;
    fill_step(e, a, t, 0, false, 0, 0, 0);
    a.x = x.p;
    a.mask = nullptr;
    MCXCHK(launch_propose(e, a));
    HIPCHK(hipMemcpyAsync(ptrial, e->ptrial.p, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(cfac, e->cfac.p, (size_t)e->nchain * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MCX_OK;
  };
  const int rc = run();
  x.release();
  return rc;
}

extern "C" int mcx_gen_remote(mcx_engine *e, uint32_t t, const float *pvals, const float *musigall,
                              float *ptrial, float *cfac, float *mutrial, float *sigtrial, int *npass)
{
  MCXCHK(enter(e));
  if (!e || !pvals || !musigall || !ptrial || !cfac) return fail(MCX_ERR_INVALID, "bad arguments");
  hipStream_t st = e->stream;
  DevBuf<float> x, ms;
  MCXCHK(x.alloc((size_t)e->ntot));
  int rc = ms.alloc(2 * (size_t)e->tchains * e->nparam);
  if (rc != MCX_OK) { x.release(); return rc; }
  auto run = [&]() -> int {
    HIPCHK(hipMemcpyAsync(x.p, pvals, (size_t)e->ntot * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ms.p, musigall, 2 * (size_t)e->tchains * e->nparam * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(e->cfac.p, 0, (size_t)e->nchain * sizeof(float), st));
    MCXCHK(remote_device(e, t, x.p, ms.p, e->ptrial.p, e->cfac.p, e->mutrial.p, e->sigtrial.p, npass));
    HIPCHK(hipMemcpyAsync(ptrial, e->ptrial.p, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(cfac, e->cfac.p, (size_t)e->nchain * sizeof(float), hipMemcpyDeviceToHost, st));
    if (mutrial) HIPCHK(hipMemcpyAsync(mutrial, e->mutrial.p, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, st));
    if (sigtrial) HIPCHK(hipMemcpyAsync(sigtrial, e->sigtrial.p, (size_t)e->ntot * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MCX_OK;
  };
  rc = run();
  x.release(); ms.release();
  return rc;
}

// ---------------------------------------------------------------------------------------------
// getters
// ---------------------------------------------------------------------------------------------
template <typename T>
static int d2h(mcx_engine *e, T *dst, const T *src, size_t count)
{
  if (!e || !dst) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(enter(e));  // (also: an asynchronous run in flight is finished first)
  HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MCX_OK;
}

extern "C" int mcx_get_counters(mcx_engine *e, mcx_counters *c)
{
  if (!e || !c) return fail(MCX_ERR_INVALID, "bad arguments");
  if (e->pend.active) MCXCHK(enter(e));  // (an asynchronous run's counters exist once it is over)
  *c = e->cnt;
  c->meet_timeouts_total = e->meet_total;
  return MCX_OK;
}
extern "C" int mcx_get_state(mcx_engine *e, float *v) { return d2h(e, v, e ? e->pvals.p : nullptr, e ? (size_t)e->ntot : 0); }
extern "C" int mcx_get_loglike(mcx_engine *e, float *v) { return d2h(e, v, e ? e->lylast.p : nullptr, e ? (size_t)e->nchain : 0); }
extern "C" int mcx_get_mean(mcx_engine *e, float *v) { return d2h(e, v, e ? e->mu.p : nullptr, e ? (size_t)e->ntot : 0); }
extern "C" int mcx_get_var(mcx_engine *e, float *v) { return d2h(e, v, e ? e->sig.p : nullptr, e ? (size_t)e->ntot : 0); }
extern "C" int mcx_get_musigall(mcx_engine *e, float *v)
{
  MCXCHK(enter(e));
  MCXCHK(finish_tail(e));
  return d2h(e, v, e->musigall.p, 2 * (size_t)e->tchains * e->nparam);
}

extern "C" int mcx_synchronize(mcx_engine *e)
{
  MCXCHK(enter(e));
  MCXCHK(finish_tail(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MCX_OK;
}
extern "C" int mcx_get_chol(mcx_engine *e, float *v) { return d2h(e, v, e ? e->cov.p : nullptr, e ? (size_t)e->ncov : 0); }
extern "C" int mcx_get_accept_counts(mcx_engine *e, uint32_t *v) { return d2h(e, v, e ? e->acc_cnt.p : nullptr, e ? (size_t)e->nchain : 0); }

extern "C" int mcx_get_accept_mask(mcx_engine *e, uint8_t *mask)
{
  MCXCHK(enter(e));
  if (!e || !mask) return fail(MCX_ERR_INVALID, "bad arguments");
  if (!e->have_run || !e->opt_mask || !e->mask.p) return fail(MCX_ERR_INVALID, "no accept mask recorded (MCX_OPT_ACCEPT_MASK)");
  return d2h(e, mask, e->mask.p, (size_t)(e->last_nburn + e->last_nsamp) * e->nchain);
}

extern "C" int mcx_get_tuner_trace(mcx_engine *e, float *scales, int maxn, int *n)
{
  MCXCHK(enter(e));
  if (!e || !n) return fail(MCX_ERR_INVALID, "bad arguments");
  int nt = 0;
  MCXCHK(d2h(e, &nt, e->ntrace.p, 1));
  *n = nt;
  const int k = std::min(std::min(nt, maxn), 256);
  if (k > 0 && scales) MCXCHK(d2h(e, scales, e->trace.p, (size_t)k));
  return MCX_OK;
}

extern "C" int mcx_samples_steps(mcx_engine *e, int *nsteps)
{
  if (!e || !nsteps) return fail(MCX_ERR_INVALID, "bad arguments");
  *nsteps = e->samp_steps;
  return MCX_OK;
}

// Rows are interleaved into MCout's (np+1)-column layout on the device, moved through two pinned
// staging buffers (the D2H of chunk k+1 overlaps the host copy of chunk k) and land in the caller's
// pageable buffer.  Off the hot path: this is the MCout::add / collect side of the boundary.
extern "C" int mcx_samples_copy(mcx_engine *e, int first_step, int nsteps, float *rows)
{
  MCXCHK(enter(e));
  if (!e || !rows || first_step < 0 || nsteps < 0) return fail(MCX_ERR_INVALID, "bad arguments");
  if (first_step + nsteps > e->samp_steps) return fail(MCX_ERR_INVALID, "steps [%d,%d) not in the sample store (%d steps)", first_step, first_step + nsteps, e->samp_steps);
  const size_t n = (size_t)e->nchain, d = (size_t)e->nparam, ncol = d + 1, nr = (size_t)nsteps * n;
  if (nr == 0) return MCX_OK;
  const size_t chunk_rows = std::max<size_t>(1, std::min<size_t>(nr, ((size_t)32 << 20) / (ncol * sizeof(float))));
  DevBuf<float> stage[2];
  float *pin[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  auto run = [&]() -> int {
    for (int b = 0; b < 2; ++b) {
      MCXCHK(stage[b].alloc(chunk_rows * ncol));
      HIPCHK(hipHostMalloc((void **)&pin[b], chunk_rows * ncol * sizeof(float), hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    }
    const float *sx = e->samp_x.p + (size_t)first_step * n * d, *sl = e->samp_ly.p + (size_t)first_step * n;
    size_t issued = 0, copied = 0;
    int ib = 0, cb = 0;
    size_t rows_in[2] = {0, 0};
    while (copied < nr) {
      while (issued < nr && issued - copied < 2 * chunk_rows) {  // keep both buffers in flight
        const size_t r = std::min(chunk_rows, nr - issued);
        hipLaunchKernelGGL(k_rows_interleave, dim3(nblocks(r * ncol)), dim3(BLOCK), 0, e->stream, sx + issued * d,
                           sl + issued, stage[ib].p, r, (int)d);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(pin[ib], stage[ib].p, r * ncol * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipEventRecord(done[ib], e->stream));
        rows_in[ib] = r;
        issued += r;
        ib ^= 1;
      }
      HIPCHK(hipEventSynchronize(done[cb]));
      std::memcpy(rows + copied * ncol, pin[cb], rows_in[cb] * ncol * sizeof(float));
      copied += rows_in[cb];
      cb ^= 1;
    }
    return MCX_OK;
  };
  const int rc = run();
  (void)hipStreamSynchronize(e->stream);
  for (int b = 0; b < 2; ++b) {
    stage[b].release();
    if (pin[b]) (void)hipHostFree(pin[b]);
    if (done[b]) (void)hipEventDestroy(done[b]);
  }
  return rc;
}

extern "C" int mcx_samples_maxlike(mcx_engine *e, float *lmax, float *params)
{
  MCXCHK(enter(e));
  if (!e || !lmax || !params) return fail(MCX_ERR_INVALID, "bad arguments");
  const size_t n = (size_t)e->nchain, d = (size_t)e->nparam;
  if (!e->run_sink) {  // arg-max over the whole HBM-resident store, on the device (first strict maximum, src/mcout.cc:140)
    const size_t nr = (size_t)e->samp_steps * n;
    if (nr == 0) return fail(MCX_ERR_INVALID, "sample store is empty");
    if (nr > 0xfffffff0ull) return fail(MCX_ERR_UNSUPPORTED, "sample store too large for the arg-max key");
    MCXCHK(e->best_row.alloc(d + 1));
    MCXCHK(e->best_key.alloc(1));
    hipLaunchKernelGGL(k_best_reset, dim3(nblocks(d + 1)), dim3(BLOCK), 0, e->stream, e->best_row.p, (int)d, e->best_key.p);
    hipLaunchKernelGGL(k_argmax_first, dim3(std::min<unsigned>(nblocks(nr), 2048u)), dim3(BLOCK), 0, e->stream, e->samp_ly.p, nr, e->best_key.p);
    hipLaunchKernelGGL(k_best_update, dim3(1), dim3(BLOCK), 0, e->stream, e->best_key.p, e->samp_ly.p, e->samp_x.p, (int)d, e->best_row.p);
    HIPCHK(hipGetLastError());
  } else if (!e->have_run) {
    return fail(MCX_ERR_INVALID, "sample store is empty");
  }
  std::vector<float> h(d + 1);
  MCXCHK(d2h(e, h.data(), e->best_row.p, d + 1));
  *lmax = h[0];
  std::copy(h.begin() + 1, h.end(), params);
  return MCX_OK;
}

extern "C" int mcx_get_profile(mcx_engine *e, mcx_profile *p)
{
  if (!e || !p) return fail(MCX_ERR_INVALID, "bad arguments");
  prof_collect(e);
  *p = e->prof;
  return MCX_OK;
}

extern "C" int mcx_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
  if (!dst_host || !src_dev) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return MCX_OK;
}

extern "C" int mcx_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
  if (!dst_dev || !src_host) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return MCX_OK;
}

// ---------------------------------------------------------------------------------------------
// numerics test hooks
// ---------------------------------------------------------------------------------------------
extern "C" int mcx_debug_numerics(int what, int n, const uint32_t *in, uint32_t *out_bits)
{
  if (n < 0 || (n > 0 && (!in || !out_bits))) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  if (n == 0) return MCX_OK;
  DevBuf<uint32_t> di, dout;
  MCXCHK(di.alloc((size_t)n));
  int rc = dout.alloc((size_t)n);
  if (rc == MCX_OK) {
    auto run = [&]() -> int {
      HIPCHK(hipMemcpy(di.p, in, (size_t)n * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_debug_numerics, dim3(nblocks((size_t)n)), dim3(BLOCK), 0, 0, what, n, di.p, dout.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(out_bits, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
      return MCX_OK;
    };
    rc = run();
  }
  di.release(); dout.release();
  return rc;
}

extern "C" int mcx_debug_copy_bandwidth(size_t bytes, int reps, double *gbps)
{
  if (!gbps || bytes == 0 || reps <= 0) return fail(MCX_ERR_INVALID, "copy bandwidth: bytes > 0, reps > 0, gbps != NULL");
  DevBuf<unsigned char> a, b;
  MCXCHK(a.alloc(bytes));
  MCXCHK(b.alloc(bytes));
  hipEvent_t t0 = nullptr, t1 = nullptr;
  int rc = MCX_OK;
  float ms = 0.0f;
  do {
    if (hipMemset(a.p, 1, bytes) != hipSuccess || hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess ||
        hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||  // warm
        hipEventRecord(t0, nullptr) != hipSuccess) { rc = MCX_ERR_HIP; break; }
    for (int r = 0; r < reps && rc == MCX_OK; ++r)
      if (hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = MCX_ERR_HIP;
    if (rc != MCX_OK) break;
    if (hipEventRecord(t1, nullptr) != hipSuccess || hipEventSynchronize(t1) != hipSuccess ||
        hipEventElapsedTime(&ms, t0, t1) != hipSuccess || !(ms > 0.0f)) rc = MCX_ERR_HIP;
  } while (0);
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  a.release();
  b.release();
  if (rc != MCX_OK) return fail(rc, "copy bandwidth: %s", hipGetErrorString(hipGetLastError()));
  *gbps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
  return MCX_OK;
}

extern "C" int mcx_debug_sqrt_sweep(uint32_t lo_bits, uint32_t hi_bits, uint64_t *nbad, uint32_t *first_bad)
{
  if (!nbad || !first_bad || hi_bits < lo_bits) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  DevBuf<unsigned long long> dn;
  DevBuf<uint32_t> df;
  MCXCHK(dn.alloc(1));
  int rc = df.alloc(1);
  if (rc == MCX_OK) {
    auto run = [&]() -> int {
      const uint32_t init = 0xffffffffu;
      HIPCHK(hipMemset(dn.p, 0, sizeof(unsigned long long)));
      HIPCHK(hipMemcpy(df.p, &init, 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_debug_sqrt_sweep, dim3(4096), dim3(BLOCK), 0, 0, lo_bits, hi_bits, dn.p, df.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipDeviceSynchronize());
      unsigned long long n = 0;
      HIPCHK(hipMemcpy(&n, dn.p, sizeof n, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(first_bad, df.p, 4, hipMemcpyDeviceToHost));
      *nbad = n;
      return MCX_OK;
    };
    rc = run();
  }
  dn.release(); df.release();
  return rc;
}

extern "C" int mcx_debug_normals(uint32_t seed, uint32_t stream, uint32_t t, uint32_t g0, uint32_t a,
                                 uint32_t q, int n, float *out)
{
  if (n < 0 || (n > 0 && !out)) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  if (n == 0) return MCX_OK;
  DevBuf<float> d;
  MCXCHK(d.alloc((size_t)n * 4));
  auto run = [&]() -> int {
    hipLaunchKernelGGL(k_debug_normals, dim3(nblocks((size_t)n)), dim3(BLOCK), 0, 0, seed, stream, t, g0, a, q, n, d.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return MCX_OK;
  };
  const int rc = run();
  d.release();
  return rc;
}

