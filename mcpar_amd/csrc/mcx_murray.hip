// mcx_murray.hip -- MCPar::genRemote (src/mcpar.cc:315-451) on device buffers: the schedule of draws, all-pairs sweeps and
// decisions over the kernels of mcx_remote.hpp, with the two exact screens that let a sweep skip rows (boxes of four
// coordinates: mcx_remote.hpp; one direction: mcx_cull_proj.hpp).  A translation unit of its own: its kernels are the
// largest of the library after the step kernels.
#include <atomic>

#include "mcx_engine_internal.hpp"
#include <sched.h>
#include "mcx_remote.hpp"
#include "mcx_cull_proj.hpp"
#include "mcx_screen.hpp"

static_assert(NACT_CULL_CELLS == CULL_NCOUNT, "mcx_engine::nact is sized for the screens' counter cells");

constexpr int SROW_UNMASKED_MAX_CHAINS = 8192;  // see launch_sweep_exact
constexpr int SROW_MAX_BLOCKS_PER_WAVE = 8;
// Few chains left (the late rejection passes: a few hundred chains, then tens, then a handful -- three or four passes per
// Murray step that cost 60-75 us each whatever they sweep, launches and one wait for the host): MULTI_K passes at once.
// The proposal and the test of pass p for chain j are functions of (step, j, p) and of sums over all Gaussians, not of
// what other chains or earlier passes did, so the next MULTI_K proposals of every remaining chain are drawn, swept and
// tested together and each chain takes the first that passes: same proposals, same tests, same order -- same bits, and
// the pass count the reference would have reached.
constexpr int MULTI_MAX_CHAINS = 1024, MULTI_K = 4;  // (from 128 / 448 / 1024 / 2048 chains: C3-murray 15.66 / 15.47 / 15.45 /
                                                     // 15.47 ms, C5 23.10 / 22.15 / 21.93 / 22.12; never: 16.7 / 23.9)

// the all-pairs sweep over chains whose np is a power of two (d == DMAX): one or two chains per lane (SWEEP_CPL).
// Workgroups of 512 / 1024 threads (fewer copies of a block's Gaussians staged through LDS) were measured on the
// two-chain kernels: C3 R-murray 39.2 ms with 256 threads, 40.0 with 512, 49.4 with 1024; the 32-D mixture 43.4 / 42.7
// / 42.7 -- the staging is not what a sweep waits for.
// Returns the number of entries per chain the sweep leaves in pmax for k_remote_cmax_combine (!SUMS): the blocks, or
// fewer where a wavefront carries its minimum through several of them.
// does the row-by-row kernel take this sweep, and in what grid: blocks per wavefront, grid y (see launch_sweep_exact)
template <int DM, bool SUMS>
static bool sweep_is_srow(int cnt, const unsigned long long *excl, int ngroups, int S, int *bpw_out, int *gy_out)
{
  if constexpr (DM == 16 || DM == 32) {
    if (excl || (DM == 16 && cnt <= SROW_UNMASKED_MAX_CHAINS)) {
      int bpw = (excl && !SUMS) ? (int)(((long long)ngroups * S) / 8192) : 1;
      bpw = bpw < 1 ? 1 : (bpw > SROW_MAX_BLOCKS_PER_WAVE ? SROW_MAX_BLOCKS_PER_WAVE : bpw);
      *bpw_out = bpw;
      *gy_out = (S + bpw - 1) / bpw;
      return true;
    }
  }
  return false;
}

// rows [y0, y0 + ny) of the row-by-row sweep's grid (a column chunk of a pass), on stream st
template <int DM, bool SUMS>
static void launch_srow_range(const float *x, const int *list, int cnt, const float *qpar, float *psum, float *pmax, int N, int own0,
                              const unsigned long long *excl, int ngroups, int bpw, int y0, int ny, hipStream_t st)
{
  if constexpr (DM == 16 || DM == 32)
    hipLaunchKernelGGL((k_remote_sweep_srow<DM, SUMS>), dim3((unsigned)((ngroups + BLOCK / 64 - 1) / (BLOCK / 64)), (unsigned)ny), dim3(BLOCK), 0, st,
                       x, list, cnt, qpar, psum, pmax, N, own0, excl, ngroups, bpw, y0);
}

template <int DM, bool SUMS>
static int launch_sweep_exact(const float *x, const int *list, int cnt, const float *qpar, float *psum, float *pmax,
                              int N, int own0, const unsigned long long *excl, int ngroups, int S, hipStream_t st)
{
  constexpr int CPL = SWEEP_CPL(DM);
  if constexpr (DM == 16 || DM == 32) {
    // 16-D masked, or over few chains: every wavefront reads its own rows through the scalar cache (no LDS, no
    // barriers); unmasked over many chains the rows are better staged once per 512 chains (4.3 GB through L2 otherwise).
    // 32-D: masked only (a row is two scalar fetches there) -- and then whatever share of the rows the masks leave: C5 25.5 ms
    // per job so, 26.0-26.8 with the LDS kernel taking over from 15 / 30 / 50 % of the rows kept or everywhere.
    if (excl || (DM == 16 && cnt <= SROW_UNMASKED_MAX_CHAINS)) {
      // masked min-arg sweep: several blocks per wavefront while 8192 wavefronts (eight per SIMD) remain -- 310 -> 72 us
      // per 65 536 x 65 536 sweep at 1 % of the rows.  The sum sweep has no own-Gaussian args to redo per block and
      // its surviving rows are many more in some groups than in others: eight blocks per wavefront 508 -> 600 us.
      int bpw = (excl && !SUMS) ? (int)(((long long)ngroups * S) / 8192) : 1;
      bpw = bpw < 1 ? 1 : (bpw > SROW_MAX_BLOCKS_PER_WAVE ? SROW_MAX_BLOCKS_PER_WAVE : bpw);
      const int gy = (S + bpw - 1) / bpw;
      hipLaunchKernelGGL((k_remote_sweep_srow<DM, SUMS>), dim3((unsigned)((ngroups + BLOCK / 64 - 1) / (BLOCK / 64)), gy), dim3(BLOCK), 0, st,
                         x, list, cnt, qpar, psum, pmax, N, own0, excl, ngroups, bpw, 0);
      return SUMS ? S : gy;
    }
  }
  hipLaunchKernelGGL((k_remote_sweep<DM, SUMS, true, CPL>), dim3(nblocks(((size_t)cnt + CPL - 1) / CPL), S), dim3(BLOCK), 0, st, x,
                     list, cnt, qpar, psum, pmax, DM, N, own0, excl, ngroups);
  return S;
}

// Sort the active chains by their spatial key, box every group of CULL_W of them and test every (group, Q_i)
// pair (mcx_remote.hpp, "Exact exclusion of far Gaussians").  Leaves the sorted list in e->cull_sorted and the
// masks in e->cull_excl ([group][words]); the pairs kept are added to the device counter behind e->nact.
constexpr int CULL_MIN_CHAINS = 4096, CULL_MIN_GAUSSIANS = 4096;
// the per-pair screen: any number of chains (a late rejection pass over a few hundred costs 60-160 us unscreened -- every
// one of them against every Gaussian, on a fraction of the chip -- and has 20-30 of them per job), sorted only where there
// are enough of them for neighbours to be near
constexpr int SCREEN_MIN_CHAINS = 1, SCREEN_SORT_MIN_CHAINS = 2048;

template <int DMAX>
static int cull_prepare(mcx_engine *e, const float *xrows, const int *ain, int nact, bool sums, int own0, hipStream_t st)
{
  const int d = e->nparam, N = e->tchains;
  const int ng = (nact + CULL_W - 1) / CULL_W, nw = (N + 63) / 64;
  // (the sums and the histogram are zero here: zeroed when allocated, and again by every k_cull_boxes)
  hipLaunchKernelGGL(k_cull_stats, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, d, e->cull_stats.p);
  hipLaunchKernelGGL(k_cull_keys, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, d, e->cull_stats.p,
                     e->cull_keys.p, e->cull_hist.p);
  hipLaunchKernelGGL(k_cull_scan, dim3(1), dim3(1024), 0, st, e->cull_hist.p);
  hipLaunchKernelGGL(k_cull_scatter, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, ain, e->cull_keys.p, nact, e->cull_hist.p,
                     e->cull_sorted.p);
  const dim3 gb((unsigned)((ng + BLOCK / 64 - 1) / (BLOCK / 64)));
  if (sums)
    hipLaunchKernelGGL((k_cull_boxes<DMAX, true>), gb, dim3(BLOCK), 0, st, xrows, (const int *)e->cull_sorted.p, nact,
                       (const float *)e->winvall.p, own0, e->cull_box.p, e->cull_lim.p, e->cull_stats.p, e->cull_hist.p);
  else
    hipLaunchKernelGGL((k_cull_boxes<DMAX, false>), gb, dim3(BLOCK), 0, st, xrows, (const int *)e->cull_sorted.p, nact,
                       (const float *)e->winvall.p, own0, e->cull_box.p, e->cull_lim.p, e->cull_stats.p, e->cull_hist.p);
  const int gchunk = 64;  // (every chunk re-reads the Gaussians' key dimensions: 74 us per 65 536 x 65 536 test with 16, 35 with 64)
  hipLaunchKernelGGL((k_cull_test<DMAX>), dim3((unsigned)((nw + BLOCK / 64 - 1) / (BLOCK / 64)), (unsigned)((ng + gchunk - 1) / gchunk)),
                     dim3(BLOCK), 0, st, (const float *)e->winvall.p, N, (const float *)e->cull_box.p, (const float *)e->cull_lim.p, ng, nact,
                     gchunk, e->cull_excl.p, nw, reinterpret_cast<unsigned long long *>(e->nact.p) + 1 + (sums ? CULL_NCOUNT : 0));
  HIPCHK(hipGetLastError());
  e->cnt.kernel_launches += 6;
  return MCX_OK;
}

// The same with the chains sorted along ONE direction and every (group, Q_i) row bounded by Cauchy-Schwarz along it
// (mcx_cull_proj.hpp): for chain clouds no box of a few coordinates separates.
template <int DMAX>
static int cull_prepare_proj(mcx_engine *e, const float *xrows, const int *ain, int nact, bool sums, int own0, hipStream_t st)
{
  const int N = e->tchains;
  const int ng = (nact + CULL_W - 1) / CULL_W, nw = (N + 63) / 64;
  double *acc0 = e->proj_acc.p, *acc1 = e->proj_acc.p + PROJ_ACC;
  // (acc0 and the histogram are zero here: zeroed when allocated, and again by every k_proj_groups; acc1 is zeroed now)
  HIPCHK(hipMemsetAsync(acc1, 0, PROJ_ACC * sizeof(double), st));
  hipLaunchKernelGGL((k_proj_moments<DMAX>), dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, (const double *)acc0, 0, acc0);
  hipLaunchKernelGGL((k_proj_moments<DMAX>), dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, (const double *)acc0, 1, acc1);
  hipLaunchKernelGGL((k_proj_keys<DMAX>), dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, (const double *)acc1,
                     e->proj_p.p, e->cull_keys.p, e->cull_hist.p);
  hipLaunchKernelGGL(k_cull_scan, dim3(1), dim3(1024), 0, st, e->cull_hist.p);
  hipLaunchKernelGGL(k_cull_scatter, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, ain, e->cull_keys.p, nact, e->cull_hist.p,
                     e->cull_sorted.p);
  const dim3 gb((unsigned)((ng + BLOCK / 64 - 1) / (BLOCK / 64)));
  if (sums)
    hipLaunchKernelGGL((k_proj_groups<DMAX, true>), gb, dim3(BLOCK), 0, st, xrows, (const int *)e->cull_sorted.p, nact,
                       (const float *)e->winvall.p, own0, (const double *)e->proj_p.p, e->proj_lohi.p, e->cull_lim.p, acc0, e->cull_hist.p);
  else
    hipLaunchKernelGGL((k_proj_groups<DMAX, false>), gb, dim3(BLOCK), 0, st, xrows, (const int *)e->cull_sorted.p, nact,
                       (const float *)e->winvall.p, own0, (const double *)e->proj_p.p, e->proj_lohi.p, e->cull_lim.p, acc0, e->cull_hist.p);
  const int gchunk = 64;
  hipLaunchKernelGGL((k_proj_test<DMAX>), dim3((unsigned)((nw + BLOCK / 64 - 1) / (BLOCK / 64)), (unsigned)((ng + gchunk - 1) / gchunk)),
                     dim3(BLOCK), 0, st, (const float *)e->winvall.p, N, (const double *)acc1, (const double *)e->proj_lohi.p,
                     (const float *)e->cull_lim.p, ng, nact, gchunk, e->cull_excl.p, nw,
                     reinterpret_cast<unsigned long long *>(e->nact.p) + 1 + (sums ? CULL_NCOUNT : 0));
  HIPCHK(hipGetLastError());
  e->cnt.kernel_launches += 8;
  return MCX_OK;
}

// The same sort, then the per-pair bound on the matrix cores (mcx_screen.hpp) in place of the boxes.  `fresh_q`: the
// Gaussians' side (centre, B') has not been built for this genRemote call yet.  *order_out: the list the masks' groups
// are cut from (the sorted list, or `ain` itself -- null = every chain in index order -- where nothing was sorted).
// the screen's matrix-core kernel as screen_prepare would launch it, kept for launching by column chunks
struct ScreenLaunch {
  int gx = 0, bchunk = 1, nblk = 0, nact = 0, N = 0, ng = 0, nw = 0;
  unsigned long long *nkept = nullptr;
};

template <int DMAX>
static void screen_gemm_go(mcx_engine *e, const ScreenLaunch &L, int blk0, int blk1, hipStream_t st)
{
  hipLaunchKernelGGL((k_screen_gemm<DMAX>), dim3((unsigned)L.gx, (unsigned)((blk1 - blk0 + L.bchunk - 1) / L.bchunk)), dim3(SCR_WAVES * 64), 0, st,
                     (const unsigned short *)e->scr_a.p, (const unsigned short *)e->scr_b.p, L.nact, L.N, L.ng, L.bchunk, e->cull_excl.p, L.nw,
                     L.nkept, blk0, blk1);
}

template <int DMAX>
static int screen_prepare(mcx_engine *e, const float *xrows, const int *ain, int nact, bool sums, int own0, bool *fresh_q,
                          const int **order_out, hipStream_t st, bool may_sort = true, ScreenLaunch *defer = nullptr, int chunks = 1)
{
  const int d = e->nparam, N = e->tchains;
  const int ng = (nact + CULL_W - 1) / CULL_W, nw = (N + 63) / 64, nblk = (N + SCR_BLK - 1) / SCR_BLK;
  if (*fresh_q) {
    hipLaunchKernelGGL(k_screen_centre, dim3(1), dim3(1024), 0, st, (const float *)e->winvall.p, N, d, e->scr_centre.p);
    hipLaunchKernelGGL((k_screen_prep_q<DMAX>), dim3(nblocks((size_t)nblk * SCR_BLK)), dim3(BLOCK), 0, st, (const float *)e->winvall.p, N,
                       nblk * SCR_BLK, (const float *)e->scr_centre.p, e->scr_b.p);
    e->cnt.kernel_launches += 2;
    *fresh_q = false;
  }
  const int *order = ain;  // (few chains: in the order they come)
  if (may_sort && nact >= SCREEN_SORT_MIN_CHAINS) {
    hipLaunchKernelGGL(k_cull_stats, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, d, e->cull_stats.p);
    hipLaunchKernelGGL(k_cull_keys, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, xrows, ain, nact, d, e->cull_stats.p,
                       e->cull_keys.p, e->cull_hist.p);
    hipLaunchKernelGGL(k_cull_scan, dim3(1), dim3(1024), 0, st, e->cull_hist.p);
    hipLaunchKernelGGL(k_cull_scatter, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, ain, e->cull_keys.p, nact, e->cull_hist.p,
                       e->cull_sorted.p);
    order = e->cull_sorted.p;
    e->cnt.kernel_launches += 4;
  }
  *order_out = order;
  const dim3 gp(nblocks((size_t)ng * CULL_W));
  if (sums)
    hipLaunchKernelGGL((k_screen_prep_x<DMAX, true>), gp, dim3(BLOCK), 0, st, xrows, order, nact, ng * CULL_W,
                       (const float *)e->winvall.p, own0, (const float *)e->scr_centre.p, e->scr_a.p, e->cull_stats.p, e->cull_hist.p);
  else
    hipLaunchKernelGGL((k_screen_prep_x<DMAX, false>), gp, dim3(BLOCK), 0, st, xrows, order, nact, ng * CULL_W,
                       (const float *)e->winvall.p, own0, (const float *)e->scr_centre.p, e->scr_a.p, e->cull_stats.p, e->cull_hist.p);
  // The Gaussians' blocks are cut into chunks so that the grid is two chipfuls of workgroups (the kernel's own occupancy
  // x the CUs; one to four chipfuls, or 4096 workgroups whatever the chip holds: the same time within 2 %)
  constexpr int rounds = 2;
  static std::atomic<int> blocks_per_cu{0};  // (per instantiation; engines on several threads may ask at once: the same answer)
  int nb = blocks_per_cu.load(std::memory_order_relaxed);
  if (!nb) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_screen_gemm<DMAX>, SCR_WAVES * 64, 0) != hipSuccess || nb < 1) nb = 2;
    blocks_per_cu.store(nb, std::memory_order_relaxed);
  }
  const int resident = nb * (e->ncu > 0 ? e->ncu : 256);
  const int gx = (ng + SCR_WAVES - 1) / SCR_WAVES;
  const int per_launch = (nblk + chunks - 1) / chunks;  // (by column chunks: every chunk's grid is sized like a whole screen's)
  const int gy_want = std::max(1, (rounds * resident) / gx);
  int bchunk = (per_launch + gy_want - 1) / gy_want;
  if (bchunk < 1) bchunk = 1;
  ScreenLaunch L;
  L.gx = gx; L.bchunk = bchunk; L.nblk = nblk; L.nact = nact; L.N = N; L.ng = ng; L.nw = nw;
  L.nkept = reinterpret_cast<unsigned long long *>(e->nact.p) + 1 + (sums ? CULL_NCOUNT : 0);
  if (defer) {
    *defer = L;
    HIPCHK(hipGetLastError());
    return MCX_OK;
  }
  {
    ProfScope sg(e, MCX_K_REMOTE_SCREEN, (uint64_t)nact * (uint64_t)N);
    screen_gemm_go<DMAX>(e, L, 0, nblk, st);
  }
  HIPCHK(hipGetLastError());
  e->cnt.kernel_launches += 1;  // (+1: the screen's own scope)
  return MCX_OK;
}

// A pass over many chains with the per-pair screen, its Gaussians cut into `chunks` column chunks: chunk c + 1 is screened
// on the step stream (matrix cores) while chunk c is swept on a side stream (vector units) -- the two kernels alternated
// before, each leaving the other's pipe idle (screen 36 % + sweep 36 % of C3-murray's job).  Nothing about
// the sums changes: block partials are per block of QBLOCK Gaussians and combined in index order as ever.  Returns false
// (nothing launched but the preparation) where the pass does not cut evenly or is not the row-by-row kernel's.
constexpr int OVERLAP_MIN_CHAINS = 8192;
template <int DM, bool SUMS>
static int screen_sweep_chunked(mcx_engine *e, const float *xrows, const int *ain, int nact, int own0, bool *fresh_q, const int **order_out,
                                float *psum, float *pmax, int S, int chunks, int *entries_out, bool *done, hipStream_t st)
{
  *done = false;
  if constexpr (DM == 16 || DM == 32) {
    const int N = e->tchains, ng = (nact + CULL_W - 1) / CULL_W;
    int bpw = 1, gy = S;
    if (!sweep_is_srow<DM, SUMS>(nact, e->cull_excl.p, ng, S, &bpw, &gy)) return MCX_OK;
    const int nblk = (N + SCR_BLK - 1) / SCR_BLK;
    // a chunk = gy / chunks rows of the sweep's grid = bpw * QBLOCK Gaussians each = whole blocks of the screen
    static_assert(QBLOCK % SCR_BLK == 0, "a sweep block is a whole number of screen blocks");
    while (chunks > 1 && (gy % chunks != 0 || N % QBLOCK != 0)) chunks >>= 1;
    if (chunks < 2) return MCX_OK;
    const int ny = gy / chunks, blk_per = ny * bpw * (QBLOCK / SCR_BLK);
    ScreenLaunch L;
    MCXCHK((screen_prepare<DM>(e, xrows, ain, nact, SUMS, own0, fresh_q, order_out, st, true, &L, chunks)));
    if (!e->mstream) HIPCHK(hipStreamCreateWithFlags(&e->mstream, hipStreamNonBlocking));
    while ((int)e->mev.size() < chunks + 1) {
      hipEvent_t ev = nullptr;
      HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      e->mev.push_back(ev);
    }
    {
      // (profile: the phase as a whole counts as the sweep -- the screen has no time of its own any more)
      ProfScope sw(e, MCX_K_REMOTE_SWEEP, (uint64_t)nact * (uint64_t)N);
      for (int c = 0; c < chunks; ++c) {
        const int b0 = c * blk_per, b1 = std::min(nblk, b0 + blk_per);
        screen_gemm_go<DM>(e, L, b0, b1, st);
        HIPCHK(hipEventRecord(e->mev[c], st));
        HIPCHK(hipStreamWaitEvent(e->mstream, e->mev[c], 0));
        launch_srow_range<DM, SUMS>(xrows, *order_out, nact, e->winvall.p, psum, pmax, N, own0, e->cull_excl.p, ng, bpw, c * ny, ny, e->mstream);
      }
      HIPCHK(hipEventRecord(e->mev[chunks], e->mstream));
      HIPCHK(hipStreamWaitEvent(st, e->mev[chunks], 0));
    }
    HIPCHK(hipGetLastError());
    e->cnt.kernel_launches += 2 * chunks - 1;
    *entries_out = SUMS ? S : gy;
    *done = true;
  }
  return MCX_OK;
}

// MCPar::genRemote on device buffers (src/mcpar.cc:315-451)
int remote_device(mcx_engine *e, uint32_t t, const float *pvals, const float *musigall,
                         float *ptrial, float *cfac, float *mutrial, float *sigtrial, int *npass_out)
{
  const int n = e->nchain, d = e->nparam, N = e->tchains, dm = d <= 64 ? dmax_for(d) : 64;
  const bool big = d > 64;  // chain vector in registers up to np = 64, re-read from memory above
  const int S = (N + QBLOCK - 1) / QBLOCK;
  if (!big && S > 65535)  // blocks of Gaussians go in gridDim.y
    return fail(MCX_ERR_UNSUPPORTED, "Murray proposals over %d chains in all: at most %d", N, 65535 * QBLOCK);
  hipStream_t st = e->stream;
  ProfScope ps(e, MCX_K_REMOTE, (uint64_t)n);
  const size_t rows_max = std::max<size_t>((size_t)n, (size_t)MULTI_MAX_CHAINS * MULTI_K);  // chains, or candidates of few chains
  if (!big) {
    MCXCHK(e->psum.alloc(rows_max * S));
    MCXCHK(e->pmax.alloc(rows_max * S));
    MCXCHK(e->racpt.alloc((size_t)n));
    MCXCHK(e->cand.alloc((size_t)MULTI_MAX_CHAINS * MULTI_K * (3 * (size_t)d + 1)));
  }
  // exclusion of far Gaussians: the two-chains-per-lane sweeps (np = 16, 32) over enough chains and Gaussians to
  // pay for the sort and the tests (or whenever possible: MCX_OPT_CULL = 1, for the tests)
  const bool cull_can = !big && d == dm && SWEEP_CPL(dm) == 2 && e->opt_cull != 0;
  const bool gemm = e->opt_cull == 3 || e->opt_cull < 0;
  auto cull_now = [&](int na) {
    return cull_can && (e->opt_cull > 0 || (na >= (gemm ? SCREEN_MIN_CHAINS : CULL_MIN_CHAINS) && N >= CULL_MIN_GAUSSIANS));
  };
  if (cull_can) {
    const size_t ngmax = (rows_max + CULL_W - 1) / CULL_W, nw = ((size_t)N + 63) / 64;
    const bool fresh = !e->cull_hist.p || !e->cull_stats.p;
    MCXCHK(e->cull_keys.alloc(rows_max)); MCXCHK(e->cull_hist.alloc(CULL_BINS)); MCXCHK(e->cull_sorted.alloc(rows_max));
    MCXCHK(e->cull_stats.alloc(2 * CULL_KD));
    if (fresh) {  // (k_cull_boxes leaves them zero for the next sort)
      HIPCHK(hipMemsetAsync(e->cull_stats.p, 0, 2 * CULL_KD * sizeof(float), st));
      HIPCHK(hipMemsetAsync(e->cull_hist.p, 0, CULL_BINS * sizeof(unsigned), st));
    } MCXCHK(e->cull_box.alloc(ngmax * 2 * CULL_KD)); MCXCHK(e->cull_lim.alloc(ngmax));
    MCXCHK(e->cull_excl.alloc(ngmax * nw));
    const bool fresh_p = !e->proj_acc.p;
    MCXCHK(e->proj_acc.alloc(2 * PROJ_ACC)); MCXCHK(e->proj_p.alloc(rows_max)); MCXCHK(e->proj_lohi.alloc(2 * ngmax));
    if (fresh_p) HIPCHK(hipMemsetAsync(e->proj_acc.p, 0, 2 * PROJ_ACC * sizeof(double), st));
    MCXCHK(e->scr_centre.alloc(64)); MCXCHK(e->scr_a.alloc(ngmax * CULL_W * scr_k(dm)));
    MCXCHK(e->scr_b.alloc((((size_t)N + SCR_BLK - 1) / SCR_BLK) * SCR_BLK * scr_k(dm)));
  }
  // which exact screen: boxes of four coordinates, or -- on request only -- one direction (mcx_cull_proj.hpp).  Measured
  // in round 4: on C3's shape the direction keeps 0.58 of the pairs where the boxes keep 0.41; on C5's mixture it keeps
  // 0.999 like the boxes -- there a pair is dead because the chain is far from the Gaussian in the 31 directions ACROSS
  // the mixture's axis (the per-chain Gaussians are still narrow), which no bound for 128 chains at once can see.
  const bool proj = e->opt_cull == 2;
  const int overlap = e->opt_murray_overlap;  // column chunks of a big pass: the next chunk's screen beside this chunk's sweep (0 / 1: off)
  bool fresh_q = true;
  // (the screens' cells and, right behind them, the word k_remote_decide counts its workgroups in)
  HIPCHK(hipMemsetAsync(e->nact.p + 2, 0, 2 * CULL_NCOUNT * sizeof(unsigned long long) + 2 * sizeof(int), st));
  uint64_t evaluated_host = 0;  // pairs of the sweeps that ran without an exclusion test
  // auto mode gives the test up where it excludes too little to pay for itself (the 32-D mixture: per-chain
  // Gaussians too broad for any 128-chain box), per kind of sweep, and tries again every eighth call
  constexpr double CULL_USELESS = 0.85;
  bool cull_min = true, cull_sums = true;
  if (e->opt_cull < 0) {
    if (e->cull_skip[0] > 0) { cull_min = false; e->cull_skip[0]--; }
    if (e->cull_skip[1] > 0) { cull_sums = false; e->cull_skip[1]--; }
  }
  uint64_t tested_min = 0, tested_sums = 0;
  hipLaunchKernelGGL(k_remote_prep, dim3(nblocks((size_t)N * d)), dim3(BLOCK), 0, st, musigall,
                     e->winvall.p, (size_t)N * d);
  const int own0 = e->rank * e->nchain;
  if (big) {
    hipLaunchKernelGGL(k_remote_cmax_big, dim3(nblocks((size_t)n)), dim3(BLOCK), 0, st, pvals, e->winvall.p,
                       e->cmax.p, n, d, N);
    evaluated_host += (uint64_t)n * (uint64_t)N;
  } else {
    const bool cull = cull_now(n) && cull_min;
    if (cull) tested_min = (uint64_t)n * (uint64_t)N;
    const int *order = nullptr;
    const unsigned long long *excl = nullptr;
    int S_min = S;  // entries per chain the min-arg sweep leaves for the combining kernel
    bool swept = false;  // the screen and the sweep went out together, by column chunks
    if (cull) {
      order = e->cull_sorted.p;
      if (gemm && overlap > 1 && n >= OVERLAP_MIN_CHAINS && d == dm) {
        DISPATCH_DMAX(dm, MCXCHK((screen_sweep_chunked<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16, false>(
                              e, pvals, nullptr, n, own0, &fresh_q, &order, (float *)nullptr, e->pmax.p, S, overlap, &S_min, &swept, st))));
      }
      if (swept) {
      } else if (gemm) {  // (boxes here and the per-pair bound for the sums only: C3-murray 18.1 -> 25.6 ms, C5 26.0 -> 38.8)
        DISPATCH_DMAX(dm, MCXCHK((screen_prepare<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, pvals, nullptr, n, false, own0, &fresh_q, &order, st))));
      } else if (proj) {
        DISPATCH_DMAX(dm, MCXCHK((cull_prepare_proj<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, pvals, nullptr, n, false, own0, st))));
      } else {
        DISPATCH_DMAX(dm, MCXCHK((cull_prepare<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, pvals, nullptr, n, false, own0, st))));
      }
      excl = e->cull_excl.p;
    } else {
      evaluated_host += (uint64_t)n * (uint64_t)N;
    }
    if (!swept) {
      ProfScope sw(e, MCX_K_REMOTE_SWEEP, (uint64_t)n * (uint64_t)N);
      if (d == dm) {
        DISPATCH_DMAX(dm, (S_min = launch_sweep_exact<DMAX_, false>(pvals, order, n, e->winvall.p, (float *)nullptr, e->pmax.p,
                                                                   N, own0, excl, (n + CULL_W - 1) / CULL_W, S, st)));
      } else {
        DISPATCH_DMAX(dm, hipLaunchKernelGGL((k_remote_sweep<DMAX_, false, false>), dim3(nblocks((size_t)n), S), dim3(BLOCK),
                                             0, st, pvals, (const int *)nullptr, n, e->winvall.p,
                                             (float *)nullptr, e->pmax.p, d, N, own0, (const unsigned long long *)nullptr, 0));
      }
    }
    hipLaunchKernelGGL(k_remote_cmax_combine, dim3(nblocks((size_t)n)), dim3(BLOCK), 0, st, e->pmax.p, e->cmax.p, n, S_min, order);
  }
  HIPCHK(hipGetLastError());
  e->cnt.remote_pairs += (uint64_t)n * (uint64_t)N;
  int nact = n, pass = 0;
  int *ain = nullptr, *aout = e->active0.p;
  unsigned long long kept_min = 0, kept_sums = 0;
  constexpr int NCOUNTS_ALL = 1 + 2 * CULL_NCOUNT + 1 + MULTI_K / 2;  // survivors, the screens' cells, `done`, tried[MULTI_K]
  if (!e->h_nact.p) {  // pinned and mapped: k_remote_decide writes the pass's counters and serial there
    MCXCHK(e->h_nact.alloc(NCOUNTS_ALL + 1));
    memset(e->h_nact.p, 0, (NCOUNTS_ALL + 1) * sizeof(unsigned long long));
  }
  int it = 0;  // kernels' turns (a turn over candidates stands for several passes)
  while (nact > 0) {
    const bool multi = !big && pass > 0 && nact <= MULTI_MAX_CHAINS;
    // two survivor counters in turn: a pass counts in one and zeroes the other for the next pass (the first pass's
    // draw zeroes its own) -- no fill between the passes
    int *const cnt_here = e->nact.p + (it & 1);
    if (big) HIPCHK(hipMemsetAsync(cnt_here, 0, sizeof(int), st));
    RemoteArgs a;
    a.active_in = ain; a.nact = nact; a.active_out = aout; a.nact_out = cnt_here;
    a.nact_zero = big ? nullptr : e->nact.p + ((it + 1) & 1);
    a.musigall = musigall; a.winv = e->winvall.p; a.cmax = e->cmax.p;
    a.ptrial = ptrial; a.mutrial = mutrial; a.sigtrial = sigtrial; a.cfac = cfac;
    a.racpt = e->racpt.p; a.psum = e->psum.p; a.pmax = e->pmax.p;
    a.n = n; a.d = d; a.N = N; a.pass = pass; a.S = S;
    a.g0 = (uint32_t)(e->rank * e->nchain); a.t = t; a.seed = e->seed;
    a.counts = reinterpret_cast<const unsigned long long *>(e->nact.p);
    a.counts_host = big ? nullptr : e->h_nact.p;  // (np > 64: no k_remote_decide; the copy below)
    a.done = reinterpret_cast<unsigned *>(e->nact.p) + 2 * (1 + 2 * NACT_CULL_CELLS);
    a.ncounts = multi ? NCOUNTS_ALL : (cull_can ? 1 + 2 * CULL_NCOUNT : 1);
    a.nflag = NCOUNTS_ALL;  // (a word of its own whatever ncounts is: serial numbers only ever grow there)
    a.serial = ++e->remote_serial;
    a.ncand = MULTI_K;
    a.cand_p = e->cand.p;
    a.cand_mu = a.cand_p + (size_t)MULTI_MAX_CHAINS * MULTI_K * d;
    a.cand_sig = a.cand_mu + (size_t)MULTI_MAX_CHAINS * MULTI_K * d;
    a.cand_racpt = a.cand_sig + (size_t)MULTI_MAX_CHAINS * MULTI_K * d;
    a.tried = e->nact.p + 2 * (1 + 2 * NACT_CULL_CELLS) + 2;
    if (big) {
      hipLaunchKernelGGL(k_remote_pass_big, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, a);
      evaluated_host += (uint64_t)nact * (uint64_t)N;
    } else if (multi) {
      const int nv = nact * MULTI_K;
      hipLaunchKernelGGL(k_remote_draw_multi, dim3(nblocks((size_t)nv)), dim3(BLOCK), 0, st, a);
      const bool cull = gemm && cull_now(nv) && cull_sums;  // (the other screens sort, and the candidates keep their rows)
      const int *list = nullptr;
      const unsigned long long *excl = nullptr;
      if (cull) {
        DISPATCH_DMAX(dm, MCXCHK((screen_prepare<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, a.cand_p, nullptr, nv, true, -1, &fresh_q, &list, st, false))));
        excl = e->cull_excl.p;
      } else {
        evaluated_host += (uint64_t)nv * (uint64_t)N;
      }
      {
        ProfScope sw(e, MCX_K_REMOTE_SWEEP, (uint64_t)nv * (uint64_t)N);
        if (d == dm) {
          DISPATCH_DMAX(dm, (launch_sweep_exact<DMAX_, true>(a.cand_p, list, nv, e->winvall.p, e->psum.p, e->pmax.p, N,
                                                            -1, excl, (nv + CULL_W - 1) / CULL_W, S, st)));
        } else {
          DISPATCH_DMAX(dm, hipLaunchKernelGGL((k_remote_sweep<DMAX_, true, false>), dim3(nblocks((size_t)nv), S), dim3(BLOCK),
                                               0, st, (const float *)a.cand_p, (const int *)nullptr, nv, e->winvall.p,
                                               e->psum.p, e->pmax.p, d, N, -1, (const unsigned long long *)nullptr, 0));
        }
      }
      static_assert(MULTI_K == 4, "k_remote_decide_multi: a chain's candidates are a quad of lanes");
      hipLaunchKernelGGL(k_remote_decide_multi, dim3(nblocks((size_t)nv)), dim3(BLOCK), 0, st, a);
    } else {
      // (later passes: the k_remote_decide that rejected a chain has drawn its next proposal already)
      if (pass == 0) DISPATCH_DMAX(dm, hipLaunchKernelGGL((k_remote_draw<DMAX_>), dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, a));
      const bool cull = cull_now(nact) && cull_sums;
      const int *list = ain;
      const unsigned long long *excl = nullptr;
      bool swept = false;
      if (cull) {  // the proposals have just been drawn: sort, box and test them
        list = e->cull_sorted.p;
        if (gemm && overlap > 1 && nact >= OVERLAP_MIN_CHAINS && d == dm) {
          int entries = S;
          DISPATCH_DMAX(dm, MCXCHK((screen_sweep_chunked<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16, true>(
                                e, ptrial, ain, nact, -1, &fresh_q, &list, e->psum.p, e->pmax.p, S, overlap, &entries, &swept, st))));
        }
        if (swept) {
        } else if (gemm) {
          DISPATCH_DMAX(dm, MCXCHK((screen_prepare<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, ptrial, ain, nact, true, -1, &fresh_q, &list, st))));
        } else if (proj) {
          DISPATCH_DMAX(dm, MCXCHK((cull_prepare_proj<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, ptrial, ain, nact, true, -1, st))));
        } else {
          DISPATCH_DMAX(dm, MCXCHK((cull_prepare<(DMAX_ == 16 || DMAX_ == 32) ? DMAX_ : 16>(e, ptrial, ain, nact, true, -1, st))));
        }
        excl = e->cull_excl.p;
        a.active_in = list;  // positions of psum / pmax are positions of the sorted list
      } else {
        evaluated_host += (uint64_t)nact * (uint64_t)N;
      }
      if (!swept) {
      ProfScope sw(e, MCX_K_REMOTE_SWEEP, (uint64_t)nact * (uint64_t)N);
      if (d == dm) {
        DISPATCH_DMAX(dm, (launch_sweep_exact<DMAX_, true>(ptrial, list, nact, e->winvall.p, e->psum.p, e->pmax.p, N,
                                                          -1, excl, (nact + CULL_W - 1) / CULL_W, S, st)));
      } else {
        DISPATCH_DMAX(dm, hipLaunchKernelGGL((k_remote_sweep<DMAX_, true, false>), dim3(nblocks((size_t)nact), S), dim3(BLOCK),
                                             0, st, ptrial, (const int *)ain, nact, e->winvall.p,
                                             e->psum.p, e->pmax.p, d, N, -1, (const unsigned long long *)nullptr, 0));
      }
      }
      hipLaunchKernelGGL(k_remote_decide, dim3(nblocks((size_t)nact)), dim3(BLOCK), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    if (!multi) e->cnt.remote_pairs += (uint64_t)nact * (uint64_t)N;  // (candidates: what the passes would have swept, below)
    unsigned long long *back = e->h_nact.p;  // survivors (low word), cells of the pairs kept by the min-arg / sum tests so far
    if (big) {
      HIPCHK(hipMemcpyAsync(back, e->nact.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    } else {
      // k_remote_decide's last workgroup writes the counters, then the pass's serial number: spinning on that word
      // instead of a copy kernel and a wait for the stream (once per pass, 30-50 passes per job; against the copy and
      // the wait on one box: C3-murray 16.2 -> 16.1 ms, C5 23.6 -> 23.4: the counters' way over PCIe is 6-11 us of the
      // kernel now, what the copy kernel and its launch were) -- with a look at the stream now and then, so that a
      // launch that failed or a device that is gone ends the wait
      // The busy wait is BOUNDED: a short pass answers within tens of microseconds and is worth a core's
      // attention; past SPIN_US the thread yields between looks (8 ranks + their MCout formatting threads + MPI progress
      // on 16 cores: a spinning waiter must not hold a core from the work it waits for), and past YIELD_US it sleeps in
      // hipStreamSynchronize like any other wait.
      constexpr long SPIN_US = 60, YIELD_US = 2000;
      volatile unsigned long long *const flag = back + a.nflag;
      const auto t_wait = std::chrono::steady_clock::now();
      bool polite = false;
      for (unsigned spins = 0; __atomic_load_n(const_cast<unsigned long long *>(flag), __ATOMIC_ACQUIRE) != a.serial; ++spins) {
        if (!polite) {
#if defined(__x86_64__) || defined(__i386__)
          __builtin_ia32_pause();  // (the sibling hyperthread gets the core's issue slots)
#endif
        } else {
          sched_yield();
        }
        if ((spins & 0xffu) == 0xffu || polite) {
          const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wait).count();
          if (us > YIELD_US) break;  // (the blocking wait below)
          polite = us > SPIN_US;
        }
        if ((spins & 0xfffu) == 0xfffu) {
          const hipError_t q = hipStreamQuery(st);
          if (q == hipSuccess) break;  // the stream is empty: the word is there (or never will be: checked below)
          if (q != hipErrorNotReady) return fail(MCX_ERR_HIP, "Murray pass: %s", hipGetErrorString(q));
        }
      }
      if (__atomic_load_n(const_cast<unsigned long long *>(flag), __ATOMIC_ACQUIRE) != a.serial) {
        HIPCHK(hipStreamSynchronize(st));
        if (__atomic_load_n(const_cast<unsigned long long *>(flag), __ATOMIC_ACQUIRE) != a.serial)
          return fail(MCX_ERR_HIP, "Murray pass: the pass's counters did not arrive");
      }
    }
    const unsigned long long before = kept_sums;
    const uint64_t pairs_now = (uint64_t)nact * (uint64_t)N * (multi ? MULTI_K : 1);
    nact = (int)(unsigned)((it & 1) ? back[0] >> 32 : back[0] & 0xffffffffull);
    int passes_now = 1;
    if (multi) {
      // tried[c] chains came as far as candidate c: pass + c would have swept them.  Survivors tried them all.
      const unsigned *tried = reinterpret_cast<const unsigned *>(back + 1 + 2 * CULL_NCOUNT + 1);
      passes_now = 0;
      for (int c = 0; c < MULTI_K; ++c)
        if (tried[c]) {
          passes_now = c + 1;
          e->cnt.remote_pairs += (uint64_t)tried[c] * (uint64_t)N;
        }
      if (nact > 0) passes_now = MULTI_K;
    }
    if (cull_can) {
      kept_min = kept_sums = 0;
      for (int c = 0; c < CULL_NCOUNT; ++c) { kept_min += back[1 + c]; kept_sums += back[1 + CULL_NCOUNT + c]; }
    }
    if (pairs_now && kept_sums > before) {  // this pass was tested
      tested_sums += pairs_now;
      if (e->opt_cull < 0 && (double)(kept_sums - before) > CULL_USELESS * (double)pairs_now) cull_sums = false;
    }
    ain = aout;
    aout = (aout == e->active0.p) ? e->active1.p : e->active0.p;
    e->cnt.kernel_launches += (big || pass > 0) ? 1 : 2;  // (+1: the sweep's own scope)
    pass += passes_now;
    ++it;
  }
  e->cnt.remote_pairs_evaluated += evaluated_host + (uint64_t)kept_min + (uint64_t)kept_sums;
  if (e->opt_cull < 0) {
    if (tested_min && (double)kept_min > CULL_USELESS * (double)tested_min) e->cull_skip[0] = 7;
    if (tested_sums && (double)kept_sums > CULL_USELESS * (double)tested_sums) e->cull_skip[1] = 7;
  }
  hipLaunchKernelGGL(k_square, dim3(nblocks((size_t)e->ntot)), dim3(BLOCK), 0, st, sigtrial, (size_t)e->ntot);
  HIPCHK(hipGetLastError());
  if (npass_out) *npass_out = pass;
  return MCX_OK;
}


// The per-pair screen alone, for tests: masks[word][group] (mcx_screen.hpp's layout; group = 128 consecutive chains, in
// the order given) of nact chains against N Gaussians given as (mu, sig2) pairs.
extern "C" int mcx_debug_murray_screen(int d, int nact, int N, const float *x, const float *musig, int own0, int sums,
                                       unsigned long long *masks)
{
  if ((d != 16 && d != 32) || nact < 1 || N < 1 || !x || !musig || !masks) return fail(MCX_ERR_INVALID, "mcx_debug_murray_screen: np = 16 or 32");
  if (!sums && (own0 < 0 || own0 + nact > N)) return fail(MCX_ERR_INVALID, "mcx_debug_murray_screen: own Gaussians outside the N");
  const int ng = (nact + CULL_W - 1) / CULL_W, nw = (N + 63) / 64, nblk = (N + SCR_BLK - 1) / SCR_BLK, K = scr_k(d);
  DevBuf<float> dx, dms, q, centre, stats;
  DevBuf<unsigned> hist;
  DevBuf<unsigned short> A, B;
  DevBuf<unsigned long long> excl, kept;
  MCXCHK(dx.alloc((size_t)nact * d)); MCXCHK(dms.alloc((size_t)N * d * 2)); MCXCHK(q.alloc((size_t)N * d * 2));
  MCXCHK(centre.alloc(64)); MCXCHK(stats.alloc(2 * CULL_KD)); MCXCHK(hist.alloc(CULL_BINS));
  MCXCHK(A.alloc((size_t)ng * CULL_W * K)); MCXCHK(B.alloc((size_t)nblk * SCR_BLK * K));
  MCXCHK(excl.alloc((size_t)ng * nw)); MCXCHK(kept.alloc(CULL_NCOUNT));
  int rc = MCX_OK;
  auto body = [&]() -> int {
    HIPCHK(hipMemcpy(dx.p, x, (size_t)nact * d * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dms.p, musig, (size_t)N * d * 2 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(kept.p, 0, CULL_NCOUNT * sizeof(unsigned long long)));
    HIPCHK(hipMemset(excl.p, 0, (size_t)ng * nw * sizeof(unsigned long long)));
    hipStream_t st = nullptr;
    hipLaunchKernelGGL(k_remote_prep, dim3(nblocks((size_t)N * d)), dim3(BLOCK), 0, st, (const float *)dms.p, q.p, (size_t)N * d);
    hipLaunchKernelGGL(k_screen_centre, dim3(1), dim3(1024), 0, st, (const float *)q.p, N, d, centre.p);
    const dim3 gq(nblocks((size_t)nblk * SCR_BLK)), gp(nblocks((size_t)ng * CULL_W));
    const int gx = (ng + SCR_WAVES - 1) / SCR_WAVES;
    const dim3 gg((unsigned)gx, (unsigned)nblk);
#define SCREEN_FOR(DM)                                                                                                          \
    hipLaunchKernelGGL((k_screen_prep_q<DM>), gq, dim3(BLOCK), 0, st, (const float *)q.p, N, nblk * SCR_BLK, (const float *)centre.p, B.p); \
    if (sums)                                                                                                                   \
      hipLaunchKernelGGL((k_screen_prep_x<DM, true>), gp, dim3(BLOCK), 0, st, (const float *)dx.p, (const int *)nullptr, nact, ng * CULL_W, \
                         (const float *)q.p, -1, (const float *)centre.p, A.p, stats.p, hist.p);                                 \
    else                                                                                                                        \
      hipLaunchKernelGGL((k_screen_prep_x<DM, false>), gp, dim3(BLOCK), 0, st, (const float *)dx.p, (const int *)nullptr, nact, ng * CULL_W, \
                         (const float *)q.p, own0, (const float *)centre.p, A.p, stats.p, hist.p);                               \
    hipLaunchKernelGGL((k_screen_gemm<DM>), gg, dim3(SCR_WAVES * 64), 0, st, (const unsigned short *)A.p, (const unsigned short *)B.p, nact, N, \
                       ng, 1, excl.p, nw, kept.p, 0, nblk);
    if (d == 16) { SCREEN_FOR(16) } else { SCREEN_FOR(32) }
#undef SCREEN_FOR
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(masks, excl.p, (size_t)ng * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MCX_OK;
  };
  rc = body();
  dx.release(); dms.release(); q.release(); centre.release(); stats.release(); hist.release(); A.release(); B.release();
  excl.release(); kept.release();
  return rc;
}
