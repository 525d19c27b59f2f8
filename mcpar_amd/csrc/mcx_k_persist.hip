// mcx_k_persist.hip -- instantiations and launcher of k_run_small<LPC, LIK> (mcx_persist.hpp)
#include "mcx_launch.hpp"
#include "mcx_persist.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

using namespace mcx;

// Recorder wavefronts (one per owner: Welford, snapshot, sample emission one phase behind) pay where a workgroup holds ONE
// set of 64 lanes' worth of chains -- there the owner's dependent chain is the bound and everything taken out of it is
// time won (8-D x 4096: 0.29 ms with, 0.31 without).  From two sets on the workgroup is bound by the instructions its
// four SIMDs issue: recording in the owner's own loop saves the hand-over through LDS, frees the recorders' wavefronts
// and LDS for generating, and ends the launch one phase earlier (round 4, one box: 16-D x 8192 0.444 -> 0.422 ms,
// 16-D x 12288 0.730 -> 0.574, 16-D x 16384 0.767 -> 0.698, 32-D x 8192 0.779 -> 0.684).
bool mcxk_persist_recorders(int own, int bpl)
{
  static const char *env = getenv("MCX_PERSIST_REC");  // tuning only (tools/persist_config_sweep.py)
  if (env && *env) return atoi(env) != 0;
  return own * bpl <= 1;
}

static size_t lds_for(int lpc2, int bpl, int own, int rec, int K)
{
  return (size_t)2 * (1 + rec) * K * own * bpl * 64 * sizeof(float4) + (size_t)2 * (1 + rec) * K * own * (64 / lpc2) * sizeof(float);
}

// Blocks per lane (mcx_persist.hpp).  Measured (tools/persist_config_sweep.py, one box, 500 + 1000 steps): with two
// owner wavefronts per workgroup two blocks per lane change nothing (16-D x 8192: 0.422 ms as 2 owners x 1 block, 0.424
// as 1 owner x 2 blocks; 8-D x 16384: 0.415 / 0.433) -- a workgroup's time is the instructions its SIMDs issue for
// generating its chains' normals, and those do not care how the owners hold the chains; with four or more they save the
// per-chain share of the owners' loops: 16-D x 16384 0.732 -> 0.698, 32-D x 8192 0.714 -> 0.684, 16-D x 24576 1.153 -> 1.079.
// Never where halving the owner wavefronts would leave CUs without a workgroup (16-D x 12288: 768 owner wavefronts
// = 3 per CU; as 384 they fill 192 CUs: 0.574 -> 0.689).  `opt` = MCX_OPT_BLOCKS_PER_LANE.
int mcxk_persist_bpl(int lpc, int d, int n, int ncu, int opt)
{
  auto legal = [&](int bpl) { return bpl >= 1 && bpl <= lpc && d % (4 * bpl) == 0; };
  if (opt > 0) return legal(opt) && (opt == 1 || opt == 2 || opt == 4) ? opt : 1;
  if (!legal(2) || ncu < 1) return 1;
  const long long nown1 = ((long long)n * lpc + 63) / 64, nown2 = ((long long)n * (lpc / 2) + 63) / 64;
  const long long own1 = (nown1 + std::min<long long>(nown1, ncu) - 1) / std::min<long long>(nown1, ncu);
  return own1 >= 4 && nown2 >= ncu && nown2 % ncu == 0 ? 2 : 1;
}

bool mcxk_persist_deal_fits(int lpc2, int bpl, int own, int rec, int K);

// Steps per phase: as many as the LDS double buffers hold where a workgroup has one or two sets of chains (fewer phase
// changes: 8-D x 4096 16 -> 0.308 ms, 24 -> 0.293, 32 -> 0.291; 16-D x 8192 without recorders 16 -> 0.446, 24 -> 0.437,
// 32 -> 0.422), 20 with three (16-D x 12288, equal shares per wavefront: 12 -> 0.590, 16 -> 0.603, 20 -> 0.571, 24 -> 0.587),
// 16 from four on (32-D x 8192 with two blocks per lane: 12 -> 0.728, 16 -> 0.684, 24 -> 0.705; 16-D x 16384: 16 to 24
// within 0.3 %) -- round 4's sweeps, one box each.  Even: a generator item is two steps.
int mcxk_persist_ksteps(int lpc2, int bpl, int own)
{
  const int rec = mcxk_persist_recorders(own, bpl) ? 1 : 0;
  int k = own * bpl <= 2 ? 32 : (own * bpl == 3 ? 20 : 16);
  static const char *env = getenv("MCX_PERSIST_KSTEPS");  // tuning only (tools/persist_config_sweep.py)
  if (env && *env) k = std::max(atoi(env), 2);
  k &= ~1;  // even: a normals item is two consecutive steps, and both are always stored (mcx_persist.hpp)
  while (k > 2 && (k > PKMAX || lds_for(lpc2, bpl, own, rec, k) > MCXK_PERSIST_LDS_LIMIT || !mcxk_persist_deal_fits(lpc2, bpl, own, rec, k))) k -= 2;
  return k;
}

size_t mcxk_persist_lds_bytes(int lpc2, int bpl, int own)
{
  return lds_for(lpc2, bpl, own, mcxk_persist_recorders(own, bpl) ? 1 : 0, mcxk_persist_ksteps(lpc2, bpl, own));
}

// Who generates what.  A phase's generator items -- the normals of two consecutive steps of one (owner, block) set, the
// acceptance logs of one owner, the 1/pwgt values -- were dealt round-robin over the generator wavefronts until round 4:
// every wavefront got the same share, and the SIMD that also carries an owner (36-52 instructions per step at the top
// priority) finished last: with one owner and two blocks per lane it issues 3/14 of the generators' work PLUS the
// owner's, against 4/14 on the SIMDs without one -- 117 against 81 instructions per step, and a phase lasts as long as
// its busiest SIMD (measured: 8192 x 16-D 0.40 ms per launch = 640 cycles per step; the busiest SIMD's 117 instructions
// at the 4.8 cycles per instruction the hot-path kernel sustains are 560).  Here the items are dealt by cost: longest
// first, each to the SIMD with the least work so far (owners' and recorders' steps counted in), within it to the
// wavefront with the least.  Wavefront w of a workgroup runs on SIMD w % 4.  Costs in issued instructions (kernel_asm):
// cn per step of a normals item, ca per Philox block of an acceptance item, co / cr per owner / recorder step.
bool mcxk_persist_deal(int lpc2, int bpl, int own, int rec, int K, uint32_t *tab)
{
  bool fits = true;
  float cn = 185.0f, ca = 110.0f, co = bpl == 1 ? 34.0f : 50.0f, cr = bpl == 1 ? 24.0f : 40.0f;
  int simd_div = 0, singles = 0;  // (single-step items measured slower wherever they were dealt: 8192 x 16-D +1..7 %)
  // From three sets of chains per workgroup on, equal shares per WAVEFRONT (what the round-robin deal of rounds 2-3 gave)
  // measured better than equal work per SIMD: 16-D x 12288 0.557 -> 0.53 ms per launch
  int per_wave = own * bpl >= 3;
  static const char *env = getenv("MCX_PERSIST_COST");  // tuning only: "cn,ca,co,cr[,map[,singles[,per_wave]]]"
  if (env && *env) {
    float v[7] = {cn, ca, co, cr, 0.0f, 0.0f, (float)per_wave};
    (void)sscanf(env, "%f,%f,%f,%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]);
    cn = v[0]; ca = v[1]; co = v[2]; cr = v[3]; simd_div = v[4] != 0.0f; singles = v[5] != 0.0f; per_wave = v[6] != 0.0f;
  }
  const int nrec = rec ? own : 0, OB = own * bpl, nsteps = K * OB;  // (K is even)
  const float acc_cost = (float)((K / 4 + 1 + lpc2 - 1) / lpc2) * ca + 20.0f;
  auto simd_of = [&](int w) { return simd_div ? w / 4 : w % 4; };
  for (int t = 0; t < 3; ++t) {
    float simd[4] = {0, 0, 0, 0}, wave[PWAVES] = {};
    int quota[PWAVES] = {}, extra[PWAVES] = {};  // single steps of normals per wavefront; its other items
    uint32_t other[PWAVES][2 * POWN_MAX] = {};
    uint32_t *T = tab + (size_t)t * PWAVES * PDEAL;
    for (int i = 0; i < PWAVES * PDEAL; ++i) T[i] = PDEAL_END;
    const int first_filler = t == 0 ? own : own + nrec;
    if (t > 0)
      for (int w = 0; w < own; ++w) simd[simd_of(w)] += (float)K * (co + (t == 2 && !rec ? cr : 0.0f));
    if (t == 2)
      for (int w = own; w < own + nrec; ++w) simd[simd_of(w)] += (float)K * cr;
    auto place = [&](float cost) {  // the wavefront of the least loaded SIMD that has done the least itself
      int best = first_filler;
      for (int w = first_filler + 1; w < PWAVES; ++w) {
        const float sb = per_wave ? wave[best] : simd[simd_of(best)], sw = per_wave ? wave[w] : simd[simd_of(w)];
        if (sw < sb || (sw == sb && wave[w] < wave[best])) best = w;
      }
      simd[simd_of(best)] += cost;
      wave[best] += cost;
      return best;
    };
    // the big odd items first, then the normals in units of `unit` steps (pairs only: 2; pairs and singles: 1)
    for (int o = 0; o < own; ++o) { const int w = place(acc_cost); other[w][extra[w]++] = 1u << 14 | (uint32_t)o; }
    const int unit = singles ? 1 : 2;
    for (int i = 0; i < nsteps; i += unit) quota[place((float)unit * cn)] += unit;
    { const int w = place(10.0f); other[w][extra[w]++] = 2u << 14; }
    // quotas -> items: walk the (owner-block, step) grid; a wavefront with an odd quota shares a pair with the next one
    int w = first_filler, used = 0, cnt[PWAVES] = {};
    auto next_wave = [&]() { while (w < PWAVES && used >= quota[w]) { ++w; used = 0; } };
    next_wave();
    for (int ob = 0; ob < OB; ++ob)
      for (int gp = 0; gp < K / 2; ++gp) {
        next_wave();
        auto push = [&](int v, uint32_t code) {
          if (v < PWAVES && cnt[v] < PDEAL) T[v * PDEAL + cnt[v]++] = code;
          else fits = false;
        };
        if (w < PWAVES && quota[w] - used >= 2) {
          push(w, (uint32_t)(gp << 4 | ob));
          used += 2;
        } else {  // one step left here: the pair is split between this wavefront and the next with room
          push(w, 3u << 14 | (uint32_t)((2 * gp) << 4 | ob));
          used += 1;
          next_wave();
          push(w, 3u << 14 | (uint32_t)((2 * gp + 1) << 4 | ob));
          used += 1;
        }
      }
    for (int v = first_filler; v < PWAVES; ++v)
      for (int i = 0; i < extra[v]; ++i) {
        if (cnt[v] < PDEAL) T[v * PDEAL + cnt[v]++] = other[v][i];
        else fits = false;
      }
  }
  return fits;
}

// every item of a phase must find a place in its wavefront's list
bool mcxk_persist_deal_fits(int lpc2, int bpl, int own, int rec, int K)
{
  uint32_t scratch[MCXK_PERSIST_DEAL_WORDS];
  return mcxk_persist_deal(lpc2, bpl, own, rec, K, scratch);
}

template <int LPC2, int BPL, int LIK, bool REC>
static hipError_t go2(const RunArgs &a, hipStream_t st)
{
  {  // per instantiation and device; the value is the largest the kernel can be launched with
    static std::mutex amu;
    static std::vector<int> adevs;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(amu);
    if (std::find(adevs.begin(), adevs.end(), dev) == adevs.end()) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_run_small<LPC2, BPL, LIK, REC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)MCXK_PERSIST_LDS_LIMIT);
      if (e != hipSuccess) return e;
      adevs.push_back(dev);
    }
  }
  const unsigned nwg = (unsigned)((a.nown + a.own - 1) / a.own);
  const size_t lds = mcxk_persist_lds_bytes(LPC2, BPL, a.own);
  if (a.nburn > 0) {
    // tuner meetings inside: every workgroup of the grid must be resident at once.  What the occupancy
    // calculator and the device's CU count promise is checked here; what they cannot see (a CU mask, a
    // foreign kernel holding LDS) is caught by the meetings' own timeout.
    // (the answer is cached per instantiation, device and LDS request: the query costs microseconds of a 0.3 ms job;
    // engines on several devices and threads share this code, hence the lock)
    struct Cap { int dev; size_t lds; long long resident; };
    static std::mutex mu;
    static std::vector<Cap> caps;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    long long resident = -1;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (const Cap &c : caps)
        if (c.dev == dev && c.lds == lds) resident = c.resident;
      if (resident < 0) {
        int per_cu = 0, ncu = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_run_small<LPC2, BPL, LIK, REC>), PBLOCK, lds);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        resident = (long long)per_cu * ncu;
        caps.push_back({dev, lds, resident});
      }
    }
    if (resident < (long long)nwg) return hipErrorCooperativeLaunchTooLarge;
  }
  hipLaunchKernelGGL((k_run_small<LPC2, BPL, LIK, REC>), dim3(nwg), dim3(PBLOCK), lds, st, a);
  return hipGetLastError();
}

template <int LPC2, int BPL, int LIK>
static hipError_t go(const RunArgs &a, hipStream_t st)
{
  return mcxk_persist_recorders(a.own, BPL) ? go2<LPC2, BPL, LIK, true>(a, st) : go2<LPC2, BPL, LIK, false>(a, st);
}

template <int LPC2, int BPL>
static hipError_t by_lik(int lik, const RunArgs &a, hipStream_t st)
{
  switch (lik) {
  case LIK_ROSEN1: return go<LPC2, BPL, LIK_ROSEN1>(a, st);
  case LIK_GAUSS: return go<LPC2, BPL, LIK_GAUSS>(a, st);
  case LIK_MIX: return go<LPC2, BPL, LIK_MIX>(a, st);
  default: return hipErrorInvalidValue;
  }
}

// lpc = 4-parameter blocks per chain (a power of two), bpl of them per lane
hipError_t mcxk_launch_persist(int lpc, int bpl, int lik, const RunArgs &a, hipStream_t st)
{
  switch (bpl * 16 + lpc / bpl) {
  case 16 + 1: return by_lik<1, 1>(lik, a, st);
  case 16 + 2: return by_lik<2, 1>(lik, a, st);
  case 16 + 4: return by_lik<4, 1>(lik, a, st);
  case 16 + 8: return by_lik<8, 1>(lik, a, st);
  case 32 + 1: return by_lik<1, 2>(lik, a, st);
  case 32 + 2: return by_lik<2, 2>(lik, a, st);
  case 32 + 4: return by_lik<4, 2>(lik, a, st);
  case 64 + 1: return by_lik<1, 4>(lik, a, st);
  case 64 + 2: return by_lik<2, 4>(lik, a, st);
  default: return hipErrorInvalidValue;
  }
}
