// mcx_remote.hpp -- the Murray inter-chain proposal, MCPar::genRemote (src/mcpar.cc:315-451), as gfx950 kernels.
// Included by mcx_engine.hip only (the fused step kernels never see it).
#pragma once
#include "mcx_device.hpp"

namespace mcx {

// ---------------------------------------------------------------------------------------------
// genRemote (src/mcpar.cc:315-451).  One lane per chain (two chains per lane at 16-D and 32-D), chain vector in registers, the N
// per-chain Gaussians Q_i staged through LDS a block at a time and read back as broadcasts.
// ---------------------------------------------------------------------------------------------
// qpar[i] = (mu_i, w_i), w = 1/sig2: sum_k (mu_k - x_k)^2 / sig2_k (src/mcpar.cc:369-383) is formed the way the
// reference forms it -- xm = mu - x first, so a chain on a Gaussian's mean gives exactly 0 however narrow the
// Gaussian, then xm * xm -- with the division replaced by a multiplication with w (arithmetic v3: sub, mul,
// fma per pair-dimension; v2's fma(-x, s, mu s) lost the cancellation when |mu| s was large).  One Q_i is 2d
// contiguous floats.
static __global__ void k_remote_prep(const float *__restrict__ musigall, float *__restrict__ qpar, size_t nd)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nd) {
    const float2 ms = reinterpret_cast<const float2 *>(musigall)[i];
    reinterpret_cast<float2 *>(qpar)[i] = make_float2(ms.x, 1.0f / ms.y);
  }
}

// The sweep loop over the rows of Gaussians staged in LDS (row stride 2*DMAX floats, zero-padded past 2*DD), with
// wave-uniform early outs.  The partial sums of arg only grow (every term is a square), so once every chain of
// the wavefront has `arg > bound(lane)` after a group of dimensions the remaining dimensions cannot bring any of
// them back under its bound, and this Q_i is dropped for the whole wavefront; completed sums are the same bits
// as the oracle's qarg().  One Q_i against 64 chains is mostly far away from all of them, so most of the
// sweep ends after the first group.  use(arg) consumes a completed sum; bound() is re-read per Q_i.
// Every lane reads the same address (a broadcast ds_read_b128 = two dimensions), so both operands of
// xm = mu - x, fma(xm xm, w, arg) arrive in VGPRs.  Round 1 streamed the rows through wave-uniform scalar loads instead:
// VALU instructions take one scalar operand, so m' went through a v_mov (3 instead of 2 instructions per
// pair-dimension), and scalar loads return out of order -- every wait is a wait for all of them -- which
// exposed their full latency once per group (VALU 60 % busy).  Same operations in the same order: same bits.
// `valid` = the lane holds a chain; lanes without one take part in the wave-uniform tests as "out of range".
template <int DMAX, bool EXACT, typename Bound, typename Use>
__device__ __forceinline__ void sweep_rows(const float *rows, int nrow, const float x[DMAX], int DD, bool valid,
                                           Bound bound, Use use)
{
  constexpr int G = DMAX >= 16 ? DMAX / 4 : DMAX;  // dimensions per group: 4 groups from 16-D up
  constexpr int G4 = G / 2, R4 = DMAX / 2;          // float4 per group / per row
  const float4 *rp = reinterpret_cast<const float4 *>(rows);
  float4 cur[G4];
#pragma unroll
  for (int k = 0; k < G4; ++k) cur[k] = rp[k];
  auto two_dims = [&](float4 v, int k, float arg) {  // dimensions k, k + 1 (k even)
    if (EXACT || k < DD) {
      const float xm = v.x - x[k];
      arg = __builtin_fmaf(xm * xm, v.y, arg);
    }
    if (EXACT || k + 1 < DD) {
      const float xm = v.z - x[k + 1];
      arg = __builtin_fmaf(xm * xm, v.w, arg);
    }
    return arg;
  };
  const unsigned long long everyone = __ballot(true), nochain = __ballot(!valid);
  for (int r = 0; r < nrow; ++r, rp += R4) {
    const float4 *rn = r + 1 < nrow ? rp + R4 : rp;
    float arg = 0.0f;
#pragma unroll
    for (int k = 0; k < G4; ++k) arg = two_dims(cur[k], 2 * k, arg);
    // the next row's first group, into the registers that have just been read (loads issued any earlier would
    // need a second set and a copy per row); in flight during this row's other groups / the next wavefronts' turn
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < G4; ++k) cur[k] = rn[k];
    __builtin_amdgcn_sched_barrier(0);
    const float b = bound();
    bool live = (__ballot(arg > b) | nochain) != everyone;
    if (live && G < DMAX) {
#pragma unroll
      for (int c = G; c < DMAX; c += G) {
#pragma unroll
        for (int k = 0; k < G4; ++k) arg = two_dims(rp[c / 2 + k], c + 2 * k, arg);
        if ((__ballot(arg > b) | nochain) == everyone) {  // also after the last group: the consumer's exp is skipped
          live = false;
          break;
        }
      }
    }
    if (live) use(arg);
  }
}

// Two chains per lane (components .x / .y of every pair): a row read from LDS then serves 128 chains of the
// wavefront instead of 64 -- the broadcast reads are what the one-chain loop is bound by (LDS 82-90 % busy at
// d = 32, VALU 70 %) -- and the arithmetic is packed (v_pk_add / v_pk_mul / v_pk_fma_f32 with the row's mu, w
// selected by op_sel: component-wise the same operations in the same order, hence the same bits).  d == DMAX only.
// `todo` names the rows of the stage to sweep, one bit per row (wave-uniform): the rows the wavefront's exclusion
// test (k_cull_test) could not rule out, or all of them.
// (Testing for the early out once per Q_i instead of after every group of dimensions, for the regime where the
// exclusion masks exclude nothing -- the 32-D mixture -- was measured slower, 47.0 against 43.3 ms per job: the
// tests after the second and third group do fire there.)
template <int DMAX, typename Bound, typename Use>
__device__ __forceinline__ void sweep_rows2(const float *rows, unsigned long long todo, const f32x2 x[DMAX], bool valid_a,
                                            bool valid_b, Bound bound, Use use)
{
  constexpr int G = DMAX >= 16 ? DMAX / 4 : DMAX;
  constexpr int G4 = G / 2, R4 = DMAX / 2;
  if (!todo) return;
  const float4 *base = reinterpret_cast<const float4 *>(rows);
  const float4 *rp = base + (size_t)__builtin_ctzll(todo) * R4;
  todo &= todo - 1;
  float4 cur[G4];
#pragma unroll
  for (int k = 0; k < G4; ++k) cur[k] = rp[k];
  auto two_dims = [&](float4 v, int k, f32x2 arg) {
    const f32x2 xm0 = splat2(v.x) - x[k];
    arg = fma2(xm0 * xm0, splat2(v.y), arg);
    const f32x2 xm1 = splat2(v.z) - x[k + 1];
    arg = fma2(xm1 * xm1, splat2(v.w), arg);
    return arg;
  };
  const unsigned long long everyone = __ballot(true), no_a = __ballot(!valid_a), no_b = __ballot(!valid_b);
  for (;;) {
    const bool more = todo != 0;
    const float4 *rn = more ? base + (size_t)__builtin_ctzll(todo) * R4 : rp;
    f32x2 arg = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < G4; ++k) arg = two_dims(cur[k], 2 * k, arg);
    __builtin_amdgcn_sched_barrier(0);  // (see sweep_rows)
#pragma unroll
    for (int k = 0; k < G4; ++k) cur[k] = rn[k];
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 b = bound();
    bool live = ((__ballot(arg.x > b.x) | no_a) & (__ballot(arg.y > b.y) | no_b)) != everyone;
    if (live && G < DMAX) {
#pragma unroll
      for (int c = G; c < DMAX; c += G) {
#pragma unroll
        for (int k = 0; k < G4; ++k) arg = two_dims(rp[c / 2 + k], c + 2 * k, arg);
        if (((__ballot(arg.x > b.x) | no_a) & (__ballot(arg.y > b.y) | no_b)) == everyone) {
          live = false;
          break;
        }
      }
    }
    if (live) use(arg);
    if (!more) break;
    todo &= todo - 1;
    rp = rn;
  }
}

// arg > ZERO_ARG  =>  expf_v2(-arg/2) == 0 exactly: -arg/2 < -88 gives n = floor(-88 log2(e) + 1/2) <= -127 < -125
constexpr float ZERO_ARG = 176.0f;

constexpr int QBLOCK = 256;  // block length of the qisum summation order
// exact exclusion of far Gaussians (further down): chains per group = the chains of one wavefront of the two-chain sweeps,
// key dimensions and bits per key dimension of the spatial sort
constexpr int CULL_W = 128, CULL_KD = 4, CULL_BITS = 3, CULL_BINS = 1 << (CULL_KD * CULL_BITS);
constexpr int CULL_NCOUNT = 64;  // cells of a "pairs kept" counter (the host adds them up): same-address atomics are slow

struct RemoteArgs {
  const int *active_in;  // compacted list of still-rejected chains (null = all chains, pass 0)
  int nact;
  int *active_out;
  int *nact_out;   // survivors of this pass (zero when the pass's decide starts)
  int *nact_zero;  // the other one of the two counters: zeroed for the next pass, or null
  const float *musigall, *winv, *cmax;  // winv = qpar: (mu, 1/sig2) pairs
  float *ptrial, *mutrial, *sigtrial, *cfac;
  float *racpt;        // [n] rejection threshold of this pass, by chain
  float *psum, *pmax;  // [S][nact] per-block partial sums / maxima, by position in the active list
  int n, d, N, pass, S;
  uint32_t g0, t, seed;
  // k_remote_decide's last workgroup hands the pass's counters to the host itself (no copy kernel behind it): `ncounts`
  // 64-bit words from `counts` (the two survivor counters, then the screens' cells) to `counts_host` (pinned, mapped);
  // `done` counts the workgroups that are through and is left zero; then `serial` goes to counts_host[nflag].
  // counts_host null: nothing of the kind.
  const unsigned long long *counts;
  unsigned long long *counts_host;
  unsigned *done;
  int ncounts;
  int nflag;                  // counts_host[nflag] takes ...
  unsigned long long serial;  // ... the pass's serial number after the counters: what the host polls for
  // Several passes at once over few chains (k_remote_draw_multi / k_remote_decide_multi): candidate c of the chain at
  // position i of the active list is the proposal pass + c would draw for it, kept in row i * ncand + c of these
  float *cand_p, *cand_mu, *cand_sig, *cand_racpt;
  int *tried;  // [ncand]: chains that came as far as candidate c (what pass + c would have had to sweep)
  int ncand;
};

// Murray draw for every still-rejected chain (src/mcpar.cc:337-352): pick a component, draw from
// its diagonal Gaussian, keep (mutrial, sigtrial); one lane per chain.
__device__ __forceinline__ void remote_draw_row(const RemoteArgs &a, int j, int pass, float *__restrict__ p, float *__restrict__ mu,
                                                float *__restrict__ sig, float *__restrict__ racpt)
{
  const int d = a.d;
  const uint32_t g = a.g0 + (uint32_t)j;
  const u32x4 w = philox4x32_10(a.t, g, (uint32_t)pass, 0u, a.seed, ST_RSEL);
  const int sel = (int)(((uint64_t)w.x * (uint64_t)a.N) >> 32);  // src/mcpar.cc:337
  for (int qb = 0; 4 * qb < d; ++qb) {
    float z[4];
    normal4_from_words(philox4x32_10(a.t, g, (uint32_t)pass, (uint32_t)qb, a.seed, ST_RNORM), z);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * qb + c;
      if (k < d) {  // src/mcpar.cc:339-352
        const float m = a.musigall[2 * ((size_t)sel * d + k)];
        const float sg = __builtin_sqrtf(a.musigall[2 * ((size_t)sel * d + k) + 1]);
        mu[k] = m;
        sig[k] = sg;
        p[k] = __builtin_fmaf(sg, z[c], m);
      }
    }
  }
  *racpt = u24(w.y);  // src/mcpar.cc:401
}
__device__ __forceinline__ void remote_draw_one(const RemoteArgs &a, int j, int pass)
{
  const size_t o = (size_t)j * a.d;
  remote_draw_row(a, j, pass, a.ptrial + o, a.mutrial + o, a.sigtrial + o, a.racpt + j);
}

// the first pass of a Murray step draws for every chain; later passes are drawn by the k_remote_decide that rejected them
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_remote_draw(const RemoteArgs a)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i == 0) *a.nact_out = 0;  // (this pass's decide counts its survivors there)
  if (i >= a.nact) return;
  remote_draw_one(a, a.active_in ? a.active_in[i] : i, a.pass);
}

// Few chains left: the proposals of this pass AND of the ncand - 1 passes behind it, for every still-rejected chain --
// they depend on (step, chain, pass) only, not on what the passes before them decided.  One lane per candidate.
static __global__ __launch_bounds__(BLOCK) void k_remote_draw_multi(const RemoteArgs a)
{
  const int v = blockIdx.x * BLOCK + threadIdx.x;
  if (v < a.ncand) a.tried[v] = 0;
  if (v >= a.nact * a.ncand) return;
  const int i = v / a.ncand, c = v - i * a.ncand;
  const size_t o = (size_t)v * a.d;
  remote_draw_row(a, a.active_in ? a.active_in[i] : i, a.pass + c, a.cand_p + o, a.cand_mu + o, a.cand_sig + o, a.cand_racpt + v);
}

// The all-pairs sweep (src/mcpar.cc:367-395 for ptrial, :421-437 for pvals): lanes = chains (vector
// in registers), blockIdx.y = one block of QBLOCK consecutive Q_i, staged through LDS by the workgroup's four
// wavefronts and read back with broadcast ds_read_b128 (sweep_rows): one fetch serves 256 chains.  SUMS: writes the block's partial
// sum and maximum of Q = exp(-arg/2); blocks are combined in index order by k_remote_decide (fixed
// summation order).  !SUMS (the cfac numerator max_i Q_i = exp(-min_i arg_i / 2)): writes the block's
// minimum of arg -- no exp per pair; k_remote_cmax_combine takes the one exp per chain.
//
// `excl` (two chains per lane only): the exclusion masks of k_cull_test, [word][group] with one bit per Q_i -- group =
// the CULL_W = 128 consecutive positions of the active list this wavefront holds.  A Q_i whose bit is clear is
// farther than the wavefront's bound from the bounding box of its 128 chains, i.e. from every one of them: its
// term is exactly +0 in every chain's sum (SUMS) or cannot lower any chain's minimum (!SUMS), so skipping it
// leaves every result bit as it was.  null = sweep everything.
// NT = threads of the workgroup (every workgroup stages all of its block's Gaussians through LDS whichever rows its
// wavefronts then skip; larger workgroups = fewer copies were measured and are not faster: mcx_engine.hip).
// (32-D with two chains per lane: asked to fit four workgroups per CU, i.e. 128 VGPRs instead of the 130-146 the
// compiler takes unasked -- four wavefronts per SIMD instead of three, no spills: the dense sweeps of the 32-D
// mixture 49 -> 52-54 "TFLOP/s"; five would spill)
template <int DMAX, bool SUMS, bool EXACT, int CPL = 1, int NT = BLOCK>
__global__ __launch_bounds__(NT, (DMAX == 32 && CPL == 2) ? 4 : 1) void k_remote_sweep(const float *__restrict__ xrows,
                                                        const int *__restrict__ active, int nact,
                                                        const float *__restrict__ qpar,
                                                        float *__restrict__ psum,
                                                        float *__restrict__ pmax, int d, int N, int own0,
                                                        const unsigned long long *__restrict__ excl, int excl_words)
{
  static_assert(CPL == 1 || (CPL == 2 && EXACT), "two chains per lane: d == DMAX only");
  // the block's Gaussians pass through LDS 16 KB (8 KB with two chains per lane) at a time
  constexpr int QSUB = DMAX >= 8 ? (2048 / CPL) / DMAX : QBLOCK;
  static_assert(CPL == 1 || QSUB <= 64, "a stage's rows are named by one 64-bit mask");
  __shared__ __attribute__((aligned(16))) float qlds[QSUB * 2 * DMAX];
  // one chain per lane: chain = position blockIdx.x * NT + threadIdx.x of the (active) list; two: the wavefront
  // holds CULL_W = 128 CONSECUTIVE positions (lane l: its first 64 + l and its second 64 + l) -- neighbours in the
  // sorted list the exclusion test was made for.  Every lane stays for the staging and its barriers.
  int pos[CPL], jj[CPL];
  bool valid[CPL];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    pos[c] = CPL == 1 ? (int)(blockIdx.x * NT + threadIdx.x)
                      : (int)blockIdx.x * (CPL * NT) + wv * (CPL * 64) + c * 64 + (int)(threadIdx.x & 63u);
    valid[c] = pos[c] < nact;
    jj[c] = valid[c] ? (active ? active[pos[c]] : pos[c]) : 0;
  }
  const unsigned long long *exw = nullptr;  // this wavefront's mask words: [word][group], excl_words = groups
  if constexpr (CPL == 2)
    if (excl) exw = excl + (size_t)((int)blockIdx.x * (NT / 64) + wv);
  const int sb = blockIdx.y;
  const int DD = EXACT ? DMAX : d;
  float x[CPL == 1 ? DMAX : 1];
  f32x2 xx[CPL == 2 ? DMAX : 1];
  if constexpr (CPL == 1) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) x[k] = 0.0f;
    if (valid[0]) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (EXACT || k < DD) x[k] = xrows[(size_t)jj[0] * DD + k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xx[k] = f32x2{0.0f, 0.0f};
    if (valid[0]) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) xx[k].x = xrows[(size_t)jj[0] * DMAX + k];
    }
    if (valid[CPL - 1]) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) xx[k].y = xrows[(size_t)jj[CPL - 1] * DMAX + k];
    }
  }
  const int q0 = sb * QBLOCK, q1 = (q0 + QBLOCK < N) ? q0 + QBLOCK : N;
  const bool wave_idle = !__any(valid[0]);
  // d == DMAX: a stage is one contiguous piece of qpar, PER float4 per thread, fetched into registers while the
  // previous stage is being swept and written to LDS between two barriers
  constexpr int PER = (QSUB * (DMAX / 2) + NT - 1) / NT;
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 hold[PER];
  if (EXACT) {
    const int n4 = (q1 - q0 < QSUB ? q1 - q0 : QSUB) * (DMAX / 2);
    const v4 *src = reinterpret_cast<const v4 *>(qpar + 2 * (size_t)q0 * DMAX);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = (int)threadIdx.x + u * NT;
      hold[u] = src[t < n4 ? t : n4 - 1];  // (unconditional: the PER loads go out back to back)
    }
  }
  f32x2 part = {0.0f, 0.0f}, m = {0.0f, 0.0f};              // SUMS: the block's sum and maximum of Q, per chain
  f32x2 amin = {__builtin_inff(), __builtin_inff()};         // !SUMS: the block's minimum of arg
  if (!SUMS && own0 >= 0) {
    // Every block starts from the chain's arg against its OWN Gaussian (global index own0 + j, one of the N: the
    // combined minimum over the blocks is unchanged), which is small -- the chain sits inside its own running
    // posterior -- so nearly every other Q_i is abandoned after its first group of dimensions.
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      if (valid[c]) {
        const float *qo = qpar + 2 * (size_t)(own0 + jj[c]) * DD;
        float a0 = 0.0f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
          if (EXACT || k < DD) {
            const float xk = CPL == 1 ? x[CPL == 1 ? k : 0] : (c == 0 ? xx[CPL == 2 ? k : 0].x : xx[CPL == 2 ? k : 0].y);
            const float xm = qo[2 * k] - xk;
            a0 = __builtin_fmaf(xm * xm, qo[2 * k + 1], a0);
          }
        const float old = c == 0 ? amin.x : amin.y;
        const float now = a0 < old ? a0 : old;  // a NaN stays out, like below
        if (c == 0) amin.x = now;
        else amin.y = now;
      }
  }
  for (int c0 = q0; c0 < q1; c0 += QSUB) {
    const int nrow = q1 - c0 < QSUB ? q1 - c0 : QSUB;
    __syncthreads();  // the previous rows have been consumed
    if (EXACT) {
      v4 *dst = reinterpret_cast<v4 *>(qlds);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int t = (int)threadIdx.x + u * NT;
        if (t < nrow * (DMAX / 2)) dst[t] = hold[u];
      }
    } else {
      const float *src = qpar + 2 * (size_t)c0 * DD;
      for (int t = threadIdx.x; t < nrow * 2 * DMAX; t += NT) {
        const int r = t / (2 * DMAX), k = t % (2 * DMAX);
        qlds[t] = k < 2 * DD ? src[(size_t)r * 2 * DD + k] : 0.0f;
      }
    }
    __syncthreads();
    if (EXACT && c0 + QSUB < q1) {
      const int cn = c0 + QSUB;
      const int n4 = (q1 - cn < QSUB ? q1 - cn : QSUB) * (DMAX / 2);
      const v4 *src = reinterpret_cast<const v4 *>(qpar + 2 * (size_t)cn * DMAX);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int t = (int)threadIdx.x + u * NT;
        hold[u] = src[t < n4 ? t : n4 - 1];
      }
    }
    if (wave_idle) continue;
    // a dropped Q_i is +0 for every chain of the wavefront: part + 0 = part, m unchanged; an arg equal to the
    // bound cannot lower the minimum either: `>` serves both sweeps
    if constexpr (CPL == 1) {
      if (SUMS) {
        sweep_rows<DMAX, EXACT>(qlds, nrow, x, DD, valid[0], [] { return ZERO_ARG; }, [&](float av) {
          const float gv = expf_v2(-0.5f * av);
          part.x = part.x + gv;
          m.x = gv > m.x ? gv : m.x;
        });
      } else {
        sweep_rows<DMAX, EXACT>(qlds, nrow, x, DD, valid[0], [&] { return amin.x < ZERO_ARG ? amin.x : ZERO_ARG; },
                                [&](float av) { amin.x = av < amin.x ? av : amin.x; });
      }
    } else {
      // rows of this stage left to sweep: Q_i c0 .. c0 + nrow - 1 are bits (c0 & 63) .. of word c0 / 64
      unsigned long long todo = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);
      if (exw) {
        const unsigned long long w = exw[(size_t)(c0 >> 6) * (size_t)excl_words];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)w), hi = __builtin_amdgcn_readfirstlane((unsigned)(w >> 32));
        todo &= (((unsigned long long)hi << 32) | lo) >> (c0 & 63);
      }
      if (SUMS) {
        sweep_rows2<DMAX>(qlds, todo, xx, valid[0], valid[CPL - 1], [] { return splat2(ZERO_ARG); }, [&](f32x2 av) {
          const f32x2 gv = expf_v2x2_nonpos(splat2(-0.5f) * av);
          part = part + gv;
          m.x = gv.x > m.x ? gv.x : m.x;
          m.y = gv.y > m.y ? gv.y : m.y;
        });
      } else {
        // (an arg beyond ZERO_ARG cannot matter: k_remote_cmax_combine's exp1(-min / 2) is exactly 0 from there on)
        sweep_rows2<DMAX>(qlds, todo, xx, valid[0], valid[CPL - 1],
                          [&] { return f32x2{amin.x < ZERO_ARG ? amin.x : ZERO_ARG, amin.y < ZERO_ARG ? amin.y : ZERO_ARG}; }, [&](f32x2 av) {
          amin.x = av.x < amin.x ? av.x : amin.x;
          amin.y = av.y < amin.y ? av.y : amin.y;
        });
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    if (valid[c]) {
      const size_t o = (size_t)sb * nact + pos[c];  // [block][position]: coalesced here and in the combining kernels
      if (SUMS) {
        psum[o] = c == 0 ? part.x : part.y;
        pmax[o] = c == 0 ? m.x : m.y;
      } else {
        pmax[o] = c == 0 ? amin.x : amin.y;
      }
    }
}

// The masked sweep at 16-D without LDS: when the exclusion masks leave a wavefront a third of its block's Gaussians,
// staging all 256 of them through LDS behind two workgroup barriers per 64 rows is what the wavefronts wait for (VALU
// issue 0.39 / 0.58 of the slots, min-arg / sum sweep).  Here every wavefront walks its OWN surviving rows and reads
// each straight from L2 through the scalar cache: the row index comes from the mask (wave-uniform), so the 32 numbers
// of a row arrive in SGPRs, and with arithmetic v3 every packed operation needs at most one scalar operand
// (xm = mu - x; q = xm xm; arg = fma(q, w, arg)) -- no v_mov, no barrier, no row fetched that nobody wants.
// One wavefront = one group of 128 consecutive chains x one block of QBLOCK Gaussians; same operations in the same
// order as sweep_rows2, same early outs, hence the same bits.  excl == null sweeps every row: used for the late
// rejection passes over a few thousand chains, where the LDS kernel's grid is too small to hide its barriers (VALU
// issue 0.25) and the rows every wavefront then pulls through L2 are few.
// np = 32: the same with a row in two halves of 16 dimensions, each the 32 SGPRs a 16-D row takes -- the second half is
// on its way while the first is swept and is swept only if the first left the row alive.
template <int DMAX, bool SUMS>
__global__ __launch_bounds__(BLOCK) void k_remote_sweep_srow(const float *__restrict__ xrows, const int *__restrict__ active,
                                                               int nact, const float *__restrict__ qpar,
                                                               float *__restrict__ psum, float *__restrict__ pmax, int N,
                                                               int own0, const unsigned long long *__restrict__ excl,
                                                               int ngroups, int bpw, int y0)
{
  static_assert(DMAX == 16 || DMAX == 32, "rows of one or two 16-dimension halves");
  const int by = (int)blockIdx.y + y0;  // (y0 > 0: a column chunk of the pass, launched on its own: mcx_murray.hip)
  // dimensions per half-row, halves, dimensions between two early-out tests (two tests per row: behind the per-pair
  // screen few rows leave early, and four tests cost 1.5 % of both Murray jobs more than two or one)
  constexpr int H = 16, NH = DMAX / H, G = DMAX / 2;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
  const int g = (int)blockIdx.x * (BLOCK / 64) + wv;
  if (g >= ngroups) return;  // (no barriers in this kernel)
  // `bpw` consecutive blocks per wavefront: what a wavefront does before its first row -- its chains' vectors, their
  // args against their own Gaussians -- is 1-2 GB through L2 per 65 536 x 65 536 sweep when every (group, block) pair
  // does it anew, more than the rows the masks leave.  SUMS: one partial sum per block as ever (the summation order,
  // DESIGN.md S3.5); !SUMS: the minimum runs on through the wavefront's blocks and is stored once, at "block"
  // blockIdx.y (the combining kernel takes a minimum: fewer entries, the same result).
  const int nsb = (N + QBLOCK - 1) / QBLOCK;
  const int sb0 = by * bpw, sb1 = sb0 + bpw < nsb ? sb0 + bpw : nsb;
  int pos[2], jj[2];
  bool valid[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    pos[c] = g * CULL_W + c * 64 + lane;
    valid[c] = pos[c] < nact;
    jj[c] = valid[c] ? (active ? active[pos[c]] : pos[c]) : 0;
  }
  f32x2 xx[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) xx[k] = f32x2{0.0f, 0.0f};
  if (valid[0]) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xx[k].x = xrows[(size_t)jj[0] * DMAX + k];
  }
  if (valid[1]) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xx[k].y = xrows[(size_t)jj[1] * DMAX + k];
  }
  f32x2 part = {0.0f, 0.0f}, m = {0.0f, 0.0f};
  f32x2 amin = {__builtin_inff(), __builtin_inff()};
  if (!SUMS && own0 >= 0) {  // (see k_remote_sweep)
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (valid[c]) {
        const float *qo = qpar + 2 * (size_t)(own0 + jj[c]) * DMAX;
        float a0 = 0.0f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
          const float xm = qo[2 * k] - (c == 0 ? xx[k].x : xx[k].y);
          a0 = __builtin_fmaf(xm * xm, qo[2 * k + 1], a0);
        }
        const float old = c == 0 ? amin.x : amin.y;
        const float now = a0 < old ? a0 : old;
        if (c == 0) amin.x = now;
        else amin.y = now;
      }
  }
  const unsigned long long everyone = __ballot(true), no_a = __ballot(!valid[0]), no_b = __ballot(!valid[1]);
  for (int sb = sb0; sb < sb1; ++sb) {
  if (SUMS) { part = f32x2{0.0f, 0.0f}; m = f32x2{0.0f, 0.0f}; }
  const int q0 = sb * QBLOCK;
  // the block's four mask words (bits beyond N are clear: k_cull_test), as scalars
  unsigned long long words[QBLOCK / 64];
#pragma unroll
  for (int wd = 0; wd < QBLOCK / 64; ++wd) {
    unsigned long long w = 0;
    if (q0 + 64 * wd < N) {
      if (excl) {
        w = excl[(size_t)((q0 >> 6) + wd) * (size_t)ngroups + g];
      } else {  // no masks: every Gaussian of the block that exists
        const int left = N - (q0 + 64 * wd);
        w = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
      }
    }
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)w), hi = __builtin_amdgcn_readfirstlane((unsigned)(w >> 32));
    words[wd] = ((unsigned long long)hi << 32) | lo;
  }
  int wd = 0;
  unsigned long long todo = words[0];
  // next surviving row of the block (index within the block), or -1
  auto next_row = [&]() -> int {
#pragma unroll
    for (int u = 1; u < QBLOCK / 64; ++u)
      if (!todo && wd + 1 < QBLOCK / 64) { ++wd; todo = wd == 1 ? words[1] : (wd == 2 ? words[2] : words[3]); }
    if (!todo) return -1;
    const int r = 64 * wd + __builtin_ctzll(todo);
    todo &= todo - 1;
    return r;
  };
  // one row = 32 floats read through the CONSTANT address space at a wave-uniform address: scalar loads
  // (s_load_dwordx4 .. x16 into SGPRs), waited for by the compiler's own s_waitcnt
  typedef float v4 __attribute__((ext_vector_type(4)));
  struct Row { v4 v[H / 2]; };
  auto fetch = [&](int r, int half, Row &row) {
    const __attribute__((address_space(4))) v4 *rp =
        (const __attribute__((address_space(4))) v4 *)(qpar + 2 * ((size_t)(q0 + r) * DMAX + (size_t)half * H));
#pragma unroll
    for (int k = 0; k < H / 2; ++k) row.v[k] = rp[k];
  };
  auto two_dims = [&](v4 v, int k, f32x2 arg) {  // (mu_k, w_k, mu_k+1, w_k+1)
    const f32x2 xm0 = splat2(v.x) - xx[k];
    arg = fma2(xm0 * xm0, splat2(v.y), arg);
    const f32x2 xm1 = splat2(v.z) - xx[k + 1];
    arg = fma2(xm1 * xm1, splat2(v.w), arg);
    return arg;
  };
  auto bound_now = [&]() {
    return SUMS ? splat2(ZERO_ARG) : f32x2{amin.x < ZERO_ARG ? amin.x : ZERO_ARG, amin.y < ZERO_ARG ? amin.y : ZERO_ARG};
  };
  // dimensions half * H .. half * H + H - 1 of a row: false = every chain of the wavefront is past its bound
  auto sweep_half = [&](const Row &row, int half, f32x2 &arg, const f32x2 b) {
#pragma unroll
    for (int c = 0; c < H; c += G) {
#pragma unroll
      for (int k = 0; k < G / 2; ++k) arg = two_dims(row.v[c / 2 + k], half * H + c + 2 * k, arg);
      if (((__ballot(arg.x > b.x) | no_a) & (__ballot(arg.y > b.y) | no_b)) == everyone) return false;
    }
    return true;
  };
  auto consume = [&](const f32x2 arg) {
    if (SUMS) {
      const f32x2 gv = expf_v2x2_nonpos(splat2(-0.5f) * arg);
      part = part + gv;
      m.x = gv.x > m.x ? gv.x : m.x;
      m.y = gv.y > m.y ? gv.y : m.y;
    } else {
      amin.x = arg.x < amin.x ? arg.x : amin.x;
      amin.y = arg.y < amin.y ? arg.y : amin.y;
    }
  };
  auto sweep_one = [&](const Row &row) {  // (a whole 16-D row)
    f32x2 arg = {0.0f, 0.0f};
    if (sweep_half(row, 0, arg, bound_now())) consume(arg);
  };
  // Two row buffers in turn: the next surviving row is requested before this one is swept.  Scalar loads return
  // out of order, so the only wait there is is "all of them": the wait for THIS row must come before the request
  // for the next one, or it would wait for that too -- `landed` is a use of the row's first and last number that the
  // compiler has to satisfy at that point.
  auto landed = [&](const Row &row) { asm volatile("" ::"s"(row.v[0].x), "s"(row.v[H / 2 - 1].w)); };
  Row ra, rb;
  int r = next_row();
  if (r >= 0) {
    fetch(r, 0, ra);
    if constexpr (NH == 1) {
      for (;;) {
        landed(ra);
        r = next_row();
        if (r >= 0) fetch(r, 0, rb);
        sweep_one(ra);
        if (r < 0) break;
        landed(rb);
        r = next_row();
        if (r >= 0) fetch(r, 0, ra);
        sweep_one(rb);
        if (r < 0) break;
      }
    } else {
      for (;;) {
        landed(ra);
        fetch(r, 1, rb);
        f32x2 arg = {0.0f, 0.0f};
        const f32x2 b = bound_now();
        bool live = sweep_half(ra, 0, arg, b);
        landed(rb);
        const int rn = next_row();
        if (rn >= 0) fetch(rn, 0, ra);
        if (live) live = sweep_half(rb, 1, arg, b);
        if (live) consume(arg);
        if (rn < 0) break;
        r = rn;
      }
    }
  }
  if (SUMS) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (valid[c]) {
        const size_t o = (size_t)sb * nact + pos[c];
        psum[o] = c == 0 ? part.x : part.y;
        pmax[o] = c == 0 ? m.x : m.y;
      }
  }
  }
  if (!SUMS) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (valid[c]) pmax[(size_t)by * nact + pos[c]] = c == 0 ? amin.x : amin.y;
  }
}

// numerator of cfac: max_i Q_i(pvals_j) = exp(-min_i arg_i / 2) (src/mcpar.cc:421-437); does not depend on the pass
// (position i of the list the sweep ran over holds chain order[i]; null = identity)
static __global__ void k_remote_cmax_combine(const float *__restrict__ pmin, float *__restrict__ cmax, int n, int S,
                                             const int *__restrict__ order)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float amin = __builtin_inff();
  constexpr int U = 16;  // loads in flight per lane (see k_remote_decide)
  int sb = 0;
  for (; sb + U <= S; sb += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = pmin[(size_t)(sb + u) * n + i];
#pragma unroll
    for (int u = 0; u < U; ++u) amin = v[u] < amin ? v[u] : amin;
  }
  for (; sb < S; ++sb) {
    const float v = pmin[(size_t)sb * n + i];
    amin = v < amin ? v : amin;
  }
  cmax[order ? order[i] : i] = expf_v2(-0.5f * amin);
}

// One slot of a global list for every lane that calls this (inside a divergent branch): ONE atomic per wavefront
// for all its callers -- tens of thousands of same-address atomics cost ~10 ns each on this chip, which made the
// survivor compaction of a 65 536-chain pass a 70 us kernel -- and the lanes take consecutive slots.
__device__ __forceinline__ int wave_slot(int *counter)
{
  const unsigned long long callers = __ballot(true);  // the lanes active here
  const int lane = (int)(threadIdx.x & 63u), leader = __builtin_ctzll(callers);
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, (int)__popcll(callers));
  base = __shfl(base, leader);
  return base + (int)__popcll(callers & ((1ull << lane) - 1ull));
}

// the last workgroup of a deciding kernel copies the pass's counters out: what the host waits for is this kernel, not a
// copy behind it
__device__ __forceinline__ void decide_hand_over(const RemoteArgs &a)
{
  if (!a.counts_host) return;
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(a.done, 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (last) {
    for (int k = (int)threadIdx.x; k < a.ncounts; k += (int)blockDim.x)
      a.counts_host[k] = __hip_atomic_load(a.counts + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) *a.done = 0u;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {  // the counters are on their way: now the pass's serial number, which the host spins on
      __hip_atomic_store(a.counts_host + a.nflag, a.serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// rejection test of the pass (src/mcpar.cc:397-441); survivors are compacted for the next pass
static __global__ void k_remote_decide(const RemoteArgs a)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.nact) {
    const int j = a.active_in ? a.active_in[i] : i;
    float qs = FPEPS, qm = FPEPS;  // src/mcpar.cc:355-365
    // The partials are added in block order (the arithmetic contract), but they are REQUESTED 16 blocks at a time:
    // a 65 536-chain pass reads 134 MB here with one wavefront per SIMD, so the loads in flight per lane are its speed
    // (66 -> 14 us; 64 at a time for the late passes over a few hundred chains: 13 -> 18 us, no).
    constexpr int U = 16;
    const float *ps = a.psum + i, *pm = a.pmax + i;
    int sb = 0;
    for (; sb + U <= a.S; sb += U) {
      float s[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = ps[(size_t)(sb + u) * a.nact];
        v[u] = pm[(size_t)(sb + u) * a.nact];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qs = qs + s[u];
        qm = v[u] > qm ? v[u] : qm;
      }
    }
    for (; sb < a.S; ++sb) {
      qs = qs + ps[(size_t)sb * a.nact];
      const float v = pm[(size_t)sb * a.nact];
      qm = v > qm ? v : qm;
    }
    const float pacpt = qm / qs;
    if (i == 0 && a.nact_zero) *a.nact_zero = 0;  // the next pass's counter (this pass counts in the other one)
    if (a.racpt[j] < pacpt) {
      a.cfac[j] = a.cmax[j] / qm;
    } else {
      a.active_out[wave_slot(a.nact_out)] = j;
      remote_draw_one(a, j, a.pass + 1);  // rejected: its next proposal, now (src/mcpar.cc:337-352 of the next pass)
    }
  }
  decide_hand_over(a);
}

// The same over candidates (k_remote_draw_multi): a chain takes the first of its ncand proposals that passes its test --
// what ncand passes one after the other would have done, every test being a function of (step, chain, pass) and of the
// sums over ALL Gaussians only -- and is a survivor if none does.  tried[c] counts the chains that got as far as
// candidate c: the chains pass + c would have swept.  One lane per CANDIDATE (the four of a chain are a quad of lanes:
// their sums are read side by side, not one after the other), ncand == 4.
static __global__ __launch_bounds__(BLOCK) void k_remote_decide_multi(const RemoteArgs a)
{
  const int v = blockIdx.x * BLOCK + threadIdx.x;  // row i * 4 + c
  const int nv = a.nact * 4;
  const bool have = v < nv;
  const int i = v >> 2, c = v & 3;
  const int j = have ? (a.active_in ? a.active_in[i] : i) : 0;
  bool ok = false;
  float qm = FPEPS;
  if (have) {
    float qs = FPEPS;  // src/mcpar.cc:355-365
    constexpr int U = 16;
    const float *ps = a.psum + v, *pm = a.pmax + v;
    int sb = 0;
    for (; sb + U <= a.S; sb += U) {
      float s[U], x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = ps[(size_t)(sb + u) * nv];
        x[u] = pm[(size_t)(sb + u) * nv];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qs = qs + s[u];
        qm = x[u] > qm ? x[u] : qm;
      }
    }
    for (; sb < a.S; ++sb) {
      qs = qs + ps[(size_t)sb * nv];
      const float x = pm[(size_t)sb * nv];
      qm = x > qm ? x : qm;
    }
    ok = a.cand_racpt[v] < qm / qs;
  }
  // the chain's quad: which of its candidates pass, and the first of them
  const unsigned long long pass_mask = __ballot(ok);
  const unsigned quad = (unsigned)(pass_mask >> ((threadIdx.x & 63u) & ~3u)) & 0xfu;
  const int first = quad ? __builtin_ctz(quad) : 4;
  if (v == 0 && a.nact_zero) *a.nact_zero = 0;
  if (have && c == first) {  // this one becomes the chain's proposal
    a.cfac[j] = a.cmax[j] / qm;
    for (int k = 0; k < a.d; ++k) {
      a.ptrial[(size_t)j * a.d + k] = a.cand_p[(size_t)v * a.d + k];
      a.mutrial[(size_t)j * a.d + k] = a.cand_mu[(size_t)v * a.d + k];
      a.sigtrial[(size_t)j * a.d + k] = a.cand_sig[(size_t)v * a.d + k];
    }
  }
  if (have && c == 0 && first == 4) a.active_out[wave_slot(a.nact_out)] = j;
  const int ntried = (have && c == 0) ? (first < 4 ? first + 1 : 4) : 0;  // (counted once per chain)
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {  // (every lane is here: whole-wavefront ballots)
    const int cnt = (int)__popcll(__ballot(ntried > cc));
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(a.tried + cc, cnt);
  }
  decide_hand_over(a);
}

// ---------------------------------------------------------------------------------------------
// Exact exclusion of far Gaussians.  A Murray sweep is all pairs (chain, Q_i), but a chain's sum has a handful of
// non-zero terms: the per-chain Gaussians are narrow against the spread of the chains (measured on C3's shape:
// 0.01-0.5 % of the pairs have arg <= 176), only never for all 128 chains of a wavefront at once -- unless the
// 128 are NEIGHBOURS.  So, before a sweep over many chains:
//   1. the active chains are sorted by a coarse spatial key of their vectors (k_cull_stats / keys / scan / scatter:
//      CULL_KD evenly spaced dimensions, CULL_BITS bits each, bit-interleaved = Z-order; a counting sort);
//   2. each group of CULL_W consecutive chains of the sorted list gets its bounding box and its bound
//      (k_cull_boxes): 176 for the sum sweep (beyond it exp1 is exactly 0), the largest own-Gaussian arg of the
//      group for the min-arg sweep (every chain's minimum starts at or below it);
//   3. every (group, Q_i) pair is tested (k_cull_test, one Q_i per lane): lower bound of arg over the box =
//      sum over the key dimensions of dist_k^2 w_k, dist_k = max(lo_k - mu_k, mu_k - hi_k, 0), accumulated with the
//      very operations of the sweep (sub, mul, fma in ascending k).  Rounding is monotone, |mu_k - x_k| >= dist_k
//      for every chain of the group and the terms left out are >= 0, so the sweep's own float result is >= this
//      float bound: bound > limit  =>  arg > limit for all 128 chains, bit for bit.  One ballot = one mask word;
//   4. the sweep skips the excluded rows.  Results do not depend on the order of the chains or on what was skipped:
//      the bit-exact tests against the oracle (which knows nothing of this) are the proof.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ int cull_keydim(int c, int d) { return d >= CULL_KD ? c * (d / CULL_KD) : (c < d ? c : d - 1); }

// sums and sums of squares of the key dimensions over the active chains: st[0..KD) / st[KD..2KD)
static __global__ __launch_bounds__(BLOCK) void k_cull_stats(const float *__restrict__ xrows, const int *__restrict__ active, int nact,
                                                             int d, float *__restrict__ st)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  float v[CULL_KD];
#pragma unroll
  for (int c = 0; c < CULL_KD; ++c) v[c] = 0.0f;
  if (i < nact) {
    const int j = active ? active[i] : i;
#pragma unroll
    for (int c = 0; c < CULL_KD; ++c) v[c] = xrows[(size_t)j * d + cull_keydim(c, d)];
  }
  __shared__ float part[BLOCK / 64][2 * CULL_KD];
#pragma unroll
  for (int c = 0; c < CULL_KD; ++c) {
    float a = v[c], b = v[c] * v[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if ((threadIdx.x & 63u) == 0) {
      part[threadIdx.x >> 6][c] = a;
      part[threadIdx.x >> 6][CULL_KD + c] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * CULL_KD) {  // (same-address atomics are ~10 ns each: one per workgroup and counter)
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(st + threadIdx.x, t);
  }
}

// key of every active chain (bins of half a standard deviation over mean +- 2 sd, clamped) and the histogram
static __global__ __launch_bounds__(BLOCK) void k_cull_keys(const float *__restrict__ xrows, const int *__restrict__ active, int nact,
                                                            int d, const float *__restrict__ st, unsigned *__restrict__ keys,
                                                            unsigned *__restrict__ hist)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= nact) return;
  const int j = active ? active[i] : i;
  unsigned q[CULL_KD];
  const float inv = 1.0f / (float)nact;
#pragma unroll
  for (int c = 0; c < CULL_KD; ++c) {
    const float m = st[c] * inv, var = st[CULL_KD + c] * inv - m * m;
    const float sd = __builtin_sqrtf(var > 1e-30f ? var : 1e-30f);
    const float t = (xrows[(size_t)j * d + cull_keydim(c, d)] - (m - 2.0f * sd)) * ((float)(1 << CULL_BITS) / (4.0f * sd));
    q[c] = (unsigned)(t > 0.0f ? (t < (float)((1 << CULL_BITS) - 1) ? (int)t : (1 << CULL_BITS) - 1) : 0);  // NaN -> 0
  }
  unsigned key = 0;
#pragma unroll
  for (int b = CULL_BITS - 1; b >= 0; --b)
#pragma unroll
    for (int c = 0; c < CULL_KD; ++c) key = (key << 1) | ((q[c] >> b) & 1u);
  keys[i] = key;
  atomicAdd(hist + key, 1u);
}

// exclusive scan of the CULL_BINS counts, in place (one workgroup)
static __global__ __launch_bounds__(1024) void k_cull_scan(unsigned *__restrict__ hist)
{
  constexpr int PER = CULL_BINS / 1024;
  __shared__ unsigned part[1024];
  unsigned v[PER], s = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    v[u] = hist[threadIdx.x * PER + u];
    s += v[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - s;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    hist[threadIdx.x * PER + u] = run;
    run += v[u];
  }
}

// sorted[p] = chain; the order inside a bin is whatever the atomics give (no result depends on it)
static __global__ __launch_bounds__(BLOCK) void k_cull_scatter(const int *__restrict__ active, const unsigned *__restrict__ keys,
                                                               int nact, unsigned *__restrict__ offs, int *__restrict__ sorted)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= nact) return;
  const unsigned p = atomicAdd(offs + keys[i], 1u);
  sorted[p] = active ? active[i] : i;
}

// bounding box of every group of CULL_W consecutive positions in the CULL_KD key dimensions -- box[g][c] = (lo, hi)
// -- and the group's bound; one wavefront per group.  Only the key dimensions are boxed: 128 chains out of tens of
// thousands can be neighbours in a few dimensions at most, in the others their box spans nearly the whole
// population and contributes nothing to the bound (measured on C3's shape: 26 % of the (group, Q_i) pairs survive
// the bound over all 16 dimensions, 29 % the bound over the 4 key dimensions, at a quarter of the work).
template <int DMAX, bool SUMS>
__global__ __launch_bounds__(BLOCK) void k_cull_boxes(const float *__restrict__ xrows, const int *__restrict__ order, int nact,
                                                      const float *__restrict__ qpar, int own0, float *__restrict__ box,
                                                      float *__restrict__ lim, float *__restrict__ stats_done,
                                                      unsigned *__restrict__ hist_done)
{
  // the sort is over (this kernel runs behind k_cull_scatter): its sums and its histogram are left zero for the next one
  for (int i = (int)(blockIdx.x * BLOCK + threadIdx.x); i < CULL_BINS; i += (int)(gridDim.x * BLOCK)) hist_done[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < 2 * CULL_KD) stats_done[threadIdx.x] = 0.0f;
  const int g = (int)blockIdx.x * (BLOCK / 64) + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  if (g * CULL_W >= nact) return;
  float lo[CULL_KD], hi[CULL_KD];
#pragma unroll
  for (int c = 0; c < CULL_KD; ++c) { lo[c] = __builtin_inff(); hi[c] = -__builtin_inff(); }
  float worst = 0.0f;  // !SUMS: the largest own-Gaussian arg of the group
#pragma unroll
  for (int h = 0; h < CULL_W / 64; ++h) {
    const int pos = g * CULL_W + h * 64 + lane;
    if (pos < nact) {
      const int j = order[pos];
      const float *x = xrows + (size_t)j * DMAX;
#pragma unroll
      for (int c = 0; c < CULL_KD; ++c) {
        const float xk = x[cull_keydim(c, DMAX)];
        lo[c] = xk < lo[c] ? xk : lo[c];
        hi[c] = xk > hi[c] ? xk : hi[c];
      }
      if (!SUMS) {
        const float *qo = qpar + 2 * (size_t)(own0 + j) * DMAX;
        float a0 = 0.0f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
          const float xm = qo[2 * k] - x[k];
          a0 = __builtin_fmaf(xm * xm, qo[2 * k + 1], a0);
        }
        // a chain whose own arg is not an ordinary number (the sweep then starts its minimum from +inf) excludes nothing
        worst = (a0 < __builtin_inff()) ? (a0 > worst ? a0 : worst) : __builtin_inff();
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int c = 0; c < CULL_KD; ++c) {
      const float l2 = __shfl_xor(lo[c], o), h2 = __shfl_xor(hi[c], o);
      lo[c] = l2 < lo[c] ? l2 : lo[c];
      hi[c] = h2 > hi[c] ? h2 : hi[c];
    }
    if (!SUMS) {
      const float w2 = __shfl_xor(worst, o);
      worst = w2 > worst ? w2 : worst;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < CULL_KD; ++c) {
      box[(size_t)g * 2 * CULL_KD + 2 * c] = lo[c];
      box[(size_t)g * 2 * CULL_KD + 2 * c + 1] = hi[c];
    }
    // min-arg sweep: beyond ZERO_ARG an arg cannot matter either -- exp1(-arg / 2) is exactly 0 there, whatever
    // the minimum turns out to be (if every arg of a chain is beyond it, its cfac numerator is 0 both ways)
    lim[g] = SUMS ? ZERO_ARG : (worst < ZERO_ARG ? worst : ZERO_ARG);
  }
}

// excl[w][g] bit b = Q_{64 w + b} may matter to group g.  One Q_i per lane (its key dimensions in registers), the
// wavefront walks over a chunk of the groups, box and bound arrive through scalar loads.  The bound is the sweep's
// own accumulation restricted to the key dimensions (ascending k, the other terms -- all >= 0 -- left out):
// still <= the sweep's float result for every chain of the group.
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_cull_test(const float *__restrict__ qpar, int N, const float *__restrict__ box,
                                                     const float *__restrict__ lim, int ngroups, int nact, int gchunk,
                                                     unsigned long long *__restrict__ excl, int excl_words,
                                                     unsigned long long *__restrict__ nkept)
{
  const int w = (int)blockIdx.x * (BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // mask word
  const int i = w * 64 + (int)(threadIdx.x & 63u);
  if (w >= excl_words) return;
  const bool have = i < N;
  float mu[CULL_KD], wk[CULL_KD];
  const float2 *src = reinterpret_cast<const float2 *>(qpar + 2 * (size_t)(have ? i : 0) * DMAX);
#pragma unroll
  for (int c = 0; c < CULL_KD; ++c) {
    const float2 v = src[cull_keydim(c, DMAX)];
    mu[c] = v.x;
    wk[c] = v.y;
  }
  const int g0 = (int)blockIdx.y * gchunk, g1 = g0 + gchunk < ngroups ? g0 + gchunk : ngroups;
  unsigned long long kept = 0;
  for (int g = g0; g < g1; ++g) {
    const float *bx = box + (size_t)g * 2 * CULL_KD;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CULL_KD; ++c) {
      const float a = bx[2 * c] - mu[c], b = mu[c] - bx[2 * c + 1];
      const float dist = __builtin_fmaxf(__builtin_fmaxf(a, b), 0.0f);
      acc = __builtin_fmaf(dist * dist, wk[c], acc);
    }
    const unsigned long long m = __ballot(have && !(acc > lim[g]));  // (a NaN bound excludes nothing)
    if ((threadIdx.x & 63u) == 0) excl[(size_t)w * ngroups + g] = m;  // [word][group]: this wavefront's words are neighbours
    const int members = nact - g * CULL_W < CULL_W ? nact - g * CULL_W : CULL_W;
    kept += (unsigned long long)__popcll(m) * (unsigned long long)members;
  }
  if ((threadIdx.x & 63u) == 0 && kept) atomicAdd(nkept + ((blockIdx.x + blockIdx.y) & (CULL_NCOUNT - 1)), kept);
}

// np > 32: the chain vector does not fit the register budget; it is re-read from global memory
// (L1/L2-resident).  Same arithmetic and order as the register kernels.
__device__ __forceinline__ float q_arg_mem(const float *__restrict__ qp, const float *__restrict__ x, int d)
{
  float arg = 0.0f;
  for (int k = 0; k < d; ++k) {
    const float xm = qp[2 * k] - x[k];
    arg = __builtin_fmaf(xm * xm, qp[2 * k + 1], arg);
  }
  return arg;
}

static __global__ __launch_bounds__(BLOCK) void k_remote_cmax_big(const float *__restrict__ pvals,
                                                           const float *__restrict__ qpar,
                                                           float *__restrict__ cmax, int n, int d, int N)
{
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const float *x = pvals + (size_t)j * d;
  float amin = __builtin_inff();
  for (int qi = 0; qi < N; ++qi) {
    const float av = q_arg_mem(qpar + 2 * (size_t)qi * d, x, d);
    amin = av < amin ? av : amin;
  }
  cmax[j] = expf_v2(-0.5f * amin);
}

static __global__ __launch_bounds__(BLOCK) void k_remote_pass_big(const RemoteArgs a)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.nact) return;
  const int j = a.active_in ? a.active_in[i] : i;
  const int d = a.d;
  const uint32_t g = a.g0 + (uint32_t)j;
  const u32x4 w = philox4x32_10(a.t, g, (uint32_t)a.pass, 0u, a.seed, ST_RSEL);
  const int sel = (int)(((uint64_t)w.x * (uint64_t)a.N) >> 32);
  float *x = a.ptrial + (size_t)j * d;
  for (int qb = 0; 4 * qb < d; ++qb) {
    float z[4];
    normal4_from_words(philox4x32_10(a.t, g, (uint32_t)a.pass, (uint32_t)qb, a.seed, ST_RNORM), z);
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * qb + c;
      if (k < d) {
        const float m = a.musigall[2 * ((size_t)sel * d + k)];
        const float sg = __builtin_sqrtf(a.musigall[2 * ((size_t)sel * d + k) + 1]);
        x[k] = __builtin_fmaf(sg, z[c], m);
        a.mutrial[(size_t)j * d + k] = m;
        a.sigtrial[(size_t)j * d + k] = sg;
      }
    }
  }
  float qs = FPEPS, qm = FPEPS;
  for (int b0 = 0; b0 < a.N; b0 += QBLOCK) {  // blocked summation order, DESIGN.md §3.5
    float part = 0.0f;
    for (int qi = b0; qi < a.N && qi < b0 + QBLOCK; ++qi) {
      const float gv = expf_v2(-0.5f * q_arg_mem(a.winv + 2 * (size_t)qi * d, x, d));
      part = part + gv;
      qm = gv > qm ? gv : qm;
    }
    qs = qs + part;
  }
  const float pacpt = qm / qs;
  if (u24(w.y) < pacpt) {
    a.cfac[j] = a.cmax[j] / qm;
  } else {
    a.active_out[wave_slot(a.nact_out)] = j;
  }
}

// src/mcpar.cc:447-448
static __global__ void k_square(float *v, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = v[i] * v[i];
}

}  // namespace mcx
