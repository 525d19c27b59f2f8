// mcx_device.hpp -- gfx950 kernels of the chain-step hot path.
//
// Work decomposition: a chain's parameter vector is split into blocks of 4
// parameters; one lane owns one block, LPC = next_pow2(ceil(d/4)) adjacent lanes own one chain,
// 64/LPC chains share a wavefront.  With d % 4 == 0 the reference's own row-major chain-state
// matrix pvals[nchain][nparam] (src/mcpar.hh:63-65) is then read and written as one fully
// coalesced 16-byte access per lane, one Philox4x32 block feeds exactly one lane's four normals,
// and the likelihood / acceptance reductions are xor-butterflies over LPC lanes.
//
// Reference functions restated here:
//   genLocal            src/mcpar.cc:302-312   -> propose_block()
//   VLFunc builtins     src/rosenbrock.cc      -> Lik<...>
//   accept/reject       src/mcpar.cc:62-75,162-175 -> accept_decision()
//   Welford + publish   src/mcpar.cc:184-209   -> welford_block(), publish
//   genRemote           src/mcpar.cc:315-451   -> k_remote_*
//   burn-in tuner       src/mcpar.cc:77-96     -> k_tuner
#pragma once
#include "mcx_numerics.hpp"

namespace mcx {

constexpr int BLOCK = 256;  // 4 wavefronts per workgroup
constexpr int MAXD_LDS = 32;  // full-covariance factor is staged in LDS up to this np, read from L2 above
constexpr int MAXD = 256;     // lanes per chain <= 64

// LIK_USER: a user's own likelihood, HIP source compiled INTO these kernels at run time (MCX_VL_SOURCE, mcx_user.hip);
// its code paths exist only in that translation unit (MCX_USER_LIK defined)
enum LikKind : int { LIK_ROSEN1 = 1, LIK_ROSEN2 = 2, LIK_GAUSS = 3, LIK_MIX = 5, LIK_ROSEN2F = 6, LIK_USER = 7 };

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the LPC lanes of a chain, every lane gets the result.  Pairing = xor-butterfly over the
// block index.  Up to 16 lanes it is done with DPP row operations: after the two
// quad steps every lane of a quad holds the quad's sum, so the mirror steps add the other quad's /
// other half-row's sum -- the same pairs as xor 4 / xor 8 (addition commutes).  Wider groups finish
// with LDS-crossbar shuffles.
template <int LPC>
__device__ __forceinline__ float group_sum(float p)
{
  if (LPC >= 2) p = p + dpp_mov<0xB1>(p);    // quad_perm [1,0,3,2]  == xor 1
  if (LPC >= 4) p = p + dpp_mov<0x4E>(p);    // quad_perm [2,3,0,1]  == xor 2
  if (LPC >= 8) p = p + dpp_mov<0x141>(p);   // row_half_mirror      ~  xor 4
  if (LPC >= 16) p = p + dpp_mov<0x140>(p);  // row_mirror           ~  xor 8
#pragma unroll
  for (int s = 16; s < LPC; s <<= 1) p = p + __shfl_xor(p, s);
  return p;
}

// ---------------------------------------------------------------------------------------------
// Likelihood block partials.  xb = this lane's 4 parameters, nv = how many are real, k0 = index
// of the first one.  lik = device parameter block.
// ---------------------------------------------------------------------------------------------
template <int LIK, int LPC>
struct Lik;

template <int LPC>
struct Lik<LIK_ROSEN1, LPC> {  // src/rosenbrock.cc:4-21
  __device__ __forceinline__ void init(const float *, int, int, int, int) {}
  __device__ __forceinline__ float eval(const float xb[4], int nv) const
  {
    float acc = 0.0f;
    if (nv >= 2) {
      const float t1 = 1.0f - xb[0];
      const float t2 = __builtin_fmaf(-xb[0], xb[0], xb[1]);
      acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
    }
    if (nv >= 4) {
      const float t1 = 1.0f - xb[2];
      const float t2 = __builtin_fmaf(-xb[2], xb[2], xb[3]);
      acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
    }
    return 0.0f - group_sum<LPC>(acc);  // 0 - s like the reference's fx = 0; fx -= ...: never -0
  }
};

// The overlapping Rosenbrock function made well-posed (MCX_VL_ROSENBROCK2_FIXED: src/rosenbrock.cc:25-41 with '+' at
// :38 and the loop kept inside the set): term k = (1 - x_k)^2 + 100 (x_{k+1} - x_k^2)^2 for k < d - 1 belongs to
// block k / 4; the block's last term takes x_{k+1} from the next lane of the chain.
template <int LPC>
struct Lik<LIK_ROSEN2F, LPC> {
  int d, k0;
  __device__ __forceinline__ void init(const float *, int d_, int k0_, int, int) { d = d_; k0 = k0_; }
  __device__ __forceinline__ float eval(const float xb[4], int) const
  {
    const float nxt = __shfl_down(xb[0], 1);  // (every lane of the wavefront is here; the chain's last block ignores it)
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (k0 + c + 1 < d) {
        const float a = xb[c], b = c < 3 ? xb[c < 3 ? c + 1 : 0] : nxt;
        const float t1 = 1.0f - a;
        const float t2 = __builtin_fmaf(-a, a, b);
        acc = acc + __builtin_fmaf(100.0f * t2, t2, t1 * t1);
      }
    return 0.0f - group_sum<LPC>(acc);
  }
};

template <int LPC>
struct Lik<LIK_GAUSS, LPC> {  // src/rosenbrock.cc:44-61; lik = mu[d], s2inv[d]
  float mu[4], si[4];
  __device__ __forceinline__ void init(const float *lik, int d, int k0, int nv, int)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mu[k] = k < nv ? lik[k0 + k] : 0.0f;
      si[k] = k < nv ? lik[d + k0 + k] : 0.0f;
    }
  }
  __device__ __forceinline__ float eval(const float xb[4], int nv) const
  {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) {
        const float a = xb[k] - mu[k];
        acc = __builtin_fmaf((0.5f * a) * a, si[k], acc);
      }
    return 0.0f - group_sum<LPC>(acc);  // 0 - s like the reference's fx = 0; fx -= ...: never -0
  }
};

template <int LPC>
struct Lik<LIK_MIX, LPC> {  // log sum_c w_c exp(-|x-m_c|^2/2); lik = means[K*d], logw[K]
  const float *means, *logw;
  int K, d, k0;
  __device__ __forceinline__ void init(const float *lik, int d_, int k0_, int, int K_)
  {
    means = lik; K = K_; d = d_; k0 = k0_;
    logw = lik + (size_t)K_ * d_;
  }
  __device__ __forceinline__ float comp(const float xb[4], int nv, int c) const
  {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) {
        const float a = xb[k] - means[c * d + k0 + k];
        acc = __builtin_fmaf(a, a, acc);
      }
    return __builtin_fmaf(-0.5f, group_sum<LPC>(acc), logw[c]);
  }
  __device__ __forceinline__ float eval(const float xb[4], int nv) const
  {
    if (K <= 8) {  // keep the component exponents in registers (same values, one evaluation each)
      float e[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) e[c] = c < K ? comp(xb, nv, c) : 0.0f;
      float emax = e[0];
#pragma unroll
      for (int c = 1; c < 8; ++c)
        if (c < K) emax = e[c] > emax ? e[c] : emax;
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < K) s = s + expf_v2(e[c] - emax);
      return emax + logf_v1(s);
    }
    float emax = comp(xb, nv, 0);
    for (int c = 1; c < K; ++c) {
      const float e = comp(xb, nv, c);
      emax = e > emax ? e : emax;
    }
    float s = 0.0f;
    for (int c = 0; c < K; ++c) s = s + expf_v2(comp(xb, nv, c) - emax);
    return emax + logf_v1(s);
  }
};

#ifdef MCX_USER_LIK
// The user's likelihood (src/vlfunc.hh:9-12 as device functions in the global namespace, include/mcx.h MCX_VL_SOURCE).
//   MCX_USER_LIK == 1: log L = mcx_user_finish(sum over the 4-parameter blocks of mcx_user_block(...)) -- the partials are
//     added in the MCX order (xor-butterfly over the block index), so a restatement of a built-in gives its bits;
//   MCX_USER_LIK == 2: log L = mcx_user_loglike(x[d], d, par) on the whole parameter vector, staged through LDS; every
//     lane of the chain evaluates it (a wavefront's instruction costs the same with one lane active or all).
template <int LPC>
struct Lik<LIK_USER, LPC> {
  const float *par;
  int d, k0;
  __device__ __forceinline__ void init(const float *lik, int d_, int k0_, int, int) { par = lik; d = d_; k0 = k0_; }
  __device__ __forceinline__ float eval(const float xb[4], int nv) const
  {
#if MCX_USER_LIK == 1
    const float acc = nv > 0 ? ::mcx_user_block(xb, nv, k0, d, par) : 0.0f;
    return ::mcx_user_finish(group_sum<LPC>(acc), d, par);
#else
    // the chain's vector, contiguous (lane q holds x[4q .. 4q+3]); an odd stride between chains: the lanes of a wavefront
    // read x[k] of different chains from different banks
    constexpr int CS = 4 * LPC + 1;
    __shared__ float xs[(BLOCK / LPC) * CS];
    float *mine = xs + ((int)threadIdx.x / LPC) * CS;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) mine[k0 + k] = xb[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // a chain never spans wavefronts; LDS is in order per wavefront
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const float v = ::mcx_user_loglike(mine, d, par);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // this call's reads precede the next call's writes
    return v;
#endif
  }
};
#endif

// ---------------------------------------------------------------------------------------------
// block load / store helpers: row-major [n][d], this lane's block at column k0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_block(const float *__restrict__ base, size_t row, int d,
                                           int k0, int nv, bool vec4, float v[4])
{
  const float *p = base + row * (size_t)d + k0;
  if (vec4) {
    if (nv == 4) {
      const float4 f = *reinterpret_cast<const float4 *>(p);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
      v[0] = v[1] = v[2] = v[3] = 0.0f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = k < nv ? p[k] : 0.0f;
  }
}

__device__ __forceinline__ void store_block(float *__restrict__ base, size_t row, int d, int k0,
                                            int nv, bool vec4, const float v[4])
{
  float *p = base + row * (size_t)d + k0;
  if (vec4) {
    if (nv == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) p[k] = v[k];
  }
}

// (mu, sig^2) interleaved pairs: musigall slot layout of src/mcpar.cc:205-208
__device__ __forceinline__ void store_pairs(float *__restrict__ base, size_t row, int d, int k0,
                                            int nv, bool vec4, const float a[4], const float b[4])
{
  float *p = base + 2 * (row * (size_t)d + k0);
  if (vec4) {
    if (nv == 4) {
      reinterpret_cast<float4 *>(p)[0] = make_float4(a[0], b[0], a[1], b[1]);
      reinterpret_cast<float4 *>(p)[1] = make_float4(a[2], b[2], a[3], b[3]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) { p[2 * k] = a[k]; p[2 * k + 1] = b[k]; }
  }
}

// ---------------------------------------------------------------------------------------------
// genLocal for one block (src/mcpar.cc:302-312): pt = x + T z, T lower triangular (Cholesky
// factor, scaled by the tuner).  diag: T is diagonal (identity covariance, any tuner history).
// Tl = T staged in LDS (full path only).
// ---------------------------------------------------------------------------------------------
// zbuf: per-workgroup LDS scratch of BLOCK float4 (or null): with T in LDS and d % 4 == 0 the lane
// group exchanges z through it (one b128 write, broadcast b128 reads) and reads T four columns at a
// time; otherwise z goes through LDS-crossbar shuffles and T is read element by element.
template <int LPC>
__device__ __forceinline__ void propose_block(const float x[4], float pt[4], const float tdiag[4],
                                              const float *Tl, bool diag, int d, int q, int nv,
                                              uint32_t t, uint32_t g, uint32_t seed, float *zbuf = nullptr,
                                              int ldt = 0)
{
  if (ldt == 0) ldt = d;  // row stride of Tl (LDS copy is padded against bank conflicts)
  float z[4];
  {
    f32x2 ze, zo;  // packed Box-Muller: same bits as normal4_from_words, half the instructions
    normal4_packed(philox4x32_10(t, g, (uint32_t)q, 0u, seed, ST_LOCAL), ze, zo);
    z[0] = ze.x; z[1] = zo.x; z[2] = ze.y; z[3] = zo.y;
  }
  if (diag) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pt[k] = __builtin_fmaf(tdiag[k], z[k], x[k]);
  } else if (zbuf) {
    float4 *zb = reinterpret_cast<float4 *>(zbuf);
    zb[threadIdx.x] = make_float4(z[0], z[1], z[2], z[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // a lane group never spans wavefronts; LDS is in order per wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int base = (int)threadIdx.x & ~(LPC - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) pt[k] = x[k];
#pragma unroll
    for (int qq = 0; qq < LPC; ++qq) {
      if (qq <= q && nv == 4) {  // columns j = 4qq .. 4qq+3 in ascending order; entries above the
        const float4 zz = zb[base + qq];  // diagonal are exact zeros, i.e. exact no-ops
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 tr = *reinterpret_cast<const float4 *>(&Tl[(4 * q + k) * ldt + 4 * qq]);
          pt[k] = __builtin_fmaf(tr.x, zz.x, pt[k]);
          pt[k] = __builtin_fmaf(tr.y, zz.y, pt[k]);
          pt[k] = __builtin_fmaf(tr.z, zz.z, pt[k]);
          pt[k] = __builtin_fmaf(tr.w, zz.w, pt[k]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // all reads of this step precede the next step's write
  } else {
    const int lane0 = (int)(threadIdx.x & 63u) & ~(LPC - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) pt[k] = x[k];
#pragma unroll
    for (int qq = 0; qq < LPC; ++qq) {
      float zz[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) zz[c] = __shfl(z[c], lane0 + qq);
      if (qq <= q) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < nv) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (4 * qq + c < d) pt[k] = __builtin_fmaf(Tl[(4 * q + k) * ldt + 4 * qq + c], zz[c], pt[k]);
          }
      }
    }
  }
}

// accept test of a Murray step (src/mcpar.cc:166-169): u < exp(ly' - ly) * cfac
__device__ __forceinline__ bool accept_decision(float lytrial, float ly, float cfac, uint32_t word)
{
  const float pacpt = expf_v2(lytrial - ly) * cfac;
  return u24(word) < pacpt;
}
// accept test of a local step (cfac = 1; src/mcpar.cc:66-69, 166-169) in the log domain: lu = accept_lu(word)
__device__ __forceinline__ bool accept_local(float lytrial, float ly, float lu) { return lu < lytrial - ly; }

// Welford update of one block (src/mcpar.cc:199-202)
__device__ __forceinline__ void welford_block(const float x[4], float mu[4], float ps[4],
                                              float winv)
{
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float delta = x[k] - mu[k];
    mu[k] = __builtin_fmaf(delta, winv, mu[k]);
    ps[k] = __builtin_fmaf(delta, x[k] - mu[k], ps[k]);
  }
}

// ---------------------------------------------------------------------------------------------
// Fused multi-step kernel for local steps: burn-in loop body (src/mcpar.cc:58-75) or main loop
// body (src/mcpar.cc:152-209) for nsteps consecutive steps, chain state held in registers.
// ---------------------------------------------------------------------------------------------
struct SegArgs {
  float *x, *ly, *mu, *psum2;
  uint32_t *acc_cnt;
  uint32_t *acc_slots;  // [number of wavefronts] accepted proposals, one plain slot per wavefront
  const float *T;
  float *samp_x, *samp_ly;  // sample store rows of the segment's first step (stride 1) / of the run's
                            // step 0 (stride > 1), or null
  int samp_stride;          // keep main-loop steps with isamp % samp_stride == 0
  uint8_t *mask;            // accept mask row of the segment's first step, or null
  const float *lik;
  int ncomp;
  int n, d, nsteps;
  uint32_t g0, t0, seed;
  int isamp0;
  int diag, vec4;
  const float *winv;  // winv[i] = 1/(i+1), i = main-loop step index (host-computed, correctly rounded)
  float *musig_own;   // this shard's musigall slot, written after local step snap_after (or never: -1)
  int snap_after;
  const float *zpre;  // pre-generated normals Z[nsteps][n][d] and accept thresholds U[nsteps][n] of this
  const float *upre;  // launch (k_gen_normals), or null: small-n mode, see k_fused_fast<..., PREGEN>
  float *trash;       // PREGEN: 16 B per lane where lanes that own no parameters dump their stores
  int init_moments;   // main loop: start from mu = 0, psum2 = FPEPS (src/mcpar.cc:99-104) instead of loading them
  float *sig_out;     // with the snapshot after step snap_after: the variances psum2 / (steps so far) as well, or null
  // The burn-in tuner (src/mcpar.cc:77-96) at the end of this launch, by the workgroup that finishes last -- on the
  // headline job the ten one-block k_tuner launches and the gaps around them were 4 % of the run.  on = 0: not here;
  // on = 2 (main loop): the same count, added to the run's accepted main-loop proposals (ctr[4]).
  struct Tuner {
    int on, check, ncov, nslots;
    unsigned long long *ctr;
    unsigned long long add_trials;
    float *T, *trace;
    int *ntrace;
    unsigned long long *cells;  // TUN_CELLS + 1 words (left at 0): finished workgroups << 40 | their accepted proposals
    float armin, armax, dfac, ifac;
  } tun;
};

// sum (and clear) the per-wavefront accept slots: one block
__device__ __forceinline__ unsigned long long block_sum_slots(uint32_t *slots, int nslots)
{
  __shared__ unsigned long long red[BLOCK / 64];
  unsigned long long v = 0;
  for (int i = threadIdx.x; i < nslots; i += blockDim.x) {
    v += slots[i];
    slots[i] = 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  __syncthreads();
  return t;
}

// src/mcpar.cc:77-96, by one workgroup: the segment's accepted proposals (the wavefronts' slots) join the counters;
// at a check the acceptance rate since the last rescale decides on a rescale of the factor
__device__ __forceinline__ void tuner_block(unsigned long long seg, unsigned long long *ctr, float *T, int ncov,
                                            unsigned long long add_trials, int check, float armin, float armax, float dfac,
                                            float ifac, float *trace, int *ntrace)
{
  __shared__ float fac;
  if (threadIdx.x == 0) {
    unsigned long long na = ctr[1] + seg, nt = ctr[2] + add_trials;
    ctr[3] += seg;
    float f = 1.0f;
    if (check) {
      const float arate = (float)na / (float)nt;
      if (arate < armin) { na = nt = 0; f = dfac; }
      else if (arate > armax) { na = nt = 0; f = ifac; }
    }
    ctr[1] = na;
    ctr[2] = nt;
    fac = f;
  }
  __syncthreads();
  if (check) {
    const float f = fac;
    if (f != 1.0f)
      for (int i = threadIdx.x; i < ncov; i += blockDim.x) T[i] *= f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int k = *ntrace;
      if (k < 256) trace[k] = T[0];
      *ntrace = k + 1;
    }
  }
}

// End of a fused burn-in launch with SegArgs::tun.on.  No wavefront has written its accept slot: every workgroup adds
// (1 << 40 | its accepted proposals) to one of TUN_CELLS words, the workgroup that completes a word carries the word's
// sum to the root word the same way, and the one that completes the root has the segment's total IN HAND -- nothing
// that another workgroup wrote has to be visible to it, so no fences (a device-scope release is a write-back of the L2,
// which holds the whole chain state at that point: +20 us per launch when tried) -- and is the tuner; nobody reads the
// factor any more.  Same-address atomics cost ~10 ns each on this chip: 64 per word instead of 1024 on one.
// Reached by EVERY thread of the workgroup (wacc = 0 for wavefronts without chains).
constexpr int TUN_CELLS = 16;
__device__ __forceinline__ void tuner_epilogue(const SegArgs &a, uint32_t wacc)
{
  if (!a.tun.on) return;
  constexpr unsigned long long ONE = 1ull << 40, SUM = ONE - 1ull;
  __shared__ unsigned part[BLOCK / 64];
  __shared__ unsigned long long lds_seg;
  __shared__ unsigned is_last;
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = wacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned last = 0;
    unsigned long long mine = ONE;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) mine += part[w];
    const unsigned cell = blockIdx.x % TUN_CELLS, ncell = gridDim.x < TUN_CELLS ? gridDim.x : TUN_CELLS;
    const unsigned members = (gridDim.x - cell + TUN_CELLS - 1) / TUN_CELLS;  // workgroups with blockIdx % TUN_CELLS == cell
    const unsigned long long old = atomicAdd(a.tun.cells + cell, mine);
    if ((unsigned)(old >> 40) + 1u == members) {
      (void)atomicExch(a.tun.cells + cell, 0ull);  // (complete: nobody adds to it again in this launch)
      const unsigned long long up = ONE | ((old + mine) & SUM);
      const unsigned long long oldr = atomicAdd(a.tun.cells + TUN_CELLS, up);
      if ((unsigned)(oldr >> 40) + 1u == ncell) {
        (void)atomicExch(a.tun.cells + TUN_CELLS, 0ull);
        lds_seg = (oldr + up) & SUM;
        last = 1;
      }
    }
    is_last = last;
  }
  __syncthreads();
  if (is_last) {
    if (a.tun.on == 1)
      tuner_block(lds_seg, a.tun.ctr, a.tun.T, a.tun.ncov, a.tun.add_trials, a.tun.check, a.tun.armin, a.tun.armax,
                  a.tun.dfac, a.tun.ifac, a.tun.trace, a.tun.ntrace);
    else if (threadIdx.x == 0)
      a.tun.ctr[4] += lds_seg;  // main loop: accepted proposals of the run so far (what k_reduce_slots adds up otherwise)
  }
}

template <int LPC, int LIK, bool MAIN>
__device__ __forceinline__ void fused_steps_body(const SegArgs &a);

template <int LPC, int LIK, bool MAIN>
__global__ __launch_bounds__(BLOCK) void k_fused_steps(const SegArgs a)
{
  fused_steps_body<LPC, LIK, MAIN>(a);
}

template <int LPC, int LIK, bool MAIN>
__device__ __forceinline__ void fused_steps_body(const SegArgs &a)
{
  // rows padded by 4 floats: the 4-row blocks of different lanes then start 16 banks apart instead of
  // on the same bank (8-way -> 2-way conflict at d = 32) and stay 16-byte aligned
  __shared__ __attribute__((aligned(16))) float Tlds[MAXD_LDS * (MAXD_LDS + 4)];
  __shared__ __attribute__((aligned(16))) float zlds[BLOCK * 4];
  const bool diag = a.diag != 0, vec4 = a.vec4 != 0;
  const int d = a.d;
  const float *Tl = a.T;
  float *zbuf = nullptr;
  int ldt = d;
  if (!diag && d <= MAXD_LDS) {
    ldt = d + 4;
    for (int i = threadIdx.x; i < d * d; i += BLOCK) Tlds[(i / d) * ldt + (i % d)] = a.T[i];
    __syncthreads();
    Tl = Tlds;
    if (vec4) zbuf = zlds;
  }
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t chain = gid / LPC;
  const int q = (int)(gid % LPC);
  if (chain >= (size_t)a.n) return;
  const int k0 = 4 * q;
  const int nv = d - k0 >= 4 ? 4 : (d - k0 > 0 ? d - k0 : 0);
  const uint32_t g = a.g0 + (uint32_t)chain;

  float x[4], mu[4], ps[4], tdiag[4];
  load_block(a.x, chain, d, k0, nv, vec4, x);
  if (MAIN) {
    load_block(a.mu, chain, d, k0, nv, vec4, mu);
    load_block(a.psum2, chain, d, k0, nv, vec4, ps);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) tdiag[k] = (diag && k < nv) ? a.T[(k0 + k) * d + k0 + k] : 0.0f;
  float ly = a.ly[chain];
  Lik<LIK, LPC> L;
  L.init(a.lik, d, k0, nv, a.ncomp);

  uint32_t cnt = 0, wacc = 0;
  float alu[4] = {0, 0, 0, 0};  // log of the four acceptance draws of the current ACCEPT block
  uint32_t ablk = 0xffffffffu;
  float winv = 1.0f;

  for (int s = 0; s < a.nsteps; ++s) {
    const uint32_t t = a.t0 + (uint32_t)s;
    float pt[4];
    propose_block<LPC>(x, pt, tdiag, Tl, diag, d, q, nv, t, g, a.seed, zbuf, ldt);
    const float lyt = L.eval(pt, nv);
    if ((t >> 2) != ablk) {  // one Philox block serves four consecutive steps
      ablk = t >> 2;
      const u32x4 aw = philox4x32_10(ablk, g, 0u, 0u, a.seed, ST_ACCEPT);
      const f32x2 l01 = accept_lu_x2(aw.x, aw.y), l23 = accept_lu_x2(aw.z, aw.w);
      alu[0] = l01.x; alu[1] = l01.y; alu[2] = l23.x; alu[3] = l23.y;
    }
    const uint32_t wi = t & 3u;
    const bool take = accept_local(lyt, ly, wi == 0u ? alu[0] : (wi == 1u ? alu[1] : (wi == 2u ? alu[2] : alu[3])));
    if (take) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = pt[k];
      ly = lyt;
      cnt += 1;
    }
    wacc += (uint32_t)__popcll(__ballot(take && q == 0));
    if (a.mask && q == 0) a.mask[(size_t)s * a.n + chain] = take ? 1 : 0;
    if (MAIN) {
      const float pwgt = (float)(a.isamp0 + s + 1);  // src/mcpar.cc:186-187
      winv = 1.0f / pwgt;
      welford_block(x, mu, ps, winv);
      if (s == a.snap_after) {  // the (mu, sig^2) this shard ships at the next exchange (src/mcpar.cc:202-208)
        float sg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[k] = ps[k] * winv;
        store_pairs(a.musig_own, chain, d, k0, nv, vec4, mu, sg);
      }
      if (a.samp_x) {  // src/mcpar.cc:177-182
        size_t row = (size_t)s;
        bool keep = true;
        if (a.samp_stride > 1) {
          const int is = a.isamp0 + s;
          keep = is % a.samp_stride == 0;
          row = (size_t)(is / a.samp_stride);
        }
        if (keep) {
          store_block(a.samp_x, row * a.n + chain, d, k0, nv, vec4, x);
          if (q == 0) a.samp_ly[row * a.n + chain] = ly;
        }
      }
    }
  }

  store_block(a.x, chain, d, k0, nv, vec4, x);
  if (q == 0) {
    a.ly[chain] = ly;
    a.acc_cnt[chain] += cnt;
  }
  if (MAIN) {
    store_block(a.mu, chain, d, k0, nv, vec4, mu);
    store_block(a.psum2, chain, d, k0, nv, vec4, ps);
  }
  // one slot per wavefront, owned by it: no atomics (4096 same-address atomics cost ~40 us per launch)
  if ((threadIdx.x & 63u) == 0 && wacc) a.acc_slots[gid >> 6] += wacc;
}


// ---------------------------------------------------------------------------------------------
// Hot-path specialisation of k_fused_steps: Rosenbrock1, diagonal Cholesky factor (identity
// covariance under any tuner history), d % 4 == 0, no accept-mask recording.  Same arithmetic as the
// generic kernel, restructured for gfx950's issue limits:
//   * the lane's four parameters are held as two packed pairs E = (x0, x2), O = (x1, x3); Box-Muller,
//     the proposal, the Rosenbrock terms and the Welford update run on v_pk_{fma,mul,add}_f32;
//   * the lane-group reductions are DPP row operations instead of LDS-crossbar shuffles;
//   * 1/pwgt comes from a host-built table through a scalar load instead of a VALU division.
// ---------------------------------------------------------------------------------------------
// Random numbers of nsteps consecutive local steps for every chain of the shard (small-n mode):
// Z[s][chain][k] = the normal of parameter k at step t0+s (LOCAL stream), U[s][chain] = the log of the
// acceptance draw (ACCEPT stream; one Philox block serves steps 4b..4b+3, drawn by the lane of the first of
// them that lies in this launch).  One lane per (step, chain, 4-parameter block): fully parallel.
template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_gen_normals(float *__restrict__ Z, float *__restrict__ U, int n,
                                                       int d, int nsteps, uint32_t t0, uint32_t g0,
                                                       uint32_t seed)
{
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t per_step = (size_t)n * LPC;
  const int s = (int)(gid / per_step);
  if (s >= nsteps) return;
  const size_t rem = gid - (size_t)s * per_step;
  const size_t chain = rem / LPC;
  const int q = (int)(rem % LPC);
  const uint32_t t = t0 + (uint32_t)s, g = g0 + (uint32_t)chain;
  if (4 * q < d) {
    f32x2 ze, zo;
    normal4_packed(philox4x32_10(t, g, (uint32_t)q, 0u, seed, ST_LOCAL), ze, zo);
    *reinterpret_cast<float4 *>(Z + ((size_t)s * n + chain) * d + 4 * q) = make_float4(ze.x, zo.x, ze.y, zo.y);
  }
  if (q == 0 && ((t & 3u) == 0u || s == 0)) {
    const u32x4 aw = philox4x32_10(t >> 2, g, 0u, 0u, seed, ST_ACCEPT);
    for (uint32_t w = t & 3u; w < 4u; ++w) {
      const int s2 = s + (int)(w - (t & 3u));
      if (s2 < nsteps) U[(size_t)s2 * n + chain] = accept_lu(pick_word(aw, w));
    }
  }
}

// value of lane `owner` of this lane's LPC-group (owner is wave-uniform), by DPP
template <int LPC>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, uint32_t owner, int q)
{
  if (LPC == 1) return v;
  const int iv = (int)v;
  int r = iv;
  if (LPC == 2) {
    r = (owner & 1u) ? __builtin_amdgcn_update_dpp(0, iv, 0xF5, 0xF, 0xF, true)   // quad_perm [1,1,3,3]
                     : __builtin_amdgcn_update_dpp(0, iv, 0xA0, 0xF, 0xF, true);  // quad_perm [0,0,2,2]
    return (uint32_t)r;
  }
  switch (owner & 3u) {
  case 0: r = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, true); break;
  case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, true); break;
  case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, true); break;
  default: r = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, true); break;
  }
  if (LPC >= 8) {  // the owner's quad holds the value; the other quad of the 8-lane group mirrors it in
    const int other = __builtin_amdgcn_update_dpp(0, r, 0x141, 0xF, 0xF, true);  // row_half_mirror
    r = ((int)(owner >> 2) == (q >> 2)) ? r : other;
  }
  return (uint32_t)r;
}

// PREGEN: the normals and accept thresholds of the launch were produced beforehand by k_gen_normals
// (same functions, same bits) and are streamed in with a 4-step register prefetch ring.  With few
// chains the fused kernel is bound by the latency of one wave's instruction stream, two thirds of which
// is the random-number work -- which does not depend on the chain state and can run on the idle SIMDs.
//
// FULL: proposals x' = x + T z with the full lower-triangular Cholesky factor (src/mcpar.cc:302-312 with
// covar_setup's factor, :454-484).  A lane needs its four rows of T and the whole z of its chain:
//   * T is staged in LDS as [column block qq][column c][lane of the chain q][rows 0, 2, 1, 3 of the lane]: for a given
//     column the lanes of a wavefront read LPC consecutive 16-byte slots -- every ds_read_b128 lane group sees each
//     slot's address on all its readers (broadcast) and no bank twice -- and one read is the pair of packed operands
//     (rows 0, 2 | rows 1, 3) of the two v_pk_fma_f32 that column costs (round 2 read rows and issued scalar
//     multiply-adds plus the moves that packed them: 741 -> ~600 instructions per wavefront-step at 32-D);
//   * z travels by DPP quad permutes for chains of <= 4 lanes (no memory: 4 moves per column block, the multiply-adds
//     packed by the compiler; feeding z as the DPP operand of v_fmac_f32 through inline asm measured the same); chains of 8 lanes write
//     their z block to LDS (9 slots per chain, so the four chains of a ds_read_b128 lane group sit on
//     different banks) and read the 8 blocks back;
//   * every lane runs all columns, in ascending order like the oracle's loop over k <= i: the entries above
//     the diagonal are exact zeros and fma(0, z, acc) == acc (z is finite; acc = -0 would need x = -0).
// value of lane qq of this lane's chain (2 or 4 lanes per chain; qq is a compile-time constant after unrolling): one DPP move
template <int LPC>
__device__ __forceinline__ float quad_bcast(float v, int qq)
{
  const int iv = (int)as_u32(v);
  int r;
  if (LPC == 2) {
    r = qq ? __builtin_amdgcn_update_dpp(0, iv, 0xF5, 0xF, 0xF, true)   // quad_perm [1,1,3,3]
           : __builtin_amdgcn_update_dpp(0, iv, 0xA0, 0xF, 0xF, true);  // quad_perm [0,0,2,2]
  } else {
    switch (qq) {
    case 0: r = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, true); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, true); break;
    }
  }
  return as_f32((uint32_t)r);
}

template <int LPC, bool MAIN, int LIK, bool PREGEN, bool FULL>
__device__ __forceinline__ uint32_t fused_fast_body(const SegArgs &a);

// (the body returns early for threads without a chain, with the wavefront's accepted proposals otherwise; the tuner
// at the end is every thread's)
template <int LPC, bool MAIN, int LIK = LIK_ROSEN1, bool PREGEN = false, bool FULL = false>
__global__ __launch_bounds__(BLOCK) void k_fused_fast(const SegArgs a)
{
  const uint32_t wacc = fused_fast_body<LPC, MAIN, LIK, PREGEN, FULL>(a);
  tuner_epilogue(a, wacc);
}

template <int LPC, bool MAIN, int LIK, bool PREGEN, bool FULL>
__device__ __forceinline__ uint32_t fused_fast_body(const SegArgs &a)
{
  static_assert(LIK == LIK_ROSEN1 || LIK == LIK_GAUSS || LIK == LIK_MIX || (LIK == LIK_ROSEN2F && !PREGEN && !FULL) || (LIK == LIK_USER && !PREGEN),
                "fast path: Rosenbrock1, diagonal Gaussian, a mixture of <= 8 unit Gaussians, (plain kernel only) the overlapping Rosenbrock, or a user's source");
  static_assert(!(FULL && PREGEN), "the pre-generated normals are laid out for diagonal proposals");
  // MCX_FULL_T_REGS: up to 16-D (<= 64 registers) a lane's four rows of the factor live in REGISTERS for the whole launch
  // instead of being re-read from LDS every step: 16-D 2.78 -> 2.73 ms per job (tools/fullcov_ab.sh).  At 32-D the same
  // takes 128 registers -- two wavefronts per SIMD instead of four -- and LOSES: 6.31 -> 6.91 ms, although 32 of the
  // lane's 41 LDS reads per step go away; there the factor stays in LDS and the work is cut by k_fused_fastb's mirrored
  // layout instead (mcx_fastb.hpp).  0: the factor in LDS everywhere (the round-2 kernel), for A/B.
#ifndef MCX_FULL_T_REGS
#define MCX_FULL_T_REGS 1
#endif
#ifndef MCX_FAST_UNROLL4
#define MCX_FAST_UNROLL4 1
#endif
  constexpr bool TREGS = FULL && LPC <= 4 && (MCX_FULL_T_REGS != 0);
  __shared__ __attribute__((aligned(16))) float4 lds_T[FULL && !TREGS ? 4 * LPC * LPC : 1];
  __shared__ __attribute__((aligned(16))) float4 lds_z[FULL && LPC == 8 ? (BLOCK / 8) * 9 : 1];
  if (FULL && !TREGS) {
    const int dd = a.d;
    // slot [(qq * 4 + c) * LPC + qv] = column 4 qq + c of the four rows of lane qv, as (row 0, row 2, row 1, row 3):
    // the two halves are the packed operands of the lane's (x0, x2) / (x1, x3) accumulators
    for (int i = threadIdx.x; i < 16 * LPC * LPC; i += BLOCK) {
      const int h = i & 3, qv = (i >> 2) % LPC, c = ((i >> 2) / LPC) & 3, qq = (i >> 2) / (4 * LPC);
      const int row = 4 * qv + (h == 0 ? 0 : (h == 1 ? 2 : (h == 2 ? 1 : 3))), col = 4 * qq + c;
      reinterpret_cast<float *>(lds_T)[i] = (row < dd && col < dd) ? a.T[row * dd + col] : 0.0f;
    }
    if (LIK != LIK_MIX) __syncthreads();
  }
  // mixture: component means and log-weights staged in LDS (every lane group reads the same rows)
  __shared__ __attribute__((aligned(16))) float lds_means[LIK == LIK_MIX ? 8 * MAXD_LDS : 4];
  __shared__ float lds_logw[8];
  if (LIK == LIK_MIX) {
    const int kd = a.ncomp * a.d;
    for (int i = threadIdx.x; i < kd; i += BLOCK) lds_means[i] = a.lik[i];
    if (threadIdx.x < (unsigned)a.ncomp) lds_logw[threadIdx.x] = a.lik[kd + threadIdx.x];
    __syncthreads();
  }
  const int d = a.d;
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t chain = gid / LPC;
  const int q = (int)(gid % LPC);
  if (chain >= (size_t)a.n) return 0u;
  const int k0 = 4 * q;
  const bool live = k0 < d;  // d % 4 == 0: a lane owns 4 parameters or none (d = 12, 20, ...)
  const uint32_t g = a.g0 + (uint32_t)chain;
  const size_t off = chain * (size_t)d + k0;

  f32x2 xe = {0, 0}, xo = {0, 0}, me = {0, 0}, mo = {0, 0}, se = {0, 0}, so = {0, 0}, te = {0, 0}, to = {0, 0};
  if (live) {
    const float4 f = *reinterpret_cast<const float4 *>(a.x + off);
    xe = f32x2{f.x, f.z}; xo = f32x2{f.y, f.w};
    te = f32x2{a.T[(k0 + 0) * d + k0 + 0], a.T[(k0 + 2) * d + k0 + 2]};
    to = f32x2{a.T[(k0 + 1) * d + k0 + 1], a.T[(k0 + 3) * d + k0 + 3]};
    if (MAIN && a.init_moments) {  // src/mcpar.cc:99-104
      se = f32x2{FPEPS, FPEPS}; so = f32x2{FPEPS, FPEPS};
    } else if (MAIN) {
      const float4 m = *reinterpret_cast<const float4 *>(a.mu + off);
      const float4 p = *reinterpret_cast<const float4 *>(a.psum2 + off);
      me = f32x2{m.x, m.z}; mo = f32x2{m.y, m.w};
      se = f32x2{p.x, p.z}; so = f32x2{p.y, p.w};
    }
  }
  // FULL, factor in registers: column col of this lane's rows (0, 2) and (1, 3) -- the packed operands of the two
  // multiply-adds a column costs; zero beyond the matrix (and, the factor being lower triangular, beyond the diagonal)
  f32x2 tre[TREGS ? 4 * LPC : 1], tro[TREGS ? 4 * LPC : 1];
  if (TREGS) {
#pragma unroll
    for (int col = 0; col < 4 * LPC; ++col) {
      const bool in = live && col < d;
      tre[col] = in ? f32x2{a.T[(k0 + 0) * d + col], a.T[(k0 + 2) * d + col]} : f32x2{0.0f, 0.0f};
      tro[col] = in ? f32x2{a.T[(k0 + 1) * d + col], a.T[(k0 + 3) * d + col]} : f32x2{0.0f, 0.0f};
    }
#pragma unroll
    for (int col = 0; col < 4 * LPC; ++col) asm volatile("" ::"v"(tre[col]), "v"(tro[col]));  // (awaited here, once: see below)
  }
  f32x2 gme = {0, 0}, gmo = {0, 0};  // Gaussian: this lane's means and 1/sigma^2 (lik = mu[d], s2inv[d])
  float gs0 = 0, gs1 = 0, gs2 = 0, gs3 = 0;
  if (LIK == LIK_GAUSS && live) {
    gme = f32x2{a.lik[k0 + 0], a.lik[k0 + 2]}; gmo = f32x2{a.lik[k0 + 1], a.lik[k0 + 3]};
    gs0 = a.lik[d + k0 + 0]; gs1 = a.lik[d + k0 + 1]; gs2 = a.lik[d + k0 + 2]; gs3 = a.lik[d + k0 + 3];
  }
  float ly = a.ly[chain];
  uint32_t cnt = 0, wacc = 0;
  f32x2 al01 = {0, 0}, al23 = {0, 0};  // log of the four acceptance draws of this lane's current ACCEPT block
  uint32_t ablk = 0xffffffffu;
  float *sx = a.samp_x ? a.samp_x + off : nullptr;
  float *sl = a.samp_x ? a.samp_ly + chain : nullptr;
  const size_t sx_stride = (size_t)a.n * d, sl_stride = (size_t)a.n;
  float *sxv = (PREGEN && a.samp_x) ? (live ? a.samp_x + off : a.trash + 4 * gid) : nullptr;  // per-lane pointer
  const size_t sxv_stride = live ? sx_stride : 0;

  // one Metropolis step given this lane's four normals (ze = z0,z2; zo = z1,z3) and the log of the acceptance draw
  auto step = [&](int s, f32x2 ze, f32x2 zo, float u, float winv_s) {
    f32x2 pe, po;
    if (!FULL) {
      pe = fma2(te, ze, xe);  // src/mcpar.cc:302-312
      po = fma2(to, zo, xo);
    } else {
      f32x2 ae = xe, ao = xo;  // rows (0, 2) and (1, 3): every row accumulates its columns in ascending order
      const float zv[4] = {ze.x, zo.x, ze.y, zo.y};
      const int zslot = ((int)threadIdx.x >> 3) * 9;
      if (LPC == 8) {
        lds_z[zslot + q] = make_float4(zv[0], zv[1], zv[2], zv[3]);
        __builtin_amdgcn_wave_barrier();  // a chain never spans wavefronts; LDS is in order per wavefront
      }
#pragma unroll
      for (int qq = 0; qq < LPC; ++qq) {
        float4 zz;
        if (LPC == 8) zz = lds_z[zslot + qq];
        else if (LPC == 1) zz = make_float4(zv[0], zv[1], zv[2], zv[3]);
        else zz = make_float4(quad_bcast<LPC>(zv[0], qq), quad_bcast<LPC>(zv[1], qq), quad_bcast<LPC>(zv[2], qq), quad_bcast<LPC>(zv[3], qq));
        const float zc[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // column 4 qq + c of the lane's four rows: two packed multiply-adds
          if (TREGS) {
            ae = fma2(tre[qq * 4 + c], splat2(zc[c]), ae);
            ao = fma2(tro[qq * 4 + c], splat2(zc[c]), ao);
          } else {
            const float4 tr = lds_T[(qq * 4 + c) * LPC + q];  // (one read = the column's four entries)
            ae = fma2(f32x2{tr.x, tr.y}, splat2(zc[c]), ae);
            ao = fma2(f32x2{tr.z, tr.w}, splat2(zc[c]), ao);
          }
        }
        // (pinned once per column block: the multiply-adds are pure and get sunk behind ALL the step's reads of T
        // otherwise -- every entry read stays alive until then: mcx_fastb.hpp found 380 registers wanted that way)
        if (!TREGS) asm volatile("" : "+v"(ae), "+v"(ao));
      }
      if (LPC == 8) __builtin_amdgcn_wave_barrier();  // this step's reads precede the next step's write
      pe = ae;
      po = ao;
    }
    float acc = 0.0f;
    if (LIK == LIK_ROSEN1) {
      // src/rosenbrock.cc:4-21 on the pairs (x0,x1), (x2,x3)
      const f32x2 t1 = splat2(1.0f) - pe;
      const f32x2 t2 = fma2(-pe, pe, po);
      const f32x2 term = fma2(splat2(100.0f) * t2, t2, t1 * t1);
      if (live) acc = term.x + term.y;  // == (0 + term.x) + term.y: the terms are >= +0
    } else if (LIK == LIK_ROSEN2F) {
      // the overlapping Rosenbrock (MCX_VL_ROSENBROCK2_FIXED): terms k = k0 .. k0+3 on the pairs (x0,x1), (x1,x2),
      // (x2,x3), (x3, x4) with x4 = the first parameter of the chain's next lane (DPP row_shl 1); the chain's last
      // block has no fourth term
      const float nx = dpp_mov<0x101>(pe.x);
      const f32x2 b2 = f32x2{pe.y, nx};
      const f32x2 u1 = splat2(1.0f) - pe, v1 = splat2(1.0f) - po;
      const f32x2 u2 = fma2(-pe, pe, po), v2 = fma2(-po, po, b2);
      const f32x2 te_ = fma2(splat2(100.0f) * u2, u2, u1 * u1);  // terms k0, k0+2
      const f32x2 to_ = fma2(splat2(100.0f) * v2, v2, v1 * v1);  // terms k0+1, k0+3
      if (live) {
        acc = (te_.x + to_.x) + te_.y;
        if (k0 + 4 < d) acc = acc + to_.y;
      }
    } else if (LIK == LIK_GAUSS) {
      // src/rosenbrock.cc:44-61: acc = fma((0.5 a) a, 1/sigma^2, acc) for k = 0..3 in order
      const f32x2 ae = pe - gme, ao = po - gmo;
      const f32x2 he = (splat2(0.5f) * ae) * ae, ho = (splat2(0.5f) * ao) * ao;
      if (live) {
        acc = __builtin_fmaf(he.x, gs0, 0.0f);
        acc = __builtin_fmaf(ho.x, gs1, acc);
        acc = __builtin_fmaf(he.y, gs2, acc);
        acc = __builtin_fmaf(ho.y, gs3, acc);
      }
    }
    float lyt;
#ifdef MCX_USER_LIK
    if (LIK == LIK_USER) {  // the user's device functions on this lane's four trial parameters (x0, x1, x2, x3)
      Lik<LIK_USER, LPC> UL;
      UL.init(a.lik, d, k0, 0, 0);
      const float ub[4] = {pe.x, po.x, pe.y, po.y};
      lyt = UL.eval(ub, live ? 4 : 0);
    } else
#endif
    if (LIK == LIK_MIX) {
      // log sum_c w_c exp(-|x - m_c|^2 / 2) as a log-sum-exp (DualGaussian: src/rosenbrock.cc:63-78)
      const int K = a.ncomp;
      float e[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        e[c] = 0.0f;
        if (c < K) {
          float s2 = 0.0f;
          if (live) {
            const float4 m = *reinterpret_cast<const float4 *>(&lds_means[c * d + k0]);
            const f32x2 ae = pe - f32x2{m.x, m.z}, ao = po - f32x2{m.y, m.w};
            s2 = __builtin_fmaf(ae.x, ae.x, 0.0f);
            s2 = __builtin_fmaf(ao.x, ao.x, s2);
            s2 = __builtin_fmaf(ae.y, ae.y, s2);
            s2 = __builtin_fmaf(ao.y, ao.y, s2);
          }
          e[c] = __builtin_fmaf(-0.5f, group_sum<LPC>(s2), lds_logw[c]);
        }
      }
      float emax = e[0];
#pragma unroll
      for (int c = 1; c < 8; ++c)
        if (c < K) emax = e[c] > emax ? e[c] : emax;
      float ssum = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; c += 2) {  // exp two components at a time, add them in component order
        if (c < K) {
          const f32x2 ex = expf_v2x2(f32x2{e[c] - emax, e[c + 1] - emax});
          ssum = ssum + ex.x;
          if (c + 1 < K) ssum = ssum + ex.y;
        }
      }
      lyt = emax + logf_v1(ssum);
    } else {
      lyt = 0.0f - group_sum<LPC>(acc);
    }
    // src/mcpar.cc:62-75 (cfac = 1 for local proposals): log u < ly' - ly
    const bool take = accept_local(lyt, ly, u);
    xe = take ? pe : xe;
    xo = take ? po : xo;
    ly = take ? lyt : ly;
    cnt += take ? 1u : 0u;
    wacc += (uint32_t)__popcll(__ballot(take && q == 0));
    if (MAIN) {
      const f32x2 w2 = splat2(winv_s);               // src/mcpar.cc:186-187
      const f32x2 de = xe - me, dO = xo - mo;         // src/mcpar.cc:199-202
      me = fma2(de, w2, me);
      mo = fma2(dO, w2, mo);
      se = fma2(de, xe - me, se);
      so = fma2(dO, xo - mo, so);
      if (s == a.snap_after && live) {  // snapshot for the next exchange (src/mcpar.cc:202-208)
        const f32x2 ve = se * w2, vo = so * w2;
        float4 *slot = reinterpret_cast<float4 *>(a.musig_own + 2 * off);
        slot[0] = make_float4(me.x, ve.x, mo.x, vo.x);
        slot[1] = make_float4(me.y, ve.y, mo.y, vo.y);
        // the run's last step: the variances as mcx_get_var returns them (k_variance otherwise)
        if (a.sig_out) *reinterpret_cast<float4 *>(a.sig_out + off) = make_float4(ve.x, vo.x, ve.y, vo.y);
      }
      if (sx) {  // src/mcpar.cc:177-182
        if (PREGEN && a.samp_stride <= 1) {
          // latency-bound mode: no exec-mask regions -- idle lanes store to their trash slot, every lane
          // of the chain stores the (same) log-likelihood
          *reinterpret_cast<float4 *>(sxv) = make_float4(xe.x, xo.x, xe.y, xo.y);
          *sl = ly;
          sxv += sxv_stride;
          sl += sl_stride;
        } else if (a.samp_stride <= 1) {
          if (live) *reinterpret_cast<float4 *>(sx) = make_float4(xe.x, xo.x, xe.y, xo.y);
          if (q == 0) *sl = ly;
          sx += sx_stride;
          sl += sl_stride;
        } else if ((a.isamp0 + s) % a.samp_stride == 0) {  // thinned store: row = isamp / stride
          const size_t row = (size_t)((a.isamp0 + s) / a.samp_stride);
          if (live) *reinterpret_cast<float4 *>(sx + row * sx_stride) = make_float4(xe.x, xo.x, xe.y, xo.y);
          if (q == 0) sl[row * sl_stride] = ly;
        }
      }
    }
  };

  // Every load of the chain state is awaited here, once: the compiler's s_waitcnt bookkeeping is path-insensitive, so
  // a wait left to the first use inside the step loop stays there for every later step -- where the only vector-memory
  // operations in flight are the previous step's sample STORES, and "vmcnt(0)" means "until they have reached HBM".
  asm volatile("" ::"v"(xe), "v"(xo), "v"(te), "v"(to), "v"(me), "v"(mo), "v"(se), "v"(so), "v"(gme), "v"(gmo), "v"(gs0),
               "v"(gs1), "v"(gs2), "v"(gs3), "v"(ly));
  // 1/pwgt of the launch's steps: read through the CONSTANT address space, i.e. by scalar loads.  As a plain global
  // load (what the compiler made of the uniform address) it sat in the vector-memory queue behind the previous
  // step's sample stores, and the wait for it was a wait for those stores to reach HBM: 21 % of the wavefronts'
  // cycles parked in s_waitcnt (SQ_WAIT_ANY) on the headline job.
  const __attribute__((address_space(4))) float *wtab = (const __attribute__((address_space(4))) float *)a.winv;
  if (!PREGEN) {
    float wnext = MAIN ? wtab[a.isamp0] : 1.0f;  // requested one step ahead (the table is padded past nsamp)
    // accept threshold: Philox block (t >> 2) of the ACCEPT stream serves steps 4b..4b+3.  The LPC
    // lanes of a chain split the work: lane q draws block b for b % LPC == q, once per 4*LPC steps.
    auto refresh = [&](uint32_t blk) {
      if ((blk & ~(uint32_t)(LPC - 1)) != ablk) {  // the four logs are taken here, once per 4*LPC steps per lane
        ablk = blk & ~(uint32_t)(LPC - 1);
        const u32x4 aw = philox4x32_10(ablk + (uint32_t)q, g, 0u, 0u, a.seed, ST_ACCEPT);
        al01 = accept_lu_x2(aw.x, aw.y);
        al23 = accept_lu_x2(aw.z, aw.w);
      }
    };
    auto one_step = [&](int s) {
      const uint32_t t = a.t0 + (uint32_t)s;
      const float wthis = wnext;
      if (MAIN) wnext = wtab[a.isamp0 + s + 1];
      f32x2 ze, zo;
      normal4_packed(philox4x32_10(t, g, (uint32_t)q, 0u, a.seed, ST_LOCAL), ze, zo);
      const uint32_t blk = t >> 2;
      refresh(blk);
      const uint32_t wi = t & 3u;
      const float mine = wi == 0u ? al01.x : (wi == 1u ? al01.y : (wi == 2u ? al23.x : al23.y));
      const float lu = as_f32(group_bcast<LPC>(as_u32(mine), blk & (uint32_t)(LPC - 1), q));
      step(s, ze, zo, lu, wthis);
    };
    int s = 0;
#if MCX_FAST_UNROLL4
    // Four steps at a time from a step index that is a multiple of 4 on: which of the block's four logs a step takes is then
    // known at compile time, and the lane that holds them is looked up once per block instead of once per step (two
    // wave-uniform switches per step otherwise: some ten scalar branches and two moves)
    if (!FULL) {
      for (; s < a.nsteps && ((a.t0 + (uint32_t)s) & 3u); ++s) one_step(s);
      for (; s + 4 <= a.nsteps; s += 4) {
        const uint32_t t = a.t0 + (uint32_t)s, blk = t >> 2;
        refresh(blk);
        const uint32_t holder = blk & (uint32_t)(LPC - 1);
        const float lu4[4] = {as_f32(group_bcast<LPC>(as_u32(al01.x), holder, q)), as_f32(group_bcast<LPC>(as_u32(al01.y), holder, q)),
                              as_f32(group_bcast<LPC>(as_u32(al23.x), holder, q)), as_f32(group_bcast<LPC>(as_u32(al23.y), holder, q))};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float wthis = wnext;
          if (MAIN) wnext = wtab[a.isamp0 + s + u + 1];
          f32x2 ze, zo;
          normal4_packed(philox4x32_10(t + (uint32_t)u, g, (uint32_t)q, 0u, a.seed, ST_LOCAL), ze, zo);
          step(s + u, ze, zo, lu4[u], wthis);
        }
      }
    }
#endif
    for (; s < a.nsteps; ++s) one_step(s);
  } else {
    // Batches of P steps, double buffered: at the top of a batch every load of it (issued one whole
    // batch earlier) is awaited at once and moved to `cur`, then the loads of the next batch are issued,
    // then the P steps run without touching memory counters (a per-step wait would expose the full
    // load latency every step: the counters retire in order).  Z, U and the 1/pwgt table are padded by
    // 2P steps, so the prefetch never needs a bounds check; pointers advance by addition only.
    constexpr int P = 8;
    const float *zq = a.zpre + (live ? off : 0), *uq = a.upre + chain;  // idle lanes read a valid address
    const float *wq = MAIN ? a.winv + a.isamp0 : a.upre;                 // (burn-in: any valid address)
    float4 nxt[P], cur[P];
    float nxu[P], cuu[P], nxw[P], cuw[P];
    auto fetch = [&]() {
      const float *zk = zq, *uk = uq;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        nxt[k] = *reinterpret_cast<const float4 *>(zk);
        nxu[k] = *uk;
        nxw[k] = wq[k];
        zk += sx_stride;
        uk += sl_stride;
      }
      zq = zk;
      uq = uk;
      wq += P;
    };
    fetch();
    int s0 = 0;
    for (; s0 + P <= a.nsteps; s0 += P) {  // full batches: no per-step bounds test
#pragma unroll
      for (int k = 0; k < P; ++k) {
        cur[k] = nxt[k];
        cuu[k] = nxu[k];
        cuw[k] = nxw[k];
      }
      fetch();
#pragma unroll
      for (int k = 0; k < P; ++k) step(s0 + k, f32x2{cur[k].x, cur[k].z}, f32x2{cur[k].y, cur[k].w}, cuu[k], cuw[k]);
    }
#pragma unroll
    for (int k = 0; k < P; ++k)  // the last, partial batch
      if (s0 + k < a.nsteps) step(s0 + k, f32x2{nxt[k].x, nxt[k].z}, f32x2{nxt[k].y, nxt[k].w}, nxu[k], nxw[k]);
  }

  if (live) *reinterpret_cast<float4 *>(a.x + off) = make_float4(xe.x, xo.x, xe.y, xo.y);
  if (q == 0) {
    a.ly[chain] = ly;
    a.acc_cnt[chain] += cnt;
  }
  if (MAIN && live) {
    *reinterpret_cast<float4 *>(a.mu + off) = make_float4(me.x, mo.x, me.y, mo.y);
    *reinterpret_cast<float4 *>(a.psum2 + off) = make_float4(se.x, so.x, se.y, so.y);
  }
  // one slot per wavefront, owned by it: no atomics (4096 same-address atomics cost ~40 us per launch); with the tuner
  // inside the launch the count travels through tuner_epilogue instead
  if ((threadIdx.x & 63u) == 0 && wacc && !a.tun.on) a.acc_slots[gid >> 6] += wacc;
  return wacc;
}

// ---------------------------------------------------------------------------------------------
// Unfused per-step kernels (host-callback likelihoods, Rosenbrock2-as-written, remote steps,
// MCX_OPT_FUSE=0).  Same device functions, hence the same bits as k_fused_steps.
// ---------------------------------------------------------------------------------------------
struct StepArgs {
  float *x, *ly, *mu, *psum2;
  float *ptrial, *lytrial, *cfac;
  const float *mutrial, *sigtrial;  // remote adoption (src/mcpar.cc:189-196)
  uint32_t *acc_cnt;
  uint32_t *acc_slots;
  const float *T;
  float *samp_x, *samp_ly;
  uint8_t *mask;
  const float *lik;
  int ncomp;
  int n, d;
  uint32_t g0, t, seed;
  int isamp;
  int diag, vec4, remote;
};

template <int LPC>
__global__ __launch_bounds__(BLOCK) void k_propose_local(const StepArgs a)
{
  // rows padded by 4 floats: the 4-row blocks of different lanes then start 16 banks apart instead of
  // on the same bank (8-way -> 2-way conflict at d = 32) and stay 16-byte aligned
  __shared__ __attribute__((aligned(16))) float Tlds[MAXD_LDS * (MAXD_LDS + 4)];
  __shared__ __attribute__((aligned(16))) float zlds[BLOCK * 4];
  const bool diag = a.diag != 0, vec4 = a.vec4 != 0;
  const int d = a.d;
  const float *Tl = a.T;
  float *zbuf = nullptr;
  int ldt = d;
  if (!diag && d <= MAXD_LDS) {
    ldt = d + 4;
    for (int i = threadIdx.x; i < d * d; i += BLOCK) Tlds[(i / d) * ldt + (i % d)] = a.T[i];
    __syncthreads();
    Tl = Tlds;
    if (vec4) zbuf = zlds;
  }
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t chain = gid / LPC;
  const int q = (int)(gid % LPC);
  if (chain >= (size_t)a.n) return;
  const int k0 = 4 * q;
  const int nv = d - k0 >= 4 ? 4 : (d - k0 > 0 ? d - k0 : 0);
  float x[4], pt[4], tdiag[4];
  load_block(a.x, chain, d, k0, nv, vec4, x);
#pragma unroll
  for (int k = 0; k < 4; ++k) tdiag[k] = (diag && k < nv) ? a.T[(k0 + k) * d + k0 + k] : 0.0f;
  propose_block<LPC>(x, pt, tdiag, Tl, diag, d, q, nv, a.t, a.g0 + (uint32_t)chain, a.seed, zbuf, ldt);
  store_block(a.ptrial, chain, d, k0, nv, vec4, pt);
  if (q == 0) a.cfac[chain] = 1.0f;  // src/mcpar.cc:309
}

// batched likelihood, the VLFunc call: x[n][d] -> y[n]
template <int LPC, int LIK>
__device__ __forceinline__ void eval_body(const float *__restrict__ x, float *__restrict__ y, int n, int d, const float *lik,
                                          int ncomp, int vec4);

template <int LPC, int LIK>
__global__ __launch_bounds__(BLOCK) void k_eval(const float *__restrict__ x, float *__restrict__ y,
                                                int n, int d, const float *lik, int ncomp, int vec4)
{
  eval_body<LPC, LIK>(x, y, n, d, lik, ncomp, vec4);
}

template <int LPC, int LIK>
__device__ __forceinline__ void eval_body(const float *__restrict__ x, float *__restrict__ y, int n, int d, const float *lik,
                                          int ncomp, int vec4)
{
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t chain = gid / LPC;
  const int q = (int)(gid % LPC);
  if (chain >= (size_t)n) return;
  const int k0 = 4 * q;
  const int nv = d - k0 >= 4 ? 4 : (d - k0 > 0 ? d - k0 : 0);
  float xb[4];
  load_block(x, chain, d, k0, nv, vec4 != 0, xb);
  Lik<LIK, LPC> L;
  L.init(lik, d, k0, nv, ncomp);
  const float v = L.eval(xb, nv);
  if (q == 0) y[chain] = v;
}

// Rosenbrock2 exactly as written (src/rosenbrock.cc:25-41): flat index, x[i+1] read across the
// set boundary, '-' on the second term, last set one term short.
static __global__ __launch_bounds__(BLOCK) void k_eval_rosen2(const float *__restrict__ x,
                                                       float *__restrict__ y, int n, int d)
{
  const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j >= (size_t)n) return;
  const size_t ntot = (size_t)n * d;
  float acc = 0.0f;
  for (size_t i = j * d; i < (j + 1) * d; ++i)
    if (i + 1 < ntot) {
      const float t1 = 1.0f - x[i];
      const float t2 = __builtin_fmaf(-x[i], x[i], x[i + 1]);
      acc = acc + __builtin_fmaf(-(100.0f * t2), t2, t1 * t1);
    }
  y[j] = 0.0f - acc;
}

template <int LPC, bool MAIN>
__global__ __launch_bounds__(BLOCK) void k_accept(const StepArgs a)
{
  const bool vec4 = a.vec4 != 0;
  const int d = a.d;
  const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t chain = gid / LPC;
  const int q = (int)(gid % LPC);
  if (chain >= (size_t)a.n) return;
  const int k0 = 4 * q;
  const int nv = d - k0 >= 4 ? 4 : (d - k0 > 0 ? d - k0 : 0);
  const uint32_t g = a.g0 + (uint32_t)chain;
  float x[4];
  load_block(a.x, chain, d, k0, nv, vec4, x);
  float ly = a.ly[chain];
  const float lyt = a.lytrial[chain];
  const u32x4 aw = philox4x32_10(a.t >> 2, g, 0u, 0u, a.seed, ST_ACCEPT);
  const uint32_t aword = pick_word(aw, a.t & 3u);
  const bool take = a.remote ? accept_decision(lyt, ly, a.cfac[chain], aword) : accept_local(lyt, ly, accept_lu(aword));
  if (take) {
    load_block(a.ptrial, chain, d, k0, nv, vec4, x);
    ly = lyt;
    store_block(a.x, chain, d, k0, nv, vec4, x);
    if (q == 0) {
      a.ly[chain] = ly;
      a.acc_cnt[chain] += 1;
    }
  }
  const uint32_t wacc = (uint32_t)__popcll(__ballot(take && q == 0));
  if (a.mask && q == 0) a.mask[chain] = take ? 1 : 0;
  if (MAIN) {
    float mu[4], ps[4];
    const float pwgt = (float)(a.isamp + 1);
    const float winv = 1.0f / pwgt;
    if (a.remote && take) {  // src/mcpar.cc:189-196
      float sg[4];
      load_block(a.mutrial, chain, d, k0, nv, vec4, mu);
      load_block(a.sigtrial, chain, d, k0, nv, vec4, sg);
#pragma unroll
      for (int k = 0; k < 4; ++k) ps[k] = sg[k] * (pwgt - 1.0f);
    } else {
      load_block(a.mu, chain, d, k0, nv, vec4, mu);
      load_block(a.psum2, chain, d, k0, nv, vec4, ps);
    }
    welford_block(x, mu, ps, winv);
    store_block(a.mu, chain, d, k0, nv, vec4, mu);
    store_block(a.psum2, chain, d, k0, nv, vec4, ps);
    if (a.samp_x) {
      store_block(a.samp_x, chain, d, k0, nv, vec4, x);
      if (q == 0) a.samp_ly[chain] = ly;
    }
  }
  // one slot per wavefront, owned by it: no atomics (4096 same-address atomics cost ~40 us per launch)
  if ((threadIdx.x & 63u) == 0 && wacc) a.acc_slots[gid >> 6] += wacc;
}

// start of a run on the multi-launch paths, one launch instead of three fills and a copy: per-wavefront and
// per-chain accept counters and the tuner-trace length to zero, the staged initial state into place (src/mcpar.cc:47-50),
// the proposal factor back to the one installed
static __global__ void k_run_reset(uint32_t *slots, size_t nslots, uint32_t *acc_cnt, size_t n, int *ntrace,
                                   float *__restrict__ x, const float *__restrict__ x0, size_t ntot,
                                   float *__restrict__ cov, const float *__restrict__ cov0, size_t ncov)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nslots) slots[i] = 0u;
  if (i < n) acc_cnt[i] = 0u;
  if (i == 0) *ntrace = 0;
  if (x0 && i < ntot) x[i] = x0[i];
  if (cov0)  // the factor the tuner is about to rescale, from the factor as installed
    for (size_t k = i; k < ncov; k += (size_t)gridDim.x * blockDim.x) cov[k] = cov0[k];
}

// start of the main loop: mu = 0, psum2 = FPEPS (src/mcpar.cc:99-104)
static __global__ void k_init_moments(float *mu, float *psum2, size_t ntot)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntot) { mu[i] = 0.0f; psum2[i] = FPEPS; }
}

// Publish this shard's (mu, sig^2) pairs into its musigall slot (src/mcpar.cc:202-208).  The
// reference rewrites the slot every step; the slot is only read by genRemote and by the exchange,
// so it is written once, right before either of them, from the resident moments.
static __global__ void k_publish(const float *__restrict__ mu, const float *__restrict__ psum2,
                          float *__restrict__ slot, size_t ntot, float winv)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntot) reinterpret_cast<float2 *>(slot)[i] = make_float2(mu[i], psum2[i] * winv);
}

// sig = psum2 / pwgt for the getter (src/mcpar.cc:202)
static __global__ void k_variance(const float *psum2, float *sig, size_t ntot, float winv)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntot) sig[i] = psum2[i] * winv;
}

// ---------------------------------------------------------------------------------------------
// Burn-in acceptance-rate tuner (src/mcpar.cc:77-96), on device so that burn-in needs no host
// round trip.  slots = per-wavefront accepts of the segment just run, ctr[1] = tuner naccept, ctr[2] = tuner
// ntrial, ctr[3] = total burn-in accepts.  Integer counters (the reference's float counters stop
// counting at 2^24: SURVEY §7).
// ---------------------------------------------------------------------------------------------
// accepted proposals of the main loop so far: *dst += sum(slots)
static __global__ void k_reduce_slots(uint32_t *slots, int nslots, unsigned long long *dst)
{
  const unsigned long long t = block_sum_slots(slots, nslots);
  if (threadIdx.x == 0) *dst += t;
}

static __global__ void k_tuner(unsigned long long *ctr, float *T, int ncov, unsigned long long add_trials,
                        int check, float armin, float armax, float dfac, float ifac, float *trace,
                        int *ntrace, uint32_t *slots, int nslots)
{
  tuner_block(block_sum_slots(slots, nslots), ctr, T, ncov, add_trials, check, armin, armax, dfac, ifac, trace, ntrace);
}

// MCout row format (src/mcout.cc:129-137): (np parameters, log-likelihood) per (step, chain)
static __global__ void k_rows_interleave(const float *__restrict__ sx, const float *__restrict__ sl,
                                  float *__restrict__ rows, size_t nrows, int d)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ncol = (size_t)d + 1;
  if (i >= nrows * ncol) return;
  const size_t r = i / ncol;
  const int c = (int)(i - r * ncol);
  rows[i] = c < d ? sx[r * d + c] : sl[r];
}

// Running maximum-likelihood sample (MCout::add's maxlval, src/mcout.cc:140-144: the FIRST strict maximum in
// (step, chain) order).  key = order-preserving bits of the value << 32 | ~row: the largest key is the largest
// value at the lowest row; 0 = nothing above -inf seen (NaN never compares greater and stays out).
static __global__ __launch_bounds__(BLOCK) void k_argmax_first(const float *__restrict__ ly, size_t nrows,
                                                               unsigned long long *__restrict__ key_out)
{
  unsigned long long best = 0;
  for (size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x; r < nrows; r += (size_t)gridDim.x * BLOCK) {
    const float v = ly[r];
    if (v > -__builtin_inff()) {
      uint32_t b = as_u32(v);
      b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
      const unsigned long long k = ((unsigned long long)b << 32) | (unsigned long long)(0xffffffffu - (uint32_t)r);
      best = k > best ? k : best;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o);
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63u) == 0 && best) atomicMax(key_out, best);
}

// best = {log-likelihood, parameters}: replaced when the block's first maximum is strictly greater (an equal
// value in a later block is not the FIRST maximum); clears the key for the next block
static __global__ void k_best_update(unsigned long long *__restrict__ key, const float *__restrict__ ly,
                                     const float *__restrict__ x, int d, float *__restrict__ best)
{
  const unsigned long long k = *key;
  if (k) {
    const size_t r = (size_t)(0xffffffffu - (uint32_t)k);
    const float v = ly[r];
    if (v > best[0])
      for (int i = threadIdx.x; i < d; i += blockDim.x) best[1 + i] = x[r * (size_t)d + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      if (v > best[0]) best[0] = v;
      *key = 0;
    }
  }
}

static __global__ void k_best_reset(float *best, int d, unsigned long long *key)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { best[0] = -__builtin_inff(); *key = 0; }
  if (i >= 1 && i <= d) best[i] = 0.0f;
}

// test hooks -----------------------------------------------------------------------------------
static __global__ void k_debug_numerics(int what, int n, const uint32_t *in, uint32_t *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t w = in[i];
  float s, c;
  uint32_t r = 0;
  switch (what) {
  case 0: r = as_u32(logf_v1(as_f32(w))); break;
  case 1: r = as_u32(expf_v2(as_f32(w))); break;
  case 2: sincos2pi_v1(w, s, c); r = as_u32(s); break;
  case 3: sincos2pi_v1(w, s, c); r = as_u32(c); break;
  case 4: r = as_u32(u24(w)); break;
  case 5: r = as_u32(uopen(w)); break;
  case 6: r = philox4x32_10(w, 0, 0, 0, 0, 0).x; break;
  case 7: r = as_u32(sqrt_rn_pos(as_f32(w))); break;
  case 8: r = as_u32(__builtin_sqrtf(as_f32(w))); break;
  case 9: r = as_u32(expf_v2x2(f32x2{as_f32(w), 1.0f}).x); break;
  case 10: r = as_u32(expf_v2x2(f32x2{-3.0f, as_f32(w)}).y); break;
  case 11: r = as_u32(logf_v1x2(f32x2{as_f32(w), 0.5f}).x); break;
  case 12: { f32x2 s2, c2; sincos2pi_v1x2(w, w ^ 0x9e3779b9u, s2, c2); r = as_u32(s2.x) ^ (as_u32(c2.x) << 1); } break;
  case 13: { float s1, c1; sincos2pi_v1(w, s1, c1); r = as_u32(s1) ^ (as_u32(c1) << 1); } break;
  case 14: r = as_u32(accept_lu(w)); break;
  case 15: r = as_u32(accept_lu_x2(w ^ 0x5bd1e995u, w).y); break;
  default: break;
  }
  out[i] = r;
}

// counts bit patterns in [lo, hi) where sqrt_rn_pos differs from the compiler's IEEE sqrtf
static __global__ void k_debug_sqrt_sweep(uint32_t lo, uint32_t hi, unsigned long long *nbad, uint32_t *first_bad)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (uint64_t w = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < hi; w += stride) {
    const float x = as_f32((uint32_t)w);
    if (as_u32(sqrt_rn_pos(x)) != as_u32(__builtin_sqrtf(x))) {
      ++bad;
      atomicMin(first_bad, (uint32_t)w);
    }
  }
  if (bad) atomicAdd(nbad, bad);
}

static __global__ void k_debug_normals(uint32_t seed, uint32_t stream, uint32_t t, uint32_t g0, uint32_t a,
                                uint32_t q, int n, float *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float z[4];
  normal4_from_words(philox4x32_10(t, g0 + (uint32_t)i, a, q, seed, stream), z);
  for (int k = 0; k < 4; ++k) out[4 * i + k] = z[k];
}

}  // namespace mcx
