// mcx_screen.hpp -- the third exact screen of the Murray sweeps (mcx_remote.hpp has the first: boxes of four coordinates;
// mcx_cull_proj.hpp the second: one direction), and the only one that looks at every PAIR: a lower bound of
//     arg(j, i) = sum_k w_ik (mu_ik - x_jk)^2        (src/mcpar.cc:355-372, :404-426 through mcx_remote.hpp's sweeps)
// for each (chain j, Gaussian Q_i) on the matrix cores, reduced over the 128 chains a wavefront of the sweep holds to the
// one bit per (group, Q_i) the sweep's exclusion masks carry.  The sweep itself -- the arithmetic whose bits must match
// the reference's -- is untouched; the screen only decides which rows it may skip, and must never skip a row that could
// matter.  Measured before it was written (tools/murray_gemm_screen_probe.py, EXPERIMENTS.md): the rows it can rule out
// are 84-95 % of C3-murray's sum sweeps (boxes: 63-70 %), 90-99 % of every min-arg sweep of C3-murray and of the 32-D
// mixture C5 (boxes: none on C5), 74-96 % of C5's early sum sweeps and next to none of its late ones.
//
// The bound.  With y_j = x_j - c and m_i = mu_i - c (any centre c):
//     arg(j, i) = sum_k w_ik m_ik^2  +  sum_k (-2 w_ik m_ik) y_jk  +  sum_k w_ik y_jk^2  =  c_i + A_j . B_i,
//     A_j = (y_j, y_j^2),  B_i = (-2 w_i m_i, w_i)                                   (K = 2 np products per pair).
// A and B are rounded to bf16 (8 significant bits, unit roundoff u = 2^-8): every product moves by at most
// (2u + u^2) |a b|, their sum by at most (2^-7 + 2^-16) sum |a_k b_k| <= (2^-7 + 2^-16) |A_j| |B_i| (Cauchy-Schwarz), so
//     arg(j, i)  >=  c_i + bf16(A_j) . bf16(B_i) - S |A_j| |B_i|,        S = 1.002 * 2^-7 + 2^-12
// where the 2^-12 pays for the fp32 accumulation inside the matrix core.  If every one of its at most K + 5 additions
// rounds with relative error 2^-23 on partial sums no larger than the sum of the terms' magnitudes, that is < 2^-16 of
// that sum; how v_mfma_f32_32x32x16_bf16 aligns and truncates the sixteen terms of one instruction is not documented to
// follow that model, so the allowance is sixteen times the model's (3 % of S: the screen keeps ~0.1 % more
// rows) and tests/test_gpu_murray_screen.py holds adversarial cases: a centre far from the chains (|A| |B| >> arg), args
// within 1e-6 .. 1e-2 of the bounds on either side, weights and coordinates of very different magnitudes.
// (y_j = x_j - c is formed in double precision from floats: exact unless the exponents differ by more than 29 bits,
// and then off by < 2^-53 of the larger one -- far inside the same allowance.)  The comparison
// with the chain's bound L_j (176 for a sum sweep: beyond it Q_i is exactly 0; min(176, the chain's arg against its own
// Gaussian) for the min-arg sweep: the minimum starts at or below it) rides in sixteen more products of the same GEMM:
//     A'_j = (A_j | 1, 1, 1, -up(S |A_j|), -up(L_j (1 + 1e-4) + 1e-3), 0 ...)
//     B'_i = (B_i | c_hi, c_mid, c_lo, up(|B_i|), 1, 0 ...)
// c_hi + c_mid + c_lo = three bf16 pieces of a float <= c_i (1 - 2^-14) (exact: 24 bits), up() = the next bf16 at or
// above, so that  A'_j . B'_i > 0  =>  arg(j, i) > L_j (1 + 1e-4)  in real numbers; the sweep's float arg is at least
// (1 - 3e-6) of the real one (np + 2 roundings, all terms >= 0), hence > L_j too: the row cannot matter to chain j.
// The epilogue is a running minimum of the accumulators over the group's 128 chains: a row is skipped iff that minimum
// is > 0.  Chains or Gaussians with anything that is not an ordinary number (or with norms beyond 1e15) are replaced by
// vectors that make every product with them negative: they exclude nothing.  What remains is the bit-exact tests against
// the oracle, which knows nothing of any of this.
//
// Matrix-core shape: v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD): chains on the rows, Gaussians on the columns -- a
// lane's 16 accumulators then all belong to ONE Gaussian, and the minimum over the chains needs no lane traffic but one
// exchange between the two half-wavefronts (done by the ballot).  A wavefront keeps its group's A' in registers
// (4 row tiles x K/16 fragments), the four wavefronts of a workgroup (four groups) share each block of 128 Gaussians
// through LDS: B' passes through L2 once per 512 chains.
#pragma once
#include "mcx_remote.hpp"

constexpr int SCR_EXTRA = 16;      // the products that carry the comparison
constexpr int SCR_BLK = 128;       // Gaussians per LDS block
constexpr int SCR_WAVES = 4;       // groups (wavefronts) per workgroup
constexpr double SCR_S = 1.002 * 0.0078125 + 0.000244140625;
constexpr float SCR_HUGE = 2.9e38f;  // (finite in bf16)
constexpr double SCR_NORM_MAX = 1e15, SCR_CONST_MAX = 1e30;
__host__ __device__ constexpr int scr_k(int dmax) { return 2 * dmax + SCR_EXTRA; }

typedef __bf16 scr_bf16x8 __attribute__((ext_vector_type(8)));
typedef float scr_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned scr_u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a struct: an array of them stays in scratch)

__device__ __forceinline__ unsigned short scr_rn(float f)  // nearest even (ordinary numbers only)
{
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned short scr_up(float f)  // the smallest bf16 >= f
{
  const unsigned u = __float_as_uint(f);
  unsigned short h = (unsigned short)(u >> 16);
  if ((u & 0xffffu) && !(u >> 31)) h = (unsigned short)(h + 1);  // (a negative number truncates upwards)
  return h;
}
__device__ __forceinline__ unsigned short scr_trunc(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__device__ __forceinline__ float scr_val(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// the centre: the mean of (a sample of 2048 of) the Gaussians' means -- any centre is exact, a central one keeps |A| |B|
// small.  One workgroup; d a power of two <= 64 (thread t: dimension t % d of the rows t / d, t / d + 1024 / d, ...).
static __global__ __launch_bounds__(1024) void k_screen_centre(const float *__restrict__ qpar, int N, int d, float *__restrict__ centre)
{
  __shared__ float sum[1024], cnt[1024];  // [slice][dimension]: added up in a fixed order (the groups the masks are made
                                          // for still depend on the counting sort's atomics: the count of pairs left
                                          // varies by a few in ten thousand from run to run, the results by nothing)
  const int stride = N > 2048 ? N / 2048 : 1;
  const int rows = (N + stride - 1) / stride;
  const int k = (int)threadIdx.x % d, per = 1024 / d;
  float s = 0.0f, c = 0.0f;
  for (int r = (int)threadIdx.x / d; r < rows; r += per) {
    const float v = qpar[2 * ((size_t)r * stride * d + k)];
    if (v - v == 0.0f) { s += v; c += 1.0f; }
  }
  sum[threadIdx.x] = s;
  cnt[threadIdx.x] = c;
  __syncthreads();
  if ((int)threadIdx.x < d) {
    float ts = 0.0f, tc = 0.0f;
    for (int u = 0; u < per; ++u) { ts += sum[u * d + k]; tc += cnt[u * d + k]; }
    const float mean = tc > 0.0f ? ts / tc : 0.0f;
    centre[threadIdx.x] = (mean - mean == 0.0f) ? mean : 0.0f;
  }
}

// B'_i, one Gaussian per thread; rows N .. nrows-1 (the padding of the last block) are zero -- their bits are masked.
template <int DMAX>
__global__ __launch_bounds__(BLOCK) void k_screen_prep_q(const float *__restrict__ qpar, int N, int nrows,
                                                         const float *__restrict__ centre, unsigned short *__restrict__ B)
{
  constexpr int K = scr_k(DMAX);
  const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
  if (i >= nrows) return;
  unsigned short row[K];
#pragma unroll
  for (int k = 0; k < K; ++k) row[k] = 0;
  if (i < N) {
    const float2 *src = reinterpret_cast<const float2 *>(qpar + 2 * (size_t)i * DMAX);
    double cs = 0.0, nb2 = 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      const float2 v = src[k];
      const double m = (double)v.x - (double)centre[k], w = (double)v.y;
      const double t = -2.0 * w * m;
      row[k] = scr_rn((float)t);
      row[DMAX + k] = scr_rn(v.y);
      cs += w * m * m;
      nb2 += t * t + w * w;
      ok = ok && (v.y >= 0.0f);
    }
    const double nb = sqrt(nb2);
    ok = ok && (nb < SCR_NORM_MAX) && (cs < SCR_CONST_MAX);  // (false for anything that is not an ordinary number)
    if (ok) {
      const float cf = (float)(cs * (1.0 - 0.00006103515625));
      const unsigned short hi = scr_trunc(cf);
      const float r1 = cf - scr_val(hi);
      const unsigned short mid = scr_trunc(r1);
      const float r2 = r1 - scr_val(mid);
      row[2 * DMAX + 0] = hi;
      row[2 * DMAX + 1] = mid;
      row[2 * DMAX + 2] = scr_trunc(r2);
      row[2 * DMAX + 3] = scr_up((float)(nb * (1.0 + 1e-6)));
    } else {
#pragma unroll
      for (int k = 0; k < 2 * DMAX; ++k) row[k] = 0;
      row[2 * DMAX + 0] = scr_trunc(-SCR_HUGE);  // every chain's product with this row is negative: never excluded
    }
    row[2 * DMAX + 4] = 0x3f80;  // 1: meets the chain's -L
  }
  uint4 *dst = reinterpret_cast<uint4 *>(B + (size_t)i * K);
#pragma unroll
  for (int c = 0; c < K / 8; ++c) {
    uint4 v;
    v.x = row[8 * c + 0] | ((unsigned)row[8 * c + 1] << 16);
    v.y = row[8 * c + 2] | ((unsigned)row[8 * c + 3] << 16);
    v.z = row[8 * c + 4] | ((unsigned)row[8 * c + 5] << 16);
    v.w = row[8 * c + 6] | ((unsigned)row[8 * c + 7] << 16);
    dst[c] = v;
  }
}

// A'_j, one position of the (sorted) active list per thread; positions nact .. 128 ngroups - 1 get a row that is past
// every bound (they are no chains).  Runs behind the sort's k_cull_scatter and, like k_cull_boxes, leaves the sort's sums
// and histogram zero for the next one.
template <int DMAX, bool SUMS>
__global__ __launch_bounds__(BLOCK) void k_screen_prep_x(const float *__restrict__ xrows, const int *__restrict__ order, int nact,
                                                         int nrows, const float *__restrict__ qpar, int own0,
                                                         const float *__restrict__ centre, unsigned short *__restrict__ A,
                                                         float *__restrict__ stats_done, unsigned *__restrict__ hist_done)
{
  constexpr int K = scr_k(DMAX);
  for (int i = (int)(blockIdx.x * BLOCK + threadIdx.x); i < CULL_BINS; i += (int)(gridDim.x * BLOCK)) hist_done[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < 2 * CULL_KD) stats_done[threadIdx.x] = 0.0f;
  const int p = (int)(blockIdx.x * BLOCK + threadIdx.x);
  if (p >= nrows) return;
  unsigned short row[K];
#pragma unroll
  for (int k = 0; k < K; ++k) row[k] = 0;
  if (p < nact) {
    const int j = order ? order[p] : p;
    const float *x = xrows + (size_t)j * DMAX;
    double na2 = 0.0;
    float a0 = 0.0f;
    const float *qo = SUMS ? qpar : qpar + 2 * (size_t)(own0 + j) * DMAX;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      const float xk = x[k];
      const double y = (double)xk - (double)centre[k], y2 = y * y;
      row[k] = scr_rn((float)y);
      row[DMAX + k] = scr_rn((float)y2);
      na2 += y2 + y2 * y2;
      if (!SUMS) {  // the chain's arg against its own Gaussian, with the very operations of the sweep
        const float xm = qo[2 * k] - xk;
        a0 = __builtin_fmaf(xm * xm, qo[2 * k + 1], a0);
      }
    }
    const double na = sqrt(na2);
    const float lim = SUMS ? ZERO_ARG : (a0 < ZERO_ARG ? a0 : ZERO_ARG);  // (an own arg that is no number: 176)
    row[2 * DMAX + 0] = 0x3f80;
    row[2 * DMAX + 1] = 0x3f80;
    row[2 * DMAX + 2] = 0x3f80;
    if (na < SCR_NORM_MAX && lim >= 0.0f) {
      row[2 * DMAX + 3] = (unsigned short)(scr_up((float)(SCR_S * na * (1.0 + 1e-6))) | 0x8000u);
      row[2 * DMAX + 4] = (unsigned short)(scr_up(lim * (1.0f + 1e-4f) + 1e-3f) | 0x8000u);
    } else {
#pragma unroll
      for (int k = 0; k < 2 * DMAX; ++k) row[k] = 0;
      row[2 * DMAX + 4] = scr_trunc(-SCR_HUGE);  // this chain excludes nothing
    }
  } else {
    row[2 * DMAX + 4] = scr_trunc(SCR_HUGE);
  }
  uint4 *dst = reinterpret_cast<uint4 *>(A + (size_t)p * K);
#pragma unroll
  for (int c = 0; c < K / 8; ++c) {
    uint4 v;
    v.x = row[8 * c + 0] | ((unsigned)row[8 * c + 1] << 16);
    v.y = row[8 * c + 2] | ((unsigned)row[8 * c + 3] << 16);
    v.z = row[8 * c + 4] | ((unsigned)row[8 * c + 5] << 16);
    v.w = row[8 * c + 6] | ((unsigned)row[8 * c + 7] << 16);
    dst[c] = v;
  }
}

// excl[w][g] bit b = Q_{64 w + b} may matter to group g (the layout of k_cull_test).  Workgroup = SCR_WAVES groups x
// `bchunk` blocks of SCR_BLK Gaussians.  A has 128 ngroups rows, B a whole number of blocks.
template <int DMAX>
__global__ __launch_bounds__(SCR_WAVES * 64, 2) void k_screen_gemm(const unsigned short *__restrict__ A, const unsigned short *__restrict__ B,
                                                                int nact, int N, int ngroups, int bchunk,
                                                                unsigned long long *__restrict__ excl, int excl_words,
                                                                unsigned long long *__restrict__ nkept, int blk0, int blk1)
{
  constexpr int K = scr_k(DMAX), KS = K / 16, ROWB = K * 2, LROW = ROWB + 16;
  constexpr int NT = SCR_WAVES * 64, CH = SCR_BLK * ROWB / 16, PER = (CH + NT - 1) / NT;
  static_assert(CH % NT == 0, "a block of Gaussians is a whole number of 16-byte pieces per thread");
  __shared__ __attribute__((aligned(16))) unsigned char blds[SCR_BLK * LROW];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
  const int r = lane & 31, h = lane >> 5;
  const int g = (int)blockIdx.x * SCR_WAVES + wv;
  const bool have_g = g < ngroups;  // (a wavefront without a group still helps with the staging)
  scr_bf16x8 a[4][KS];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const size_t pos = (size_t)(have_g ? g : 0) * CULL_W + rt * 32 + r;
      a[rt][s] = *reinterpret_cast<const scr_bf16x8 *>(A + pos * K + s * 16 + h * 8);
    }
  // this launch's blocks of Gaussians: [blk0, blk1) of the ceil(N / SCR_BLK) there are (all of them, or one column chunk of
  // a pass whose sweep runs beside the next chunk's screen: mcx_murray.hip)
  const int nblk = blk1;
  const int b0 = blk0 + (int)blockIdx.y * bchunk, b1 = b0 + bchunk < nblk ? b0 + bchunk : nblk;
  if (b0 >= b1) return;  // (the whole workgroup)
  scr_u32x4 hold[PER];
  {
    const scr_u32x4 *src = reinterpret_cast<const scr_u32x4 *>(B + (size_t)b0 * SCR_BLK * K);
#pragma unroll
    for (int u = 0; u < PER; ++u) hold[u] = src[(int)threadIdx.x + u * NT];
  }
  const int members = have_g ? (nact - g * CULL_W < CULL_W ? nact - g * CULL_W : CULL_W) : 0;
  unsigned long long kept = 0;
  for (int b = b0; b < b1; ++b) {
    __syncthreads();  // the previous block has been consumed
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int c = (int)threadIdx.x + u * NT;  // piece c of the block: row c / (K/8), piece c % (K/8) of it
      *reinterpret_cast<scr_u32x4 *>(blds + (c / (K / 8)) * LROW + (c % (K / 8)) * 16) = hold[u];
    }
    __syncthreads();
    {  // the next block on its way while this one is multiplied (unconditional: the last block once more)
      const scr_u32x4 *src = reinterpret_cast<const scr_u32x4 *>(B + (size_t)(b + 1 < b1 ? b + 1 : b) * SCR_BLK * K);
#pragma unroll
      for (int u = 0; u < PER; ++u) hold[u] = src[(int)threadIdx.x + u * NT];
    }
    if (!have_g) continue;
    unsigned m32[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      scr_bf16x8 bf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s)
        bf[s] = *reinterpret_cast<const scr_bf16x8 *>(blds + (ct * 32 + r) * LROW + (s * 16 + h * 8) * 2);
      float mn = __builtin_inff();
      auto tile_min = [](const scr_f32x16 &acc) {
        const float m0 = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
        const float m1 = __builtin_fminf(__builtin_fminf(acc[3], acc[4]), acc[5]);
        const float m2 = __builtin_fminf(__builtin_fminf(acc[6], acc[7]), acc[8]);
        const float m3 = __builtin_fminf(__builtin_fminf(acc[9], acc[10]), acc[11]);
        const float m4 = __builtin_fminf(__builtin_fminf(acc[12], acc[13]), acc[14]);
        const float m5 = __builtin_fminf(__builtin_fminf(m0, m1), acc[15]);
        const float m6 = __builtin_fminf(__builtin_fminf(m2, m3), m4);
        return __builtin_fminf(m5, m6);
      };
      // (at most 256 registers asked for above: the accumulators are plain VGPRs and the minimum reads them directly --
      // from AGPRs every one of them costs a v_accvgpr_read first.  Measured and not faster: two row tiles at a time on
      // two accumulators (32-D +3 %, 16-D spills at five wavefronts per SIMD), s_setprio around the products, five
      // wavefronts per SIMD instead of four, grids of exactly one to four chipfuls of workgroups: the matrix cores stay
      // 65 % busy, SQ_VALU_MFMA_BUSY_CYCLES)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        scr_f32x16 acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt][s], bf[s], acc, 0, 0, 0);
        mn = __builtin_fminf(mn, tile_min(acc));
      }
      const bool col_ok = b * SCR_BLK + ct * 32 + r < N;
      const unsigned long long bal = __ballot(col_ok && !(mn > 0.0f));  // both halves hold the same 32 Gaussians
      m32[ct] = (unsigned)bal | (unsigned)(bal >> 32);
    }
    const unsigned long long w0 = (unsigned long long)m32[0] | ((unsigned long long)m32[1] << 32);
    const unsigned long long w1 = (unsigned long long)m32[2] | ((unsigned long long)m32[3] << 32);
    if (lane == 0) {
      excl[(size_t)(2 * b) * ngroups + g] = w0;
      if (2 * b + 1 < excl_words) excl[(size_t)(2 * b + 1) * ngroups + g] = w1;
    }
    kept += (unsigned long long)(__popcll(w0) + __popcll(w1)) * (unsigned long long)members;
  }
  if (lane == 0 && kept) atomicAdd(nkept + ((blockIdx.x * SCR_WAVES + wv + blockIdx.y) & (CULL_NCOUNT - 1)), kept);
}
