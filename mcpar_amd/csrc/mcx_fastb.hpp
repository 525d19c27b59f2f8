// mcx_fastb.hpp -- k_fused_fastb<LPC2, BPL, MAIN, LIK>: the hot-path kernel k_fused_fast (mcx_device.hpp) with BPL
// consecutive 4-parameter blocks per lane instead of one: a chain of LPC = LPC2 * BPL blocks takes LPC2 lanes.
//
// Why: everything a lane does per CHAIN -- the log-likelihood difference, the acceptance test, the five selects that
// take or keep the state, the accept counters, the ballot, the broadcast of the acceptance draw, the loop itself --
// is repeated by every lane of the chain; with two (four) blocks per lane that share is paid half (a quarter) as
// often per parameter, the lane-group reductions lose their first stage(s) to plain in-lane adds, and a wavefront
// carries BPL independent Philox / Box-Muller streams whose instructions fill each other's dependency stalls.
//
// Same functions on the same values in the same order as k_fused_fast, hence the same bits:
//   * block q = q2 * BPL + b of the chain draws Philox counter (t, g, q, 0) whichever lane holds it;
//   * the xor-butterfly over the block index starts with the stages xor 1 (.. xor BPL/2), which pair
//     blocks of ONE lane here: p[q] + p[q ^ 1] is the same sum in either order; the remaining stages are the lane-group
//     butterfly over q2;
//   * the acceptance draw of ACCEPT block blk is taken by lane blk % LPC2 of the chain and broadcast.
// Diagonal factor, d % 4 == 0, no accept mask, no pre-generated normals (the plain hot path only).
#pragma once
#include "mcx_device.hpp"

namespace mcx {

// FULL (BPL = 2 only): proposals x' = x + T z with the full lower-triangular factor (src/mcpar.cc:302-312 with
// covar_setup's factor, :454-484), in a MIRRORED layout: lane q2 of a chain of NB = 2 LPC2 blocks holds blocks q2 and
// NB - 1 - q2.  Row block r of a lower-triangular factor needs column blocks 0 .. r only, and a wavefront must run whatever
// ANY of its lanes needs: with one block per lane (k_fused_fast<..., FULL>) that is all NB column blocks for every lane --
// the zeros above the diagonal are multiplied through as exact no-ops.  Here every lane's FIRST block (q2 < LPC2) needs
// column blocks 0 .. LPC2 - 1 at most, so its multiply-adds and reads of T for the upper half of the columns are never
// issued at all: 3/4 of the work of the square, decided at compile time; z travels between the lanes of a chain by DPP
// quad permutes (LPC2 <= 4) instead of through LDS, and what a lane does per chain is paid once per two blocks.
// Same bits: every row adds its columns in ascending order; the butterfly over the block index pairs (q2, q2 ^ 1), ... on
// the first and on the second blocks separately -- the second blocks run it mirrored, which permutes the operands of
// commutative additions only -- and its last stage (blocks 0 .. LPC2 - 1 against the rest) is the in-lane add.
template <int LPC2, int BPL, bool MAIN, int LIK, bool FULL>
__device__ __forceinline__ uint32_t fused_fastb_body(const SegArgs &a);

template <int LPC2, int BPL, bool MAIN, int LIK, bool FULL = false>
__global__ __launch_bounds__(BLOCK) void k_fused_fastb(const SegArgs a)
{
  const uint32_t wacc = fused_fastb_body<LPC2, BPL, MAIN, LIK, FULL>(a);
  tuner_epilogue(a, wacc);  // (every thread, also those the body let go early)
}

template <int LPC2, int BPL, bool MAIN, int LIK, bool FULL = false>
__device__ __forceinline__ uint32_t fused_fastb_body(const SegArgs &a)
{
  static_assert(BPL == 2 || BPL == 4, "two or four blocks per lane");
  static_assert(!FULL || (BPL == 2 && LPC2 <= 4), "full covariance: two mirrored blocks per lane, z by DPP quad permutes");
  constexpr int NB = LPC2 * BPL;  // blocks per chain
  // FULL: the factor by column, [(4 qq + c) * NB + row block] = column 4 qq + c of the block's rows (0, 2, 1, 3): one
  // read is the pair of packed operands of the two multiply-adds a column costs a block (k_fused_fast's table)
  __shared__ __attribute__((aligned(16))) float4 lds_T[FULL ? 4 * NB * NB : 1];
  if (FULL) {
    const int dd = a.d;
    for (int i = threadIdx.x; i < 16 * NB * NB; i += BLOCK) {
      const int h = i & 3, qv = (i >> 2) % NB, c = ((i >> 2) / NB) & 3, qq = (i >> 2) / (4 * NB);
      const int row = 4 * qv + (h == 0 ? 0 : (h == 1 ? 2 : (h == 2 ? 1 : 3))), col = 4 * qq + c;
      reinterpret_cast<float *>(lds_T)[i] = (row < dd && col < dd) ? a.T[row * dd + col] : 0.0f;
    }
    if (LIK != LIK_MIX) __syncthreads();
  }
  static_assert(LIK == LIK_ROSEN1 || LIK == LIK_GAUSS || LIK == LIK_MIX || LIK == LIK_USER, "hot-path likelihoods (or a user's source)");
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
  const size_t chain = gid / LPC2;
  const int q2 = (int)(gid % LPC2);
  if (chain >= (size_t)a.n) return 0u;
  const uint32_t g = a.g0 + (uint32_t)chain;
  // block b of this lane is block qb[b] of the chain: consecutive, or (FULL) mirrored.  The consecutive case keeps the
  // forms `off + 4 b` / `k00 + 4 b` (constant offsets from one base: immediates in the memory instructions -- spelled
  // per block they cost the 32-D mixture 17 registers and a wavefront per SIMD)
  const int k00 = 4 * BPL * q2;
  const size_t off = chain * (size_t)d + k00;
  int qb[BPL], kb[BPL];
  size_t offb[BPL];
#pragma unroll
  for (int b = 0; b < BPL; ++b) {
    qb[b] = FULL ? (b == 0 ? q2 : NB - 1 - q2) : q2 * BPL + b;
    kb[b] = FULL ? 4 * qb[b] : k00 + 4 * b;
    offb[b] = FULL ? chain * (size_t)d + kb[b] : off + 4 * b;
  }
  const size_t row0 = FULL ? chain * (size_t)d : off;  // base of the sample-store pointer; block b at + rowk[b]
  int rowk[BPL];
#pragma unroll
  for (int b = 0; b < BPL; ++b) rowk[b] = FULL ? kb[b] : 4 * b;
  bool live[BPL];                               // d % 4 == 0: a block is whole or absent (d = 12, 20, ...)
  f32x2 xe[BPL], xo[BPL], me[BPL], mo[BPL], se[BPL], so[BPL], te[BPL], to[BPL], gme[BPL], gmo[BPL];
  float gs[BPL][4];
#pragma unroll
  for (int b = 0; b < BPL; ++b) {
    const int k0 = kb[b];
    live[b] = k0 < d;
    xe[b] = xo[b] = me[b] = mo[b] = se[b] = so[b] = te[b] = to[b] = gme[b] = gmo[b] = f32x2{0, 0};
    gs[b][0] = gs[b][1] = gs[b][2] = gs[b][3] = 0.0f;
    if (live[b]) {
      const float4 f = *reinterpret_cast<const float4 *>(a.x + offb[b]);
      xe[b] = f32x2{f.x, f.z}; xo[b] = f32x2{f.y, f.w};
      if (!FULL) {
        te[b] = f32x2{a.T[(k0 + 0) * d + k0 + 0], a.T[(k0 + 2) * d + k0 + 2]};
        to[b] = f32x2{a.T[(k0 + 1) * d + k0 + 1], a.T[(k0 + 3) * d + k0 + 3]};
      }
      if (MAIN && a.init_moments) {  // src/mcpar.cc:99-104
        se[b] = f32x2{FPEPS, FPEPS}; so[b] = f32x2{FPEPS, FPEPS};
      } else if (MAIN) {
        const float4 m = *reinterpret_cast<const float4 *>(a.mu + offb[b]);
        const float4 p = *reinterpret_cast<const float4 *>(a.psum2 + offb[b]);
        me[b] = f32x2{m.x, m.z}; mo[b] = f32x2{m.y, m.w};
        se[b] = f32x2{p.x, p.z}; so[b] = f32x2{p.y, p.w};
      }
      if (LIK == LIK_GAUSS) {  // lik = mu[d], 1/sigma^2[d]
        gme[b] = f32x2{a.lik[k0 + 0], a.lik[k0 + 2]}; gmo[b] = f32x2{a.lik[k0 + 1], a.lik[k0 + 3]};
        gs[b][0] = a.lik[d + k0 + 0]; gs[b][1] = a.lik[d + k0 + 1]; gs[b][2] = a.lik[d + k0 + 2]; gs[b][3] = a.lik[d + k0 + 3];
      }
    }
  }
  float ly = a.ly[chain];
  uint32_t cnt = 0, wacc = 0;
  f32x2 al01 = {0, 0}, al23 = {0, 0};  // log of the four acceptance draws of this lane's current ACCEPT block
  uint32_t ablk = 0xffffffffu;
  float *sx = a.samp_x ? a.samp_x + row0 : nullptr;
  float *sl = a.samp_x ? a.samp_ly + chain : nullptr;
  const size_t sx_stride = (size_t)a.n * d, sl_stride = (size_t)a.n;

  // the first log2(BPL) stages of the butterfly over the block index, inside the lane; the lane group does the rest
  auto blocks_sum = [&](const float p[BPL]) -> float {
    if (FULL) return group_sum<LPC2>(p[0]) + group_sum<LPC2>(p[1]);  // mirrored blocks: two butterflies, then the last stage
    if (BPL == 2) return group_sum<LPC2>(p[0] + p[1]);
    return group_sum<LPC2>((p[0] + p[1]) + (p[BPL == 4 ? 2 : 0] + p[BPL == 4 ? 3 : 1]));
  };

  // every load of the chain state is awaited here, once (see k_fused_fast): no vmcnt wait may end up in the step loop
#pragma unroll
  for (int b = 0; b < BPL; ++b)
    asm volatile("" ::"v"(xe[b]), "v"(xo[b]), "v"(te[b]), "v"(to[b]), "v"(me[b]), "v"(mo[b]), "v"(se[b]), "v"(so[b]),
                 "v"(gme[b]), "v"(gmo[b]), "v"(gs[b][0]), "v"(gs[b][1]), "v"(gs[b][2]), "v"(gs[b][3]));
  asm volatile("" ::"v"(ly));
  // 1/pwgt by scalar loads (constant address space): a plain global load would queue behind the sample stores
  const __attribute__((address_space(4))) float *wtab = (const __attribute__((address_space(4))) float *)a.winv;
  // acceptance draw: Philox block (t >> 2) of the ACCEPT stream serves steps 4b..4b+3; lane q2 draws block b for
  // b % LPC2 == q2, once per 4 * LPC2 steps
  auto refresh = [&](uint32_t blk) {
    if ((blk & ~(uint32_t)(LPC2 - 1)) != ablk) {
      ablk = blk & ~(uint32_t)(LPC2 - 1);
      const u32x4 aw = philox4x32_10(ablk + (uint32_t)q2, g, 0u, 0u, a.seed, ST_ACCEPT);
      al01 = accept_lu_x2(aw.x, aw.y);
      al23 = accept_lu_x2(aw.z, aw.w);
    }
  };
  auto step = [&](const int s, const float lu) {
    const uint32_t t = a.t0 + (uint32_t)s;
    f32x2 pe[BPL], po[BPL];
    if (!FULL) {
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        f32x2 ze, zo;
        normal4_packed(philox4x32_10(t, g, (uint32_t)(q2 * BPL + b), 0u, a.seed, ST_LOCAL), ze, zo);
        pe[b] = fma2(te[b], ze, xe[b]);  // src/mcpar.cc:302-312
        po[b] = fma2(to[b], zo, xo[b]);
      }
    } else {
      float zv[BPL][4];
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        f32x2 ze, zo;
        normal4_packed(philox4x32_10(t, g, (uint32_t)qb[b], 0u, a.seed, ST_LOCAL), ze, zo);
        zv[b][0] = ze.x; zv[b][1] = zo.x; zv[b][2] = ze.y; zv[b][3] = zo.y;
        pe[b] = xe[b];  // rows (0, 2) and (1, 3) of the block: every row accumulates its columns in ascending order
        po[b] = xo[b];
      }
#pragma unroll
      for (int qq = 0; qq < NB; ++qq) {
        // column block qq is the first block of lane qq (qq < LPC2) or the second of lane NB - 1 - qq: one DPP move each
        constexpr int HALF = LPC2;
        const int holder = qq < HALF ? qq : NB - 1 - qq, which = qq < HALF ? 0 : 1;
        float zc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) zc[c] = LPC2 == 1 ? zv[which][c] : quad_bcast<(LPC2 > 1 ? LPC2 : 2)>(zv[which][c], holder);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 t1 = lds_T[(qq * 4 + c) * NB + qb[1]];  // the second block's rows: every column block may matter
          pe[1] = fma2(f32x2{t1.x, t1.y}, splat2(zc[c]), pe[1]);
          po[1] = fma2(f32x2{t1.z, t1.w}, splat2(zc[c]), po[1]);
          if (qq < HALF) {  // the first block (q2 < LPC2): columns beyond block LPC2 - 1 are above its diagonal for every lane
            const float4 t0 = lds_T[(qq * 4 + c) * NB + qb[0]];
            pe[0] = fma2(f32x2{t0.x, t0.y}, splat2(zc[c]), pe[0]);
            po[0] = fma2(f32x2{t0.z, t0.w}, splat2(zc[c]), po[0]);
          }
        }
        // The accumulators are pinned down once per column block: left alone, the multiply-adds are sunk behind all 48
        // reads of the step (they are pure: nothing holds them in place), every T entry read stays alive until then --
        // 380 registers wanted, 124 spilled at one wavefront per SIMD.
        asm volatile("" : "+v"(pe[0]), "+v"(po[0]), "+v"(pe[1]), "+v"(po[1]));
      }
    }
    float lyt;
#ifdef MCX_USER_LIK
    if (LIK == LIK_USER) {  // a user's source (mcx_user.hip): this lane's BPL blocks of the proposal
#if MCX_USER_LIK == 1
      float acc[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        const float xb[4] = {pe[b].x, po[b].x, pe[b].y, po[b].y};
        acc[b] = live[b] ? ::mcx_user_block(xb, 4, kb[b], d, a.lik) : 0.0f;
      }
      lyt = ::mcx_user_finish(blocks_sum(acc), d, a.lik);
#else
      // whole-vector form: the chain's proposal, contiguous in LDS (an odd stride between chains: the lanes of a wavefront
      // read x[k] of different chains from different banks), every lane of the chain calls the function -- with four blocks
      // per lane a 16-D chain IS one lane: nothing is evaluated twice
      constexpr int CS = 4 * LPC2 * BPL + 1;
      __shared__ float xs[(BLOCK / LPC2) * CS];
      float *mine = xs + ((int)threadIdx.x / LPC2) * CS;
#pragma unroll
      for (int b = 0; b < BPL; ++b)
        if (live[b]) {
          float *dst = mine + kb[b];
          dst[0] = pe[b].x; dst[1] = po[b].x; dst[2] = pe[b].y; dst[3] = po[b].y;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();  // a chain never spans wavefronts; LDS is in order per wavefront
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      lyt = ::mcx_user_loglike(mine, d, a.lik);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();  // this step's reads precede the next step's writes
#endif
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
          float s2[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b) {
            s2[b] = 0.0f;
            if (live[b]) {
              const float4 m = *reinterpret_cast<const float4 *>(&lds_means[c * d + kb[b]]);
              const f32x2 ae = pe[b] - f32x2{m.x, m.z}, ao = po[b] - f32x2{m.y, m.w};
              s2[b] = __builtin_fmaf(ae.x, ae.x, 0.0f);
              s2[b] = __builtin_fmaf(ao.x, ao.x, s2[b]);
              s2[b] = __builtin_fmaf(ae.y, ae.y, s2[b]);
              s2[b] = __builtin_fmaf(ao.y, ao.y, s2[b]);
            }
          }
          e[c] = __builtin_fmaf(-0.5f, blocks_sum(s2), lds_logw[c]);
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
      float acc[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        acc[b] = 0.0f;
        if (LIK == LIK_ROSEN1) {  // src/rosenbrock.cc:4-21 on the pairs (x0,x1), (x2,x3)
          const f32x2 t1 = splat2(1.0f) - pe[b];
          const f32x2 t2 = fma2(-pe[b], pe[b], po[b]);
          const f32x2 term = fma2(splat2(100.0f) * t2, t2, t1 * t1);
          if (live[b]) acc[b] = term.x + term.y;
        } else {  // src/rosenbrock.cc:44-61: acc = fma((0.5 a) a, 1/sigma^2, acc) for k = 0..3 in order
          const f32x2 ae = pe[b] - gme[b], ao = po[b] - gmo[b];
          const f32x2 he = (splat2(0.5f) * ae) * ae, ho = (splat2(0.5f) * ao) * ao;
          if (live[b]) {
            acc[b] = __builtin_fmaf(he.x, gs[b][0], 0.0f);
            acc[b] = __builtin_fmaf(ho.x, gs[b][1], acc[b]);
            acc[b] = __builtin_fmaf(he.y, gs[b][2], acc[b]);
            acc[b] = __builtin_fmaf(ho.y, gs[b][3], acc[b]);
          }
        }
      }
      lyt = 0.0f - blocks_sum(acc);
    }
    // src/mcpar.cc:62-75 (cfac = 1 for local proposals): log u < ly' - ly
    const bool take = accept_local(lyt, ly, lu);
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
      xe[b] = take ? pe[b] : xe[b];
      xo[b] = take ? po[b] : xo[b];
    }
    ly = take ? lyt : ly;
    cnt += take ? 1u : 0u;
    wacc += (uint32_t)__popcll(__ballot(take && q2 == 0));
    if (MAIN) {
      const f32x2 w2 = splat2(wtab[a.isamp0 + s]);  // src/mcpar.cc:186-187
#pragma unroll
      for (int b = 0; b < BPL; ++b) {
        const f32x2 de = xe[b] - me[b], dO = xo[b] - mo[b];  // src/mcpar.cc:199-202
        me[b] = fma2(de, w2, me[b]);
        mo[b] = fma2(dO, w2, mo[b]);
        se[b] = fma2(de, xe[b] - me[b], se[b]);
        so[b] = fma2(dO, xo[b] - mo[b], so[b]);
      }
      if (s == a.snap_after) {  // snapshot for the next exchange (src/mcpar.cc:202-208)
#pragma unroll
        for (int b = 0; b < BPL; ++b)
          if (live[b]) {
            const f32x2 ve = se[b] * w2, vo = so[b] * w2;
            float4 *slot = reinterpret_cast<float4 *>(a.musig_own + 2 * offb[b]);
            slot[0] = make_float4(me[b].x, ve.x, mo[b].x, vo.x);
            slot[1] = make_float4(me[b].y, ve.y, mo[b].y, vo.y);
            if (a.sig_out) *reinterpret_cast<float4 *>(a.sig_out + offb[b]) = make_float4(ve.x, vo.x, ve.y, vo.y);
          }
      }
      if (sx) {  // src/mcpar.cc:177-182
        if (a.samp_stride <= 1) {
#pragma unroll
          for (int b = 0; b < BPL; ++b)
            if (live[b]) *reinterpret_cast<float4 *>(sx + rowk[b]) = make_float4(xe[b].x, xo[b].x, xe[b].y, xo[b].y);
          if (q2 == 0) *sl = ly;
          sx += sx_stride;
          sl += sl_stride;
        } else if ((a.isamp0 + s) % a.samp_stride == 0) {  // thinned store: row = isamp / stride
          const size_t row = (size_t)((a.isamp0 + s) / a.samp_stride);
#pragma unroll
          for (int b = 0; b < BPL; ++b)
            if (live[b]) *reinterpret_cast<float4 *>(sx + row * sx_stride + rowk[b]) = make_float4(xe[b].x, xo[b].x, xe[b].y, xo[b].y);
          if (q2 == 0) sl[row * sl_stride] = ly;
        }
      }
    }
  };
  auto one_step = [&](const int s) {
    const uint32_t t = a.t0 + (uint32_t)s, blk = t >> 2;
    refresh(blk);
    const uint32_t wi = t & 3u;
    const float mine = wi == 0u ? al01.x : (wi == 1u ? al01.y : (wi == 2u ? al23.x : al23.y));
    step(s, as_f32(group_bcast<LPC2>(as_u32(mine), blk & (uint32_t)(LPC2 - 1), q2)));
  };
  int s = 0;
#if MCX_FAST_UNROLL4
  // (four steps per iteration from an aligned step on: k_fused_fast's driver loop, mcx_device.hpp)
  for (; s < a.nsteps && ((a.t0 + (uint32_t)s) & 3u); ++s) one_step(s);
  for (; s + 4 <= a.nsteps; s += 4) {
    const uint32_t blk = (a.t0 + (uint32_t)s) >> 2;
    refresh(blk);
    const uint32_t holder = blk & (uint32_t)(LPC2 - 1);
    const float lu4[4] = {as_f32(group_bcast<LPC2>(as_u32(al01.x), holder, q2)), as_f32(group_bcast<LPC2>(as_u32(al01.y), holder, q2)),
                          as_f32(group_bcast<LPC2>(as_u32(al23.x), holder, q2)), as_f32(group_bcast<LPC2>(as_u32(al23.y), holder, q2))};
#pragma unroll
    for (int u = 0; u < 4; ++u) step(s + u, lu4[u]);
  }
#endif
  for (; s < a.nsteps; ++s) one_step(s);

#pragma unroll
  for (int b = 0; b < BPL; ++b)
    if (live[b]) {
      *reinterpret_cast<float4 *>(a.x + offb[b]) = make_float4(xe[b].x, xo[b].x, xe[b].y, xo[b].y);
      if (MAIN) {
        *reinterpret_cast<float4 *>(a.mu + offb[b]) = make_float4(me[b].x, mo[b].x, me[b].y, mo[b].y);
        *reinterpret_cast<float4 *>(a.psum2 + offb[b]) = make_float4(se[b].x, so[b].x, se[b].y, so[b].y);
      }
    }
  if (q2 == 0) {
    a.ly[chain] = ly;
    a.acc_cnt[chain] += cnt;
  }
  // one slot per wavefront, owned by it (no atomics); the tuner adds the slots up whatever their number
  if ((threadIdx.x & 63u) == 0 && wacc && !a.tun.on) a.acc_slots[gid >> 6] += wacc;  // (else: tuner_epilogue)
  return wacc;
}

}  // namespace mcx
