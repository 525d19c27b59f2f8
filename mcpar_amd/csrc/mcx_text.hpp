// mcx_text.hpp -- the sample rows as the TEXT the reference's MCout::output prints (src/mcout.cc:41-45: every field
// through `ostream << float`, two blanks behind it, a newline behind the last column of a row), produced on the GPU
// from the HBM-resident sample store.  The characters of a number come from fmtg6 (fmt_g6.hpp: exact, integer only,
// the routine the facade uses on the host), one number per lane; three passes:
//   k_text_sizes   bytes of every number's field, summed per workgroup
//   k_text_scan    one workgroup: the sums become byte offsets, the last one the total
//   k_text_write   the fields again, placed: neighbouring lanes write neighbouring bytes
// Included by mcx_engine.hip only.
#pragma once
#include "fmt_g6.hpp"
#include "mcx_device.hpp"

namespace mcx {

// number i of the dump: row i / (d + 1) = (kept step, chain), column i % (d + 1) (the last one is the log-likelihood);
// sl == null: sx holds the rows as they are printed, d + 1 columns each
__device__ __forceinline__ uint32_t text_bits(const float *__restrict__ sx, const float *__restrict__ sl, size_t i, int d)
{
  if (!sl) return as_u32(sx[i]);
  const size_t ncol = (size_t)d + 1, r = i / ncol;
  const int c = (int)(i - r * ncol);
  return as_u32(c < d ? sx[r * (size_t)d + c] : sl[r]);
}

__device__ __forceinline__ unsigned text_field_bytes(const fmtg6::Text &t, size_t i, int d)
{
  return (unsigned)t.len + 2u + (((i + 1) % ((size_t)d + 1)) == 0 ? 1u : 0u);
}

// sum over the workgroup (BLOCK threads); every thread gets it
__device__ __forceinline__ unsigned text_block_sum(unsigned v, unsigned *lds)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned t = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) t += lds[w];
  __syncthreads();
  return t;
}

static __global__ __launch_bounds__(BLOCK) void k_text_sizes(const float *__restrict__ sx, const float *__restrict__ sl,
                                                             size_t count, int d, unsigned long long *__restrict__ wg)
{
  __shared__ unsigned lds[BLOCK / 64];
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  unsigned b = 0;
  if (i < count) b = text_field_bytes(fmtg6::format(text_bits(sx, sl, i, d)), i, d);
  const unsigned s = text_block_sum(b, lds);
  if (threadIdx.x == 0) wg[blockIdx.x] = s;
}

// wg[0 .. nwg): sums -> exclusive offsets, wg[nwg] = total.  One workgroup of 1024 threads, a contiguous stretch each.
static __global__ __launch_bounds__(1024) void k_text_scan(unsigned long long *__restrict__ wg, size_t nwg)
{
  __shared__ unsigned long long part[1024];
  const size_t per = (nwg + 1023) / 1024, a = (size_t)threadIdx.x * per, b = a + per < nwg ? a + per : nwg;
  unsigned long long s = 0;
  for (size_t k = a; k < b; ++k) s += wg[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {  // (1024 additions: nothing against the passes around it)
    unsigned long long run = 0;
    for (int t = 0; t < 1024; ++t) {
      const unsigned long long v = part[t];
      part[t] = run;
      run += v;
    }
    wg[nwg] = run;
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (size_t k = a; k < b; ++k) {
    const unsigned long long v = wg[k];
    wg[k] = run;
    run += v;
  }
}

static __global__ __launch_bounds__(BLOCK) void k_text_write(const float *__restrict__ sx, const float *__restrict__ sl,
                                                             size_t count, int d, const unsigned long long *__restrict__ wg,
                                                             char *__restrict__ out)
{
  __shared__ unsigned wave_sum[BLOCK / 64];
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  fmtg6::Text t;
  t.lo = t.hi = 0;
  t.len = 0;
  unsigned bytes = 0;
  if (i < count) {
    t = fmtg6::format(text_bits(sx, sl, i, d));
    bytes = text_field_bytes(t, i, d);
  }
  // exclusive prefix of `bytes` over the workgroup: within the wavefront by shuffles, across wavefronts through LDS
  const unsigned lane = threadIdx.x & 63u;
  unsigned incl = bytes;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o);
    if (lane >= (unsigned)o) incl += v;
  }
  if (lane == 63u) wave_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned before = 0;
  for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) before += wave_sum[w];
  if (i >= count) return;
  char *p = out + wg[blockIdx.x] + before + (incl - bytes);
  for (int k = 0; k < t.len; ++k) p[k] = (char)((k < 8 ? t.lo >> (8 * k) : t.hi >> (8 * (k - 8))) & 0xffu);
  p[t.len] = ' ';
  p[t.len + 1] = ' ';
  if (bytes == (unsigned)t.len + 3u) p[t.len + 2] = '\n';
}

}  // namespace mcx
