// ubench.hip -- VALU issue-cost microbenchmark for gfx950 (build: hipcc --offload-arch=gfx950 -O2).
// Each kernel runs ITER iterations of 8 independent chains of one instruction per lane; grid fills
// the chip with W waves per SIMD.  Prints ns per wave-instruction per SIMD and the ratio to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(name, decl, body, fin_stmt)                                                   \
  __global__ void name(unsigned *out, unsigned seed)                               \
  {                                                                                \
    decl;                                                                          \
    for (int i = 0; i < ITER; ++i) { body; }                                       \
    fin_stmt;                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = fin;                              \
  }

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define DF(k) float a##k = seed * 1e-9f + k + threadIdx.x;
#define DU(k) unsigned a##k = seed + k * 77 + threadIdx.x;
#define DL(k) unsigned long long a##k = seed + k * 77 + threadIdx.x;
#define DP(k) float a##k = seed * 1e-9f + k + threadIdx.x, b##k = a##k + 0.5f;
#define FINF unsigned fin = __float_as_uint(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define FINU unsigned fin = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define FINL unsigned fin = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define FINP unsigned fin = __float_as_uint(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7)

#define OP_FMA(k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a##k));
#define OP_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a##k) : "v"((unsigned)a##k), "v"(seed) : "vcc");
#define OP_MUL24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
#define OP_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
#define OP_CVT(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##k));
#define OP_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(seed));
#define OP_CND2(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a##k) : "v"(seed), "s"(smask));
#define OP_CMPCND(k) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(seed) : "vcc");
#define OP_MED3(k) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a##k) : "v"(sf));
#define OP_MAXF(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##k) : "v"(sf));
#define OP_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a##k) : : "vcc");
#define OP_CMPADDC(k) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a##k) : "v"(seed) : "vcc");
#define OP_ADD(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_BITOP3(k) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a##k) : "v"(seed));
#define OP_BITOP3S(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##k) : "v"(seed), "s"(sseed));
#define OP_FREXPM(k) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a##k));
#define OP_LDEXP(k) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a##k) : "v"(seed));
#define OP_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a##k) : "v"(seed));
#define OP_DPP(k) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a##k));

#define DV(k) f2 a##k = {seed * 1e-9f + k + threadIdx.x, 0.5f + k};
#define FINV unsigned fin = __float_as_uint(a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y)
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a##k));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a##k));

// the Murray sweep's per-dimension patterns (counted per instruction of the group, like the others):
//   mov3: v_mov s->v ; v_fma -x, s, v ; v_fmac     (two scalar operands: one has to go through a VGPR)
//   fma2: v_fma (one scalar operand) ; v_fmac      (the other operand already in a VGPR)
//   mov64: v_mov_b64 s[..]->v[..] counted alone
//   lds_b128: ds_read_b128 of one address by all lanes, 1 per 8 v_fma (what feeding 4 dimensions' operands costs)
#define OP_MOV3(k) asm volatile("v_mov_b32 %1, %2\n\tv_fma_f32 %1, -%0, %3, %1\n\tv_fmac_f32 %0, %1, %1" : "+v"(a##k), "=&v"(b##k) : "s"(sf), "s"(sg));
#define OP_FMA2(k) asm volatile("v_fma_f32 %1, -%0, %2, %1\n\tv_fmac_f32 %0, %1, %1" : "+v"(a##k), "+v"(b##k) : "s"(sf));
#define OP_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(a##k) : "s"(sl));
#define OP_FMADPP(k) asm volatile("v_fmac_f32_dpp %0, %1, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a##k) : "v"(b##k));
#define DSF float sf = __uint_as_float(__builtin_amdgcn_readfirstlane(seed)) * 1e-9f, sg = sf + 1.0f;
#define DSL unsigned long long sl = __builtin_amdgcn_readfirstlane(seed) | ((unsigned long long)__builtin_amdgcn_readfirstlane(seed + 1) << 32);
__global__ void k_mov3(unsigned *out, unsigned seed)
{
  DSF R8(DP)
  for (int i = 0; i < ITER / 3; ++i) { R8(OP_MOV3) }
  FINP; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
__global__ void k_fma2(unsigned *out, unsigned seed)
{
  DSF R8(DP)
  for (int i = 0; i < ITER / 2; ++i) { R8(OP_FMA2) }
  FINP; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
__global__ void k_mov64(unsigned *out, unsigned seed)
{
  DSL R8(DL)
  for (int i = 0; i < ITER; ++i) { R8(OP_MOV64) }
  FINL; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
__global__ void k_fmac_dpp(unsigned *out, unsigned seed)
{
  R8(DP)
  for (int i = 0; i < ITER; ++i) { R8(OP_FMADPP) }
  FINP; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
// 8 v_fma + 1 broadcast ds_read_b128 per group, counted as 8 instructions: the LDS read rides along if this equals fma
__global__ void k_fma_lds(unsigned *out, unsigned seed)
{
  __shared__ float4 tab[256];
  tab[threadIdx.x] = make_float4(seed * 1e-9f, 1.f, 2.f, 3.f);
  __syncthreads();
  R8(DF)
  float4 v = tab[seed & 255];
  for (int i = 0; i < ITER; ++i) {
    const float4 w = tab[(seed + i) & 255];  // wave-uniform address: every lane reads the same 16 bytes
    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a0) : "v"(v.x)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a1) : "v"(v.y));
    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a2) : "v"(v.z)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a3) : "v"(v.w));
    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a4) : "v"(v.x)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a5) : "v"(v.y));
    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a6) : "v"(v.z)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a7) : "v"(v.w));
    v = w;
  }
  FINF; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
//   pk2: two chains per lane: v_pk_fma t2 = -x2 * (s,s) + (m',m') with both scalars taken from ONE SGPR pair by op_sel ; v_pk_fma arg2
#define OP_PK2(k) asm volatile("v_pk_fma_f32 %1, %0, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 %0, %1, %1, %0" : "+v"(a##k), "+v"(b##k) : "s"(sl));
#define DV2(k) f2 a##k = {seed * 1e-9f + k + threadIdx.x, 0.5f + k}, b##k = a##k;
__global__ void k_pk2(unsigned *out, unsigned seed)
{
  DSL R8(DV2)
  for (int i = 0; i < ITER / 2; ++i) { R8(OP_PK2) }
  unsigned fin = __float_as_uint(a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + b0.x + b7.y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
KERNEL(k_fma, R8(DF), R8(OP_FMA), FINF)
KERNEL(k_xor, R8(DU), R8(OP_XOR), FINU)
KERNEL(k_mul_lo_u32, R8(DU), R8(OP_MULLO), FINU)
KERNEL(k_mul_hi_u32, R8(DU), R8(OP_MULHI), FINU)
KERNEL(k_mad_u64_u32, R8(DL), R8(OP_MAD64), FINL)
KERNEL(k_mul_u32_u24, R8(DU), R8(OP_MUL24), FINU)
KERNEL(k_sqrt, R8(DF), R8(OP_SQRT), FINF)
KERNEL(k_rcp, R8(DF), R8(OP_RCP), FINF)
KERNEL(k_cvt_f32_u32, R8(DU), R8(OP_CVT), FINU)
KERNEL(k_cndmask, R8(DU), R8(OP_CND), FINU)
KERNEL(k_cndmask_s, unsigned long long smask = 0x5555555555555555ull | __builtin_amdgcn_readfirstlane(seed); R8(DU), R8(OP_CND2), FINU)
__global__ void k_cmp_cnd(unsigned *out, unsigned seed)
{
  R8(DU)
  for (int i = 0; i < ITER / 2; ++i) { R8(OP_CMPCND) }
  FINU; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
KERNEL(k_med3, float sf = seed * 1e-9f; R8(DF), R8(OP_MED3), FINF)
KERNEL(k_max_f32, float sf = seed * 1e-9f; R8(DF), R8(OP_MAXF), FINF)
KERNEL(k_addc, R8(DU), R8(OP_ADDC), FINU)
__global__ void k_cmp_addc(unsigned *out, unsigned seed)
{
  R8(DU)
  for (int i = 0; i < ITER / 2; ++i) { R8(OP_CMPADDC) }
  FINU; out[blockIdx.x * blockDim.x + threadIdx.x] = fin;
}
KERNEL(k_add_u32, R8(DU), R8(OP_ADD), FINU)
KERNEL(k_add_f32_dpp, R8(DF), R8(OP_DPP), FINF)
KERNEL(k_bitop3, R8(DU), R8(OP_BITOP3), FINU)
KERNEL(k_bitop3_s, unsigned sseed = __builtin_amdgcn_readfirstlane(seed * 3u); R8(DU), R8(OP_BITOP3S), FINU)
KERNEL(k_frexp_mant, R8(DF), R8(OP_FREXPM), FINF)
KERNEL(k_ldexp, R8(DF), R8(OP_LDEXP), FINF)
KERNEL(k_and_or, R8(DU), R8(OP_ANDOR), FINU)
KERNEL(k_pk_fma, R8(DV), R8(OP_PKFMA), FINV)
KERNEL(k_pk_mul, R8(DV), R8(OP_PKMUL), FINV)

typedef void (*kfn)(unsigned *, unsigned);
struct K { const char *name; kfn f; };
int main()
{
  K ks[] = {{"fma", k_fma},{"xor", k_xor},{"mul_lo_u32", k_mul_lo_u32},{"mul_hi_u32", k_mul_hi_u32},{"mad_u64_u32", k_mad_u64_u32},{"mul_u32_u24", k_mul_u32_u24},{"sqrt", k_sqrt},{"rcp", k_rcp},{"cvt_f32_u32", k_cvt_f32_u32},{"cndmask", k_cndmask},{"cndmask sgpr", k_cndmask_s},{"cmp+cndmask", k_cmp_cnd},{"med3_f32", k_med3},{"max_f32", k_max_f32},{"addc_co_u32", k_addc},{"cmp+addc", k_cmp_addc},{"add_u32", k_add_u32},{"add_f32_dpp", k_add_f32_dpp},{"bitop3_b32", k_bitop3},{"bitop3 v,v,s", k_bitop3_s},{"frexp_mant_f32", k_frexp_mant},{"ldexp_f32", k_ldexp},{"and_or_b32", k_and_or},{"pk_fma", k_pk_fma},{"pk_mul", k_pk_mul},{"sweep mov3", k_mov3},{"sweep fma2", k_fma2},{"mov_b64 s->v", k_mov64},{"fmac_f32_dpp", k_fmac_dpp},{"8fma+lds_b128", k_fma_lds},{"sweep pk2", k_pk2}};
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned *out; CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int wps = 1; wps <= 8; wps *= 2) {
    printf("== %d wave(s) per SIMD (%d CUs, clock %d MHz)\n", wps, cus, p.clockRate / 1000);
    double base = 0;
    for (auto &k : ks) {
      const int blocks = cus * wps;  // 256 threads = 4 waves -> one per SIMD per block
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CHK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        CHK(hipEventRecord(a)); hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 12345u); CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      const double ninst = (double)ITER * 8 * wps;  // wave-instructions per SIMD
      const double ns = best * 1e6 / ninst;
      if (base == 0) base = ns;
      printf("  %-14s %7.3f ns/wave-instr/SIMD  = %5.2f x v_fma  (~%.1f cycles at 2.4 GHz)\n", k.name, ns, ns / base, ns * 2.4);
    }
  }
  return 0;
}
