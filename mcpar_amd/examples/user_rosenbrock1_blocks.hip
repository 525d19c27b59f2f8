// A user's likelihood as SOURCE (MCX_VL_SOURCE, include/mcx.h), block form: Rosenbrock1 of src/rosenbrock.cc:4-21
// restated with the operation order of the built-in, so that a run with it must equal the
// built-in's -- and the CPU oracle's -- bit for bit.
#define MCX_USER_BLOCK_FORM
__device__ float mcx_user_block(const float xb[4], int nv, int k0, int d, const float *par)
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
  return acc;
}
#define MCX_USER_FINISH
__device__ float mcx_user_finish(float sum, int d, const float *par) { return 0.0f - sum; }
