// A user's likelihood as SOURCE with no built-in counterpart (MCX_VL_SOURCE, whole-vector form): a "banana" -- a
// Gaussian in (x0, x1 + b x0^2 - 100 b) and in the remaining coordinates, par = (b, 1/s0^2, 1/s^2).  Plain fp32
// operations in a fixed left-to-right order (no fma), so that four numpy float32 expressions on the host reproduce
// it bit for bit (tests/test_gpu_user_source.py): the VLFunc of src/vlfunc.hh:9-12 for one parameter set.
__device__ float mcx_user_loglike(const float *x, int d, const float *par)
{
  const float b = par[0], w0 = par[1], w = par[2];
  const float y1 = (x[1] + (b * x[0]) * x[0]) - 100.0f * b;
  float acc = (x[0] * x[0]) * w0;
  acc = acc + (y1 * y1) * w;
  for (int k = 2; k < d; ++k) acc = acc + (x[k] * x[k]) * w;
  return -0.5f * acc;
}
