// How does v_mfma_f32_32x32x2_f32 round its fp32 accumulation?  A Gram matrix accumulated over n rows in MFMA registers carries n / 2
// roundings per element: with round-to-nearest their sum grows like sqrt(n) ulps, with truncation like n ulps and it is a BIAS (DESIGN.md 4.4:
// the z2 / z3 statistics derived from Gram(h1) / Gram(h2) divide by variances that can be 1e-3 of the Gram's entries).
// Every lane feeds a = b = 1 + 2^-12 (the product 1 + 2^-11 + 2^-24 needs 25 bits), K = 2 per instruction; the exact sum after s steps is
// 2 s (1 + 2^-11 + 2^-24).  Prints accumulated / exact - 1 for a few s, next to the same recurrence done with fmaf (round-to-nearest) on the host.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_round mfma_round.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int steps, float a, float* out)
{
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
  if (threadIdx.x == 0) *out = acc[0];
}
int main()
{
  float* d; hipMalloc(&d, 4);
  const float a = 1.f + ldexpf(1.f, -12);
  const double p = (double)a * (double)a;
  for (int steps : {32, 512, 2048, 40960, 400000}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, steps, a, d);
    float got; hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
    float h = 0.f;
    for (int s = 0; s < steps; ++s) { h = fmaf(a, a, h); h = fmaf(a, a, h); }   // two round-to-nearest fused steps per instruction
    float h1 = 0.f;
    for (int s = 0; s < steps; ++s) h1 = (float)((double)h1 + 2.0 * p);         // one rounding per instruction (the K = 2 sum formed exactly first)
    const double exact = 2.0 * steps * p;
    printf("steps %7d: mfma %.9g  rel err %+.3e | host fma x2 (RNE) %+.3e | host exact-pair then RNE %+.3e\n", steps, got, got / exact - 1.0, h / exact - 1.0, h1 / exact - 1.0);
  }
  return 0;
}
