// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 as a function of waves per SIMD and of independent
// accumulators per wave.  Prints cycles per MFMA per SIMD (s_memtime of wave 0 around the loop).
//   hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue && ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, long long* cyc, int iters, float a0, float b0)
{
  f32x16 acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int waves)
{
  float* out; long long* cyc; long long h;
  hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters, 1.f, 2.f);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave = (double)h / (iters * 4.0 * NACC);
  const int per_simd = (waves + 3) / 4;
  printf("acc/wave %d  waves/WG %2d (%d per SIMD): %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", NACC, waves, per_simd, per_wave,
         per_wave / per_simd);
  hipFree(out); hipFree(cyc);
}

int main()
{
  for (int w : {1, 4, 8, 12, 16}) run<4>(w);
  for (int w : {1, 8}) run<1>(w);
  for (int w : {1, 8}) run<2>(w);
  for (int w : {1, 8}) run<8>(w);
  return 0;
}
