// What does a seam between two dependent phases cost on this part: a kernel boundary on one stream, or a device-wide barrier inside one launch?
// (the head chains gemm -> row-BatchNorm -> gemm -> ... are ~5 dependent launches of 5 - 9 us each; a persistent kernel would
//  replace four boundaries by four grid barriers.)  Measured here, per seam, for G workgroups of 256 threads (G = 16 .. 512):
//   A  N dependent launches of a kernel whose body is one read-modify-write per workgroup  -> us per launch (boundary + minimal body)
//   B  one launch of N phases separated by an atomic-counter grid barrier (device-scope release / acquire, every workgroup arrives and spins)
//   C  as B, with a realistic producer in front of every barrier: each workgroup writes 64 KB that the NEXT phase reads from another workgroup
//      (the release then has dirty lines to write back -- what "the statistics are ready" means for a GEMM tile)
//   D  the phases of C as N dependent launches (what the chain is today)
// build: hipcc --offload-arch=gfx950 -O2 -o launch_vs_barrier launch_vs_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // device scope: this workgroup's writes are visible before the arrival
    while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <bool WORK>
__global__ void phases(float* p, float* buf, unsigned* counter, int n)
{
  const unsigned G = gridDim.x;
  for (int ph = 0; ph < n; ++ph) {
    if (WORK) {   // 64 KB per workgroup: written here, read (another workgroup's) in the next phase
      float* mine = buf + (size_t)blockIdx.x * 16384;
      const float* other = buf + (size_t)((blockIdx.x + 1) % G) * 16384;
      float acc = 0.f;
      if (ph) for (int i = threadIdx.x; i < 16384; i += 256) acc += other[i];
      for (int i = threadIdx.x; i < 16384; i += 256) mine[i] = acc + (float)ph;
    } else if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
    grid_barrier(counter, (unsigned)(ph + 1) * G);
  }
}

__global__ void phase_launch(float* buf, int ph)
{
  const unsigned G = gridDim.x;
  float* mine = buf + (size_t)blockIdx.x * 16384;
  const float* other = buf + (size_t)((blockIdx.x + 1) % G) * 16384;
  float acc = 0.f;
  if (ph) for (int i = threadIdx.x; i < 16384; i += 256) acc += other[i];
  __syncthreads();   // (C reads the neighbour's block of the previous phase before anyone overwrites: a launch boundary gives that for free; here `mine` != `other`)
  for (int i = threadIdx.x; i < 16384; i += 256) mine[i] = acc + (float)ph;
}

int main()
{
  hipStream_t s; CK(hipStreamCreate(&s));
  float *p, *buf; unsigned* c;
  CK(hipMalloc(&p, 4096 * 4)); CK(hipMalloc(&buf, (size_t)512 * 16384 * 4)); CK(hipMalloc(&c, 4));
  CK(hipMemset(p, 0, 4096 * 4)); CK(hipMemset(buf, 0, (size_t)512 * 16384 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200, REP = 7;
  printf("%6s | %18s | %24s | %36s | %34s\n", "G", "A: us per launch", "B: us per grid barrier", "C: us per phase (64 KB/WG + barrier)", "D: us per launch (64 KB/WG each)");
  for (int G : {16, 32, 64, 128, 256, 512}) {
    std::vector<float> ta, tb, tc, td;
    for (int r = 0; r < REP; ++r) {
      float ms;
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, s, p);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ta.push_back(ms * 1e3f / N);
      CK(hipMemsetAsync(c, 0, 4, s));
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(phases<false>, dim3(G), dim3(256), 0, s, p, buf, c, N);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tb.push_back(ms * 1e3f / N);
      CK(hipMemsetAsync(c, 0, 4, s));
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(phases<true>, dim3(G), dim3(256), 0, s, p, buf, c, N);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tc.push_back(ms * 1e3f / N);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(phase_launch, dim3(G), dim3(256), 0, s, buf, i);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); td.push_back(ms * 1e3f / N);
    }
    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%6d | %18.2f | %24.2f | %36.2f | %34.2f\n", G, med(ta), med(tb), med(tc), med(td));
  }
  // the same producer as C split into N launches (what the chain is today): us per launch with 64 KB per workgroup of work
  return 0;
}
