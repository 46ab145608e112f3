// What does hipEventElapsedTime return for two STOP events bound to two consecutive kernels (hipExtLaunchKernelGGL, no start event)?
// build: hipcc --offload-arch=gfx950 -O2 -o event_bind event_bind.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
int main() {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t eA, eB, sB, r0, r1; hipEventCreate(&eA); hipEventCreate(&eB); hipEventCreate(&sB); hipEventCreate(&r0); hipEventCreate(&r1);
  const long long us = 100;   // wall_clock64 ticks at 100 MHz
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(r0, s);
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, eA, 0, 100 * us, (int*)nullptr);   // A: 100 us
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, sB, eB, 0, 300 * us, (int*)nullptr);        // B: 300 us, with a start event too
    hipEventRecord(r1, s);
    hipStreamSynchronize(s);
    float ab = -1, sb = -1, aa = -1, tot = -1, ra = -1;
    hipError_t e1 = hipEventElapsedTime(&ab, eA, eB), e2 = hipEventElapsedTime(&sb, sB, eB), e3 = hipEventElapsedTime(&aa, eA, eA), e4 = hipEventElapsedTime(&tot, r0, r1),
               e5 = hipEventElapsedTime(&ra, r0, eA);
    printf("stopA->stopB %.3f ms (%d) | startB->stopB %.3f ms (%d) | stopA->stopA %.3f (%d) | record0->stopA %.3f (%d) | record0->record1 %.3f (%d)\n", ab, e1, sb, e2, aa, e3, ra, e5, tot, e4);
  }
  return 0;
}
