// Cycle stamps of one kNN wave at its phase boundaries (table build | distances of two queries | then for the first query: bound | compaction | rank+emit | and the whole round).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DALIGNNET_KNN_STAMP -I alignnet-3d_amd/csrc tools/microbench/knn_phases.hip -o tools/microbench/knn_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_dgcnn.h"
using namespace alignnet;

// self-check: one thread per query, same key expression, k smallest by (key, index)
__global__ void knn_check(const float* pcs1, const float* pcs2, const float* center, int B, int N, int k, const int* nn, int* bad)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x, cloud = blockIdx.y, tower = cloud >= B, b = cloud - tower * B;
  if (q >= N) return;
  const float* pc = (tower ? pcs2 : pcs1) + (size_t)b * N * 3;
  const float cx = center[cloud * 3], cy = center[cloud * 3 + 1], cz = center[cloud * 3 + 2];
  const float qx = pc[q * 3] - cx, qy = pc[q * 3 + 1] - cy, qz = pc[q * 3 + 2] - cz;
  const float qq = fmaf(qz, qz, fmaf(qy, qy, qx * qx));
  const float q2x = -2.0f * qx, q2y = -2.0f * qy, q2z = -2.0f * qz;
  unsigned long long prev = 0; bool first = true;
  unsigned long long want[32];
  for (int s = 0; s < k; ++s) {
    unsigned long long best = ~0ull;
    for (int j = 0; j < N; ++j) {
      const float x = pc[j * 3] - cx, y = pc[j * 3 + 1] - cy, z = pc[j * 3 + 2] - cz;
      const float inner = fmaf(q2z, z, fmaf(q2y, y, q2x * x));
      const unsigned long long kk = ((unsigned long long)fkey(qq + inner + fmaf(z, z, fmaf(y, y, x * x))) << 32) | (unsigned)j;
      if ((first || kk > prev) && kk < best) best = kk;
    }
    want[s] = best; prev = best; first = false;
  }
  const int* got = nn + ((size_t)cloud * N + q) * k;
  int miss = 0;
  for (int s = 0; s < k; ++s) {
    bool f = false;
    for (int u = 0; u < k; ++u) f |= got[u] == (int)(want[s] & 0xffffffffu);
    miss += !f;
  }
  if (miss && atomicAdd(bad, 1) == 0) {   // first differing query: (index, key) pairs of both selections
    for (int s = 0; s < k; ++s) {
      const int j = got[s];
      const float x = pc[j * 3] - cx, y = pc[j * 3 + 1] - cy, z = pc[j * 3 + 2] - cz;
      const float inner = fmaf(q2z, z, fmaf(q2y, y, q2x * x));
      printf("q %d cloud %d  got %4d key %08x | want %4d key %08x\n", q, cloud, j, fkey(qq + inner + fmaf(z, z, fmaf(y, y, x * x))),
             (int)(want[s] & 0xffffffffu), (unsigned)(want[s] >> 32));
    }
  }
}
int main(int argc, char** argv)
{
  const int N = argc > 1 ? atoi(argv[1]) : 4096, B = argc > 2 ? atoi(argv[2]) : 64, k = 20;
  std::vector<float> h((size_t)2 * B * N * 3), c((size_t)2 * B * 3, 0.f);
  srand(1);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (size_t i = 0; i + 6 <= h.size(); i += 30) { h[i + 3] = h[i]; h[i + 4] = h[i + 1]; h[i + 5] = h[i + 2]; }   // duplicate points: exact distance ties
  float *d1, *d2, *dc; int* nn;
  hipMalloc(&d1, (size_t)B * N * 3 * 4); hipMalloc(&d2, (size_t)B * N * 3 * 4); hipMalloc(&dc, c.size() * 4);
  hipMalloc(&nn, (size_t)2 * B * N * k * 4);
  hipMemcpy(d1, h.data(), (size_t)B * N * 3 * 4, hipMemcpyHostToDevice);
  hipMemcpy(d2, h.data() + (size_t)B * N * 3, (size_t)B * N * 3 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    launch_knn(0, 0, d1, d2, dc, B, N, k, nn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_knn_stamp), sizeof(st));
    printf("N=%d clouds=%d  %.3f ms | cycles: table %lld | last round: distances %lld bound %lld compaction %lld rank+emit %lld round %lld\n", N, 2 * B, ms,
           st[1] - st[0], st[2] - st[7], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[7]);
  }
  int* dbad; hipMalloc(&dbad, 4); hipMemset(dbad, 0, 4);
  const int cb = 2 * B < 4 ? 2 * B : 4;   // check the first clouds
  hipLaunchKernelGGL(knn_check, dim3((N + 63) / 64, cb), dim3(64), 0, 0, d1, d2, dc, B, N, k, nn, dbad);
  int bad = -1; hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("self-check: %d of %d queries differ from the brute-force (key, index) selection\n", bad, cb * N);
  return bad != 0;
}
