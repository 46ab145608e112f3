// ICP refinement of the network's prediction on the FULL clouds, gfx950 only.
//
// Replaces icp.icp_p2point (reference icp.py:69-78: o3.registration_icp with
// TransformationEstimationPointToPoint(with_constraint=True, with_scaling=False)) as called from the evaluation loop
// (train.py:463-484: radius 0.1, `--its` iterations, init = get_mat_angle(prediction)).  Open3D (a private fork,
// README.md:32) is not part of the reference tree; the loop and the z-constrained point-to-point estimate are
// restated in oracle/icp_ref.py, which this kernel follows step for step (same order of evaluate / estimate / stop).
//
// One workgroup per pair, all arithmetic in fp64 (Open3D computes in double; nearest-neighbour decisions then agree
// with the oracle).  The target cloud sits in LDS as doubles (structure of arrays: no conversion in the inner loop),
// the source cloud is re-read from HBM/L2 each iteration (a few thousand points).  Brute force n1 x n2 per
// iteration: at the datasets' cloud sizes that is a few million distance evaluations per pair -- a KD-tree would be
// slower to build than this is to run.
// Round 6: sixteen waves per pair instead of four, and FOUR LANES PER SOURCE POINT (lane s of a quad scans targets s, s + 4, ...; the
// quad's (distance, index) minima meet in two shuffles, equal distances going to the lower index = the oracle's argmin): 256 pairs x 256
// threads put one wave on every SIMD and 1500 points on 256 threads (six rounds of a serial 1500-long scan each); now a CU holds four
// waves per SIMD and a scan is 375 long.
// And the scan is CERTIFIED in fp32 before it is decided in fp64: one pass takes, per lane, the smallest and second smallest fp32 distance of its slice
// (packed arithmetic, two targets per instruction) and where the smallest sits; the quad's minimum m gives a threshold m + 4 eps with
// eps >= |fp32 distance - exact distance| for every target that can matter (icp_prefilter_eps: the rounding of the transformed source point to fp32, of
// the differences, of the three products and sums).  Any target whose fp64 distance equals the minimum has an fp32 distance under the threshold, so: a
// lane whose smallest is under it and whose second smallest is not evaluates that ONE target in fp64 (the old arithmetic); a lane with two or more under
// it (a near-tie inside its slice: rare) walks its slice in fp64 as before; the quad's (distance, index) minima meet as before.  Same decisions as the
// all-fp64 scan at a third of its vector-issue time (a first form with the fp64 evaluation inside the scan loop spilled around the branch: 7.7 ms).
#include "engine.h"
#include <cmath>
#include <vector>

namespace {

int fail(const alignnet_handle* h, const std::string& m) { h->err = m; return 1; }

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(h, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

constexpr int kIcpThreads = 1024, kIcpSums = 12, kIcpSplit = 4;   // threads per pair; lanes per source point
constexpr float kIcpFar = 1e18f;      // padding of the fp32 slices: a distance of 3e36, under no threshold
typedef float icp_f32x2 __attribute__((ext_vector_type(2)));

// slice-major fp32 copy of the LDS-resident targets: lane s of a quad reads targets s, s + 4, ... as consecutive floats, two per 8-byte read
__host__ __device__ constexpr int icp_slice_len(int lds_points) { return (((lds_points + 3) / 4 + 1) & ~1) + 2; }
constexpr size_t icp_lds_bytes(int lds_points) { return (size_t)lds_points * 24 + (size_t)3 * kIcpSplit * icp_slice_len(lds_points) * 4; }

// eps >= |e32 - d| for a target at exact squared distance d <= M from a source point with |coordinates| <= P (u = 2^-24): the source point is rounded to
// fp32 (u P per coordinate), each difference once more (u (P + a) in all, a = |difference|), so a squared difference moves by <= 2 a u (P + a) + u^2 (P + a)^2,
// and the product and two fmas round the running sum three times (3 u d): eps = u (2 sqrt(3) P sqrt(d) + 5 d) + 3 u^2 (P + sqrt(d))^2.  Returned with room:
// 2^-23 (4 P sqrt(M) + 6 M) + 2^-44 (P + sqrt(M))^2.
__device__ __forceinline__ float icp_prefilter_eps(float P, float M)
{
  const float r = sqrtf(M);
  return 1.1920929e-7f * (4.f * P * r + 6.f * M) + 5.6843419e-14f * (P + r) * (P + r);
}

struct IcpArgs {
  const float* pts[2];          // point blobs
  const long long* off;         // [n + 1][2] row offsets into the blobs
  const int* rows;              // [B] example rows, or null: pair b is row b
  const double* init;           // [B][16] row-major 4x4
  double radius; int its;
  int lds_points;               // target points that fit in LDS
  double* out;                  // [B][16]
  double* fitness; double* rmse; int* iters;   // [B] each, may be null
};

__device__ __forceinline__ void block_reduce(double (&v)[kIcpSums], double* red /*[waves][kIcpSums]*/, double* tot /*[kIcpSums]*/)
{
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < kIcpSums; ++k) red[(threadIdx.x >> 6) * kIcpSums + k] = v[k];
  __syncthreads();
  if (threadIdx.x < kIcpSums) {
    double s = 0.0;
    for (int w = 0; w < kIcpThreads / 64; ++w) s += red[w * kIcpSums + threadIdx.x];   // fixed order: deterministic
    tot[threadIdx.x] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kIcpThreads) void icp_kernel(const IcpArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double tgt[];   // [3][lds_points] doubles | [3][4 slices][S4] floats
  __shared__ double T[12];            // rows 0..2 of the 4x4
  __shared__ double red[(kIcpThreads / 64) * kIcpSums], tot[kIcpSums];
  const int b = blockIdx.x, tid = threadIdx.x, sub = tid & (kIcpSplit - 1);
  const long long row = a.rows ? a.rows[b] : b;
  const long long s_lo = a.off[row * 2], n1 = a.off[(row + 1) * 2] - s_lo;
  const long long t_lo = a.off[row * 2 + 1], n2 = a.off[(row + 1) * 2 + 1] - t_lo;
  const float* src = a.pts[0] + s_lo * 3;
  const float* dst = a.pts[1] + t_lo * 3;
  if (tid < 12) T[tid] = a.init[(size_t)b * 16 + tid];
  const int nl = (int)min((long long)a.lds_points, n2);
  const double* tx = tgt; const double* ty = tgt + a.lds_points; const double* tz = tgt + 2 * a.lds_points;
  const int S4 = icp_slice_len(a.lds_points);
  float* t32 = reinterpret_cast<float*>(tgt + 3 * (size_t)a.lds_points);   // [coordinate][slice][S4]
  for (int j = tid; j < 3 * kIcpSplit * S4; j += kIcpThreads) t32[j] = kIcpFar;
  __syncthreads();
  for (int j = tid; j < nl; j += kIcpThreads) {
    const float x = dst[j * 3], y = dst[j * 3 + 1], z = dst[j * 3 + 2];
    tgt[j] = (double)x; tgt[a.lds_points + j] = (double)y; tgt[2 * a.lds_points + j] = (double)z;
    const int o = (j & (kIcpSplit - 1)) * S4 + (j >> 2);
    t32[o] = x; t32[kIcpSplit * S4 + o] = y; t32[2 * kIcpSplit * S4 + o] = z;
  }
  __syncthreads();
  const int Lp = ((nl + kIcpSplit - 1) / kIcpSplit + 1) & ~1;   // slice positions walked (even; the padding behind a slice's end is kIcpFar)
  const float* s32x = t32 + sub * S4; const float* s32y = s32x + kIcpSplit * S4; const float* s32z = s32y + kIcpSplit * S4;
  const double r2 = a.radius * a.radius;
  // the correspondence sums are taken about a pivot near the clouds (the first target point), not about the origin: one-pass centring
  // sum p q - n mean(p) mean(q) cancels |offset|^2 / extent^2 of its digits (a cloud 4 km from the origin: the rotation came out 6e-8 off
  // Open3D's two-pass estimate, tests/test_icp_gpu.py::test_icp_exact_ties_and_far_frames); about the pivot the terms are of the clouds' size
  const double cx = n2 > 0 ? (double)dst[0] : 0.0, cy = n2 > 0 ? (double)dst[1] : 0.0, cz = n2 > 0 ? (double)dst[2] : 0.0;
  double fit_prev = 0.0, rmse_prev = 0.0, fit = 0.0, rmse = 0.0;
  int k = 0;
  if (n1 > 0 && n2 > 0)
    for (k = 0;; ++k) {
      // ---- evaluate(T): nearest target point of every transformed source point, sums over the inliers ----
      double v[kIcpSums];
#pragma unroll
      for (int q = 0; q < kIcpSums; ++q) v[q] = 0.0;
      for (long long i0 = 0; i0 < n1; i0 += kIcpThreads / kIcpSplit) {
        const long long i = i0 + (tid / kIcpSplit);
        const bool active = i < n1;                      // (inactive quads run along on the last point: the shuffles below want every lane)
        const long long ic = active ? i : n1 - 1;
        const double sx = src[ic * 3], sy = src[ic * 3 + 1], sz = src[ic * 3 + 2];
        const double px = T[0] * sx + T[1] * sy + T[2] * sz + T[3];
        const double py = T[4] * sx + T[5] * sy + T[6] * sz + T[7];
        const double pz = T[8] * sx + T[9] * sy + T[10] * sz + T[11];
        double best = 1e300; int bj = 0x7fffffff;
        {
          // one fp32 pass over the slice: its smallest distance m1 with the position it sits at, and its second smallest m2 (v_med3 of the
          // ordered pair and the newcomer) -- four vector instructions per target next to the packed arithmetic, no branch
          const float pxf = (float)px, pyf = (float)py, pzf = (float)pz;
          const icp_f32x2 qx2 = {pxf, pxf}, qy2 = {pyf, pyf}, qz2 = {pzf, pzf};
          float m1 = 3.0e38f, m2 = 3.0e38f;
          int im = 0;
#pragma unroll 4
          for (int m = 0; m < Lp; m += 2) {
            const icp_f32x2 X = *reinterpret_cast<const icp_f32x2*>(s32x + m), Y = *reinterpret_cast<const icp_f32x2*>(s32y + m), Z = *reinterpret_cast<const icp_f32x2*>(s32z + m);
            const icp_f32x2 dx = qx2 - X, dy = qy2 - Y, dz = qz2 - Z;
            const icp_f32x2 e = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const bool lt = e[u] < m1;
              m2 = __builtin_amdgcn_fmed3f(m1, m2, e[u]);   // m1 <= m2: the median of the three is the new second smallest
              im = lt ? m + u : im;
              m1 = fminf(m1, e[u]);
            }
          }
          float qm = fminf(m1, __shfl_xor(m1, 1));
          qm = fminf(qm, __shfl_xor(qm, 2));
          const float P = fabsf(pxf) + fabsf(pyf) + fabsf(pzf) + 1e-30f;
          const float thr = qm + 4.f * icp_prefilter_eps(P, 1.01f * qm + 1e-6f * P * P);
          // every target whose fp64 distance is the minimum has an fp32 distance <= thr.  A lane with ONE such target decides it in fp64; a lane
          // with two or more (a near-tie inside its slice: rare) walks its slice in fp64 as the all-fp64 kernel did
          if (m2 <= thr) {
#pragma unroll 1
            for (int j = sub; j < nl; j += kIcpSplit) {
              const double ddx = px - tx[j], ddy = py - ty[j], ddz = pz - tz[j];
              const double d = ddx * ddx + ddy * ddy + ddz * ddz;
              if (d < best) { best = d; bj = j; }   // strict: the first index of this lane's slice wins ties
            }
          } else if (m1 <= thr) {
            const int j = sub + kIcpSplit * im;
            if (j < nl) {
              const double ddx = px - tx[j], ddy = py - ty[j], ddz = pz - tz[j];
              best = ddx * ddx + ddy * ddy + ddz * ddz; bj = j;
            }
          }
        }
#pragma unroll 1
        for (long long j = nl + sub; j < n2; j += kIcpSplit) {   // clouds larger than the LDS budget: the tail comes from L2
          const double dx = px - (double)dst[j * 3], dy = py - (double)dst[j * 3 + 1], dz = pz - (double)dst[j * 3 + 2];
          const double d = dx * dx + dy * dy + dz * dz;
          if (d < best) { best = d; bj = (int)j; }
        }
        // the quad's minimum; equal distances go to the lower index (oracle: argmin = the first index)
#pragma unroll
        for (int o = 1; o < kIcpSplit; o <<= 1) {
          const double ob = __shfl_xor(best, o);
          const int oj = __shfl_xor(bj, o);
          if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (active && sub == 0 && best <= r2) {
          const double qx = bj < nl ? tx[bj] : (double)dst[(long long)bj * 3];
          const double qy = bj < nl ? ty[bj] : (double)dst[(long long)bj * 3 + 1];
          const double qz = bj < nl ? tz[bj] : (double)dst[(long long)bj * 3 + 2];
          const double ax = px - cx, ay = py - cy, az = pz - cz, bx = qx - cx, by = qy - cy, bz = qz - cz;
          v[0] += 1.0; v[1] += ax; v[2] += ay; v[3] += az; v[4] += bx; v[5] += by; v[6] += bz;
          v[7] += ax * bx + ay * by; v[8] += ax * by - ay * bx; v[9] += best;
        }
      }
      block_reduce(v, red, tot);
      const double cnt = tot[0];
      fit = cnt / (double)n1;
      rmse = cnt > 0.0 ? sqrt(tot[9] / cnt) : 0.0;
      if (k > 0 && fabs(fit - fit_prev) < 1e-6 && fabs(rmse - rmse_prev) < 1e-6) break;
      if (k == a.its) break;
      fit_prev = fit; rmse_prev = rmse;
      // ---- estimate: rotation about z + translation minimising sum |Rz p + t - q|^2 over the correspondences ----
      if (tid == 0 && cnt > 0.0) {
        const double apx = tot[1] / cnt, apy = tot[2] / cnt, apz = tot[3] / cnt;   // means about the pivot
        const double aqx = tot[4] / cnt, aqy = tot[5] / cnt, aqz = tot[6] / cnt;
        const double sxx = tot[7] - cnt * (apx * aqx + apy * aqy);
        const double sxy = tot[8] - cnt * (apx * aqy - apy * aqx);
        const double th = atan2(sxy, sxx), c = cos(th), s = sin(th);
        const double mpx = cx + apx, mpy = cy + apy, mqx = cx + aqx, mqy = cy + aqy;
        const double tx = mqx - (c * mpx - s * mpy), ty = mqy - (s * mpx + c * mpy), tz = aqz - apz;
        // T <- U T,  U = [[c,-s,0,tx],[s,c,0,ty],[0,0,1,tz]]
        double n[12];
        for (int col = 0; col < 4; ++col) {
          n[col] = c * T[col] - s * T[4 + col];
          n[4 + col] = s * T[col] + c * T[4 + col];
          n[8 + col] = T[8 + col];
        }
        n[3] += tx; n[7] += ty; n[11] += tz;
        for (int q = 0; q < 12; ++q) T[q] = n[q];
      }
      __syncthreads();
    }
  __syncthreads();
  if (tid < 12) a.out[(size_t)b * 16 + tid] = T[tid];
  if (tid >= 12 && tid < 16) a.out[(size_t)b * 16 + tid] = tid == 15 ? 1.0 : 0.0;
  if (tid == 0) {
    if (a.fitness) a.fitness[b] = fit;
    if (a.rmse) a.rmse[b] = rmse;
    if (a.iters) a.iters[b] = k;
  }
}

// shared driver: tables already on the device
int run_icp(alignnet_handle* h, const float* d_p0, const float* d_p1, const long long* d_off, const int* d_rows, long long max_n2,
            int B, const double* init, double radius, int its, double* out, double* fitness, double* rmse, int* iters)
{
  if (!init || !out) return fail(h, "icp: null init / out");
  if (!(radius > 0.0) || its < 0) return fail(h, "icp: radius must be > 0 and its >= 0");
  double *d_init = nullptr, *d_out = nullptr, *d_fr = nullptr; int* d_it = nullptr;
  HIP_TRY(h, hipMalloc(&d_init, (size_t)B * 16 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&d_out, (size_t)B * 16 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&d_fr, (size_t)B * 2 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&d_it, (size_t)B * sizeof(int)));
  HIP_TRY(h, hipMemcpyAsync(d_init, init, (size_t)B * 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  IcpArgs a;
  a.pts[0] = d_p0; a.pts[1] = d_p1; a.off = d_off; a.rows = d_rows; a.init = d_init; a.radius = radius; a.its = its;
  const long long budget = (150 * 1024) / 36;   // doubles x 3 + floats x 3 per point within one CU's LDS
  a.lds_points = (int)std::max<long long>(1, std::min(budget, max_n2));
  a.out = d_out; a.fitness = d_fr; a.rmse = d_fr + B; a.iters = d_it;
  static alignnet::PerDeviceOnce attr;
  if (attr.need(h->cfg.device)) {
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(icp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
    attr.mark(h->cfg.device);
  }
  hipLaunchKernelGGL(icp_kernel, dim3(B), dim3(kIcpThreads), icp_lds_bytes(a.lds_points), h->stream, a);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, d_out, (size_t)B * 16 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (fitness) HIP_TRY(h, hipMemcpyAsync(fitness, d_fr, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (rmse) HIP_TRY(h, hipMemcpyAsync(rmse, d_fr + B, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (iters) HIP_TRY(h, hipMemcpyAsync(iters, d_it, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  hipFree(d_init); hipFree(d_out); hipFree(d_fr); hipFree(d_it);
  return 0;
}

}  // namespace

extern "C" int alignnet_icp_refine(alignnet_handle* h, const float* points1, const float* points2, const int64_t* offsets, int32_t B,
                                   const double* init, double radius, int32_t its, double* out, double* fitness, double* rmse,
                                   int32_t* iterations)
{
  if (!h) return 1;
  if (!offsets || B < 1) return fail(h, "alignnet_icp_refine: null offsets or B < 1");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  long long max_n2 = 0;
  for (int i = 0; i < B; ++i) {
    if (offsets[(i + 1) * 2] < offsets[i * 2] || offsets[(i + 1) * 2 + 1] < offsets[i * 2 + 1]) return fail(h, "alignnet_icp_refine: offsets must be non-decreasing");
    max_n2 = std::max<long long>(max_n2, offsets[(i + 1) * 2 + 1] - offsets[i * 2 + 1]);
  }
  const size_t n0 = (size_t)offsets[B * 2], n1 = (size_t)offsets[B * 2 + 1];
  if ((n0 && !points1) || (n1 && !points2)) return fail(h, "alignnet_icp_refine: null point blob");
  float *d0 = nullptr, *d1 = nullptr; long long* doff = nullptr;
  HIP_TRY(h, hipMalloc(&d0, std::max<size_t>(n0, 1) * 3 * sizeof(float)));
  HIP_TRY(h, hipMalloc(&d1, std::max<size_t>(n1, 1) * 3 * sizeof(float)));
  HIP_TRY(h, hipMalloc(&doff, (size_t)(B + 1) * 2 * sizeof(long long)));
  if (n0) HIP_TRY(h, hipMemcpyAsync(d0, points1, n0 * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (n1) HIP_TRY(h, hipMemcpyAsync(d1, points2, n1 * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(doff, offsets, (size_t)(B + 1) * 2 * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  const int rc = run_icp(h, d0, d1, doff, nullptr, max_n2, B, init, radius, its, out, fitness, rmse, iterations);
  hipFree(d0); hipFree(d1); hipFree(doff);
  return rc;
}

extern "C" int alignnet_icp_refine_dataset(alignnet_handle* h, const int32_t* rows, int32_t B, const double* init, double radius,
                                           int32_t its, double* out, double* fitness, double* rmse, int32_t* iterations)
{
  if (!h) return 1;
  alignnet::DatasetTables t;
  if (!alignnet_dataset_tables(h, &t)) return fail(h, "alignnet_icp_refine_dataset: no dataset uploaded");
  if (!rows || B < 1) return fail(h, "alignnet_icp_refine_dataset: null rows or B < 1");
  for (int i = 0; i < B; ++i)
    if (rows[i] < 0 || rows[i] >= t.n) return fail(h, "alignnet_icp_refine_dataset: row " + std::to_string(rows[i]) + " out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  int* d_rows = nullptr;
  HIP_TRY(h, hipMalloc(&d_rows, (size_t)B * sizeof(int)));
  HIP_TRY(h, hipMemcpyAsync(d_rows, rows, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  // the largest target cloud is not known on the host: size the LDS stage for the budget, the kernel clamps per pair
  const int rc = run_icp(h, t.pts[0], t.pts[1], t.off, d_rows, (150 * 1024) / 36, B, init, radius, its, out, fitness, rmse, iterations);
  hipFree(d_rows);
  return rc;
}
