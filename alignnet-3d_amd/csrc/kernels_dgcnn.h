// DGCNN backbone, eval mode (models/tp8.py:30-46 with utils/tf_util_dgcnn.py:638-706), gfx950 only.
//
//   knn_kernel    pairwise_distance + knn: k nearest neighbours (self included) per point.  The reference
//                 materialises the [B,N,N] distance matrix (64 MB per cloud at N=4096); here one wave owns one query
//                 point, keeps its N candidate distances in registers and finds the k-th smallest by bisection on
//                 order-preserving integer keys, so N^2 never leaves the register file.
//   dgcnn_fused   edge feature [x_i, x_j - x_i] -> 1x1 convs widths[:-1] over the k neighbours -> max over k ->
//                 conv widths[-1] -> max over points, fused; activations stay in LDS / registers (the reference
//                 round-trips [B,N,20,C] tensors: 872 MB per pair at N=4096).
// The kNN graph is computed once per cloud in the mean-centred frame and shared by the three stages: the later
// frames differ by a rigid motion, which leaves the graph unchanged (SURVEY.md 8.A6 vii; ties at the k-th
// neighbour aside).
#pragma once
#include "ablate.h"
#include "engine.h"
#include "kernels_infer.h"

namespace alignnet {

constexpr int kKnnMaxPerLane = 64;   // N <= 64 * 64 = 4096 candidates per query
constexpr int kKnnList = 128;        // survivors of the first bound that are selected from LDS
constexpr int kKnnWaves = 8;         // waves per workgroup, one query point per wave and round
constexpr int kKnnQueries = 256;     // query points per workgroup (the cloud's candidate table is built once for all of them); fewer when the grid would not fill the chip (launch_knn)

// order-preserving map float -> uint32 (handles the slightly negative "distances" the TF formula can produce) and its inverse
__device__ __forceinline__ uint32_t fkey(float f)
{
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

#ifdef ALIGNNET_KNN_STAMP   // tools/microbench/knn_phases.hip: cycle stamps of one wave at the phase boundaries
__device__ long long g_knn_stamp[8];
#define KNN_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) g_knn_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define KNN_STAMP(i) do {} while (0)
#endif

// grid: (ceil(N / qpw), 2B), block 512 = 8 waves; qpw = query points per workgroup.  PER: candidate slots per lane compiled in (N <= 64 PER).
//
// The workgroup first builds the cloud's candidate table in LDS -- mean-centred x, y, z and |x|^2 per point, 16 bytes each, rows past
// N padded with |x|^2 = +inf -- laid out for packed-fp32 arithmetic: entry (t2, lane) holds the candidates of slots 2 t2 and 2 t2 + 1
// of that lane as {xA, xB, yA, yB} (plane 0) and {zA, zB, ppA, ppB} (plane 1), so that one 16-byte LDS read per plane feeds
// v_pk_mul / v_pk_fma_f32 without register shuffles.  Each wave then takes one query point per round, keeps its 64 PER candidate
// distances in registers as floats, and selects: an upper bound from the per-lane minima, survivors compacted into the wave's
// LDS list, ranks within the list.
//
// Against the first form (candidates read from global memory for every query, integer keys for all of them) this removes, per
// candidate and query: the three global loads and their address arithmetic (L1 was 47 % busy), the centring, |x|^2, the
// float -> key conversion and the `j < N` select -- 19 VALU instructions down to 3 (5 packed operations and a min3 per pair).
// (Two queries per wave halve the LDS reads but need 246 VGPRs, two waves per SIMD: the selection phases -- short dependent
//  scalar / vector chains -- then ran unhidden and the kernel was no faster than the first form: DESIGN.md 4.5.)
template <int PER = kKnnMaxPerLane>
[[maybe_unused]] static __global__ __launch_bounds__(kKnnWaves * 64, 4) void knn_kernel(const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                 const float* __restrict__ center, int B, int N, int k, int* __restrict__ nn, int qpw)
{
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float knn_smem[];
  f32x4* plane0 = reinterpret_cast<f32x4*>(knn_smem);                 // [PER / 2][64] {xA, xB, yA, yB}
  f32x4* plane1 = plane0 + (PER / 2) * 64;                            // [PER / 2][64] {zA, zB, ppA, ppB}
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint2* list = reinterpret_cast<uint2*>(plane1 + (PER / 2) * 64) + wave * kKnnList;   // [8][kKnnList] {distance bits, point index}
  const int cloud = blockIdx.y, tower = cloud >= B, b = cloud - tower * B;
  const float* pc = (tower ? pcs2 : pcs1) + (size_t)b * N * 3;
  const float cx = center[cloud * 3], cy = center[cloud * 3 + 1], cz = center[cloud * 3 + 2];
  KNN_STAMP(0);
  // ---- 0. candidate table ----
  for (int j = tid; j < PER * 64; j += kKnnWaves * 64) {
    const int t = j >> 6, half = t & 1, e = (t >> 1) * 64 + (j & 63);
    float x = 0.f, y = 0.f, z = 0.f, pp = INFINITY;
    if (j < N) {
      x = pc[j * 3] - cx; y = pc[j * 3 + 1] - cy; z = pc[j * 3 + 2] - cz;
      pp = fmaf(z, z, fmaf(y, y, x * x));                             // reduce_sum(square(x)) (tf_util_dgcnn.py:655)
    }
    float* p0 = reinterpret_cast<float*>(plane0 + e);
    float* p1 = reinterpret_cast<float*>(plane1 + e);
    p0[half] = x; p0[2 + half] = y; p1[half] = z; p1[2 + half] = pp;
  }
  __syncthreads();
  KNN_STAMP(1);
  const int per2 = (N + 127) >> 7;     // populated pair slots
  const int qend = min(N, (int)(blockIdx.x + 1) * qpw);
  for (int q = blockIdx.x * qpw + wave; q < qend; q += kKnnWaves) {
    // ---- 1. distances of the wave's query to every candidate; per-lane minimum ----
    KNN_STAMP(7);
    float d[PER];
    const float qx = pc[q * 3] - cx, qy = pc[q * 3 + 1] - cy, qz = pc[q * 3 + 2] - cz;
    const float qq = fmaf(qz, qz, fmaf(qy, qy, qx * qx));
    // -2 * matmul (:653-654): the factor folded into the query, exact.  (The fused multiply-adds are spelled out so that every
    //  evaluation of this formula -- here, the table above, tools/microbench/knn_phases.hip's brute force -- rounds alike.)
    const f32x2 q2x = {-2.0f * qx, -2.0f * qx}, q2y = {-2.0f * qy, -2.0f * qy}, q2z = {-2.0f * qz, -2.0f * qz};
    float lminf = INFINITY;
#pragma unroll
    for (int t0 = 0; t0 < PER / 2; t0 += 4) {
      if (t0 < per2) {
#pragma unroll
        for (int t2 = t0; t2 < t0 + 4; ++t2) {
          const f32x4 a0 = plane0[t2 * 64 + lane], a1 = plane1[t2 * 64 + lane];
          const f32x2 x = {a0[0], a0[1]}, y = {a0[2], a0[3]}, z = {a1[0], a1[1]}, pp = {a1[2], a1[3]};
          const f32x2 inner = __builtin_elementwise_fma(q2z, z, __builtin_elementwise_fma(q2y, y, q2x * x));
          const f32x2 dd = qq + inner + pp;                               // square + inner + square^T (:657)
          d[2 * t2] = dd[0]; d[2 * t2 + 1] = dd[1];
          lminf = fminf(lminf, fminf(dd[0], dd[1]));
        }
        asm volatile("" ::: "memory");   // eight table reads in flight, not all 64 (hoisted together they cost 182 spills at 128 VGPRs)
      } else {
#pragma unroll
        for (int t2 = t0; t2 < t0 + 4; ++t2) { d[2 * t2] = INFINITY; d[2 * t2 + 1] = INFINITY; }
      }
    }
    KNN_STAMP(2);
    // ---- 2. upper bound: the k-th smallest of the 64 per-lane minima (k lanes hold a value <= it).  Bisection over the upper
    //         16 key bits only: the bound may be the top of the k-th minimum's bucket, which lets a few more candidates through
    //         to the list and halves this phase.  Clamped to FLT_MAX: the padding (+inf) never passes. ----
    uint32_t lo = 0u, hi = 0xffffu;
    if (k <= 64) {
      const uint32_t lmin16 = fkey(lminf) >> 16;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (__popcll(__ballot(lmin16 <= mid)) >= k) hi = mid; else lo = mid + 1;
      }
      lo = (lo << 16) | 0xffffu;
    } else lo = 0xffffffffu;
    const uint32_t T0 = min(lo, 0xff7fffffu);   // fkey(FLT_MAX)
    const float T0f = fkey_inv(T0);
    KNN_STAMP(3);
    // ---- 3. compact the survivors (d <= T0), in increasing point index, into this wave's LDS list ----
    int M = 0;
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));   // (the 64 point indices t * 64 + lane are loop invariants: hoisted out of the query loop they held 64 VGPRs)
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      // about k + a few of the N candidates survive, so most 64-candidate slots hold none: skip those on the scalar unit
      const bool sel = d[t] <= T0f;
      const unsigned long long m = __ballot(sel);
      if (__builtin_expect(m == 0ull, 1)) continue;   // the empty slot falls through: v_cmp + an untaken branch
      const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, M));   // M + set bits below this lane
      if (sel && pos < kKnnList) list[pos] = make_uint2(__float_as_uint(d[t]), (unsigned)(t * 64 + lane_q));
      M += __popcll(m);
    }
    int* out = nn + ((size_t)cloud * N + q) * k;
    KNN_STAMP(4);
    if (M <= 64) {
      // ---- 4a. the usual case, one list entry per lane: rank every entry among the others by its (key, index) pair --
      //          one 64-bit compare per entry -- and let the entries of rank < k write themselves (nearest first; ties at
      //          the k-th key go to the lower point indices as tf.nn.top_k) ----
      const bool in = lane < M;
      const uint2 e = list[in ? lane : 0];
      const unsigned long long mine = in ? ((unsigned long long)fkey(__uint_as_float(e.x)) << 32) | e.y : ~0ull;
      int rank = 0;
      for (int i = 0; i < M; i += 4) {   // lanes past M hold ~0: never below a list entry
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const unsigned long long other = __shfl(mine, (i + v) & 63);   // uniform source lane: a v_readlane pair
          rank += other < mine;
        }
      }
      if (in && rank < k) out[rank] = (int)(mine & 0xffffffffu);
      KNN_STAMP(5);
    } else if (M <= kKnnList) {
      // ---- 4b. k-th smallest of the list by bisection (<= 2 entries per lane), emit in list (= index) order ----
      uint32_t lk[kKnnList / 64];
#pragma unroll
      for (int v = 0; v < kKnnList / 64; ++v) lk[v] = v * 64 + lane < M ? fkey(__uint_as_float(list[v * 64 + lane].x)) : 0xffffffffu;
      lo = 0u; hi = T0;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int v = 0; v < kKnnList / 64; ++v) c += __popcll(__ballot(lk[v] <= mid));
        if (c >= k) hi = mid; else lo = mid + 1;
      }
      const uint32_t T = lo;
      int written = 0;
      for (int pass = 0; pass < 2 && written < k; ++pass)
#pragma unroll
        for (int v = 0; v < kKnnList / 64; ++v) {
          const bool sel = (pass == 0 ? lk[v] < T : lk[v] == T) && v * 64 + lane < M;
          const unsigned long long m = __ballot(sel);
          const int pos = written + __popcll(m & ((1ull << lane) - 1ull));
          if (sel && pos < k) out[pos] = (int)list[v * 64 + lane].y;
          written += __popcll(m);
        }
    } else {
      // ---- 4c. (rare: more than kKnnList survivors, i.e. many exact ties) full bisection over all candidates ----
      lo = 0u; hi = T0;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const float midf = fkey_inv(mid);
        int c = 0;
#pragma unroll
        for (int t = 0; t < PER; ++t) c += d[t] <= midf;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (c >= k) hi = mid; else lo = mid + 1;
      }
      const float Tf = fkey_inv(lo);
      int written = 0;
      for (int pass = 0; pass < 2 && written < k; ++pass) {
#pragma unroll
        for (int t = 0; t < PER; ++t) {
          const bool sel = pass == 0 ? d[t] < Tf : d[t] == Tf;
          const unsigned long long m = __ballot(sel);
          const int pos = written + __popcll(m & ((1ull << lane) - 1ull));
          if (sel && pos < k) out[pos] = t * 64 + lane;
          written += __popcll(m);
        }
      }
    }
    KNN_STAMP(6);
  }
}

constexpr size_t knn_lds_bytes(int per) { return (size_t)per * 64 * 16 + (size_t)kKnnWaves * kKnnList * 8; }

// the kNN graph of 2B clouds (N <= 4096, k <= N): nn [2B][N][k].  (static: the kernels are per translation unit, so is the flag)
[[maybe_unused]] static hipError_t launch_knn(int device, hipStream_t stream, const float* p1, const float* p2, const float* center, int B, int N,
                                              int k, int* nn, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr)   // (t0, t1: optional kernel timer, engine.h ProfScope)
{
  // 256 queries per workgroup amortise the candidate table (64 KB at N = 4096: ~3 us to build); a serving-size batch would then be 32 workgroups walking
  // 32 rounds each (157 us at B = 1 .. 4, N = 4096) -- fewer queries per workgroup until the grid has two workgroups per CU
  int qpw = kKnnQueries;
  while (qpw > 32 && (long)2 * B * ((N + qpw - 1) / qpw) < 512) qpw >>= 1;
  const dim3 grid((N + qpw - 1) / qpw, 2 * B), block(kKnnWaves * 64);
  static PerDeviceOnce attr_done;
  if (attr_done.need(device)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds_bytes(16));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds_bytes(32));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds_bytes(64));
    if (e != hipSuccess) return e;
    attr_done.mark(device);
  }
  if (N <= 1024) hipExtLaunchKernelGGL(knn_kernel<16>, grid, block, knn_lds_bytes(16), stream, t0, t1, 0, p1, p2, center, B, N, k, nn, qpw);
  else if (N <= 2048) hipExtLaunchKernelGGL(knn_kernel<32>, grid, block, knn_lds_bytes(32), stream, t0, t1, 0, p1, p2, center, B, N, k, nn, qpw);
  else hipExtLaunchKernelGGL(knn_kernel<64>, grid, block, knn_lds_bytes(64), stream, t0, t1, 0, p1, p2, center, B, N, k, nn, qpw);
  return hipGetLastError();
}

struct DgcnnArgs {
  const float* pcs[2]; const float* xform; const int* nn;   // nn: [2B][N][k]
  float* pooled; long tower_stride, row_stride;
  int B, N, k, nlayers;
  int ld[2];
  ConvLayerDev L[kMaxConv];   // L[0].w = raw [6][C1]; others packed images
  long long* stamps;          // debug (ALIGNNET_DBG & 64): cycle stamps of thread 0 / workgroup (1, 0) around neighbour slot 5
};
#define DG_STAMP(i) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 1 && blockIdx.y == 0 && tid == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

#ifndef DGLB
#define DGLB 6
#endif
#ifndef DG_LIFTPRIO
#define DG_LIFTPRIO 1   // measured (tools/ab_dgcnn_variants.sh): 0.777 -> 0.789 of the fp32-MFMA roofline at N = 4096; 3 gives the same
#endif
constexpr int kDgTile = 64;   // points per workgroup (two 32-row MFMA tiles)

// LDS: es [64][8] | buf0 [64][ld0] | buf1 [64][ld1] | buf0' [64][ld0] (second lift buffer)
// Per neighbour slot: the gather of x_j for slot s+1 is issued before the MFMAs of slot s (its global round trip hides
// behind them), and the lift output is double-buffered so that lifting slot s+1 does not wait for slot s's readers.
__device__ __forceinline__ void dg_gather(const DgcnnArgs& a, const float* pc, int cloud, int tile, int slot, int tid, float (&v)[6])
{
  const int n = min(tile * kDgTile + tid, a.N - 1);
  const int j = a.nn[((size_t)cloud * a.N + n) * a.k + slot];
  const float* p = pc + (size_t)n * 3;
  const float* pj = pc + (size_t)j * 3;
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  v[3] = pj[0] - p[0]; v[4] = pj[1] - p[1]; v[5] = pj[2] - p[2];
}

// software-pipelined form of the gather: the neighbour INDEX of slot s + 2 and the neighbour POINT of slot s + 1 are requested
// at the top of slot s, so no load of a slot depends on another load of the same slot (index -> point was two exposed global
// round trips in front of the edge features of every slot); the thread's own point is loaded once per tile.
__device__ __forceinline__ int dg_nn_index(const DgcnnArgs& a, int cloud, int tile, int slot, int tid)
{
  const int n = min(tile * kDgTile + tid, a.N - 1);
  return a.nn[((size_t)cloud * a.N + n) * a.k + min(slot, a.k - 1)];
}

__device__ __forceinline__ void dg_edge_to_lds(const float* xf, const float (&v)[6], float* e)
{
  const float x = v[0] - xf[0], y = v[1] - xf[1], z = v[2] - xf[2];
  e[0] = x * xf[3] + y * xf[6] + z * xf[9];
  e[1] = x * xf[4] + y * xf[7] + z * xf[10];
  e[2] = x * xf[5] + y * xf[8] + z * xf[11];
  e[3] = v[3] * xf[3] + v[4] * xf[6] + v[5] * xf[9];
  e[4] = v[3] * xf[4] + v[4] * xf[7] + v[5] * xf[10];
  e[5] = v[3] * xf[5] + v[4] * xf[8] + v[5] * xf[11];
}

// edge layer 0: K = 6 lift on the VALU, es -> out[64][ldo].  The thread's weights / scale / shift (channels c0 and
// c0 + 32 for widths <= 64) are passed in registers: reloading them from global memory in every neighbour slot put an L2
// round trip on each slot's critical path.
struct DgLiftRegs { float w[2][6]; float sc[2], sh[2]; };

__device__ __forceinline__ DgLiftRegs dg_lift_load(const ConvLayerDev& L, int tower, int tid)
{
  DgLiftRegs R;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = (tid & 31) + 32 * g;
    const bool live = c < L.cout;
#pragma unroll
    for (int d = 0; d < 6; ++d) R.w[g][d] = live ? L.w[d * L.cout + c] : 0.f;
    R.sc[g] = live ? L.scale[tower * L.cout + c] : 0.f;
    R.sh[g] = live ? L.shift[tower * L.cout + c] : 0.f;
  }
  return R;
}

__device__ __forceinline__ void dg_lift(const ConvLayerDev& L, int tower, const float* es, float* out, int ldo, int tid,
                                        const DgLiftRegs* regs = nullptr)
{
  const int c0 = tid & 31, r0 = tid >> 5;
  const int cw = (L.cout + 7) & ~7;
  for (int c = c0; c < cw; c += 32) {
    const bool live = c < L.cout;
    const int g = (c - c0) >> 5;
    float w[6];
    float sc, sh;
    if (regs && g < 2) {
#pragma unroll
      for (int d = 0; d < 6; ++d) w[d] = g ? regs->w[1][d] : regs->w[0][d];
      sc = g ? regs->sc[1] : regs->sc[0]; sh = g ? regs->sh[1] : regs->sh[0];
    } else {
#pragma unroll
      for (int d = 0; d < 6; ++d) w[d] = live ? L.w[d * L.cout + c] : 0.f;
      sc = live ? L.scale[tower * L.cout + c] : 0.f; sh = live ? L.shift[tower * L.cout + c] : 0.f;
    }
#pragma unroll
    for (int rr = 0; rr < kDgTile / 16; ++rr) {
      const int row = rr * 16 + r0;
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(es + row * 8);
      const f32x4 e1 = *reinterpret_cast<const f32x4*>(es + row * 8 + 4);
      float acc = e0[0] * w[0];
      acc = fmaf(e0[1], w[1], acc); acc = fmaf(e0[2], w[2], acc); acc = fmaf(e0[3], w[3], acc);
      acc = fmaf(e1[0], w[4], acc); acc = fmaf(e1[1], w[5], acc);
      out[row * ldo + c] = fmaxf(fmaf(acc, sc, sh), 0.f);
    }
  }
}

// LD0 / LD1 != 0: the shipped shape compiled in -- widths [LD0 - 4, LD1 - 4, C3], three layers -- so that LDS strides, k-depths
// and tile counts are constants (as for pointnet_fused: fewer address registers, no generic layer dispatch)
template <int LD0 = 0, int LD1 = 0>
[[maybe_unused]] static __global__ __launch_bounds__(kWaves * 64, LD0 ? DGLB : 2) void dgcnn_fused(const DgcnnArgs a)
{
  const int ld0 = LD0 ? LD0 : a.ld[0], ld1 = LD1 ? LD1 : a.ld[1];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.y, tile = blockIdx.x;
  const int tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  float* es = smem;                                   // edge features of the slot being lifted
  // generic: es | buf0 | buf1 | buf0'.
  // (shipped shape: edge-feature rows 9 floats apart -- the MFMA lift reads them 4 bytes per lane, 16 rows per k, and at 8 floats
  //  rows r and r + 8 share a bank: that was the kernel's 26 M LDS conflict cycles per launch)
  constexpr int kEsLd = LD0 ? 9 : 8, kEs = kDgTile * kEsLd;
  // Shipped shape: es | W2 image [4][8][64][4] | buf0 -- ONE lift buffer (the lift of slot s + 1 runs behind the barrier that ends slot
  // s's MFMAs, so it can overwrite what they read), and the pooled edge features (buf1, written after the neighbour loop) alias the
  // W2 image and the head of buf0, both dead by then: 52.5 KiB, THREE workgroups per CU (6 waves per SIMD at <= 80 VGPRs).
  constexpr int kW2Off = kEs;                                       // shipped shape only
  constexpr int kW2 = 4 * 8 * 64 * 4;
  const int boff[2] = {LD0 ? kEs + kW2 : kEs, LD0 ? kEs : kEs + kDgTile * ld0};
  const int boff0b = LD0 ? kEs + kW2 : kEs + kDgTile * (ld0 + ld1);   // second lift buffer (generic shape only)
  static_assert(!LD0 || kDgTile * LD1 <= kW2 + kDgTile * LD0, "the pooled edge features must fit the regions they alias");
  const int nl = LD0 ? 3 : a.nlayers;                           // edge convs: layers 0 .. nl-2 ; point conv: layer nl-1
  const int nvalid = min(kDgTile, a.N - tile * kDgTile);
  const bool two_edge_layers = nl == 3;               // pipelined path: lift (VALU) + one MFMA edge layer

  // running max over the k neighbours of the LAST edge layer's pre-activation (sc*acc+sh); relu folded after the max
  const ConvLayerDev& LE = a.L[nl - 2];
  const int le_cin = LD0 ? LD0 - 4 : LE.cin, le_cout = LD1 ? LD1 - 4 : LE.cout;
  const int CTE = (le_cout + 31) >> 5;
  // item = (32-row tile m of the 64 points, channel tile ct): wave handles items wave, wave+8, ... (<= 2 slots: C <= 256)
  constexpr int kSlots = 2;
  f32x16 best[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) best[s][r] = -INFINITY;

  // Register-resident weights for the common case (last edge layer with cin <= 64, e.g. 64 -> 128): the item's
  // 8 k-groups x float4 are loaded once per workgroup instead of once per neighbour slot.
  const bool wreg = two_edge_layers && le_cin <= 64;
  f32x4 breg[kSlots][8];
  if (LD0) {
    // shipped shape: the 64 -> 128 layer's weight image (4 channel tiles x 8 k-groups x 1 KiB) is staged in LDS once per workgroup
    // and every wave reads its item's fragments from there in each neighbour slot.  Keeping them in 32 registers per lane (the
    // generic path below) does not fit the 128-VGPR budget of two workgroups per CU: it cost 12 spill slots per lane.
    const f32x4* src = reinterpret_cast<const f32x4*>(LE.w);
    f32x4* dst = reinterpret_cast<f32x4*>(smem + kW2Off);
#pragma unroll
    for (int i = 0; i < (4 * 8 * 64) / (kWaves * 64); ++i) dst[i * kWaves * 64 + tid] = src[i * kWaves * 64 + tid];
  } else if (wreg) {
    const int KG = (le_cin + 7) >> 3;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int item = wave + s * kWaves;
      const int ct = min(item >> 1, CTE - 1);
#pragma unroll
      for (int kg = 0; kg < 8; ++kg)
        breg[s][kg] = reinterpret_cast<const f32x4*>(LE.w)[((size_t)ct * KG + min(kg, KG - 1)) * 64 + lane];
    }
  }
  // scale / shift of the wave's (<= 2) items stay in registers across the neighbour slots (reloaded per slot they were an L2
  // round trip between every slot's MFMAs and its running max)
  float esc[kSlots], esh[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const int col = (min(wave + s * kWaves, CTE * 2 - 1) >> 1) * 32 + (lane & 31);
    const bool live = col < le_cout;
    esc[s] = live ? LE.scale[tower * le_cout + col] : 0.f;
    esh[s] = live ? LE.shift[tower * le_cout + col] : 0.f;
  }
  auto edge_mfma_reg = [&](const float* in, int ldi) {
    const int KG = (le_cin + 7) >> 3;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int item = wave + s * kWaves;
      if (item < CTE * 2) {
        const int m = item & 1;
        const float* arow = in + (m * 32 + (lane & 31)) * ldi + (lane >> 5) * 4;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (LD0) {
          if (s == 0) {   // 8 items = 8 waves: the second slot does not exist for C2 = 128
            const f32x4* bl = reinterpret_cast<const f32x4*>(smem + kW2Off) + (item >> 1) * 8 * 64 + lane;
            f32x4 av[8], bv[8];
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) { av[kg] = *reinterpret_cast<const f32x4*>(arow + kg * 8); bv[kg] = bl[kg * 64]; }
#pragma unroll
            for (int kg = 0; kg < 8; ++kg)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kg][q], bv[kg][q], acc, 0, 0, 0);
          }
        } else {
        f32x4 av[8];
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) av[kg] = *reinterpret_cast<const f32x4*>(arow + min(kg, KG - 1) * 8);
#pragma unroll
        for (int kg = 0; kg < 8; ++kg)
          if (kg < KG) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kg][q], breg[s][kg][q], acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) best[s][r] = fmaxf(best[s][r], fmaf(acc[r], esc[s], esh[s]));
      }
    }
  };
  auto edge_mfma = [&](const float* in, int ldi) {
    const int KG = (le_cin + 7) >> 3;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int item = wave + s * kWaves;
      if (item < CTE * 2) {
        const int ct = item >> 1, m = item & 1;
        f32x16 acc[1];
        mfma_rows<1>(in + m * 32 * ldi, ldi, reinterpret_cast<const f32x4*>(LE.w) + (size_t)ct * KG * 64, KG, lane, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) best[s][r] = fmaxf(best[s][r], fmaf(acc[0][r], esc[s], esh[s]));
      }
    }
  };

  if (two_edge_layers) {
    // the gather of x_j for slot s+1 is issued before the MFMAs of slot s; the lift output is double-buffered.
    // (A one-barrier variant that issued the VALU lift behind the wave's own MFMAs measured 13 % slower: a wave
    //  issues in order, so only OTHER waves' VALU work overlaps its MFMAs.)
    float v[6];
    float own[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f};
    int jn = 0;   // neighbour index of the slot after next
    // the cloud's rotation, read once: inside the slot loop (divergent branch, behind a memory clobber) hipcc re-read the nine
    // floats with vector loads and waited for them -- an exposed L2 round trip in every slot's edge-feature phase (2.3 k cycles)
    float rot[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) rot[i] = xf[3 + i];
    if (tid < kDgTile) {
      dg_gather(a, pc, cloud, tile, 0, tid, v);
      own[0] = v[0]; own[1] = v[1]; own[2] = v[2];
      jn = dg_nn_index(a, cloud, tile, 1, tid);
      dg_edge_to_lds(xf, v, es + tid * kEsLd);
    }
    // shipped shape: the K = 6 lift runs on the matrix pipe as 16 x 16 x 4 tiles (k padded to 8 with zero weights and
    // zero edge-feature columns): the [64 rows][64 channels] output is 16 tiles, two per wave, two instructions each; the wave's
    // B fragments (2 tiles x 2 k-steps) and the tiles' scale / shift stay in 8 registers.  Per lane and slot: 4 LDS reads, 4 MFMAs,
    // 8 fma+max, 8 LDS writes -- against 16 16-byte LDS reads and 64 VALU operations for the VALU form (whose weights, at the 128-VGPR
    // budget of two workgroups per CU, had to live in LDS or spill: profiles/r01_dgcnn_pmc_by_kernel.json, 1.3 GB of scratch writes).
    float lw[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, lsc[2] = {0.f, 0.f}, lsh[2] = {0.f, 0.f};
    if (LD0) {
      const ConvLayerDev& L0 = a.L[0];
      if (tid < kDgTile) { es[tid * kEsLd + 6] = 0.f; es[tid * kEsLd + 7] = 0.f; }   // never written again (dg_edge_to_lds fills 0..5)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = 16 * ((wave * 2 + i) & 3) + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int kk = 4 * ks + (lane >> 4);
          lw[i][ks] = kk < 6 ? L0.w[kk * (LD0 - 4) + c] : 0.f;
        }
        lsc[i] = L0.scale[tower * (LD0 - 4) + c];
        lsh[i] = L0.shift[tower * (LD0 - 4) + c];
      }
    }
    auto lift_mfma = [&](float* out) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = wave * 2 + i, rt = t >> 2, c = 16 * (t & 3) + (lane & 15);
        const float* ar = es + (16 * rt + (lane & 15)) * kEsLd + (lane >> 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], lw[i][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4], lw[i][1], acc, 0, 0, 0);
        float* o = out + (16 * rt + 4 * (lane >> 4)) * ld0 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * ld0] = fmaxf(fmaf(acc[r], lsc[i], lsh[i]), 0.f);
      }
    };
    // (A one-barrier form -- edge features double-buffered too, each wave's lift MFMAs issued in front of its own edge MFMAs --
    //  measured 0.776 of the roofline against 0.815 for the two-barrier loop below: DESIGN.md 4.5.)
    __syncthreads();
    DgLiftRegs lregs;
    if (!LD0) lregs = dg_lift_load(a.L[0], tower, tid);
    if (LD0) lift_mfma(smem + boff[0]);
    else dg_lift(a.L[0], tower, es, smem + boff[0], ld0, tid, &lregs);
    __syncthreads();
    for (int slot = 0; slot < a.k; ++slot) {
      const bool more = slot + 1 < a.k;
      if (slot == 5) DG_STAMP(0);
      if (more && tid < kDgTile) {                    // in flight during the MFMAs: x_j of slot + 1 (index already here), index of slot + 2
        const float* pj = pc + (size_t)jn * 3;
        nb[0] = pj[0]; nb[1] = pj[1]; nb[2] = pj[2];
        jn = dg_nn_index(a, cloud, tile, slot + 2, tid);
        asm volatile("" ::: "memory");                // keep the four requests in front of the MFMAs (hipcc sinks loads to their use)
      }
      if (wreg) edge_mfma_reg(smem + ((slot & 1) ? boff0b : boff[0]), ld0);
      else edge_mfma(smem + ((slot & 1) ? boff0b : boff[0]), ld0);
      if (slot == 5) DG_STAMP(1);
#if DG_LIFTPRIO
      // the VALU / LDS phases of a slot run at raised priority: the co-resident workgroup's MFMA stream needs one issue slot per 64
      // cycles and loses nothing, while a lift issued at equal priority next to it took 3x its stand-alone time
      __builtin_amdgcn_s_setprio(DG_LIFTPRIO);
#endif
      if (more && tid < kDgTile) {
        if (LD0) {   // the rotated own point e[0..2] does not change over the slots: only (x_j - x_i) R is rewritten
          const float v3 = nb[0] - own[0], v4 = nb[1] - own[1], v5 = nb[2] - own[2];
          float* e = es + tid * kEsLd;
          e[3] = v3 * rot[0] + v4 * rot[3] + v5 * rot[6];
          e[4] = v3 * rot[1] + v4 * rot[4] + v5 * rot[7];
          e[5] = v3 * rot[2] + v4 * rot[5] + v5 * rot[8];
        } else {
          v[0] = own[0]; v[1] = own[1]; v[2] = own[2]; v[3] = nb[0] - own[0]; v[4] = nb[1] - own[1]; v[5] = nb[2] - own[2];
          dg_edge_to_lds(xf, v, es + tid * kEsLd);
        }
      }
      if (slot == 5) DG_STAMP(2);
      __syncthreads();
      if (slot == 5) DG_STAMP(3);
      if (more) {
        if (LD0) lift_mfma(smem + (((slot + 1) & 1) ? boff0b : boff[0]));
        else dg_lift(a.L[0], tower, es, smem + (((slot + 1) & 1) ? boff0b : boff[0]), ld0, tid, &lregs);
      }
      if (slot == 5) DG_STAMP(4);
#if DG_LIFTPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      __syncthreads();
      if (slot == 5) DG_STAMP(5);
    }
    DG_STAMP(6);
  } else {
    for (int slot = 0; slot < a.k; ++slot) {
      __syncthreads();
      if (tid < kDgTile) { float v[6]; dg_gather(a, pc, cloud, tile, slot, tid, v); dg_edge_to_lds(xf, v, es + tid * 8); }
      __syncthreads();
      dg_lift(a.L[0], tower, es, smem + boff[0], ld0, tid);
      __syncthreads();
      // ---- middle edge layers 1 .. nl-3 ----
      for (int l = 1; l < nl - 2; ++l) {
        const ConvLayerDev& L = a.L[l];
        const float* in = smem + (((l - 1) & 1) ? boff[1] : boff[0]);
        float* out = smem + ((l & 1) ? boff[1] : boff[0]);
        hidden_layer<2, kDgTile>(in, (((l - 1) & 1) ? ld1 : ld0), out, ((l & 1) ? ld1 : ld0), L, tower, wave, lane);
        __syncthreads();
      }
      const int l = nl - 2;
      edge_mfma(smem + (((l - 1) & 1) ? boff[1] : boff[0]), (((l - 1) & 1) ? ld1 : ld0));
    }
  }
  __syncthreads();
  // ---- relu(max_k) -> LDS point features [64][C_E] (tp8.py:42) ----
  const int lh = (nl - 2) & 1;   // buffer that receives the pooled edge features
  {
    float* out = smem + (lh ? boff[1] : boff[0]);
    const int ldo = (lh ? ld1 : ld0);
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int item = wave + s * kWaves;
      if (item < CTE * 2) {
        const int ct = item >> 1, m = item & 1;
        const int col = ct * 32 + (lane & 31);
        if (col < ((le_cout + 7) & ~7)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[row * ldo + col] = (col < le_cout && row < nvalid) ? fmaxf(best[s][r], 0.f) : 0.f;
          }
        }
      }
    }
  }
  __syncthreads();
  DG_STAMP(7);
  // ---- point conv (widths[-1]) + max over the tile's points (tp8.py:43-45) ----
  {
    const ConvLayerDev& L = a.L[nl - 1];
    const float* in = smem + (lh ? boff[1] : boff[0]);
    const int ldi = (lh ? ld1 : ld0);
    const int KG = LD1 ? (LD1 - 4) / 8 : (L.cin + 7) >> 3, CT = (L.cout + 31) >> 5;
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
    for (int ct = wave; ct < CT; ct += kWaves) {
      f32x16 acc[2];
#ifndef DG_PC
#define DG_PC 0
#endif
      // shipped shape: four waves per SIMD -- variant chosen by measurement (DG_PC: 0 compiler-managed loads, 1 hand-issued stream,
      // 2 compiler-managed three k-groups deep); the generic instantiation (two waves per SIMD) keeps the hand-issued stream
      if (LD0 && DG_PC == 0) mfma_rows<2, true, false>(in, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc);
      else if (LD0 && DG_PC == 2) mfma_rows_deep<2>(in, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc);
      else mfma_rows<2>(in, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc);
      const int col = ct * 32 + (lane & 31);
      const bool live = col < L.cout;
      const float sc = live ? L.scale[tower * L.cout + col] : 0.f, sh = live ? L.shift[tower * L.cout + col] : 0.f;
      float mx = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < nvalid) mx = fmaxf(mx, fmaf(acc[m][r], sc, sh));
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // (values only grow and start at 0: a tile that does not beat what is already there -- most of a cloud's 64 tiles -- skips the
      //  atomic; a stale read can only cause a redundant one.  The per-tile atomics were 156 MB of write traffic per launch.)
      if (lane < 32 && live && mx > dst[col]) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));
    }
  }
  DG_STAMP(8);
}

}  // namespace alignnet
