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
#include "kernels_infer.h"

namespace alignnet {

constexpr int kKnnMaxPerLane = 64;   // N <= 64 * 64 = 4096 candidates per query
constexpr int kKnnList = 256;        // survivors of the first bound that are selected from LDS

// order-preserving map float -> uint32 (handles the slightly negative "distances" the TF formula can produce)
__device__ __forceinline__ uint32_t fkey(float f)
{
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

#ifdef ALIGNNET_KNN_STAMP   // tools/microbench/knn_phases.hip: cycle stamps of one wave at the phase boundaries
__device__ long long g_knn_stamp[8];
#define KNN_STAMP(i) do { if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) g_knn_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define KNN_STAMP(i) do {} while (0)
#endif
// grid: (ceil(N / 4), 2B), block 256 = 4 waves, one query point per wave
// PER: candidate slots per lane compiled in (N <= 64 PER): 64 for the full 4096-point range, 16 for N <= 1024 (a quarter of the
// key registers, twice the waves per SIMD)
template <int PER = kKnnMaxPerLane>
[[maybe_unused]] static __global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                 const float* __restrict__ center, int B, int N, int k, int* __restrict__ nn)
{
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cloud = blockIdx.y, tower = cloud >= B, b = cloud - tower * B;
  const int q = blockIdx.x * 4 + wave;
  if (q >= N) return;
  const float* pc = (tower ? pcs2 : pcs1) + (size_t)b * N * 3;
  const float cx = center[cloud * 3], cy = center[cloud * 3 + 1], cz = center[cloud * 3 + 2];
  const float qx = pc[q * 3] - cx, qy = pc[q * 3 + 1] - cy, qz = pc[q * 3 + 2] - cz;
  const float qq = qx * qx + qy * qy + qz * qz;   // reduce_sum(square(x)) (tf_util_dgcnn.py:655)
  KNN_STAMP(0);
  uint32_t key[PER];
  const int per = (N + 63) >> 6;
  // Candidates are loaded eight slots at a time with clamped (always valid) indices and no branch around the loads: guarded
  // by `if (j < N)` every slot's three loads were issued and waited for inside their own exec-masked block -- 64 exposed
  // memory round trips per query (50 k of the 75 k cycles of a query at N = 4096).
#pragma unroll
  for (int t0 = 0; t0 < PER; t0 += 8) {
    if (t0 < per) {
      float px[8], py[8], pz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min((t0 + u) * 64 + lane, N - 1);
        px[u] = pc[j * 3]; py[u] = pc[j * 3 + 1]; pz[u] = pc[j * 3 + 2];
      }
      // two candidates per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32): same operations per component
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        const f32x2 x = f32x2{px[u], px[u + 1]} - cx, y = f32x2{py[u], py[u + 1]} - cy, z = f32x2{pz[u], pz[u + 1]} - cz;
        const f32x2 inner = -2.0f * (qx * x + qy * y + qz * z);          // -2 * matmul (:653-654)
        const f32x2 d = qq + inner + (x * x + y * y + z * z);             // square + inner + square^T (:657)
        key[t0 + u] = (t0 + u) * 64 + lane < N ? fkey(d[0]) : 0xffffffffu;
        key[t0 + u + 1] = (t0 + u + 1) * 64 + lane < N ? fkey(d[1]) : 0xffffffffu;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) key[t0 + u] = 0xffffffffu;
    }
  }
  KNN_STAMP(1);
  // ---- 1. upper bound: the k-th smallest of the 64 per-lane minima (k lanes hold a value <= it) ----
  uint32_t lmin = 0xffffffffu;
  // (the slot loops below test `per` once per block of eight slots: slots past N hold the sentinel key, which no bound admits;
  //  a uniform `t < per` per slot made hipcc keep 64 such predicates in SGPRs across the loops -- 587 scalar spills)
#pragma unroll
  for (int t0 = 0; t0 < PER; t0 += 8)
    if (t0 < per) {
#pragma unroll
      for (int t = t0; t < t0 + 8; ++t) lmin = min(lmin, key[t]);
    }
  // (bisection over the upper 16 key bits only: the bound may be the top of the k-th minimum's 2^-7-wide bucket, which lets a few
  //  more candidates through to the list and halves this phase)
  uint32_t lo = 0u, hi = 0xffffu;
  if (k <= 64) {
    const uint32_t lmin16 = lmin >> 16;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (__popcll(__ballot(lmin16 <= mid)) >= k) hi = mid; else lo = mid + 1;
    }
    lo = (lo << 16) | 0xffffu;
    if (lo == 0xffffffffu) lo = 0xfffffffeu;   // never admit the sentinel keys of slots past N
  } else lo = 0xfffffffeu;   // every real candidate; the sentinel keys (0xffffffff) of slots past N stay out without an index test
  const uint32_t T0 = lo;
  KNN_STAMP(2);
  // ---- 2. compact the survivors (key <= T0), in increasing point index, into this wave's LDS list ----
  __shared__ uint32_t s_key[4][kKnnList];
  __shared__ int s_idx[4][kKnnList];
  int M = 0;
#pragma unroll
  for (int t0 = 0; t0 < PER; t0 += 8)
    if (t0 < per) {
#pragma unroll
      for (int t = t0; t < t0 + 8; ++t) {
        // about k + a few of the N candidates survive, so most 64-candidate slots hold none: skip those on the scalar unit
        const bool sel = key[t] <= T0;   // T0 <= 0xfffffffe: no sentinel passes, no `t * 64 + lane < N` needed
        const unsigned long long m = __ballot(sel);
        if (m == 0ull) continue;
        const int pos = M + __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < kKnnList) { s_key[wave][pos] = key[t]; s_idx[wave][pos] = t * 64 + lane; }
        M += __popcll(m);
      }
    }
  int* out = nn + ((size_t)cloud * N + q) * k;
  KNN_STAMP(3);
  if (M <= 64) {
    // ---- 3a'. the usual case, one list entry per lane: rank every entry among the others by its (key, index) pair --
    //           one 64-bit compare per entry -- and let the entries of rank < k write themselves (nearest first; ties at
    //           the k-th key go to the lower point indices as tf.nn.top_k).  A 32-step bisection was 5.2 k cycles. ----
    const bool in = lane < M;
    const unsigned long long mine = in ? ((unsigned long long)s_key[wave][lane] << 32) | (unsigned)s_idx[wave][lane] : ~0ull;
    int rank = 0;
    for (int i = 0; i < M; ++i) {
      const unsigned long long other = __shfl(mine, i);   // uniform source lane: a v_readlane pair
      rank += other < mine;
    }
    if (in && rank < k) out[rank] = (int)(mine & 0xffffffffu);
    KNN_STAMP(4);
    KNN_STAMP(5);
    return;
  }
  if (M <= kKnnList) {
    // ---- 3a. k-th smallest of the list by bisection (<= 4 entries per lane), emit in list (= index) order ----
    uint32_t lk[kKnnList / 64];
#pragma unroll
    for (int u = 0; u < kKnnList / 64; ++u) lk[u] = u * 64 + lane < M ? s_key[wave][u * 64 + lane] : 0xffffffffu;
    lo = 0u; hi = T0;
    const int nU = (M + 63) >> 6;   // populated registers of the list (usually one)
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int u = 0; u < kKnnList / 64; ++u)
        if (u < nU) c += __popcll(__ballot(lk[u] <= mid));
      if (c >= k) hi = mid; else lo = mid + 1;
    }
    const uint32_t T = lo;
    KNN_STAMP(4);
    int written = 0;
    for (int pass = 0; pass < 2 && written < k; ++pass)
#pragma unroll
      for (int u = 0; u < kKnnList / 64; ++u) {
        const bool sel = (pass == 0 ? lk[u] < T : lk[u] == T) && u * 64 + lane < M;
        const unsigned long long m = __ballot(sel);
        const int pos = written + __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < k) out[pos] = s_idx[wave][u * 64 + lane];
        written += __popcll(m);
      }
    KNN_STAMP(5);
    return;
  }
  // ---- 3b. (rare: more than kKnnList survivors) full bisection over all candidates ----
  lo = 0u; hi = T0;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    int c = 0;
#pragma unroll
    for (int t0 = 0; t0 < PER; t0 += 8)
      if (t0 < per) {
#pragma unroll
        for (int t = t0; t < t0 + 8; ++t) c += key[t] <= mid;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (c >= k) hi = mid; else lo = mid + 1;
  }
  const uint32_t T = lo;
  int written = 0;
  for (int pass = 0; pass < 2 && written < k; ++pass) {
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if ((t & ~7) < per) {
        const bool sel = pass == 0 ? key[t] < T : key[t] == T;   // T <= T0 <= 0xfffffffe
        const unsigned long long m = __ballot(sel);
        const int pos = written + __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < k) out[pos] = t * 64 + lane;
        written += __popcll(m);
      }
  }
}

struct DgcnnArgs {
  const float* pcs[2]; const float* xform; const int* nn;   // nn: [2B][N][k]
  float* pooled; long tower_stride, row_stride;
  int B, N, k, nlayers;
  int ld[2];
  ConvLayerDev L[kMaxConv];   // L[0].w = raw [6][C1]; others packed images
  long long* stamps;          // debug (ALIGNNET_DBG & 64): cycle stamps of thread 0 / workgroup (1, 0) around neighbour slot 5
};
#define DG_STAMP(i) do { if (a.stamps && blockIdx.x == 1 && blockIdx.y == 0 && tid == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

#ifndef DGLB
#define DGLB 4
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
                                        const DgLiftRegs* regs = nullptr, const float* wl = nullptr)
{
  const int c0 = tid & 31, r0 = tid >> 5;
  const int cw = (L.cout + 7) & ~7;
  for (int c = c0; c < cw; c += 32) {
    const bool live = c < L.cout;
    const int g = (c - c0) >> 5;
    float w[6];
    float sc, sh;
    if (wl) {   // the layer's [w0..w5, scale, shift] rows staged in LDS by the caller (two 16-byte reads per channel)
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(wl + c * 8), q1 = *reinterpret_cast<const f32x4*>(wl + c * 8 + 4);
      w[0] = q0[0]; w[1] = q0[1]; w[2] = q0[2]; w[3] = q0[3]; w[4] = q1[0]; w[5] = q1[1]; sc = q1[2]; sh = q1[3];
    } else if (regs && g < 2) {
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
  // generic: es | buf0 | buf1 | buf0'.  Shipped shape (LD0): es | buf0 | buf0' | lift table [64][8] | W2 image [4][8][64][4], and
  // buf1 (the pooled edge features, written after the neighbour loop when both lift buffers are dead) aliases buf0 / buf0'.
  const int boff[2] = {kDgTile * 8, LD0 ? kDgTile * 8 : kDgTile * 8 + kDgTile * ld0};
  const int boff0b = LD0 ? kDgTile * 8 + kDgTile * ld0 : kDgTile * 8 + kDgTile * (ld0 + ld1);   // second lift buffer
  constexpr int kWlOff = kDgTile * 8 + kDgTile * 2 * (LD0 ? LD0 : 1);      // shipped shape only
  constexpr int kW2Off = kWlOff + 64 * 8;
  static_assert(!LD0 || kDgTile * LD1 <= 2 * kDgTile * LD0, "the pooled edge features must fit the two lift buffers they alias");
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
    if (tid < kDgTile) {
      dg_gather(a, pc, cloud, tile, 0, tid, v);
      own[0] = v[0]; own[1] = v[1]; own[2] = v[2];
      jn = dg_nn_index(a, cloud, tile, 1, tid);
      dg_edge_to_lds(xf, v, es + tid * 8);
    }
    // shipped shape (128-VGPR budget): the lift's weights / scale / shift live in LDS ([C1][8] floats behind the activation
    // buffers) instead of 16 registers per thread -- with them the kernel needed 12 spill slots per lane, i.e. 44 B x 512 threads
    // of scratch written once by each of the 65 k workgroups of a launch (1.3 GB of HBM writes, profiles/r01_dgcnn_pmc_by_kernel.json)
    float* wl = LD0 ? smem + kWlOff : nullptr;
    if (LD0 && tid < (LD0 - 4)) {
      const ConvLayerDev& L0 = a.L[0];
#pragma unroll
      for (int d = 0; d < 6; ++d) wl[tid * 8 + d] = L0.w[d * L0.cout + tid];
      wl[tid * 8 + 6] = L0.scale[tower * L0.cout + tid];
      wl[tid * 8 + 7] = L0.shift[tower * L0.cout + tid];
    }
    __syncthreads();
    DgLiftRegs lregs;
    if (!LD0) lregs = dg_lift_load(a.L[0], tower, tid);
    dg_lift(a.L[0], tower, es, smem + boff[0], ld0, tid, LD0 ? nullptr : &lregs, wl);
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
        v[0] = own[0]; v[1] = own[1]; v[2] = own[2]; v[3] = nb[0] - own[0]; v[4] = nb[1] - own[1]; v[5] = nb[2] - own[2];
        dg_edge_to_lds(xf, v, es + tid * 8);
      }
      if (slot == 5) DG_STAMP(2);
      __syncthreads();
      if (slot == 5) DG_STAMP(3);
      if (more) dg_lift(a.L[0], tower, es, smem + (((slot + 1) & 1) ? boff0b : boff[0]), ld0, tid, LD0 ? nullptr : &lregs, wl);
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
      if (lane < 32 && live) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));
    }
  }
  DG_STAMP(8);
}

}  // namespace alignnet
