// Phase 3 of the training forward (kernels_train_fwd.h) on 128-point tiles for the shipped widths C1 = 64, C2 = 128
// (models/tp8.py:49-59 with is_training = True; every configs/*.json of the reference but default.json), gfx950 only.
//
// Why a second tile shape: with 64-point tiles (train_fwd_phase23<3, ...>) a weight fragment of the 128 -> C3 lift feeds two
// 32-row MFMA tiles, with 128-point tiles four -- half the L2 -> CU weight stream per FLOP.  The inference kernel measured exactly
// this trade (0.76 of the fp32-MFMA roofline with two row tiles per fragment, 0.85 with four), and the 64-point
// training kernel sat at 0.725.  One workgroup of eight waves per cloud (two per SIMD, 104 KiB of LDS: one workgroup per CU);
// same outputs, same layouts as train_fwd_phase23<3> (ext / idx per lane half, column sums of h2 in four slices, h2 stored
// row-major), so everything downstream is unchanged.
#pragma once
#include "kernels_train_fwd.h"

namespace alignnet {

constexpr int kWT = 128;   // points per tile
constexpr int kWW = 8;     // waves per workgroup
constexpr int kWSlots = 4; // channel tiles of the lift per wave: C3 <= 1024

static inline size_t lds_p3_wide_f32() { return ((size_t)kWT * 4 + (size_t)kWT * (64 + 4) + (size_t)kWT * (128 + 4)) * sizeof(float); }

// K = 3 lift for a 128-row tile with 512 threads: 32 lanes cover 32 channels of one row, 16 rows per pass
__device__ __forceinline__ void layer1_wide(const float* __restrict__ xs, const Layer1W& L, float* __restrict__ out, int tid)
{
  constexpr int ldo = 64 + 4;
  const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = c0 + 32 * g;
    const float w0 = L.w0[g], wa = L.wa[g], wb = L.wb[g], s = L.s[g], t = L.t[g];
#pragma unroll
    for (int rr = 0; rr < kWT / (kWW * 2); ++rr) {
      const int row = rr * (kWW * 2) + r0;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
      const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
      out[row * ldo + c] = fmaxf(fmaf(acc, s, t), 0.f);   // rows past the cloud's end repeat its last point
    }
  }
}

// e = max(e, v) with the ordinal of the winning element: strictly greater wins, so the first of equal values is kept (as the compare /
// select form of train_fwd_phase23<3>); cnt counts the calls
__device__ __forceinline__ void argmax_step(float v, float& e, int& eo, int& cnt)
{
  asm volatile("v_cmp_gt_f32 vcc, %3, %0\n\t"
               "v_cndmask_b32 %0, %0, %3, vcc\n\t"
               "v_cndmask_b32 %1, %1, %2, vcc\n\t"
               "v_add_u32 %2, 1, %2"
               : "+v"(e), "+v"(eo), "+v"(cnt) : "v"(v) : "vcc");
}

// fp32: exact fp32 MFMA, hand-issued weight stream (mfma_rows<4>: the inference kernel's inner loop)
__global__ __launch_bounds__(kWW * 64, 2) void train_fwd_phase3_wide(const TrainFwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int C1 = 64, C2 = 128, ld0 = C1 + 4, ld1 = C2 + 4, KG2 = C1 / 8, KG3 = C2 / 8;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  float* xs = smem;
  const int off0 = kWT * 4, off1 = kWT * 4 + kWT * ld0;   // integer offsets keep the LDS address space visible (ds_read_b128)
  const int CT3 = (a.C3 + 31) >> 5;
  const int ntiles = (a.N + kWT - 1) / kWT;

  const Layer1W l1w = layer1_load(a.w1, C1, a.sc1 + tower * C1, a.sh1 + tower * C1, tid);
  // layer-2 item of this wave: channel tile wave >> 1, 64-row group wave & 1 (a weight fragment feeds two row tiles)
  const int ct2 = wave >> 1, rg2 = wave & 1, col2 = ct2 * 32 + (lane & 31);
  const float sc2 = a.sc2[tower * C2 + col2], sh2 = a.sh2[tower * C2 + col2];
  double cs2 = 0.0;                       // column sum of h2: this lane's rows of column col2, whole cloud
  // running extreme of sgn * (z3 - bias) and its point, per (channel tile slot, lane): registers for the whole cloud
  float be[kWSlots], sg[kWSlots];
  int bi[kWSlots];
#pragma unroll
  for (int q = 0; q < kWSlots; ++q) {
    const int col = (wave + q * kWW) * 32 + (lane & 31);
    be[q] = -INFINITY; bi[q] = 0;
    sg[q] = col < a.C3 ? a.sgn3[tower * a.C3 + col] : 1.f;
  }

  // the tile's raw points are requested one tile ahead (threads 0 .. 127)
  float nx = 0.f, ny = 0.f, nz = 0.f;
  auto request_xyz = [&](int t) {
    if (tid < kWT) {
      const float* q = pc + (size_t)min(t * kWT + tid, a.N - 1) * 3;
      nx = q[0]; ny = q[1]; nz = q[2];
    }
  };
  request_xyz(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    const int nvalid = min(kWT, a.N - tile * kWT);
    if (tile) __syncthreads();            // the previous tile's readers are done with xs / h1 / h2
    if (tid < kWT) {
      const float x = nx - xf[0], y = ny - xf[1], z = nz - xf[2];
      xs[tid * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
      xs[tid * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
      xs[tid * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
    }
    if (tile + 1 < ntiles) request_xyz(tile + 1);
    __syncthreads();
#ifndef X_NOL1
    layer1_wide(xs, l1w, smem + off0, tid);
#endif
    __syncthreads();

    // ---- layer 2: h2 = relu(bn2(h1 W2 + b2)) -> LDS, column sums ----
#ifndef X_NOL2
    {
      f32x16 acc[2];
      mfma_rows<2, true, true>(smem + off0 + rg2 * 64 * ld0, ld0, reinterpret_cast<const f32x4*>(a.wp2) + (size_t)ct2 * KG2 * 64, KG2, lane, acc);
      float lsum = 0.f;
      float* out = smem + off1;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rg2 * 64 + acc_row(m, r, lane);
          const float hv = fmaxf(fmaf(acc[m][r], sc2, sh2), 0.f);
          lsum += row < nvalid ? hv : 0.f;   // (rows past the cloud's end: copies of the last point, not part of the batch)
          out[row * ld1 + col2] = hv;
        }
      cs2 += (double)lsum;
    }
#endif
    __syncthreads();

    // ---- keep h2 for the Gram and the sparse (arg-max) part of the backward: coalesced rows out of the LDS tile ----
#ifndef X_NOSTORE
    if (!(a.dbg & 2)) {
      float* dst = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kWT) * C2;
      constexpr int c4 = C2 / 4;
#pragma unroll
      for (int j = 0; j < kWT * c4 / (kWW * 64); ++j) {
        const int i = tid + j * kWW * 64, row = i / c4, q = i % c4;
        if (row < nvalid)
          *reinterpret_cast<f32x4*>(dst + (size_t)row * C2 + q * 4) = *reinterpret_cast<const f32x4*>(smem + off1 + row * ld1 + q * 4);
      }
    }
#endif

    // ---- layer 3: z3 = h2 W3 + b3: extreme of sgn * (z3 - b3) over the cloud's points (statistics: stat3_pool_finish_kernel) ----
    // (not unrolled: with the four channel tiles unrolled hipcc keeps the A fragments -- the same LDS reads for every channel tile -- live
    //  across them and spills; the slot's running extreme is selected in and out of its register by compares against the loop counter)
#pragma unroll 1
    for (int q = 0; q < kWSlots; ++q) {
      const int ct = wave + q * kWW;
      if (ct >= CT3) break;
      float e = be[0], s = sg[0]; int ei = bi[0];
#pragma unroll
      for (int u = 1; u < kWSlots; ++u)
        if (u == q) { e = be[u]; ei = bi[u]; s = sg[u]; }
      asm volatile("" ::: "memory");
      f32x16 acc[4];
      mfma_rows<4, true, true>(smem + off1, ld1, reinterpret_cast<const f32x4*>(a.wp3) + (size_t)ct * KG3 * 64,
#ifdef WIDE_KG_RT
                               a.C2 >> 3,
#else
                               KG3,
#endif
                               lane, acc);
      // running extreme + its accumulator ordinal (16 m + r).  Spelled in asm with a counter register: written as
      // `if (v > e) { e = v; ei = <row constant>; }` hipcc materialises the 64 row constants in VGPRs, hoists them out of the tile loop
      // and the kernel needs 120 registers more (167 spills).  The ordinal is turned into the point index when the slot is written back.
      int cnt = 0, eo = -1;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) argmax_step(acc[m][r] * s, e, eo, cnt);
      // (rows past the cloud's end -- last tile of a cloud whose size is not a multiple of 128 -- are copies of the last point: they tie
      //  with it, a lane keeps the first of equal values = the lower row, and the index is clamped to N - 1 when it is written)
      if (eo >= 0) ei = tile * kWT + (eo >> 4) * 32 + (eo & 3) + 8 * ((eo & 15) >> 2) + 4 * half;
#pragma unroll
      for (int u = 0; u < kWSlots; ++u)
        if (u == q) { be[u] = e; bi[u] = ei; }
    }
  }
  {
    float* my_ext = a.ext + ((size_t)cloud * 2 + half) * a.C3;
    int* my_idx = a.idx + ((size_t)cloud * 2 + half) * a.C3;
#pragma unroll
    for (int q = 0; q < kWSlots; ++q) {
      const int col = (wave + q * kWW) * 32 + (lane & 31);
      if (col < a.C3 && !(a.dbg & 4)) { my_ext[col] = be[q]; my_idx[col] = min(bi[q], a.N - 1); }
    }
    a.colsum_part[((size_t)cloud * 4 + rg2 * 2 + half) * C2 + col2] = cs2;
  }
}

}  // namespace alignnet
