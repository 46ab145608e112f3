// Split-bf16 ("bf16x3") variant of the fused PointNet backbone, eval mode, gfx950 only.  Opt-in
// (alignnet_set_option "infer_matmul_bf16x3"); the default inference path is the exact-fp32 pointnet_fused.
//
// Why: fp32-input MFMA runs at 1/16 of the bf16 MFMA rate on gfx950 (157 vs 2500 TFLOP/s).  Writing every fp32 operand as
// x = x_hi + x_lo with x_hi = bf16(x), x_lo = bf16(x - x_hi) (16 significant bits together) and forming
//     x w  ~=  x_hi w_hi + x_hi w_lo + x_lo w_hi          (the dropped x_lo w_lo term is 2^-16 of the product)
// with fp32 accumulation costs three bf16 MFMAs per product -- 16/3 = 5.3x the fp32 MFMA rate -- and lands where fp32
// rounding itself does: against the fp64 oracle the network's outputs differ by 1.4e-5 (centres, the fp32 ulp of a 20 m
// coordinate) and 1.7e-6 (logits), the exact-fp32 path by 1.3e-5 and 1.1e-6 (tests/test_forward_gpu.py).
//
// Structure = pointnet_fused<128>: one workgroup (8 waves) per 128-point tile, layer 1 on the VALU, the hidden layer and
// the last layer on MFMA, max over the tile's points + atomicMax.  Activation tiles live in LDS as two bf16 tiles (hi, lo),
// row stride = K16 + 8 elements (conflict-free 16-byte reads); weight images hold, per (channel tile, 16-wide k block),
// one hi and one lo 16-byte fragment per lane.  The weight fragments of a whole channel tile (<= 16 fragments) sit in
// registers; each is re-requested for the wave's NEXT channel tile right after its last use, so the stream runs a full
// tile (3 k cycles of MFMA) ahead without inline asm.
#pragma once
#include "kernels_infer.h"

namespace alignnet {

constexpr int kSplitTP = 128, kSplitKB1 = 4, kSplitKB2 = 8;   // hidden width <= 64, lift input width <= 128

__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo)
{
  hi = to_bf16_bits(x);
  lo = to_bf16_bits(x - __uint_as_float((unsigned)hi << 16));
}

// Ws[ct][kb][part][lane][8], part 0 = hi, 1 = lo;  element k = 16 kb + 8 (lane >> 5) + s, c = 32 ct + (lane & 31)
static __global__ void pack_weights_split_kernel(const float* __restrict__ W, int K, int C, unsigned short* __restrict__ Ws)
{
  const int KB = (K + 15) >> 4, CT = (C + 31) >> 5;
  const size_t total = (size_t)CT * KB * 512;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int s8 = idx & 7, lane = (idx >> 3) & 63;
    const size_t q = idx >> 9;
    const int kb = q % KB, ct = q / KB;
    const int k = 16 * kb + 8 * (lane >> 5) + s8, c = 32 * ct + (lane & 31);
    unsigned short hi = 0, lo = 0;
    if (k < K && c < C) split_bf16(W[(size_t)k * C + c], hi, lo);
    const size_t base = (((size_t)ct * KB + kb) * 2) * 512 + (size_t)lane * 8 + s8;
    Ws[base] = hi;
    Ws[base + 512] = lo;
  }
}

struct SplitArgs {
  const float* pcs[2]; const float* xform;
  float* pooled; long tower_stride, row_stride;
  int B, N;
  int C1, C2, C3;
  const float* w1;                  // [3][C1] fp32 (K = 3 lift stays on the VALU)
  const unsigned short* w2s;        // split image of W2 [C1][C2]
  const unsigned short* w3s;        // split image of W3 [C2][C3]
  const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3;   // folded BN [2 towers][C]
};

// acc[m] += A[rows 32 m ..][16 kb ..] * W block, three bf16 MFMAs per product
template <int MR>
__device__ __forceinline__ void split_mfma(const bf16x8 (&ah)[MR], const bf16x8 (&al)[MR], const bf16x8& bh, const bf16x8& bl,
                                           f32x16 (&acc)[MR])
{
#ifdef ALIGNNET_SETPRIO
  __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
#ifdef ALIGNNET_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// C1T / C2T != 0: the shipped widths compiled in (strides, k-depths and tile counts become constants)
template <int C1T = 0, int C2T = 0>
static __global__ __launch_bounds__(kWaves * 64, 2) void pointnet_split(const SplitArgs a)
{
  const int kC1 = C1T ? C1T : a.C1, kC2 = C2T ? C2T : a.C2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.y, tile = blockIdx.x;
  const int tower = cloud >= a.B, b = cloud - tower * a.B;
  const int K1 = (kC1 + 15) & ~15, K2 = (kC2 + 15) & ~15, ld1 = K1 + 8, ld2 = K2 + 8;
  const int KB1 = K1 >> 4, KB2 = K2 >> 4;
  float* xs = smem;                                                         // [128][4]
  unsigned short* s16 = reinterpret_cast<unsigned short*>(smem + kSplitTP * 4);
  const int o1h = 0, o1l = kSplitTP * ld1, o2h = 2 * kSplitTP * ld1, o2l = o2h + kSplitTP * ld2;   // element offsets

  // ---- weight fragments first: the hidden layer's (this wave's item) and the first channel tile's of the last layer are
  //      requested before anything else, so their L2 round trips overlap the xyz load and the VALU lift ----
  const int CT2 = (kC2 + 31) >> 5, CT3 = (a.C3 + 31) >> 5;
  const bf16x8* img2 = reinterpret_cast<const bf16x8*>(a.w2s);
  const bf16x8* img3 = reinterpret_cast<const bf16x8*>(a.w3s);
  bf16x8 w2h[kSplitKB1], w2l[kSplitKB1];
  if (wave < CT2 * 2) {
#pragma unroll
    for (int kb = 0; kb < kSplitKB1; ++kb)
      if (kb < KB1) { w2h[kb] = img2[(((size_t)(wave >> 1) * KB1 + kb) * 2) * 64 + lane]; w2l[kb] = img2[(((size_t)(wave >> 1) * KB1 + kb) * 2 + 1) * 64 + lane]; }
  }
  bf16x8 bh[kSplitKB2], bl[kSplitKB2];
  if (wave < CT3) {
#pragma unroll
    for (int kb = 0; kb < kSplitKB2; ++kb)
      if (kb < KB2) { bh[kb] = img3[(((size_t)wave * KB2 + kb) * 2) * 64 + lane]; bl[kb] = img3[(((size_t)wave * KB2 + kb) * 2 + 1) * 64 + lane]; }
  }

  // ---- prologue: p' = (p - c) @ R ----
  if (tid < kSplitTP) {
    const int n = min(tile * kSplitTP + tid, a.N - 1);   // tail rows repeat the last point: max unaffected
    const float* p = a.pcs[tower] + ((size_t)b * a.N + n) * 3;
    const float* xf = a.xform + (size_t)cloud * 12;
    const float x = p[0] - xf[0], y = p[1] - xf[1], z = p[2] - xf[2];
    xs[tid * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
    xs[tid * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
    xs[tid * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
  }
  __syncthreads();

  // ---- layer 1 (K = 3, VALU, fp32) -> h1 as hi / lo bf16 tiles; columns C1 .. K1 are zero padding ----
  {
    const int c0 = tid & 31, r0 = tid >> 5;
    for (int c = c0; c < K1; c += 32) {
      const bool live = c < kC1;
      const float w0 = live ? a.w1[c] : 0.f, w1 = live ? a.w1[kC1 + c] : 0.f, w2 = live ? a.w1[2 * kC1 + c] : 0.f;
      const float sc = live ? a.sc1[tower * kC1 + c] : 0.f, sh = live ? a.sh1[tower * kC1 + c] : 0.f;
#pragma unroll
      for (int rr = 0; rr < kSplitTP / 16; ++rr) {
        const int row = rr * 16 + r0;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float v = fmaxf(fmaf(fmaf(p[2], w2, fmaf(p[1], w1, p[0] * w0)), sc, sh), 0.f);
        unsigned short hi, lo;
        split_bf16(v, hi, lo);
        s16[o1h + row * ld1 + c] = hi;
        s16[o1l + row * ld1 + c] = lo;
      }
    }
  }
  __syncthreads();

  // ---- layer 2: item = (channel tile, 64-row half), one per wave at C2 = 128 ----
  {
    for (int item = wave; item < CT2 * 2; item += kWaves) {
      const int ct = item >> 1, rg = item & 1;
      if (item != wave) {   // wider hidden layers: later items load theirs here (the first item's were requested up front)
#pragma unroll
        for (int kb = 0; kb < kSplitKB1; ++kb)
          if (kb < KB1) { w2h[kb] = img2[(((size_t)ct * KB1 + kb) * 2) * 64 + lane]; w2l[kb] = img2[(((size_t)ct * KB1 + kb) * 2 + 1) * 64 + lane]; }
      }
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      const int arow = (rg * 64 + (lane & 31)) * ld1 + half * 8;
#pragma unroll
      for (int kb = 0; kb < kSplitKB1; ++kb)
        if (kb < KB1) {
          bf16x8 ah[2], al[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            ah[m] = *reinterpret_cast<const bf16x8*>(s16 + o1h + arow + m * 32 * ld1 + kb * 16);
            al[m] = *reinterpret_cast<const bf16x8*>(s16 + o1l + arow + m * 32 * ld1 + kb * 16);
          }
          split_mfma<2>(ah, al, w2h[kb], w2l[kb], acc);
        }
      const int col = ct * 32 + (lane & 31);
      const bool live = col < kC2;
      const float sc = live ? a.sc2[tower * kC2 + col] : 0.f, sh = live ? a.sh2[tower * kC2 + col] : 0.f;
      if (col < K2) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rg * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            unsigned short hi, lo;
            split_bf16(fmaxf(fmaf(acc[m][r], sc, sh), 0.f), hi, lo);   // zero in the padding columns (sc = sh = 0)
            s16[o2h + row * ld2 + col] = hi;
            s16[o2l + row * ld2 + col] = lo;
          }
      }
    }
  }
  __syncthreads();

  // ---- last layer + max over the tile's points.  Wave w owns channel tiles w, w + 8, ...; its weight fragments roll one
  //      full tile ahead ----
  {
    constexpr int MR = kSplitTP / 32;
    const bf16x8* img = img3;
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
    const int arow = (lane & 31) * ld2 + half * 8;
    for (int ct = wave; ct < CT3; ct += kWaves) {
      const int col = ct * 32 + (lane & 31);
      const bool live = col < a.C3;
      const float sc = live ? a.sc3[tower * a.C3 + col] : 0.f, sh = live ? a.sh3[tower * a.C3 + col] : 0.f;
      const int nct = ct + kWaves;
      f32x16 acc[MR];
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < kSplitKB2; ++kb)
        if (kb < KB2) {
          bf16x8 ah[MR], al[MR];
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            ah[m] = *reinterpret_cast<const bf16x8*>(s16 + o2h + arow + m * 32 * ld2 + kb * 16);
            al[m] = *reinterpret_cast<const bf16x8*>(s16 + o2l + arow + m * 32 * ld2 + kb * 16);
          }
          split_mfma<MR>(ah, al, bh[kb], bl[kb], acc);
          if (nct < CT3) {   // this slot's fragments for the wave's next channel tile
            bh[kb] = img[(((size_t)nct * KB2 + kb) * 2) * 64 + lane];
            bl[kb] = img[(((size_t)nct * KB2 + kb) * 2 + 1) * 64 + lane];
          }
        }
      // max_n relu(sc z_n + sh) = max(0, sc * (max_n z_n or min_n z_n) + sh): one v_max3 + one v_min3 per two accumulators,
      // scale / shift / relu once per channel tile (fmaf is monotone in z, so the result is bit-identical)
      float hi = -INFINITY, lo = INFINITY;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(hi), "v"(acc[m][r]), "v"(acc[m][r + 1]));
          asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(lo), "v"(acc[m][r]), "v"(acc[m][r + 1]));
        }
      float mx = fmaxf(fmaxf(fmaf(hi, sc, sh), fmaf(lo, sc, sh)), 0.f);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32 && live) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));   // >= 0: monotone bit pattern
    }
  }
}

}  // namespace alignnet
