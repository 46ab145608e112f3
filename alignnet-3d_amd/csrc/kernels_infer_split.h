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

// two fp32 -> (hi, lo) bf16 pairs: hi = RNE(x) (one v_cvt_pk_bf16_f32 for both), lo = RNE(x - hi); the same values split_bf16 gives
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16_pair(float x0, float x1, unsigned& hi, unsigned& lo)
{
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 d = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
}

// The K = 3 lift of a 128-point tile for C1 = 64 on the matrix pipe (round 6; the DGCNN kernels lift this way since rounds 3 / 5): k padded to 4 -- xs rows are
// {x', y', z', 0} and the weight fragment of the fourth k is zero -- and the [128 rows][64 channels] output cut into 16 x 16 tiles of v_mfma_f32_16x16x4_f32, ONE
// instruction per tile: wave w takes rows 16 w .. 16 w + 15 and the four channel tiles, as the TRANSPOSED product (weights as the A operand), so a lane ends up with
// one row and four adjacent channels -- one v_cvt_pk pair and one 8-byte store each for the hi and the lo tile.  Per lane and tile: one 4-byte LDS read, four
// MFMAs, 4 x (two 16-byte parameter reads, ~20 vector instructions, two 8-byte stores) instead of 16 x (5 vector instructions) + the splits of the VALU form.
// w1 [3][64], sc / sh [64] (LDS tables or global memory); th / tl: the tile's hi / lo images, row stride 72.  Both split kernels of the shipped widths call
// this, so they stay bit-identical to each other (tests/test_forward_gpu.py); against the VALU lift the products are summed by the MFMA's own chain (the 1e-4
// oracle bar holds with three decimal orders to spare).
__device__ __forceinline__ void split_lift64_mfma(const float* __restrict__ xs, const float* __restrict__ w1, const float* __restrict__ sc, const float* __restrict__ sh,
                                                  unsigned short* __restrict__ th, unsigned short* __restrict__ tl, int wave, int lane)
{
  constexpr int ld1 = 72;
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const int n = lane & 15, kq = lane >> 4, row = 16 * wave + n;
  const float b = xs[row * 4 + kq];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const float wv = w1[min(kq, 2) * 64 + 16 * ct + n];
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kq < 3 ? wv : 0.f, b, acc, 0, 0, 0);
    const f32x4_ s4 = *reinterpret_cast<const f32x4_*>(sc + 16 * ct + 4 * kq), t4 = *reinterpret_cast<const f32x4_*>(sh + 16 * ct + 4 * kq);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[r], s4[r], t4[r]), 0.f);
    unsigned h0, l0, h1, l1;
    split_bf16_pair(v[0], v[1], h0, l0);
    split_bf16_pair(v[2], v[3], h1, l1);
    const int o = row * ld1 + 16 * ct + 4 * kq;
    *reinterpret_cast<u32x2*>(th + o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(tl + o) = u32x2{l0, l1};
  }
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
  int prio_mask = 0x7f;             // pointnet_split_persist: the k-blocks of an item in which waves 0 .. 3 run at priority 2 (else 0; waves 4 .. 7 at 1)
  long long* stamps = nullptr;      // ablation build only, dbg & 64: cycle stamps of workgroup 0's waves at the phase boundaries of its third tile
  int dbg = 0;                      // ablation build only (ablate.h): 1 = last layer without its LDS reads, 2 = without its weight requests, 4 = no lift / hidden layer, 8 = no last layer
};

// acc[m] += A[rows 32 m ..][16 kb ..] * W block, three bf16 MFMAs per product
template <int MR, bool PRIO = true>
__device__ __forceinline__ void split_mfma(const bf16x8 (&ah)[MR], const bf16x8 (&al)[MR], const bf16x8& bh, const bf16x8& bl,
                                           f32x16 (&acc)[MR])
{
#ifdef ALIGNNET_SETPRIO
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
#ifdef ALIGNNET_SETPRIO
  if (PRIO) __builtin_amdgcn_s_setprio(0);
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
    xs[tid * 4 + 0] = fmaf(z, xf[9], fmaf(y, xf[6], x * xf[3]));
    xs[tid * 4 + 1] = fmaf(z, xf[10], fmaf(y, xf[7], x * xf[4]));
    xs[tid * 4 + 2] = fmaf(z, xf[11], fmaf(y, xf[8], x * xf[5]));
    xs[tid * 4 + 3] = 0.f;   // (the k = 4 padding of the matrix-pipe lift)
  }
  __syncthreads();

  // ---- layer 1 (K = 3, fp32) -> h1 as hi / lo bf16 tiles; columns C1 .. K1 are zero padding ----
  if (C1T == 64) split_lift64_mfma(xs, a.w1, a.sc1 + tower * 64, a.sh1 + tower * 64, s16 + o1h, s16 + o1l, wave, lane);   // shipped width: on the matrix pipe
  else {
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


// ---- round 5: the shipped widths (64, 128) as a PERSISTENT workgroup with a two-channel-tile register block in the last layer ----
// What the one-tile-per-workgroup kernel above loses: (1) with 108 KB of LDS there is one workgroup per CU, so every
// 128-point tile pays the workgroup launch, the cold xyz round trip and the first weight fragments' round trip with nothing to run under
// them (~2 of the ~3.9 us a tile spends before its last layer); (2) in the last layer every wave reads the whole h2 tile (hi + lo, 64 KB)
// from LDS for each 32-channel tile: 2 ds_read_b128 per 3 MFMAs = the LDS pipe 67 % busy next to a matrix pipe that wants 100 % (the ablation
// build later showed these reads are not what the last layer waits for: what the two-tile block buys is the smaller weight ring).
// Here a workgroup walks tiles t = blockIdx.x, + gridDim.x, ...: the next tile's points are requested before the last layer and sit in three
// registers under it, the rolling weight ring runs across the tile boundary (the first item's fragments are the same for every tile), the
// per-column parameters of all three layers live in LDS tables filled once per workgroup; and a wave's item in the last layer is TWO channel
// tiles x MR row blocks (MR = 4: all 128 rows, C3 >= 512; MR = 2: a 64-row half, so that C3 = 256 still has eight items), each A fragment
// feeding 6 MFMAs instead of 3: LDS reads per MFMA halve.  The weight ring holds R k-blocks (x 2 channel tiles x hi / lo) instead of a whole
// channel tile, so 128 accumulator registers fit.  Same arithmetic, same order of accumulation per output as pointnet_split<64, 128>:
// results are bit-identical to it.
constexpr int split_ring(int MR) { return MR == 4 ? 2 : 4; }   // k-blocks of weight fragments in flight per wave: what 256 registers leave

struct SplitPersistLds {   // float offsets of the tables behind the activation tiles
  int p1, p2, p3, total_bytes;
};
__host__ __device__ inline SplitPersistLds split_persist_lds(int C3)
{
  SplitPersistLds o;
  const int act = kSplitTP * 4 + (2 * kSplitTP * (72 + 136)) / 2;   // xs + h1 hi / lo + h2 hi / lo, in floats
  o.p1 = act; o.p2 = o.p1 + 2 * 5 * 64; o.p3 = o.p2 + 2 * 2 * 128;
  o.total_bytes = (o.p3 + 2 * 2 * C3) * 4;
  return o;
}

template <int MR, int R>
__device__ __forceinline__ void split_last_layer(const SplitArgs& a, const unsigned short* s16, const float* p3, __amdgpu_buffer_rsrc_t w3r, int CT3, int wave, int lane,
                                                 int tower, float* dst, bf16x8 (&bh)[2][R], bf16x8 (&bl)[2][R])
{
  constexpr int ld2 = 136, o2h = 2 * kSplitTP * 72, o2l = o2h + kSplitTP * ld2, KB2 = 8;
  const int half = lane >> 5;
  const unsigned loff = (unsigned)lane * 16u;
  const int nitems = MR == 4 ? (CT3 >> 1) : CT3;        // (pair) or (pair, 64-row half)
  for (int item = wave; item < nitems; item += kWaves) {
    const int pair = MR == 4 ? item : (item >> 1), rg = MR == 4 ? 0 : (item & 1);
    const int nitem = item + kWaves < nitems ? item + kWaves : wave;     // past this tile's last item: the next tile's first
    const int npair = MR == 4 ? nitem : (nitem >> 1);
    const int arow = (rg * 64 + (lane & 31)) * ld2 + half * 8;
    float hi0 = -INFINITY, lo0 = INFINITY;
    f32x16 acc[2][MR];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb) {
      const int slot = kb % R;
      bf16x8 ah[MR], al[MR];
      if (ALN_ABL(a.dbg, 1)) {
#pragma unroll
        for (int m = 0; m < MR; ++m) { ah[m] = bh[0][m % R]; al[m] = bl[0][m % R]; }
      } else {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(s16 + o2h + arow + m * 32 * ld2 + kb * 16);
          al[m] = *reinterpret_cast<const bf16x8*>(s16 + o2l + arow + m * 32 * ld2 + kb * 16);
        }
      }
      // Issue arbitration is oldest-first at equal priority: left alone, waves 0 .. 3 take the matrix pipe whenever they can and finish a
      // tile's last layer ~10 k cycles before waves 4 .. 7 of the same SIMDs, which then run alone at half the pipe's rate (cycle stamps,
      // DESIGN.md 4.1b).  The older wave alternates between priority 2 and 0 per k-block, the younger one stays at 1: each wins half the ties.
      if (wave < 4) { if ((a.prio_mask >> kb) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
      else if (kb == 0) __builtin_amdgcn_s_setprio(1);
      split_mfma<MR, false>(ah, al, bh[0][slot], bl[0][slot], acc[0]);
      if (kb < KB2 - 1) {
        split_mfma<MR, false>(ah, al, bh[1][slot], bl[1][slot], acc[1]);
      } else {
        // channel tile 0 is complete: its maxima / minima go BETWEEN channel tile 1's last 3 MR MFMAs (a wave's epilogue is otherwise time
        // in which it issues no MFMA, and the SIMD's other wave reaches its own epilogue at the same moment): same MFMA order as split_mfma
#pragma unroll
        for (int g = 0; g < 3 * MR; ++g) {
          const int m = g % MR, prod = g / MR;
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(prod == 0 ? al[m] : ah[m], prod == 1 ? bl[1][slot] : bh[1][slot], acc[1][m], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 8 * MR; ++st)
            if ((3 * st) / 8 == g) {
              const int mm = st / 8, r = 2 * (st % 8);
              hi0 = fmaxf(fmaxf(hi0, acc[0][mm][r]), acc[0][mm][r + 1]);
              lo0 = fminf(fminf(lo0, acc[0][mm][r]), acc[0][mm][r + 1]);
            }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
      }
      if (ALN_ABL(a.dbg, 2)) continue;
      const int qp = kb + R < KB2 ? pair : npair, qk = (kb + R) % KB2;     // this slot's next user: R k-blocks on
      // buffer loads: resource + scalar offset + one 32-bit lane offset, no 64-bit address pair per stream
      const unsigned wq = (unsigned)(((2 * qp) * KB2 + qk) * 2048);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bh[c][slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w3r, loff, wq + c * KB2 * 2048, 0));
        bl[c][slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w3r, loff, wq + c * KB2 * 2048 + 1024, 0));
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = (2 * pair + c) * 32 + (lane & 31);
      const float sc = p3[(tower * 2 + 0) * a.C3 + col], sh = p3[(tower * 2 + 1) * a.C3 + col];
      float hi = hi0, lo = lo0;
      if (c == 1) {
        hi = -INFINITY; lo = INFINITY;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            hi = fmaxf(fmaxf(hi, acc[1][m][r]), acc[1][m][r + 1]);
            lo = fminf(fminf(lo, acc[1][m][r]), acc[1][m][r + 1]);
          }
      }
      float mx = fmaxf(fmaxf(fmaf(hi, sc, sh), fmaf(lo, sc, sh)), 0.f);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));
    }
  }
}

// grid = (workgroups, 1); C1 = 64, C2 = 128, C3 a multiple of 64, >= 512 for MR = 4 (an item is a channel-tile pair over all 128 rows) and
// >= 256 for MR = 2; dynamic LDS = split_persist_lds(C3).total_bytes
template <int MR>
static __global__ __launch_bounds__(kWaves * 64, 2) void pointnet_split_persist(const SplitArgs a)
{
  constexpr int kC1 = 64, kC2 = 128, ld1 = 72, ld2 = 136, KB1 = 4, KB2 = 8, R = split_ring(MR);
  constexpr int o1h = 0, o1l = kSplitTP * ld1, o2h = 2 * kSplitTP * ld1, o2l = o2h + kSplitTP * ld2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* xs = smem;
  unsigned short* s16 = reinterpret_cast<unsigned short*>(smem + kSplitTP * 4);
  const SplitPersistLds L = split_persist_lds(a.C3);
  float* p1 = smem + L.p1;   // [tower][w0, w1, w2, sc, sh][64]
  float* p2 = smem + L.p2;   // [tower][sc, sh][128]
  float* p3 = smem + L.p3;   // [tower][sc, sh][C3]
  const int CT3 = a.C3 >> 5;
  const int tiles_per_cloud = (a.N + kSplitTP - 1) / kSplitTP, ntiles = 2 * a.B * tiles_per_cloud;
  const bf16x8* img2 = reinterpret_cast<const bf16x8*>(a.w2s);
  const bf16x8* img3 = reinterpret_cast<const bf16x8*>(a.w3s);
  constexpr bool wide = MR == 4;
  // the W3 image as a buffer resource (raw, dword-3 flags of gfx9: 32-bit data format), CT3 x 8 k-blocks x (hi, lo) x 1 KB
  const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w3s), 0, CT3 * KB2 * 2048, 0x00020000);
  const int first_pair = wide ? wave : (wave >> 1);

  // the ring's first R k-blocks of this wave's first item: in flight under the table fill and the first tile's prologue
  bf16x8 bh[2][R], bl[2][R];
#pragma unroll
  for (int kb = 0; kb < R; ++kb)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bh[c][kb] = img3[(((size_t)(2 * first_pair + c) * KB2 + kb) * 2) * 64 + lane];
      bl[c][kb] = img3[(((size_t)(2 * first_pair + c) * KB2 + kb) * 2 + 1) * 64 + lane];
    }
  int t = blockIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (t < ntiles && tid < kSplitTP) {
    const int cloud = t / tiles_per_cloud, tile = t - cloud * tiles_per_cloud, tower = cloud >= a.B, b = cloud - tower * a.B;
    const float* p = a.pcs[tower] + ((size_t)b * a.N + min(tile * kSplitTP + tid, a.N - 1)) * 3;
    px = p[0]; py = p[1]; pz = p[2];
  }
  for (int i = tid; i < 2 * 5 * 64; i += kWaves * 64) {
    const int tw = i / 320, j = i - tw * 320, which = j >> 6, c = j & 63;
    p1[i] = which < 3 ? a.w1[which * kC1 + c] : (which == 3 ? a.sc1 : a.sh1)[tw * kC1 + c];
  }
  for (int i = tid; i < 2 * 2 * 128; i += kWaves * 64) {
    const int tw = i >> 8, which = (i >> 7) & 1, c = i & 127;
    p2[i] = (which ? a.sh2 : a.sc2)[tw * kC2 + c];
  }
  for (int i = tid; i < 4 * a.C3; i += kWaves * 64) {
    const int tw = i / (2 * a.C3), j = i - tw * 2 * a.C3, which = j >= a.C3, c = j - which * a.C3;
    p3[i] = (which ? a.sh3 : a.sc3)[tw * a.C3 + c];
  }

#define PSP_STAMP(i) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && lane == 0 && t == 2 * (int)gridDim.x) a.stamps[wave * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
  for (; t < ntiles; t += gridDim.x) {
    PSP_STAMP(0);
    const int cloud = t / tiles_per_cloud, tower = cloud >= a.B, b = cloud - tower * a.B;
    // the hidden layer's fragments (this wave's item): requested here, used two barriers on
    bf16x8 w2h[KB1], w2l[KB1];
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
      w2h[kb] = img2[(((size_t)(wave >> 1) * KB1 + kb) * 2) * 64 + lane];
      w2l[kb] = img2[(((size_t)(wave >> 1) * KB1 + kb) * 2 + 1) * 64 + lane];
    }
    // ---- p' = (p - c) @ R from the points requested one tile ago ----
    if (tid < kSplitTP) {
      const float* xf = a.xform + (size_t)cloud * 12;
      const float x = px - xf[0], y = py - xf[1], z = pz - xf[2];
      const f32x4 pr = {fmaf(z, xf[9], fmaf(y, xf[6], x * xf[3])), fmaf(z, xf[10], fmaf(y, xf[7], x * xf[4])), fmaf(z, xf[11], fmaf(y, xf[8], x * xf[5])), 0.f};
      *reinterpret_cast<f32x4*>(xs + tid * 4) = pr;   // {x', y', z', 0}: the k = 4 padding of the matrix-pipe lift
    }
    __syncthreads();   // also: the tables (first tile)
    PSP_STAMP(1);
    // ---- layer 1 (K = 3, fp32, matrix pipe: split_lift64_mfma) -> h1 hi / lo ----
    if (!(ALN_ABL(a.dbg, 4))) {
      const float* q = p1 + tower * 320;
      split_lift64_mfma(xs, q, q + 192, q + 256, s16 + o1h, s16 + o1l, wave, lane);
    }
    PSP_STAMP(2);
    __syncthreads();   // h1 complete; every wave has left the previous tile's last layer (h2 may be overwritten)
    PSP_STAMP(3);
    // ---- the next tile's points: three registers under the hidden layer and the last layer ----
    {
      // unconditional on a clamped tile and row (a conditional load is waited for inside its exec-masked block)
      const int tn = min(t + (int)gridDim.x, ntiles - 1);
      const int cn = tn / tiles_per_cloud, tile = tn - cn * tiles_per_cloud, twn = cn >= a.B, bn = cn - twn * a.B;
      const float* p = a.pcs[twn] + ((size_t)bn * a.N + min(tile * kSplitTP + (tid & (kSplitTP - 1)), a.N - 1)) * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    // ---- layer 2: wave = (channel tile, 64-row half) ----
    if (!(ALN_ABL(a.dbg, 4))) {
      const int ct = wave >> 1, rg = wave & 1;
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      const int arow = (rg * 64 + (lane & 31)) * ld1 + half * 8;
#pragma unroll
      for (int kb = 0; kb < KB1; ++kb) {
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(s16 + o1h + arow + m * 32 * ld1 + kb * 16);
          al[m] = *reinterpret_cast<const bf16x8*>(s16 + o1l + arow + m * 32 * ld1 + kb * 16);
        }
        // TRANSPOSED product (weights as the A operand): a lane ends up with one point and sixteen channels, four adjacent ones per
        // accumulator quad -- h2 leaves as 8-byte packed stores.  Same products in the same order per output as split_mfma.
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[kb], al[m], acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2l[kb], ah[m], acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[kb], ah[m], acc[m], 0, 0, 0);
      }
      const float* q2 = p2 + tower * 256 + ct * 32 + 4 * half;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(q2 + 8 * qd), sh = *reinterpret_cast<const f32x4*>(q2 + 128 + 8 * qd);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int row = rg * 64 + m * 32 + (lane & 31), col = ct * 32 + 8 * qd + 4 * half;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(acc[m][4 * qd + j], sc[j], sh[j]), 0.f);
          unsigned h0, l0, h1, l1;
          split_bf16_pair(v[0], v[1], h0, l0);
          split_bf16_pair(v[2], v[3], h1, l1);
          const u32x2 hi = {h0, h1}, lo = {l0, l1};
          *reinterpret_cast<u32x2*>(s16 + o2h + row * ld2 + col) = hi;
          *reinterpret_cast<u32x2*>(s16 + o2l + row * ld2 + col) = lo;
        }
      }
    }
    PSP_STAMP(4);
    __syncthreads();
    PSP_STAMP(5);
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
    if (!(ALN_ABL(a.dbg, 8))) split_last_layer<MR, R>(a, s16, p3, w3r, CT3, wave, lane, tower, dst, bh, bl);
    PSP_STAMP(6);
  }
}

}  // namespace alignnet
