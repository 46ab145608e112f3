// Phase 3 of the training forward (kernels_train_fwd.h) on 128-point tiles for the shipped widths C1 = 64, C2 = 128
// (models/tp8.py:49-59 with is_training = True; every configs/*.json of the reference but default.json), gfx950 only.
//
// Why a second tile shape: with 64-point tiles (train_fwd_phase23<3, ...>) a weight fragment of the 128 -> C3 lift feeds two
// 32-row MFMA tiles, with 128-point tiles four -- half the L2 -> CU weight stream per FLOP.  The inference kernel measured exactly
// this trade (0.76 of the fp32-MFMA roofline with two row tiles per fragment, 0.85 with four), and the 64-point
// training kernel sat at 0.725.  One workgroup of eight waves per cloud (two per SIMD, one workgroup per CU); same outputs, same
// layouts as train_fwd_phase23<3> (ext / idx per lane half, column sums of h2 in four slices, h2 stored row-major, upper Gram
// blocks per cloud), so everything downstream is unchanged.
//   train_fwd_phase3_wide<GIVEN, GRAM>   fp32: hand-issued weight stream, running arg-max in registers, optionally on given features
//                                        (dgcnn / hybrid point conv) and with the Gram of the features in the same pass
//   train_fwd_phase3_wide_bf16           bf16 operands: the next tile's prologue software-pipelined around the current tile's lift
// Option "train_phase3_tile64" (or ALIGNNET_P3_TILE64=1) switches back to the 64-point kernels; tests/test_train_gpu.py::
// test_phase3_tile_shapes_agree compares the two (bf16: bit-identical gradients in the test's shapes; over random shapes about one case in three,
// the others differ by a value pushed over a bf16 rounding boundary downstream of the regrouped column sums: tools/stress_tile_shapes.py).
#pragma once
#include "ablate.h"
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

// One step of the running arg-max, e = max(e, v) with eo = 1 + the ordinal of the winning element (0: none yet); strictly greater wins, so
// the first of equal values is kept (as the compare / select form of train_fwd_phase23<3>).  gfx940 / gfx950 need two wait states between
// a VALU write of VCC and a VALU read of it; hipcc inserts them for its own instructions, not inside an asm block (without them the selects
// can read the PREVIOUS compare's mask).  The two slots are filled with work: the next element's product vn = an * s and the counter.
__device__ __forceinline__ void argmax_step(float v, float& e, int& eo, int& cnt, float an, float s, float& vn)
{
  asm volatile("v_cmp_gt_f32 vcc, %4, %0\n\t"
               "v_mul_f32 %3, %5, %6\n\t"
               "v_add_u32 %2, 1, %2\n\t"
               "v_cndmask_b32 %0, %0, %4, vcc\n\t"
               "v_cndmask_b32 %1, %1, %2, vcc"
               : "+v"(e), "+v"(eo), "+v"(cnt), "=&v"(vn) : "v"(v), "v"(an), "v"(s) : "vcc");
}

// fp32: exact fp32 MFMA, hand-issued weight stream (mfma_rows<4>: the inference kernel's inner loop)
// GIVEN: the tile of hidden features is read from h2_store (the DGCNN branch's pooled edge features p = max_k h2, kernels_train_dgcnn.h)
// instead of being recomputed from xyz; column sums and the store belong to the producer (as train_fwd_phase23<3, false, true>).
// GRAM: the Gram matrix h2^T h2 (upper 32 x 32 blocks, what gram_h2_kernel produces from the stored h2 in a pass of its own: 268 MB read per
// launch) is accumulated here from the LDS tile, register-resident for the whole cloud.  Ten blocks on eight waves, balanced: wave w owns
// block w over all rows and a quarter of the rows of block 8 (waves 0-3) or 9 (waves 4-7) -- 80 MFMAs per wave and tile next to the lift's
// 1024; the four quarters meet in LDS after the last tile (fixed order: deterministic).
template <bool GIVEN = false, bool GRAM = false>
__global__ __launch_bounds__(kWW * 64, 2) void train_fwd_phase3_wide(const TrainFwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int C1 = 64, C2 = 128, ld0 = C1 + 4, ld1 = C2 + 4, KG2 = C1 / 8, KG3 = C2 / 8;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts3, part = vcloud - cloud * a.parts3, tower = cloud >= a.B, b = cloud - tower * a.B;   // (TrainFwdArgs::parts3)
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const XForm XF = xform_load(xf);   // (scalar registers: store_xs read the frame from memory for every tile)
  float* xs = smem;
  const int off0 = kWT * 4, off1 = kWT * 4 + kWT * ld0;   // integer offsets keep the LDS address space visible (ds_read_b128)
  const int CT3 = (a.C3 + 31) >> 5;
  const int nt_all = (a.N + kWT - 1) / kWT, tile0 = part * nt_all / a.parts3, ntiles = (part + 1) * nt_all / a.parts3;   // this workgroup's tiles [tile0, ntiles)

  Layer1W l1w = {};
  if (!GIVEN) l1w = layer1_load(a.w1, C1, a.sc1 + tower * C1, a.sh1 + tower * C1, tid);
  // layer-2 item of this wave: channel tile wave >> 1, 64-row group wave & 1 (a weight fragment feeds two row tiles)
  const int ct2 = wave >> 1, rg2 = wave & 1, col2 = ct2 * 32 + (lane & 31);
  const float sc2 = GIVEN ? 0.f : a.sc2[tower * C2 + col2], sh2 = GIVEN ? 0.f : a.sh2[tower * C2 + col2];
  double cs2 = 0.0;                       // column sum of h2: this lane's rows of column col2, whole cloud
  // running extreme of sgn * (z3 - bias) and its point, per (channel tile slot, lane): registers for the whole cloud
  float be[kWSlots];
  int bi[kWSlots];
#pragma unroll
  for (int q = 0; q < kWSlots; ++q) { be[q] = -INFINITY; bi[q] = 0; }

  // Gram blocks in the order (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
  f32x16 gacc[GRAM ? 2 : 1];
  int git = 0, gjt = 0;
  const int git1 = wave < 4 ? 2 : 3, gq = wave & 3;   // second unit: rows 32 gq .. 32 gq + 31 of block (2,3) / (3,3)
  if (GRAM) {
    int rem = wave;
    while (rem >= 4 - git) { rem -= 4 - git; ++git; }
    gjt = git + rem;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
  }

  // the tile's raw points are requested one tile ahead (threads 0 .. 127)
  float nx = 0.f, ny = 0.f, nz = 0.f;
  auto request_xyz = [&](int t) {
    if (tid < kWT) {
      const float* q = pc + (size_t)min(t * kWT + tid, a.N - 1) * 3;
      nx = q[0]; ny = q[1]; nz = q[2];
    }
  };
  auto store_xs = [&]() {
    if (tid < kWT) {
      const float x = nx - XF.v[0], y = ny - XF.v[1], z = nz - XF.v[2];
      xs[tid * 4 + 0] = x * XF.v[3] + y * XF.v[6] + z * XF.v[9];
      xs[tid * 4 + 1] = x * XF.v[4] + y * XF.v[7] + z * XF.v[10];
      xs[tid * 4 + 2] = x * XF.v[5] + y * XF.v[8] + z * XF.v[11];
    }
  };
  // The K = 3 lift of tile t + 1 runs INSIDE the lift of tile t (behind the wave's first channel tile): h1's buffer is free once the hidden
  // layer of tile t is done, so the points -> barrier -> lift -> barrier chain no longer sits in front of every tile's first MFMA.
  if (!GIVEN) {
    request_xyz(tile0);
    store_xs();
    if (tile0 + 1 < ntiles) request_xyz(tile0 + 1);
    __syncthreads();
    layer1_wide(xs, l1w, smem + off0, tid);
  }
  for (int tile = tile0; tile < ntiles; ++tile) {
    const int nvalid = min(kWT, a.N - tile * kWT);
    __syncthreads();            // h1 of this tile is complete; the previous tile's readers are done with h2
    if (GIVEN) {
      // rows past the cloud's end repeat its last row (they tie with it in the max; the index is clamped when it is written)
      const float* src = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kWT) * C2;
      constexpr int c4 = C2 / 4;
#pragma unroll
      for (int j = 0; j < kWT * c4 / (kWW * 64); ++j) {
        const int i = tid + j * kWW * 64, row = i / c4, q = i % c4;
        *reinterpret_cast<f32x4*>(smem + off1 + row * ld1 + q * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)min(row, nvalid - 1) * C2 + q * 4);
      }
      __syncthreads();
    } else {
    // ---- layer 2: h2 = relu(bn2(h1 W2 + b2)) -> LDS, column sums ----
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
    __syncthreads();
    if (tile + 1 < ntiles) {   // (xs: last read by this tile's lift, a tile ago)
      store_xs();
      if (tile + 2 < ntiles) request_xyz(tile + 2);
    }

    // ---- keep h2 for the Gram and the sparse (arg-max) part of the backward: coalesced rows out of the LDS tile ----
    if (!(ALN_ABL(a.dbg, 2))) {
      float* dst = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kWT) * C2;
      constexpr int c4 = C2 / 4;
      // (the thread's eight row / column offsets are tile-invariant: hoisted out of the tile loop they are 16 address registers held -- or
      //  spilled -- across the lift; laundering tid makes them a few integer operations per tile instead)
      int tl = tid;
      asm volatile("" : "+v"(tl));
#pragma unroll
      for (int j = 0; j < kWT * c4 / (kWW * 64); ++j) {
        const int i = tl + j * kWW * 64, row = i / c4, q = i % c4;
        if (row < nvalid)
          *reinterpret_cast<f32x4*>(dst + (size_t)row * C2 + q * 4) = *reinterpret_cast<const f32x4*>(smem + off1 + row * ld1 + q * 4);
      }
    }

    }   // !GIVEN

    if (GRAM) {
      const float* hb = smem + off1 + half * ld1 + (lane & 31);
      const float* pa0 = hb + git * 32;
      const float* pb0 = hb + gjt * 32;
      const float* pa1 = hb + git1 * 32 + gq * 32 * ld1;
      const float* pb1 = hb + 3 * 32 + gq * 32 * ld1;
      if (nvalid == kWT) {
#pragma unroll 8
        for (int r = 0; r < kWT; r += 2) gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0[r * ld1], pb0[r * ld1], gacc[0], 0, 0, 0);
#pragma unroll 8
        for (int r = 0; r < 32; r += 2) gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1[r * ld1], pb1[r * ld1], gacc[1], 0, 0, 0);
      } else {   // rows past the cloud's end hold copies of its last point (for the max): not part of the batch
#pragma unroll 8
        for (int r = 0; r < kWT; r += 2)
          gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r + half < nvalid ? pa0[r * ld1] : 0.f, pb0[r * ld1], gacc[0], 0, 0, 0);
#pragma unroll 8
        for (int r = 0; r < 32; r += 2)
          gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gq * 32 + r + half < nvalid ? pa1[r * ld1] : 0.f, pb1[r * ld1], gacc[1], 0, 0, 0);
      }
    }

    // ---- layer 3: z3 = h2 W3 + b3: extreme of sgn * (z3 - b3) over the cloud's points (statistics: stat3_pool_finish_kernel) ----
    // (not unrolled: with the four channel tiles unrolled hipcc keeps the A fragments -- the same LDS reads for every channel tile -- live
    //  across them and spills; the slot's running extreme is selected in and out of its register by compares against the loop counter)
#pragma unroll 1
    for (int q = 0; q < kWSlots; ++q) {
      if (!GIVEN && q == 1 && tile + 1 < ntiles) {   // every wave, whatever its number of channel tiles
        __syncthreads();
        layer1_wide(xs, l1w, smem + off0, tid);
      }
      const int ct = wave + q * kWW;
      if (ct >= CT3) continue;
      float e = be[0]; int ei = bi[0];
#pragma unroll
      for (int u = 1; u < kWSlots; ++u)
        if (u == q) { e = be[u]; ei = bi[u]; }
      // sign(gamma3) of the lane's column: requested in front of the MFMA loop (older than the weight stream: its first counted wait covers it)
      const float s = ct * 32 + (lane & 31) < a.C3 ? a.sgn3[tower * a.C3 + ct * 32 + (lane & 31)] : 1.f;
      asm volatile("" ::: "memory");
      f32x16 acc[4];
      mfma_rows<4, true, true>(smem + off1, ld1, reinterpret_cast<const f32x4*>(a.wp3) + (size_t)ct * KG3 * 64,
                               KG3,
                               lane, acc);
      // running extreme + its accumulator ordinal (16 m + r).  Spelled in asm with a counter register: written as
      // `if (v > e) { e = v; ei = <row constant>; }` hipcc materialises the 64 row constants in VGPRs, hoists them out of the tile loop
      // and the kernel needs 120 registers more (167 spills).  The ordinal is turned into the point index when the slot is written back.
      int cnt = 0, eo = 0;
      float v = acc[0][0] * s, vn;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        argmax_step(v, e, eo, cnt, acc[(i + 1 < 64 ? i + 1 : i) >> 4][(i + 1 < 64 ? i + 1 : i) & 15], s, vn);
        v = vn;
      }
      --eo;
      // (rows past the cloud's end -- last tile of a cloud whose size is not a multiple of 128 -- are copies of the last point: they tie
      //  with it, a lane keeps the first of equal values = the lower row, and the index is clamped to N - 1 when it is written)
      if (eo >= 0) ei = tile * kWT + (eo >> 4) * 32 + (eo & 3) + 8 * ((eo & 15) >> 2) + 4 * half;
#pragma unroll
      for (int u = 0; u < kWSlots; ++u)
        if (u == q) { be[u] = e; bi[u] = ei; }
    }
  }
  {
    float* my_ext = a.ext + ((size_t)vcloud * 2 + half) * a.C3;
    int* my_idx = a.idx + ((size_t)vcloud * 2 + half) * a.C3;
#pragma unroll
    for (int q = 0; q < kWSlots; ++q) {
      const int col = (wave + q * kWW) * 32 + (lane & 31);
      if (col < a.C3 && !(ALN_ABL(a.dbg, 4))) { my_ext[col] = be[q]; my_idx[col] = min(bi[q], a.N - 1); }
    }
    if (!GIVEN) a.colsum_part[((size_t)vcloud * 4 + rg2 * 2 + half) * C2 + col2] = cs2;
  }
  if (GRAM) {
    float* my_gram = a.gram_part + (size_t)vcloud * C2 * C2;
    const float zero[16] = {};
    tile_commit(my_gram, C2, git, gjt, C2, C2, gacc[0], lane, zero);
    __syncthreads();                       // every wave is done with the last tile's LDS reads
    float* q4 = smem + off1;               // [8 waves][16][64]
#pragma unroll
    for (int r = 0; r < 16; ++r) q4[(wave * 16 + r) * 64 + lane] = gacc[1][r];
    __syncthreads();
    if (wave == 0 || wave == 4) {
      f32x16 tot;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tot[r] = ((q4[((wave + 0) * 16 + r) * 64 + lane] + q4[((wave + 1) * 16 + r) * 64 + lane]) + q4[((wave + 2) * 16 + r) * 64 + lane]) +
                 q4[((wave + 3) * 16 + r) * 64 + lane];
      tile_commit(my_gram, C2, git1, 3, C2, C2, tot, lane, zero);
    }
  }
}


// ---------------------------------------------------------------------------------
// bf16 operands (train_matmul_bf16, BASELINE.json configs[2]): the same phase on 128-point tiles, software-pipelined.
// In bf16 the lift of a tile is only ~4 k matrix-pipe cycles per wave, so -- unlike fp32 -- the tile's prologue (K = 3 lift, hidden layer,
// Gram, h2 store) is as long as the lift itself; the 64-point kernel hides it behind the second workgroup of the CU, but streams
// every weight fragment for two row tiles only and is bound by that L2 -> CU stream (matrix pipe busy 0.28).  Here one workgroup of
// eight waves owns the CU, a weight fragment feeds four row tiles, and the prologue of tile t + 1 runs under the lift of tile t:
//   S0: { Gram(t), h2 store(t), K = 3 lift(t + 1) }  ||  lift(t), first half of the wave's channel tiles      -- barrier
//   S1: { points(t + 2), hidden layer(t + 1) }       ||  lift(t), second half                                  -- barrier
// with h2 double-buffered (row-major bf16 = A operand of the lift) and one transposed tile (both operands of the Gram).  Waves 0-3
// take the prologue piece first, waves 4-7 the lift first, so that the two waves of a SIMD are in different kinds of work.
// LDS: xs 2 KiB | h1 [128][72] bf16 | h2 2 x [128][136] bf16 | h2^T [128][136] bf16 = 123 KiB.
// Outputs and layouts are those of train_fwd_phase23<3, true, false, 64, 128>.
// ---------------------------------------------------------------------------------
static inline size_t lds_p3_wide_bf16() { return (size_t)kWT * 4 * sizeof(float) + ((size_t)kWT * 72 + 3 * (size_t)kWT * 136) * sizeof(unsigned short); }

__global__ __launch_bounds__(kWW * 64, 2) void train_fwd_phase3_wide_bf16(const TrainFwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int C1 = 64, C2 = 128, ld0h = C1 + 8, ldh = C2 + 8, ldT = kWT + 8, KG2 = C1 / 16, KG3 = C2 / 16;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts3, part = vcloud - cloud * a.parts3, tower = cloud >= a.B, b = cloud - tower * a.B;   // (TrainFwdArgs::parts3)
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const XForm XF = xform_load(xf);   // (scalar registers: store_xs read the frame from memory for every tile)
  float* xs = smem;
  unsigned short* h1h = reinterpret_cast<unsigned short*>(smem + kWT * 4);
  unsigned short* h2h = h1h + kWT * ld0h;            // two tiles
  unsigned short* bufT = h2h + 2 * kWT * ldh;
  const int CT3 = (a.C3 + 31) >> 5;
  const int nct = (CT3 - wave + kWW - 1) / kWW;       // channel tiles wave, wave + 8, ... of this wave (<= kWSlots)
  const int nA = (nct + 1) >> 1;                      // ... of which the first nA are lifted in S0, the rest in S1
  const int nt_all = (a.N + kWT - 1) / kWT, tile0 = part * nt_all / a.parts3, ntiles = (part + 1) * nt_all / a.parts3;   // this workgroup's tiles [tile0, ntiles)
  const bool early = wave < kWW / 2;                  // prologue piece first

  const Layer1W l1w = layer1_load(a.w1, C1, a.sc1 + tower * C1, a.sh1 + tower * C1, tid);
  // hidden-layer item of this wave: channel tile wave >> 1, row tiles 2 (wave & 1) + {0, 1}
  const int ct2 = wave >> 1, rg2 = wave & 1, col2 = ct2 * 32 + (lane & 31);
  const float sc2 = a.sc2[tower * C2 + col2], sh2 = a.sh2[tower * C2 + col2];
  double cs2 = 0.0;
  // the item's four weight fragments stay in registers for the whole cloud (requested per tile they were an exposed L2 round trip in
  // front of eight MFMAs: 3 k cycles per tile)
  bf16x8 w2f[KG2];
#pragma unroll
  for (int kg = 0; kg < KG2; ++kg) w2f[kg] = reinterpret_cast<const bf16x8*>(a.wp2h)[((size_t)ct2 * KG2 + kg) * 64 + lane];
  // Gram: the ten upper 32 x 32 blocks on eight waves (blocks wave, wave + 8), register-resident for the whole cloud
  constexpr int nblk = 10, CT2 = 4;
  f32x16 gacc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
  float rbe[kWSlots]; int rbi[kWSlots];
#pragma unroll
  for (int q = 0; q < kWSlots; ++q) { rbe[q] = -INFINITY; rbi[q] = 0; }
  const bf16x8* wimg = reinterpret_cast<const bf16x8*>(a.wp3h) + (size_t)tower * CT3 * KG3 * 64;
  bf16x8 wf[KG3];                                     // the lift's weight stream (lift3 below)
  if (nct > 0) {
#pragma unroll
    for (int kg = 0; kg < KG3; ++kg) wf[kg] = wimg[((size_t)wave * KG3 + kg) * 64 + lane];
  }

  float nx = 0.f, ny = 0.f, nz = 0.f;
  auto request_xyz = [&](int t) {
    if (tid < kWT) {
      const float* q = pc + (size_t)min(t * kWT + tid, a.N - 1) * 3;
      nx = q[0]; ny = q[1]; nz = q[2];
    }
  };
  auto store_xs = [&]() {
    if (tid < kWT) {
      const float x = nx - XF.v[0], y = ny - XF.v[1], z = nz - XF.v[2];
      xs[tid * 4 + 0] = x * XF.v[3] + y * XF.v[6] + z * XF.v[9];
      xs[tid * 4 + 1] = x * XF.v[4] + y * XF.v[7] + z * XF.v[10];
      xs[tid * 4 + 2] = x * XF.v[5] + y * XF.v[8] + z * XF.v[11];
    }
  };
  // K = 3 lift of tile t -> h1h (bf16; rows past the cloud's end are zero)
  auto lift = [&](int t) {
    const int nvalid = min(kWT, a.N - t * kWT);
    const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int c = c0 + 32 * g;
#pragma unroll
      for (int rr = 0; rr < kWT / (kWW * 2); ++rr) {
        const int row = rr * (kWW * 2) + r0;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float acc = fmaf(p[2], l1w.wb[g], fmaf(p[1], l1w.wa[g], p[0] * l1w.w0[g]));
        // (and-mask, not a conditional: hipcc turned the conditional into an exec-masked block -- point read, lift, convert -- per element)
        h1h[row * ld0h + c] = (unsigned short)(to_bf16_bits(fmaxf(fmaf(acc, l1w.s[g], l1w.t[g]), 0.f)) & (row < nvalid ? 0xffffu : 0u));
      }
    }
  };
  // hidden layer of tile t: h2 = round(relu(bn2(h1 W2))) -> h2h[t & 1] (row-major) and bufT (transposed); column sums of the rounded values
  auto hidden = [&](int t) {
    const int nvalid = min(kWT, a.N - t * kWT);
    unsigned short* dst = h2h + (t & 1) * kWT * ldh;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    {
      const unsigned short* arow = h1h + (rg2 * 64 + (lane & 31)) * ld0h + (lane >> 5) * 8;
#pragma unroll
      for (int kg = 0; kg < KG2; ++kg)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + m * 32 * ld0h + kg * 16), w2f[kg], acc[m], 0, 0, 0);
    }
    float lsum = 0.f;
    const int lim = nvalid - rg2 * 64 - 4 * half;   // row < nvalid  <=>  32 m + 8 (r >> 2) + (r & 3) < lim
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned short hb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = q * 4 + e, row = rg2 * 64 + acc_row(m, r, lane);
          const float hv = m * 32 + (r & 3) + 8 * (r >> 2) < lim ? fmaxf(fmaf(acc[m][r], sc2, sh2), 0.f) : 0.f;
          hb[e] = to_bf16_bits(hv);
          lsum += __uint_as_float((unsigned)hb[e] << 16);
          dst[row * ldh + col2] = hb[e];
        }
        // rows 8 q + 4 half + 0..3 of row tile m: consecutive in the transposed tile
        uint2 pk; pk.x = hb[0] | ((unsigned)hb[1] << 16); pk.y = hb[2] | ((unsigned)hb[3] << 16);
        *reinterpret_cast<uint2*>(bufT + col2 * ldT + rg2 * 64 + m * 32 + q * 8 + half * 4) = pk;
      }
    cs2 += (double)lsum;
  };
  auto gram = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int item = wave + q * kWW;
      if (item < nblk) {
        int it = 0, rem = item;
        while (rem >= CT2 - it) { rem -= CT2 - it; ++it; }
        const int jt = it + rem;
        const unsigned short* pa = bufT + (it * 32 + (lane & 31)) * ldT + half * 8;
        const unsigned short* pb = bufT + (jt * 32 + (lane & 31)) * ldT + half * 8;
#pragma unroll
        for (int kg = 0; kg < kWT / 16; ++kg)
          gacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pa + kg * 16),
                                                            *reinterpret_cast<const bf16x8*>(pb + kg * 16), gacc[q], 0, 0, 0);
      }
    }
  };
  auto store_h2 = [&](int t) {
    const int nvalid = min(kWT, a.N - t * kWT);
    const unsigned short* src = h2h + (t & 1) * kWT * ldh;
    unsigned short* dst = reinterpret_cast<unsigned short*>(a.h2_store) + ((size_t)cloud * a.N + (size_t)t * kWT) * C2;
    constexpr int c8 = C2 / 8;
#pragma unroll
    for (int j = 0; j < kWT * c8 / (kWW * 64); ++j) {
      const int i = tid + j * kWW * 64, row = i / c8, q = i % c8;
      if (row < nvalid) *reinterpret_cast<f32x4*>(dst + (size_t)row * C2 + q * 8) = *reinterpret_cast<const f32x4*>(src + row * ldh + q * 8);
    }
  };
  // lift of tile t for this wave's channel tile slots [q0, q1): acc = sgn * (z3 - b3) (sign folded into the bf16 image); the epilogue is
  // the key max of train_fwd_phase23<3, true> (low 4 mantissa bits = accumulator register number), over four row tiles
  auto lift3 = [&](int t, int q0, int q1) {
    const int nvalid = min(kWT, a.N - t * kWT);
    const unsigned short* arow = h2h + (t & 1) * kWT * ldh + (lane & 31) * ldh + (lane >> 5) * 8;
    // (not unrolled over the slots: the A fragments are the same LDS reads for every channel tile and hipcc would keep them live across
    //  unrolled copies; the slot's running extreme is selected in and out of its register by compares against the loop counter)
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
      // wf[kg] holds the k-group's fragment of THIS channel tile; right behind the MFMAs that read it, it is re-requested for the next
      // channel tile of the wave's cyclic stream (slots 0 .. nct - 1, tile after tile): seven k-groups = 0.9 k matrix-pipe cycles of lookahead
      const int qn = q + 1 < nct ? q + 1 : 0;
      const bf16x8* wpn = wimg + (size_t)(wave + qn * kWW) * KG3 * 64 + lane;
      f32x16 acc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      bf16x8 av[4], an[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const bf16x8*>(arow + m * 32 * ldh);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kg = 0; kg < KG3; ++kg) {
        if (kg + 1 < KG3) {
#pragma unroll
          for (int m = 0; m < 4; ++m) an[m] = *reinterpret_cast<const bf16x8*>(arow + m * 32 * ldh + (kg + 1) * 16);
        }
        asm volatile("" ::: "memory");   // the A fragments one k-group ahead, not all 32 up front (128 registers)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], wf[kg], acc[m], 0, 0, 0);
        wf[kg] = wpn[kg * 64];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < 4; ++m) av[m] = an[m];
      }
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (nvalid == kWT) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float k0 = __uint_as_float((__float_as_uint(acc[m][r]) & ~15u) | (unsigned)r);
            const float k1 = __uint_as_float((__float_as_uint(acc[m][r + 1]) & ~15u) | (unsigned)(r + 1));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx[m]) : "v"(mx[m]), "v"(k0), "v"(k1));
          }
      } else {
        // rows past the cloud's end (zero rows of h2) never win.  lim is laundered per channel tile: the 64 compares are invariant in q,
        // and hoisted out of the loop they are 64 SGPR pairs (140 scalar spills)
        int lim = nvalid - 4 * half;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = m * 32 + (r & 3) + 8 * (r >> 2) < lim;
            const float k = ok ? __uint_as_float((__float_as_uint(acc[m][r]) & ~15u) | (unsigned)r) : -INFINITY;
            asm("v_max_f32 %0, %1, %2" : "=v"(mx[m]) : "v"(mx[m]), "v"(k));
          }
      }
      // near-ties resolve to the lower row tile
      int msel = 0; float cand = mx[0];
#pragma unroll
      for (int m = 1; m < 4; ++m)
        if (mx[m] > cand) { cand = mx[m]; msel = m; }
      const int ci = t * kWT + acc_row(msel, (int)(__float_as_uint(cand) & 15u), lane);
#pragma unroll
      for (int u = 0; u < kWSlots; ++u)
        if (u == q && cand > rbe[u]) { rbe[u] = cand; rbi[u] = ci; }
    }
  };
  auto pieceA = [&](int t, bool more) { if (!(ALN_ABL(a.dbg, 64))) gram(); if (!(ALN_ABL(a.dbg, 2))) store_h2(t); if (more && !(ALN_ABL(a.dbg, 256))) lift(t + 1); };

  // ---- prologue of tile 0 (not overlapped) ----
  request_xyz(tile0);
  store_xs();
  if (tile0 + 1 < ntiles) request_xyz(tile0 + 1);
  __syncthreads();
  lift(tile0);
  __syncthreads();
  if (tile0 + 1 < ntiles) store_xs();
  hidden(tile0);
  __syncthreads();
  for (int t = tile0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    // ---- S0 ----
    if (early) pieceA(t, more);
    if (!(ALN_ABL(a.dbg, 8))) lift3(t, 0, nA);
    if (!early) pieceA(t, more);
    if (t + 2 < ntiles) request_xyz(t + 2);
    __syncthreads();
    // ---- S1 ----
    if (early && more && !(ALN_ABL(a.dbg, 128))) hidden(t + 1);
    if (!(ALN_ABL(a.dbg, 8))) lift3(t, nA, nct);
    if (!early && more && !(ALN_ABL(a.dbg, 128))) hidden(t + 1);
    if (t + 2 < ntiles) store_xs();
    __syncthreads();
  }
  {
    float* my_ext = a.ext + ((size_t)vcloud * 2 + half) * a.C3;
    int* my_idx = a.idx + ((size_t)vcloud * 2 + half) * a.C3;
#pragma unroll
    for (int q = 0; q < kWSlots; ++q) {
      const int col = (wave + q * kWW) * 32 + (lane & 31);
      if (col < a.C3 && !(ALN_ABL(a.dbg, 4))) { my_ext[col] = rbe[q]; my_idx[col] = rbi[q]; }
    }
    a.colsum_part[((size_t)vcloud * 4 + rg2 * 2 + half) * C2 + col2] = cs2;
    float* my_gram = a.gram_part + (size_t)vcloud * C2 * C2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int item = wave + q * kWW;
      if (item < nblk) {
        int it = 0, rem = item;
        while (rem >= CT2 - it) { rem -= CT2 - it; ++it; }
        const float zero[16] = {};
        tile_commit(my_gram, C2, it, it + rem, C2, C2, gacc[q], lane, zero);
      }
    }
  }
}

}  // namespace alignnet
