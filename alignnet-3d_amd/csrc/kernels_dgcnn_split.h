// Split-bf16 variant of dgcnn_fused (see kernels_infer_split.h for the arithmetic: x = hi + lo bf16, three bf16 MFMAs per
// product, fp32 accumulate), eval mode, gfx950 only.  Covers the shape of every DGCNN use of the reference
// (models/tp8.py:30-46 with widths [C_a <= 64, C_b <= 128, C3]): edge feature -> K = 6 lift (VALU) -> 1x1 conv C_a -> C_b over
// the k neighbours (MFMA) -> max over k -> point conv C_b -> C3 (MFMA) -> max over points.  Other shapes run dgcnn_fused.
//
// Per neighbour slot the edge conv is 12 bf16 MFMAs per wave instead of 32 fp32 MFMAs of twice the length; its split weight
// fragments (4 k blocks x hi/lo) stay in registers for all 20 slots.  The lift writes its output directly as hi / lo bf16
// tiles (double-buffered across slots), and the pooled edge features are split once for the point conv, whose weight
// fragments roll one channel tile ahead as in pointnet_split.
// LDS: es [64][8] f32 | lift tiles [2 buffers][hi, lo][64][Ka+8] bf16 | point features [hi, lo][64][Kb+8] bf16 | es' [64][8] f32.
#pragma once
#include "ablate.h"
#include "kernels_dgcnn.h"
#include "kernels_infer_split.h"

namespace alignnet {

struct DgcnnSplitArgs {
  const float* pcs[2]; const float* xform; const int* nn;
  float* pooled; long tower_stride, row_stride;
  int B, N, k;
  int Ca, Cb, C3;
  const float* w1;                  // [6][Ca] fp32
  const unsigned short* w2s;        // split image of the edge conv [Ca][Cb]
  const unsigned short* w3s;        // split image of the point conv [Cb][C3]
  const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3;   // folded BN [2 towers][C]
  int dbg;
};

// edge layer 0: K = 6 lift on the VALU, es -> hi / lo tiles [64][lda]; columns Ca .. Ka are zero padding.  The thread's
// weights, scale and shift (channels c0 and c0 + 32: Ca <= 64) live in registers for all neighbour slots:
// reloading them from global memory in every slot put an L2 round trip on each slot's critical path.
struct DgLiftW { float w[2][6]; float sc[2], sh[2]; };

__device__ __forceinline__ DgLiftW dg_lift_weights(const DgcnnSplitArgs& a, int tower, int tid)
{
  DgLiftW L;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = (tid & 31) + 32 * g;
    const bool live = c < a.Ca;
    L.sc[g] = live ? a.sc1[tower * a.Ca + c] : 0.f;
    L.sh[g] = live ? a.sh1[tower * a.Ca + c] : 0.f;
#pragma unroll
    for (int d = 0; d < 6; ++d) L.w[g][d] = live ? a.w1[d * a.Ca + c] : 0.f;
  }
  return L;
}

__device__ __forceinline__ void dg_lift_split(const DgcnnSplitArgs& a, int tower, const DgLiftW& L, const float* es, unsigned short* th,
                                              unsigned short* tl, int lda, int Ka, int tid)
{
  const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = c0 + 32 * g;
    if (c < Ka) {
#pragma unroll
      for (int rr = 0; rr < kDgTile / 16; ++rr) {
        const int row = rr * 16 + r0;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(es + row * 8);
        const f32x4 e1 = *reinterpret_cast<const f32x4*>(es + row * 8 + 4);
        float acc = e0[0] * L.w[g][0];
        acc = fmaf(e0[1], L.w[g][1], acc); acc = fmaf(e0[2], L.w[g][2], acc); acc = fmaf(e0[3], L.w[g][3], acc);
        acc = fmaf(e1[0], L.w[g][4], acc); acc = fmaf(e1[1], L.w[g][5], acc);
        unsigned short hi, lo;
        split_bf16(fmaxf(fmaf(acc, L.sc[g], L.sh[g]), 0.f), hi, lo);
        th[row * lda + c] = hi;
        tl[row * lda + c] = lo;
      }
    }
  }
}

// The same lift on the matrix pipe for Ca = 64 (round 5; dgcnn_fused's shipped shape has lifted this way since round 3): k padded to 8 with
// zero weights (the caller zeroes es columns 6 / 7 once), the [64 rows][64 channels] output cut into 16 x 16 tiles of
// v_mfma_f32_16x16x4_f32 (two instructions per tile), two tiles per wave -- as the TRANSPOSED product (weights as the A operand), so a lane
// ends up with one row and four adjacent channels: one v_cvt_pk pair and one 8-byte store each for the hi and the lo tile.  Per thread and
// slot 2 x (two 4-byte LDS reads, two MFMAs, 22 vector instructions, two 8-byte stores) instead of 8 x (14 vector instructions, two 2-byte
// stores) + eight 16-byte reads: the VALU lift was what the edge phase waited for (~1800 vector instructions per SIMD and slot pair next
// to 3 k cycles of MFMA).
struct DgLiftMS { float w[2][2]; float sc[2][4], sh[2][4]; };

__device__ __forceinline__ DgLiftMS dg_liftms_load(const DgcnnSplitArgs& a, int tower, int wave, int lane)
{
  DgLiftMS L;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ctile = (2 * wave + i) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = 4 * ks + (lane >> 4);
      L.w[i][ks] = kk < 6 ? a.w1[kk * 64 + 16 * ctile + (lane & 15)] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      L.sc[i][r] = a.sc1[tower * 64 + 16 * ctile + 4 * (lane >> 4) + r];
      L.sh[i][r] = a.sh1[tower * 64 + 16 * ctile + 4 * (lane >> 4) + r];
    }
  }
  return L;
}

__device__ __forceinline__ void dg_liftms(const DgLiftMS& L, const float* es, unsigned short* th, unsigned short* tl, int wave, int lane)
{
  constexpr int lda = 72;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = 2 * wave + i, rt = t >> 2, ctile = t & 3;
    const float* ar = es + (16 * rt + (lane & 15)) * 8 + (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(L.w[i][0], ar[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(L.w[i][1], ar[4], acc, 0, 0, 0);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[r], L.sc[i][r], L.sh[i][r]), 0.f);
    unsigned h0, l0, h1, l1;
    split_bf16_pair(v[0], v[1], h0, l0);
    split_bf16_pair(v[2], v[3], h1, l1);
    const int o = (16 * rt + (lane & 15)) * lda + 16 * ctile + 4 * (lane >> 4);
    *reinterpret_cast<u32x2*>(th + o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(tl + o) = u32x2{l0, l1};
  }
}

// CaT / CbT != 0: the shipped edge widths compiled in (strides, k-depths and tile counts become constants)
template <int CaT = 0, int CbT = 0>
static __global__ __launch_bounds__(kWaves * 64, 2) void dgcnn_split(const DgcnnSplitArgs a0)
{
  DgcnnSplitArgs a = a0;
  if (CaT) a.Ca = CaT;
  if (CbT) a.Cb = CbT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.y, tile = blockIdx.x;
  const int tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int nvalid = min(kDgTile, a.N - tile * kDgTile);
  const int Ka = (a.Ca + 15) & ~15, Kb = (a.Cb + 15) & ~15, lda = Ka + 8, ldb = Kb + 8;
  const int KBa = Ka >> 4, KBb = Kb >> 4;
  float* es = smem;                                                       // [64][8]
  unsigned short* s16 = reinterpret_cast<unsigned short*>(smem + kDgTile * 8);
  const int tsz = kDgTile * lda;                                          // one lift tile
  const int oP = 4 * tsz, psz = kDgTile * ldb;                            // point features: hi at oP, lo at oP + psz

  // ---- this wave's edge-conv item (channel tile wave >> 1, 32-row block wave & 1): weight fragments resident for all slots ----
  const int CTE = (a.Cb + 31) >> 5;
  const int ect = wave >> 1, em = wave & 1;
  const bool eactive = ect < CTE;
  const bf16x8* img2 = reinterpret_cast<const bf16x8*>(a.w2s);
  bf16x8 eh[kSplitKB1], el[kSplitKB1];
  if (eactive) {
#pragma unroll
    for (int kb = 0; kb < kSplitKB1; ++kb)
      if (kb < KBa) { eh[kb] = img2[(((size_t)ect * KBa + kb) * 2) * 64 + lane]; el[kb] = img2[(((size_t)ect * KBa + kb) * 2 + 1) * 64 + lane]; }
  }
  const int ecol = ect * 32 + (lane & 31);
  const bool elive = eactive && ecol < a.Cb;
  const float esc = elive ? a.sc2[tower * a.Cb + ecol] : 0.f, esh = elive ? a.sh2[tower * a.Cb + ecol] : 0.f;
  f32x16 best;   // running max over the k neighbours of the edge conv's pre-activation; relu folded after the max
#pragma unroll
  for (int r = 0; r < 16; ++r) best[r] = -INFINITY;

  auto edge_conv = [&](int buf) {
    if (!eactive) return;
    const int oh = (2 * buf) * tsz, ol = oh + tsz;
    const int arow = (em * 32 + (lane & 31)) * lda + half * 8;
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < kSplitKB1; ++kb)
      if (kb < KBa) {
        bf16x8 ah[1], al[1];
        ah[0] = *reinterpret_cast<const bf16x8*>(s16 + oh + arow + kb * 16);
        al[0] = *reinterpret_cast<const bf16x8*>(s16 + ol + arow + kb * 16);
        split_mfma<1>(ah, al, eh[kb], el[kb], acc);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = fmaxf(best[r], fmaf(acc[0][r], esc, esh));
  };

  // ---- all (point, neighbour slot) edge features are gathered ONCE, up front: lane = point of the tile, wave w holds slots
  //      w, w + 8, w + 16.  With the edge conv down to ~400 cycles per slot, a per-slot gather (index load, then the
  //      dependent point load: two HBM/L2 round trips) was exposed 20 times per tile. ----
  constexpr int kSlotRegs = 3;   // k <= 24
  float ev[kSlotRegs][6];
  {
    const int n = min(tile * kDgTile + lane, a.N - 1);
    const float* p = pc + (size_t)n * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float cxp = px - xf[0], cyp = py - xf[1], czp = pz - xf[2];
    const float e0 = cxp * xf[3] + cyp * xf[6] + czp * xf[9], e1 = cxp * xf[4] + cyp * xf[7] + czp * xf[10],
                e2 = cxp * xf[5] + cyp * xf[8] + czp * xf[11];
#pragma unroll
    for (int q = 0; q < kSlotRegs; ++q) {
      const int slot = wave + q * kWaves;
      ev[q][0] = e0; ev[q][1] = e1; ev[q][2] = e2; ev[q][3] = 0.f; ev[q][4] = 0.f; ev[q][5] = 0.f;
      if (slot < a.k) {
        const int j = a.nn[((size_t)cloud * a.N + n) * a.k + slot];
        const float* pj = pc + (size_t)j * 3;
        const float dx = pj[0] - px, dy = pj[1] - py, dz = pj[2] - pz;
        ev[q][3] = dx * xf[3] + dy * xf[6] + dz * xf[9];
        ev[q][4] = dx * xf[4] + dy * xf[7] + dz * xf[10];
        ev[q][5] = dx * xf[5] + dy * xf[8] + dz * xf[11];
      }
    }
  }
  float* es2 = reinterpret_cast<float*>(s16 + oP + 2 * psz);   // second edge-feature buffer [64][8], behind the point tiles
  auto write_es = [&](int slot) {   // the wave that holds `slot` publishes its 64 edge features
    if ((slot % kWaves) != wave) return;
    float* dstp = ((slot & 1) ? es2 : es) + lane * 8;
    const int q = slot / kWaves;
#pragma unroll
    for (int u = 0; u < kSlotRegs; ++u)
      if (u == q) {
        *reinterpret_cast<f32x4*>(dstp) = f32x4{ev[u][0], ev[u][1], ev[u][2], ev[u][3]};
        dstp[4] = ev[u][4]; dstp[5] = ev[u][5];
      }
  };
  // two neighbour slots per iteration (slot s -> es / tile buffer 0, slot s+1 -> es' / tile buffer 1): publish both edge
  // features, lift both, run both edge convs -- three barriers per two slots instead of four
  constexpr bool kLiftM = CaT == 64;                       // the lift on the matrix pipe (dg_liftms)
  DgLiftW LW;
  DgLiftMS LM;
  if constexpr (kLiftM) {
    LM = dg_liftms_load(a, tower, wave, lane);
    if (tid < kDgTile) { es[tid * 8 + 6] = 0.f; es[tid * 8 + 7] = 0.f; es2[tid * 8 + 6] = 0.f; es2[tid * 8 + 7] = 0.f; }   // k padding: never written again
  } else LW = dg_lift_weights(a, tower, tid);
  for (int slot = 0; slot < a.k; slot += 2) {
    const bool two = slot + 1 < a.k;
    write_es(slot);
    if (two) write_es(slot + 1);
    __syncthreads();
    if (!(ALN_ABL(a.dbg, 2))) {
      if constexpr (kLiftM) {
        dg_liftms(LM, es, s16, s16 + tsz, wave, lane);
        if (two) dg_liftms(LM, es2, s16 + 2 * tsz, s16 + 3 * tsz, wave, lane);
      } else {
        dg_lift_split(a, tower, LW, es, s16, s16 + tsz, lda, Ka, tid);
        if (two) dg_lift_split(a, tower, LW, es2, s16 + 2 * tsz, s16 + 3 * tsz, lda, Ka, tid);
      }
    }
    __syncthreads();
    if (!(ALN_ABL(a.dbg, 1))) {
      edge_conv(0);
      if (two) edge_conv(1);
    }
    __syncthreads();   // the next pair overwrites es / es' and the tiles
  }

  // ---- relu(max_k) -> point features as hi / lo tiles [64][ldb] (tp8.py:42); rows past the cloud and padding columns are zero ----
  if (eactive && ecol < Kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = em * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      unsigned short hi, lo;
      split_bf16((elive && row < nvalid) ? fmaxf(best[r], 0.f) : 0.f, hi, lo);
      s16[oP + row * ldb + ecol] = hi;
      s16[oP + psz + row * ldb + ecol] = lo;
    }
  }
  // the first channel tile's fragments of the point conv are requested before the barrier
  const int CT3 = (a.C3 + 31) >> 5;
  const bf16x8* img3 = reinterpret_cast<const bf16x8*>(a.w3s);
  bf16x8 bh[kSplitKB2], bl[kSplitKB2];
  if (wave < CT3) {
#pragma unroll
    for (int kb = 0; kb < kSplitKB2; ++kb)
      if (kb < KBb) { bh[kb] = img3[(((size_t)wave * KBb + kb) * 2) * 64 + lane]; bl[kb] = img3[(((size_t)wave * KBb + kb) * 2 + 1) * 64 + lane]; }
  }
  __syncthreads();

  // ---- point conv + max over the tile's points (tp8.py:43-45) ----
  {
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
    const int arow = (lane & 31) * ldb + half * 8;
    for (int ct = wave; ct < ((ALN_ABL(a.dbg, 4)) ? 0 : CT3); ct += kWaves) {
      const int col = ct * 32 + (lane & 31);
      const bool live = col < a.C3;
      const float sc = live ? a.sc3[tower * a.C3 + col] : 0.f, sh = live ? a.sh3[tower * a.C3 + col] : 0.f;
      const int nct = ct + kWaves;
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < kSplitKB2; ++kb)
        if (kb < KBb) {
          bf16x8 ah[2], al[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            ah[m] = *reinterpret_cast<const bf16x8*>(s16 + oP + arow + m * 32 * ldb + kb * 16);
            al[m] = *reinterpret_cast<const bf16x8*>(s16 + oP + psz + arow + m * 32 * ldb + kb * 16);
          }
          split_mfma<2>(ah, al, bh[kb], bl[kb], acc);
          if (nct < CT3) {
            bh[kb] = img3[(((size_t)nct * KBb + kb) * 2) * 64 + lane];
            bl[kb] = img3[(((size_t)nct * KBb + kb) * 2 + 1) * 64 + lane];
          }
        }
      float mx = 0.f;   // relu folded into the max
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < nvalid) mx = fmaxf(mx, fmaf(acc[m][r], sc, sh));
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32 && live && mx > dst[col]) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));   // >= 0: monotone bit pattern; skipped when the tile does not beat the running max
    }
  }
}

}  // namespace alignnet
