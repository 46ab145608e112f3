// Training-mode forward / backward of the DGCNN backbone (models/tp8.py:30-46 with is_training=True,
// utils/tf_util_dgcnn.py:638-706; BatchNorm with batch statistics utils/tf_util_dgcnn.py batch_norm_template), gfx950 only.
//
// Shape handled: widths [C1, C2, C3] = edge conv 6 -> C1, edge conv C1 -> C2 over the B*N*k edge rows, max over the k
// neighbours, point conv C2 -> C3 over the B*N rows, max over the points.  The point conv and everything behind it is
// the shared-MLP machinery (kernels_train_fwd.h / kernels_train_bwd.h) fed with the stored pooled edge features
// p = max_k h2 instead of a recomputed h2 (their GIVEN variants).  What is new here is the edge part:
//
//   forward   phase 1: BN statistics of z1 = e W1 + b1.  z1 is linear in the 6-vector e = [x_i, x_j - x_i], so its
//                      per-channel sum / sum of squares follow from the cloud's first and second moments of e
//                      (27 numbers, fp64) -- no [B*N*k, C1] pass at all.
//             phase 2: h1 = relu(bn1(z1)) recomputed per (tile, slot); z2 = h1 W2 + b2 on MFMA: its statistics AND, because
//                      max_k relu(g zhat_k + b) = relu(g zhat* + b) with zhat* the extreme of sign(g) z over the slots and
//                      sign(gamma2) known up front, the per-(point, channel) extreme and its slot ([B*N, C2] floats +
//                      bytes) -- the edge conv runs once; column sums of h1.
//             finish:  p = relu(bn2(extreme)) in place, column sums of p (elementwise).
//   backward  B2 (GIVEN) gives dp; only the arg-k row of each (point, channel) receives it:
//                      dy2[n,s,c] = dp[n,c] [s == argk[n,c]] [p > 0].
//             edge pass: per (tile, slot): h1_s recomputed, dy2_s built from the register-resident dp tile,
//                      dh1 = dy2_s V2 + h1_s Q2 + q2b, dy1 = dh1 [h1 > 0]; U2 += h1_s^T dy2_s and Gram(h1) in registers;
//                      dy1 is never stored: the first layer is linear in e, so dbeta1, dgamma1, dW1 and the per-cloud
//                      frame gradients follow from  Pdy = sum e^T dy1 (6 x C1), sum dy1 (C1)  and the moments of e.
#pragma once
#include "ablate.h"
#include "kernels_train_bwd.h"
#include "kernels_dgcnn.h"

namespace alignnet {

constexpr int kDgK = 20;          // neighbours per point (models/tp8.py:33)
constexpr int kDgMom = 27;        // sum e (6) | upper triangle of sum e e^T (21, row-major d <= d2)

struct DgTrainArgs {
  const float* pcs[2]; const float* xform; const int* nn;   // nn: [2B][N][k]
  int B, N, k, C1, C2;
  int ld0;                        // LDS leading dim of the h1 tile
  const float* w1; const float* b1;   // [6][C1], [C1]
  const float* wp2; const float* b2;  // MFMA image of W2 [C1][C2], [C2]
  const unsigned short* wp2h;         // bf16 MFMA image of W2 (dg_train_fwd<C1, true>), no sign folded
  const float *sc1, *sh1, *sc2, *sh2; // [2][C] batch-stat scale / shift (acc -> y)
  const float* gamma2[2];         // BatchNorm gamma of the second edge layer per tower (its sign picks max or min over the slots)
  double* mom;                    // [2B][27]                       (phase 1)
  double* stat_part;              // phase 1: [2B][C1][2]
  float* g1_part;                 // phase 2: [2B][C1*C1] Gram(h1) over the cloud's (point, slot) rows, upper blocks
  float* p_store;                 // [2B*N][C2]                     (phase 3)
  unsigned char* argk;            // [2B*N][C2]
  double* colsum_part;            // [2B][2 halves][C2]   column sums of p
  double* s1_part;                // [2B][1024 / C1][C1]  column sums of h1 over all (point, slot) rows, per row group
  long long* stamps;              // debug (ALIGNNET_DBG & 32): cycle stamps of thread 0 / block 0, iteration 25
  int dbg = 0;                    // ablation (timing only): bit 0 = no Gram(h1) MFMAs in dg_train_fwd
  // parts > 1: a cloud's tiles are dealt to `parts` workgroups (grid 2B * parts, workgroup = cloud * parts + part, tiles [part nt / parts, (part + 1) nt / parts)):
  // one workgroup per cloud leaves three quarters of the chip idle at 128 clouds (B = 64).  Everything a workgroup hands on is a per-workgroup PARTIAL
  // (g1_part, s1_part: indexed by the workgroup, reduced over B * parts slices) or per point (p_store, argk), so nothing else changes.
  int parts = 1;
};
#define FE_STAMP(i) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 0 && it == 25) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ void dgt_gather(const float* __restrict__ pc, const int* __restrict__ nnc, int N, int k, int n, int slot,
                                           float (&v)[6])
{
  const int j = nnc[(size_t)n * k + slot];
  const float* p = pc + (size_t)n * 3;
  const float* pj = pc + (size_t)j * 3;
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  v[3] = pj[0] - p[0]; v[4] = pj[1] - p[1]; v[5] = pj[2] - p[2];
}

// The same gather as two pipeline stages: the neighbour index of edge slot j (tile j / k, slot j % k) one step ahead of the point loads
// that depend on it -- issued back to back, the index round trip (an L2 / HBM latency) stalls the gathering wave, and the whole
// workgroup behind the slot's barrier, once per slot.
__device__ __forceinline__ int dgt_index(const int* __restrict__ nnc, int N, int k, int j, int tid)
{
  const int jt = j / k, js = j - jt * k;
  return nnc[(size_t)min(jt * kTT + tid, N - 1) * k + js];
}
__device__ __forceinline__ void dgt_points(const float* __restrict__ pc, int N, int k, int j, int tid, int nbr, float (&v)[6])
{
  const int n = min((j / k) * kTT + tid, N - 1);
  const float* p = pc + (size_t)n * 3;
  const float* pj = pc + (size_t)nbr * 3;
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  v[3] = pj[0] - p[0]; v[4] = pj[1] - p[1]; v[5] = pj[2] - p[2];
}

// ---------------------------------------------------------------------------------
// phase 1: moments of e per cloud -> statistics of z1.   grid 2B, block kP1T
// (a gather-latency chain per edge row: sixteen waves per cloud instead of four -- at N = 4096, 128 clouds of 256 threads left most of the chip's wave
//  slots empty; the fp64 moments are summed in another grouping, nothing else changes)
// ---------------------------------------------------------------------------------
constexpr int kP1T = 1024;
__global__ __launch_bounds__(kP1T) void dg_train_phase1(const DgTrainArgs a)
{
  __shared__ double red[kP1T / 64][kDgMom];
  __shared__ double tot[kDgMom];
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  double m[kDgMom];
#pragma unroll
  for (int i = 0; i < kDgMom; ++i) m[i] = 0.0;
  const int rows = a.N * a.k;
  for (int r = threadIdx.x; r < rows; r += kP1T) {
    const int n = r / a.k, slot = r - n * a.k;
    float v[6], e[6];
    dgt_gather(pc, nnc, a.N, a.k, n, slot, v);
    dg_edge_to_lds(xf, v, e);
    int q = 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      m[d] += (double)e[d];
#pragma unroll
      for (int d2 = d; d2 < 6; ++d2) m[q++] += (double)e[d] * (double)e[d2];
    }
  }
#pragma unroll
  for (int i = 0; i < kDgMom; ++i) {
    double v = m[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kDgMom) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < kP1T / 64; ++wv) t += red[wv][threadIdx.x];
    tot[threadIdx.x] = t;
    a.mom[(size_t)cloud * kDgMom + threadIdx.x] = t;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C1; c += kP1T) {
    double w[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) w[d] = (double)a.w1[d * a.C1 + c];
    double sw = 0.0, qd = 0.0;
    int q = 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      sw += w[d] * tot[d];
#pragma unroll
      for (int d2 = d; d2 < 6; ++d2) qd += (d == d2 ? 1.0 : 2.0) * w[d] * w[d2] * tot[q++];
    }
    const double bb = (double)a.b1[c], n = (double)rows;
    a.stat_part[((size_t)cloud * a.C1 + c) * 2] = sw + n * bb;
    a.stat_part[((size_t)cloud * a.C1 + c) * 2 + 1] = qd + 2.0 * bb * sw + n * bb * bb;
  }
}

// ---------------------------------------------------------------------------------
// K = 6 lift: out[row][c] = relu((e . w[:,c]) * sc + sh), rows >= nvalid are written as 0.
// ---------------------------------------------------------------------------------
// On the matrix pipe (C1 a multiple of 16): k padded to 8 -- zero weights, and es columns 6 / 7 zeroed once by the
// caller -- and the [64 rows][C1] output cut into 16 x 16 tiles (v_mfma_f32_16x16x4_f32, two instructions per tile), the tiles
// dealt round-robin to the NW waves; a wave's B fragments and its tiles' scale / shift stay in registers.  Per lane and tile: two
// 4-byte LDS reads, two MFMAs, four fma + max + select, four LDS writes -- the VALU form (round 1) cost 16 rows x (two 16-byte reads +
// 8 operations) per lane and ran 3.5x slower whenever the co-resident workgroup streamed MFMAs.
template <int C1, int NW>
struct DgtLiftM {
  static constexpr int kCT = C1 / 16, kTiles = (kTT / 16) * kCT, kPer = kTiles / NW;
  static_assert(C1 % 16 == 0 && kTiles % NW == 0, "lift tiles must divide over the waves");
  float w[kPer][2], sc[kPer], sh[kPer];
};

template <int C1, int NW>
__device__ __forceinline__ DgtLiftM<C1, NW> dgt_liftm_load(const float* __restrict__ w1, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, int wave, int lane)
{
  DgtLiftM<C1, NW> R;
#pragma unroll
  for (int i = 0; i < DgtLiftM<C1, NW>::kPer; ++i) {
    const int t = wave * DgtLiftM<C1, NW>::kPer + i, c = 16 * (t % DgtLiftM<C1, NW>::kCT) + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = 4 * ks + (lane >> 4);
      R.w[i][ks] = kk < 6 ? w1[kk * C1 + c] : 0.f;
    }
    R.sc[i] = sc[c]; R.sh[i] = sh[c];
  }
  return R;
}

template <int C1, int NW>
__device__ __forceinline__ void dgt_liftm(const DgtLiftM<C1, NW>& R, const float* __restrict__ es, float* __restrict__ out, int ldo,
                                          int nvalid, int wave, int lane)
{
#pragma unroll
  for (int i = 0; i < DgtLiftM<C1, NW>::kPer; ++i) {
    const int t = wave * DgtLiftM<C1, NW>::kPer + i, rt = t / DgtLiftM<C1, NW>::kCT, c = 16 * (t % DgtLiftM<C1, NW>::kCT) + (lane & 15);
    const float* ar = es + (16 * rt + (lane & 15)) * 8 + (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], R.w[i][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4], R.w[i][1], acc, 0, 0, 0);
    const int row0 = 16 * rt + 4 * (lane >> 4);
    float* o = out + row0 * ldo + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * ldo] = row0 + r < nvalid ? fmaxf(fmaf(acc[r], R.sc[i], R.sh[i]), 0.f) : 0.f;
  }
}

// bf16 mode of the backward edge pass: the fp32 tile holds the ROUNDED h1 (what the forward multiplied: U2 = h1^T dy2, the relu
// mask and the sparse products read it) and a bf16 row-major copy feeds the h1 Q2 MFMAs
template <int C1, int NW>
__device__ __forceinline__ void dgt_liftm_both(const DgtLiftM<C1, NW>& R, const float* __restrict__ es, float* __restrict__ out, int ldo,
                                               unsigned short* __restrict__ Xh, int ldh, int nvalid, int wave, int lane)
{
#pragma unroll
  for (int i = 0; i < DgtLiftM<C1, NW>::kPer; ++i) {
    const int t = wave * DgtLiftM<C1, NW>::kPer + i, rt = t / DgtLiftM<C1, NW>::kCT, c = 16 * (t % DgtLiftM<C1, NW>::kCT) + (lane & 15);
    const float* ar = es + (16 * rt + (lane & 15)) * 8 + (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], R.w[i][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4], R.w[i][1], acc, 0, 0, 0);
    const int row0 = 16 * rt + 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned short hb = row0 + r < nvalid ? to_bf16_bits(fmaxf(fmaf(acc[r], R.sc[i], R.sh[i]), 0.f)) : (unsigned short)0;
      out[(row0 + r) * ldo + c] = __uint_as_float((unsigned)hb << 16);
      Xh[(row0 + r) * ldh + c] = hb;
    }
  }
}

// bf16 mode: the same lift written straight as the two bf16 tiles the bf16 MFMAs read -- row-major Xh[row][ldh] (A operand of
// z2 = h1 W2) and transposed XhT[c][ldT] (both operands of Gram(h1) = h1^T h1); a lane's four rows of a column are one 8-byte store
// into the transposed tile.
template <int C1, int NW>
__device__ __forceinline__ void dgt_liftm_bf16(const DgtLiftM<C1, NW>& R, const float* __restrict__ es, unsigned short* __restrict__ Xh, int ldh,
                                               unsigned short* __restrict__ XhT, int ldT, int nvalid, int wave, int lane)
{
#pragma unroll
  for (int i = 0; i < DgtLiftM<C1, NW>::kPer; ++i) {
    const int t = wave * DgtLiftM<C1, NW>::kPer + i, rt = t / DgtLiftM<C1, NW>::kCT, c = 16 * (t % DgtLiftM<C1, NW>::kCT) + (lane & 15);
    const float* ar = es + (16 * rt + (lane & 15)) * 8 + (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], R.w[i][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4], R.w[i][1], acc, 0, 0, 0);
    const int row0 = 16 * rt + 4 * (lane >> 4);
    unsigned short hb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hb[r] = row0 + r < nvalid ? to_bf16_bits(fmaxf(fmaf(acc[r], R.sc[i], R.sh[i]), 0.f)) : (unsigned short)0;
      Xh[(row0 + r) * ldh + c] = hb[r];
    }
    uint2 pk;
    pk.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16); pk.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
    *reinterpret_cast<uint2*>(XhT + c * ldT + row0) = pk;
  }
}

// The same lift, also handing back the wave's own h1 values (bf16 bits, tile i, accumulator element r): the relu mask [h1 > 0] of the
// dense edge backward, whose dh1 tiles are the wave's lift tiles -- it never reads h1 back for the mask.
template <int C1, int NW>
__device__ __forceinline__ void dgt_liftm_bf16_keep(const DgtLiftM<C1, NW>& R, const float* __restrict__ es, unsigned short* __restrict__ Xh, int ldh,
                                                    unsigned short* __restrict__ XhT, int ldT, int nvalid, int wave, int lane,
                                                    unsigned short (&hv)[DgtLiftM<C1, NW>::kPer][4])
{
#pragma unroll
  for (int i = 0; i < DgtLiftM<C1, NW>::kPer; ++i) {
    const int t = wave * DgtLiftM<C1, NW>::kPer + i, rt = t / DgtLiftM<C1, NW>::kCT, c = 16 * (t % DgtLiftM<C1, NW>::kCT) + (lane & 15);
    const float* ar = es + (16 * rt + (lane & 15)) * 8 + (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], R.w[i][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4], R.w[i][1], acc, 0, 0, 0);
    const int row0 = 16 * rt + 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[i][r] = row0 + r < nvalid ? to_bf16_bits(fmaxf(fmaf(acc[r], R.sc[i], R.sh[i]), 0.f)) : (unsigned short)0;
      Xh[(row0 + r) * ldh + c] = hv[i][r];
    }
    uint2 pk;
    pk.x = (unsigned)hv[i][0] | ((unsigned)hv[i][1] << 16); pk.y = (unsigned)hv[i][2] | ((unsigned)hv[i][3] << 16);
    *reinterpret_cast<uint2*>(XhT + c * ldT + row0) = pk;
  }
}

// ---------------------------------------------------------------------------------
// phase 2: one workgroup (4 waves) per cloud walks (tile, slot); wave w owns channel tile w of the C2 <= 128
// edge-conv outputs and both 32-row groups.  LDS: es [64][8] | X0 [64][ld0] | X1 [64][ld0] (lift double-buffered;
// the gather for the next slot is in flight during this slot's MFMAs).
// ---------------------------------------------------------------------------------
// C1 is a template parameter: with a compile-time LDS row stride the per-row tile addresses are immediate offsets; as
// run-time values the compiler hoists them out of the slot loop into ~50 registers and spills.
// BF16 ("train_matmul_bf16" with the dgcnn backbone): the edge conv behind the lift, z2 = h1 W2, and Gram(h1) run on
// v_mfma_f32_32x32x16_bf16 with h1 and W2 rounded to bf16 (fp32 accumulate); the lift writes bf16 tiles, the column sums are those of
// the rounded h1, and the statistics of z2 follow from the Gram of the rounded h1 with the rounded W2 (stat2_from_gram_kernel).
// LDS: es [64][8] fp32 | per buffer Xh [64][C1 + 8] | XhT [C1][64 + 8] bf16.  Everything behind the accumulators is unchanged.
template <int C1, bool BF16 = false>
__global__ __launch_bounds__(kTW * 64, 2) void dg_train_fwd(const DgTrainArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const long long t_begin = ALN_STAMPS(a.stamps) ? (long long)__builtin_readcyclecounter() : 0;   // (debug: whole-kernel cycles per workgroup)
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  constexpr int ld0 = C1 + 4;
  constexpr int KG2 = (C1 + 7) >> 3;   // <= 8 (C1 <= 64)
  constexpr int ldh = C1 + 8, ldT = kTT + 8, KG16 = C1 / 16;   // bf16 tiles
  constexpr int kBufH = kTT * ldh + C1 * ldT;                    // bf16 elements per lift buffer (Xh | XhT)
  float* const xbuf = smem + 2 * kTT * 8;   // behind the two edge-feature buffers
  unsigned short* hbuf = reinterpret_cast<unsigned short*>(xbuf);
  const int CT2 = (a.C2 + 31) >> 5;
  const int ntiles = (a.N + kTT - 1) / kTT;
  const int it0 = (part * ntiles / a.parts) * a.k, total = ((part + 1) * ntiles / a.parts) * a.k;   // this workgroup's (tile, slot) range [it0, total); it0 is even (k = 20)
  const int ct = wave, col = ct * 32 + (lane & 31);
  const bool mine = ct < CT2, live = mine && col < a.C2;
  // max_k relu(g zhat + b) = relu(g zhat* + b) with zhat* the extreme of sign(g) z over the slots: the sign of gamma2 is known
  // before the statistics are, so this one pass records the extreme (and its slot) next to the sums -- no second edge-conv pass
  const float sgn = (live && a.gamma2[tower][col] < 0.f) ? -1.f : 1.f;
  const float* sc1 = a.sc1 + tower * C1;
  const float* sh1 = a.sh1 + tower * C1;
  const DgtLiftM<C1, kTW> lw = dgt_liftm_load<C1, kTW>(a.w1, sc1, sh1, wave, lane);
  // the wave's W2 fragments stay in registers for the whole cloud (C1 <= 64: 8 k-groups), with the column's sign(gamma2)
  // folded in (exact): the accumulator is sgn * (z2 - bias), so the extreme over the slots is a plain max and the sums are
  // those of sgn * (z2 - bias), put right at the end
  f32x4 breg[BF16 ? 1 : KG2];
  bf16x8 bregh[BF16 ? KG16 : 1];
  if (mine) {
    if constexpr (BF16) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const unsigned flip = sgn < 0.f ? 0x80008000u : 0u;   // a lane's eight k values belong to one column: its sign, folded bit-wise
#pragma unroll
      for (int kg = 0; kg < KG16; ++kg) {
        u32x4 v = reinterpret_cast<const u32x4*>(a.wp2h)[((size_t)ct * KG16 + kg) * 64 + lane];
        v[0] ^= flip; v[1] ^= flip; v[2] ^= flip; v[3] ^= flip;
        bregh[kg] = __builtin_bit_cast(bf16x8, v);
      }
    } else {
#pragma unroll
      for (int kg = 0; kg < KG2; ++kg) {
        breg[kg] = reinterpret_cast<const f32x4*>(a.wp2)[((size_t)ct * KG2 + kg) * 64 + lane];
        breg[kg][0] *= sgn; breg[kg][1] *= sgn; breg[kg][2] *= sgn; breg[kg][3] *= sgn;
      }
    }
  }
  double s1q[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int CT1 = (C1 + 31) >> 5, nG = CT1 * (CT1 + 1) / 2;   // <= 3 blocks (C1 <= 64) <= 4 waves
  int git = 0, gjt = 0;
  {
    int rem = min(wave, nG - 1);
    while (rem >= CT1 - git) { rem -= CT1 - git; ++git; }
    gjt = git + rem;
  }
  f32x16 gacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
  // The cloud's Gram blocks in fp64, in LDS (one [16][64] slab per owning wave, touched by that wave only: no barrier): the fp32 MFMA
  // accumulators are folded in at the end of every tile (k slots x 64 rows).  Carried in fp32 for the whole cloud -- N k = 82 k rows at
  // N = 4096 -- each element collects ~40 k roundings at the magnitude of the running sum, and the statistics of z2 derived from the Gram
  // (stat2_from_gram_kernel: w^T G w / M - mean^2) divide by variances that can be 1e-3 of its entries: at N = 4096 the step sat 8x further
  // from the fp64 oracle than a plain fp32 evaluation of the graph (profiles/r06_relu_pin_diag_dg.log).
  double* const gsum = reinterpret_cast<double*>(BF16 ? reinterpret_cast<float*>(hbuf + 2 * kBufH) : xbuf + 2 * kTT * ld0) + (size_t)wave * 16 * 64 + lane;
  if (wave < nG) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gsum[r * 64] = 0.0;
  }
  f32x16 best[2];
  int bk[2][16];

  // One barrier per slot: at iteration it the waves lift slot it + 1 (es[(it + 1) & 1] -> h buffer (it + 1) & 1) and wave 0 writes the
  // edge features of slot it + 2 into es[it & 1] (last read by the lift of slot it, an iteration and a barrier ago) while the MFMAs
  // of slot it read h buffer it & 1; the points of slot it + 3 and the neighbour index of slot it + 4 are in flight meanwhile.
  // (Two barriers per slot -- features, barrier, lift, barrier -- left the matrix pipe idle during every lift.)
  float v[6];
  int jnext = 0;
  auto es_of = [&](int j) { return smem + (j & 1) * kTT * 8; };
  auto lift_slot = [&](int j) {
    const int nv = min(kTT, a.N - (j / a.k) * kTT);
    if constexpr (BF16) {
      unsigned short* nh = hbuf + (j & 1) * kBufH;
      dgt_liftm_bf16<C1, kTW>(lw, es_of(j), nh, ldh, nh + kTT * ldh, ldT, nv, wave, lane);
    } else dgt_liftm<C1, kTW>(lw, es_of(j), xbuf + (j & 1) * kTT * ld0, ld0, nv, wave, lane);
  };
  if (tid < kTT) {
    smem[tid * 8 + 6] = 0.f; smem[tid * 8 + 7] = 0.f;   // k padding of the MFMA lift in both buffers: never written again
    smem[kTT * 8 + tid * 8 + 6] = 0.f; smem[kTT * 8 + tid * 8 + 7] = 0.f;
    dgt_points(pc, a.N, a.k, it0, tid, dgt_index(nnc, a.N, a.k, it0, tid), v);
    dg_edge_to_lds(xf, v, es_of(it0) + tid * 8);
    if (total > it0 + 1) {
      dgt_points(pc, a.N, a.k, it0 + 1, tid, dgt_index(nnc, a.N, a.k, it0 + 1, tid), v);
      dg_edge_to_lds(xf, v, es_of(it0 + 1) + tid * 8);
    }
    if (total > it0 + 2) dgt_points(pc, a.N, a.k, it0 + 2, tid, dgt_index(nnc, a.N, a.k, it0 + 2, tid), v);
    if (total > it0 + 3) jnext = dgt_index(nnc, a.N, a.k, it0 + 3, tid);
  }
  __syncthreads();
  lift_slot(it0);
  __syncthreads();
  for (int it = it0; it < total; ++it) {
    const int tile = it / a.k, slot = it - tile * a.k;
    const int nvalid = min(kTT, a.N - tile * kTT);
    FE_STAMP(0);
    if (it + 1 < total) lift_slot(it + 1);
    if (it + 2 < total && tid < kTT) {
      dg_edge_to_lds(xf, v, es_of(it) + tid * 8);   // slot it + 2
      if (it + 3 < total) dgt_points(pc, a.N, a.k, it + 3, tid, jnext, v);
      if (it + 4 < total) jnext = dgt_index(nnc, a.N, a.k, it + 4, tid);
    }
    const float* X = xbuf + (it & 1) * kTT * ld0;
    const unsigned short* Xh = hbuf + (it & 1) * kBufH;
    const unsigned short* XhT = Xh + kTT * ldh;
    if (slot == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { best[m][r] = -INFINITY; bk[m][r] = 0; }
    }
    if (mine) {
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      if constexpr (BF16) {
        const unsigned short* arow = Xh + (lane & 31) * ldh + half * 8;
#pragma unroll
        for (int kg = 0; kg < KG16; ++kg) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + kg * 16), bregh[kg], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + 32 * ldh + kg * 16), bregh[kg], acc[1], 0, 0, 0);
        }
      } else {
      const float* arow = X + (lane & 31) * ld0 + half * 4;
#pragma unroll
      for (int kg = 0; kg < KG2; ++kg) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + kg * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + 32 * ld0 + kg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], breg[kg][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], breg[kg][s], acc[1], 0, 0, 0);
        }
      }
      }
      FE_STAMP(1);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool up = acc[m][r] > best[m][r];   // first slot wins ties
          bk[m][r] = up ? slot : bk[m][r];
          best[m][r] = fmaxf(best[m][r], acc[m][r]);
        }
      if (slot == a.k - 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = acc_row(m, r, lane);
            if (row < nvalid && live) {
              const size_t o = ((size_t)cloud * a.N + (size_t)tile * kTT + row) * a.C2 + col;
              a.p_store[o] = best[m][r] * sgn;   // the accumulator (z2 - bias) at the arg-extreme slot; dg_pool_finish turns it into p
              a.argk[o] = (unsigned char)bk[m][r];
            }
          }
      }
    }
    // Gram(h1) += h1_s^T h1_s (upper 32 x 32 blocks, one per wave): with the column sums below it gives the statistics of
    // z2 = h1 W2 + b2 (linear in h1: stat2_from_gram_kernel) and the layer-2 weight gradient of the backward -- the 32 sums
    // per lane and slot this replaces were a third of the kernel's VALU work, and the kernel is bound by that, not by the matrix pipe
    if (wave < nG && !(ALN_ABL(a.dbg, 1))) {
      if constexpr (BF16) {
        const unsigned short* pa = XhT + (git * 32 + (lane & 31)) * ldT + half * 8;
        const unsigned short* pb = XhT + (gjt * 32 + (lane & 31)) * ldT + half * 8;
#pragma unroll
        for (int kk = 0; kk < kTT / 16; ++kk)
          gacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pa + kk * 16), *reinterpret_cast<const bf16x8*>(pb + kk * 16), gacc, 0, 0, 0);
      } else {
      const float* pa = X + half * ld0 + git * 32 + (lane & 31);
      const float* pb = X + half * ld0 + gjt * 32 + (lane & 31);
#pragma unroll 8
      for (int r = 0; r < kTT; r += 2) gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld0], pb[r * ld0], gacc, 0, 0, 0);
      }
      if (slot == a.k - 1) {   // end of the tile: fp32 block -> the cloud's fp64 block
#pragma unroll
        for (int r = 0; r < 16; ++r) { gsum[r * 64] += (double)gacc[r]; gacc[r] = 0.f; }
      }
    }
    FE_STAMP(2);
    {   // column sums of h1 (rows past nvalid are zero): thread = (4 columns, one of kQG row groups), float4 reads
      constexpr int kQ = C1 / 4, kQG = (kTW * 64) / kQ;   // C1 = 64: 16 column quads x 16 row groups
      const int cq = tid % kQ, g = tid / kQ;
      f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < kTT / kQG; ++r) {
        if constexpr (BF16) {
          const uint2 hv = *reinterpret_cast<const uint2*>(Xh + (g + r * kQG) * ldh + cq * 4);
          sm[0] += __uint_as_float(hv.x << 16); sm[1] += __uint_as_float(hv.x & 0xffff0000u);
          sm[2] += __uint_as_float(hv.y << 16); sm[3] += __uint_as_float(hv.y & 0xffff0000u);
        } else {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(X + (g + r * kQG) * ld0 + cq * 4);
          sm[0] += hv[0]; sm[1] += hv[1]; sm[2] += hv[2]; sm[3] += hv[3];
        }
      }
      s1q[0] += (double)sm[0]; s1q[1] += (double)sm[1]; s1q[2] += (double)sm[2]; s1q[3] += (double)sm[3];
    }
    FE_STAMP(3);
    __syncthreads();   // the slot's only barrier
    FE_STAMP(7);
  }
  if (wave < nG) {
    const float zero[16] = {};
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[r] = (float)(gsum[r * 64] + (double)gacc[r]);   // (gacc is zero unless the ablation switch skipped the folds)
    tile_commit(a.g1_part + (size_t)vcloud * C1 * C1, C1, git, gjt, C1, C1, gacc, lane, zero);
  }
  if (ALN_STAMPS(a.stamps) && tid == 0) {   // stamps 8 / 9 / 10: longest and (2^40 - shortest) workgroup, workgroup 0 -- zeroed by the host
    const long long d = (long long)__builtin_readcyclecounter() - t_begin;
    atomicMax(reinterpret_cast<unsigned long long*>(a.stamps + 8), (unsigned long long)d);
    atomicMax(reinterpret_cast<unsigned long long*>(a.stamps + 9), (unsigned long long)((1ll << 40) - d));
    if (blockIdx.x == 0) a.stamps[10] = d;
  }
  {   // s1_part [cloud][kQG row groups][C1]
    constexpr int kQ = C1 / 4, kQG = (kTW * 64) / kQ;
    const int cq = tid % kQ, g = tid / kQ;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.s1_part[((size_t)vcloud * kQG + g) * C1 + cq * 4 + e] = s1q[e];
  }
}

// p = relu(scale * acc* + shift) in place over the cloud's [N][C2] rows, and the column sums of p.  grid 2B * parts, block 256: workgroup = cloud * parts + part
// streams rows [N part / parts, N (part + 1) / parts) and leaves its own column-sum slice (a pure stream: 4 MB per cloud at N = 4096 -- one workgroup per cloud is
// 128 of them on 256 CUs at B = 64)
__global__ __launch_bounds__(256) void dg_pool_finish(float* __restrict__ p, int B, int N, int C2, const float* __restrict__ sc2,
                                                      const float* __restrict__ sh2, double* __restrict__ colsum_part, int parts)
{
  __shared__ double red[4][256];   // [e][thread]: consecutive lanes on consecutive banks (as [thread][4] every lane of a store sat on the same eight banks: conflict share 0.56)
  const int vcloud = blockIdx.x, cloud = vcloud / parts, part = vcloud - cloud * parts, tower = cloud >= B, tid = threadIdx.x;
  const int c4 = C2 >> 2, G = 256 / c4, q = tid % c4, g = tid / c4;
  float* base = p + (size_t)cloud * N * C2;
  const int rbeg = (int)((long)N * part / parts), rend = (int)((long)N * (part + 1) / parts);
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (g < G) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(sc2 + tower * C2 + q * 4), sh = *reinterpret_cast<const f32x4*>(sh2 + tower * C2 + q * 4);
    for (int r0 = rbeg + g; r0 < rend; r0 += G * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * G;
        v[u] = r < rend ? *reinterpret_cast<const f32x4*>(base + (size_t)r * C2 + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * G;
        if (r < rend) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = fmaxf(fmaf(v[u][e], sc[e], sh[e]), 0.f); ps[e] += o[e]; }
          *reinterpret_cast<f32x4*>(base + (size_t)r * C2 + q * 4) = o;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += (double)ps[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[e][tid] = g < G ? s[e] : 0.0;
  __syncthreads();
  if (tid < c4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double t = 0.0;
      for (int gg = 0; gg < G; ++gg) t += red[e][gg * c4 + tid];
      colsum_part[((size_t)vcloud * 2) * C2 + tid * 4 + e] = t;
      colsum_part[((size_t)vcloud * 2 + 1) * C2 + tid * 4 + e] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------
// backward edge pass (see the header comment).  One workgroup of 8 waves per cloud walks (tile, slot).
// dy2_s = dp [argk == s] has one non-zero per (point, channel) over the k slots, so its two products are done sparsely
// on the VALU from per-tile index lists (built once per tile with wave ballots, deterministic order):
//   dh1_s[row,:] += sum over the row's slot-s columns c of dp[row,c] V2[c,:]          (row lists)
//   U2[:,c]      += sum over the column's slot-s rows of dp[row,c] h1_s[row,:]        (column lists)
// -- 1/k of the dense MFMA work each.  What stays on MFMA is h1_s Q2 and Gram(h1).
// LDS: es [2][64][8] | X = h1_s [64][ld0] | D = sparse part of dh1_s [64][ld0] (arg-k staging at tile start) |
//      DP [64][C2+4] | V2 [C2][C1+4] | row lists [64][C2] + offsets [64][24] | column lists [C2][64] + offsets [C2][24]  (bytes)
// ---------------------------------------------------------------------------------
constexpr int kBEW = 8;

template <int I, int N, class F>
__device__ __forceinline__ void dg_static_for(F&& f)   // f(integral_constant<int, I>) for I in [I, N): loop indices usable as asm immediates
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); dg_static_for<I + 1, N>(f); }
}

struct DgBwdArgs {
  const float* pcs[2]; const float* xform; const int* nn; int B, N, k, C1, C2;
  int ld0;
  const float* w1; const float *sc1, *sh1;
  const float* v2; long v2_stride;     // per tower: V2 = (W2 diag(k2))^T  [C2][C1] row-major
  const float* q2img; long q2img_stride;   // per-tower MFMA image of Q2 [C1][C1]
  const unsigned short* q2imgh; long q2imgh_stride;   // bf16 mode: per-tower bf16 MFMA image of Q2
  const float* q2b;                    // [2][C1]
  const float* k2 = nullptr;           // [2][C2] gamma2 rstd2 (dense form: folded into the staged dp)
  const unsigned short* w2th = nullptr;   // dense form: bf16 MFMA image of round(W2)^T (K = C2, C = C1), shared by the towers
  const float* dyp;                    // [2B*N][C2]  dp * [p > 0]   (pass B2, GIVEN)
  const unsigned char* argk;           // [2B*N][C2]
  float* u2_part; float* g1_part;      // [2B][C1*C2], [2B][C1*C1] (upper blocks)
  double* pdy_part;                    // [2B][4 = 2 row groups x 2 halves][7][C1]: sum e_d dy1 (d < 6), sum dy1
  long long* stamps;                   // debug: cycle stamps of thread 0 / block 0 at the phase boundaries of iteration 25
  int parts = 1;                       // as DgTrainArgs::parts: u2_part / pdy_part are per-workgroup partials ([2B * parts] slices)
};
#define BE_STAMP(i) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 0 && it == 25) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
// stamps 10..15: the tile-start block of iteration 40; stamp 16: iteration 45 (20 iterations = one tile after stamp 0)
#define BE_TSTAMP(i, at) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 0 && it == at) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

static inline size_t dg_bwd_edge_lds(int C1, int C2, bool bf16 = false)
{
  const int ld0 = ((C1 + 7) & ~7) + 4;
  return ((size_t)2 * kTT * 8 + 2 * (size_t)kTT * ld0 + (size_t)kTT * (C2 + 4) + (size_t)C2 * (C1 + 4)) * sizeof(float) +
         (size_t)kTT * (C2 + 4) + kTT * 24 + (size_t)C2 * (kTT + 4) + (size_t)C2 * 24 + (bf16 ? (size_t)kTT * (C1 + 8) * sizeof(unsigned short) : 0);
}

// BF16 ("train_matmul_bf16", dgcnn): h1 is the ROUNDED h1 of the forward (fp32 tile of rounded values + a bf16 copy) and the dense
// product h1_s Q2 runs on v_mfma_f32_32x32x16_bf16 (Q2 as a bf16 image); the sparse products, Pdy and all reductions stay fp32.
template <int C1, int C2, bool BF16 = false>
__global__ __launch_bounds__(kBEW * 64) void dg_train_bwd_edge(const DgBwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  constexpr int ld0 = C1 + 4, ldv = C1 + 4;
  constexpr int ldp = C2 + 4;
  float* X = smem + 2 * kTT * 8;
  float* D = X + kTT * ld0;
  float* DP = D + kTT * ld0;
  float* V2 = DP + kTT * ldp;
  // byte tiles with row strides of 4 bytes more than a power of two: lanes that walk a COLUMN of them (the P2 team reads SLc[c][j] for 64
  // consecutive c, the column-list build reads AK[row][c] for 64 rows) then fall into 64 different banks instead of 4 (stride 64 B:
  // 16-way conflicts) or 2 (stride 128 B: 32-way) -- a third of this kernel's LDS cycles were bank conflicts (profiles/r02_train_dgcnn_pmc)
  constexpr int ldsl = C2 + 4, ldsc = kTT + 4, ldak = C2 + 4;
  unsigned char* SL = reinterpret_cast<unsigned char*>(V2 + C2 * ldv);   // [64][ldsl]  columns of the row, grouped by slot
  unsigned char* SO = SL + kTT * ldsl;                                   // [64][24]  group offsets
  unsigned char* SLc = SO + kTT * 24;                                      // [C2][ldsc]  rows of the column, grouped by slot
  unsigned char* SOc = SLc + C2 * ldsc;                                  // [C2][24]
  unsigned char* AK = reinterpret_cast<unsigned char*>(D);                 // [64][ldak] staging (tile start only)
  constexpr int ldh = C1 + 8, KG16 = C1 / 16;
  unsigned short* Xh = reinterpret_cast<unsigned short*>(SOc + C2 * 24);   // bf16 mode: h1 [64][C1 + 8]  (all sizes above are multiples of 16 bytes)
  constexpr int CT1 = (C1 + 31) >> 5, KGq = (C1 + 7) >> 3;
  const int ntiles = (a.N + kTT - 1) / kTT;
  const int it0 = (part * ntiles / a.parts) * a.k, total = ((part + 1) * ntiles / a.parts) * a.k;   // this workgroup's (tile, slot) range; it0 is even (k = 20)
  const f32x4* q2img = reinterpret_cast<const f32x4*>(a.q2img + tower * a.q2img_stride);
  const float* sc1 = a.sc1 + tower * C1;
  const float* sh1 = a.sh1 + tower * C1;
  const DgtLiftM<C1, kBEW> lw = dgt_liftm_load<C1, kBEW>(a.w1, sc1, sh1, wave, lane);
  if (tid < 2 * kTT) { smem[tid * 8 + 6] = 0.f; smem[tid * 8 + 7] = 0.f; }   // k padding of the MFMA lift in both es buffers: never written again
  // roles: waves [0, nitems) own one dh1 item (channel tile, 32-row group), nitems <= 4; waves 4..7 the sparse U2 units (P2)
  constexpr int nitems = CT1 * 2;
  // the dh1 waves keep their Q2 fragments in registers for the whole cloud (C1 <= 64: 8 k-groups); streaming them per slot
  // put eight dependent L2 round trips in front of every slot's MFMAs
  f32x4 qreg[BF16 ? 1 : 8];
  bf16x8 qregh[BF16 ? KG16 : 1];
  if (wave < nitems) {
    if constexpr (BF16) {
      const bf16x8* qh = reinterpret_cast<const bf16x8*>(a.q2imgh + tower * a.q2imgh_stride);
#pragma unroll
      for (int kg = 0; kg < KG16; ++kg) qregh[kg] = qh[((size_t)(wave >> 1) * KG16 + kg) * 64 + lane];
    } else {
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) qreg[kg] = q2img[((size_t)(wave >> 1) * KGq + min(kg, KGq - 1)) * 64 + lane];
    }
  }
  double pd[4];
  f32x16 pacc;
#pragma unroll
  for (int d = 0; d < 4; ++d) pd[d] = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) pacc[q] = 0.f;
  // sparse units: P1 = (row, 8-channel chunk of C1), P2 = (column of C2, 16-channel chunk of C1)
  constexpr int p1n = C1 >> 3;
  const int p1row = tid / p1n, p1ch = tid - p1row * p1n;
  const bool p1on = tid < kTT * p1n;
  // P2 is the job of waves 4..7 alone, next to the dh1 items of waves 0..3 (the Gram blocks those waves once held are the
  // forward's now): 256 threads, so the shipped shape walks a column's list once for a 32-channel chunk
  constexpr int kP2W = (C1 * C2 >= 256 * 32) ? 32 : 16, kP2Q = kP2W / 4;
  constexpr int p2n = C1 / kP2W;
  const int p2t = tid - 4 * 64;
  const int p2c = (p2t & 0x7fffffff) % C2, p2j = (p2t & 0x7fffffff) / C2;
  const bool p2on = p2t >= 0 && p2t < C2 * p2n;
  float u2[kP2W];
#pragma unroll
  for (int i = 0; i < kP2W; ++i) u2[i] = 0.f;

  {   // V2 of this tower -> LDS, once per cloud
    const float* src = a.v2 + tower * a.v2_stride;
    const int c4 = C1 >> 2;
    for (int i = tid; i < C2 * c4; i += kBEW * 64) {
      const int r = i / c4, q = i % c4;
      *reinterpret_cast<f32x4*>(V2 + r * ldv + q * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)r * C1 + q * 4);
    }
  }
  float v[6];
  int jnext = 0;
  if (tid < kTT) {
    dgt_points(pc, a.N, a.k, it0, tid, dgt_index(nnc, a.N, a.k, it0, tid), v);
    if (total > it0 + 1) jnext = dgt_index(nnc, a.N, a.k, it0 + 1, tid);
  }
  for (int it = it0; it < total; ++it) {
    const int tile = it / a.k, slot = it - tile * a.k;
    const int nvalid = min(kTT, a.N - tile * kTT);
    const bool more = it + 1 < total;
    float* es = smem + (it & 1) * kTT * 8;
    if (slot == 0) {
      BE_TSTAMP(10, 40);
      __syncthreads();   // the previous tile's readers of D / DP / the lists are done
      BE_TSTAMP(11, 40);
      const size_t base = ((size_t)cloud * a.N + (size_t)tile * kTT) * C2;
      const int c4 = C2 >> 2;
      for (int i = tid; i < kTT * c4; i += kBEW * 64) {
        const int row = i / c4, q = i % c4;
        const bool ok = row < nvalid;
        *reinterpret_cast<f32x4*>(DP + row * ldp + q * 4) =
            ok ? *reinterpret_cast<const f32x4*>(a.dyp + base + (size_t)row * C2 + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<unsigned*>(AK + row * ldak + q * 4) =
            ok ? *reinterpret_cast<const unsigned*>(a.argk + base + (size_t)row * C2 + q * 4) : 0xffffffffu;
      }
      BE_TSTAMP(12, 40);
      __syncthreads();
      BE_TSTAMP(13, 40);
      // stable counting sort by slot without a per-lane select per slot: one compare per slot gives the wave mask of the slot's
      // entries; the mask and the running offset are parked in lane `slot` of a register (v_writelane), and after the loop every
      // lane fetches the mask / offset of ITS slot with ds_bpermute and ranks itself in it (mbcnt).  One byte store per entry.
      // (As a run-time loop with a guarded store per slot this took 162 k cycles per tile -- 45 % of the kernel; with per-slot
      // selects, which hipcc turns into three exec-masked regions per slot, 76 k.)
      // (No writelane builtin in this hipcc, hence asm: one scalar data operand per instruction, the lane select an immediate --
      // hence the compile-time slot loop.  gfx950 needs two wait states between a VALU write of an SGPR (the compare) and a VALU
      // read of it; hipcc inserts them for its own instructions but does not look inside an asm block: without the s_nop the
      // writelane reads the previous compare's mask.)
#define DG_PARK3(o, l, h, so, sl, sh, ln) \
  asm("s_nop 1\n\tv_writelane_b32 %0, %3, %6\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6" \
      : "+v"(o), "+v"(l), "+v"(h) : "s"(so), "s"(sl), "s"(sh), "n"(ln))
#define DG_PARK2(l, h, sl, sh, ln) \
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(l), "+v"(h) : "s"(sl), "s"(sh), "n"(ln))
#define DG_PARK1(o, so, ln) asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(o) : "s"(so), "n"(ln))
      auto rank_in = [](int lo, int hi) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)hi, __builtin_amdgcn_mbcnt_lo((unsigned)lo, 0u)); };
      auto from_lane = [](int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); };
      for (int row = wave; row < kTT; row += kBEW) {   // row lists: lane <-> columns lane, lane + 64
        const int c0 = lane, c1 = lane + 64;
        const float d0 = DP[row * ldp + c0];
        const int k0 = AK[row * ldak + c0];
        const int a0 = d0 != 0.f ? k0 : 255;
        int a1 = 255;
        if constexpr (C2 > 64) {
          const float d1 = DP[row * ldp + c1];
          const int k1 = AK[row * ldak + c1];
          a1 = d1 != 0.f ? k1 : 255;
        }
        int pos = 0, offv = 0, l0 = 0, h0 = 0, l1 = 0, h1 = 0;
        dg_static_for<0, kDgK>([&](auto S) {
          constexpr int s = decltype(S)::value;
          const unsigned long long m0 = __ballot(a0 == s), m1 = __ballot(a1 == s);
          DG_PARK3(offv, l0, h0, pos, (int)(unsigned)m0, (int)(unsigned)(m0 >> 32), s);
          if constexpr (C2 > 64) DG_PARK2(l1, h1, (int)(unsigned)m1, (int)(unsigned)(m1 >> 32), s);
          pos += __popcll(m0) + __popcll(m1);
        });
        DG_PARK1(offv, pos, kDgK);
        if (lane <= kDgK) SO[row * 24 + lane] = (unsigned char)offv;
        const int s0 = a0 & 31, s1 = a1 & 31;   // (255 -> lane 31: fetched, not used)
        const int p0 = from_lane(s0, offv) + rank_in(from_lane(s0, l0), from_lane(s0, h0));
        if (a0 < kDgK) SL[row * ldsl + p0] = (unsigned char)c0;
        if constexpr (C2 > 64) {
          const int n0 = __popc(l0) + __popc(h0);   // lane s: the slot's entries among the first 64 columns
          const int p1 = from_lane(s1, offv) + from_lane(s1, n0) + rank_in(from_lane(s1, l1), from_lane(s1, h1));
          if (a1 < kDgK) SL[row * ldsl + p1] = (unsigned char)c1;
        }
      }
      BE_TSTAMP(14, 40);
      for (int c = wave; c < C2; c += kBEW) {        // column lists: lane <-> row
        const float dv = DP[lane * ldp + c];
        const int kv = AK[lane * ldak + c];
        const int av = dv != 0.f ? kv : 255;
        int pos = 0, offv = 0, l0 = 0, h0 = 0;
        dg_static_for<0, kDgK>([&](auto S) {
          constexpr int s = decltype(S)::value;
          const unsigned long long m = __ballot(av == s);
          DG_PARK3(offv, l0, h0, pos, (int)(unsigned)m, (int)(unsigned)(m >> 32), s);
          pos += __popcll(m);
        });
        DG_PARK1(offv, pos, kDgK);
        if (lane <= kDgK) SOc[c * 24 + lane] = (unsigned char)offv;
        const int sv = av & 31;
        const int pp = from_lane(sv, offv) + rank_in(from_lane(sv, l0), from_lane(sv, h0));
        if (av < kDgK) SLc[c * ldsc + pp] = (unsigned char)lane;
      }
#undef DG_PARK3
#undef DG_PARK2
#undef DG_PARK1
      BE_TSTAMP(15, 40);
    }
    BE_STAMP(0);
    BE_TSTAMP(16, 45);
    if (tid < kTT) dg_edge_to_lds(xf, v, es + tid * 8);
    BE_STAMP(1);
    __syncthreads();   // es and the lists are ready; every wave is done with the previous slot's X / D (and the arg-k staging)
    if (more && tid < kTT) {
      dgt_points(pc, a.N, a.k, it + 1, tid, jnext, v);
      if (it + 2 < total) jnext = dgt_index(nnc, a.N, a.k, it + 2, tid);
    }
    BE_STAMP(2);
    if constexpr (BF16) dgt_liftm_both<C1, kBEW>(lw, es, X, ld0, Xh, ldh, nvalid, wave, lane);
    else dgt_liftm<C1, kBEW>(lw, es, X, ld0, nvalid, wave, lane);
    BE_STAMP(3);
    if (p1on) {   // D[row][8 ch ..] = sum_c dp[row,c] V2[c][8 ch ..] over the row's slot columns
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      const int j0 = SO[p1row * 24 + slot], j1 = SO[p1row * 24 + slot + 1];
      // four list entries per trip: index -> dp -> V2 row is a chain of three dependent LDS reads, the trip count is data
      // dependent and the slowest lane of the workgroup sets the barrier -- so the chains of four entries run side by side
      for (int j = j0; j < j1; j += 4) {
        int cc[4]; float g[4]; f32x4 w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cc[u] = SL[p1row * ldsl + min(j + u, j1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = j + u < j1 ? DP[p1row * ldp + cc[u]] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          w0[u] = *reinterpret_cast<const f32x4*>(V2 + cc[u] * ldv + p1ch * 8);
          w1[u] = *reinterpret_cast<const f32x4*>(V2 + cc[u] * ldv + p1ch * 8 + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s0[0] = fmaf(g[u], w0[u][0], s0[0]); s0[1] = fmaf(g[u], w0[u][1], s0[1]); s0[2] = fmaf(g[u], w0[u][2], s0[2]); s0[3] = fmaf(g[u], w0[u][3], s0[3]);
          s1[0] = fmaf(g[u], w1[u][0], s1[0]); s1[1] = fmaf(g[u], w1[u][1], s1[1]); s1[2] = fmaf(g[u], w1[u][2], s1[2]); s1[3] = fmaf(g[u], w1[u][3], s1[3]);
        }
      }
      *reinterpret_cast<f32x4*>(D + p1row * ld0 + p1ch * 8) = s0;
      *reinterpret_cast<f32x4*>(D + p1row * ld0 + p1ch * 8 + 4) = s1;
    }
    BE_STAMP(4);
    __syncthreads();
    BE_STAMP(5);
    if (wave >= 4) {
      if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 256 && it == 25) a.stamps[17] = (long long)__builtin_readcyclecounter();
      if (p2on) {   // U2[kP2W j ..][c] += sum over the column's slot rows of dp[row,c] h1_s[row][kP2W j ..]
        const int j0 = SOc[p2c * 24 + slot], j1 = SOc[p2c * 24 + slot + 1];
        for (int j = j0; j < j1; j += 2) {   // two entries per trip (see P1)
          int rr[2]; float g[2]; f32x4 hv[2][kP2Q];
#pragma unroll
          for (int u = 0; u < 2; ++u) rr[u] = SLc[p2c * ldsc + min(j + u, j1 - 1)];
#pragma unroll
          for (int u = 0; u < 2; ++u) g[u] = j + u < j1 ? DP[rr[u] * ldp + p2c] : 0.f;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < kP2Q; ++q) hv[u][q] = *reinterpret_cast<const f32x4*>(X + rr[u] * ld0 + p2j * kP2W + q * 4);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < kP2Q; ++q) {
              u2[q * 4 + 0] = fmaf(g[u], hv[u][q][0], u2[q * 4 + 0]); u2[q * 4 + 1] = fmaf(g[u], hv[u][q][1], u2[q * 4 + 1]);
              u2[q * 4 + 2] = fmaf(g[u], hv[u][q][2], u2[q * 4 + 2]); u2[q * 4 + 3] = fmaf(g[u], hv[u][q][3], u2[q * 4 + 3]);
            }
        }
      }
      if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 256 && it == 25) a.stamps[18] = (long long)__builtin_readcyclecounter();
    }
    BE_STAMP(6);
    if (wave < nitems) {
      // ---- dh1 = (sparse part) + h1_s Q2 + q2b ; dy1 = dh1 [h1 > 0] ; Pdy += e^T dy1 ----
      const int ct = wave >> 1, rg = wave & 1;
      const int col = ct * 32 + (lane & 31);
      const bool live = col < C1;
      const float qb = live ? a.q2b[tower * C1 + col] : 0.f;
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = (live ? D[(rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ld0 + col] : 0.f) + qb;
      // mask (h1 > 0) and e^T operands of the epilogue: unconditional loads, in flight behind the MFMAs below (written as
      // short-circuit conditions they became 32 exec-masked LDS reads, each waited for on the spot: 4.4 k cycles per slot)
      const int ei = lane & 31;
      float xv[16], ev[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        xv[r] = X[row * ld0 + col];
        ev[r] = es[row * 8 + (ei & 7)];
      }
      asm volatile("" ::: "memory");
      BE_STAMP(8);
      if constexpr (BF16) {
        const unsigned short* arow = Xh + (rg * 32 + (lane & 31)) * ldh + half * 8;
#pragma unroll
        for (int kg = 0; kg < KG16; ++kg)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + kg * 16), qregh[kg], acc[0], 0, 0, 0);
      } else {
        const float* arow = X + (rg * 32 + (lane & 31)) * ld0 + half * 4;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg)
          if (kg < KGq) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + kg * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], qreg[kg][q], acc[0], 0, 0, 0);
          }
      }
      asm volatile("" ::: "memory");
      BE_STAMP(9);
      // Pdy[d][col] += sum_rows e[row][d] dy1[row][col] as one more MFMA chain: the masked accumulator element r IS the B
      // operand of step r (rows paired as the accumulator layout pairs them), A = e^T with a row of ones for sum dy1.
      // (A VALU epilogue -- two float4 LDS reads and 7 FMAs per row and lane -- cost 4.4 k cycles per slot.)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float dy = (row < nvalid && xv[r] > 0.f) ? acc[0][r] : 0.f;
        const float ea = ei < 6 ? ev[r] : (ei == 6 ? 1.f : 0.f);
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea, dy, pacc, 0, 0, 0);
      }
      if (slot == a.k - 1) {   // fp32 sums of one tile (k * 64 rows) folded into fp64; lane (col, half) holds d = 4 half + q
#pragma unroll
        for (int q = 0; q < 4; ++q) { pd[q] += (double)pacc[q]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) pacc[q] = 0.f;
      }
    }
    BE_STAMP(7);
  }
  if (p2on) {
#pragma unroll
    for (int i = 0; i < kP2W; ++i) a.u2_part[((size_t)vcloud * C1 + p2j * kP2W + i) * C2 + p2c] = u2[i];
  }
  if (wave < nitems) {
    const int ct = wave >> 1, rg = wave & 1, col = ct * 32 + (lane & 31);
    if (col < C1) {
      double* dst = a.pdy_part + ((size_t)vcloud * 4 + rg * 2 + half) * 7 * C1 + col;
#pragma unroll
      for (int d = 0; d < 7; ++d) {
        const int q = d - 4 * half;   // this half-wave holds d = 4 half .. 4 half + 3 (d = 7 does not exist); the rest of the slice is zero
        dst[(size_t)d * C1] = (q >= 0 && q < 4) ? pd[q & 3] : 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// backward edge pass, dense form ("train_matmul_bf16" with the dgcnn backbone).  Same contract as dg_train_bwd_edge (same arguments,
// same outputs), but the two products with dy2_s = dp [argk == s] -- 1 / k dense -- run on the matrix pipe as bf16 MFMAs instead of
// as VALU walks over per-tile index lists:
//     dh1_s = (dy2_s diag(k2)) round(W2)^T + h1_s Q2 + q2b          A = P masked (row-major, hi + lo),  B = image of W2^T (registers)
//     U2'  += h1_s^T (dy2_s diag(k2))                               A = Xh^T,  B = P^T masked (hi + lo);   U2' = U2 diag(k2)
// P = dp diag(k2) is split into bf16 hi + lo (16 significant bits) ONCE per tile, row-major and transposed, next to the arg-k bytes in
// both layouts; per neighbour slot the MFMA operand fragments are masked on the way into the registers (8 bytes of arg-k per 16-byte
// fragment: byte == slot -> keep), so nothing is rebuilt per slot.  h1 is the rounded h1 of the forward, W2 the rounded operand of
// the forward (what the oracle's straight-through backward uses).  No row / column lists (48 k of a tile's 265 k cycles), no sparse
// loops whose slowest lane sets the barrier, no fp32 copy of h1; the h1 tiles are double-buffered: one barrier per slot.
// LDS: es [3][64][8] | 2 x (Xh [64][C1+8] | XhT [C1][72]) | Ph, Pl [64][C2+8] | PhT, PlT [C2][72] | AK [64][C2+4] | AKT [C2][72] bytes
// ---------------------------------------------------------------------------------
static inline size_t dg_bwd_edge_dense_lds(int C1, int C2)
{
  return (size_t)3 * kTT * 8 * sizeof(float) + 2 * ((size_t)kTT * (C1 + 8) + (size_t)C1 * (kTT + 8)) * 2 +
         (2 * (size_t)kTT * (C2 + 8) + 2 * (size_t)C2 * (kTT + 8)) * 2 + (size_t)kTT * (C2 + 4) + (size_t)C2 * (kTT + 8);
}

// 8 arg-k bytes (w0 = elements 0..3, w1 = 4..7) against the slot: 16-bit lane masks for the 8 bf16 values of a fragment.
// All bytes are < 0x80 (slots 0 .. 19; 0x7f marks rows past the cloud), so byte-parallel arithmetic has no carries: x + 0x7f has its
// top bit set iff the byte of x = w ^ slot is non-zero; the matching bytes become 0xff, v_perm_b32 doubles each byte into a 16-bit lane.
__device__ __forceinline__ uint4 dg_slot_mask(unsigned w0, unsigned w1, unsigned slot4)
{
  auto bytes = [&](unsigned wv) {
    const unsigned x = wv ^ slot4;
    const unsigned z = ~(x + 0x7f7f7f7fu) & 0x80808080u;   // 0x80 where arg-k == slot
    return z | (z - (z >> 7));                              // 0xff there
  };
  const unsigned m0 = bytes(w0), m1 = bytes(w1);
  return uint4{__builtin_amdgcn_perm(m0, m0, 0x01010000u), __builtin_amdgcn_perm(m0, m0, 0x03030202u),
               __builtin_amdgcn_perm(m1, m1, 0x01010000u), __builtin_amdgcn_perm(m1, m1, 0x03030202u)};
}
__device__ __forceinline__ bf16x8 dg_masked(const unsigned short* p, const uint4& m)
{
  uint4 v = *reinterpret_cast<const uint4*>(p);
  v.x &= m.x; v.y &= m.y; v.z &= m.z; v.w &= m.w;
  return *reinterpret_cast<bf16x8*>(&v);
}

template <int C1, int C2>
__global__ __launch_bounds__(kBEW * 64) void dg_train_bwd_edge_dense(const DgBwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  constexpr int ldh = C1 + 8, ldT = kTT + 8, ldak = C2 + 4, ldd = C2 + 8;
  constexpr int CT1 = C1 / 32, CT2 = C2 / 32, KG1 = C1 / 16, KG2 = C2 / 16;
  constexpr int kXbuf = kTT * ldh + C1 * ldT;                       // bf16 elements of one (Xh | XhT) buffer
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem + 3 * kTT * 8);
  unsigned short* Ph = Xb + 2 * kXbuf;
  unsigned short* Pl = Ph + kTT * ldd;
  unsigned short* PhT = Pl + kTT * ldd;
  unsigned short* PlT = PhT + C2 * ldT;
  unsigned char* AK = reinterpret_cast<unsigned char*>(PlT + C2 * ldT);   // [64][ldak]
  unsigned char* AKT = AK + kTT * ldak;                                    // [C2][ldT]
  const int ntiles = (a.N + kTT - 1) / kTT;
  const int it0 = (part * ntiles / a.parts) * a.k, total = ((part + 1) * ntiles / a.parts) * a.k;   // this workgroup's (tile, slot) range; it0 is even (k = 20)
  const float* sc1 = a.sc1 + tower * C1;
  const float* sh1 = a.sh1 + tower * C1;
  const DgtLiftM<C1, kBEW> lw = dgt_liftm_load<C1, kBEW>(a.w1, sc1, sh1, wave, lane);
  if (tid < 3 * kTT) { smem[tid * 8 + 6] = 0.f; smem[tid * 8 + 7] = 0.f; }   // k padding of the MFMA lift in the three es buffers
  // Every wave owns the same share of both products (round 3: four dh1 waves at ~7 k cycles per slot were what the barrier waited for):
  //   dh1  16-row group rg = wave >> 1, the NCT column tiles (16 wide) ct16 = (wave & 1) NCT + j -- exactly the tiles whose h1 the wave
  //        lifts, so the relu mask [h1 > 0] stays in registers -- on v_mfma_f32_16x16x32_bf16; a masked A fragment feeds all NCT tiles;
  //        Pdy on v_mfma_f32_16x16x4_f32 (the accumulator element r of lane group g IS the B operand of step r: rows 4 g + r)
  //   U2'  P column tile jt2 = wave >> 1, K (row) half kh = wave & 1, both h1 column tiles: a masked B fragment feeds CT1 tiles; the two
  //        K halves are summed through LDS after the last slot
  constexpr int NCT = C1 / 32, K2 = C2 / 32, K1 = C1 / 32;
  static_assert(kBEW == 8, "wave roles are laid out for eight waves");
  const int rg = wave >> 1, chalf = wave & 1, g4 = lane >> 4, n16 = lane & 15;
  bf16x8 w2f[NCT][K2], q2f[NCT][K1];   // B fragments of W2^T (K = C2) and Q2 (K = C1), read out of the 32x32x16 images
  float qb[NCT];
  {
    const bf16x8* wi = reinterpret_cast<const bf16x8*>(a.w2th);
    const bf16x8* qh = reinterpret_cast<const bf16x8*>(a.q2imgh + tower * a.q2imgh_stride);
#pragma unroll
    for (int j2 = 0; j2 < NCT; ++j2) {
      const int ct16 = chalf * NCT + j2, ct32 = ct16 >> 1, l32 = (g4 & 1) * 32 + 16 * (ct16 & 1) + n16;   // image lane: k half, column
#pragma unroll
      for (int kg = 0; kg < K2; ++kg) w2f[j2][kg] = wi[((size_t)ct32 * KG2 + 2 * kg + (g4 >> 1)) * 64 + l32];
#pragma unroll
      for (int kg = 0; kg < K1; ++kg) q2f[j2][kg] = qh[((size_t)ct32 * KG1 + 2 * kg + (g4 >> 1)) * 64 + l32];
      qb[j2] = a.q2b[tower * C1 + ct16 * 16 + n16];
    }
  }
  f32x16 uacc[CT1];
#pragma unroll
  for (int i = 0; i < CT1; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) uacc[i][r] = 0.f;
  double pd[NCT][4];
  f32x4 pacc[NCT];
#pragma unroll
  for (int j2 = 0; j2 < NCT; ++j2)
#pragma unroll
    for (int d = 0; d < 4; ++d) { pd[j2][d] = 0.0; pacc[j2][d] = 0.f; }

  // edge features run two slots ahead of the MFMAs: es(it + 1) is written while slot it is lifted (three buffers: the dh1 waves of a
  // slow slot it may still be reading es(it) when a fast wave writes es(it + 2)), the gather of slot it + 2 is in flight meanwhile
  float v[6];
  int jnext = 0;   // neighbour index of slot it + 3 at the top of iteration it: one step ahead of the point loads that need it
  if (tid < kTT) {
    dgt_points(pc, a.N, a.k, it0, tid, dgt_index(nnc, a.N, a.k, it0, tid), v);
    dg_edge_to_lds(xf, v, smem + tid * 8);
    if (total > it0 + 1) dgt_points(pc, a.N, a.k, it0 + 1, tid, dgt_index(nnc, a.N, a.k, it0 + 1, tid), v);
    if (total > it0 + 2) jnext = dgt_index(nnc, a.N, a.k, it0 + 2, tid);
  }
  __syncthreads();
  for (int it = it0; it < total; ++it) {
    const int tile = it / a.k, slot = it - tile * a.k;
    const int nvalid = min(kTT, a.N - tile * kTT);
    const bool more = it + 1 < total;
    float* es = smem + ((it - it0) % 3) * kTT * 8;   // (the three edge-feature buffers rotate from this workgroup's first slot)
    unsigned short* Xh = Xb + (it & 1) * kXbuf;
    unsigned short* XhT = Xh + kTT * ldh;
    if (slot == 0) {
      __syncthreads();   // the previous tile's readers of the P / arg-k tiles are done
      // P = dp diag(k2) as bf16 hi / lo, row-major and transposed, and the arg-k bytes in both layouts: once per tile
      const size_t base = ((size_t)cloud * a.N + (size_t)tile * kTT) * C2;
      constexpr int c8 = C2 / 8;
      for (int item = tid; item < kTT * c8; item += kBEW * 64) {
        const int r = item / c8, o = item % c8;
        const bool ok = r < nvalid;
        const float* src = a.dyp + base + (size_t)r * C2 + o * 8;
        const f32x4 d0 = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 d1 = ok ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(a.k2 + tower * C2 + o * 8), k1 = *reinterpret_cast<const f32x4*>(a.k2 + tower * C2 + o * 8 + 4);
        uint2 ak = {0x7f7f7f7fu, 0x7f7f7f7fu};   // (rows past the cloud: no slot; < 0x80 for dg_slot_mask)
        if (ok) ak = *reinterpret_cast<const uint2*>(a.argk + base + (size_t)r * C2 + o * 8);
        const float dv[8] = {d0[0] * k0[0], d0[1] * k0[1], d0[2] * k0[2], d0[3] * k0[3], d1[0] * k1[0], d1[1] * k1[1], d1[2] * k1[2], d1[3] * k1[3]};
        unsigned hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          hi[e] = to_bf16_bits(dv[e]);
          lo[e] = to_bf16_bits(dv[e] - __uint_as_float(hi[e] << 16));
        }
        *reinterpret_cast<uint4*>(Ph + r * ldd + o * 8) = uint4{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16)};
        *reinterpret_cast<uint4*>(Pl + r * ldd + o * 8) = uint4{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16)};
        *reinterpret_cast<unsigned*>(AK + r * ldak + o * 8) = ak.x;
        *reinterpret_cast<unsigned*>(AK + r * ldak + o * 8 + 4) = ak.y;
      }
      __syncthreads();
      // transposed copies, lanes along the columns (conflict-free: consecutive elements of a row in, 16 / 8 bytes of a column out; written
      // element by element from the loop above the 2-byte stores of a wave fell into two banks)
      for (int item = tid; item < C2 * (kTT / 8); item += kBEW * 64) {
        const int c = item % C2, ro = item / C2;
        unsigned wh[4], wl[4], wa[2] = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int rr = ro * 8 + e;
          const unsigned h1 = Ph[rr * ldd + c], l1 = Pl[rr * ldd + c], a1 = AK[rr * ldak + c];
          if (e & 1) { wh[e >> 1] |= h1 << 16; wl[e >> 1] |= l1 << 16; } else { wh[e >> 1] = h1; wl[e >> 1] = l1; }
          wa[e >> 2] |= a1 << (8 * (e & 3));
        }
        *reinterpret_cast<uint4*>(PhT + c * ldT + ro * 8) = uint4{wh[0], wh[1], wh[2], wh[3]};
        *reinterpret_cast<uint4*>(PlT + c * ldT + ro * 8) = uint4{wl[0], wl[1], wl[2], wl[3]};
        *reinterpret_cast<uint2*>(AKT + c * ldT + ro * 8) = uint2{wa[0], wa[1]};
      }
    }
    unsigned short hv[NCT][4];
    dgt_liftm_bf16_keep<C1, kBEW>(lw, es, Xh, ldh, XhT, ldT, nvalid, wave, lane, hv);   // (this h1 buffer's readers, two slots back, are behind the last barrier)
    if (more && tid < kTT) {
      dg_edge_to_lds(xf, v, smem + ((it + 1 - it0) % 3) * kTT * 8 + tid * 8);
      if (it + 2 < total) dgt_points(pc, a.N, a.k, it + 2, tid, jnext, v);
      if (it + 3 < total) jnext = dgt_index(nnc, a.N, a.k, it + 3, tid);
    }
    __syncthreads();   // the slot's only barrier: h1 tiles (and at a tile start the P tiles) complete, es of the next slot written
    const unsigned slot4 = (unsigned)slot * 0x01010101u;
    {
      // ---- dh1 = (P masked) W2^T + h1_s Q2 + q2b ; dy1 = dh1 [h1 > 0] ; Pdy += e^T dy1 ----
      const int arow = rg * 16 + n16;
      const unsigned short* ah = Ph + arow * ldd + g4 * 8;
      const unsigned short* al = Pl + arow * ldd + g4 * 8;
      const unsigned char* am = AK + arow * ldak + g4 * 8;
      const unsigned short* ax = Xh + arow * ldh + g4 * 8;
      f32x4 acc[NCT];
#pragma unroll
      for (int j2 = 0; j2 < NCT; ++j2) acc[j2] = f32x4{qb[j2], qb[j2], qb[j2], qb[j2]};
#pragma unroll
      for (int kg = 0; kg < K2; ++kg) {
        const uint4 m = dg_slot_mask(*reinterpret_cast<const unsigned*>(am + kg * 32), *reinterpret_cast<const unsigned*>(am + kg * 32 + 4), slot4);
        const bf16x8 fh = dg_masked(ah + kg * 32, m), fl = dg_masked(al + kg * 32, m);
#pragma unroll
        for (int j2 = 0; j2 < NCT; ++j2) {
          acc[j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, w2f[j2][kg], acc[j2], 0, 0, 0);
          acc[j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, w2f[j2][kg], acc[j2], 0, 0, 0);
        }
      }
#pragma unroll
      for (int kg = 0; kg < K1; ++kg) {
        const bf16x8 fx = *reinterpret_cast<const bf16x8*>(ax + kg * 32);
#pragma unroll
        for (int j2 = 0; j2 < NCT; ++j2) acc[j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, q2f[j2][kg], acc[j2], 0, 0, 0);
      }
      float ea[4];   // A operand of the Pdy steps: e[row 4 g + r][d = n16] for d < 6, a row of ones (d = 6) for sum dy1
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ev = es[(rg * 16 + 4 * g4 + r) * 8 + (n16 & 7)];
        ea[r] = n16 < 6 ? ev : (n16 == 6 ? 1.f : 0.f);
      }
#pragma unroll
      for (int j2 = 0; j2 < NCT; ++j2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dy = hv[j2][r] ? acc[j2][r] : 0.f;   // (rows past the cloud lift to h1 = 0)
          pacc[j2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[r], dy, pacc[j2], 0, 0, 0);
        }
      if (slot == a.k - 1) {   // fp32 sums of one tile (k * 16 rows) folded into fp64
#pragma unroll
        for (int j2 = 0; j2 < NCT; ++j2)
#pragma unroll
          for (int q = 0; q < 4; ++q) { pd[j2][q] += (double)pacc[j2][q]; pacc[j2][q] = 0.f; }
      }
    }
    {
      // ---- U2' tiles (it2, jt2) += Xh^T[32 it2 .., rows] . (P^T masked)[32 jt2 .., rows] over this wave's half of the tile's 64 rows ----
      const int jt2 = wave >> 1, kh = wave & 1;
      if (jt2 < CT2) {
        const int brow = jt2 * 32 + (lane & 31);
        const unsigned short* ph = PhT + brow * ldT + half * 8;
        const unsigned short* pl = PlT + brow * ldT + half * 8;
        const unsigned char* pm = AKT + brow * ldT + half * 8;
#pragma unroll
        for (int kk = 0; kk < kTT / 32; ++kk) {
          const int kg = kh * (kTT / 32) + kk;
          const uint2 mk = *reinterpret_cast<const uint2*>(pm + kg * 16);
          const uint4 m = dg_slot_mask(mk.x, mk.y, slot4);
          const bf16x8 bh = dg_masked(ph + kg * 16, m), bl = dg_masked(pl + kg * 16, m);
#pragma unroll
          for (int it2 = 0; it2 < CT1; ++it2) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(XhT + (it2 * 32 + (lane & 31)) * ldT + half * 8 + kg * 16);
            uacc[it2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, uacc[it2], 0, 0, 0);
            uacc[it2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl, uacc[it2], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();   // the last slot's readers of the P tiles are done: their LDS takes the K-half partials of U2'
  {
    float* red = reinterpret_cast<float*>(Ph);   // [CT2 * CT1 tiles][16][64]
    const int jt2 = wave >> 1, kh = wave & 1;
    if (kh == 1 && jt2 < CT2) {
#pragma unroll
      for (int it2 = 0; it2 < CT1; ++it2)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((jt2 * CT1 + it2) * 16 + r) * 64 + lane] = uacc[it2][r];
    }
    __syncthreads();
    if (kh == 0 && jt2 < CT2) {
#pragma unroll
      for (int it2 = 0; it2 < CT1; ++it2) {
        float other[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) other[r] = red[((jt2 * CT1 + it2) * 16 + r) * 64 + lane];
        tile_commit(a.u2_part + (size_t)vcloud * C1 * C2, C2, it2, jt2, C1, C2, uacc[it2], lane, other);
      }
    }
  }
  if (g4 < 2) {   // Pdy [cloud][slice = 16-row group][d][C1]: lane group g holds d = 4 g + q
#pragma unroll
    for (int j2 = 0; j2 < NCT; ++j2) {
      const int col = (chalf * NCT + j2) * 16 + n16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = 4 * g4 + q;
        if (d < 7) a.pdy_part[(((size_t)vcloud * 4 + rg) * 7 + d) * C1 + col] = pd[j2][q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// first-layer backward from the reduced quantities (no [B*N*k, C1] tensor is read):
//   dz1_r = k (dy1_r - dbeta/M - zhat_r dgamma/M),  zhat_r = (e_r . w_c + b - mu) rstd
//   dbeta = sum sdy,  dgamma = rstd (sum_d w_dc Pdy_dc + (b - mu) sdy)                                (dg_b0_totals)
//   per cloud:  P_dc = sum_r e_rd dz1_rc = k [Pdy_dc - mb Se_d - mg rstd (sum_d' Ge_dd' w_d'c + (b - mu) Se_d)]
//               S_c  = sum_r dz1_rc      = k [sdy_c - n mb - mg rstd (Se . w_c + n (b - mu))]
//               gx_d = sum_c w_dc S_c (d < 3),  grot = sum_c (w_0c P_1c - w_1c P_0c + w_3c P_4c - w_4c P_3c)   (dg_b0_cloud)
//   dW1 = sum over clouds of P.
// ---------------------------------------------------------------------------------
// D = 6: DGCNN (e = [x_i, x_j - x_i]);  D = 3: the PointNet first layer (e = x'), where the same identities replace pass B0 and
// the stored dy1: moments [2B][D + D(D+1)/2] = sum e | upper triangle of sum e e^T (row-major, d <= d2); Pdy [2B][4][D + 1][C1].
struct DgB0Args {
  const double* pdy_part;   // [2B][slices][D + 1][C1]
  int slices;               // partial sums per cloud: 4 (row group x half-wave), or 1 when the producer reduced them itself
  const double* mom;        // [2B][D + D (D + 1) / 2]
  const float* w1; const float* b1; const float *mean1, *rstd1, *k1;   // [D][C1], [C1], [2][C1] x 3
  int B, C1, rows;          // rows per cloud (DGCNN: N * k)
  double count;             // M = B * rows
  float* dbeta[2]; float* dgamma[2];
  float* dbg1;              // [2][C1][2] totals (dbeta1, dgamma1)
  float* p_part;            // [2B][D][C1]
  float* gx; float* grot;   // [2B][3], [2B]
  // glue between the stages, folded into dg_b0_cloud (one thread per cloud does what stage3/2/1_glue_bwd_kernel do, kernels_train_head.h):
  // 3 = after the stage-3 backbone (frame of stage 3: x3 = (p - s2c) R(-theta)), 2 = after the stage-2 backbone (x2 = p - s1c), 0 = none
  int glue = 0; const float* xform = nullptr; const int* pcls = nullptr; int nb = 0;
  float* d_s2c = nullptr; float* d_o2 = nullptr; int ldo2 = 0; float* d_s1c = nullptr; float* d_o0 = nullptr;
};

constexpr int kB0C = 8;   // channels per workgroup of dg_b0_totals (x 128 cloud groups): with 32 x 32 the fp32 PointNet step (C1 = 64, four slices per cloud) ran it
                          // on four workgroups walking eight dependent trips of L2 round trips each -- 17 us per launch
template <int D>
__global__ __launch_bounds__(1024) void dg_b0_totals(const DgB0Args a)   // grid (ceil(C1 / kB0C), 2), block kB0C channels x 128 cloud groups
{
  constexpr int kG = 1024 / kB0C;
  __shared__ double red[kG][kB0C][2];
  const int cl = threadIdx.x % kB0C, g = threadIdx.x / kB0C, c = blockIdx.x * kB0C + cl, t = blockIdx.y;
  double sdy = 0.0, wp = 0.0;
  if (c < a.C1) {
    double w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = (double)a.w1[d * a.C1 + c];
    // four slices per iteration: 4 (D + 1) independent loads in flight
    const int S = a.B * a.slices;
    for (int bs = g; bs < S; bs += 4 * kG) {
      double v[4][D + 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bu = min(bs + kG * u, S - 1);
        const double* p = a.pdy_part + ((size_t)t * S + bu) * (D + 1) * a.C1 + c;
#pragma unroll
        for (int d = 0; d <= D; ++d) v[u][d] = p[(size_t)d * a.C1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (bs + kG * u < S) {
#pragma unroll
          for (int d = 0; d < D; ++d) wp += w[d] * v[u][d];
          sdy += v[u][D];
        }
    }
  }
  red[g][cl][0] = sdy; red[g][cl][1] = wp;
  __syncthreads();
  // 128 partials per channel: sixteen lanes of a channel sum eight each, then a four-step shuffle (fixed order)
  if (threadIdx.x >= kB0C * 16) return;
  const int ch = threadIdx.x >> 4, part = threadIdx.x & 15, cc = blockIdx.x * kB0C + ch;
  sdy = 0.0; wp = 0.0;
#pragma unroll
  for (int q = 0; q < kG / 16; ++q) { sdy += red[part * (kG / 16) + q][ch][0]; wp += red[part * (kG / 16) + q][ch][1]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { sdy += __shfl_xor(sdy, o); wp += __shfl_xor(wp, o); }
  if (part != 0 || cc >= a.C1) return;
  const double dg = (double)a.rstd1[t * a.C1 + cc] * (wp + ((double)a.b1[cc] - (double)a.mean1[t * a.C1 + cc]) * sdy);
  a.dbeta[t][cc] = (float)sdy;
  a.dgamma[t][cc] = (float)dg;
  a.dbg1[(t * a.C1 + cc) * 2] = (float)sdy;
  a.dbg1[(t * a.C1 + cc) * 2 + 1] = (float)dg;
}

template <int D>
__global__ __launch_bounds__(128) void dg_b0_cloud(const DgB0Args a)   // grid 2B, block 128 (C1 <= 128)
{
  constexpr int kMom = D + D * (D + 1) / 2;
  __shared__ double mo[kMom];
  __shared__ double red[2][4];
  const int cloud = blockIdx.x, t = cloud >= a.B, c = threadIdx.x;
  if (threadIdx.x < kMom) mo[threadIdx.x] = a.mom[(size_t)cloud * kMom + threadIdx.x];
  __syncthreads();
  double g[4] = {0.0, 0.0, 0.0, 0.0};   // gx0, gx1, gx2, grot
  if (c < a.C1) {
    double w[D], pdy[D + 1];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = (double)a.w1[d * a.C1 + c];
#pragma unroll
    for (int d = 0; d < D + 1; ++d) {
      double s = 0.0;
      for (int sl = 0; sl < a.slices; ++sl) s += a.pdy_part[(((size_t)cloud * a.slices + sl) * (D + 1) + d) * a.C1 + c];
      pdy[d] = s;
    }
    const double k = (double)a.k1[t * a.C1 + c], rs = (double)a.rstd1[t * a.C1 + c];
    const double bm = (double)a.b1[c] - (double)a.mean1[t * a.C1 + c];
    const double mb = (double)a.dbg1[(t * a.C1 + c) * 2] / a.count, mg = (double)a.dbg1[(t * a.C1 + c) * 2 + 1] / a.count;
    const double n = (double)a.rows;
    double P[D];
    double sew = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) sew += mo[d] * w[d];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      double gw = 0.0;
#pragma unroll
      for (int d2 = 0; d2 < D; ++d2) {   // symmetric second moment: index of (d, d2) in the packed upper triangle
        const int lo = d < d2 ? d : d2, hi = d < d2 ? d2 : d;
        const int q = D + lo * D - lo * (lo - 1) / 2 + (hi - lo);
        gw += mo[q] * w[d2];
      }
      P[d] = k * (pdy[d] - mb * mo[d] - mg * rs * (gw + bm * mo[d]));
      a.p_part[((size_t)cloud * D + d) * a.C1 + c] = (float)P[d];
    }
    const double S = k * (pdy[D] - n * mb - mg * rs * (sew + n * bm));
    g[0] = w[0] * S; g[1] = w[1] * S; g[2] = w[2] * S;
    g[3] = w[0] * P[1] - w[1] * P[0];
    if (D == 6) g[3] += w[3] * P[4] - w[4] * P[3];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double v = g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) a.gx[cloud * 3 + threadIdx.x] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x]);
  if (threadIdx.x == 3) a.grot[cloud] = (float)(red[0][3] + red[1][3]);
  if (a.glue && threadIdx.x == 0) {
    const float g0 = (float)(red[0][0] + red[1][0]), g1 = (float)(red[0][1] + red[1][1]), g2 = (float)(red[0][2] + red[1][2]), gr = (float)(red[0][3] + red[1][3]);
    if (a.glue == 3) {
      // x3 = (p - s2c) R(-theta):  dL/ds2c = -(gx R^T),  dL/dtheta = -grot -> the residual logit of the predicted class (tp8.py:298-301);
      // s2c = o2[:, :3] + s1c (tp8.py:117):  d_o2[:, :3] = d_s2c,  d_s1c += d_s2c
      const float* R = a.xform + cloud * 12 + 3;
      float d[3];
      d[0] = a.d_s2c[cloud * 3 + 0] - (g0 * R[0] + g1 * R[1] + g2 * R[2]);
      d[1] = a.d_s2c[cloud * 3 + 1] - (g0 * R[3] + g1 * R[4] + g2 * R[5]);
      d[2] = a.d_s2c[cloud * 3 + 2] - (g0 * R[6] + g1 * R[7] + g2 * R[8]);
      const float pinb = (float)(3.141592653589793 / (double)a.nb);
      a.d_o2[(size_t)cloud * a.ldo2 + 3 + a.nb + a.pcls[cloud]] += -gr * pinb;
#pragma unroll
      for (int e = 0; e < 3; ++e) { a.d_s2c[cloud * 3 + e] = d[e]; a.d_o2[(size_t)cloud * a.ldo2 + e] = d[e]; a.d_s1c[cloud * 3 + e] += d[e]; }
    } else {
      // x2 = p - s1c (tp8.py:113):  d_s1c -= gx;  s1c = o1 + center_mean:  d_o1 = d_s1c
      const float gg[3] = {g0, g1, g2};
#pragma unroll
      for (int e = 0; e < 3; ++e) { const float v = a.d_s1c[cloud * 3 + e] - gg[e]; a.d_s1c[cloud * 3 + e] = v; a.d_o0[cloud * 3 + e] = v; }
    }
  }
}

// first and second moments of the stage-frame points x' of every cloud (the D = 3 input of the two kernels above), fp64.  grid 2B
// With w1 / b1 / stat_part given it is also phase 1 of the PointNet forward: z1 = x' W1 + b1 is linear in x', so the per-channel
// sum and sum of squares of z1 over the cloud follow from the nine moments (as dg_train_phase1 does for the edge layer).
__device__ __forceinline__ void pn_moments_body(const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                const float* __restrict__ xform, int B, int N, double* __restrict__ mom,
                                                const float* __restrict__ w1, const float* __restrict__ b1, int C1,
                                                double* __restrict__ stat_part, int cloud)   // 256 threads
{
  __shared__ double red[4][9];
  __shared__ double tot[9];
  const int tower = cloud >= B, b = cloud - tower * B, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pc = (tower ? pcs2 : pcs1) + (size_t)b * N * 3;
  const float* xf = xform + (size_t)cloud * 12;
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int n = threadIdx.x; n < N; n += 256) {
    const float x = pc[n * 3] - xf[0], y = pc[n * 3 + 1] - xf[1], z = pc[n * 3 + 2] - xf[2];
    const double e[3] = {(double)(x * xf[3] + y * xf[6] + z * xf[9]), (double)(x * xf[4] + y * xf[7] + z * xf[10]),
                         (double)(x * xf[5] + y * xf[8] + z * xf[11])};
    int q = 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      m[d] += e[d];
#pragma unroll
      for (int d2 = d; d2 < 3; ++d2) m[q++] += e[d] * e[d2];
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double v = m[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    tot[threadIdx.x] = t;
    mom[(size_t)cloud * 9 + threadIdx.x] = t;
  }
  if (!stat_part) return;
  __syncthreads();
  for (int c = threadIdx.x; c < C1; c += 256) {
    const double w[3] = {(double)w1[c], (double)w1[C1 + c], (double)w1[2 * C1 + c]};
    double sw = 0.0, qd = 0.0;
    int q = 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      sw += w[d] * tot[d];
#pragma unroll
      for (int d2 = d; d2 < 3; ++d2) qd += (d == d2 ? 1.0 : 2.0) * w[d] * w[d2] * tot[q++];
    }
    const double bb = (double)b1[c], n = (double)N;
    stat_part[((size_t)cloud * C1 + c) * 2] = sw + n * bb;
    stat_part[((size_t)cloud * C1 + c) * 2 + 1] = qd + 2.0 * bb * sw + n * bb * bb;
  }
}
__global__ __launch_bounds__(256) void pn_moments_kernel(const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                         const float* __restrict__ xform, int B, int N, double* __restrict__ mom,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, int C1,
                                                         double* __restrict__ stat_part)
{
  pn_moments_body(pcs1, pcs2, xform, B, N, mom, w1, b1, C1, stat_part, blockIdx.x);
}

// The moments of a stage's frame need nothing but that frame, and a frame is produced per cloud: the kernel that writes cloud c's frame -- the
// start kernel (centroid), the stage-1 / stage-2 glue -- goes straight on to cloud c's moments (one workgroup per cloud, 256 threads), instead of a
// launch of its own per stage.  mom == nullptr: no tail (the stage is not a specialised PointNet stage).
struct MomentsTail { double* mom = nullptr; const float* w1 = nullptr; const float* b1 = nullptr; int C1 = 0; double* stat_part = nullptr; };
__device__ __forceinline__ void moments_tail(const MomentsTail& mt, const float* pcs1, const float* pcs2, const float* xform, int B, int N, int cloud)
{
  if (!mt.mom) return;
  __syncthreads();   // the frame just written by this workgroup (workgroup-scope release / acquire: it is read through the CU's own L1 / L2 path)
  pn_moments_body(pcs1, pcs2, xform, B, N, mt.mom, mt.w1, mt.b1, mt.C1, mt.stat_part, cloud);
}
// The first launch of a training step: the clouds' centroids (frame of stage 1) and, as further blocks of the same grid, the step's weight images --
// the fp32 MFMA images of the conv layers and / or the bf16 images; nothing in one part reads what another writes.  (Three launches before: 6 us each.)
constexpr int kStartF32Blocks = 256, kStartBf16Blocks = 128;   // blocks per image (<= 2 / 4 elements per thread)
__global__ __launch_bounds__(256) void train_start_kernel(const float* __restrict__ pcs1, const float* __restrict__ pcs2, int B, int N, float* __restrict__ xform,
                                                          float* __restrict__ center_mean, const PackJob* __restrict__ f32jobs, int nf32, const PackBf16Jobs pj, int nbf16,
                                                          const MomentsTail mt)
{
  int bx = blockIdx.x;
  if (bx < 2 * B) { centroid_body(pcs1, pcs2, B, N, xform, center_mean, nullptr, 0, bx, 2 * B); moments_tail(mt, pcs1, pcs2, xform, B, N, bx); return; }
  bx -= 2 * B;
  if (bx < nf32 * kStartF32Blocks) { pack_weights_multi_body(f32jobs, bx % kStartF32Blocks, bx / kStartF32Blocks, kStartF32Blocks); return; }
  bx -= nf32 * kStartF32Blocks;
  if (bx < nbf16 * kStartBf16Blocks) pack_bf16_jobs_body(pj, bx % kStartBf16Blocks, bx / kStartBf16Blocks, kStartBf16Blocks);
}

// stage-1 / stage-2 glue (kernels_infer.h: stage1_finish_cloud, stage2_finish_cloud) + the next stage's moments, one workgroup per cloud
__global__ __launch_bounds__(256) void stage1_finish_moments_kernel(const float* __restrict__ o1, const float* __restrict__ center_mean, int B, int N,
                                                                    float* __restrict__ s1c, float* __restrict__ xform, float* __restrict__ out_c1,
                                                                    float* __restrict__ out_c2, const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                                    const MomentsTail mt)
{
  const int cloud = blockIdx.x;
  if (threadIdx.x == 0) stage1_finish_cloud(o1, center_mean, B, s1c, xform, out_c1, out_c2, cloud);
  moments_tail(mt, pcs1, pcs2, xform, B, N, cloud);
}
__global__ __launch_bounds__(256) void stage2_finish_moments_kernel(const float* __restrict__ o2, int ldo, const float* __restrict__ s1c, int B, int N, int nb,
                                                                    float* __restrict__ s2c, float* __restrict__ xform, float* __restrict__ theta_out,
                                                                    int* __restrict__ cls_out, float* __restrict__ out_c1, float* __restrict__ out_c2,
                                                                    float* __restrict__ out_l1, float* __restrict__ out_l2, const float* __restrict__ pcs1,
                                                                    const float* __restrict__ pcs2, const MomentsTail mt)
{
  const int cloud = blockIdx.x;
  if (threadIdx.x < 64) stage2_finish_cloud(o2, ldo, s1c, B, nb, s2c, xform, theta_out, cls_out, out_c1, out_c2, out_l1, out_l2, cloud, threadIdx.x);
  moments_tail(mt, pcs1, pcs2, xform, B, N, cloud);
}

}  // namespace alignnet
