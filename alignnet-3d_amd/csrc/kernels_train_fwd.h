// Training-mode forward of the shared-MLP backbone (models/tp8.py:49-59 with is_training=True,
// utils/tf_util.py:455-492 batch-statistics BatchNorm), gfx950 only.
//
// Training BN couples all B*N points of a tower, so the forward is three grid-wide phases; each phase
// recomputes the cheap earlier layers from xyz (inputs are 12 B/point) instead of round-tripping
// [B*N, C] activations through HBM:
//   phase 1: statistics of z1 = x' W1 + b1                      (from the moments of x', kernels_train_dgcnn.h)
//   phase 2: h1 = relu(bn1(z1));  statistics of z2 = h1 W2 + b2 (MFMA)
//   phase 3: h1, h2;  z3 = h2 W3 + b3: statistics, and -- because
//              max_n relu(g*z_n + b) = relu(g * max_n z_n + b)   for g >= 0   (min_n for g < 0)
//            -- only the per-(cloud, channel) extreme of sign(gamma)*z3 and its point index.  z3
//            ([B*N, 1024]) is never materialised.  The same pass accumulates the Gram matrix
//            h2^T h2 and the column sums of h2, which is all the dense part of the conv3 backward needs
//            (kernels_train_bwd.h), and stores h2 for the sparse (arg-max) part.
// One workgroup per cloud walks its 128-point tiles; sums are kept in fp64 per lane (sum, sum of
// squares -> biased variance E[z^2]-E[z]^2 without fp32 cancellation), then one partial per cloud.
#pragma once
#include "ablate.h"
#include "kernels_infer.h"

namespace alignnet {

// Training kernels walk 64-point tiles: LDS <= 77 KiB per workgroup -> two workgroups (two clouds) per CU overlap
// each other's barriers, global round trips and epilogues.
constexpr int kTT = 64;
constexpr int kTW = 4;     // waves per training workgroup: 2 workgroups/CU = 2 waves/SIMD -> 256 VGPRs each, no spills

struct TrainFwdArgs {
  const float* pcs[2];
  const float* xform;      // [2B][12]
  int B, N;
  int C1, C2, C3;
  int ld[2];               // LDS leading dims: buf0 holds h1 ([128][C1r+4]), buf1 holds h2 ([128][C2r+4])
  const float* w1;         // [3][C1]
  const float* wp2;        // MFMA image of W2 [C1][C2]
  const float* wp3;        // MFMA image of W3 [C2][C3]
  const unsigned short* wp3h;   // bf16 MFMA image of W3 (train_bf16 mode; packed by pack_bf16_jobs_kernel)
  const unsigned short* wp2h;   // bf16 MFMA image of W2 (train_bf16 mode, no sign folding)
  const float *b1, *b2, *b3;       // conv biases (added before BN: utils/tf_util.py:161)
  const float *sc1, *sh1;  // [2][C1] batch-stat scale/shift of layer 1 (phase >= 2)
  const float *sc2, *sh2;  // [2][C2] (phase 3)
  const float* sgn3;       // [2][C3] sign(gamma3) as +1/-1 (phase 3)
  double* stat_part;       // [2B][2 halves][C][2] partial (sum, sumsq) of this phase's layer (phase 1: [2B][1][C][2])
  float* ext;              // [2B][2 halves][C3] extreme of sgn*z3   (phase 3)
  int* idx;                // [2B][2 halves][C3] its point index
  float* gram_part;        // [2B][C2*C2]
  double* colsum_part;     // [2B][4 = 2 row halves x 2 lane halves][C2]
  float* h2_store;         // [2B*N][C2]
  int gram_inline;         // fp32 phase 3: 1 = accumulate the Gram per tile in this kernel (fallback), 0 = gram_h2_kernel does it
  int dbg;                 // debug/ablation flags (0 in production)
  long long* stamps;       // debug (dbg & 32): cycle stamps of thread 0 / block 0 at the phase boundaries of tile 3
  int parts3 = 1;          // phase 3 (all three kernels): the same split; ext / idx / gram_part / colsum_part are per-WORKGROUP then ([2B * parts3] slices) and
                           // merge_ext_parts_kernel folds the running extremes of a cloud's parts (exactly: a scan in tile order) before anything reads them
  int parts = 1;           // PHASE 2 only: a cloud's tiles dealt to `parts` workgroups (grid 2B * parts, workgroup = cloud * parts + part); stat_part is then a per-WORKGROUP
                           // partial ([2B * parts][4] slices).  One workgroup per cloud leaves the chip half empty below 512 clouds (the reference's shipped batch is 128).
};
#define P3_STAMP(i) do { if (PHASE == 3 && ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 0 && tile == 3) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

// Accumulate one 32x32 MFMA result tile into a lane-owned global matrix.  The old values are requested BEFORE the
// MFMA loop that produces the new ones and pinned there (compiler memory barrier: hipcc otherwise sinks the loads
// next to their use and exposes one L2 round trip per item), then added and stored afterwards.
__device__ __forceinline__ void tile_prefetch(const float* __restrict__ base, long ld, int it, int jt, int rows, int cols, bool first,
                                              int lane, float (&old)[16])
{
  const int j = jt * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    old[r] = (first || i >= rows || j >= cols) ? 0.f : base[(size_t)i * ld + j];
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void tile_commit(float* __restrict__ base, long ld, int it, int jt, int rows, int cols, const f32x16& v,
                                            int lane, const float (&old)[16])
{
  const int j = jt * 32 + (lane & 31);
  if (j >= cols) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (i < rows) base[(size_t)i * ld + j] = old[r] + v[r];
  }
}

// x' = (p - c) @ R for the tile's 128 points -> xs[128][4]; rows past N repeat the last point (masked later)
__device__ __forceinline__ void load_tile_xform(const float* __restrict__ pc, const float* __restrict__ xf, int N,
                                                int tile, float* __restrict__ xs, int tid)
{
  if (tid < kTT) {
    const int n = min(tile * kTT + tid, N - 1);
    const float* p = pc + (size_t)n * 3;
    const float x = p[0] - xf[0], y = p[1] - xf[1], z = p[2] - xf[2];
    xs[tid * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
    xs[tid * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
    xs[tid * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
  }
}

// the same in two halves -- request the tile's raw points (registers), transform + store them later -- so that a pass can ask for tile
// t + 1 while it works on tile t (the load's HBM / L2 round trip sat exposed at the head of every tile: 2.1 - 2.6 k cycles in pass B2)
struct TilePoint { float x, y, z; };
__device__ __forceinline__ TilePoint tile_point_request(const float* __restrict__ pc, int N, int tile, int tid)
{
  TilePoint p = {0.f, 0.f, 0.f};
  if (tid < kTT) {
    const float* q = pc + (size_t)min(tile * kTT + tid, N - 1) * 3;
    p.x = q[0]; p.y = q[1]; p.z = q[2];
  }
  return p;
}
__device__ __forceinline__ void tile_point_store(const TilePoint& p, const float* __restrict__ xf, float* __restrict__ xs, int tid)
{
  if (tid < kTT) {
    const float x = p.x - xf[0], y = p.y - xf[1], z = p.z - xf[2];
    xs[tid * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
    xs[tid * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
    xs[tid * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
  }
}

// The cloud's frame (centre + rotation, 12 floats) once per workgroup in scalar registers: read through the pointer, tile_point_store
// fetched it from memory for every tile -- a vector load of a uniform address waited for on the spot by wave 0, with the other waves at
// the next barrier (~700 cycles per tile in every training pass).
struct XForm { float v[12]; };
__device__ __forceinline__ XForm xform_load(const float* __restrict__ xf)
{
  XForm X;
#pragma unroll
  for (int i = 0; i < 12; ++i) X.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xf[i])));
  return X;
}
__device__ __forceinline__ void tile_point_store(const TilePoint& p, const XForm& X, float* __restrict__ xs, int tid)
{
  if (tid < kTT) {
    const float x = p.x - X.v[0], y = p.y - X.v[1], z = p.z - X.v[2];
    xs[tid * 4 + 0] = x * X.v[3] + y * X.v[6] + z * X.v[9];
    xs[tid * 4 + 1] = x * X.v[4] + y * X.v[7] + z * X.v[10];
    xs[tid * 4 + 2] = x * X.v[5] + y * X.v[8] + z * X.v[11];
  }
}

// layer 1 on the VALU: out[row][c] = relu((x' . w[:,c]) * sc + sh); rows >= nvalid are written as 0.
// Layer1W holds the thread's weights / scale / shift for its (<= 4) channel groups c0, c0 + 32, ...: loaded once per
// workgroup (layer1_load) instead of once per tile -- the reload put an L2 round trip in front of every tile's first barrier.
struct Layer1W { float w0[2], wa[2], wb[2], s[2], t[2]; const float *w1, *sc, *sh; };   // groups 0, 1 in registers (C1 <= 64)

__device__ __forceinline__ Layer1W layer1_load(const float* __restrict__ w1, int C1, const float* __restrict__ sc,
                                               const float* __restrict__ sh, int tid)
{
  Layer1W L;
  L.w1 = w1; L.sc = sc; L.sh = sh;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = (tid & 31) + 32 * g;
    const bool live = c < C1;
    L.w0[g] = live ? w1[c] : 0.f; L.wa[g] = live ? w1[C1 + c] : 0.f; L.wb[g] = live ? w1[2 * C1 + c] : 0.f;
    L.s[g] = live ? sc[c] : 0.f; L.t[g] = live ? sh[c] : 0.f;
  }
  return L;
}

template <bool ROUND = false>   // ROUND: the value is rounded to bf16 (and kept as fp32): what the bf16 hidden layer consumes
__device__ __forceinline__ void layer1_to_lds(const float* __restrict__ xs, const Layer1W& L, int C1, float* __restrict__ out, int ldo,
                                              int nvalid, int tid)
{
  constexpr int kRowsPerPass = kTW * 2;   // 32 lanes cover 32 channels of one row
  const int c0 = tid & 31, r0 = tid >> 5;
  const int cw = (C1 + 7) & ~7;           // C1 <= 128: at most four channel groups
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = c0 + 32 * g;
    if (c < cw) {
      float w0, wa, wb, s, t;
      if (g < 2) { w0 = L.w0[g & 1]; wa = L.wa[g & 1]; wb = L.wb[g & 1]; s = L.s[g & 1]; t = L.t[g & 1]; }
      else {   // wider first layers: loaded per tile
        const bool live = c < C1;
        w0 = live ? L.w1[c] : 0.f; wa = live ? L.w1[C1 + c] : 0.f; wb = live ? L.w1[2 * C1 + c] : 0.f;
        s = live ? L.sc[c] : 0.f; t = live ? L.sh[c] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
        const int row = rr * kRowsPerPass + r0;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
        float hv = row < nvalid ? fmaxf(fmaf(acc, s, t), 0.f) : 0.f;
        if (ROUND) hv = __uint_as_float((unsigned)to_bf16_bits(hv) << 16);
        out[row * ldo + c] = hv;
      }
    }
  }
}

// bf16 copy of the lift's output for the bf16 hidden layer (train_matmul_bf16): out16[row][c], row stride ldh elements,
// columns C1 .. K16 zero
__device__ __forceinline__ void layer1_to_lds_bf16(const float* __restrict__ xs, const Layer1W& L, int C1, unsigned short* __restrict__ out16,
                                                   int ldh, int K16, int nvalid, int tid)
{
  constexpr int kRowsPerPass = kTW * 2;
  const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = c0 + 32 * g;
    if (c < K16) {
      float w0, wa, wb, s, t;
      if (g < 2) { w0 = L.w0[g & 1]; wa = L.wa[g & 1]; wb = L.wb[g & 1]; s = L.s[g & 1]; t = L.t[g & 1]; }
      else {
        const bool live = c < C1;
        w0 = live ? L.w1[c] : 0.f; wa = live ? L.w1[C1 + c] : 0.f; wb = live ? L.w1[2 * C1 + c] : 0.f;
        s = live ? L.sc[c] : 0.f; t = live ? L.sh[c] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
        const int row = rr * kRowsPerPass + r0;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
        out16[row * ldh + c] = (row < nvalid && c < C1) ? to_bf16_bits(fmaxf(fmaf(acc, s, t), 0.f)) : (unsigned short)0;
      }
    }
  }
}

// same lift with the weights loaded from global memory on every call (pass B2 has no registers to spare for Layer1W)
__device__ __forceinline__ void layer1_to_lds_global(const float* __restrict__ xs, const float* __restrict__ w1, int C1,
                                                     const float* __restrict__ sc, const float* __restrict__ sh,
                                                     float* __restrict__ out, int ldo, int nvalid, int tid)
{
  constexpr int kRowsPerPass = kTW * 2;
  const int c0 = tid & 31, r0 = tid >> 5;
  const int cw = (C1 + 7) & ~7;
  for (int c = c0; c < cw; c += 32) {
    const bool live = c < C1;
    const float w0 = live ? w1[c] : 0.f, wa = live ? w1[C1 + c] : 0.f, wb = live ? w1[2 * C1 + c] : 0.f;
    const float s = live ? sc[c] : 0.f, t = live ? sh[c] : 0.f;
#pragma unroll
    for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
      const int row = rr * kRowsPerPass + r0;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
      const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
      out[row * ldo + c] = row < nvalid ? fmaxf(fmaf(acc, s, t), 0.f) : 0.f;
    }
  }
}

// the same lift written straight as a bf16 tile (row stride ldh elements, columns C1 .. K16 zero): pass B2 in bf16 mode consumes h1
// only as the A operand of the bf16 hidden layer (and for its column sums), so the fp32 tile + conversion pass + barrier are not needed
__device__ __forceinline__ void layer1_to_lds_bf16_global(const float* __restrict__ xs, const float* __restrict__ w1, int C1,
                                                          const float* __restrict__ sc, const float* __restrict__ sh,
                                                          unsigned short* __restrict__ out16, int ldh, int K16, int nvalid, int tid,
                                                          float (*csum)[4] = nullptr)
{
  // csum: the thread's share of the tile's column sums of the ROUNDED h1 -- columns c0 + 32 j, rows r0 + 8 rr -- added up as the values
  // are produced (a separate pass over the tile was 16 dependent LDS reads per thread, 1.5 k cycles per tile in pass B2)
  constexpr int kRowsPerPass = kTW * 2;
  const int c0 = tid & 31, r0 = tid >> 5;
  for (int c = c0; c < K16; c += 32) {
    const bool live = c < C1;
    const float w0 = live ? w1[c] : 0.f, wa = live ? w1[C1 + c] : 0.f, wb = live ? w1[2 * C1 + c] : 0.f;
    const float s = live ? sc[c] : 0.f, t = live ? sh[c] : 0.f;
    float cs = 0.f;
#pragma unroll
    for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
      const int row = rr * kRowsPerPass + r0;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
      const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
      const unsigned short hb = (row < nvalid && live) ? to_bf16_bits(fmaxf(fmaf(acc, s, t), 0.f)) : (unsigned short)0;
      out16[row * ldh + c] = hb;
      cs += __uint_as_float((unsigned)hb << 16);
    }
    if (csum) (*csum)[(c - c0) >> 5] = cs;
  }
}

// fp32 lift from the LDS parameter table par[5][64] (see layer1_to_lds_bf16_par; pass B2, fp32, shipped widths)
__device__ __forceinline__ void layer1_to_lds_par(const float* __restrict__ xs, const float* __restrict__ par, float* __restrict__ out, int ldo, int nvalid, int tid)
{
  constexpr int kRowsPerPass = kTW * 2;
  const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = c0 + 32 * g;
    const float w0 = par[c], wa = par[64 + c], wb = par[128 + c], s = par[192 + c], t = par[256 + c];
#pragma unroll
    for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
      const int row = rr * kRowsPerPass + r0;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
      const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
      out[row * ldo + c] = row < nvalid ? fmaxf(fmaf(acc, s, t), 0.f) : 0.f;
    }
  }
}

// the same lift with the thread's weights / scale / shift read from an LDS table par[5][64] = {w0, wa, wb, scale, shift} (C1 = 64) that the
// workgroup filled once per cloud: pass B2 has neither the registers for Layer1W nor the time for five global loads per tile
__device__ __forceinline__ void layer1_to_lds_bf16_par(const float* __restrict__ xs, const float* __restrict__ par, unsigned short* __restrict__ out16,
                                                       int ldh, int nvalid, int tid, float (*csum)[4])
{
  constexpr int kRowsPerPass = kTW * 2;
  const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = c0 + 32 * g;
    const float w0 = par[c], wa = par[64 + c], wb = par[128 + c], s = par[192 + c], t = par[256 + c];
    float cs = 0.f;
#pragma unroll
    for (int rr = 0; rr < kTT / kRowsPerPass; ++rr) {
      const int row = rr * kRowsPerPass + r0;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
      const float acc = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0));
      const unsigned short hb = row < nvalid ? to_bf16_bits(fmaxf(fmaf(acc, s, t), 0.f)) : (unsigned short)0;
      out16[row * ldh + c] = hb;
      cs += __uint_as_float((unsigned)hb << 16);
    }
    (*csum)[g] = cs;
  }
}

__device__ __forceinline__ int acc_row(int m, int r, int lane) { return m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- bf16 operands (BASELINE.json configs[2]: "training ... bf16 with grad step"): the 128 -> C3 lift, 90 % of the
// training FLOPs, on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); everything else stays fp32 -------------------------

// bf16 weight image: Wh[t][ct][kg][lane][8] = sign(gamma_t[c]) * W[16 kg + 8 (lane>>5) + s][c],  c = 32 ct + (lane&31)
// (one 16-byte fragment per lane per MFMA).  The sign of the following BatchNorm's gamma is folded into the column
// (exact), so that the kernel's accumulator is already sgn * (z - bias): the pooled extreme is a plain max.
// (packed by pack_bf16_jobs_kernel, kernels_train_bwd.h: up to nine images per launch)

// acc[m] = A[rows 32 m.., :16 KG] * W tile, A a bf16 LDS tile with row stride lda (elements), one bf16x8 read per MFMA
template <int MR, bool CLEAR = true>
__device__ __forceinline__ void mfma_rows_bf16(const unsigned short* __restrict__ A, int lda, const bf16x8* __restrict__ Wh,
                                               int KG, int lane, f32x16 (&acc)[MR])
{
  if (CLEAR) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  }
  const unsigned short* arow = A + (lane & 31) * lda + (lane >> 5) * 8;
  bf16x8 bcur = Wh[lane];
  for (int kg = 0; kg < KG; ++kg) {
    const bf16x8 bnext = Wh[min(kg + 1, KG - 1) * 64 + lane];
    bf16x8 av[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) av[m] = *reinterpret_cast<const bf16x8*>(arow + m * 32 * lda + kg * 16);
#ifdef ALIGNNET_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bcur, acc[m], 0, 0, 0);
#ifdef ALIGNNET_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    bcur = bnext;
  }
}

// same product with every weight fragment requested up front (KG <= 8): for the short bf16 products of pass B2 (two MFMAs = 64
// cycles per k-group) a one-deep lookahead leaves an L2 round trip exposed per k-group
template <int MR, bool CLEAR = true>
__device__ __forceinline__ void mfma_rows_bf16_all(const unsigned short* __restrict__ A, int lda, const bf16x8* __restrict__ Wh,
                                                   int KG, int lane, f32x16 (&acc)[MR])
{
  if (CLEAR) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  }
  bf16x8 b[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) b[kg] = Wh[min(kg, KG - 1) * 64 + lane];
  const unsigned short* arow = A + (lane & 31) * lda + (lane >> 5) * 8;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
    if (kg < KG) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + m * 32 * lda + kg * 16), b[kg], acc[m], 0, 0, 0);
    }
}

// (phase 1, the statistics of z1, comes from the cloud's nine moments of x': pn_moments_kernel in kernels_train_dgcnn.h)

// ---------------------------------------------------------------------------------
// phases 2 and 3 (one workgroup of 8 waves per cloud).
// Per-lane running sums live in lane-owned global slots ("slices", L2-resident, read-modify-write once per
// tile, no atomics): every accumulator element has exactly one owner lane, so the result is deterministic.
// Slice layout: part[(cloud * S + slice) * n + i]; the two half-waves of a column are slices 0/1.
// ---------------------------------------------------------------------------------
// GIVEN (phase 3, fp32): the tile of hidden features is read from h2_store (the DGCNN branch's pooled edge features,
// kernels_train_dgcnn.h) instead of being recomputed from xyz; column sums and the store belong to the producer.
template <int PHASE, bool BF16 = false, bool GIVEN = false, int C1T = 0, int C2T = 0>   // C1T / C2T: compile-time widths, see train_bwd_b2
__global__ __launch_bounds__(kTW * 64, 2) void train_fwd_phase23(const TrainFwdArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, pparts = PHASE == 2 ? a.parts : a.parts3, cloud = vcloud / pparts, part = vcloud - cloud * pparts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const XForm XF = xform_load(xf);
  float* xs = smem;
  float* buf0 = smem + kTT * 4;
  const int kC1 = C1T ? C1T : a.C1, kC2 = C2T ? C2T : a.C2;
  const int ld0 = C1T ? C1T + 4 : a.ld[0], ld1 = C2T ? C2T + 4 : a.ld[1];
  float* buf1 = buf0 + kTT * ld0;
  // bf16 mode keeps h2 only as bf16: row-major (A operand of the lift, row stride C2 + 8 elements = 4 dwords mod 64 at
  // C2 = 128) and transposed [channel][row] (both operands of the Gram, row stride kTT + 8); no fp32 tile
  const int K16 = (kC2 + 15) & ~15, ldh = K16 + 8;
  constexpr int ldT = kTT + 8;
  unsigned short* buf1h = reinterpret_cast<unsigned short*>(buf1);
  unsigned short* bufT = buf1h + kTT * ldh;
  const int KG2 = (kC1 + 7) >> 3, CT2 = (kC2 + 31) >> 5;
  const int KG3 = (kC2 + 7) >> 3, CT3 = (a.C3 + 31) >> 5;
  const int nt_all = (a.N + kTT - 1) / kTT, tile0 = part * nt_all / pparts, ntiles = (part + 1) * nt_all / pparts;   // this workgroup's tiles [tile0, ntiles) (phase 3: all of them)
  float* my_ext = PHASE == 3 ? a.ext + ((size_t)vcloud * 2 + half) * a.C3 : nullptr;
  int* my_idx = PHASE == 3 ? a.idx + ((size_t)vcloud * 2 + half) * a.C3 : nullptr;
  float* my_gram = PHASE == 3 ? a.gram_part + (size_t)vcloud * kC2 * kC2 : nullptr;

  // Two workgroups share a CU and run identical per-tile timelines; started together they stay in lockstep and
  // their non-MFMA phases coincide.  Stagger the second resident wave of workgroups by about half a tile.
  if (PHASE == 3 && (ALN_ABL(a.dbg, 16)) && ((blockIdx.x / 256) & 1)) {
    const int reps = ALN_ABL(a.dbg >> 8, 0xff);
    for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
  }

  constexpr int kGramSlots = 3;               // ceil(10 upper blocks of a 128 x 128 Gram / 4 waves)
  const int nblk = CT2 * (CT2 + 1) / 2;
  f32x16 gacc[BF16 ? kGramSlots : 1];
  if (BF16) {
#pragma unroll
    for (int q = 0; q < kGramSlots; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
  }

  const Layer1W l1w = layer1_load(a.w1, kC1, a.sc1 + tower * kC1, a.sh1 + tower * kC1, tid);
  constexpr int kBfSlots = 8;                 // channel tiles wave, wave + 4, ... of the lift: C3 <= 1024
  float rbe[BF16 ? kBfSlots : 1];
  int rbi[BF16 ? kBfSlots : 1];
  if (BF16) {
#pragma unroll
    for (int q = 0; q < kBfSlots; ++q) { rbe[q] = -INFINITY; rbi[q] = 0; }
  }

  // layer-2 items of this wave (static slots): batch-stat scale / shift and the running sums stay in registers for the whole cloud
  // (read / read-modify-written per tile they were dependent L2 round trips inside every tile: ~3 k of its 6.3 k layer-2 cycles)
  // (phase 2 only -- in the fp32 phase 3 the hand-issued weight stream must stay spill-free, and these eight registers tip it over)
  constexpr bool kRegSums = PHASE == 2;   // (the bf16 phase 3 spills 27 registers with them and its layer-2 part gets slower: 6.3 k -> 9.8 k cycles per tile)
  float l2sc[2], l2sh[2];
  double l2s[2] = {0.0, 0.0}, l2ss[2] = {0.0, 0.0};
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
    const int col = (min(wave + q2 * kTW, CT2 * 2 - 1) >> 1) * 32 + (lane & 31);
    const bool live = kRegSums && PHASE == 3 && !GIVEN && col < kC2;
    l2sc[q2] = live ? a.sc2[tower * kC2 + col] : 0.f;
    l2sh[q2] = live ? a.sh2[tower * kC2 + col] : 0.f;
  }
  double gcs[4] = {0.0, 0.0, 0.0, 0.0};   // GIVEN && BF16: column sums of the rounded features, this thread's four columns
  // phase 2: the next tile's points are requested while this tile computes (their round trip sat in front of every tile's first barrier);
  // phase 3 has no register to spare for it (see kRegSums)
  constexpr bool kPfPts = PHASE == 2 && !GIVEN;
  TilePoint npt = {0.f, 0.f, 0.f};
  if (kPfPts) npt = tile_point_request(pc, a.N, tile0, tid);
  for (int tile = tile0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    const bool first = tile == tile0;
    __syncthreads();   // previous tile's readers are done with xs/buf0/buf1
    P3_STAMP(0);
    if (GIVEN && BF16) {
      // the given features (DGCNN: pooled edge features p, fp32 in HBM) rounded to bf16 on the way into the two tiles the bf16
      // MFMAs read; the column sums are those of the ROUNDED values (centred Gram = exact covariance of one data matrix).
      // 256 % (C2 / 4) == 0 (C2 = 64 / 128): a thread's four columns are the same in every trip.
      const float* src = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;
      float ls[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = tid; i < kTT * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kC2 + q * 4);
        unsigned short hb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hb[e] = to_bf16_bits(v[e]);
          ls[e] += __uint_as_float((unsigned)hb[e] << 16);
        }
        uint2 pk; pk.x = hb[0] | ((unsigned)hb[1] << 16); pk.y = hb[2] | ((unsigned)hb[3] << 16);
        *reinterpret_cast<uint2*>(buf1h + row * ldh + q * 4) = pk;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) gcs[e] += (double)ls[e];
      // the transposed tile from the row-major one: lanes along the COLUMNS (reads: consecutive 2-byte elements of a row; writes:
      // 16 bytes = eight rows of one column, row stride 36 dwords: conflict-free).  Written element by element from the load loop --
      // lanes along q, 144 q dwords apart -- the 2-byte stores fell into four banks: 67 % of this kernel's LDS cycles were conflicts
      // (profiles/r03_train_dgcnn_bf16_pmc_by_kernel.json, first pass).
      __syncthreads();
      for (int i = tid; i < kC2 * (kTT / 8); i += kTW * 64) {
        const int c = i % kC2, ro = i / kC2;
        unsigned wv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          wv[e] = (unsigned)buf1h[(ro * 8 + 2 * e) * ldh + c] | ((unsigned)buf1h[(ro * 8 + 2 * e + 1) * ldh + c] << 16);
        *reinterpret_cast<uint4*>(bufT + c * ldT + ro * 8) = uint4{wv[0], wv[1], wv[2], wv[3]};
      }
    } else if (GIVEN) {
      const float* src = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;
      for (int i = tid; i < kTT * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kC2 + q * 4);
        *reinterpret_cast<f32x4*>(buf1 + row * ld1 + q * 4) = v;
      }
    } else {
    if (kPfPts) {
      tile_point_store(npt, XF, xs, tid);
      if (tile + 1 < ntiles) npt = tile_point_request(pc, a.N, tile + 1, tid);
    } else load_tile_xform(pc, xf, a.N, tile, xs, tid);
    __syncthreads();
    P3_STAMP(1);
    // bf16 mode: the hidden layer's operands are bf16 too (h1 tile in the buf0 region, row stride K16(C1) + 8 elements)
    const int K16a = (kC1 + 15) & ~15, ld0h = K16a + 8;
    if (BF16) layer1_to_lds_bf16(xs, l1w, kC1, reinterpret_cast<unsigned short*>(buf0), ld0h, K16a, nvalid, tid);
    else layer1_to_lds(xs, l1w, kC1, buf0, ld0, nvalid, tid);
    __syncthreads();
    P3_STAMP(2);

    // ---- layer 2: z2 = h1 W2 + b2; item = (channel tile, 32-row group): C2 = 128 -> 8 items, one per wave ----
    auto l2_item = [&](const int q2, const int item) {
      const int ct = item >> 1, rg = item & 1;
      f32x16 acc[1];
      if (BF16 && C1T != 0)   // every weight fragment requested up front: with the one-deep lookahead each of the four 32-cycle k-groups
                              // waited for its own L2 round trip (layer 2: 4.4 k cycles per tile).  (Run-time widths: eight fragments, 66 spills.)
        mfma_rows_bf16_all<1>(reinterpret_cast<const unsigned short*>(buf0) + rg * 32 * ld0h, ld0h,
                              reinterpret_cast<const bf16x8*>(a.wp2h) + (size_t)ct * (K16a >> 4) * 64, K16a >> 4, lane, acc);
      else if (BF16)
        mfma_rows_bf16<1>(reinterpret_cast<const unsigned short*>(buf0) + rg * 32 * ld0h, ld0h,
                          reinterpret_cast<const bf16x8*>(a.wp2h) + (size_t)ct * (K16a >> 4) * 64, K16a >> 4, lane, acc);
      else
        mfma_rows<1, true, true>(buf0 + rg * 32 * ld0, ld0, reinterpret_cast<const f32x4*>(a.wp2) + (size_t)ct * KG2 * 64, KG2, lane, acc);
      const int col = ct * 32 + (lane & 31);
      const bool live = col < kC2;
      if (PHASE == 2) {
        const float bias = live ? a.b2[col] : 0.f;
        // shifted sums in fp32 (shift = this lane's first value), folded into fp64 once per tile:
        //   sum z = S1 + n z0,  sum z^2 = S2 + 2 z0 S1 + n z0^2   -- no E[z^2]-E[z]^2 cancellation in fp32
        const float z0 = acc[0][0] + bias;
        float s1 = 0.f, s2 = 0.f; int cnt = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rg * 32 + acc_row(0, r, lane) < nvalid) {
            const float dlt = (acc[0][r] + bias) - z0;
            s1 += dlt; s2 = fmaf(dlt, dlt, s2); ++cnt;
          }
        {
          const double zd = (double)z0, n = (double)cnt;
          l2s[q2] += (double)s1 + n * zd;
          l2ss[q2] += (double)s2 + 2.0 * zd * (double)s1 + n * zd * zd;
        }
      } else {
        const float sc = kRegSums ? l2sc[q2] : (live ? a.sc2[tower * kC2 + col] : 0.f), sh = kRegSums ? l2sh[q2] : (live ? a.sh2[tower * kC2 + col] : 0.f);
        float lsum = 0.f;
        const bool wr = col < ((kC2 + 7) & ~7);
        if (BF16) {
          // the column sums are those of the ROUNDED values, so that the centred Gram G - s s^T / M (kernels_train_bwd.h)
          // is the exact covariance of one data matrix
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            unsigned short hb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = q * 4 + e, row = rg * 32 + acc_row(0, r, lane);
              const float h = row < nvalid ? fmaxf(fmaf(acc[0][r], sc, sh), 0.f) : 0.f;
              hb[e] = to_bf16_bits(h);
              lsum += __uint_as_float((unsigned)hb[e] << 16);
              if (col < K16) buf1h[row * ldh + col] = hb[e];
            }
            if (col < K16) {   // rows 8q + 4 half + 0..3 of this column are consecutive in the transposed tile
              uint2 pk; pk.x = hb[0] | ((unsigned)hb[1] << 16); pk.y = hb[2] | ((unsigned)hb[3] << 16);
              *reinterpret_cast<uint2*>(bufT + col * ldT + rg * 32 + q * 8 + half * 4) = pk;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rg * 32 + acc_row(0, r, lane);
            const float h = row < nvalid ? fmaxf(fmaf(acc[0][r], sc, sh), 0.f) : 0.f;
            lsum += h;
            if (wr) buf1[row * ld1 + col] = h;
          }
        }
        if (kRegSums) l2s[q2] += (double)lsum;   // phase 3: the column sum of h2
        else if (live) {
          double* cs = a.colsum_part + ((size_t)vcloud * 4 + rg * 2 + half) * kC2 + col;
          *cs = first ? (double)lsum : *cs + (double)lsum;
        }
      }
    };
    if constexpr (kRegSums) {   // C2 <= 128: at most two items per wave, static slots (their running sums live in registers)
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
        if (wave + q2 * kTW < CT2 * 2) l2_item(q2, wave + q2 * kTW);
    } else {
      for (int item = wave; item < CT2 * 2; item += kTW) l2_item(0, item);
    }
    }   // !GIVEN
    P3_STAMP(3);
    if (PHASE == 2) continue;
    __syncthreads();
    P3_STAMP(4);

    // ---- keep h2 for the sparse (arg-max) part of the backward: coalesced rows out of the LDS tile ----
    if (GIVEN) {
    } else if (BF16 && !(ALN_ABL(a.dbg, 2))) {
      unsigned short* dst = reinterpret_cast<unsigned short*>(a.h2_store) + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c8 = kC2 >> 3;
      for (int i = tid; i < nvalid * c8; i += kTW * 64) {
        const int row = i / c8, q = i % c8;
        *reinterpret_cast<f32x4*>(dst + (size_t)row * kC2 + q * 8) = *reinterpret_cast<const f32x4*>(buf1h + row * ldh + q * 8);
      }
    } else if (!BF16 && !(ALN_ABL(a.dbg, 2))) {
      float* dst = a.h2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;
      for (int i = tid; i < nvalid * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        *reinterpret_cast<f32x4*>(dst + (size_t)row * kC2 + q * 4) = *reinterpret_cast<const f32x4*>(buf1 + row * ld1 + q * 4);
      }
    }

    P3_STAMP(5);
    // ---- Gram: G += h2^T h2 (32x32 tiles of the C2 x C2 matrix), K = the tile's rows ----
    if (BF16 && !(ALN_ABL(a.dbg, 1))) {
      // upper-triangle blocks only (the Gram is symmetric; centre_gram_kernel mirrors them), register-resident for the
      // whole cloud: a per-tile read-modify-write of the 64 KiB per-cloud matrix does not stay in L2 (512 clouds in flight)
#pragma unroll
      for (int q = 0; q < kGramSlots; ++q) {
        const int item = wave + q * kTW;
        if (item < nblk) {
          int it = 0, rem = item;
          while (rem >= CT2 - it) { rem -= CT2 - it; ++it; }
          const int jt = it + rem;
          const unsigned short* pa = bufT + (it * 32 + (lane & 31)) * ldT + half * 8;
          const unsigned short* pb = bufT + (jt * 32 + (lane & 31)) * ldT + half * 8;
#pragma unroll
          for (int kg = 0; kg < kTT / 16; ++kg)
            gacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pa + kg * 16),
                                                              *reinterpret_cast<const bf16x8*>(pb + kg * 16), gacc[q], 0, 0, 0);
        }
      }
    }
    for (int item = wave; !BF16 && item < (((ALN_ABL(a.dbg, 1)) || !a.gram_inline) ? 0 : CT2 * CT2); item += kTW) {
      const int it = item / CT2, jt = item % CT2;
      const float* pa = buf1 + half * ld1 + it * 32 + (lane & 31);
      const float* pb = buf1 + half * ld1 + jt * 32 + (lane & 31);
      float old[16];
      tile_prefetch(my_gram, kC2, it, jt, kC2, kC2, first, lane, old);
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll 8
      for (int r = 0; r < kTT; r += 2) g = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld1], pb[r * ld1], g, 0, 0, 0);
      tile_commit(my_gram, kC2, it, jt, kC2, kC2, g, lane, old);
    }

    P3_STAMP(6);
    // ---- layer 3: z3 = h2 W3 + b3: statistics + extreme of sgn*z3 over the cloud's points ----
    if (BF16) {
      // acc = sgn * (z3 - bias) (sign folded into the bf16 image).  VALU-bound epilogue, 4 ops per element: sum, sum of
      // squares, and a max over keys = the value with its low 4 mantissa bits replaced by the accumulator register
      // number (2^-19 relative, far below the bf16 operand rounding), so value and row come out of one v_max3_f32 chain.
      // The per-(channel, lane) running sums / extreme / index stay in registers across the cloud's tiles (fp32 sums of
      // <= N/2 values per lane: 1e-6 relative, below the operand rounding) and are written once: re-reading and
      // re-writing those 48 KiB per cloud for every tile was the same L2-overflowing traffic as the Gram's.
      const int KG16 = K16 >> 4;
      const bf16x8* wimg = reinterpret_cast<const bf16x8*>(a.wp3h) + (size_t)tower * CT3 * KG16 * 64;
      // One continuous weight stream over the wave's channel tiles, kRing fragments deep (KG16 % kRing == 0): a bf16 k-group is
      // two MFMAs = 64 cycles, so the one-deep lookahead of mfma_rows_bf16 left most of every L2 round trip exposed (3.1 k
      // cycles per channel tile against 0.5 k of MFMA); the ring keeps loading through the epilogues.
      constexpr int kRing = 4;
      bf16x8 ring[kRing];
      const bf16x8* wnext = wimg + (size_t)wave * KG16 * 64 + lane;      // next fragment to request
      int knext = 0, left = wave < CT3 ? ((CT3 - wave + kTW - 1) / kTW) * KG16 : 0;   // fragments of this wave not yet requested
      auto request = [&](bf16x8& dst) {
        dst = *wnext;
        if (left > 1) {
          --left;
          wnext += 64;
          if (++knext == KG16) { knext = 0; wnext += (size_t)(kTW - 1) * KG16 * 64; }
        }
      };
      if (wave < CT3) {
#pragma unroll
        for (int j = 0; j < kRing; ++j) request(ring[j]);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < kBfSlots; ++q) {
        const int ct = wave + q * kTW;
        if (ct < CT3 && !(ALN_ABL(a.dbg, 8))) {
          f32x16 acc[2];
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
          {
            const unsigned short* arow = buf1h + (lane & 31) * ldh + (lane >> 5) * 8;
            for (int kg0 = 0; kg0 < KG16; kg0 += kRing) {
#pragma unroll
              for (int j = 0; j < kRing; ++j) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(arow + (kg0 + j) * 16);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow + 32 * ldh + (kg0 + j) * 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ring[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ring[j], acc[1], 0, 0, 0);
                request(ring[j]);
                asm volatile("" ::: "memory");
              }
            }
          }
          // (sum z3 and sum z3^2 are not accumulated here: z3 is linear in h2, so they follow from the column sums and the Gram
          //  of h2 that this pass produces anyway -- stat3_pool_finish_kernel; the epilogue is the key max alone)
          float mx[2] = {-INFINITY, -INFINITY};
          if (nvalid == kTT) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const float k0 = __uint_as_float((__float_as_uint(acc[m][r]) & ~15u) | (unsigned)r);
                const float k1 = __uint_as_float((__float_as_uint(acc[m][r + 1]) & ~15u) | (unsigned)(r + 1));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx[m]) : "v"(mx[m]), "v"(k0), "v"(k1));
              }
          } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const bool ok = acc_row(m, r, lane) < nvalid;
                const float k = ok ? __uint_as_float((__float_as_uint(acc[m][r]) & ~15u) | (unsigned)r) : -INFINITY;
                asm("v_max_f32 %0, %1, %2" : "=v"(mx[m]) : "v"(mx[m]), "v"(k));
              }
          }
          const int msel = mx[1] > mx[0];   // near-ties resolve to the lower row block
          const float cand = msel ? mx[1] : mx[0];
          if (cand > rbe[q]) { rbe[q] = cand; rbi[q] = tile * kTT + acc_row(msel, (int)(__float_as_uint(cand) & 15u), lane); }
        }
      }
      P3_STAMP(7);
      continue;
    }
    for (int ct = wave; ct < CT3; ct += kTW) {
      const int col = ct * 32 + (lane & 31);
      const bool live = col < a.C3;
      const float sg = live ? a.sgn3[tower * a.C3 + col] : 1.f;
      float be = (first || !live) ? -INFINITY : my_ext[col];
      int bi = (first || !live) ? 0 : my_idx[col];
      asm volatile("" ::: "memory");   // keep the two loads above the MFMA loop (see tile_prefetch)
      f32x16 acc[2];
      mfma_rows<2, true, true>(buf1, ld1, reinterpret_cast<const f32x4*>(a.wp3) + (size_t)ct * KG3 * 64, KG3, lane, acc);
      // extreme of sgn * (z3 - bias) and its row.  The batch statistics of z3 are not accumulated here: z3 is linear in h2, so sum z3
      // and sum z3^2 follow from the column sums and the Gram of h2 (stat3_pool_finish_kernel) -- the shifted fp32 sums, their fp64
      // fold and the read-modify-write of a [C3][2] double slice per tile were half of this epilogue.
      if (nvalid == kTT) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[m][r] * sg;
            if (v > be) { be = v; bi = tile * kTT + acc_row(m, r, lane); }
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = acc_row(m, r, lane);
            const float v = acc[m][r] * sg;
            if (row < nvalid && v > be) { be = v; bi = tile * kTT + row; }
          }
      }
      if (live && !(ALN_ABL(a.dbg, 4))) { my_ext[col] = be; my_idx[col] = bi; }
    }
  }
  if (!GIVEN && kRegSums) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int item = wave + q2 * kTW;
      if (item < CT2 * 2) {
        const int ct = item >> 1, rg = item & 1, col = ct * 32 + (lane & 31);
        if (col < kC2) {
          if (PHASE == 2) {
            double* st = a.stat_part + (((size_t)vcloud * 4 + rg * 2 + half) * kC2 + col) * 2;   // slice (rg, half) of this workgroup
            st[0] = l2s[q2]; st[1] = l2ss[q2];
          } else {
            a.colsum_part[((size_t)vcloud * 4 + rg * 2 + half) * kC2 + col] = l2s[q2];
          }
        }
      }
    }
  }
  if (PHASE == 3 && BF16 && GIVEN) {   // colsum_part [cloud][256 / (C2 / 4) row groups][C2]
    const int c4 = kC2 >> 2, q = tid % c4, g = tid / c4, slices = (kTW * 64) / c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.colsum_part[((size_t)vcloud * slices + g) * kC2 + q * 4 + e] = gcs[e];
  }
  if (PHASE == 3 && BF16) {
#pragma unroll
    for (int q = 0; q < kBfSlots; ++q) {
      const int col = (wave + q * kTW) * 32 + (lane & 31);
      if (col < a.C3 && !(ALN_ABL(a.dbg, 4))) { my_ext[col] = rbe[q]; my_idx[col] = rbi[q]; }
    }
#pragma unroll
    for (int q = 0; q < kGramSlots; ++q) {
      const int item = wave + q * kTW;
      if (item < nblk) {
        int it = 0, rem = item;
        while (rem >= CT2 - it) { rem -= CT2 - it; ++it; }
        const float zero[16] = {};
        tile_commit(my_gram, kC2, it, it + rem, kC2, kC2, gacc[q], lane, zero);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Phase 2 without the hidden layer: z2 = h1 W2 + b2 is linear in h1, so its batch statistics follow from the column sums
// s1 = sum h1 and the Gram G1 = sum h1^T h1 of the tower (sum z2_c = s1 . w_c + M b_c, sum z2_c^2 = w_c^T G1 w_c + 2 b_c s1 . w_c +
// M b_c^2) -- three 32 x 32 Gram blocks per tile instead of the 64 x 128 product and its statistics epilogue, and the same s1 / G1
// are what the backward needs for the layer-2 weight gradient (pass B2 no longer sums h1, pass B1 no longer accumulates the Gram).
// The per-cloud Gram stays in fp32 registers, clouds are added in fp64: the variance comes out to ~1e-7 relative
// (tools note in DESIGN.md).  BF16: h1 is rounded to bf16 first -- the statistics are those of the bf16 product exactly.
// grid 2B, block 4 waves; LDS xs | X [64][ld0]
// ---------------------------------------------------------------------------------
struct Gram1Args {
  const float* pcs[2]; const float* xform; int B, N, C1; int ld0;
  const float* w1; const float *sc1, *sh1;
  float* g1_part;      // [2B][C1*C1] upper blocks
  double* s1_part;     // [2B][sG][C1]
  int parts = 1;       // as TrainFwdArgs::parts: g1_part / s1_part per workgroup
};

template <bool BF16, int C1T = 0>   // C1T: compile-time width (0 = from the arguments)
__global__ __launch_bounds__(kTW * 64, 2) void train_fwd_gram1(const Gram1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  float* xs = smem;
  float* X = smem + kTT * 4;
  const int ld0 = C1T ? C1T + 4 : a.ld0, C1 = C1T ? C1T : a.C1, CT1 = (C1 + 31) >> 5, nblk = CT1 * (CT1 + 1) / 2;
  const int nt_all = (a.N + kTT - 1) / kTT, tile0 = part * nt_all / a.parts, ntiles = (part + 1) * nt_all / a.parts;   // this workgroup's tiles [tile0, ntiles)
  const int sG = max(1, (kTW * 64) / C1);
  const Layer1W l1w = layer1_load(a.w1, C1, a.sc1 + tower * C1, a.sh1 + tower * C1, tid);
  constexpr int kSlots = 3;
  f32x16 gacc[kSlots];
  int bit[kSlots], bjt[kSlots];
#pragma unroll
  for (int q = 0; q < kSlots; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
    int it = 0, rem = wave + q * kTW;
    while (it < CT1 && rem >= CT1 - it) { rem -= CT1 - it; ++it; }
    bit[q] = it; bjt[q] = it + rem;
  }
  double s1c = 0.0;
  // The cloud's blocks in fp64: every tile's fp32 MFMA block (32 accumulation steps) is folded in and cleared.  Carried in fp32 for the whole
  // cloud an element collects N / 2 roundings at the magnitude of the running sum (tools/microbench/mfma_round.hip: sequential round-to-nearest);
  // on the dgcnn edge rows (82 k per cloud at N = 4096, dg_train_fwd) that put the step 4.2e-3 from the pinned fp64 oracle.  At N = 1024 the
  // effect is below what the fully pinned comparison resolves (profiles/r06_gramfix_ab.log); folded per tile the chain length no longer grows with N.
  double gd[kSlots][16];
#pragma unroll
  for (int q = 0; q < kSlots; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) gd[q][r] = 0.0;
  for (int tile = tile0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    __syncthreads();
    load_tile_xform(pc, xf, a.N, tile, xs, tid);
    __syncthreads();
    layer1_to_lds<BF16>(xs, l1w, C1, X, ld0, nvalid, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSlots; ++q)
      if (wave + q * kTW < nblk) {
        const float* pa = X + half * ld0 + bit[q] * 32 + (lane & 31);
        const float* pb = X + half * ld0 + bjt[q] * 32 + (lane & 31);
#pragma unroll 8
        for (int r = 0; r < kTT; r += 2) gacc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld0], pb[r * ld0], gacc[q], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) { gd[q][r] += (double)gacc[q][r]; gacc[q][r] = 0.f; }
      }
    if (tid < sG * C1) {   // column sums of h1: sG row groups x C1 columns (rows past nvalid are zero)
      const int c = tid % C1, g = tid / C1;
      float sm = 0.f;
      for (int r = g; r < kTT; r += sG) sm += X[r * ld0 + c];
      s1c += (double)sm;
    }
  }
#pragma unroll
  for (int q = 0; q < kSlots; ++q)
    if (wave + q * kTW < nblk) {
      const float zero[16] = {};
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[q][r] = (float)gd[q][r];
      tile_commit(a.g1_part + (size_t)vcloud * C1 * C1, C1, bit[q], bjt[q], C1, C1, gacc[q], lane, zero);
    }
  if (tid < sG * C1) a.s1_part[(size_t)vcloud * sG * C1 + tid] = s1c;
}

// (sum z2, sum (z2 - mean)^2) per tower and channel from the reduced s1 [2][C1] and G1 [2][C1*C1] (upper 32 x 32 blocks valid), fp64.
// round_w: the hidden layer runs on bf16 operands -- W2 as rounded.  grid (C2, 2), block 256.  out: [2][C2][2] doubles.
// (thread 0 returns true with the channel's (sum z2, centred sum of squares) in *o0 / *o1)
__device__ __forceinline__ bool stat2_from_gram_body(const double* __restrict__ G1, const double* __restrict__ s1, const float* __restrict__ W2,
                                                     const float* __restrict__ b2, int C1, int C2, double M, int round_w, int c, int t, double* o0, double* o1)
{
  __shared__ double red[4][2];
  const int tid = threadIdx.x;
  auto wv = [&](int i) { const float w = W2[(size_t)i * C2 + c]; return (double)(round_w ? __uint_as_float((unsigned)to_bf16_bits(w) << 16) : w); };
  // CENTRED form: w^T (G - s s^T / M) w.  (Uncentred -- sum z^2 - (sum z)^2 / M afterwards -- the two terms agree to all but a few digits for a
  // channel whose mean is tens of its deviation; the fp64 reduction's totals are read unrounded for the same reason.)
  const double* G = G1 + (size_t)t * C1 * C1;
  const double* sv = s1 + (size_t)t * C1;
  const double invM = 1.0 / M;
  double q = 0.0, sw = 0.0;
  for (int e = tid; e < C1 * C1; e += 256) {
    const int i = e / C1, j = e % C1;
    const double g = ((i >> 5) <= (j >> 5) ? G[(size_t)i * C1 + j] : G[(size_t)j * C1 + i]) - sv[i] * (sv[j] * invM);
    q += wv(i) * g * wv(j);
  }
  for (int i = tid; i < C1; i += 256) sw += sv[i] * wv(i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { q += __shfl_xor(q, o); sw += __shfl_xor(sw, o); }
  if ((tid & 63) == 0) { red[tid >> 6][0] = q; red[tid >> 6][1] = sw; }
  __syncthreads();
  if (tid == 0) {
    const double Q = red[0][0] + red[1][0] + red[2][0] + red[3][0], S = red[0][1] + red[1][1] + red[2][1] + red[3][1], bb = (double)b2[c];
    *o0 = S + M * bb;          // sum z2
    *o1 = Q;                   // sum (z2 - mean)^2: the centred quadratic form
    return true;
  }
  return false;
}
__global__ __launch_bounds__(256) void stat2_from_gram_kernel(const double* __restrict__ G1, const double* __restrict__ s1, const float* __restrict__ W2,
                                                              const float* __restrict__ b2, int C1, int C2, double M, int round_w,
                                                              double* __restrict__ out)
{
  double v0, v1;
  if (stat2_from_gram_body(G1, s1, W2, b2, C1, C2, M, round_w, blockIdx.x, blockIdx.y, &v0, &v1)) {
    out[((size_t)blockIdx.y * C2 + blockIdx.x) * 2] = v0;
    out[((size_t)blockIdx.y * C2 + blockIdx.x) * 2 + 1] = v1 + v0 * (v0 / M);   // stat_finish_kernel takes (sum, sum of squares): fp64 carries the 1e-16 (mean / deviation)^2 this costs
  }
}

// ---------------------------------------------------------------------------------
// Gram of the stored fp32 h2 (fp32 training): G[cloud] = sum over the cloud's rows of h2^T h2, upper 32 x 32 blocks only
// (centre_gram_kernel mirrors them).  The blocks stay in registers for the whole cloud -- accumulating them per tile
// inside phase 3 was a read-modify-write of 64 KiB per cloud that does not stay in L2 with 512 clouds in flight.
// One workgroup (4 waves) per cloud, <= 3 blocks per wave (C2 <= 128), tiles double-buffered through LDS.
// ---------------------------------------------------------------------------------
template <int C2T = 0>   // C2T: compile-time width (0 = from the argument)
__global__ __launch_bounds__(kTW * 64) void gram_h2_kernel(const float* __restrict__ h2, int N, int C2rt, float* __restrict__ gram_part)
{
  const int C2 = C2T ? C2T : C2rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.x;
  const int ld = C2 + 4, CT2 = (C2 + 31) >> 5, nblk = CT2 * (CT2 + 1) / 2, c4 = C2 >> 2;
  const int ntiles = (N + kTT - 1) / kTT;
  const float* src = h2 + (size_t)cloud * N * C2;
  constexpr int kSlots = 3;
  f32x16 gacc[kSlots];
  int bit[kSlots], bjt[kSlots];
#pragma unroll
  for (int q = 0; q < kSlots; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
    int it = 0, rem = wave + q * kTW;
    while (it < CT2 && rem >= CT2 - it) { rem -= CT2 - it; ++it; }
    bit[q] = it; bjt[q] = it + rem;
  }
  // the next tile is requested into registers (all loads in flight at once) BEFORE this tile's MFMAs and written to the
  // other LDS buffer after them: a load-then-store loop serialised eight HBM round trips per tile
  constexpr int kRegs = (kTT * 128 / 4) / (kTW * 64);   // f32x4 per thread at C2 = 128
  f32x4 nxt[kRegs];
  auto request_tile = [&](int tile) {
    const int nvalid = min(kTT, N - tile * kTT);
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      const int i = tid + u * kTW * 64, row = i / c4, q = i % c4;
      nxt[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < kTT * c4 && row < nvalid) nxt[u] = *reinterpret_cast<const f32x4*>(src + ((size_t)tile * kTT + row) * C2 + q * 4);
    }
  };
  auto store_tile = [&](float* buf) {
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      const int i = tid + u * kTW * 64, row = i / c4, q = i % c4;
      if (i < kTT * c4) *reinterpret_cast<f32x4*>(buf + row * ld + q * 4) = nxt[u];
    }
  };
  request_tile(0);
  store_tile(smem);
  for (int tile = 0; tile < ntiles; ++tile) {
    float* cur = smem + (tile & 1) * kTT * ld;
    __syncthreads();   // cur is complete; the other buffer's readers (tile - 1) are done
    if (tile + 1 < ntiles) request_tile(tile + 1);
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      if (wave + q * kTW < nblk) {
        const float* pa = cur + half * ld + bit[q] * 32 + (lane & 31);
        const float* pb = cur + half * ld + bjt[q] * 32 + (lane & 31);
#pragma unroll 8
        for (int r = 0; r < kTT; r += 2) gacc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld], pb[r * ld], gacc[q], 0, 0, 0);
      }
    }
    if (tile + 1 < ntiles) store_tile(smem + ((tile + 1) & 1) * kTT * ld);
  }
#pragma unroll
  for (int q = 0; q < kSlots; ++q)
    if (wave + q * kTW < nblk) {
      const float zero[16] = {};
      tile_commit(gram_part + (size_t)cloud * C2 * C2, C2, bit[q], bjt[q], C2, C2, gacc[q], lane, zero);
    }
}

// ---------------------------------------------------------------------------------
// statistics finish: partials [B clouds of one tower] -> mean / biased variance (tf.nn.moments),
// batch-stat scale/shift for the recompute passes, EMA shadows (utils/tf_util.py:476-485),
// and (last layer) the pooled feature  relu(gamma*inv*(z* - mean) + beta).
// grid: (ceil(C/256), 2 towers)
// ---------------------------------------------------------------------------------
struct StatFinishArgs {
  const double* part;        // [2B * slices][C][2]
  int B, C, slices; double count;    // rows per tower = B*N
  const float* bias;         // [C] shared
  const float* beta[2]; const float* gamma[2];
  float* mov_mean[2]; float* mov_var[2];
  float bn_decay; int update_ema;
  float* mean; float* var;   // [2][C] batch statistics (kept for the backward)
  float* scale; float* shift;   // [2][C]: y = acc*scale + shift  (acc = z - bias)
  float* sgn;                // [2][next_C] or null: sign(gamma) of the NEXT layer (needed by phase 3 before its own statistics)
  const float* next_gamma[2]; int next_C;
  float* rstd; float* k;     // [2][C]: rsqrt(var+eps) and gamma*rsqrt(var+eps) (backward passes)
  double* totals_out = nullptr;   // sync_bn: only reduce the partials to [2][C][2] (sum, sum of squares) here; after the all-reduce over the
                                  // ranks a second call finishes from them (part = the totals, B = slices = 1, count = the global count)
};

// (body and kernel apart: the step also runs it as one job of a merged launch, kernels_train_bwd.h)
constexpr int kSfC = 8;   // channels per workgroup (x 128 slice groups): with 32 x 32 a C = 64 layer ran on four workgroups walking 32-deep chains of loads
__device__ __forceinline__ void stat_finish_body(const StatFinishArgs& a, int bx, int t, int gx)   // grid (gx = ceil(C / kSfC), 2), block 1024
{
  __shared__ double red[1024 / kSfC][kSfC][2];
  constexpr int kG = 1024 / kSfC;
  const int cl = threadIdx.x % kSfC, g = threadIdx.x / kSfC, c = bx * kSfC + cl;
  if (a.sgn)   // sign(gamma) of the next layer, spread over this launch's threads
    for (int i = bx * 1024 + threadIdx.x; i < a.next_C; i += gx * 1024) a.sgn[t * a.next_C + i] = a.next_gamma[t][i] >= 0.f ? 1.f : -1.f;
  const int S = a.B * a.slices;
  // the finishing thread's parameters travel with the partials (loaded behind the barrier they were a round trip of their own)
  const int cq = min(c, a.C - 1);
  float f_gamma = 0.f, f_bias = 0.f, f_beta = 0.f, f_mm = 0.f, f_mv = 0.f;
  if (g == 0 && !a.totals_out) {
    f_gamma = a.gamma[t][cq]; f_bias = a.bias[cq]; f_beta = a.beta[t][cq];
    if (a.update_ema) { f_mm = a.mov_mean[t][cq]; f_mv = a.mov_var[t][cq]; }
  }
  double s = 0.0, ss = 0.0;
  if (c < a.C)
    for (int b = g; b < S; b += kG * 4) {   // four slices per trip: eight independent loads in flight
      double v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* p = a.part + ((size_t)(t * S + min(b + u * kG, S - 1)) * a.C + c) * 2;
        v0[u] = p[0]; v1[u] = p[1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (b + u * kG < S) { s += v0[u]; ss += v1[u]; }
    }
  red[g][cl][0] = s; red[g][cl][1] = ss;
  __syncthreads();
  if (g != 0 || c >= a.C) return;
  s = 0.0; ss = 0.0;
  for (int q = 0; q < kG; ++q) { s += red[q][cl][0]; ss += red[q][cl][1]; }
  if (a.totals_out) { a.totals_out[((size_t)t * a.C + c) * 2] = s; a.totals_out[((size_t)t * a.C + c) * 2 + 1] = ss; return; }
  const double mean = s / a.count;
  const double var = fmax(ss / a.count - mean * mean, 0.0);
  const float mf = (float)mean, vf = (float)var;
  a.mean[t * a.C + c] = mf;
  a.var[t * a.C + c] = vf;
  const float inv = f_gamma * (1.0f / sqrtf(vf + kBnEps));
  a.scale[t * a.C + c] = inv;
  a.shift[t * a.C + c] = (f_bias - mf) * inv + f_beta;
  a.rstd[t * a.C + c] = 1.0f / sqrtf(vf + kBnEps);
  a.k[t * a.C + c] = inv;
  if (a.update_ema) {
    // ExponentialMovingAverage.apply: shadow -= (1 - decay) * (shadow - value)
    a.mov_mean[t][c] = f_mm - (1.f - a.bn_decay) * (f_mm - mf);
    a.mov_var[t][c] = f_mv - (1.f - a.bn_decay) * (f_mv - vf);
  }
}
__global__ __launch_bounds__(1024) void stat_finish_kernel(const StatFinishArgs a) { stat_finish_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// The hidden layer's statistics from Gram(h1) AND their finish in one launch (fp32 training; they were stat2_from_gram_kernel + stat_finish_kernel on a
// [2][C2][2] "partial" table: a launch whose 16 workgroups each read two doubles per channel).  grid (C2, 2), block 256; f.part / B / slices unused.
__global__ __launch_bounds__(256) void stat2_from_gram_finish_kernel(const double* __restrict__ G1, const double* __restrict__ s1, const float* __restrict__ W2,
                                                                     const float* __restrict__ b2, int C1, int C2, double M, int round_w, const StatFinishArgs a)
{
  const int c = blockIdx.x, t = blockIdx.y;
  if (a.sgn)   // sign(gamma) of the next layer, spread over this launch's threads
    for (int i = c * 256 + threadIdx.x; i < a.next_C; i += C2 * 256) a.sgn[t * a.next_C + i] = a.next_gamma[t][i] >= 0.f ? 1.f : -1.f;
  double s, ss;
  if (!stat2_from_gram_body(G1, s1, W2, b2, C1, C2, M, round_w, c, t, &s, &ss)) return;
  const double mean = s / a.count;
  const double var = fmax(ss / a.count, 0.0);   // (ss: the centred sum of squares)
  const float mf = (float)mean, vf = (float)var;
  a.mean[t * a.C + c] = mf;
  a.var[t * a.C + c] = vf;
  const float inv = a.gamma[t][c] * (1.0f / sqrtf(vf + kBnEps));
  a.scale[t * a.C + c] = inv;
  a.shift[t * a.C + c] = (a.bias[c] - mf) * inv + a.beta[t][c];
  a.rstd[t * a.C + c] = 1.0f / sqrtf(vf + kBnEps);
  a.k[t * a.C + c] = inv;
  if (a.update_ema) {
    a.mov_mean[t][c] -= (1.f - a.bn_decay) * (a.mov_mean[t][c] - mf);
    a.mov_var[t][c] -= (1.f - a.bn_decay) * (a.mov_var[t][c] - vf);
  }
}

__global__ void sign_kernel(const float* __restrict__ gamma, int C, float* __restrict__ sgn)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) sgn[c] = gamma[c] >= 0.f ? 1.f : -1.f;
}

// rstd[t][c] = rsqrt(var+eps), k[t][c] = gamma_t[c]*rstd
__global__ void rstd_k_kernel(const float* __restrict__ var, const float* __restrict__ g0, const float* __restrict__ g1, int C,
                              float* __restrict__ rstd, float* __restrict__ k)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  if (c >= C) return;
  const float rs = 1.0f / sqrtf(var[t * C + c] + kBnEps);
  rstd[t * C + c] = rs;
  k[t * C + c] = (t ? g1[c] : g0[c]) * rs;
}

// pooled[t,b,c] = relu(scale*(sgn*ext - bias) + shift);  also keeps zhat* = (z* - mean)*rsqrt(var+eps) and the
// final arg-extreme index (combining the two half-wave slices; first occurrence wins ties)
// Phase 3 split over `parts` workgroups per cloud (TrainFwdArgs::parts3): fold the parts' running extremes per (cloud, lane half, channel).  A part
// scanned its tiles keeping the first of equal values (strictly greater wins); folding the parts in tile order under the same rule IS the scan over
// the whole cloud -- bit-equal (ext, idx) to one workgroup per cloud.  n = 2B * 2 * C3 outputs.
static __global__ __launch_bounds__(256) void merge_ext_parts_kernel(const float* __restrict__ extp, const int* __restrict__ idxp, int parts, int C3, size_t n,
                                                                     float* __restrict__ ext, int* __restrict__ idx)
{
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= n) return;
  const size_t per = (size_t)2 * C3, cloud = i / per, rem = i - cloud * per;
  const size_t o = cloud * parts * per + rem;
  float e = extp[o]; int bi = idxp[o];
  for (int p = 1; p < parts; ++p) {
    const float v = extp[o + p * per];
    if (v > e) { e = v; bi = idxp[o + p * per]; }
  }
  ext[i] = e; idx[i] = bi;
}

struct PoolFinishArgs {
  const float* ext; const int* idx2; const float* sgn; const float* bias; const float* scale; const float* shift; const float* mean; const float* var;
  int B, C; float* pooled; long tower_stride, row_stride; float* zhat_star; int* idx;
  int ext_excludes_bias;   // bf16 mode: ext = extreme of sgn*(z - bias)
};
__device__ __forceinline__ void pool_finish_body(const PoolFinishArgs& a, unsigned bx)   // 256 threads per block, ceil(2 B C / 256) blocks
{
  const float* __restrict__ ext = a.ext; const int* __restrict__ idx2 = a.idx2; const float* __restrict__ sgn = a.sgn;
  const float* __restrict__ bias = a.bias; const float* __restrict__ scale = a.scale; const float* __restrict__ shift = a.shift;
  const float* __restrict__ mean = a.mean; const float* __restrict__ var = a.var;
  float* __restrict__ pooled = a.pooled; float* __restrict__ zhat_star = a.zhat_star; int* __restrict__ idx = a.idx;
  const int B = a.B, C = a.C, ext_excludes_bias = a.ext_excludes_bias;
  const long tower_stride = a.tower_stride, row_stride = a.row_stride;
  const size_t i = bx * (size_t)256 + threadIdx.x;
  if (i >= (size_t)2 * B * C) return;
  const int c = i % C, cloud = i / C, t = cloud >= B, b = cloud - t * B;
  const size_t h0 = ((size_t)cloud * 2) * C + c, h1 = h0 + C;
  float e = ext[h0]; int bi = idx2[h0];
  if (ext[h1] > e || (ext[h1] == e && idx2[h1] < bi)) { e = ext[h1]; bi = idx2[h1]; }
  idx[i] = bi;
  const float z = e * sgn[t * C + c] + (ext_excludes_bias ? bias[c] : 0.f);
  const float y = fmaf(ext_excludes_bias ? e * sgn[t * C + c] : z - bias[c], scale[t * C + c], shift[t * C + c]);
  pooled[t * tower_stride + b * row_stride + c] = fmaxf(y, 0.f);
  zhat_star[i] = (z - mean[t * C + c]) * (1.0f / sqrtf(var[t * C + c] + kBnEps));
}

}  // namespace alignnet
