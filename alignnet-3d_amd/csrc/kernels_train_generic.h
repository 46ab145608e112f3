// Training kernels for PointNet backbones of ANY depth (models/tp8.py:49-59 loops over `layer_sizes[1:]`), gfx950 only.
//
// The shipped dataset configs all use three conv layers with C1, C2 <= 128, and kernels_train_fwd.h / kernels_train_bwd.h are
// built around exactly that shape (recompute from xyz, Gram identities, nothing [B*N, C]-sized but h2 in HBM).  The reference's
// own configs/default.json, however, has five-layer backbones ([64, 64, 64, 128, 1024]), and tp8.py accepts any list.  This file
// is the general path for such stages: layer by layer, with the pre-BatchNorm activations Z_l kept in HBM (as TensorFlow keeps
// them: 1.3 k floats per point for default.json, 0.7 GB per stage at B = 64 -- small against 288 GB), every product on fp32 MFMA:
//
//   forward    x' = (x - c) R  ->  Z_1 = x' W_1 + b_1 (K = 3, VALU)  ->  Z_l = relu(bn(Z_{l-1})) W_l + b_l  (gen_gemm_fwd: the
//              BatchNorm + ReLU of the previous layer is applied while its rows are staged in LDS, so Y_l is never stored)
//              -> per-tile (mean, M2) partials of every Z_l, merged in fp64 (Chan) -> batch statistics, EMA (tf_util.py:455-492)
//              -> max over the points of relu(bn(Z_L)) with the arg-max row (tf_util.py:350-373).
//   backward   dY_L scattered from the pooled gradient -> per layer, last to first:
//              g = dY [y > 0];  dbeta = sum g, dgamma = sum g zhat;  dZ = k (g - dbeta/M - zhat dgamma/M)   (in place)
//              dW_l = relu(bn(Z_{l-1}))^T dZ_l   (gen_gemm_dw: per-slab partials, summed in a fixed order by reduce_slices_kernel)
//              dY_{l-1} = dZ_l W_l^T             (gen_gemm_dx)
//              first layer: per-cloud S = sum dZ_1, P = x'^T dZ_1 give dW_1 and the frame gradients gx / grot.
// Biases in front of a BatchNorm get an exactly-zero gradient, as in the specialised path.
// Row order: cloud-major, cloud = tower * B + b, row = cloud * N + n; a 64-row tile never straddles the two towers.
#pragma once
#include "kernels_infer.h"

namespace alignnet {

constexpr int kGenTile = 64;        // rows per workgroup tile (two 32-row MFMA tiles)
constexpr int kGenWaves = 4;
constexpr int kGenMaxK = 256;       // widest INPUT of an MFMA layer (LDS-resident A tile [64][K + 4])
constexpr int kGenSlab = 1024;      // rows per dW partial, times 1 .. 4 with the layer's width (gen_slab_rows)
// wide layers have workgroups to spare along C (grid z): longer slabs there mean fewer partials to write and to sum (0.14 ms per layer
// in gen_sum_partials at 1024 rows); narrow layers keep 1024 rows so that the launch still fills the chip
inline int gen_slab_rows(int C) { const int f = C / 128; return kGenSlab * (f < 1 ? 1 : f > 4 ? 4 : f); }

// ---- x' = (x - c) R for every point, [R][4] (w = 0) ------------------------------------------------------------------------------
__global__ void gen_xform_kernel(const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ xform, int B, int N,
                                 float* __restrict__ X0)
{
  const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (r >= (size_t)2 * B * N) return;
  const int cloud = (int)(r / N), n = (int)(r - (size_t)cloud * N), tower = cloud >= B, b = cloud - tower * B;
  const float* p = (tower ? p2 : p1) + ((size_t)b * N + n) * 3;
  const float* xf = xform + (size_t)cloud * 12;
  const float x = p[0] - xf[0], y = p[1] - xf[1], z = p[2] - xf[2];
  f32x4 o;
  o[0] = x * xf[3] + y * xf[6] + z * xf[9];
  o[1] = x * xf[4] + y * xf[7] + z * xf[10];
  o[2] = x * xf[5] + y * xf[8] + z * xf[11];
  o[3] = 0.f;
  reinterpret_cast<f32x4*>(X0)[r] = o;
}

// per-tile statistics partial of one column: n values around `shift` -> (mean, M2); merged in fp64 by gen_stat_finish
__device__ __forceinline__ void gen_tile_stat(float s1, float s2, float shift, int n, float* __restrict__ dst)
{
  const float inv = n > 0 ? 1.f / (float)n : 0.f;
  dst[0] = shift + s1 * inv;
  dst[1] = fmaxf(s2 - s1 * s1 * inv, 0.f);
}

// ---- layer 1: Z = x' W + b, K = 3 on the VALU.  grid (tiles, 2 towers), block 256 = 4 row groups x 64 columns ------------------
struct GenL1Args { const float* X0; const float* W; const float* bias; float* Z; float* part; int M, C, tiles; };

__global__ __launch_bounds__(256) void gen_layer1_fwd(const GenL1Args a)
{
  __shared__ float red[4][64][2];
  const int tile = blockIdx.x, tower = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  for (int c0 = 0; c0 < a.C; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < a.C;
    const float w0 = live ? a.W[c] : 0.f, w1 = live ? a.W[a.C + c] : 0.f, w2 = live ? a.W[2 * a.C + c] : 0.f, bb = live ? a.bias[c] : 0.f;
    const f32x4 x00 = reinterpret_cast<const f32x4*>(a.X0)[row0];
    const float shift = fmaf(x00[2], w2, fmaf(x00[1], w1, fmaf(x00[0], w0, bb)));   // the tile's first row: every row group uses the same shift
    float s1 = 0.f, s2 = 0.f;
    for (int r = g; r < nvalid; r += 4) {
      const f32x4 x = reinterpret_cast<const f32x4*>(a.X0)[row0 + r];
      const float z = fmaf(x[2], w2, fmaf(x[1], w1, fmaf(x[0], w0, bb)));
      if (live) a.Z[(row0 + r) * a.C + c] = z;
      const float d = z - shift;
      s1 += d; s2 = fmaf(d, d, s2);
    }
    red[g][lane][0] = s1; red[g][lane][1] = s2;
    __syncthreads();
    if (g == 0 && live) {
      s1 = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
      s2 = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
      gen_tile_stat(s1, s2, shift, nvalid, a.part + (((size_t)tower * a.tiles + tile) * a.C + c) * 2);
    }
    __syncthreads();
  }
}

// ---- Z = relu(bn(Zprev)) W + b on fp32 MFMA.  grid (tiles, 2), block 256, LDS [64][K + 4] floats ---------------------------------
struct GenGemmArgs {
  const float* Zprev; const float* scale; const float* shift;   // previous layer: [R][K], [2][K], [2][K]
  const float* Wimg; const float* bias;                         // this layer: MFMA image (K -> C), [C]
  float* Z; float* part;                                        // [R][C], [2][tiles][C][2]
  int M, K, C, tiles;
};

// stage rows [row0, row0 + 64) of the previous layer into LDS with its BatchNorm + ReLU applied (rows past nvalid: zeros)
__device__ __forceinline__ void gen_stage_act(const float* __restrict__ Zprev, const float* __restrict__ sc, const float* __restrict__ sh, size_t row0,
                                              int nvalid, int K, float* __restrict__ A, int lda, int tid)
{
  const int kq = K >> 2;   // K is a multiple of 8
  for (int i = tid; i < kGenTile * kq; i += kGenWaves * 64) {
    const int r = i / kq, q = i - r * kq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nvalid) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(Zprev + (row0 + r) * K + q * 4);
      const f32x4 s = *reinterpret_cast<const f32x4*>(sc + q * 4), t = *reinterpret_cast<const f32x4*>(sh + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(z[e], s[e], t[e]), 0.f);
    }
    *reinterpret_cast<f32x4*>(A + r * lda + q * 4) = v;
  }
}

__global__ __launch_bounds__(kGenWaves * 64) void gen_gemm_fwd(const GenGemmArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, tower = blockIdx.y;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  const int lda = a.K + 4, KG = a.K >> 3, CT = (a.C + 31) >> 5;
  gen_stage_act(a.Zprev, a.scale + tower * a.K, a.shift + tower * a.K, row0, nvalid, a.K, smem, lda, tid);
  __syncthreads();
  for (int ct = wave; ct < CT; ct += kGenWaves) {
    f32x16 acc[2];
    mfma_rows<2, true, false>(smem, lda, reinterpret_cast<const f32x4*>(a.Wimg) + (size_t)ct * KG * 64, KG, lane, acc);
    const int col = ct * 32 + (lane & 31);
    const bool live = col < a.C;
    const float bb = live ? a.bias[col] : 0.f;
    const float shift = __shfl(acc[0][0], lane & 31) + bb;   // row 0 of the tile (held by the lower half-wave)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float z = acc[m][r] + bb;
        if (row < nvalid) {
          if (live) a.Z[(row0 + row) * a.C + col] = z;
          const float d = z - shift;
          s1 += d; s2 = fmaf(d, d, s2);
        }
      }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lane < 32 && live) gen_tile_stat(s1, s2, shift, nvalid, a.part + (((size_t)tower * a.tiles + tile) * a.C + col) * 2);
  }
}

// ---- merge the tile partials (Chan, fp64), batch statistics + EMA (utils/tf_util.py:474-491).  grid (ceil(C / 64), 2), block 256 ---
struct GenStatArgs {
  const float* part; int tiles, M, C;
  const float* beta[2]; const float* gamma[2]; float* mov_mean[2]; float* mov_var[2];
  float bn_decay; int update_ema;
  float* mean; float* rstd; float* scale; float* shift;   // [2][C]: y = z * scale + shift
  // sync_bn (batch statistics over all data-parallel ranks, utils/tf_util.py:474 at the global batch): three launches with an all-reduce
  // of `tot` between them.  mode 1: tot[t][c] = this rank's sum of z; mode 2: tot holds all ranks' sum -> tot[2 C + ...] = this rank's
  // sum of squared differences from the GLOBAL mean; mode 3: tot holds both global sums -> statistics and EMA with the global count
  // M * world.  mode 0: one launch, this rank's batch.
  int mode = 0; double* tot = nullptr; double world = 1.0;
};

// (All tiles but a tower's last hold kGenTile rows, so the pairwise Chan merge -- a chain of fp64 divisions per partial, 4096 partials
//  per column at B = 256, N = 1024: 0.4 ms per launch -- is replaced by its closed form over all partials: mean = sum n_b mean_b / M,
//  M2 = sum [M2_b + n_b (mean_b - mean)^2]; two passes of independent loads, 16 tile groups per column.)
__global__ __launch_bounds__(1024) void gen_stat_finish(const GenStatArgs a)
{
  __shared__ double red[16][64];
  __shared__ double bmean[64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, t = blockIdx.y;
  const bool live = c < a.C;
  const float* base = a.part + ((size_t)t * a.tiles * a.C + (live ? c : 0)) * 2;
  auto rows_of = [&](int i) { return (double)min(kGenTile, a.M - i * kGenTile); };
  const double nglob = (double)a.M * a.world;
  double s = 0.0;
  if (live && a.mode <= 1) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    int i = g;
    for (; i + 48 < a.tiles; i += 64)
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += rows_of(i + 16 * u) * (double)base[(size_t)(i + 16 * u) * a.C * 2];
    for (; i < a.tiles; i += 16) s4[0] += rows_of(i) * (double)base[(size_t)i * a.C * 2];
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  red[g][lane] = s;
  __syncthreads();
  if (g == 0) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][lane];
    if (a.mode == 1) { if (live) a.tot[t * a.C + c] = tot; }
    else bmean[lane] = (a.mode >= 2 ? (live ? a.tot[t * a.C + c] : 0.0) : tot) / nglob;
  }
  if (a.mode == 1) return;
  __syncthreads();
  const double mean = bmean[lane];
  s = 0.0;
  if (live && a.mode != 3) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    int i = g;
    for (; i + 48 < a.tiles; i += 64)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = base + (size_t)(i + 16 * u) * a.C * 2;
        const double d = (double)p[0] - mean;
        s4[u] += (double)p[1] + rows_of(i + 16 * u) * d * d;
      }
    for (; i < a.tiles; i += 16) {
      const float* p = base + (size_t)i * a.C * 2;
      const double d = (double)p[0] - mean;
      s4[0] += (double)p[1] + rows_of(i) * d * d;
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  __syncthreads();
  red[g][lane] = s;
  __syncthreads();
  if (g != 0 || !live) return;
  double m2 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) m2 += red[q][lane];
  if (a.mode == 2) { a.tot[(2 + t) * a.C + c] = m2; return; }
  if (a.mode == 3) m2 = a.tot[(2 + t) * a.C + c];
  const double n = nglob;
  const float mf = (float)mean, vf = (float)(m2 / n);   // biased variance: mean of the squared difference from the mean (tf.nn.moments)
  const float rs = 1.0f / sqrtf(vf + kBnEps), k = a.gamma[t][c] * rs;
  a.mean[t * a.C + c] = mf;
  a.rstd[t * a.C + c] = rs;
  a.scale[t * a.C + c] = k;
  a.shift[t * a.C + c] = a.beta[t][c] - mf * k;
  if (a.update_ema) {   // ExponentialMovingAverage.apply: shadow -= (1 - decay) * (shadow - value)
    a.mov_mean[t][c] -= (1.f - a.bn_decay) * (a.mov_mean[t][c] - mf);
    a.mov_var[t][c] -= (1.f - a.bn_decay) * (a.mov_var[t][c] - vf);
  }
}

// ---- hybrid stages: h = relu(bn(Z)) materialised once for the fused tail (phase 3 / pass B2 with GIVEN features), with the cloud's
//      column sums of h (one fp64 slice per cloud: the centred Gram of the tail needs them).
//      grid (2B), block (C / 4) * (256 / (C / 4)): a thread owns four columns of every (256 / (C / 4))-th row -------------------------
__global__ __launch_bounds__(256) void gen_apply_colsum(const float* __restrict__ Z, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int B, int N, int C, float* __restrict__ H, double* __restrict__ colsum)
{
  __shared__ double red[32][128];
  const int cloud = blockIdx.x, tower = cloud >= B, c4 = C >> 2, groups = blockDim.x / c4;
  const int q = threadIdx.x % c4, g = threadIdx.x / c4;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + tower * C + q * 4), sh = *reinterpret_cast<const f32x4*>(shift + tower * C + q * 4);
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int r0 = 0; r0 < N; r0 += 8 * groups) {   // eight rows in flight per thread
    f32x4 z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + g + u * groups;
      z[u] = r < N ? *reinterpret_cast<const f32x4*>(Z + ((size_t)cloud * N + r) * C + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + g + u * groups;
      if (r < N) {
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[e] = fmaxf(fmaf(z[u][e], sc[e], sh[e]), 0.f); ls[e] += hv[e]; }
        *reinterpret_cast<f32x4*>(H + ((size_t)cloud * N + r) * C + q * 4) = hv;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += (double)ls[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[g][q * 4 + e] = s[e];
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double t = 0.0;
      for (int k = 0; k < groups; ++k) t += red[k][q * 4 + e];
      colsum[(size_t)cloud * C + q * 4 + e] = t;
    }
  }
}

// ---- max over the N points of relu(bn(Z_L)) + arg-max row.  grid (2B, ceil(C / 64)), block 256 = 4 row groups x 64 columns --------
struct GenPoolArgs { const float* Z; const float* scale; const float* shift; int B, N, C; float* pooled; long tower_stride, row_stride; int* idx; };

__global__ __launch_bounds__(256) void gen_pool_fwd(const GenPoolArgs a)
{
  __shared__ float bv[4][64];
  __shared__ int bi[4][64];
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.y * 64 + lane;
  const bool live = c < a.C;
  const float sc = live ? a.scale[tower * a.C + c] : 0.f, sh = live ? a.shift[tower * a.C + c] : 0.f;
  float best = -1.f; int at = 0;
  if (live)
    for (int n = g; n < a.N; n += 4) {
      const float y = fmaxf(fmaf(a.Z[((size_t)cloud * a.N + n) * a.C + c], sc, sh), 0.f);
      if (y > best) { best = y; at = n; }   // first maximum
    }
  bv[g][lane] = best; bi[g][lane] = at;
  __syncthreads();
  if (g != 0 || !live) return;
  for (int q = 1; q < 4; ++q)
    if (bv[q][lane] > best || (bv[q][lane] == best && bi[q][lane] < at)) { best = bv[q][lane]; at = bi[q][lane]; }
  a.pooled[tower * a.tower_stride + b * a.row_stride + c] = best;
  a.idx[(size_t)cloud * a.C + c] = at;
}

// dY_L[cloud * N + idx][c] = dP[cloud, c] into a zeroed [R][C] buffer.  one thread per (cloud, channel)
__global__ void gen_pool_bwd(const float* __restrict__ dP, long tower_stride, long row_stride, const int* __restrict__ idx, int B, int N, int C,
                             float* __restrict__ dY)
{
  const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (e >= (size_t)2 * B * C) return;
  const int cloud = (int)(e / C), c = (int)(e - (size_t)cloud * C), tower = cloud >= B, b = cloud - tower * B;
  dY[((size_t)cloud * N + idx[e]) * C + c] = dP[tower * tower_stride + b * row_stride + c];
}

// ---- BatchNorm backward, pass 1: per-tile sums of g = dY [y > 0] and g zhat.  grid (tiles, 2, ceil(C / 64)), block 256 -------------
struct GenBnBwdArgs {
  const float* Z; float* dY; const float* mean; const float* rstd; const float* scale; const float* shift;
  float* part;                      // [2][tiles][C][2]
  const float* cA; const float* cB; // pass 2: [2][C] dbeta / M, dgamma / M
  int M, C, tiles;
};

__global__ __launch_bounds__(256) void gen_bn_bwd_reduce(const GenBnBwdArgs a)
{
  __shared__ float red[4][64][2];
  const int tile = blockIdx.x, tower = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.z * 64 + lane;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  const bool live = c < a.C;
  const float mu = live ? a.mean[tower * a.C + c] : 0.f, rs = live ? a.rstd[tower * a.C + c] : 0.f;
  const float sc = live ? a.scale[tower * a.C + c] : 0.f, sh = live ? a.shift[tower * a.C + c] : 0.f;
  float s0 = 0.f, s1 = 0.f;
  if (live)
    for (int r = g; r < nvalid; r += 4) {
      const size_t e = (row0 + r) * a.C + c;
      const float z = a.Z[e];
      const float gg = fmaf(z, sc, sh) > 0.f ? a.dY[e] : 0.f;
      s0 += gg; s1 = fmaf(gg, (z - mu) * rs, s1);
    }
  red[g][lane][0] = s0; red[g][lane][1] = s1;
  __syncthreads();
  if (g == 0 && live) {
    float* p = a.part + (((size_t)tower * a.tiles + tile) * a.C + c) * 2;
    p[0] = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
    p[1] = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
  }
}

// sums of the tile partials in fp64 -> dbeta, dgamma (written into the gradient vector) and the pass-2 coefficients
struct GenBnFinArgs { const float* part; int tiles, M, C; float* dbeta[2]; float* dgamma[2]; float* cA; float* cB;
                      // sync_bn: mode 1 = this rank's (dbeta, dgamma) -> the gradient and tot[t][c][2] (all-reduced by the host), mode 2 = the
                      // coefficients from all ranks' totals and the global count M * world; mode 0 = both from this rank's sums
                      int mode = 0; double* tot = nullptr; double world = 1.0; };

__global__ __launch_bounds__(1024) void gen_bn_bwd_finish(const GenBnFinArgs a)   // grid (ceil(C / 64), 2), block 16 tile groups x 64 columns
{
  __shared__ double red[16][64][2];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, t = blockIdx.y;
  double s0 = 0.0, s1 = 0.0;
  if (a.mode == 2) {
    if (g == 0 && c < a.C) {
      a.cA[t * a.C + c] = (float)(a.tot[(t * a.C + c) * 2] / ((double)a.M * a.world));
      a.cB[t * a.C + c] = (float)(a.tot[(t * a.C + c) * 2 + 1] / ((double)a.M * a.world));
    }
    return;
  }
  if (c < a.C) {
    const float* base = a.part + ((size_t)t * a.tiles * a.C + c) * 2;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    int i = g;
    for (; i + 48 < a.tiles; i += 64)
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float* p = base + (size_t)(i + 16 * u) * a.C * 2; a0[u] += p[0]; a1[u] += p[1]; }
    for (; i < a.tiles; i += 16) { const float* p = base + (size_t)i * a.C * 2; a0[0] += p[0]; a1[0] += p[1]; }
    s0 = (a0[0] + a0[1]) + (a0[2] + a0[3]); s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  }
  red[g][lane][0] = s0; red[g][lane][1] = s1;
  __syncthreads();
  if (g != 0 || c >= a.C) return;
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { s0 += red[q][lane][0]; s1 += red[q][lane][1]; }
  a.dbeta[t][c] = (float)s0;
  a.dgamma[t][c] = (float)s1;
  if (a.mode == 1) { a.tot[(t * a.C + c) * 2] = s0; a.tot[(t * a.C + c) * 2 + 1] = s1; return; }
  a.cA[t * a.C + c] = (float)(s0 / (double)a.M);
  a.cB[t * a.C + c] = (float)(s1 / (double)a.M);
}

// pass 2, in place: dZ = k (g - dbeta / M - zhat dgamma / M)
__global__ __launch_bounds__(256) void gen_bn_bwd_apply(const GenBnBwdArgs a)
{
  const int tile = blockIdx.x, tower = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.z * 64 + lane;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  if (c >= a.C) return;
  const float mu = a.mean[tower * a.C + c], rs = a.rstd[tower * a.C + c], sc = a.scale[tower * a.C + c], sh = a.shift[tower * a.C + c];
  const float ca = a.cA[tower * a.C + c], cb = a.cB[tower * a.C + c];
  for (int r = g; r < nvalid; r += 4) {
    const size_t e = (row0 + r) * a.C + c;
    const float z = a.Z[e];
    const float gg = fmaf(z, sc, sh) > 0.f ? a.dY[e] : 0.f;
    a.dY[e] = sc * (gg - ca - (z - mu) * rs * cb);
  }
}

// ---- dYprev = dZ W^T: contraction over this layer's C outputs in chunks of 128.  grid (tiles, 2), block 256, LDS [64][132] --------
struct GenDxArgs { const float* dZ; const float* WTimg; float* dYprev; int M, K, C; };   // WTimg: MFMA image of W^T (C -> K)

__global__ __launch_bounds__(kGenWaves * 64) void gen_gemm_dx(const GenDxArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kChunk = 128, lda = kChunk + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, tower = blockIdx.y;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  const int KGt = (a.C + 7) >> 3, CT = (a.K + 31) >> 5;   // image: CT output tiles x KGt contraction groups
  f32x16 acc[2][2];   // output tiles ct = wave, wave + 4 (K <= 256), two row tiles each
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
  for (int c0 = 0; c0 < a.C; c0 += kChunk) {
    const int cw = min(kChunk, a.C - c0);            // multiple of 8
    __syncthreads();
    for (int i = tid; i < kGenTile * (kChunk / 4); i += kGenWaves * 64) {
      const int r = i / (kChunk / 4), q = i - r * (kChunk / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < nvalid && q * 4 < cw) v = *reinterpret_cast<const f32x4*>(a.dZ + (row0 + r) * a.C + c0 + q * 4);
      *reinterpret_cast<f32x4*>(smem + r * lda + q * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ct = wave + i * kGenWaves;
      if (ct < CT)
        mfma_rows<2, false, false>(smem, lda, reinterpret_cast<const f32x4*>(a.WTimg) + ((size_t)ct * KGt + (c0 >> 3)) * 64, cw >> 3, lane, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ct = wave + i * kGenWaves, col = ct * 32 + (lane & 31);
    if (ct < CT && col < a.K)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < nvalid) a.dYprev[(row0 + row) * a.K + col] = acc[i][m][r];
        }
  }
}

// image of W^T for gen_gemm_dx: contraction index = this layer's output channel, output = its input channel
__global__ void gen_pack_transposed(const float* __restrict__ W, int K, int C, float* __restrict__ Wp)   // W: [K][C] row-major
{
  const int KG = (C + 7) >> 3, CT = (K + 31) >> 5;
  const size_t total = (size_t)CT * KG * 256;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int s = idx & 3, lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kg = t % KG, ct = t / KG;
    const int k = 8 * kg + 4 * (lane >> 5) + s, c = 32 * ct + (lane & 31);   // k: contraction (a column of W), c: output (a row of W)
    Wp[idx] = (k < C && c < K) ? W[(size_t)c * C + k] : 0.f;
  }
}

// ---- dW partial of one row slab: relu(bn(Zprev))^T dZ.  grid (slabs, 2 towers, ceil(C / 64)), block 256 ---------------------------
// LDS: act [64][K] | dz [64][64].  Both MFMA operands are single floats read along a row of those tiles (conflict-free).
struct GenDwArgs {
  const float* Zprev; const float* scale; const float* shift;   // [R][K], [2][K]
  const float* dZ;                                              // [R][C]
  float* part;                                                  // [2 * slabs][K][C]
  int M, K, C, slabs, slab_rows;
};

__global__ __launch_bounds__(kGenWaves * 64) void gen_gemm_dw(const GenDwArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x, tower = blockIdx.y, c0 = blockIdx.z * 64;
  const int KT = (a.K + 31) >> 5;                     // <= 8 row tiles of the output (input channels)
  float* act = smem;
  float* dz = smem + kGenTile * a.K;
  f32x16 acc[2][2];                                   // kt = wave, wave + 4; ct = 0, 1
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int rbeg = slab * a.slab_rows, rend = min(a.M, rbeg + a.slab_rows);
  for (int t0 = rbeg; t0 < rend; t0 += kGenTile) {
    const int nvalid = min(kGenTile, rend - t0);
    const size_t row0 = (size_t)tower * a.M + t0;
    __syncthreads();
    gen_stage_act(a.Zprev, a.scale + tower * a.K, a.shift + tower * a.K, row0, nvalid, a.K, act, a.K, tid);
    for (int i = tid; i < kGenTile * 16; i += kGenWaves * 64) {
      const int r = i >> 4, q = i & 15;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < nvalid && c0 + q * 4 < a.C) v = *reinterpret_cast<const f32x4*>(a.dZ + (row0 + r) * a.C + c0 + q * 4);
      *reinterpret_cast<f32x4*>(dz + r * 64 + q * 4) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int st = 0; st < kGenTile / 2; ++st) {
      const int dr = 2 * st + (lane >> 5);
      const float b0 = dz[dr * 64 + (lane & 31)], b1 = dz[dr * 64 + 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kt = wave + i * kGenWaves;
        if (kt < KT) {
          const float av = act[dr * a.K + min(kt * 32 + (lane & 31), a.K - 1)];
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[i][1], 0, 0, 0);
        }
      }
    }
  }
  float* out = a.part + (size_t)(tower * a.slabs + slab) * a.K * a.C;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kt = wave + i * kGenWaves;
    if (kt >= KT) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = c0 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (k < a.K && col < a.C) out[(size_t)k * a.C + col] = acc[i][j][r];
      }
    }
  }
}

// ---- first layer backward.  Per cloud: S[c] = sum_n dZ1[n, c], P[d][c] = sum_n x'[n, d] dZ1[n, c]  (the cloud's share of dW_1), then
//      gx[d] = sum_c W1[d, c] S[c],  grot = sum_c (W1[0, c] P[1][c] - W1[1, c] P[0][c])   (kernels_train_bwd.h: same identities).
//      grid (2B), block 256 = 4 row groups x 64 columns; C1 <= 256.
struct GenL1BwdArgs { const float* X0; const float* dZ; const float* W; int B, N, C; float* Ppart; float* gx; float* grot; };   // Ppart: [2B][3][C]

__global__ __launch_bounds__(256) void gen_layer1_bwd(const GenL1BwdArgs a)
{
  __shared__ float red[4][64][4];
  __shared__ double fin[4];
  const int cloud = blockIdx.x, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  if (threadIdx.x < 4) fin[threadIdx.x] = 0.0;
  __syncthreads();
  for (int c0 = 0; c0 < a.C; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < a.C;
    float s = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (live)
      for (int n = g; n < a.N; n += 4) {
        const size_t row = (size_t)cloud * a.N + n;
        const f32x4 x = reinterpret_cast<const f32x4*>(a.X0)[row];
        const float d = a.dZ[row * a.C + c];
        s += d; p0 = fmaf(x[0], d, p0); p1 = fmaf(x[1], d, p1); p2 = fmaf(x[2], d, p2);
      }
    red[g][lane][0] = s; red[g][lane][1] = p0; red[g][lane][2] = p1; red[g][lane][3] = p2;
    __syncthreads();
    if (g == 0) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
      double q[4] = {0.0, 0.0, 0.0, 0.0};
      if (live) {
        a.Ppart[((size_t)cloud * 3 + 0) * a.C + c] = v[1];
        a.Ppart[((size_t)cloud * 3 + 1) * a.C + c] = v[2];
        a.Ppart[((size_t)cloud * 3 + 2) * a.C + c] = v[3];
        const float w0 = a.W[c], w1 = a.W[a.C + c], w2 = a.W[2 * a.C + c];
        q[0] = (double)w0 * v[0]; q[1] = (double)w1 * v[0]; q[2] = (double)w2 * v[0];
        q[3] = (double)w0 * v[2] - (double)w1 * v[1];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q[e] += __shfl_xor(q[e], o);
        if (lane == 0) fin[e] += q[e];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) a.gx[cloud * 3 + threadIdx.x] = (float)fin[threadIdx.x];
  if (threadIdx.x == 3) a.grot[cloud] = (float)fin[3];
}

// =================================================================================================================================
// DGCNN backbones of any widths / depth (models/tp8.py:30-46 loops over `sizes[:-1]` for the edge convs): the same layer-by-layer
// machinery over the B N k EDGE rows (row = ((tower B + b) N + n) k + slot), fed by three kernels of its own:
//   gen_edge_kernel      E = [x_i', (x_j - x_i) R]  per edge row, [rows][8] (columns 6, 7 zero)      (utils/tf_util_dgcnn.py:674-706)
//   gen_layer1e_fwd      Z_1 = E W_1 + b_1 (K = 6, VALU) + tile statistics
//   gen_layer1e_bwd      per cloud: S = sum dZ_1, P = E^T dZ_1 -> dW_1 partials and the frame gradients gx / grot
// The max over the k neighbours and the max over the points are gen_pool_fwd / gen_pool_bwd with (clouds, rows per cloud) =
// (2 B N, k) and (2 B, N); the point conv takes the pooled edge features through identity scale / shift arrays.
// =================================================================================================================================
__global__ void gen_edge_kernel(const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ xform, const int* __restrict__ nn,
                                int B, int N, int k, float* __restrict__ E)
{
  const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (r >= (size_t)2 * B * N * k) return;
  const size_t pt = r / k;
  const int cloud = (int)(pt / N), n = (int)(pt - (size_t)cloud * N), tower = cloud >= B, b = cloud - tower * B;
  const float* pc = (tower ? p2 : p1) + (size_t)b * N * 3;
  const float* xf = xform + (size_t)cloud * 12;
  const int j = nn[r];
  const float* p = pc + (size_t)n * 3;
  const float* pj = pc + (size_t)j * 3;
  const float x = p[0] - xf[0], y = p[1] - xf[1], z = p[2] - xf[2];
  const float dx = pj[0] - p[0], dy = pj[1] - p[1], dz = pj[2] - p[2];
  f32x4 a, c;
  a[0] = x * xf[3] + y * xf[6] + z * xf[9];
  a[1] = x * xf[4] + y * xf[7] + z * xf[10];
  a[2] = x * xf[5] + y * xf[8] + z * xf[11];
  a[3] = dx * xf[3] + dy * xf[6] + dz * xf[9];
  c[0] = dx * xf[4] + dy * xf[7] + dz * xf[10];
  c[1] = dx * xf[5] + dy * xf[8] + dz * xf[11];
  c[2] = 0.f; c[3] = 0.f;
  reinterpret_cast<f32x4*>(E)[2 * r] = a;
  reinterpret_cast<f32x4*>(E)[2 * r + 1] = c;
}

// grid (tiles, 2 towers), block 256 = 4 row groups x 64 columns (as gen_layer1_fwd, K = 6, rows of 8 floats)
__global__ __launch_bounds__(256) void gen_layer1e_fwd(const GenL1Args a)
{
  __shared__ float red[4][64][2];
  const int tile = blockIdx.x, tower = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int nvalid = min(kGenTile, a.M - tile * kGenTile);
  const size_t row0 = (size_t)tower * a.M + (size_t)tile * kGenTile;
  auto zrow = [&](size_t row, const float (&w)[6], float bb) {
    const f32x4 x = reinterpret_cast<const f32x4*>(a.X0)[2 * row], y = reinterpret_cast<const f32x4*>(a.X0)[2 * row + 1];
    return fmaf(y[1], w[5], fmaf(y[0], w[4], fmaf(x[3], w[3], fmaf(x[2], w[2], fmaf(x[1], w[1], fmaf(x[0], w[0], bb))))));
  };
  for (int c0 = 0; c0 < a.C; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < a.C;
    float w[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) w[d] = live ? a.W[d * a.C + c] : 0.f;
    const float bb = live ? a.bias[c] : 0.f;
    const float shift = zrow(row0, w, bb);
    float s1 = 0.f, s2 = 0.f;
    for (int r = g; r < nvalid; r += 4) {
      const float z = zrow(row0 + r, w, bb);
      if (live) a.Z[(row0 + r) * a.C + c] = z;
      const float d = z - shift;
      s1 += d; s2 = fmaf(d, d, s2);
    }
    red[g][lane][0] = s1; red[g][lane][1] = s2;
    __syncthreads();
    if (g == 0 && live) {
      s1 = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
      s2 = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
      gen_tile_stat(s1, s2, shift, nvalid, a.part + (((size_t)tower * a.tiles + tile) * a.C + c) * 2);
    }
    __syncthreads();
  }
}

// grid 2B, block 256; a.N = rows per cloud (N k); Ppart [2B][6][C];  gx_d = sum_c w_dc S_c (d < 3: only the x_i' half translates with the
// frame), grot = sum_c (w_0c P_1c - w_1c P_0c + w_3c P_4c - w_4c P_3c) (both halves rotate)
__global__ __launch_bounds__(256) void gen_layer1e_bwd(const GenL1BwdArgs a)
{
  __shared__ float red[4][64][7];
  __shared__ double fin[4];
  const int cloud = blockIdx.x, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  if (threadIdx.x < 4) fin[threadIdx.x] = 0.0;
  __syncthreads();
  for (int c0 = 0; c0 < a.C; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < a.C;
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // S, P_0 .. P_5
    if (live)
      for (int n = g; n < a.N; n += 4) {
        const size_t row = (size_t)cloud * a.N + n;
        const f32x4 x = reinterpret_cast<const f32x4*>(a.X0)[2 * row], y = reinterpret_cast<const f32x4*>(a.X0)[2 * row + 1];
        const float d = a.dZ[row * a.C + c];
        v[0] += d;
        v[1] = fmaf(x[0], d, v[1]); v[2] = fmaf(x[1], d, v[2]); v[3] = fmaf(x[2], d, v[3]);
        v[4] = fmaf(x[3], d, v[4]); v[5] = fmaf(y[0], d, v[5]); v[6] = fmaf(y[1], d, v[6]);
      }
#pragma unroll
    for (int e = 0; e < 7; ++e) red[g][lane][e] = v[e];
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int e = 0; e < 7; ++e) v[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
      double q[4] = {0.0, 0.0, 0.0, 0.0};
      if (live) {
#pragma unroll
        for (int d = 0; d < 6; ++d) a.Ppart[((size_t)cloud * 6 + d) * a.C + c] = v[1 + d];
        float w[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) w[d] = a.W[d * a.C + c];
        q[0] = (double)w[0] * v[0]; q[1] = (double)w[1] * v[0]; q[2] = (double)w[2] * v[0];
        q[3] = (double)w[0] * v[2] - (double)w[1] * v[1] + (double)w[3] * v[5] - (double)w[4] * v[4];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q[e] += __shfl_xor(q[e], o);
        if (lane == 0) fin[e] += q[e];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) a.gx[cloud * 3 + threadIdx.x] = (float)fin[threadIdx.x];
  if (threadIdx.x == 3) a.grot[cloud] = (float)fin[3];
}

__global__ void gen_fill_kernel(float* __restrict__ p, size_t n, float v)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace alignnet
