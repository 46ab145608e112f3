// Test hooks only (alignnet_debug_train_relu_mask, include/alignnet_hip.h): the SIGN every relu of the last training step saw
// (utils/tf_util.py:167-168,345-346), one byte per element.  Nothing here is launched by a training or inference step.
//
// Where the step keeps the relu's output (the stored h2 of a PointNet stage, the pooled features, the pooled edge features) the mask
// is read from it; where the step RECOMPUTES the activation in every pass (h1 from xyz: kernels_train_fwd.h layer1_to_lds and its
// variants; the dgcnn K = 6 lift: kernels_train_dgcnn.h dgt_liftm) the mask kernel runs the same device functions on the same frame,
// weights and batch-statistics scale / shift the step left in its workspace, so the bits are the ones the passes multiplied with.
#pragma once
#include "kernels_train_fwd.h"
#include "kernels_train_dgcnn.h"
#include "kernels_train_head.h"

namespace alignnet {

// out[i] = fmaf(z[i], sc[t][c], sh[t][c]) > 0 over [2 * rows_per_tower][C] (sc == nullptr: z[i] > 0).  T = float, or unsigned short
// for a bf16 buffer (the stored h2 of the bf16 step).
template <typename T>
__global__ void dbg_mask_rows_kernel(const T* __restrict__ z, const float* __restrict__ sc, const float* __restrict__ sh, size_t rows_per_tower, int C,
                                     unsigned char* __restrict__ out)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= 2 * rows_per_tower * C) return;
  const int c = (int)(i % C), t = (i / C) >= rows_per_tower;
  float v;
  if constexpr (sizeof(T) == 2) v = __uint_as_float((unsigned)z[i] << 16); else v = z[i];
  if (sc) v = fmaf(v, sc[t * C + c], sh[t * C + c]);
  out[i] = v > 0.f ? 1 : 0;
}

// the pooled features in their consumer's layout (StageWS::pooled): out[(t * B + b) * C + c] = pooled[t * tower_stride + b * row_stride + c] > 0
__global__ void dbg_mask_pooled_kernel(const float* __restrict__ pooled, long tower_stride, long row_stride, int B, int C, unsigned char* __restrict__ out)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)2 * B * C) return;
  const int c = (int)(i % C), cloud = (int)(i / C), t = cloud >= B, b = cloud - t * B;
  out[i] = pooled[t * tower_stride + b * row_stride + c] > 0.f ? 1 : 0;
}

// head hidden layer: the forward's expression (kernels_train_head.h bn_rows_fwd_kernel): inv = gamma / sqrt(var + eps), sh = beta - mean inv, y = relu(fma(z, inv, sh))
__global__ void dbg_mask_head_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ var, const float* g0, const float* g1,
                                     const float* b0, const float* b1, int M, int C, int rows_per_set, unsigned char* __restrict__ out)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int c = (int)(i % C), r = (int)(i / C), set = r >= rows_per_set;
  const float mf = mean[set * C + c], vf = var[set * C + c];
  const float inv = (set ? g1 : g0)[c] * (1.0f / sqrtf(vf + kBnEps)), sh = (set ? b1 : b0)[c] - mf * inv;
  out[i] = fmaf(z[i], inv, sh) > 0.f ? 1 : 0;
}

// PointNet conv1 of a fused stage: h1 = relu(fma(x' . w, sc, sh)) through the passes' own tile functions.  grid 2B, block kTW * 64;
// LDS: xs [64][4] | h1 [64][C1r + 4]
struct DbgLayer1Args { const float* pcs[2]; const float* xform; int B, N, C1, ld0; const float* w1; const float *sc1, *sh1; unsigned char* out; };
__global__ __launch_bounds__(kTW * 64) void dbg_mask_layer1_kernel(const DbgLayer1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem; float* h1 = smem + kTT * 4;
  const int tid = threadIdx.x, cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const XForm X = xform_load(a.xform + (size_t)cloud * 12);
  const Layer1W L = layer1_load(a.w1, a.C1, a.sc1 + tower * a.C1, a.sh1 + tower * a.C1, tid);
  const int ntiles = (a.N + kTT - 1) / kTT;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    tile_point_store(tile_point_request(pc, a.N, tile, tid), X, xs, tid);
    __syncthreads();
    layer1_to_lds(xs, L, a.C1, h1, a.ld0, nvalid, tid);
    __syncthreads();
    for (int e = tid; e < nvalid * a.C1; e += kTW * 64) {
      const int row = e / a.C1, c = e - row * a.C1;
      a.out[((size_t)cloud * a.N + (size_t)tile * kTT + row) * a.C1 + c] = h1[row * a.ld0 + c] > 0.f ? 1 : 0;
    }
    __syncthreads();
  }
}

// dgcnn edge conv1 of a fused stage (C1 in {32, 64}): the K = 6 lift on the matrix pipe exactly as dg_train_fwd / dg_train_bwd_edge run it
// (dgt_liftm: the same two 16x16x4 MFMAs per tile on the same operands).  out [2B][N][k][C1].  grid 2B, block kTW * 64; LDS: es [64][8] | h1 [64][C1 + 4]
struct DbgEdge1Args { const float* pcs[2]; const float* xform; const int* nn; int B, N, k; const float* w1; const float *sc1, *sh1; unsigned char* out; };
template <int C1>
__global__ __launch_bounds__(kTW * 64) void dbg_mask_edge1_kernel(const DbgEdge1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* es = smem; float* h1 = smem + kTT * 8;
  constexpr int ld0 = C1 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  const DgtLiftM<C1, kTW> lw = dgt_liftm_load<C1, kTW>(a.w1, a.sc1 + tower * C1, a.sh1 + tower * C1, wave, lane);
  if (tid < kTT) { es[tid * 8 + 6] = 0.f; es[tid * 8 + 7] = 0.f; }
  const int ntiles = (a.N + kTT - 1) / kTT, total = ntiles * a.k;
  for (int j = 0; j < total; ++j) {
    const int tile = j / a.k, slot = j - tile * a.k, nvalid = min(kTT, a.N - tile * kTT);
    if (tid < kTT) {
      float v[6];
      dgt_points(pc, a.N, a.k, j, tid, dgt_index(nnc, a.N, a.k, j, tid), v);
      dg_edge_to_lds(xf, v, es + tid * 8);
    }
    __syncthreads();
    dgt_liftm<C1, kTW>(lw, es, h1, ld0, nvalid, wave, lane);
    __syncthreads();
    for (int e = tid; e < nvalid * C1; e += kTW * 64) {
      const int row = e / C1, c = e - row * C1;
      a.out[(((size_t)cloud * a.N + (size_t)tile * kTT + row) * a.k + slot) * C1 + c] = h1[row * ld0 + c] > 0.f ? 1 : 0;
    }
    __syncthreads();
  }
}

// bf16 step (train_matmul_bf16): the ROUNDED h1 the hidden conv multiplied, as bf16 bits [2B N][C1] -- the passes' own lift (layer1_to_lds<ROUND>:
// relu(fma(x' . w, sc, sh)) rounded by to_bf16_bits).  grid 2B, block kTW * 64; LDS as dbg_mask_layer1_kernel
struct DbgRound1Args { const float* pcs[2]; const float* xform; int B, N, C1, ld0; const float* w1; const float *sc1, *sh1; unsigned short* out; };
__global__ __launch_bounds__(kTW * 64) void dbg_rounded_layer1_kernel(const DbgRound1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem; float* h1 = smem + kTT * 4;
  const int tid = threadIdx.x, cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const XForm X = xform_load(a.xform + (size_t)cloud * 12);
  const Layer1W L = layer1_load(a.w1, a.C1, a.sc1 + tower * a.C1, a.sh1 + tower * a.C1, tid);
  const int ntiles = (a.N + kTT - 1) / kTT;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    tile_point_store(tile_point_request(pc, a.N, tile, tid), X, xs, tid);
    __syncthreads();
    layer1_to_lds<true>(xs, L, a.C1, h1, a.ld0, nvalid, tid);
    __syncthreads();
    for (int e = tid; e < nvalid * a.C1; e += kTW * 64) {
      const int row = e / a.C1, c = e - row * a.C1;
      a.out[((size_t)cloud * a.N + (size_t)tile * kTT + row) * a.C1 + c] = (unsigned short)(__float_as_uint(h1[row * a.ld0 + c]) >> 16);
    }
    __syncthreads();
  }
}

// dgcnn, bf16 step: the rounded output of the K = 6 lift (input of the second edge conv) as bf16 bits [2B][N][k][C1]: dgt_liftm's fp32 value through
// to_bf16_bits -- what dgt_liftm_bf16 stores (the same expression on the same accumulator).  Launch as dbg_mask_edge1_kernel.
struct DbgEdgeR1Args { const float* pcs[2]; const float* xform; const int* nn; int B, N, k; const float* w1; const float *sc1, *sh1; unsigned short* out; };
template <int C1>
__global__ __launch_bounds__(kTW * 64) void dbg_rounded_edge1_kernel(const DbgEdgeR1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* es = smem; float* h1 = smem + kTT * 8;
  constexpr int ld0 = C1 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int* nnc = a.nn + (size_t)cloud * a.N * a.k;
  const DgtLiftM<C1, kTW> lw = dgt_liftm_load<C1, kTW>(a.w1, a.sc1 + tower * C1, a.sh1 + tower * C1, wave, lane);
  if (tid < kTT) { es[tid * 8 + 6] = 0.f; es[tid * 8 + 7] = 0.f; }
  const int ntiles = (a.N + kTT - 1) / kTT, total = ntiles * a.k;
  for (int j = 0; j < total; ++j) {
    const int tile = j / a.k, slot = j - tile * a.k, nvalid = min(kTT, a.N - tile * kTT);
    if (tid < kTT) {
      float v[6];
      dgt_points(pc, a.N, a.k, j, tid, dgt_index(nnc, a.N, a.k, j, tid), v);
      dg_edge_to_lds(xf, v, es + tid * 8);
    }
    __syncthreads();
    dgt_liftm<C1, kTW>(lw, es, h1, ld0, nvalid, wave, lane);
    __syncthreads();
    for (int e = tid; e < nvalid * C1; e += kTW * 64) {
      const int row = e / C1, c = e - row * C1;
      a.out[(((size_t)cloud * a.N + (size_t)tile * kTT + row) * a.k + slot) * C1 + c] = to_bf16_bits(h1[row * ld0 + c]);
    }
    __syncthreads();
  }
}
// [n] floats -> bf16 bits (the pooled edge features p as the bf16 point conv rounds them while staging)
__global__ void dbg_round_rows_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = to_bf16_bits(src[i]);
}

// the classes of the loss's target angles (ALIGNNET_DECISION_ANGLE_CLASS): out [2 variants: theta, theta + pi][B][W], W = B for the pair term
// (entry (i, j): label difference of row i against the decoded yaw difference of column j, models/tp8.py:327), 1 otherwise
__global__ void dbg_angle_class_kernel(const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ theta, int B, int nb, int term,
                                       int* __restrict__ out)
{
  const size_t W = term == 2 ? (size_t)B : 1, idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= 2 * (size_t)B * W) return;
  const int v = (int)(idx / (B * W)), i = (int)((idx / W) % B), j = (int)(idx % W);
  const float pi = 3.14159274101257324f;
  const float dth = theta[B + j] - theta[j];
  const float a1i = a1[i], a2i = a2[i];
  int c; float r;
  if (term < 2) angle2class((term == 0 ? a1i : a2i) + (v ? pi : 0.f), nb, &c, &r);   // (the expressions of loss_pairs_body, kernels_train_head.h)
  else angle2class((a2i - a1i) - dth + (v ? pi : 0.f), nb, &c, &r);
  out[idx] = c;
}

}  // namespace alignnet
