// Inference kernels of the AlignNet-3D tp8 engine, written for gfx950 (CDNA4) only.
//
//   pointnet_fused   models/tp8.py:49-59 (+ the re-centre / rotate prologue of
//                    :106,113,122-127): shared per-point MLP + max over points, fused.
//   fc_mfma          models/tp8.py:75-82 via utils/tf_util.py:311-347: head MLP layers.
//   centroid/finish  models/tp8.py:104,109,117-125,155-156 and :294-301 (yaw decode).
//
// BatchNorm in eval mode is a fixed per-channel affine; it is applied in the
// accumulator epilogue as  relu(acc * scale + shift)  with
//   scale = gamma * rsqrt(moving_var + 1e-3),  shift = (bias - moving_mean) * scale + beta
// (utils/tf_util.py:488-491), one (scale, shift) pair PER TOWER because beta/gamma and
// the EMA shadows are per tower while conv/fc weights are shared.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alignnet {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kTilePts = 128;     // points per workgroup
constexpr int kWaves = 8;         // 512 threads, two waves per SIMD
constexpr int kMaxConv = 6;
constexpr float kBnEps = 1e-3f;   // utils/tf_util.py:491

// s_setprio(1) around the MFMA clusters: +4.5 % on the fused backbone (waves in LDS/VALU phases yield the SIMD)
#ifndef ALIGNNET_NO_SETPRIO
#define ALIGNNET_SETPRIO 1
#endif

// ---------------------------------------------------------------------------------
// Weight image for the MFMA layers.  v_mfma_f32_32x32x2_f32 takes, per lane l,
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  A k-group is 8 consecutive k;
// lane (j, half) keeps k = 8*kg + 4*half + s (s = 0..3) in one float4, so one
// coalesced 1 KiB load feeds four MFMAs:
//   Wp[((ct * KG + kg) * 64 + lane) * 4 + s] = W[8*kg + 4*(lane>>5) + s][32*ct + (lane&31)]
// Out-of-range k / channel entries are zero.
// ---------------------------------------------------------------------------------
[[maybe_unused]] static __global__ void pack_weights_kernel(const float* __restrict__ W, int K, int C, float* __restrict__ Wp)
{
  const int KG = (K + 7) >> 3, CT = (C + 31) >> 5;
  const size_t total = (size_t)CT * KG * 256;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int s = idx & 3, lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kg = t % KG, ct = t / KG;
    const int k = 8 * kg + 4 * (lane >> 5) + s, c = 32 * ct + (lane & 31);
    Wp[idx] = (k < K && c < C) ? W[(size_t)k * C + c] : 0.f;
  }
}

// all MFMA layers in one launch (training re-packs every step): grid (blocks, jobs)
struct PackJob { const float* src; float* dst; int K, C; };
__device__ __forceinline__ void pack_weights_multi_body(const PackJob* __restrict__ jobs, unsigned bx, unsigned by, unsigned gx)
{
  const PackJob j = jobs[by];
  const int KG = (j.K + 7) >> 3, CT = (j.C + 31) >> 5;
  const size_t total = (size_t)CT * KG * 256;
  for (size_t idx = bx * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gx * blockDim.x) {
    const int s = idx & 3, lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kg = t % KG, ct = t / KG;
    const int k = 8 * kg + 4 * (lane >> 5) + s, c = 32 * ct + (lane & 31);
    j.dst[idx] = (k < j.K && c < j.C) ? j.src[(size_t)k * j.C + c] : 0.f;
  }
}
[[maybe_unused]] static __global__ void pack_weights_multi_kernel(const PackJob* __restrict__ jobs) { pack_weights_multi_body(jobs, blockIdx.x, blockIdx.y, gridDim.x); }

// scale/shift for one layer and one BN set; bn == nullptr -> plain bias.
[[maybe_unused]] static __global__ void fold_bn_kernel(const float* __restrict__ bias, const float* __restrict__ beta,
                               const float* __restrict__ gamma, const float* __restrict__ mean,
                               const float* __restrict__ var, int C, float* __restrict__ scale,
                               float* __restrict__ shift)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (beta) {
    const float inv = gamma[c] * (1.0f / sqrtf(var[c] + kBnEps));
    scale[c] = inv;
    shift[c] = (bias[c] - mean[c]) * inv + beta[c];
  } else {
    scale[c] = 1.f;
    shift[c] = bias[c];
  }
}

// ---------------------------------------------------------------------------------
// centroid: models/tp8.py:104.  One workgroup per cloud.  Writes the stage-1 frame
// (centre = mean, rotation = identity) and keeps the mean for tp8.py:109.
// xform layout per cloud: c[3], R[9] row-major (p' = (p - c) @ R).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void centroid_body(const float* __restrict__ pcs1, const float* __restrict__ pcs2, int B, int N, float* __restrict__ xform,
                                              float* __restrict__ center_mean, float* __restrict__ zero, size_t nzero, int cloud, int nclouds)
{
  // eval forward: the pooled-feature buffers (atomicMax targets of the three backbones) are cleared here, a slice per workgroup,
  // instead of by a memset launch of their own in front of this kernel
  if (zero) {
    const size_t per = (nzero + nclouds - 1) / nclouds, lo = cloud * per, hi = lo + per < nzero ? lo + per : nzero;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) zero[i] = 0.f;
  }
  const int tower = cloud >= B, b = cloud - tower * B;
  const float* pc = (tower ? pcs2 : pcs1) + (size_t)b * N * 3;
  float s[3] = {0.f, 0.f, 0.f};
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    s[0] += pc[n * 3 + 0];
    s[1] += pc[n * 3 + 1];
    s[2] += pc[n * 3 + 2];
  }
  __shared__ float red[4][3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float v = s[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float m = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)N;
    center_mean[cloud * 3 + threadIdx.x] = m;
    xform[cloud * 12 + threadIdx.x] = m;
  }
  if (threadIdx.x < 9) xform[cloud * 12 + 3 + threadIdx.x] = (threadIdx.x % 4 == 0) ? 1.f : 0.f;
}
static __global__ __launch_bounds__(256) void centroid_kernel(const float* __restrict__ pcs1, const float* __restrict__ pcs2,
                                                      int B, int N, float* __restrict__ xform,
                                                      float* __restrict__ center_mean, float* __restrict__ zero = nullptr, size_t nzero = 0)
{
  centroid_body(pcs1, pcs2, B, N, xform, center_mean, zero, nzero, blockIdx.x, gridDim.x);
}

// bf16 operands (training option train_matmul_bf16, inference option infer_matmul_bf16x3)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short to_bf16_bits(float x)   // round-to-nearest-even (v_cvt_pk_bf16_f32 semantics)
{
  const __bf16 v = (__bf16)x;
  return __builtin_bit_cast(unsigned short, v);
}

// ---------------------------------------------------------------------------------
// pointnet_fused
// ---------------------------------------------------------------------------------
struct ConvLayerDev {
  const float* w;      // layer 0: [3][cout] row-major; others: packed image (see above)
  const float* scale;  // [2][cout]  (tower-major)
  const float* shift;  // [2][cout]
  int cin, cout;
};

struct BackboneArgs {
  const float* pcs[2];   // [B][N][3] per tower
  const float* xform;    // [2B][12]
  float* pooled;         // zero-initialised; element (tower, b, c) at tower*tower_stride + b*row_stride + c
  long tower_stride, row_stride;
  int B, N, nlayers;
  int ld[2];             // leading dimensions (floats) of the two LDS activation buffers
  int tiles_per_wg;      // consecutive point tiles of one cloud walked by a workgroup (pooled max published once)
  ConvLayerDev L[kMaxConv];
#ifdef ALIGNNET_KSTAMP
  long long* stamps;     // debug build: s_memtime at every k-group of the last layer, waves 0 and 4 of workgroup (0, 0)
#endif
};

// Weight (B operand) stream: issued by hand two k-groups ahead and retired with a counted wait.
// hipcc otherwise sinks the load to the top of the consuming iteration behind `s_waitcnt vmcnt(0)`, exposing
// the L2 latency on every k-group (cdna_hip_programming.md 5.7: an asm load is invisible to the compiler's
// wait bookkeeping, so the wait below is ours).  Loads return in order, and nothing else is stored inside the
// loop, so `vmcnt(2)` means: everything except the two newest loads has landed.
__device__ __forceinline__ f32x4 wload_issue(const f32x4* p)
{
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wload_wait2(f32x4& v) { asm volatile("s_waitcnt vmcnt(2)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void wload_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same wait, naming the three stream registers as read-write operands: look-ahead loads into them may still be in flight when the loop
// ends, and only a (formal) use at the wait keeps the allocator from handing one of them to a new value BEFORE it -- an ordinary v_mov is not
// ordered against an asm volatile statement.  (Seen in train_fwd_phase3_wide<true, true>: `v_mov_b32 v142, 0` -- the arg-max counter -- was
// scheduled in front of the drain while `global_load_dwordx4 v[142:145]` was still outstanding, and the counter became a weight.)
__device__ __forceinline__ void wload_drain(f32x4& b0, f32x4& b1, f32x4& b2) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2)::"memory"); }

template <int MR>
__device__ __forceinline__ void mfma_kgroup(const f32x4 (&av)[MR], const f32x4& bv, f32x16 (&acc)[MR])
{
#ifdef ALIGNNET_SETPRIO
  __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s], bv[s], acc[m], 0, 0, 0);
#ifdef ALIGNNET_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

template <int MR>
__device__ __forceinline__ void lds_rows(const float* arow, int lda, int kg, f32x4 (&av)[MR])
{
#pragma unroll
  for (int m = 0; m < MR; ++m) av[m] = *reinterpret_cast<const f32x4*>(arow + m * 32 * lda + kg * 8);
}

// acc[m] (+)= A[rows m*32.., :K] * W  for MR row tiles and one 32-channel tile.  Unrolled by three k-groups so the
// three in-flight weight registers rotate roles WITHOUT register copies (a copy of an asm-load destination that is
// still in flight reads garbage).
// ASMB = true: hand-issued weight stream (only for kernels verified spill-free: a spilled / re-allocated asm-load
// destination that is still in flight corrupts its new owner).  ASMB = false: compiler-managed loads.
#ifdef ALIGNNET_KSTAMP
__device__ long long* g_kstamp = nullptr;   // set per lane-0 of the traced waves; null elsewhere
#define KSTAMP() do { if (kst) { *kst++ = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define KSTAMP() do {} while (0)
#endif

template <int MR, bool CLEAR = true, bool ASMB = true>
__device__ __forceinline__ void mfma_rows(const float* __restrict__ A, int lda, const f32x4* __restrict__ Wp,
                                          int KG, int lane, f32x16 (&acc)[MR], long long* kst = nullptr)
{
  (void)kst;
  if (CLEAR) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  }
  const float* arow = A + (lane & 31) * lda + (lane >> 5) * 4;
  if (!ASMB) {
    f32x4 bcur = Wp[lane];
    f32x4 av[MR];
    lds_rows<MR>(arow, lda, 0, av);
    for (int kg = 0; kg < KG; ++kg) {
      const int kn = kg + 1 < KG ? kg + 1 : kg;
      const f32x4 bnext = Wp[kn * 64 + lane];
      f32x4 an[MR];
      lds_rows<MR>(arow, lda, kn, an);
      mfma_kgroup<MR>(av, bcur, acc);
      bcur = bnext;
#pragma unroll
      for (int m = 0; m < MR; ++m) av[m] = an[m];
    }
    return;
  }
  const f32x4* wp = Wp + lane;
  const int last = KG - 1;
  // No drain here: stores / atomics of the previous epilogue may still be in flight.  Loads retire in order among
  // themselves, so after `vmcnt(2)` at most two operations are outstanding and the oldest of our three loads cannot be
  // one of them -- the wait is merely conservative while an older store is pending.
  f32x4 b0 = wload_issue(wp);
  f32x4 b1 = wload_issue(wp + min(1, last) * 64);
  f32x4 b2;
  f32x4 a0[MR], a1[MR];
  lds_rows<MR>(arow, lda, 0, a0);
  for (int kg = 0; kg < KG; kg += 3) {
    KSTAMP();
    b2 = wload_issue(wp + min(kg + 2, last) * 64);
    lds_rows<MR>(arow, lda, min(kg + 1, last), a1);
    wload_wait2(b0);
    KSTAMP();
    mfma_kgroup<MR>(a0, b0, acc);
    KSTAMP();
    b0 = wload_issue(wp + min(kg + 3, last) * 64);
    lds_rows<MR>(arow, lda, min(kg + 2, last), a0);
    wload_wait2(b1);
    if (kg + 1 < KG) mfma_kgroup<MR>(a1, b1, acc);
    b1 = wload_issue(wp + min(kg + 4, last) * 64);
    lds_rows<MR>(arow, lda, min(kg + 3, last), a1);
    wload_wait2(b2);
    if (kg + 2 < KG) mfma_kgroup<MR>(a0, b2, acc);
#pragma unroll
    for (int m = 0; m < MR; ++m) a0[m] = a1[m];
  }
  KSTAMP();
  wload_drain(b0, b1, b2);   // look-ahead loads still in flight: retire them before the registers are reused
  KSTAMP();
}

// hidden layer: out[row][col] = relu(acc*scale+shift) for this item's MR row tiles x one channel tile
// Same product with compiler-managed loads (safe under spills) and the weight stream three k-groups deep: each load is
// pinned in front of the MFMAs of the k-group two ahead of it by a compiler barrier (left alone, hipcc sinks it next to its
// use and every k-group starts with an exposed L2 round trip -- a one-deep lookahead covers 256 cycles of MFMAs at MR = 1,
// the round trip is 500-800).
template <int MR, bool CLEAR = true>
__device__ __forceinline__ void mfma_rows_deep(const float* __restrict__ A, int lda, const f32x4* __restrict__ Wp, int KG, int lane,
                                               f32x16 (&acc)[MR])
{
  if (CLEAR) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  }
  const float* arow = A + (lane & 31) * lda + (lane >> 5) * 4;
  const f32x4* wp = Wp + lane;
  const int last = KG - 1;
  f32x4 b0 = wp[0], b1 = wp[min(1, last) * 64], b2 = wp[min(2, last) * 64];
  asm volatile("" ::: "memory");
  f32x4 av[MR];
  for (int kg = 0; kg < KG; kg += 3) {
    lds_rows<MR>(arow, lda, kg, av);
    mfma_kgroup<MR>(av, b0, acc);
    b0 = wp[min(kg + 3, last) * 64];
    asm volatile("" ::: "memory");
    if (kg + 1 < KG) { lds_rows<MR>(arow, lda, kg + 1, av); mfma_kgroup<MR>(av, b1, acc); }
    b1 = wp[min(kg + 4, last) * 64];
    asm volatile("" ::: "memory");
    if (kg + 2 < KG) { lds_rows<MR>(arow, lda, kg + 2, av); mfma_kgroup<MR>(av, b2, acc); }
    b2 = wp[min(kg + 5, last) * 64];
    asm volatile("" ::: "memory");
  }
}

template <int MR, int KGC = 0>   // KGC: compile-time k-depth (0 = from the layer)
__device__ __forceinline__ void hidden_item(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                            const ConvLayerDev& L, int tower, int ct, int rg, int lane)
{
  const int KG = KGC ? KGC : (L.cin + 7) >> 3;
  f32x16 acc[MR];
  mfma_rows<MR>(in + rg * MR * 32 * ldi, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc);
  const int col = ct * 32 + (lane & 31);
  const bool live = col < L.cout;
  const float sc = live ? L.scale[tower * L.cout + col] : 0.f;
  const float sh = live ? L.shift[tower * L.cout + col] : 0.f;
  if (col < ldo - 4) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (rg * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[row * ldo + col] = fmaxf(fmaf(acc[m][r], sc, sh), 0.f);
      }
  }
}

template <int MR, int TP>
__device__ __forceinline__ void hidden_layer(const float* in, int ldi, float* out, int ldo, const ConvLayerDev& L,
                                             int tower, int wave, int lane)
{
  constexpr int RG = (TP / 32) / MR;
  const int CT = (L.cout + 31) >> 5;
  for (int item = wave; item < CT * RG; item += kWaves) hidden_item<MR>(in, ldi, out, ldo, L, tower, item / RG, item % RG, lane);
}

// LD0 / LD1: compile-time LDS row strides of the two activation buffers (0 = from the arguments): immediates instead of address
// registers in the unrolled hidden-layer epilogues
template <int TP, int LD0 = 0, int LD1 = 0, int KGL = 0>   // KGL: k-groups of the last layer (0 = from the arguments)
__global__ __launch_bounds__(kWaves * 64, 2) void pointnet_fused(const BackboneArgs a)
{
  const int lds_ld[2] = {LD0 ? LD0 : a.ld[0], LD1 ? LD1 : a.ld[1]};
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cloud = blockIdx.y;
  const int tower = cloud >= a.B, b = cloud - tower * a.B;
  float* xs = smem;                       // [TP][4]
  // integer offsets (not a runtime-selected pointer) keep the LDS address space visible -> ds_read_b128, not flat loads
  const int boff[2] = {TP * 4, TP * 4 + TP * lds_ld[0]};
  // A workgroup walks tiles_per_wg consecutive tiles of its cloud and keeps the pooled maximum of its (<= kPoolRegs per
  // wave) channel tiles in registers: one atomicMax per (workgroup, channel) instead of one per (tile, channel) -- that
  // per-tile stream was 60 % of the kernel's HBM traffic.
  constexpr int kPoolRegs = 4;
  float pool[kPoolRegs] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles_all = (a.N + TP - 1) / TP;
  const int tile_lo = blockIdx.x * a.tiles_per_wg, tile_hi = min(ntiles_all, tile_lo + a.tiles_per_wg);
  // xyz of the tile to come (threads 0 .. TP-1): requested one tile ahead, so its HBM round trip overlaps the previous
  // tile's last layer instead of sitting in front of the first barrier
  float nx = 0.f, ny = 0.f, nz = 0.f;
  auto request_xyz = [&](int t) {
    if (tid < TP) {
      const int n = min(t * TP + tid, a.N - 1);   // tail rows repeat the last point: max unaffected
      const float* p = a.pcs[tower] + ((size_t)b * a.N + n) * 3;
      nx = p[0]; ny = p[1]; nz = p[2];
    }
  };
  request_xyz(tile_lo);
  // the K = 3 lift's weights / scale / shift of this thread's first two channel groups (widths <= 64): loaded once per
  // workgroup instead of once per tile
  float l0w[2][5];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const ConvLayerDev& L0 = a.L[0];
    const int c = (tid & 31) + 32 * g;
    const bool live = c < L0.cout;
    l0w[g][0] = live ? L0.w[c] : 0.f; l0w[g][1] = live ? L0.w[L0.cout + c] : 0.f; l0w[g][2] = live ? L0.w[2 * L0.cout + c] : 0.f;
    l0w[g][3] = live ? L0.scale[tower * L0.cout + c] : 0.f; l0w[g][4] = live ? L0.shift[tower * L0.cout + c] : 0.f;
  }
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
  if (tile != tile_lo) __syncthreads();   // the previous tile's readers are done with the LDS buffers

  // ---- prologue: p' = (p - c) @ R   (models/tp8.py:106,113,122,127) ----
  if (tid < TP) {
    const float* xf = a.xform + (size_t)cloud * 12;
    const float x = nx - xf[0], y = ny - xf[1], z = nz - xf[2];
    xs[tid * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
    xs[tid * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
    xs[tid * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
  }
  if (tile + 1 < tile_hi) request_xyz(tile + 1);
  __syncthreads();

  // ---- layer 0: K = 3 lift on the VALU (not a dense GEMM); the first two channel groups' weights come from registers ----
  {
    const ConvLayerDev& L = a.L[0];
    float* out = smem + boff[0];
    const int ldo = lds_ld[0], c0 = tid & 31, r0 = tid >> 5;   // a 32-lane group writes one row, 32 consecutive channels
    for (int c = c0; c < ldo - 4; c += 32) {
      const bool live = c < L.cout;
      const int g = (c - c0) >> 5;
      float w0, w1, w2, sc, sh;
      if (g < 2) { w0 = g ? l0w[1][0] : l0w[0][0]; w1 = g ? l0w[1][1] : l0w[0][1]; w2 = g ? l0w[1][2] : l0w[0][2];
                   sc = g ? l0w[1][3] : l0w[0][3]; sh = g ? l0w[1][4] : l0w[0][4]; }
      else { w0 = live ? L.w[c] : 0.f; w1 = live ? L.w[L.cout + c] : 0.f; w2 = live ? L.w[2 * L.cout + c] : 0.f;
             sc = live ? L.scale[tower * L.cout + c] : 0.f; sh = live ? L.shift[tower * L.cout + c] : 0.f; }
#pragma unroll
      for (int rr = 0; rr < TP / 16; ++rr) {
        const int row = rr * 16 + r0;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float acc = fmaf(p[2], w2, fmaf(p[1], w1, p[0] * w0));
        out[row * ldo + c] = fmaxf(fmaf(acc, sc, sh), 0.f);
      }
    }
  }
  __syncthreads();

  // ---- hidden MFMA layers 1 .. nlayers-2 ----
  if constexpr (KGL != 0 && LD0 != 0 && LD1 != 0) {
    // shipped shape (three conv layers, widths LD0 - 4 -> LD1 - 4): one (channel tile, 64-row group) item per wave, everything constant
    constexpr int CTH = (LD1 - 4) / 32, RGH = (TP / 32) / 2;
    for (int item = wave; item < CTH * RGH; item += kWaves)
      hidden_item<2, (LD0 - 4) / 8>(smem + boff[0], LD0, smem + boff[1], LD1, a.L[1], tower, item / RGH, item % RGH, lane);
    __syncthreads();
  } else
  for (int l = 1; l < a.nlayers - 1; ++l) {
    const ConvLayerDev& L = a.L[l];
    const int CT = (L.cout + 31) >> 5;
    const float* in = smem + (((l - 1) & 1) ? boff[1] : boff[0]);
    float* out = smem + ((l & 1) ? boff[1] : boff[0]);
    const int ldi = lds_ld[(l - 1) & 1], ldo = lds_ld[l & 1];
    constexpr int MT = TP / 32;
    if (CT >= kWaves) hidden_layer<MT, TP>(in, ldi, out, ldo, L, tower, wave, lane);
    else if (CT * 2 >= kWaves || MT < 4) hidden_layer<(MT >= 2 ? 2 : 1), TP>(in, ldi, out, ldo, L, tower, wave, lane);
    else hidden_layer<1, TP>(in, ldi, out, ldo, L, tower, wave, lane);
    __syncthreads();
  }

  // ---- last layer + max over the tile's points (utils/tf_util.py:350-373) ----
  {
    const int l = a.nlayers - 1;
    const ConvLayerDev& L = a.L[l];
    const float* in = smem + (((l - 1) & 1) ? boff[1] : boff[0]);
    const int ldi = lds_ld[(l - 1) & 1];
    // (with the strides compiled in the last layer's depth is too: three conv layers, input = buffer 1 of width LD1 - 4)
    const int KG = (LD1 && KGL) ? KGL : (L.cin + 7) >> 3, CT = (L.cout + 31) >> 5;
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
    for (int ct = wave; ct < CT; ct += kWaves) {
      f32x16 acc[TP / 32];
      // scale / shift are requested BEFORE the MFMA loop (they are older than the weight stream, so its first counted
      // wait covers them): loaded after it, their L2 round trip sat exposed in every channel tile's epilogue
      const int col = ct * 32 + (lane & 31);
      const bool live = col < L.cout;
      const float sc = live ? L.scale[tower * L.cout + col] : 0.f;
      const float sh = live ? L.shift[tower * L.cout + col] : 0.f;
      asm volatile("" ::: "memory");
#ifdef ALIGNNET_KSTAMP
      long long* kst = (a.stamps && blockIdx.x == 0 && blockIdx.y == 1 && tile == tile_lo + 2 && lane == 0 && (wave == 0 || wave == 4) && ct < 16)
                           ? a.stamps + ((wave >> 2) * 2 + (ct >> 3)) * 64 : nullptr;
      if (kst) *kst++ = (long long)__builtin_readcyclecounter();
      mfma_rows<TP / 32>(in, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc, kst);
#else
      mfma_rows<TP / 32>(in, ldi, reinterpret_cast<const f32x4*>(L.w) + (size_t)ct * KG * 64, KG, lane, acc);
#endif
      float mx = 0.f;   // relu folded into the max: max_n relu(v_n) = max(0, max_n v_n)
#pragma unroll
      for (int m = 0; m < TP / 32; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaf(acc[m][r], sc, sh));
      const int q = (ct - wave) / kWaves;
      if (q < kPoolRegs) {
#pragma unroll
        for (int u = 0; u < kPoolRegs; ++u)
          if (u == q) pool[u] = fmaxf(pool[u], mx);
      } else {   // very wide last layers: beyond the register slots, publish per tile
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (lane < 32 && live) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));   // >= 0: monotone bit pattern
      }
    }
  }
  }   // tile loop
  {
    const ConvLayerDev& L = a.L[a.nlayers - 1];
    const int CT = (L.cout + 31) >> 5;
    float* dst = a.pooled + tower * a.tower_stride + b * a.row_stride;
#pragma unroll
    for (int u = 0; u < kPoolRegs; ++u) {
      const int ct = wave + u * kWaves, col = ct * 32 + (lane & 31);
      if (ct < CT) {
        const float mx = fmaxf(pool[u], __shfl_xor(pool[u], 32));
        // values are >= 0, so the IEEE bit pattern is monotone as a signed int
        if (lane < 32 && col < L.cout) atomicMax(reinterpret_cast<int*>(dst + col), __float_as_int(mx));
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// fc_mfma: out[M][Nout] = act((in[M][K] @ W[K][Nout]) * scale + shift), one 32x32 output
// tile per workgroup, the four waves split K and reduce through LDS.
// rows_per_set: rows [0, rows_per_set) use BN set 0, the rest set 1 (siamese heads).
// ---------------------------------------------------------------------------------
struct FcArgs {
  const float* in; long ldin;
  const float* wp; const float* scale; const float* shift;
  float* out; long ldout;
  int M, K, Nout, relu, rows_per_set;
  // elementwise stage glue folded into the last head layer's epilogue (each was a 5 us launch of its own behind the layer):
  //   finish = 1 (stage 1, models/tp8.py:109): s1 = head + center_mean -> s1c, next frame (s1, I), pred_s1 centres
  //   finish = 3 (pair head, models/tp8.py:155-156): pred_translations = head[:, :3] + (s2c2 - s2c1), logits = head[:, 3:]
  int finish = 0; int B = 0, nb = 0;
  const float* addend = nullptr;   // finish 1: center_mean [2B][3]; finish 3: s2c [2B][3]
  float* s1c = nullptr; float* xform = nullptr;
  float* out_a = nullptr; float* out_b = nullptr;   // finish 1: pred_s1_pc1centers / pc2centers; finish 3: pred_translations / remaining logits (any may be null)
  // Two K halves (grid.z = 2, ksplit = 2; STAGE form only): a 32 x 32 tile over K = 2048 is 1024 MFMAs = 6.8 us of one CU's matrix pipe, and the pair
  // head's first layer has only 128 tiles -- half the chip idles behind them.  Each half adds its raw partial tile onto a zeroed buffer (two addends
  // onto zero: a + b = b + a, the result does not depend on who arrives first) and the NEXT layer applies this layer's folded BatchNorm + relu to its
  // A operand while it stages it (in_scale / in_shift per input column, BN set 0).
  int ksplit = 1;
  const float* in_scale = nullptr; const float* in_shift = nullptr;
};

// NW waves split K (4: the usual head layer; 8 for K >= 1024 -- the pair head's first layer, K = 2048, ran 64 k-groups per wave on half
// the chip's CUs with one wave per SIMD: 20.8 us)
// STAGE: the A operand (32 input rows x the trip's 64 k) goes through a per-wave LDS tile, read from memory as 256-byte row segments.  Read straight
// into the MFMA layout, a load instruction's 64 lanes touch 64 different 128-byte lines (16 bytes each of 32 rows, two halves), one L1 lookup per
// lane; row by row it is 8 lines per instruction.
constexpr int kFcLd = 68;   // floats per staged row: 64 + 4 (row stride = 4 mod 8 floats: conflict-free 16-byte reads, as in pointnet_fused)
template <int NW, bool STAGE = false>
static __global__ __launch_bounds__(NW * 64) void fc_mfma(const FcArgs a)
{
  __shared__ float red[NW - 1][16][64];
  __shared__ __attribute__((aligned(16))) float stageA[STAGE ? NW : 1][STAGE ? 32 * kFcLd : 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ct = blockIdx.x, mt = blockIdx.y;
  const int KG = a.K >> 3;   // K % 8 == 0 checked on the host
  const int row_in = min(mt * 32 + (lane & 31), a.M - 1);
  const float* arow = a.in + (size_t)row_in * a.ldin + (lane >> 5) * 4;
  const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp) + (size_t)ct * KG * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int KGz = KG / a.ksplit, kz0 = (int)blockIdx.z * KGz;             // this workgroup's K range (ksplit divides KG: checked on the host)
  const int per = (KGz + NW - 1) / NW, k0 = kz0 + wave * per, k1 = min(kz0 + KGz, k0 + per);
  // the folded glue's addends are requested in front of the MFMA loop (loaded in the epilogue, each row's round trip sat exposed behind it)
  float add0[16], add1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { add0[r] = 0.f; add1[r] = 0.f; }
  if (a.finish && ct == 0) {   // (uniform: only the last layer's first column tile carries the three centre / translation columns)
    const int col = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.M - 1);
      add0[r] = col < 3 ? a.addend[row * 3 + col] : 0.f;
      add1[r] = (a.finish == 3 && col < 3) ? a.addend[(a.B + row) * 3 + col] : 0.f;
    }
  }
  // the epilogue's scale / shift of both BN sets likewise (wave 0 only uses them; both sets exist for every layer: fold_for_eval)
  const int ecol = min(ct * 32 + (lane & 31), a.Nout - 1);
  const float esc0 = a.scale[ecol], esh0 = a.shift[ecol], esc1 = a.scale[a.Nout + ecol], esh1 = a.shift[a.Nout + ecol];
  // U k-groups of operands are requested before their MFMAs (a one-deep lookahead exposed one memory round trip per
  // k-group: 32 of them per wave at K = 1024)
  constexpr int U = 8;   // (sixteen with the eight-wave form was measured: 23.0 against 19.2 us for the K = 2048 layer)
  for (int kg = k0; kg < k1; kg += U) {
    f32x4 av[U], bv[U];
    if (STAGE) {
      float* st = stageA[wave];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = u * 64 + lane, r = q >> 4, c16 = q & 15;          // 16-byte chunk c16 of row r's 64-float segment
        const int kk = min(kg + (c16 >> 1), k1 - 1);                      // its k-group, clamped like the direct form
        const int row_g = min(mt * 32 + r, a.M - 1);
        av[u] = *reinterpret_cast<const f32x4*>(a.in + (size_t)row_g * a.ldin + kk * 8 + (c16 & 1) * 4);
        bv[u] = wp[(size_t)min(kg + u, k1 - 1) * 64];
      }
      if (a.in_scale) {   // the producing layer's folded BatchNorm + relu, applied on the way into the tile (see FcArgs.ksplit)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c16 = (u * 64 + lane) & 15, kk = min(kg + (c16 >> 1), k1 - 1);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(a.in_scale + kk * 8 + (c16 & 1) * 4);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(a.in_shift + kk * 8 + (c16 & 1) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) av[u][e] = fmaxf(fmaf(av[u][e], sc[e], sh[e]), 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = u * 64 + lane, r = q >> 4, c16 = q & 15;
        *reinterpret_cast<f32x4*>(st + r * kFcLd + c16 * 4) = av[u];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < U; ++u) av[u] = *reinterpret_cast<const f32x4*>(st + (lane & 31) * kFcLd + u * 8 + (lane >> 5) * 4);
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = min(kg + u, k1 - 1);
      av[u] = *reinterpret_cast<const f32x4*>(arow + kk * 8);
      bv[u] = wp[(size_t)kk * 64];
    }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kg + u < k1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][s], bv[u][s], acc, 0, 0, 0);
      }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    const int col = ct * 32 + (lane & 31);
    if (col < a.Nout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.M) {
          const int set = row >= a.rows_per_set;
          float v = acc[r];
#pragma unroll
          for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
          if (a.ksplit > 1) { unsafeAtomicAdd(a.out + (size_t)row * a.ldout + col, v); continue; }   // raw partial tile; the consumer finishes it
          v = fmaf(v, set ? esc1 : esc0, set ? esh1 : esh0);
          v = a.relu ? fmaxf(v, 0.f) : v;
          a.out[(size_t)row * a.ldout + col] = v;
          if (a.finish == 1 && col < 3) {   // row = cloud
            const int tower = row >= a.B, b = row - tower * a.B;
            const float s = v + add0[r];
            a.s1c[row * 3 + col] = s;
            a.xform[row * 12 + col] = s;
            float* oc = tower ? a.out_b : a.out_a;
            if (oc) oc[b * 3 + col] = s;
#pragma unroll
            for (int i = 0; i < 3; ++i) a.xform[row * 12 + 3 + 3 * col + i] = (i == col) ? 1.f : 0.f;   // row `col` of the identity
          } else if (a.finish == 3) {       // row = pair
            if (col < 3) { if (a.out_a) a.out_a[row * 3 + col] = v + (add1[r] - add0[r]); }
            else if (a.out_b) a.out_b[(size_t)row * 2 * a.nb + (col - 3)] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// stage finish kernels (one thread per cloud / pair)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float floor_modf(float x, float y)   // tf.mod
{
  float r = fmodf(x, y);
  if (r != 0.f && ((r < 0.f) != (y < 0.f))) r += y;
  return r;
}

// models/tp8.py:109: s1 = head + center_mean; next frame = (s1, I)
__device__ __forceinline__ void stage1_finish_cloud(const float* __restrict__ o1, const float* __restrict__ center_mean, int B,
                                                    float* __restrict__ s1c, float* __restrict__ xform,
                                                    float* __restrict__ out_c1, float* __restrict__ out_c2, int cloud)   // one thread
{
  const int tower = cloud >= B, b = cloud - tower * B;
  float* oc = tower ? out_c2 : out_c1;
  for (int d = 0; d < 3; ++d) {
    const float v = o1[cloud * 3 + d] + center_mean[cloud * 3 + d];
    s1c[cloud * 3 + d] = v;
    xform[cloud * 12 + d] = v;
    if (oc) oc[b * 3 + d] = v;
  }
  for (int i = 0; i < 9; ++i) xform[cloud * 12 + 3 + i] = (i % 4 == 0) ? 1.f : 0.f;
}
static __global__ void stage1_finish_kernel(const float* __restrict__ o1, const float* __restrict__ center_mean, int B,
                                     float* __restrict__ s1c, float* __restrict__ xform,
                                     float* __restrict__ out_c1, float* __restrict__ out_c2)
{
  const int cloud = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloud < 2 * B) stage1_finish_cloud(o1, center_mean, B, s1c, xform, out_c1, out_c2, cloud);
}

// models/tp8.py:117-125 + :294-301,202-212: s2 centre, logits, in-graph yaw decode,
// R = rot_z(-theta) (tp8.py:26-27), next frame = (s2, R).  One wave per cloud (block = 256 threads = 4 clouds).
__device__ __forceinline__ void stage2_finish_cloud(const float* __restrict__ o2, int ldo, const float* __restrict__ s1c, int B, int nb,
                                                    float* __restrict__ s2c, float* __restrict__ xform, float* __restrict__ theta_out,
                                                    int* __restrict__ cls_out, float* __restrict__ out_c1, float* __restrict__ out_c2,
                                                    float* __restrict__ out_l1, float* __restrict__ out_l2, int cloud, int lane)   // one wave
{
  const int tower = cloud >= B, b = cloud - tower * B;
  const float* o = o2 + (size_t)cloud * ldo;
  float* oc = tower ? out_c2 : out_c1;
  float* ol = tower ? out_l2 : out_l1;
  if (lane < 3) {
    const float v = o[lane] + s1c[cloud * 3 + lane];
    s2c[cloud * 3 + lane] = v;
    xform[cloud * 12 + lane] = v;
    if (oc) oc[b * 3 + lane] = v;
  }
  const float* lg = o + 3;
  // arg-max over the class logits: first maximum wins, as tf.argmax (value desc, index asc)
  float best = -INFINITY; int cls = 0x7fffffff;
  for (int i = lane; i < nb; i += 64) { const float v = lg[i]; if (v > best) { best = v; cls = i; } }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(cls, off);
    if (ob > best || (ob == best && oi < cls)) { best = ob; cls = oi; }
  }
  if (cls >= nb) cls = 0;   // every class logit NaN (or -inf): tf.argmax returns 0 and the run continues with NaNs, it does not fault
  if (ol) for (int i = lane; i < 2 * nb; i += 64) ol[(size_t)b * 2 * nb + i] = lg[i];
  if (lane == 0) {
    const float pi = 3.14159274101257324f;   // np.float32(np.pi), tf.constant(np.pi)
    const float res = lg[nb + cls] * (pi / (float)nb);
    const float apc = 2.0f * pi / (float)nb;
    const float ang = (float)cls * apc + res;
    const float th = floor_modf(ang + pi, 2.0f * pi) - pi;
    if (theta_out) theta_out[cloud] = th;
    if (cls_out) cls_out[cloud] = cls;
    const float a = -th, c = cosf(a), sn = sinf(a);
    float* R = xform + cloud * 12 + 3;
    R[0] = c;  R[1] = -sn; R[2] = 0.f;
    R[3] = sn; R[4] = c;  R[5] = 0.f;
    R[6] = 0.f; R[7] = 0.f; R[8] = 1.f;
  }
}
static __global__ __launch_bounds__(256) void stage2_finish_kernel(const float* __restrict__ o2, int ldo, const float* __restrict__ s1c, int B, int nb,
                                     float* __restrict__ s2c, float* __restrict__ xform, float* __restrict__ theta_out,
                                     int* __restrict__ cls_out,
                                     float* __restrict__ out_c1, float* __restrict__ out_c2,
                                     float* __restrict__ out_l1, float* __restrict__ out_l2)
{
  const int cloud = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cloud < 2 * B) stage2_finish_cloud(o2, ldo, s1c, B, nb, s2c, xform, theta_out, cls_out, out_c1, out_c2, out_l1, out_l2, cloud, threadIdx.x & 63);
}

// models/tp8.py:155-156 (one thread per output element)
static __global__ void final_finish_kernel(const float* __restrict__ net, int ldn, const float* __restrict__ s2c, int B, int nb,
                                    float* __restrict__ out_t, float* __restrict__ out_l)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = 3 + 2 * nb;
  if (e >= B * w) return;
  const int b = e / w, i = e % w;
  const float v = net[(size_t)b * ldn + i];
  if (i < 3) { if (out_t) out_t[b * 3 + i] = v + (s2c[(B + b) * 3 + i] - s2c[b * 3 + i]); }
  else if (out_l) out_l[(size_t)b * 2 * nb + (i - 3)] = v;
}

}  // namespace alignnet
