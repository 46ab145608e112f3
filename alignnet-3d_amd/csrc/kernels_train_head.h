// Training kernels for the small dense parts: head MLPs (models/tp8.py:75-82), the glue between
// stages (:109,117-127,155) and the `separate` loss with its gradient (:304-354).  gfx950 only.
// These are latency-bound (M = 2B rows); the design goal is few, simple launches.
#pragma once
#include "kernels_infer.h"

namespace alignnet {

// ---------------------------------------------------------------------------------
// gemm_small: C[i][j] (+)= alpha * sum_k A(i,k) * B(k,j) (+ bias[j]) with arbitrary strides, so
// NN / TN / NT all map onto it.  One 32x32 tile per workgroup, 8 waves split K (fp32 MFMA).
// ---------------------------------------------------------------------------------
struct GemmArgs {
  const float* A; long sa_i, sa_k;
  const float* B; long sb_k, sb_j;
  float* C; long sc_i, sc_j;
  int M, N, K;
  const float* bias;   // [N] or null
  float alpha;
  int accumulate;      // C += ...
  long batch_a, batch_b, batch_c;   // element strides between batch entries (blockIdx.z)
  const float* kscale = nullptr;    // [K] or null: C = sum_k A(i,k) kscale[k] B(k,j) -- e.g. Q = W diag(E) W^T without materialising W diag(E)
  long batch_k = 0;                 // its stride between batch entries
  // optional second output: C as a bf16 MFMA B-operand image (rows of C = the operand's k, columns = its channels; layout of
  // pack_bf16_jobs_body: [ct][kg][lane][8], k = 16 kg + 8 (lane >> 5) + s, c = 32 ct + (lane & 31)) -- the backward's Q matrices are
  // consumed only in that form, and a pack launch of their own behind the product cost 5.3 us, six times per step
  unsigned short* img = nullptr; long batch_img = 0;
};

constexpr int kGemmWaves = 8, kGemmKC = 32, kGemmLd = 33;

// One 32 x 32 output tile per workgroup; K is split over 8 waves.  Each wave stages its own 32-wide K chunks of A and B
// in a private LDS region (no workgroup barrier in the loop): lanes run along whichever dimension has the smaller
// stride, so row-major, transposed and column-scaled views all load coalesced; the next chunk's 32 loads are in
// flight while the current chunk's 16 MFMAs run.
constexpr int kGemmSmemFloats = kGemmWaves * 2 * kGemmKC * kGemmLd;   // 66 KiB: two workgroups per CU
__device__ __forceinline__ void gemm_small_tile(const GemmArgs& a, int tile_x, int tile_y, int bz, float* smem)
{
  float (*stage)[2][kGemmKC * kGemmLd] = reinterpret_cast<float (*)[2][kGemmKC * kGemmLd]>(smem);   // [wave][A|B][k][i or j]; reused for the final reduction
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5;
  const int i0 = tile_y * 32, j0 = tile_x * 32;
  const float* A = a.A + bz * a.batch_a;
  const float* Bm = a.B + bz * a.batch_b;
  float* As = stage[wave][0];
  float* Bs = stage[wave][1];
  const float* ksc = a.kscale ? a.kscale + bz * a.batch_k : nullptr;
  // K range of this wave: whole chunks
  const int nchunks = (a.K + kGemmKC - 1) / kGemmKC, per = (nchunks + kGemmWaves - 1) / kGemmWaves;
  const int c0 = wave * per, c1 = min(nchunks, c0 + per);
  const bool a_kfast = a.sa_k <= a.sa_i, b_jfast = a.sb_j <= a.sb_k;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ra[16], rb[16];
  // Element addresses are fixed per lane up to the chunk offset: computed once (clamped to the matrix, so the loads need no
  // branch around them) instead of 64-bit multiply-adds and two exec-masked blocks per load in every chunk -- the address
  // arithmetic was about 500 VALU instructions per chunk and wave, more than the 16 MFMAs it feeds.
  const float* pa[16];
  const float* pb[16];
  unsigned va = 0u, vb = 0u;   // bit t: row (column) of element t is inside the matrix
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int e = lane + 64 * t;
    const int ai = a_kfast ? e >> 5 : e & 31, ak = a_kfast ? e & 31 : e >> 5;
    const int bj = b_jfast ? e & 31 : e >> 5, bk = b_jfast ? e >> 5 : e & 31;
    pa[t] = A + (size_t)min(i0 + ai, a.M - 1) * a.sa_i + (size_t)ak * a.sa_k;
    pb[t] = Bm + (size_t)bk * a.sb_k + (size_t)min(j0 + bj, a.N - 1) * a.sb_j;
    va |= (unsigned)(i0 + ai < a.M) << t;
    vb |= (unsigned)(j0 + bj < a.N) << t;
  }
  auto fetch = [&](int c) {
    const int kc = c * kGemmKC;
    if (kc + kGemmKC <= a.K) {   // whole chunk inside K (always, for K % 32 == 0)
      const size_t oa = (size_t)kc * a.sa_k, ob = (size_t)kc * a.sb_k;
      // the column scale travels with the operands (requested behind them it was a second L2 round trip per chunk: the Q = W diag(E) W^T
      // products, K = 1024, took 16 us on 32 workgroups); k-fast lanes all need the same entry, one load
      float kv[16];
      if (ksc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int e = lane + 64 * t; kv[t] = ksc[kc + (a_kfast ? e & 31 : e >> 5)]; }
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) { ra[t] = pa[t][oa]; rb[t] = pb[t][ob]; }
#pragma unroll
      for (int t = 0; t < 16; ++t) { ra[t] = (va >> t) & 1u ? ra[t] : 0.f; rb[t] = (vb >> t) & 1u ? rb[t] : 0.f; }
      if (ksc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) ra[t] *= kv[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = lane + 64 * t;
        const int ak = a_kfast ? e & 31 : e >> 5, bk = b_jfast ? e >> 5 : e & 31;
        const int ka = min(kc + ak, a.K - 1) - ak, kb = min(kc + bk, a.K - 1) - bk;   // clamped chunk offsets
        const float xa = pa[t][(size_t)ka * a.sa_k], xb = pb[t][(size_t)kb * a.sb_k];
        ra[t] = (((va >> t) & 1u) && kc + ak < a.K) ? xa * (ksc ? ksc[min(kc + ak, a.K - 1)] : 1.f) : 0.f;
        rb[t] = (((vb >> t) & 1u) && kc + bk < a.K) ? xb : 0.f;
      }
    }
  };
  if (c0 < c1) fetch(c0);
  for (int c = c0; c < c1; ++c) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = lane + 64 * t;
      const int ai = a_kfast ? e >> 5 : e & 31, ak = a_kfast ? e & 31 : e >> 5;
      const int bj = b_jfast ? e & 31 : e >> 5, bk = b_jfast ? e >> 5 : e & 31;
      As[ak * kGemmLd + ai] = ra[t];
      Bs[bk * kGemmLd + bj] = rb[t];
    }
    if (c + 1 < c1) fetch(c + 1);
    __builtin_amdgcn_wave_barrier();   // same-wave LDS operations complete in order; keep the compiler from reordering them
#pragma unroll
    for (int s2 = 0; s2 < kGemmKC / 2; ++s2) {
      const int k = 2 * s2 + half;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * kGemmLd + (lane & 31)], Bs[k * kGemmLd + (lane & 31)], acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();   // all staging regions are dead: reuse them for the cross-wave reduction
  float* red = &stage[0][0][0];   // [wave][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  // 8 waves x 2 accumulator registers each finish the tile
  const int j = j0 + (lane & 31);
  if (j < a.N) {
    const float bias = a.bias ? a.bias[j] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 16 / kGemmWaves; ++rr) {
      const int r = wave * (16 / kGemmWaves) + rr;
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < a.M) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kGemmWaves; ++w) v += red[(w * 16 + r) * 64 + lane];
        v = v * a.alpha + bias;
        float* dst = a.C + bz * a.batch_c + (size_t)row * a.sc_i + (size_t)j * a.sc_j;
        *dst = a.accumulate ? *dst + v : v;
        if (a.img) {
          const int KGi = (a.M + 15) >> 4;
          const size_t ii = ((((size_t)(j >> 5) * KGi + (row >> 4)) * 64 + (((row >> 3) & 1) * 32 + (j & 31))) << 3) + (row & 7);
          a.img[bz * a.batch_img + ii] = to_bf16_bits(v);
        }
      }
    }
  }
}

__global__ __launch_bounds__(kGemmWaves * 64) void gemm_small(const GemmArgs a)
{
  __shared__ float smem[kGemmSmemFloats];
  gemm_small_tile(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// The throughput form, for the products whose output is many tiles and whose K is a few hundred (the head layers' dW = x^T dz after the
// backward: M x N up to 2048 x 512 over K = 256 / 512 rows).  One 64 x 64 output tile per workgroup: wave = (quadrant q, K group g);
// an iteration stages 64 k of A and B once for all eight waves (16 loads per lane for 16 MFMAs, where the K-split form loads 32 and then
// reduces eight partial tiles through LDS), group g taking the k range [32 g, 32 g + 32) of it; double-buffered, one barrier per
// iteration.  The two groups' partial tiles are added in a fixed order (g = 0 first).  Plain products only: alpha / bias / accumulate as in
// gemm_small_tile, no kscale, no image.
constexpr int kGemmT = 64, kGemmTLd = 65, kGemmTK = 64;
static_assert(2 * 2 * kGemmTK * kGemmTLd <= kGemmSmemFloats, "the 64 x 64 form's two stage buffers fit the K-split form's LDS");
__device__ __forceinline__ void gemm_tile64(const GemmArgs& a, int tile_x, int tile_y, int bz, float* smem)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int q = wave & 3, g = wave >> 2, qi = q >> 1, qj = q & 1;
  const int i0 = tile_y * kGemmT, j0 = tile_x * kGemmT;
  const float* A = a.A + bz * a.batch_a;
  const float* Bm = a.B + bz * a.batch_b;
  const bool a_kfast = a.sa_k <= a.sa_i, b_jfast = a.sb_j <= a.sb_k;
  const float* pa[8];
  const float* pb[8];
  int la[8], lb[8];          // LDS offsets of this thread's eight A / B elements
  unsigned va = 0u, vb = 0u, ka = 0u, kb = 0u;   // bit u: row / column inside the matrix; k offsets (6 bits each would not fit: kept in la / lb)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + 512 * u;
    const int ai = a_kfast ? e >> 6 : e & 63, ak = a_kfast ? e & 63 : e >> 6;
    const int bj = b_jfast ? e & 63 : e >> 6, bk = b_jfast ? e >> 6 : e & 63;
    pa[u] = A + (size_t)min(i0 + ai, a.M - 1) * a.sa_i + (size_t)ak * a.sa_k;
    pb[u] = Bm + (size_t)bk * a.sb_k + (size_t)min(j0 + bj, a.N - 1) * a.sb_j;
    la[u] = ak * kGemmTLd + ai; lb[u] = bk * kGemmTLd + bj;
    va |= (unsigned)(i0 + ai < a.M) << u;
    vb |= (unsigned)(j0 + bj < a.N) << u;
  }
  (void)ka; (void)kb;
  float ra[8], rb[8];
  const int niter = (a.K + kGemmTK - 1) / kGemmTK;
  auto fetch = [&](int it) {
    const int kc = it * kGemmTK;
    if (kc + kGemmTK <= a.K) {
      const size_t oa = (size_t)kc * a.sa_k, ob = (size_t)kc * a.sb_k;
#pragma unroll
      for (int u = 0; u < 8; ++u) { ra[u] = pa[u][oa]; rb[u] = pb[u][ob]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { ra[u] = (va >> u) & 1u ? ra[u] : 0.f; rb[u] = (vb >> u) & 1u ? rb[u] : 0.f; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = tid + 512 * u;
        const int ak = a_kfast ? e & 63 : e >> 6, bk = b_jfast ? e >> 6 : e & 63;
        const int kka = min(kc + ak, a.K - 1) - ak, kkb = min(kc + bk, a.K - 1) - bk;   // clamped offsets
        const float xa = pa[u][(size_t)kka * a.sa_k], xb = pb[u][(size_t)kkb * a.sb_k];
        ra[u] = (((va >> u) & 1u) && kc + ak < a.K) ? xa : 0.f;
        rb[u] = (((vb >> u) & 1u) && kc + bk < a.K) ? xb : 0.f;
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  fetch(0);
  for (int it = 0; it < niter; ++it) {
    float* As = smem + (it & 1) * (2 * kGemmTK * kGemmTLd);
    float* Bs = As + kGemmTK * kGemmTLd;
#pragma unroll
    for (int u = 0; u < 8; ++u) { As[la[u]] = ra[u]; Bs[lb[u]] = rb[u]; }
    if (it + 1 < niter) fetch(it + 1);
    __syncthreads();   // (the buffer written now was last read two iterations ago: every wave has passed the barrier of the iteration in between)
    const float* Ag = As + (32 * g) * kGemmTLd + 32 * qi + (lane & 31);
    const float* Bg = Bs + (32 * g) * kGemmTLd + 32 * qj + (lane & 31);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int k = 2 * s2 + half;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ag[k * kGemmTLd], Bg[k * kGemmTLd], acc, 0, 0, 0);
    }
  }
  __syncthreads();
  float* red = smem;   // [wave][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const int j = j0 + 32 * qj + (lane & 31);
  if (j < a.N) {
    const float bias = a.bias ? a.bias[j] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * g + rr;
      const int row = i0 + 32 * qi + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < a.M) {
        const float v = (red[(q * 16 + r) * 64 + lane] + red[((q + 4) * 16 + r) * 64 + lane]) * a.alpha + bias;
        float* dst = a.C + bz * a.batch_c + (size_t)row * a.sc_i + (size_t)j * a.sc_j;
        *dst = a.accumulate ? *dst + v : v;
      }
    }
  }
}

// Two independent products in one launch (e.g. a head layer's dW = x^T dz and dx = dz W^T, 8 us each on their own): the tiles of
// job 0 come first in the linear grid, then those of job 1.
struct GemmPair { GemmArgs g[2]; int tx[2]; int n0; };
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_small2(const GemmPair p)
{
  const int job = (int)blockIdx.x >= p.n0;
  const int t = (int)blockIdx.x - job * p.n0;
  __shared__ float smem[kGemmSmemFloats];
  gemm_small_tile(p.g[job], t % p.tx[job], t / p.tx[job], 0, smem);
}

// Any number (<= kGemmJobs) of independent products in one launch: the weight-gradient GEMMs of a whole step (nine head layers,
// Ghat W of six conv layers) have nothing to wait for but the optimiser, so they run as one launch after the backward's critical
// path.  Tiles are numbered job after job (start[] = prefix sums); a job's batch entries (GemmArgs.batch_*) are consecutive tiles.
constexpr int kGemmJobs = 18;
struct GemmJobs { GemmArgs g[kGemmJobs]; int tx[kGemmJobs], ty[kGemmJobs], start[kGemmJobs + 1]; int n; };
__device__ __host__ inline bool gemm_job_is_big(const GemmArgs& g) { return !g.kscale && !g.img; }   // every plain product (a second launch for the few small ones costs more than their padding)
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_small_jobs(const GemmJobs jp)   // (2.6 KB of kernel arguments: under the 4 KB limit)
{
  __shared__ float smem[kGemmSmemFloats];
  int j = 0;
  while (j + 1 < jp.n && (int)blockIdx.x >= jp.start[j + 1]) ++j;   // uniform
  const int t = (int)blockIdx.x - jp.start[j], per = jp.tx[j] * jp.ty[j];
  gemm_small_tile(jp.g[j], (t % per) % jp.tx[j], (t % per) / jp.tx[j], t / per, smem);
}
// the same table form for the 64 x 64-tile jobs: a kernel of its own because the K-split body needs 216 registers (one workgroup per CU) and
// this one runs two workgroups per CU (together in one kernel at 128 registers the K-split body spills 91)
__global__ __launch_bounds__(kGemmWaves * 64, 4) void gemm_tile64_jobs(const GemmJobs jp)
{
  __shared__ float smem[kGemmSmemFloats];
  int j = 0;
  while (j + 1 < jp.n && (int)blockIdx.x >= jp.start[j + 1]) ++j;   // uniform
  const int t = (int)blockIdx.x - jp.start[j], per = jp.tx[j] * jp.ty[j];
  gemm_tile64(jp.g[j], (t % per) % jp.tx[j], (t % per) / jp.tx[j], t / per, smem);
}

// counter-based uniform in [0,1) for dropout when the host supplies none (tf.nn.dropout draws
// random_uniform; the stream cannot match TF's, only the distribution does)
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx)
{
  uint64_t x = seed ^ (idx * 0x9E3779B97F4A7C15ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (float)(x >> 40) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------
// BatchNorm over rows for FC layers (utils/tf_util.py:495-506) + ReLU (+ dropout, tp8.py:80-81).
// Rows [0, rows_per_set) use BN set 0, the rest set 1.  grid (ceil(C/64), nsets), block 256.
// ---------------------------------------------------------------------------------
struct BnRowsArgs {
  const float* z; float* y; int M, C, rows_per_set;
  const float* beta[2]; const float* gamma[2];
  float* mov_mean[2]; float* mov_var[2];
  float bn_decay; int update_ema;
  float* mean; float* var;          // [nsets][C]
  float keep;                       // <= 0: no dropout
  const float* u; long u_set_stride;   // uniforms [rows][C] per set (set s at u + s*u_set_stride), or null
  uint64_t seed;
  // sync_bn (batch statistics over all data-parallel ranks): mode 1 = only the column sums -> totals [nsets][C][2] (forward: sum z, sum z^2;
  // backward: sum g, sum g zhat), mode 2 = take them (all-reduced) from totals and do the rest, with world x the rows; 0 = one launch
  double* totals = nullptr; int mode = 0; int world = 1;
  // split-K producer (head_fwd_train: the pair head's first layer, K = 2048 on 128 tiles): z holds the first K half's product, z2 the second's,
  // neither has the bias; the statistics pass forms z = z + z2 + zbias[c] and writes it back (each element belongs to one thread)
  const float* z2 = nullptr; const float* zbias = nullptr; float* zw = nullptr;
};

__device__ __forceinline__ float dropout_scale(const BnRowsArgs& a, int set, int row_in_set, int c)
{
  if (a.keep <= 0.f) return 1.f;
  const float u = a.u ? a.u[(size_t)set * a.u_set_stride + (size_t)row_in_set * a.C + c]
                      : hash_uniform(a.seed + set, (uint64_t)row_in_set * a.C + c);
  return floorf(a.keep + u) / a.keep;   // tf.nn.dropout: x / keep * floor(keep + u)
}

constexpr int kBnCols = 32, kBnGroups = 32, kBnU = 8;   // block = 32 columns x 32 row groups; 8 independent loads per batch

__global__ __launch_bounds__(kBnCols * kBnGroups) void bn_rows_fwd_kernel(const BnRowsArgs a)
{
  __shared__ double red[kBnGroups][kBnCols][2];
  const int cl = threadIdx.x % kBnCols, rg = threadIdx.x / kBnCols, c = blockIdx.x * kBnCols + cl, set = blockIdx.y;
  const int r0 = set * a.rows_per_set, r1 = min(a.M, r0 + a.rows_per_set), R = r1 - r0;
  double s = 0.0, ss = 0.0;
  // the first trip's rows stay in registers for the second pass (with <= 256 rows per set -- every head of the shipped batch -- that is all of them:
  // the apply pass then starts without its own L2 round trip)
  float v0[kBnU];
  bool have0 = false;
  if (c < a.C && a.mode != 2)
    for (int r = r0 + rg; r < r1; r += kBnGroups * kBnU) {
      float v[kBnU];
#pragma unroll
      for (int u = 0; u < kBnU; ++u) { const int ru = r + u * kBnGroups; v[u] = ru < r1 ? a.z[(size_t)ru * a.C + c] : 0.f; }
      if (a.z2) {
        float v2[kBnU];
        const float bias = a.zbias[c];
#pragma unroll
        for (int u = 0; u < kBnU; ++u) { const int ru = r + u * kBnGroups; v2[u] = ru < r1 ? a.z2[(size_t)ru * a.C + c] : 0.f; }
#pragma unroll
        for (int u = 0; u < kBnU; ++u) {
          const int ru = r + u * kBnGroups;
          if (ru < r1) { v[u] = (v[u] + v2[u]) + bias; a.zw[(size_t)ru * a.C + c] = v[u]; }
        }
      }
      if (r == r0 + rg) {
        have0 = true;
#pragma unroll
        for (int u = 0; u < kBnU; ++u) v0[u] = v[u];
      }
#pragma unroll
      for (int u = 0; u < kBnU; ++u) { s += (double)v[u]; ss += (double)v[u] * (double)v[u]; }
    }
  red[rg][cl][0] = s; red[rg][cl][1] = ss;
  __syncthreads();
  if (c >= a.C) return;
  s = 0.0; ss = 0.0;
  for (int q = 0; q < kBnGroups; ++q) { s += red[q][cl][0]; ss += red[q][cl][1]; }
  if (a.mode == 1) { if (rg == 0) { a.totals[((size_t)set * a.C + c) * 2] = s; a.totals[((size_t)set * a.C + c) * 2 + 1] = ss; } return; }
  if (a.mode == 2) { s = a.totals[((size_t)set * a.C + c) * 2]; ss = a.totals[((size_t)set * a.C + c) * 2 + 1]; }
  const double Rg = (double)R * a.world;
  const double mean = s / Rg, var = fmax(ss / Rg - mean * mean, 0.0);
  const float mf = (float)mean, vf = (float)var;
  if (rg == 0) {
    a.mean[set * a.C + c] = mf; a.var[set * a.C + c] = vf;
    if (a.update_ema) {
      a.mov_mean[set][c] -= (1.f - a.bn_decay) * (a.mov_mean[set][c] - mf);
      a.mov_var[set][c] -= (1.f - a.bn_decay) * (a.mov_var[set][c] - vf);
    }
  }
  const float inv = a.gamma[set][c] * (1.0f / sqrtf(vf + kBnEps)), sh = a.beta[set][c] - mf * inv;
  for (int r = r0 + rg; r < r1; r += kBnGroups * kBnU) {
    float v[kBnU];
    if (r == r0 + rg && have0) {
#pragma unroll
      for (int u = 0; u < kBnU; ++u) v[u] = v0[u];
    } else {
#pragma unroll
      for (int u = 0; u < kBnU; ++u) { const int ru = r + u * kBnGroups; v[u] = ru < r1 ? a.z[(size_t)ru * a.C + c] : 0.f; }
    }
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int ru = r + u * kBnGroups;
      if (ru < r1) a.y[(size_t)ru * a.C + c] = fmaxf(fmaf(v[u], inv, sh), 0.f) * dropout_scale(a, set, ru - r0, c);
    }
  }
}

// backward: dy (wrt the layer output after relu/dropout) -> dz, dgamma, dbeta.
// dz = gamma*inv * (g - mean(g) - zhat*mean(g*zhat)), g = dy * dropout * [y > 0]
struct BnRowsBwdArgs {
  BnRowsArgs f;          // same forward description (z, stats, dropout)
  const float* dy; float* dz;
  float* dbeta[2]; float* dgamma[2];
};

__global__ __launch_bounds__(kBnCols * kBnGroups) void bn_rows_bwd_kernel(const BnRowsBwdArgs b)
{
  const BnRowsArgs& a = b.f;
  __shared__ double red[kBnGroups][kBnCols][2];
  const int cl = threadIdx.x % kBnCols, rg = threadIdx.x / kBnCols, c = blockIdx.x * kBnCols + cl, set = blockIdx.y;
  const int r0 = set * a.rows_per_set, r1 = min(a.M, r0 + a.rows_per_set), R = r1 - r0;
  float mf = 0.f, rstd = 0.f, gam = 0.f, bet = 0.f;
  if (c < a.C) { mf = a.mean[set * a.C + c]; rstd = 1.0f / sqrtf(a.var[set * a.C + c] + kBnEps); gam = a.gamma[set][c]; bet = a.beta[set][c]; }
  // the relu mask [y > 0] from the expression bn_rows_fwd_kernel evaluated (y = relu(fma(z, inv, sh))): written as fma(zhat, gamma, beta) it
  // disagreed with the forward on pre-activations within a rounding of zero -- a unit "on" in the forward and "off" in the backward
  const float inv_f = gam * rstd, sh_f = bet - mf * inv_f;
  double sb = 0.0, sg = 0.0;
  float zh0[kBnU], g0[kBnU];   // zhat and the masked gradient of the first trip's rows, kept for the second pass (see bn_rows_fwd_kernel)
  bool have0 = false;
  if (c < a.C && a.mode != 2)
    for (int r = r0 + rg; r < r1; r += kBnGroups * kBnU) {
      float zv[kBnU], dv[kBnU];
#pragma unroll
      for (int u = 0; u < kBnU; ++u) {
        const int ru = r + u * kBnGroups;
        zv[u] = ru < r1 ? a.z[(size_t)ru * a.C + c] : 0.f;
        dv[u] = ru < r1 ? b.dy[(size_t)ru * a.C + c] : 0.f;
      }
      const bool first = r == r0 + rg;
      if (first) have0 = true;
#pragma unroll
      for (int u = 0; u < kBnU; ++u) {
        const int ru = r + u * kBnGroups;
        float zh = 0.f, g = 0.f;
        if (ru < r1) {
          zh = (zv[u] - mf) * rstd;
          g = fmaf(zv[u], inv_f, sh_f) > 0.f ? dv[u] * dropout_scale(a, set, ru - r0, c) : 0.f;   // the FORWARD's expression: the same sign bit for bit
          sb += g; sg += (double)g * zh;
        }
        if (first) { zh0[u] = zh; g0[u] = g; }
      }
    }
  red[rg][cl][0] = sb; red[rg][cl][1] = sg;
  __syncthreads();
  if (c >= a.C) return;
  sb = 0.0; sg = 0.0;
  for (int q = 0; q < kBnGroups; ++q) { sb += red[q][cl][0]; sg += red[q][cl][1]; }
  if (a.mode == 1) {   // local sums: the parameter gradients (the gradient all-reduce adds the ranks') and this rank's share of the totals
    if (rg == 0) { b.dbeta[set][c] = (float)sb; b.dgamma[set][c] = (float)sg; a.totals[((size_t)set * a.C + c) * 2] = sb; a.totals[((size_t)set * a.C + c) * 2 + 1] = sg; }
    return;
  }
  if (a.mode == 2) { sb = a.totals[((size_t)set * a.C + c) * 2]; sg = a.totals[((size_t)set * a.C + c) * 2 + 1]; }
  else if (rg == 0) { b.dbeta[set][c] = (float)sb; b.dgamma[set][c] = (float)sg; }
  const double Rg = (double)R * a.world;
  const float mb = (float)(sb / Rg), mg = (float)(sg / Rg), k = gam * rstd;
  for (int r = r0 + rg; r < r1; r += kBnGroups * kBnU) {
    if (r == r0 + rg && have0) {
#pragma unroll
      for (int u = 0; u < kBnU; ++u) {
        const int ru = r + u * kBnGroups;
        if (ru < r1) b.dz[(size_t)ru * a.C + c] = k * (g0[u] - mb - zh0[u] * mg);
      }
      continue;
    }
    float zv[kBnU], dv[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int ru = r + u * kBnGroups;
      zv[u] = ru < r1 ? a.z[(size_t)ru * a.C + c] : 0.f;
      dv[u] = ru < r1 ? b.dy[(size_t)ru * a.C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int ru = r + u * kBnGroups;
      if (ru < r1) {
        const float zh = (zv[u] - mf) * rstd;
        const float g = fmaf(zv[u], inv_f, sh_f) > 0.f ? dv[u] * dropout_scale(a, set, ru - r0, c) : 0.f;
        b.dz[(size_t)ru * a.C + c] = k * (g - mb - zh * mg);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// loss `separate` forward + gradient (models/tp8.py:304-354), one workgroup.
// Shapes follow the reference exactly, including the [B] - [B,1] -> [B,B] broadcasts of
// tp8.py:279 and :327 (DESIGN.md quirks ix, x).
// ---------------------------------------------------------------------------------
struct LossArgs {
  int B, nb;
  float esf, af; int accept_inverted;
  // predictions (device): cloud-major [2B] = tower 0 rows then tower 1 rows
  const float* s1c; const float* s2c;          // [2B][3]
  const float* o2; int ldo2;                   // [2B][3+2nb], logits at +3
  const float* o3; int ldo3;                   // [B][3+2nb]
  const float* theta; const int* pcls;         // [2B] decoded yaw and its arg-max class (tp8.py:294-301)
  // labels
  const float *tr, *c1, *c2, *a1, *a2;         // [B,3],[B,3],[B,3],[B,1],[B,1]
  // outputs
  float* out;                                  // [17]: loss, then the 16 summaries (tp8.py:336-353 order)
  float* d_s1c; float* d_s2c;                  // [2B][3]   (direct loss terms; d_s2c includes the pred_translations path)
  float* d_o2;                                 // [2B][3+2nb]: only the logits part is written ([3:]); [:3] zeroed
  float* d_o3;                                 // [B][3+2nb]
  float* scratch;                              // >= 8*B floats + 4*B ints worth of space
  int want_grad;
  // the pair head's output glue (kernels_infer.h: final_finish_kernel -- pred_translations = head[:, :3] + (s2c2 - s2c1), remaining-angle logits), spread over
  // loss_prep_kernel's threads: elementwise on values the loss reads anyway, a 5 us launch of its own before.  ff_net == nullptr: not folded.
  const float* ff_net = nullptr; int ff_ldn = 0; const float* ff_s2c = nullptr; int ff_B = 0; float* ff_out_t = nullptr; float* ff_out_l = nullptr;
};

__device__ __forceinline__ double block_sum(double v, double* red)
{
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// NV sums at once: one pair of barriers instead of NV (the loss kernels reduce five / six scalars per workgroup; one after the other that
// was ten / twelve barriers of a latency-bound launch).  Same per-value summation order as block_sum.  red: [waves][NV] doubles.
template <int NV>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double* red)
{
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * NV + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w * NV + q];
    v[q] = t;
  }
}

__device__ __forceinline__ float huberf(float e, float d)
{
  const float a = fabsf(e), q = fminf(a, d);
  return 0.5f * q * q + d * (a - q);
}
__device__ __forceinline__ float clipf(float e, float d) { return fminf(fmaxf(e, -d), d); }

// tf_angle2class (tp8.py:181-199), scalar
__device__ __forceinline__ void angle2class(float angle, int nb, int* cls, float* res)
{
  const float twopi = 6.28318548202514648f;   // np.float32(2*np.pi)
  const float a = floor_modf(angle, twopi);
  const float apc = twopi / (float)nb;
  const float sh = floor_modf(a + apc * 0.5f, twopi);
  const int c = (int)(sh / apc);
  *cls = c;
  *res = sh - ((float)c * apc + apc * 0.5f);
}

// The loss runs as two small launches so that the B x B work (tp8.py:279,327 broadcasts) spreads over the chip:
//   loss_prep_kernel     ceil(3B/64) blocks: per-row class / residual targets, log-sum-exp; 1 block: Huber terms + their gradients;
//                        G blocks (loss_pairs_body): for a slice of rows i, partial sums over (i, j) of the residual Huber loss and of
//                        clip(pick_j - label_ij) per column j, for both variants (theta, theta + pi) -- independent of the other blocks
//   loss_final_kernel    reduce the partials, tf.cond variant choice, totals, angle-term gradients
// Scratch layout (floats unless noted): see LossScratch.
struct LossScratch {
  float* lse;      // [3][B]
  float* pick;     // [3][2][B]
  float* lab;      // [2][2][B]      stage-2 labels per tower / variant
  int* cls;        // [3][2][B]
  float* sjf;      // [3][B]         S_j of the chosen variant (reduced over the G partials)
  float* hub;      // [8]            the five Huber means
  double* ce;      // [3][2]         mean cross-entropy per term / variant
  double* rlp;     // [G][3][2]      partial sums of huber(pick_j - label_ij)
  float* sjp;      // [G][3][2][B]   partial sums over i of clip(pick_j - label_ij, 1)
  double* cep;     // [ceil(3B/64)][3][2]  per-workgroup partial cross-entropy sums of loss_prep_kernel
};

__device__ __forceinline__ LossScratch loss_scratch(float* base, int B, int G)
{
  LossScratch s;
  s.lse = base; s.pick = s.lse + 3 * B; s.lab = s.pick + 6 * B;
  s.cls = reinterpret_cast<int*>(s.lab + 4 * B);
  s.sjf = reinterpret_cast<float*>(s.cls + 6 * B);
  s.hub = s.sjf + 3 * B;
  s.ce = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(s.hub + 8) + 7) & ~(uintptr_t)7);
  s.rlp = s.ce + 6;
  s.sjp = reinterpret_cast<float*>(s.rlp + (size_t)G * 6);
  s.cep = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(s.sjp + (size_t)G * 6 * B) + 7) & ~(uintptr_t)7);
  return s;
}

__device__ __forceinline__ const float* term_logits(const LossArgs& a, int term, int row)
{
  return term == 0 ? a.o2 + (size_t)row * a.ldo2 + 3 : term == 1 ? a.o2 + (size_t)(a.B + row) * a.ldo2 + 3 : a.o3 + (size_t)row * a.ldo3 + 3;
}

__device__ __forceinline__ void loss_pairs_body(const LossArgs& a, int G, int g, double* red);
// grid: ceil(3B/64) workgroups for the softmax rows (16 lanes per (term, row)) + one workgroup for the Huber terms + G for the B x B part
__global__ __launch_bounds__(1024) void loss_prep_kernel(const LossArgs a, int G, int nprep)
{
  __shared__ double red[16 * 6];
  const int B = a.B, nb = a.nb, tid = threadIdx.x, nt = blockDim.x;
  const float pi = 3.14159274101257324f, pinb = (float)(3.141592653589793 / (double)nb);
  const LossScratch S = loss_scratch(a.scratch, B, G);
  if (a.ff_net) {
    const int w = 3 + 2 * nb;
    for (int e = blockIdx.x * nt + tid; e < a.ff_B * w; e += gridDim.x * nt) {
      const int b = e / w, i = e % w;
      const float v = a.ff_net[(size_t)b * a.ff_ldn + i];
      if (i < 3) { if (a.ff_out_t) a.ff_out_t[b * 3 + i] = v + (a.ff_s2c[(a.ff_B + b) * 3 + i] - a.ff_s2c[b * 3 + i]); }
      else if (a.ff_out_l) a.ff_out_l[(size_t)b * 2 * nb + (i - 3)] = v;
    }
  }
  if ((int)blockIdx.x > nprep) { loss_pairs_body(a, G, (int)blockIdx.x - nprep - 1, red); return; }
  if ((int)blockIdx.x == nprep) {
    // ---- Huber terms (tp8.py:312-323) ----
    double h[5] = {0, 0, 0, 0, 0};
    for (int e = tid; e < 3 * B; e += nt) {
      const int b = e / 3, d = e % 3;
      const float e1a = a.s1c[b * 3 + d] - a.c1[e], e1b = a.s1c[(B + b) * 3 + d] - a.c2[e];
      const float e2a = a.s2c[b * 3 + d] - a.c1[e], e2b = a.s2c[(B + b) * 3 + d] - a.c2[e];
      const float pt = a.o3[(size_t)b * a.ldo3 + d] + (a.s2c[(B + b) * 3 + d] - a.s2c[b * 3 + d]);
      const float e3 = pt - a.tr[e];
      h[0] += huberf(e1a, 1.f); h[1] += huberf(e1b, 1.f); h[2] += huberf(e2a, 1.f); h[3] += huberf(e2b, 1.f); h[4] += huberf(e3, 2.f);
      if (a.want_grad) {
        const float w = 1.0f / ((float)B * 3.0f * (float)B);   // (1/B) * mean over 3B elements
        const float g3 = w * clipf(e3, 2.f);
        a.d_s1c[b * 3 + d] = w * a.esf * 0.5f * clipf(e1a, 1.f);
        a.d_s1c[(B + b) * 3 + d] = w * a.esf * 0.5f * clipf(e1b, 1.f);
        a.d_s2c[b * 3 + d] = w * a.esf * 0.5f * clipf(e2a, 1.f) - g3;      // pred_translations = head + (s2c2 - s2c1)
        a.d_s2c[(B + b) * 3 + d] = w * a.esf * 0.5f * clipf(e2b, 1.f) + g3;
        a.d_o3[(size_t)b * a.ldo3 + d] = g3;
        a.d_o2[(size_t)b * a.ldo2 + d] = 0.f;
        a.d_o2[(size_t)(B + b) * a.ldo2 + d] = 0.f;
      }
    }
    block_sum_n<5>(h, red);
    if (tid == 0)
      for (int k = 0; k < 5; ++k) S.hub[k] = (float)(h[k] / (3.0 * B));
    return;
  }
  // ---- per-row class / residual targets and log-sum-exp for the three angle terms: 16 lanes per (term, row) ----
  const int nvar = a.accept_inverted ? 2 : 1;
  double ce[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  const int p = blockIdx.x * (nt >> 4) + (tid >> 4), sub = tid & 15;
  if (p < 3 * B) {
    const int term = p / B, i = p - term * B;
    const float* lg = term_logits(a, term, i);
    float m = -INFINITY;
    for (int k = sub; k < nb; k += 16) m = fmaxf(m, lg[k]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sm = 0.f;
    for (int k = sub; k < nb; k += 16) sm += expf(lg[k] - m);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    if (sub == 0) {
      const float l = m + logf(sm);
      S.lse[term * B + i] = l;
      // target angle of row i (term 2: column 0 of T, tp8.py:199 class_id[:, 0])
      const float tgt = term == 0 ? a.a1[i] : term == 1 ? a.a2[i] : (a.a2[i] - a.a1[i]) - (a.theta[B + 0] - a.theta[0]);
#pragma unroll
      for (int v = 0; v < 2; ++v) {   // (constant trip count: ce[][v] indexed by a run-time v lived in scratch memory)
        if (v >= nvar) break;
        int c; float r;
        angle2class(tgt + (v ? pi : 0.f), nb, &c, &r);
        const int cc = min(max(c, 0), nb - 1);
        S.cls[(term * 2 + v) * B + i] = cc;
        S.pick[(term * 2 + v) * B + i] = lg[nb + cc];
        if (term < 2) S.lab[(term * 2 + v) * B + i] = r / pinb;
        const double d = (double)(l - lg[cc]);
        if (term == 0) ce[0][v] += d; else if (term == 1) ce[1][v] += d; else ce[2][v] += d;
      }
    }
  }
  double cev[6] = {ce[0][0], ce[0][1], ce[1][0], ce[1][1], ce[2][0], ce[2][1]};
  block_sum_n<6>(cev, red);
  if (tid == 0)
    for (int q = 0; q < 6; ++q) S.cep[(size_t)blockIdx.x * 6 + q] = cev[q];
}

// block g of G owns rows i in [g*R, (g+1)*R); thread j-strided over columns.  Takes nothing from loss_prep's scratch: the picked residual
// logit of column j and the label of row i are re-derived from the inputs (the same expressions, so the same bits), which lets these
// blocks ride in loss_prep_kernel's launch instead of waiting for it (one launch and ~9 us less per step).
__device__ __forceinline__ void loss_pairs_body(const LossArgs& a, int G, int g, double* red)
{
  double rlv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int B = a.B, nb = a.nb, tid = threadIdx.x, nt = blockDim.x;
  const float pi = 3.14159274101257324f, pinb = (float)(3.141592653589793 / (double)nb);
  const LossScratch S = loss_scratch(a.scratch, B, G);
  const int R = (B + G - 1) / G, i0 = g * R, i1 = min(B, i0 + R);
  const int nvar = a.accept_inverted ? 2 : 1;
  const float dth0 = a.theta[B + 0] - a.theta[0];
  for (int j = tid; j < B; j += nt) {
    // pick_j as loss_prep_kernel forms it (class of row j's target angle, tp8.py:199 class_id[:, 0] for the pair term).  All six first: each
    // is a label load -> class -> logit load chain, and one (term, variant) after the other that was six dependent L2 round trips per thread
    const float a1j = a.a1[j], a2j = a.a2[j];
    const float dth = a.theta[B + j] - a.theta[j];
    float pj[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int term = q >> 1, v = q & 1;
      const float tgt = term == 0 ? a1j : term == 1 ? a2j : (a2j - a1j) - dth0;
      int cj; float rj;
      angle2class(tgt + (v ? pi : 0.f), nb, &cj, &rj);
      pj[q] = v < nvar ? term_logits(a, term, j)[nb + min(max(cj, 0), nb - 1)] : 0.f;
    }
    float sjv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hlv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = i0; i < i1; ++i) {   // (row loop outside, the six (term, variant) inside and fully unrolled: every array index is a constant -- a `continue`
                                      //  in the unrolled loop had left pj[] in scratch memory)
      const float a1i = a.a1[i], a2i = a.a2[i];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int term = q >> 1, v = q & 1;
        int c; float r;
        if (term < 2) angle2class((term == 0 ? a1i : a2i) + (v ? pi : 0.f), nb, &c, &r);
        else angle2class((a2i - a1i) - dth + (v ? pi : 0.f), nb, &c, &r);
        const float label = r / pinb;
        const float e = pj[q] - label;
        hlv[q] += huberf(e, 1.f);
        sjv[q] += clipf(e, 1.f);
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int term = q >> 1, v = q & 1;
      rlv[q] += (double)hlv[q];
      if (v < nvar) S.sjp[(((size_t)g * 3 + term) * 2 + v) * B + j] = sjv[q];
    }
  }
  block_sum_n<6>(rlv, red);
  if (tid == 0)
    for (int q = 0; q < 6; ++q)
      if ((q & 1) < nvar) S.rlp[(size_t)g * 6 + q] = rlv[q];
}

constexpr int kLossCols = 2;   // columns j per workgroup of loss_final_kernel (with 8 a thread walked four trips of three dependent-latency terms: 16 us)

// grid: ceil(B / kLossCols) workgroups of 256 threads.  Every workgroup repeats the (tiny) reduction of the partials and
// the tf.cond choice, then writes the angle-term gradients of its own columns; workgroup 0 writes the scalars.
__global__ __launch_bounds__(256) void loss_final_kernel(const LossArgs a, int G, int nprep)
{
  __shared__ float sres[3][2][3];   // [term][variant][tot, ce, rl]
  __shared__ int schosen[3];
  __shared__ float ssj[3][kLossCols];
  const int B = a.B, nb = a.nb, tid = threadIdx.x, nt = blockDim.x, ln = tid & 63, wave = tid >> 6;
  const LossScratch S = loss_scratch(a.scratch, B, G);
  const int nvar = a.accept_inverted ? 2 : 1;
  // the column sums S_j of BOTH variants are requested before the variant choice is known (behind it they were a round trip of their own):
  // thread (e = tid >> 2, gq = tid & 3) sums rows gq, gq + 4, ... of column j0 + e % kLossCols for term e / kLossCols
  float sjv[2] = {0.f, 0.f};
  if (a.want_grad && tid < 3 * kLossCols * 4) {
    const int e = tid >> 2, gq = tid & 3, term = e / kLossCols, j = blockIdx.x * kLossCols + e % kLossCols;
    if (j < B)
      for (int g = gq; g < G; g += 4) {
        sjv[0] += S.sjp[(((size_t)g * 3 + term) * 2 + 0) * B + j];
        if (nvar > 1) sjv[1] += S.sjp[(((size_t)g * 3 + term) * 2 + 1) * B + j];
      }
  }
  {   // one wave per (term, variant), two of them for waves 0 / 1 (256 threads): partials summed across the lanes; both rounds' loads first
    double rlq[2] = {0.0, 0.0}, ceq[2] = {0.0, 0.0};
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = wave + qi * (int)(nt >> 6);
      if (q < 6 && (q % 2) < nvar) {
        for (int g = ln; g < G; g += 64) rlq[qi] += S.rlp[((size_t)g * 3 + q / 2) * 2 + q % 2];
        for (int g = ln; g < nprep; g += 64) ceq[qi] += S.cep[(size_t)g * 6 + q];
      }
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = wave + qi * (int)(nt >> 6), term = q / 2, v = q % 2;
      double rl = rlq[qi], ce = ceq[qi];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { rl += __shfl_xor(rl, o); ce += __shfl_xor(ce, o); }
      if (q < 6 && ln == 0 && v < nvar) {
        const float r = (float)(rl / ((double)B * B)), c = (float)(ce / B);
        sres[term][v][0] = c + 20.0f * r; sres[term][v][1] = c; sres[term][v][2] = r;
      }
    }
  }
  __syncthreads();
  if (tid < 3) schosen[tid] = (a.accept_inverted && !(sres[tid][0][0] > sres[tid][1][0])) ? 1 : 0;   // tf.cond picks the LARGER
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    const float* hub = S.hub;
    const float s1 = (hub[0] + hub[1]) * 0.5f, s2t = (hub[2] + hub[3]) * 0.5f;
    const float A1 = sres[0][schosen[0]][0], A2 = sres[1][schosen[1]][0], A3 = sres[2][schosen[2]][0];
    const float lt = a.esf * (s1 + s2t) + hub[4];
    const float la = a.esf * ((A1 + A2) * 0.5f) + A3;
    float* o = a.out;
    o[0] = (lt + a.af * la) / (float)B;
    o[1] = lt; o[2] = la;
    o[3] = hub[0]; o[4] = hub[1]; o[5] = hub[2]; o[6] = hub[3]; o[7] = hub[4];
    for (int t = 0; t < 3; ++t) for (int k = 0; k < 3; ++k) o[8 + t * 3 + k] = sres[t][schosen[t]][k];
  }
  if (!a.want_grad) return;
  // ---- gradients of the angle terms (chosen variant only: tf.cond), columns j0 .. j0 + kLossCols ----
  const int j0 = blockIdx.x * kLossCols;
  if (tid < 3 * kLossCols * 4) {   // S_j = sum over the G row slices; 4 lanes share one sum
    const int e = tid >> 2, gq = tid & 3, term = e / kLossCols, j = j0 + e % kLossCols;
    float sj = schosen[term] ? sjv[1] : sjv[0];   // (same rows in the same order as before)
    (void)j; (void)gq;
    sj += __shfl_xor(sj, 1);
    sj += __shfl_xor(sj, 2);
    if (gq == 0) ssj[term][e % kLossCols] = sj;
  }
  __syncthreads();
  const float invB = 1.0f / (float)B;
  for (int e = tid; e < kLossCols * 2 * nb; e += nt) {
    const int jl = e / (2 * nb), j = j0 + jl, k = e % (2 * nb);
    if (j >= B) break;
    const float s3 = ssj[2][jl];   // S_j of the stage-3 term, needed by the towers' residual logits
    for (int term = 0; term < 3; ++term) {
      const int v = schosen[term];
      const float w = term < 2 ? invB * a.af * a.esf * 0.5f : invB * a.af;
      const float* lg = term_logits(a, term, j);
      float* dl = term == 0 ? a.d_o2 + (size_t)j * a.ldo2 + 3 : term == 1 ? a.d_o2 + (size_t)(B + j) * a.ldo2 + 3 : a.d_o3 + (size_t)j * a.ldo3 + 3;
      const int c = S.cls[(term * 2 + v) * B + j];
      float gval;
      if (k < nb) gval = w * invB * (expf(lg[k] - S.lse[term * B + j]) - (k == c ? 1.f : 0.f));
      else if (k - nb == c) gval = w * 20.0f * invB * invB * ssj[term][jl];
      else gval = 0.f;
      if (term < 2 && k >= nb) {
        // the stage-3 target depends on the towers' decoded yaw (tp8.py:325-327): gradient through the gathered residual
        // of the PREDICTED class; d(label_ij)/d(p2_j) = -1/(pi/nb), d(p)/d(logit) = pi/nb
        const int pc = a.pcls[term * B + j];
        if (k - nb == pc) gval += (term == 1 ? 1.f : -1.f) * invB * a.af * 20.0f * invB * invB * s3;
      }
      dl[k] = gval;
    }
  }
}

// ---------------------------------------------------------------------------------
// glue backward (one thread per cloud)
// ---------------------------------------------------------------------------------
// x3 = (p - s2c) @ R(-theta): given per-cloud  gx = sum_n dL/dx3[n,:]  and  grot = sum_n (g0*x3_1 - g1*x3_0):
//   dL/ds2c = -(gx @ R^T),  dL/dtheta = -grot,  dL/dlogit[nb + predcls] += dtheta * pi/nb   (tp8.py:298-301)
__global__ void stage3_glue_bwd_kernel(const float* __restrict__ gx, const float* __restrict__ grot,
                                       const float* __restrict__ xform, const int* __restrict__ pcls, int B, int nb,
                                       float* __restrict__ d_s2c, float* __restrict__ d_o2, int ldo2)
{
  const int cloud = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloud >= 2 * B) return;
  const float* R = xform + cloud * 12 + 3;
  const float g0 = gx[cloud * 3], g1 = gx[cloud * 3 + 1], g2 = gx[cloud * 3 + 2];
  d_s2c[cloud * 3 + 0] -= g0 * R[0] + g1 * R[1] + g2 * R[2];
  d_s2c[cloud * 3 + 1] -= g0 * R[3] + g1 * R[4] + g2 * R[5];
  d_s2c[cloud * 3 + 2] -= g0 * R[6] + g1 * R[7] + g2 * R[8];
  const float pinb = (float)(3.141592653589793 / (double)nb);
  d_o2[(size_t)cloud * ldo2 + 3 + nb + pcls[cloud]] += -grot[cloud] * pinb;
}

// s2c = o2[:, :3] + s1c (tp8.py:117): d_o2[:, :3] = d_s2c; d_s1c += d_s2c
__global__ void stage2_glue_bwd_kernel(const float* __restrict__ d_s2c, int B, float* __restrict__ d_o2, int ldo2,
                                       float* __restrict__ d_s1c)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * B * 3) return;
  const int cloud = e / 3, d = e % 3;
  d_o2[(size_t)cloud * ldo2 + d] = d_s2c[e];
  d_s1c[e] += d_s2c[e];
}

// x2 = p - s1c (tp8.py:113): d_s1c -= sum_n dL/dx2[n,:]
__global__ void stage1_glue_bwd_kernel(const float* __restrict__ gx, int B, float* __restrict__ d_s1c)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * B * 3) return;
  d_s1c[e] -= gx[e];
}

}  // namespace alignnet
