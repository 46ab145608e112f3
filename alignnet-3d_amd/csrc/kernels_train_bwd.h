// Backward of the shared-MLP backbone in training mode (autodiff of models/tp8.py:49-59 through
// utils/tf_util.py:455-492 batch-statistics BN and the max-pool), gfx950 only.
//
// Notation per tower and layer l: z = h_prev W + b, mu/var batch moments over the M = B*N rows,
// r = rsqrt(var+eps), zhat = (z-mu) r, y = gamma zhat + beta, h = relu(y), k = gamma r.
// BN backward:  dz = k (dy - dbeta/M - zhat dgamma/M),  dbeta = sum dy,  dgamma = sum dy zhat.
//
// Two exact identities remove the dense backward through the widest layer (conv3, 90 % of the FLOPs):
//  (1) max-pool: dy3 is non-zero only at the arg-extreme point n*(b,c) of each (cloud, channel);
//  (2) the remaining (BN-statistics) part of dz3 is per-channel affine in z3:  A_c + E_c z3[n,c],
//      E = -k r dgamma/M, so with Ghat = h2^T h2 - s2 s2^T/M (centred Gram) and m2 = s2/M
//        dW3 = Sp - m2 (k*dbeta)^T + (Ghat W3) diag(E),      Sp[:,c] = sum_b k g0[b,c] h2[b,n*,:]
//        dh2 = (h2 - m2) Q3 - W3 (k*dbeta)/M + sparse rows,  Q3 = W3 diag(E) W3^T   (C2 x C2)
// The same identity handles the statistics part of layer 2 (Q2, centred Gram of h1).
// Passes (one workgroup per cloud, walking its 128-point tiles, recomputing h1/h2 from xyz):
//   B2: dh2 -> dy2 (stored), dbeta2/dgamma2, U2 = h1^T dy2, Gram/colsum of h1
//   B1: dh1 = dy2 V2 + (h1 - m1) Q2 + ... -> dy1 (stored), dbeta1/dgamma1
//   B0: dz1 -> dW1 partials and the per-cloud input gradients (centre / yaw paths)
#pragma once
#include "ablate.h"
#include "kernels_train_fwd.h"
#include "kernels_train_head.h"

namespace alignnet {


struct BwdB2Args {
  const float* pcs[2]; const float* xform; int B, N, C1, C2, C3;
  int ld0, ldb;                 // LDS leading dims: h1 view, and the two big buffers
  const float* w1; const float* wp2;
  const float *sc1, *sh1;       // [2][C1]
  const float *sc2, *sh2;       // [2][C2]
  const float* b2; const float *mean2, *rstd2;   // [C2], [2][C2], [2][C2]
  const unsigned short* wp2h;   // bf16 image of W2 (train_matmul_bf16)
  const unsigned short* q3imgh; // bf16 images of Q3 per tower (tower stride q3imgh_stride elements)
  long q3imgh_stride;
  const float* q3img;           // per tower: MFMA image of Q3 [C2][C2]; tower stride q3img_stride floats
  long q3img_stride;
  const float* q3b;             // [2][C2]: -m2 Q3 - W3 (k*dbeta)/M
  const float* gs;              // [2B][C3]  k3 * g0
  const int* idx;               // [2B][C3]
  const float* w3t;             // [C3][C2]
  const unsigned short* w3th;   // bf16 [C3][C2]: round(W3)^T, rows gathered by the sparse part of the shipped-width bf16 instantiation
  float* dy2_store;             // [2B*N][C2]
  double* dbg2_part;            // [2B][2 halves][C2][2]  (dbeta2, dgamma2)
  int parts = 1;                // > 1 (never with ACCUM): a cloud's tiles are dealt to `parts` workgroups (grid 2B * parts, workgroup = cloud * parts + part); dbg2_part / s1_part are per-WORKGROUP
                                // partials then ([2B * parts] slices) -- one workgroup per cloud leaves the chip half empty below 2B = 512 clouds (the reference's shipped batch: 128)
  float* u2_part;               // [2B][C1*C2]
  float* g1_part;               // [2B][C1*C1]
  double* s1_part;              // [2B][G = 256 / C1 row groups][C1]
  int dbg;
  long long* stamps;            // debug: s_memtime stamps of wave 0 / block 0 at the phase boundaries of tile 3
  const float* h2_given;        // [2B*N][C2] (GIVEN variant)
};

// Work split in B2: item = (channel tile ct, 32-row group rg) = wave + 4*slot  (C2 <= 128 -> CT2*2 <= 8 items, two
// static slots per wave), so the wave that produced z2 for an item also owns its dh2 and keeps z2 in registers.
// LDS: xs | X [64][ldb] | Y [64][ldb] | hit list (entry, g)[C3] | per-wave tile offsets.
#define B2_STAMP(i) do { if (ALN_STAMPS(a.stamps) && blockIdx.x == 0 && tid == 0 && tile == 3) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
// BF16: the hidden layer is recomputed exactly as the bf16 forward did (bf16 h1 tile x bf16 W2 image), and the h2 Q3 product
// of dh2 takes h2 and Q3 as bf16 operands; accumulation, sparse rows, BN backward and all reductions stay fp32 / fp64.
// GIVEN (fp32, !ACCUM): Y is loaded from h2_given (the DGCNN branch's pooled edge features p = max_k h2) instead of being
// recomputed; the relu mask is p > 0 and zhat at the arg-max slot follows from p itself: acc = (p - sh) / sc.
// C1T / C2T: compile-time layer widths (0 = take them from the arguments).  With the widths -- hence the LDS row strides --
// known, the per-row tile addresses of the unrolled epilogues are immediates; as run-time values hipcc hoists them out of the
// tile loop into registers and spills (kernels_train_dgcnn.h has the measurements).
template <bool ACCUM, bool BF16 = false, bool GIVEN = false, int C1T = 0, int C2T = 0>
__global__ __launch_bounds__(kTW * 64, 2) void train_bwd_b2(const BwdB2Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const int kC1 = C1T ? C1T : a.C1, kC2 = C2T ? C2T : a.C2;
  const int ld0 = C1T ? C1T + 4 : a.ld0, ldb = (C1T && C2T) ? (C1T > C2T ? C1T : C2T) + 4 : a.ldb;
  const int ntiles = (a.N + kTT - 1) / kTT;
  const int tile0 = part * ntiles / a.parts, tile_end = (part + 1) * ntiles / a.parts;   // this workgroup's tiles (the hit lists below are built for the whole cloud: absolute tile indices)
  // SPM (the shipped widths in bf16 mode): the sparse rows of dh2 -- one non-zero of dy3 per (cloud, channel), at the arg-extreme row --
  // as a small dense product on the matrix pipe instead of a read-modify-write scatter on the VALU:
  //     dh2_sparse[64 rows, :] = S [64 x hits] . R [hits x C2],   S[row_j, j] = k3 g0 of hit j,   R[j, :] = round(W3)^T[c_j, :]
  // per chunk of 64 hits of the tile: R is gathered from the bf16 table (transposed into the B-operand layout on the way into LDS),
  // S is built as two bf16 tiles hi + lo (16 significant bits of the gradient), and 16 MFMAs per wave add the product to the
  // accumulators that take h2 Q3.  The scatter it replaces gave every row to one wave (rows that win many channels: up to several
  // hundred hits on one wave) and was half of the kernel.  Deterministic: fixed summation order.
  // STDF: the fp32 instantiation of the shipped widths.  Like SPM it keeps what is tile-invariant in an LDS table filled once per cloud -- the
  // first layer's weights / scale / shift, the cloud's frame, the six per-column parameters of the two epilogues -- instead of reading each
  // from memory in every tile right where it is used (the registers to hold them across the tile loop do not exist at 247 VGPRs)
  constexpr bool STDF = !BF16 && !ACCUM && !GIVEN && C1T == 64 && C2T == 128;
  constexpr bool SPM = BF16 && !ACCUM && !GIVEN && C1T == 64 && C2T == 128;   // (on given features -- the dgcnn point conv -- it was measured too: 12 spilled registers next to the feature prefetch, 266 vs 249 us per launch)
  constexpr int kSpH = 64, kSpLd = kSpH + 8;                 // hits per chunk; row stride of the S / R^T tiles (conflict-free 16-byte reads)
  constexpr int kXbytes = SPM ? (kTT * 72 + 128 * kSpLd + kTT * kSpLd) * 2 : 0;   // h1 bf16 | R^T | S lo   (X region of the SPM layout)
  float* xs = smem;
  float* X = smem + kTT * 4;
  float* Y = SPM ? X + kXbytes / 4 : X + kTT * ldb;
  unsigned short* spRT = reinterpret_cast<unsigned short*>(X) + kTT * 72;                    // [128][kSpLd]
  unsigned short* spSl = spRT + 128 * kSpLd;                                                // [64][kSpLd]
  unsigned short* spSh = reinterpret_cast<unsigned short*>(Y) + kTT * 136;                  // [64][kSpLd] behind the bf16 h2 tile
  int* hit_e = reinterpret_cast<int*>(Y + kTT * ldb);            // [C3] entry = channel | (row-in-tile << 16)
  float* hit_g = reinterpret_cast<float*>(hit_e + a.C3);          // [C3] k3*g0 of that channel
  int* hoff = reinterpret_cast<int*>(hit_g + a.C3);               // [8 waves][ntiles + 1] offsets into the wave's segment
  int* wtot = hoff + kTW * (ntiles + 1);                       // [8] segment sizes
  float* l1par = reinterpret_cast<float*>(wtot + kTW);          // SPM / STDF: [5][64] first-layer weights / scale / shift of this tower + the cloud's frame [12] (filled once per cloud)
  float* cpar = l1par + 336;                                   // STDF: [6][128] sc2, sh2, q3b, b2, mean2, rstd2 of this tower
  const int KG2 = (kC1 + 7) >> 3, CT1 = (kC1 + 31) >> 5, CT2 = (kC2 + 31) >> 5, KGq = (kC2 + 7) >> 3;
  const f32x4* q3img = reinterpret_cast<const f32x4*>(a.q3img + tower * a.q3img_stride);
  const int sG = max(1, (kTW * 64) / kC1);
  const int K16a = (kC1 + 15) & ~15, ld0h = K16a + 8, K16b = (kC2 + 15) & ~15, ldbh = K16b + 8;   // bf16 tiles
  float* my_u2 = a.u2_part + (size_t)cloud * kC1 * kC2;
  float* my_g1 = a.g1_part + (size_t)cloud * kC1 * kC1;

  if constexpr (STDF) {
    if (tid < 12) l1par[320 + tid] = xf[tid];
    if (tid < 64) {
      l1par[tid] = a.w1[tid]; l1par[64 + tid] = a.w1[kC1 + tid]; l1par[128 + tid] = a.w1[2 * kC1 + tid];
      l1par[192 + tid] = a.sc1[tower * kC1 + tid]; l1par[256 + tid] = a.sh1[tower * kC1 + tid];
    }
    if (tid < 128) {
      cpar[tid] = a.sc2[tower * kC2 + tid]; cpar[128 + tid] = a.sh2[tower * kC2 + tid]; cpar[256 + tid] = a.q3b[tower * kC2 + tid];
      cpar[384 + tid] = a.b2[tid]; cpar[512 + tid] = a.mean2[tower * kC2 + tid]; cpar[640 + tid] = a.rstd2[tower * kC2 + tid];
    }
  }
  if constexpr (SPM) {
    if (tid < 12) l1par[320 + tid] = xf[tid];   // the cloud's frame: read from here per tile (from memory it was a round trip in front of every tile's first barrier)
    if (tid < 64) {
      l1par[tid] = a.w1[tid]; l1par[64 + tid] = a.w1[kC1 + tid]; l1par[128 + tid] = a.w1[2 * kC1 + tid];
      l1par[192 + tid] = a.sc1[tower * kC1 + tid]; l1par[256 + tid] = a.sh1[tower * kC1 + tid];
    }
    // ---- per-cloud hit list, ONE segment per tile (ordered by channel): wave w counts, then fills, the tiles t = w, w + 4, ... ----
    static_assert(!SPM || kTW == 4, "tiles are dealt to four waves");
    int* cnt = wtot;   // reuse: [ntiles] counts live in hoff's tail until the prefix sum  (hoff has kTW * (ntiles + 1) + kTW ints)
    // the cloud's arg-extreme rows and gradients ONCE into registers (lane l holds channels l, 64 + l, ...; C3 <= 1024): the two passes
    // below walk them per tile -- re-read from memory per (tile, chunk) they were 2 x (ntiles / 4) x C3 / 64 dependent L2 round trips per
    // wave in front of the first tile
    constexpr int kIdQ = 16;
    int idt[kIdQ], idrow[kIdQ]; float idg[kIdQ];
    const int nq = (a.C3 + 63) >> 6;
#pragma unroll
    for (int q = 0; q < kIdQ; ++q) {
      const int c = q * 64 + lane;
      const bool ok = q < nq && c < a.C3;
      const int id = ok ? a.idx[(size_t)cloud * a.C3 + c] : -1;
      idg[q] = ok ? a.gs[(size_t)cloud * a.C3 + c] : 0.f;
      idt[q] = id >= 0 ? id / kTT : -1; idrow[q] = id >= 0 ? id % kTT : 0;
    }
    // (only this workgroup's tiles [tile0, tile_end): with the cloud dealt to several workgroups the others' hits are not its business)
    for (int t = tile0 + wave; t < tile_end; t += kTW) {
      int n = 0;
#pragma unroll
      for (int q = 0; q < kIdQ; ++q)
        if (q < nq) n += __popcll(__ballot(idt[q] == t));
      if (lane == 0) hoff[ntiles + 1 + t] = n;
    }
    __syncthreads();
    if (tid == 0) {
      int acc0 = 0;
      for (int t = tile0; t < tile_end; ++t) { hoff[t] = acc0; acc0 += hoff[ntiles + 1 + t]; }
      hoff[tile_end] = acc0;
    }
    __syncthreads();
    for (int t = tile0 + wave; t < tile_end; t += kTW) {
      int pos = hoff[t];
#pragma unroll
      for (int q = 0; q < kIdQ; ++q)
        if (q < nq) {
          const bool m = idt[q] == t;
          const unsigned long long mask = __ballot(m);
          if (m) {
            const int p = pos + __popcll(mask & ((1ull << lane) - 1ull));
            hit_e[p] = (q * 64 + lane) | (idrow[q] << 16);
            const float g = idg[q];   // split once: bf16 hi | lo << 16 (16 significant bits)
            const unsigned vh = to_bf16_bits(g), vl = to_bf16_bits(g - __uint_as_float(vh << 16));
            reinterpret_cast<unsigned*>(hit_g)[p] = vh | (vl << 16);
          }
          pos += __popcll(mask);
        }
    }
    (void)cnt;
  } else
  // ---- per-cloud hit lists: wave w owns the arg-extreme rows with (row & 7) == w, ordered by tile then channel
  //      (fixed order => deterministic summation); built once, consumed tile by tile ----
  {
    // (the cloud's arg-extreme rows once into registers, as in the SPM branch: re-read per (tile, chunk) they were (ntiles + 1) x C3 / 64
    //  dependent L2 round trips per wave in front of the first tile)
    constexpr int kIdQ = 16;
    int idr[kIdQ];
    const int nq = (a.C3 + 63) >> 6;
#pragma unroll
    for (int q = 0; q < kIdQ; ++q) {
      const int c = q * 64 + lane;
      const int id = (q < nq && c < a.C3) ? a.idx[(size_t)cloud * a.C3 + c] : -1;
      idr[q] = (id >= 0 && (id & (kTW - 1)) == wave && id >= tile0 * kTT && id < tile_end * kTT) ? id : -1;   // this wave's rows of this workgroup's tiles only
    }
    int cntw = 0;
#pragma unroll
    for (int q = 0; q < kIdQ; ++q)
      if (q < nq) cntw += __popcll(__ballot(idr[q] >= 0));
    if (lane == 0) wtot[wave] = cntw;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    int pos = 0;
    for (int t = tile0; t < tile_end; ++t) {
      if (lane == 0) hoff[wave * (ntiles + 1) + t] = woff + pos;
#pragma unroll
      for (int q = 0; q < kIdQ; ++q)
        if (q < nq) {
          const bool m = idr[q] >= 0 && (idr[q] / kTT) == t;
          const unsigned long long mask = __ballot(m);
          if (m) {
            const int p = woff + pos + __popcll(mask & ((1ull << lane) - 1ull));
            const int c = q * 64 + lane;
            hit_e[p] = c | ((idr[q] % kTT) << 16);
            hit_g[p] = a.gs[(size_t)cloud * a.C3 + c];
          }
          pos += __popcll(mask);
        }
    }
    if (lane == 0) hoff[wave * (ntiles + 1) + tile_end] = woff + pos;
  }

  f32x16 z2[2];   // [row group]
  bf16x8 w2f[SPM ? 4 : 1], q3f[SPM ? 8 : 1];   // SPM: weight fragments requested a phase ahead of their MFMAs (see the tile loop)
  const int ct = wave, col = ct * 32 + (lane & 31);
  const bool live = col < kC2;
  double db = 0.0, dg = 0.0, s1c = 0.0;
  double s1v[4] = {0.0, 0.0, 0.0, 0.0};   // bf16, !ACCUM: column sums of the rounded h1 straight from the lift (columns c0 + 32 j, row group tid >> 5)

  // SPM: tiles of one chunk of hits [hb, he), he - hb <= kSpH.  Thread roles -- gather: hit pair p = tid & 31, columns 16 q .. 16 q + 15
  // (q = tid >> 5); S build: row r = tid >> 2, hit octets 2 (tid & 3), 2 (tid & 3) + 1.
  unsigned spg0[8], spg1[8];   // the gathered 2 x 32 bytes of round(W3)^T, hits 2 p and 2 p + 1 (requested early, written into R^T by sp_write)
  unsigned spma = 0u, spmb = 0u;   // all-ones where the hit exists: applied when the rows are WRITTEN (masked here, right behind the loads, each
                                   // request waited for its four loads on the spot -- an L2 round trip in front of every tile's lift)
  auto sp_request = [&](int hb, int he) {
    const int pj = (tid & 31) * 2, q = tid >> 5;
    const int ha = hb + pj, hbb = hb + pj + 1;
    const int ca = ha < he ? (hit_e[ha] & 0xffff) : -1, cb = hbb < he ? (hit_e[hbb] & 0xffff) : -1;
    const uint4* sa = reinterpret_cast<const uint4*>(a.w3th + (size_t)max(ca, 0) * kC2 + q * 16);
    const uint4* sb = reinterpret_cast<const uint4*>(a.w3th + (size_t)max(cb, 0) * kC2 + q * 16);
    const uint4 a0 = sa[0], a1 = sa[1], b0 = sb[0], b1 = sb[1];
    spma = ca >= 0 ? 0xffffffffu : 0u; spmb = cb >= 0 ? 0xffffffffu : 0u;
    spg0[0] = a0.x; spg0[1] = a0.y; spg0[2] = a0.z; spg0[3] = a0.w; spg0[4] = a1.x; spg0[5] = a1.y; spg0[6] = a1.z; spg0[7] = a1.w;
    spg1[0] = b0.x; spg1[1] = b0.y; spg1[2] = b0.z; spg1[3] = b0.w; spg1[4] = b1.x; spg1[5] = b1.y; spg1[6] = b1.z; spg1[7] = b1.w;
  };
  auto sp_write = [&](int hb, int he) {
    {   // R^T[n][j]: one dword = hits (2 p, 2 p + 1) of column n; the two half-waves of a wave walk i in opposite halves (bank spread)
      const int pj = (tid & 31) * 2, q = tid >> 5, hsel = (tid >> 5) & 1;
      // half-wave 1 starts eight columns further on (bank spread): its registers are rotated by selects, so that every register index
      // below is a compile-time constant (hipcc folds an if / else over the two orders into one loop with a run-time index -> scratch)
      unsigned r0[8], r1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0[k] = (hsel ? spg0[(k + 4) & 7] : spg0[k]) & spma; r1[k] = (hsel ? spg1[(k + 4) & 7] : spg1[k]) & spmb; }
      unsigned short* base = spRT + (q * 16) * kSpLd + pj;
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) {
        const unsigned x0 = r0[ii >> 1], x1 = r1[ii >> 1];
        const unsigned v = (ii & 1) ? ((x0 >> 16) | (x1 & 0xffff0000u)) : ((x0 & 0xffffu) | (x1 << 16));
        *reinterpret_cast<unsigned*>(base + ((ii + 8 * hsel) & 15) * kSpLd) = v;
      }
    }
    {   // S hi / lo: S[row_j][j] = g_j.  Wave w owns rows 16 w .. 16 w + 15 of both tiles: it clears them (wide stores) and then drops
        // its hits in -- LDS operations of one wave complete in order, so no barrier between the two.  (Built row by row from the hit
        // arrays -- 32 LDS reads and 16 selects per thread -- this block was 3 - 4 k cycles of every tile.)
      constexpr int kRowQ = kSpLd * 2 / 16;   // 16-byte pieces per row (9)
      const uint4 z = {0u, 0u, 0u, 0u};
      for (int i = lane; i < 16 * kRowQ; i += 64) {
        reinterpret_cast<uint4*>(spSh + wave * 16 * kSpLd)[i] = z;
        reinterpret_cast<uint4*>(spSl + wave * 16 * kSpLd)[i] = z;
      }
      const int hi = hb + lane;
      if (hi < he) {
        const int r = hit_e[hi] >> 16;
        if ((r >> 4) == wave) {
          const unsigned pk = reinterpret_cast<const unsigned*>(hit_g)[hi];
          spSh[r * kSpLd + lane] = (unsigned short)(pk & 0xffffu);
          spSl[r * kSpLd + lane] = (unsigned short)(pk >> 16);
        }
      }
    }
  };
  auto sp_mfma = [&](int nh, f32x16 (&acc)[2]) {   // acc[m] += (S hi + S lo)[rows 32 m ..][:nh] . R^T[cols of this wave's tile]
    const unsigned short* ah = spSh + (lane & 31) * kSpLd + half * 8;
    const unsigned short* al = spSl + (lane & 31) * kSpLd + half * 8;
    const unsigned short* bb = spRT + (wave * 32 + (lane & 31)) * kSpLd + half * 8;
#pragma unroll
    for (int kg = 0; kg < kSpH / 16; ++kg)
      if (kg * 16 < nh) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bb + kg * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ah + m * 32 * kSpLd + kg * 16), bv, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(al + m * 32 * kSpLd + kg * 16), bv, acc[m], 0, 0, 0);
        }
      }
  };

  // SPM: the per-column parameters of the two epilogues are tile-invariant; loaded once per cloud (left inside the tile loop hipcc sank each
  // load into the exec-masked block of its first use -- a global round trip waited for on the spot, twice per tile)
  float pc_sc = 0.f, pc_sh = 0.f, pc_qb = 0.f, pc_bias = 0.f, pc_mu = 0.f, pc_rs = 0.f;
  if constexpr (SPM) {   // (the fp32 instantiation sits at 247 VGPRs: six more registers across the tile loop spill)
    pc_sc = a.sc2[tower * kC2 + col]; pc_sh = a.sh2[tower * kC2 + col]; pc_qb = a.q3b[tower * kC2 + col];
    pc_bias = a.b2[col]; pc_mu = a.mean2[tower * kC2 + col]; pc_rs = a.rstd2[tower * kC2 + col];
  }
  TilePoint nextp = GIVEN ? TilePoint{0.f, 0.f, 0.f} : tile_point_request(pc, a.N, tile0, tid);
  for (int tile = tile0; tile < tile_end; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    const bool first = tile == tile0;
    __syncthreads();
    B2_STAMP(0);
    const int sp_h0 = SPM ? hoff[tile] : 0, sp_h1 = SPM ? hoff[tile + 1] : 0;
    if (SPM && sp_h1 > sp_h0) {   // first chunk of this tile's hits: requested now, in LDS before the barrier in front of the dh2 MFMAs
      sp_request(sp_h0, min(sp_h1, sp_h0 + kSpH));
    }
    if (GIVEN && BF16) {
      // bf16 mode on given features: the tile is staged as the bf16 A operand of h2 Q3 only (the Y region takes the fp32 dy2 later);
      // the epilogue reads the fp32 features it needs (mask, zhat) straight from memory
      const float* src = a.h2_given + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      unsigned short* Yh = reinterpret_cast<unsigned short*>(Y);
      const int c4 = kC2 >> 2;
      for (int i = tid; i < kTT * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kC2 + q * 4);
        uint2 pk;
        pk.x = (unsigned)to_bf16_bits(v[0]) | ((unsigned)to_bf16_bits(v[1]) << 16);
        pk.y = (unsigned)to_bf16_bits(v[2]) | ((unsigned)to_bf16_bits(v[3]) << 16);
        *reinterpret_cast<uint2*>(Yh + row * ldbh + q * 4) = pk;
      }
    } else if (GIVEN) {
      const float* src = a.h2_given + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;
      for (int i = tid; i < kTT * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kC2 + q * 4);
        *reinterpret_cast<f32x4*>(Y + row * ldb + q * 4) = v;
      }
    } else {
    tile_point_store(nextp, (SPM || STDF) ? l1par + 320 : xf, xs, tid);
    if (tile + 1 < tile_end) nextp = tile_point_request(pc, a.N, tile + 1, tid);   // in flight for the whole of this tile
    if constexpr (SPM) {   // the hidden layer's four weight fragments: requested here, in flight under the barrier and the lift
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) w2f[kg] = reinterpret_cast<const bf16x8*>(a.wp2h)[((size_t)ct * 4 + kg) * 64 + lane];
    }
    __syncthreads();
    B2_STAMP(1);
    if (SPM) {
      float cst[4] = {0.f, 0.f, 0.f, 0.f};
      layer1_to_lds_bf16_par(xs, l1par, reinterpret_cast<unsigned short*>(X), ld0h, nvalid, tid, &cst);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1v[j] += (double)cst[j];
      __syncthreads();
    } else if (BF16 && !ACCUM) {   // h1 straight as a bf16 tile (the fp32 tile, its conversion pass and a barrier were 3.2 k of a tile's 19 k cycles)
      float cst[4] = {0.f, 0.f, 0.f, 0.f};
      layer1_to_lds_bf16_global(xs, a.w1, kC1, a.sc1 + tower * kC1, a.sh1 + tower * kC1, reinterpret_cast<unsigned short*>(X + (SPM ? 0 : kTT * ld0)), ld0h, K16a,
                                nvalid, tid, &cst);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1v[j] += (double)cst[j];
      __syncthreads();
    } else if (STDF) {
    layer1_to_lds_par(xs, l1par, X, ld0, nvalid, tid);
    __syncthreads();
    } else {
    layer1_to_lds_global(xs, a.w1, kC1, a.sc1 + tower * kC1, a.sh1 + tower * kC1, X, ld0, nvalid, tid);
    __syncthreads();
    }
    if (BF16 && ACCUM) {   // bf16 copy of h1 behind the fp32 one (X holds [64][ldb] floats; the fp32 h1 uses [64][ld0] of it)
      unsigned short* Xh = reinterpret_cast<unsigned short*>(X + kTT * ld0);
      const int half_cols = K16a >> 1;
      for (int i = tid; i < kTT * half_cols; i += kTW * 64) {
        const int row = i / half_cols, c = (i % half_cols) * 2;
        const float v0 = c < kC1 ? X[row * ld0 + c] : 0.f, v1 = c + 1 < kC1 ? X[row * ld0 + c + 1] : 0.f;
        *reinterpret_cast<unsigned*>(Xh + row * ld0h + c) = (unsigned)to_bf16_bits(v0) | ((unsigned)to_bf16_bits(v1) << 16);
      }
      __syncthreads();
    }
    B2_STAMP(2);

    // ---- layer 2 forward: z2 (pre-BN, minus bias) stays in registers, h2 -> Y.  Wave w owns channel tile w and both
    //      32-row groups (one weight fragment feeds two MFMAs; C2 <= 128 -> CT2 <= 4 waves) ----
    if (ct < CT2) {
      if constexpr (SPM) {
        const unsigned short* arow = reinterpret_cast<const unsigned short*>(X) + (lane & 31) * ld0h + (lane >> 5) * 8;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) z2[m][r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            z2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + m * 32 * ld0h + kg * 16), w2f[kg], z2[m], 0, 0, 0);
        // the first four of Q3's eight fragments for the product behind the next barrier: requested now, in flight under the h2 store and the
        // S / R^T build (all eight here: 256 registers and two spills)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) q3f[kg] = reinterpret_cast<const bf16x8*>(a.q3imgh + tower * a.q3imgh_stride)[((size_t)ct * 8 + kg) * 64 + lane];
      } else if (BF16)
        mfma_rows_bf16_all<2>(reinterpret_cast<const unsigned short*>(X + (SPM ? 0 : kTT * ld0)), ld0h,
                              reinterpret_cast<const bf16x8*>(a.wp2h) + (size_t)ct * (K16a >> 4) * 64, K16a >> 4, lane, z2);
      else
        mfma_rows<2, true, false>(X, ld0, reinterpret_cast<const f32x4*>(a.wp2) + (size_t)ct * KG2 * 64, KG2, lane, z2);
      const float sc = SPM ? pc_sc : STDF ? cpar[col] : (live ? a.sc2[tower * kC2 + col] : 0.f), sh = SPM ? pc_sh : STDF ? cpar[128 + col] : (live ? a.sh2[tower * kC2 + col] : 0.f);
      if (SPM) {
        // every column is live (C2 = 128 = 4 waves x 32) and rows past the cloud's end need no zeros: h2 only feeds h2 Q3 here, whose rows
        // past nvalid are masked in the epilogue (h1 of those rows is 0, so h2 = relu(shift): finite).  Branch-free: with the row test
        // hipcc built 32 exec-masked blocks (compare, save exec, branch, fma, max, convert, restore, store) per wave and tile.
        unsigned short* Yh = reinterpret_cast<unsigned short*>(Y);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Yh[acc_row(m, r, lane) * ldbh + col] = to_bf16_bits(fmaxf(fmaf(z2[m][r], sc, sh), 0.f));
      } else if (BF16) {
        if (col < K16b) {   // h2 as a bf16 tile in the Y region (row stride K16(C2) + 8 elements)
          unsigned short* Yh = reinterpret_cast<unsigned short*>(Y);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = acc_row(m, r, lane);
              Yh[row * ldbh + col] = (row < nvalid && live) ? to_bf16_bits(fmaxf(fmaf(z2[m][r], sc, sh), 0.f)) : (unsigned short)0;
            }
        }
      } else if (col < ((kC2 + 7) & ~7)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = acc_row(m, r, lane);
            Y[row * ldb + col] = row < nvalid ? fmaxf(fmaf(z2[m][r], sc, sh), 0.f) : 0.f;
          }
      }
    }
    }   // !GIVEN
    if (SPM && sp_h1 > sp_h0) sp_write(sp_h0, min(sp_h1, sp_h0 + kSpH));   // (regions disjoint from h1 / h2: no barrier needed before)
    B2_STAMP(3);
    // Gram / column sums of h1 (X): needed by the statistics part of layer 2's backward
    for (int item = wave; ACCUM && item < ((ALN_ABL(a.dbg, 32)) ? 0 : CT1 * CT1); item += kTW) {
      const int it = item / CT1, jt = item % CT1;
      const float* pa = X + half * ld0 + it * 32 + (lane & 31);
      const float* pb = X + half * ld0 + jt * 32 + (lane & 31);
      float old[16];
      tile_prefetch(my_g1, kC1, it, jt, kC1, kC1, first, lane, old);
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll 8
      for (int r = 0; r < kTT; r += 2) g = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld0], pb[r * ld0], g, 0, 0, 0);
      tile_commit(my_g1, kC1, it, jt, kC1, kC1, g, lane, old);
    }
    if (!GIVEN && !(BF16 && !ACCUM) && a.s1_part && tid < sG * kC1) {   // column sums of h1 (only when the forward did not keep them): sG row groups x C1 columns
      const int c = tid % kC1, g = tid / kC1;
      float sm = 0.f;
      if (BF16 && !ACCUM) {   // the column sums of the rounded h1: what the bf16 products of passes B2 / B1 see
        const unsigned short* Xh = reinterpret_cast<const unsigned short*>(X + kTT * ld0);
        for (int r = g; r < kTT; r += sG) sm += __uint_as_float((unsigned)Xh[r * ld0h + c] << 16);
      } else
      for (int r = g; r < kTT; r += sG) sm += X[r * ld0 + c];
      s1c += (double)sm;
    }
    __syncthreads();

    B2_STAMP(4);
    if constexpr (!SPM) {
    // ---- sparse rows of dh2: X <- 0, then X[row][:] += g * W3[:, c] for this tile's hits ----
    for (int i = tid; i < kTT * ldb / 4; i += kTW * 64) reinterpret_cast<f32x4*>(X)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // ldb % 4 == 0
    __syncthreads();
    B2_STAMP(5);
    if (ALN_ABL(a.dbg, 8)) {
      for (int base = 0; base < a.C3; base += 64) {
        const int c = base + lane;
        const int rel = c < a.C3 ? a.idx[(size_t)cloud * a.C3 + c] - tile * kTT : -1;
        const bool hit = rel >= 0 && rel < nvalid && (rel & (kTW - 1)) == wave;
        unsigned long long mask = __ballot(hit);
        while (mask) {
          const int l = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const int row = __shfl(rel, l);
          const int cc = base + l;
          const float g = a.gs[(size_t)cloud * a.C3 + cc];
          for (int k = lane; k < kC2; k += 64) X[row * ldb + k] += g * a.w3t[(size_t)cc * kC2 + k];
        }
      }
    } else {
      // one hit per half-wave (32 lanes x 4 columns), 4 hits per half per chunk: 8 W3 rows in flight per wave.
      // Hits of one wave may share a row, so the two halves apply their updates one after the other, in list order.
      const int h0 = hoff[wave * (ntiles + 1) + tile], h1 = hoff[wave * (ntiles + 1) + tile + 1];
      const int l4 = (lane & 31) * 4;
      for (int hb = h0; hb < h1; hb += 8) {
        int e[4]; float g[4]; f32x4 wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hi = hb + 2 * q + half;
          const bool ok = hi < h1;
          e[q] = ok ? hit_e[hi] : -1;
          g[q] = ok ? hit_g[hi] : 0.f;
          wv[q] = (ok && l4 < kC2) ? *reinterpret_cast<const f32x4*>(a.w3t + (size_t)(e[q] & 0xffff) * kC2 + l4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (half == hh && e[q] >= 0 && l4 < kC2) {
              f32x4* px = reinterpret_cast<f32x4*>(X + (e[q] >> 16) * ldb + l4);
              f32x4 v = *px;
              v[0] = fmaf(g[q], wv[q][0], v[0]); v[1] = fmaf(g[q], wv[q][1], v[1]);
              v[2] = fmaf(g[q], wv[q][2], v[2]); v[3] = fmaf(g[q], wv[q][3], v[3]);
              *px = v;
            }
          }
        }
      }
    }
    __syncthreads();

    }

    B2_STAMP(6);
    // ---- dh2 = sparse + q3b + h2 Q3 ; dy2 = dh2 * [y2 > 0] ; reductions ----
    if (ct < CT2) {
      const float qb = SPM ? pc_qb : STDF ? cpar[256 + col] : (live ? a.q3b[tower * kC2 + col] : 0.f);
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = SPM ? qb : (live ? X[acc_row(m, r, lane) * ldb + col] : 0.f) + qb;
      float pgv[(GIVEN && BF16) ? 32 : 1];
      if (GIVEN && BF16) {   // requested in front of the MFMAs, used behind them
        const float* src = a.h2_given + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2 + col;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = acc_row(m, r, lane);
            pgv[m * 16 + r] = (live && row < nvalid) ? src[(size_t)row * kC2] : 0.f;
          }
        asm volatile("" ::: "memory");
      }
      if constexpr (SPM) {
        const unsigned short* arow = reinterpret_cast<const unsigned short*>(Y) + (lane & 31) * ldbh + (lane >> 5) * 8;
#pragma unroll
        for (int kg = 4; kg < 8; ++kg) q3f[kg] = reinterpret_cast<const bf16x8*>(a.q3imgh + tower * a.q3imgh_stride)[((size_t)ct * 8 + kg) * 64 + lane];
#pragma unroll
        for (int kg = 0; kg < 8; ++kg)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + m * 32 * ldbh + kg * 16), q3f[kg], acc[m], 0, 0, 0);
      } else if (BF16)
        mfma_rows_bf16_all<2, false>(reinterpret_cast<const unsigned short*>(Y), ldbh,
                                     reinterpret_cast<const bf16x8*>(a.q3imgh + tower * a.q3imgh_stride) + (size_t)ct * (K16b >> 4) * 64, K16b >> 4, lane, acc);
      else
        mfma_rows<2, false, false>(Y, ldb, q3img + (size_t)ct * KGq * 64, KGq, lane, acc);
      if constexpr (SPM) {   // (CT2 == kTW: every wave is here, the barriers below are workgroup-wide)
        for (int hb = sp_h0; hb < sp_h1; hb += kSpH) {
          const int he = min(sp_h1, hb + kSpH);
          if (hb > sp_h0) {   // further chunks of a crowded tile: rebuild the three tiles
            __syncthreads();
            sp_write(hb, he);
            __syncthreads();
          }
          if (he < sp_h1) sp_request(he, min(sp_h1, he + kSpH));   // next chunk's rows travel under this chunk's MFMAs
          sp_mfma(he - hb, acc);
        }
      }
      const float sc = SPM ? pc_sc : STDF ? cpar[col] : (live ? a.sc2[tower * kC2 + col] : 0.f), sh = SPM ? pc_sh : STDF ? cpar[128 + col] : (live ? a.sh2[tower * kC2 + col] : 0.f);
      const float bias = SPM ? pc_bias : STDF ? cpar[384 + col] : (live ? a.b2[col] : 0.f), mu = SPM ? pc_mu : STDF ? cpar[512 + col] : (live ? a.mean2[tower * kC2 + col] : 0.f);
      const float rs = SPM ? pc_rs : STDF ? cpar[640 + col] : (live ? a.rstd2[tower * kC2 + col] : 0.f);
      float lb = 0.f, lg = 0.f;
      const float isc = GIVEN ? 1.0f / sc : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (GIVEN) {
            const float p = (GIVEN && BF16) ? pgv[(m * 16 + r) % ((GIVEN && BF16) ? 32 : 1)] : (live ? Y[acc_row(m, r, lane) * ldb + col] : 0.f);
            z2[m][r] = p > 0.f ? (p - sh) * isc : -INFINITY;   // the pre-BN accumulator at the arg-max slot (only used where p > 0)
          }
          const bool on = acc_row(m, r, lane) < nvalid && (GIVEN ? z2[m][r] != -INFINITY : fmaf(z2[m][r], sc, sh) > 0.f);
          const float dy = on ? acc[m][r] : 0.f;
          lb += dy; lg += (GIVEN && !on) ? 0.f : dy * ((z2[m][r] + bias - mu) * rs);
          z2[m][r] = dy;   // the registers now hold dy2
        }
      db += (double)lb; dg += (double)lg;
    }
    B2_STAMP(7);
    __syncthreads();   // everyone finished reading X (sparse) and Y (h2)
    B2_STAMP(8);

    // ---- dy2 -> Y ; h1 -> X again (only when U2 / Gram(h1) are accumulated here; pass B1 does it otherwise) ----
    if (ct < CT2 && col < ((kC2 + 7) & ~7)) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[acc_row(m, r, lane) * ldb + col] = col < kC2 ? z2[m][r] : 0.f;
    }
    if (ACCUM) layer1_to_lds_global(xs, a.w1, kC1, a.sc1 + tower * kC1, a.sh1 + tower * kC1, X, ld0, nvalid, tid);
    __syncthreads();

    B2_STAMP(9);
    // ---- store dy2 (coalesced rows) and U2 += h1^T dy2 ----
    {
      if (BF16 && !GIVEN) {   // dy2 travels to pass B1 as bf16 in this mode (half the 268 MB per stage)
        unsigned short* dsth = reinterpret_cast<unsigned short*>(a.dy2_store) + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
        const int c8 = kC2 >> 3;
        for (int i = tid; i < nvalid * c8; i += kTW * 64) {
          const int row = i / c8, q = i % c8;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(Y + row * ldb + q * 8), v1 = *reinterpret_cast<const f32x4*>(Y + row * ldb + q * 8 + 4);
          uint4 pk;
          pk.x = (unsigned)to_bf16_bits(v0[0]) | ((unsigned)to_bf16_bits(v0[1]) << 16);
          pk.y = (unsigned)to_bf16_bits(v0[2]) | ((unsigned)to_bf16_bits(v0[3]) << 16);
          pk.z = (unsigned)to_bf16_bits(v1[0]) | ((unsigned)to_bf16_bits(v1[1]) << 16);
          pk.w = (unsigned)to_bf16_bits(v1[2]) | ((unsigned)to_bf16_bits(v1[3]) << 16);
          *reinterpret_cast<uint4*>(dsth + (size_t)row * kC2 + q * 8) = pk;
        }
      } else {
      float* dst = a.dy2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;   // C2 % 4 == 0 (multiple of 32 enforced on the host)
      for (int i = tid; i < nvalid * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        *reinterpret_cast<f32x4*>(dst + (size_t)row * kC2 + q * 4) = *reinterpret_cast<const f32x4*>(Y + row * ldb + q * 4);
      }
      }
      for (int item = wave; ACCUM && item < ((ALN_ABL(a.dbg, 64)) ? 0 : CT1 * CT2); item += kTW) {
        const int it = item / CT2, jt = item % CT2;
        const float* pa = X + half * ld0 + it * 32 + (lane & 31);
        const float* pb = Y + half * ldb + jt * 32 + (lane & 31);
        float old[16];
        tile_prefetch(my_u2, kC2, it, jt, kC1, kC2, first, lane, old);
        f32x16 u;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll 8
        for (int r = 0; r < kTT; r += 2) u = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld0], pb[r * ldb], u, 0, 0, 0);
        tile_commit(my_u2, kC2, it, jt, kC1, kC2, u, lane, old);
      }
    }
    B2_STAMP(10);
  }
  if (ct < CT2 && live) {
    double* d = a.dbg2_part + (((size_t)vcloud * 2 + half) * kC2 + col) * 2;   // slice (half)
    d[0] = db; d[1] = dg;
  }
  if (!GIVEN && BF16 && !ACCUM && a.s1_part) {   // [cloud][8 row groups][C1]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (tid & 31) + 32 * j;
      if (c < kC1) a.s1_part[((size_t)vcloud * 8 + (tid >> 5)) * kC1 + c] = s1v[j];
    }
  } else if (!GIVEN && a.s1_part && tid < sG * kC1) a.s1_part[(size_t)vcloud * sG * kC1 + tid] = s1c;   // [cloud][group][C1]
}

// ---------------------------------------------------------------------------------
// B1: dh1 = dy2 V2 + h1 Q2 + q2b ; dy1 = dh1 * [h1 > 0] (stored) ; dbeta1 / dgamma1 partials
// ---------------------------------------------------------------------------------
struct BwdB1Args {
  const float* pcs[2]; const float* xform; int B, N, C1, C2;
  int ld0, ldb;
  const float* w1; const float *sc1, *sh1;
  const float* b1; const float *mean1, *rstd1;      // [C1], [2][C1], [2][C1]
  const float* v2img; const float* q2img; long v2img_stride, q2img_stride;   // per-tower MFMA images: V2 [C2][C1], Q2 [C1][C1]
  const float* q2b;                                  // [2][C1]
  const float* dy2_store;
  float* dy1_store;                                  // [2B*N][C1]
  double* dbg1_part;                                 // [2B][4 = 2 row groups x 2 halves][C1][2]
  float* u2_part; float* g1_part;                    // [2B][C1*C2], [2B][C1*C1] (upper blocks) or null: accumulated in B2
  int dy2_bf16;                                      // dy2_store holds bf16 (train_matmul_bf16)
  double* pdy_part;                                  // PDY variant: [2B][4 = 2 row groups x 2 halves][4][C1]: sum x'_d dy1 (d < 3), sum dy1
  int parts = 1;                                     // as BwdB2Args::parts: dbg1_part / u2_part / g1_part / pdy_part are per-workgroup partials
};

// PDY (C1 <= 64, one item per wave): dy1 is not stored and dbeta1 / dgamma1 are not reduced here.  The first layer is linear in
// x', so everything behind it -- dbeta1, dgamma1, dW1 and the per-cloud frame gradients -- follows from Pdy = sum x'^T dy1 (3 x C1),
// sum dy1 and the cloud's moments of x' (dg_b0_totals<3> / dg_b0_cloud<3>, kernels_train_dgcnn.h): no pass B0, no [B*N, C1]
// round trip.  Pdy rides on the accumulator layout as one more MFMA chain (the masked accumulator element r is the B operand of
// step r, A = x'^T with a row of ones).
template <int C1T = 0, int C2T = 0, bool PDY = false>   // compile-time widths (0 = from the arguments), see train_bwd_b2
__global__ __launch_bounds__(kTW * 64, 2) void train_bwd_b1(const BwdB1Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  float* xs = smem;
  float* X = smem + kTT * 4;            // h1   [64][ld0]
  const int kC1 = C1T ? C1T : a.C1, kC2 = C2T ? C2T : a.C2;
  const int ld0 = C1T ? C1T + 4 : a.ld0, ldb = C2T ? C2T + 4 : a.ldb;
  float* Y = X + kTT * ld0;             // dy2  [64][ldb]
  const int CT1 = (kC1 + 31) >> 5, KGv = (kC2 + 7) >> 3, KGq = (kC1 + 7) >> 3;
  const int nt_all = (a.N + kTT - 1) / kTT, tile0 = part * nt_all / a.parts, ntiles = (part + 1) * nt_all / a.parts;   // this workgroup's tiles [tile0, ntiles)
  const f32x4* v2img = reinterpret_cast<const f32x4*>(a.v2img + tower * a.v2img_stride);
  const f32x4* q2img = reinterpret_cast<const f32x4*>(a.q2img + tower * a.q2img_stride);
  // items: (column tile ct, 32-row group rg)
  const int nitems = CT1 * 2;
  const Layer1W l1w = layer1_load(a.w1, kC1, a.sc1 + tower * kC1, a.sh1 + tower * kC1, tid);
  // U2 = h1^T dy2 (CT1 x CT2 blocks) and the upper blocks of Gram(h1) stay in registers for the whole cloud (a per-tile
  // read-modify-write of these 48 KiB per cloud falls out of L2 with 512 clouds in flight); <= 3 blocks per wave
  constexpr int kAccSlots = 3;
  const int CT2 = (kC2 + 31) >> 5, half = lane >> 5;
  const int nblk_u = a.u2_part ? CT1 * CT2 : 0, nblk = nblk_u + ((a.u2_part && a.g1_part) ? CT1 * (CT1 + 1) / 2 : 0);   // Gram(h1) only if the forward did not keep it
  f32x16 gacc[kAccSlots];
#pragma unroll
  for (int q = 0; q < kAccSlots; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;

  f32x16 pacc;
  double pd[4] = {0.0, 0.0, 0.0, 0.0};
  if (PDY) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
  }
  const XForm XF = xform_load(xf);                        // the cloud's frame in scalar registers; the tile's points one tile ahead
  TilePoint npt = tile_point_request(pc, a.N, tile0, tid);
  for (int tile = tile0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    const bool first = tile == tile0;
    __syncthreads();
    if constexpr (C2T != 0) {
      if (!a.dy2_bf16) {
        // the fp32 dy2 tile (64 rows x C2 floats from HBM): ALL of a thread's pieces requested before the first is stored.  Written as
        // `if (row < nvalid) v = load` inside the store loop, every piece was an exec-masked block of its own -- load, wait for it, store --
        // i.e. kTT C2 / 1024 HBM round trips in a row at the head of every tile (eight for C2 = 128: a quarter of this pass).
        constexpr int c4 = C2T >> 2, kIt = kTT * c4 / (kTW * 64);
        const float* src = a.dy2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * C2T;
        f32x4 v[kIt];
#pragma unroll
        for (int j = 0; j < kIt; ++j) {
          const int i = tid + j * kTW * 64, row = i / c4, q = i % c4;
          v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)min(row, nvalid - 1) * C2T + q * 4);
        }
        tile_point_store(npt, XF, xs, tid);
        if (tile + 1 < ntiles) npt = tile_point_request(pc, a.N, tile + 1, tid);
#pragma unroll
        for (int j = 0; j < kIt; ++j) {
          const int i = tid + j * kTW * 64, row = i / c4, q = i % c4;
          *reinterpret_cast<f32x4*>(Y + row * ldb + q * 4) = row < nvalid ? v[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (C2T == 0 || a.dy2_bf16) {
      tile_point_store(npt, XF, xs, tid);
      if (tile + 1 < ntiles) npt = tile_point_request(pc, a.N, tile + 1, tid);
      if (a.dy2_bf16) {
        const unsigned short* srch = reinterpret_cast<const unsigned short*>(a.dy2_store) + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
        const int c8 = kC2 >> 3;
        for (int i = tid; i < kTT * c8; i += kTW * 64) {
          const int row = i / c8, q = i % c8;
          uint4 pk = {0u, 0u, 0u, 0u};
          if (row < nvalid) pk = *reinterpret_cast<const uint4*>(srch + (size_t)row * kC2 + q * 8);
          float* yo = Y + row * ldb + q * 8;
          *reinterpret_cast<f32x4*>(yo) = f32x4{__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                                                __uint_as_float(pk.y & 0xffff0000u)};
          *reinterpret_cast<f32x4*>(yo + 4) = f32x4{__uint_as_float(pk.z << 16), __uint_as_float(pk.z & 0xffff0000u), __uint_as_float(pk.w << 16),
                                                    __uint_as_float(pk.w & 0xffff0000u)};
        }
      } else {
      const float* src = a.dy2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC2;
      const int c4 = kC2 >> 2;
      for (int i = tid; i < kTT * c4; i += kTW * 64) {
        const int row = i / c4, q = i % c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * kC2 + q * 4);
        *reinterpret_cast<f32x4*>(Y + row * ldb + q * 4) = v;
      }
      }
    }
    __syncthreads();
    layer1_to_lds(xs, l1w, kC1, X, ld0, nvalid, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kAccSlots; ++q) {
      const int item = wave + q * kTW;
      if (item < nblk) {
        int it, jt, ldr;
        const float* pb;
        if (item < nblk_u) { it = item / CT2; jt = item % CT2; pb = Y + half * ldb; ldr = ldb; }
        else {
          int rem = item - nblk_u; it = 0;
          while (rem >= CT1 - it) { rem -= CT1 - it; ++it; }
          jt = it + rem; pb = X + half * ld0; ldr = ld0;
        }
        const float* pa = X + half * ld0 + it * 32 + (lane & 31);
        pb += jt * 32 + (lane & 31);
#pragma unroll 8
        for (int r = 0; r < kTT; r += 2) gacc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[r * ld0], pb[r * ldr], gacc[q], 0, 0, 0);
      }
    }
    for (int item = wave; item < nitems; item += kTW) {
      const int ct = item >> 1, rg = item & 1;
      const int col = ct * 32 + (lane & 31);
      const bool live = col < kC1;
      f32x16 acc[1];
      const float qb = live ? a.q2b[tower * kC1 + col] : 0.f;
      if (PDY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = qb;
        mfma_rows_deep<1, false>(Y + rg * 32 * ldb, ldb, v2img + (size_t)ct * KGv * 64, KGv, lane, acc);
        mfma_rows_deep<1, false>(X + rg * 32 * ld0, ld0, q2img + (size_t)ct * KGq * 64, KGq, lane, acc);
        const int ei = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float h1v = X[row * ld0 + (live ? col : 0)];
          const float xv = xs[row * 4 + (ei & 3)];
          const float dy = (live && row < nvalid && h1v > 0.f) ? acc[0][r] : 0.f;
          const float ea = ei < 3 ? xv : (ei == 3 ? 1.f : 0.f);
          pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea, dy, pacc, 0, 0, 0);
        }
        // fp32 sums of one tile folded into fp64: lanes of the lower half hold d = 0..2 and sum dy1 (q = 0..3), the upper half zeros
#pragma unroll
        for (int q = 0; q < 4; ++q) pd[q] += (double)pacc[q];
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
        continue;
      }
      double* dslice = a.dbg1_part + (((size_t)vcloud * 4 + rg * 2 + (lane >> 5)) * kC1 + (live ? col : 0)) * 2;   // slice (rg, half)
      const double o0 = (first || !live) ? 0.0 : dslice[0], o1 = (first || !live) ? 0.0 : dslice[1];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = qb;
      mfma_rows<1, false, false>(Y + rg * 32 * ldb, ldb, v2img + (size_t)ct * KGv * 64, KGv, lane, acc);
      mfma_rows<1, false, false>(X + rg * 32 * ld0, ld0, q2img + (size_t)ct * KGq * 64, KGq, lane, acc);
      const float w0 = live ? a.w1[col] : 0.f, wa = live ? a.w1[kC1 + col] : 0.f, wb = live ? a.w1[2 * kC1 + col] : 0.f;
      const float bias = live ? a.b1[col] : 0.f, mu = live ? a.mean1[tower * kC1 + col] : 0.f;
      const float rs = live ? a.rstd1[tower * kC1 + col] : 0.f;
      float lb = 0.f, lg = 0.f;
      float* dst = a.dy1_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * kC1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float h1v = X[row * ld0 + (live ? col : 0)];   // unconditional: behind `&&` it is an exec-masked read waited for on the spot
        const bool on = live && row < nvalid && h1v > 0.f;
        const float dy = on ? acc[0][r] : 0.f;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float z = fmaf(p[2], wb, fmaf(p[1], wa, p[0] * w0)) + bias;
        lb += dy; lg += dy * ((z - mu) * rs);
        if (live && row < nvalid) dst[(size_t)row * kC1 + col] = dy;
      }
      if (live) {
        dslice[0] = o0 + (double)lb;
        dslice[1] = o1 + (double)lg;
      }
    }
  }
  if (PDY && wave < nitems) {
    const int ct = wave >> 1, rg = wave & 1, col = ct * 32 + (lane & 31);
    if (col < kC1) {
      double* dst = a.pdy_part + ((size_t)vcloud * 4 + rg * 2 + (lane >> 5)) * 4 * kC1 + col;
#pragma unroll
      for (int d = 0; d < 4; ++d) dst[(size_t)d * kC1] = pd[d];   // the upper half-wave's slice is zero
    }
  }
#pragma unroll
  for (int q = 0; q < kAccSlots; ++q) {
    const int item = wave + q * kTW;
    if (item < nblk) {
      const float zero[16] = {};
      if (item < nblk_u) {
        tile_commit(a.u2_part + (size_t)vcloud * kC1 * kC2, kC2, item / CT2, item % CT2, kC1, kC2, gacc[q], lane, zero);
      } else {
        int rem = item - nblk_u, it = 0;
        while (rem >= CT1 - it) { rem -= CT1 - it; ++it; }
        tile_commit(a.g1_part + (size_t)vcloud * kC1 * kC1, kC1, it, it + rem, kC1, kC1, gacc[q], lane, zero);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Pass B1 on bf16 MFMA (train_matmul_bf16, the shipped widths C1 = 64, C2 = 128; replaces train_bwd_b1<64, 128, true> there).
// In bf16 mode dy2 already arrives as bf16 from pass B2, so the fp32 kernel above spent its time converting that tile to fp32
// and running 176 fp32 MFMAs (64 cycles each) per 64-row tile.  Here every operand tile is written ONCE, in bf16, by its
// producer inside the kernel -- dy2 row-major + transposed while it is staged from HBM, h1 row-major + transposed by the lift --
// and the four products run on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 5 x (4 + 8) ... 52 MFMAs of 32 cycles per tile:
//   dh1 = dy2 V2 + h1 Q2 + q2b      A = row-major tiles, B = bf16 images of V2 / Q2 (pack_bf16_jobs_kernel)
//   U2 += h1^T dy2, G1 += h1^T h1   A = h1 transposed, B = dy2 / h1 transposed, blocks register-resident for the whole cloud
// Pdy = x'^T dy1 (3 x C1) and sum dy1 stay in fp32 on the VALU (four FMAs per accumulator element; x' in bf16 would put
// 3-digit coordinates into the first layer's weight and frame gradients).  The backward treats the operand rounding as identity,
// like the rest of the bf16 mode.
// LDS (bf16): xs fp32 [64][4] | Xh [64][72] | XhT [64][72] | Yh [64][136] | YhT [128][72] | V2 image 16 KB | Q2 image 8 KB = 78 KiB: two workgroups per CU.
// ---------------------------------------------------------------------------------
constexpr int kPackBf16Jobs = 16;
struct PackBf16Jobs {
  const float* src[kPackBf16Jobs]; const float* gamma[kPackBf16Jobs]; unsigned short* dst[kPackBf16Jobs]; int K[kPackBf16Jobs], C[kPackBf16Jobs];
  const float* rowscale[kPackBf16Jobs];   // null, or [K]: element (k, c) is multiplied by rowscale[k] before rounding
  int tr[kPackBf16Jobs];                  // 1: the source is stored transposed, element (k, c) at src[c * K + k] -- e.g. V2 = (W2 diag(k2))^T straight from W2
};
// bf16 MFMA images (layout and sign folding as described in kernels_train_fwd.h; gamma null = no folding) of up to nine matrices in one launch,
// grid (blocks, jobs): a training step re-packs 9 + 3 x 6 images, each its own 4.5 us launch before
__device__ __forceinline__ void pack_bf16_jobs_body(const PackBf16Jobs& j, unsigned bx, int q, unsigned gx)
{
  const float* W = j.src[q];
  if (!W) return;
  const float* gamma = j.gamma[q];
  if (j.K[q] < 0) {   // K = -rows: a plain transposed bf16 table dst[c][k] = round(W[k][c]) (pass B2's sparse part gathers its rows)
    const int K = -j.K[q], C = j.C[q];
    for (size_t idx = bx * (size_t)256 + threadIdx.x; idx < (size_t)K * C; idx += (size_t)gx * 256) {
      const int c = idx / K, k = idx % K;
      j.dst[q][idx] = to_bf16_bits(W[(size_t)k * C + c]);
    }
    return;
  }
  const int K = j.K[q], C = j.C[q], KG = (K + 15) >> 4, CT = (C + 31) >> 5;
  const size_t total = (size_t)CT * KG * 512;
  for (size_t idx = bx * (size_t)256 + threadIdx.x; idx < total; idx += (size_t)gx * 256) {
    const int s8 = idx & 7, lane = (idx >> 3) & 63;
    const size_t t = idx >> 9;
    const int kg = t % KG, ct = t / KG;
    const int k = 16 * kg + 8 * (lane >> 5) + s8, c = 32 * ct + (lane & 31);
    float v = 0.f;
    if (k < K && c < C) {
      const float x = (j.tr[q] ? W[(size_t)c * K + k] : W[(size_t)k * C + c]) * (j.rowscale[q] ? j.rowscale[q][k] : 1.f);
      v = (!gamma || gamma[c] >= 0.f) ? x : -x;
    }
    j.dst[q][idx] = to_bf16_bits(v);
  }
}
__global__ __launch_bounds__(256) void pack_bf16_jobs_kernel(const PackBf16Jobs j) { pack_bf16_jobs_body(j, blockIdx.x, blockIdx.y, gridDim.x); }

struct BwdB1hArgs {
  const float* pcs[2]; const float* xform; int B, N;
  const float* w1; const float *sc1, *sh1;                  // [3][64], [2][64]
  const unsigned short* v2imgh; const unsigned short* q2imgh; long v2_stride, q2_stride;   // per-tower bf16 images: V2 [128 -> 64], Q2 [64 -> 64]
  const float* q2b;                                         // [2][64]
  const unsigned short* dy2_store;                          // [2B*N][128] bf16
  float* u2_part; float* g1_part;                           // [2B][64*128], [2B][64*64] (upper blocks) or null (the forward kept Gram(h1))
  double* pdy_part;                                         // [2B][4][64]: one slice per cloud (DgB0Args.slices = 1)
  int parts = 1;                                            // as BwdB2Args::parts: u2_part / g1_part / pdy_part are per-workgroup partials (DgB0Args.slices = parts)
};

__global__ __launch_bounds__(kTW * 64, 2) void train_bwd_b1_bf16(const BwdB1hArgs a)
{
  constexpr int C1 = 64, C2 = 128, ldx = C1 + 8, ldy = C2 + 8, ldT = kTT + 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vcloud = blockIdx.x, cloud = vcloud / a.parts, part = vcloud - cloud * a.parts, tower = cloud >= a.B, b = cloud - tower * a.B;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  const XForm XF = xform_load(xf);
  float* xs = smem;                                                        // [64][4] fp32
  unsigned short* Xh = reinterpret_cast<unsigned short*>(smem + kTT * 4);   // h1    [64][72]
  unsigned short* XhT = Xh + kTT * ldx;                                     // h1^T  [64][72]
  unsigned short* Yh = XhT + C1 * ldT;                                      // dy2   [64][136]
  unsigned short* YhT = Yh + kTT * ldy;                                     // dy2^T [128][72]
  // The operand images of V2 (128 -> 64: 16 KB) and Q2 (64 -> 64: 8 KB) of this tower, copied into LDS once per cloud: requested from
  // memory by every wave for every tile (twelve fragments, waited for right in front of their MFMAs) they were two exposed L2 round trips
  // per tile.  79.9 KB per workgroup: still two per CU.
  bf16x8* Vl = reinterpret_cast<bf16x8*>(YhT + C2 * ldT);                   // [2 channel tiles][8 k-groups][64 lanes]
  bf16x8* Ql = Vl + 2 * (C2 / 16) * 64;                                      // [2][4][64]
  const int nt_all = (a.N + kTT - 1) / kTT, tile0 = part * nt_all / a.parts, ntiles = (part + 1) * nt_all / a.parts;   // this workgroup's tiles [tile0, ntiles)
  {
    const bf16x8* v2img = reinterpret_cast<const bf16x8*>(a.v2imgh + tower * a.v2_stride);
    const bf16x8* q2img = reinterpret_cast<const bf16x8*>(a.q2imgh + tower * a.q2_stride);
    for (int i = tid; i < 2 * (C2 / 16) * 64; i += kTW * 64) Vl[i] = v2img[i];
    for (int i = tid; i < 2 * (C1 / 16) * 64; i += kTW * 64) Ql[i] = q2img[i];
  }
  const Layer1W l1w = layer1_load(a.w1, C1, a.sc1 + tower * C1, a.sh1 + tower * C1, tid);
  // register-resident blocks for the whole cloud: U2 (2 x 4 blocks) and, unless the forward kept it, the upper blocks of Gram(h1) (3)
  constexpr int kAccSlots = 3, CT1 = 2, CT2 = 4;
  const int nblk_u = CT1 * CT2, nblk = nblk_u + (a.g1_part ? CT1 * (CT1 + 1) / 2 : 0);
  f32x16 gacc[kAccSlots];
#pragma unroll
  for (int q = 0; q < kAccSlots; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[q][r] = 0.f;
  double pd[4] = {0.0, 0.0, 0.0, 0.0};
  const int ct = wave >> 1, rg = wave & 1, col = ct * 32 + (lane & 31);   // this wave's dh1 item: 32 rows x 32 channels
  const float qb = a.q2b[tower * C1 + col];
  // the tile's points and its dy2 rows (16 KB of bf16 from HBM) are requested one tile ahead: loaded at the head of the tile, their round trip
  // sat exposed in front of the first barrier of each of a cloud's 16 tiles
  constexpr int kDyIt = (C2 / 8) / kTW;
  TilePoint npt = tile_point_request(pc, a.N, tile0, tid);
  uint4 ndy[kDyIt];
  auto dy_request = [&](int tile) {
    const int nv = min(kTT, a.N - tile * kTT);
    const unsigned short* src = a.dy2_store + ((size_t)cloud * a.N + (size_t)tile * kTT) * C2;
#pragma unroll
    for (int it = 0; it < kDyIt; ++it) {
      ndy[it] = uint4{0u, 0u, 0u, 0u};
      if (lane < nv) ndy[it] = *reinterpret_cast<const uint4*>(src + (size_t)lane * C2 + (wave + it * kTW) * 8);
    }
  };
  dy_request(tile0);
  for (int tile = tile0; tile < ntiles; ++tile) {
    const int nvalid = min(kTT, a.N - tile * kTT);
    __syncthreads();
    tile_point_store(npt, XF, xs, tid);
    {   // dy2 tile: lane = row, wave-uniform 8-channel chunk; row-major 16-byte write + eight transposed 2-byte writes (consecutive lanes)
      const int row = lane;
#pragma unroll
      for (int it = 0; it < kDyIt; ++it) {
        const int q = wave + it * kTW;
        const uint4 pk = ndy[it];
        *reinterpret_cast<uint4*>(Yh + row * ldy + q * 8) = pk;
        const unsigned v[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          YhT[(q * 8 + 2 * e) * ldT + row] = (unsigned short)(v[e] & 0xffffu);
          YhT[(q * 8 + 2 * e + 1) * ldT + row] = (unsigned short)(v[e] >> 16);
        }
      }
    }
    if (tile + 1 < ntiles) { npt = tile_point_request(pc, a.N, tile + 1, tid); dy_request(tile + 1); }
    __syncthreads();
    {   // lift: thread (channel c0 + 32 g, rows 8 r0 .. 8 r0 + 7): eight bf16 values -> one 16-byte transposed write + eight row-major ones
      const int c0 = tid & 31, r0 = tid >> 5;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int c = c0 + 32 * g;
        unsigned short hv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = r0 * 8 + e;
          const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
          const float acc = fmaf(p[2], l1w.wb[g], fmaf(p[1], l1w.wa[g], p[0] * l1w.w0[g]));
          // rows past the cloud's end must read 0 (they are contracted over in U2 / Gram): an and-mask, not a branch -- written as a
          // conditional hipcc built sixteen exec-masked blocks (point read, lift, convert) per thread and tile
          hv[e] = (unsigned short)(to_bf16_bits(fmaxf(fmaf(acc, l1w.s[g], l1w.t[g]), 0.f)) & (row < nvalid ? 0xffffu : 0u));
          Xh[row * ldx + c] = hv[e];
        }
        uint4 pk;
        pk.x = hv[0] | ((unsigned)hv[1] << 16); pk.y = hv[2] | ((unsigned)hv[3] << 16);
        pk.z = hv[4] | ((unsigned)hv[5] << 16); pk.w = hv[6] | ((unsigned)hv[7] << 16);
        *reinterpret_cast<uint4*>(XhT + c * ldT + r0 * 8) = pk;
      }
    }
    __syncthreads();
    // ---- U2 += h1^T dy2, G1 += h1^T h1: contraction over the tile's 64 rows = 4 bf16 k-groups ----
#pragma unroll
    for (int q = 0; q < kAccSlots; ++q) {
      const int item = wave + q * kTW;
      if (item < nblk) {
        int it, jt;
        const unsigned short* pb;
        if (item < nblk_u) { it = item / CT2; jt = item % CT2; pb = YhT; }
        else {
          int rem = item - nblk_u; it = 0;
          while (rem >= CT1 - it) { rem -= CT1 - it; ++it; }
          jt = it + rem; pb = XhT;
        }
        const unsigned short* pa = XhT + (it * 32 + (lane & 31)) * ldT + half * 8;
        pb += (jt * 32 + (lane & 31)) * ldT + half * 8;
#pragma unroll
        for (int kg = 0; kg < kTT / 16; ++kg)
          gacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pa + kg * 16), *reinterpret_cast<const bf16x8*>(pb + kg * 16),
                                                            gacc[q], 0, 0, 0);
      }
    }
    // ---- dh1 = dy2 V2 + h1 Q2 + q2b for this wave's (32 rows, 32 channels); dy1 = dh1 [h1 > 0]; Pdy on the VALU ----
    {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = qb;
      {
        const unsigned short* ay = Yh + (rg * 32 + (lane & 31)) * ldy + half * 8;
        const unsigned short* ax = Xh + (rg * 32 + (lane & 31)) * ldx + half * 8;
#pragma unroll
        for (int kg = 0; kg < C2 / 16; ++kg)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ay + kg * 16), Vl[(ct * (C2 / 16) + kg) * 64 + lane], acc[0], 0, 0, 0);
#pragma unroll
        for (int kg = 0; kg < C1 / 16; ++kg)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ax + kg * 16), Ql[(ct * (C1 / 16) + kg) * 64 + lane], acc[0], 0, 0, 0);
      }
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rg * 32 + acc_row(0, r, lane);
        const unsigned short h1v = Xh[row * ldx + col];   // rows past nvalid hold 0
        const f32x4 p = *reinterpret_cast<const f32x4*>(xs + row * 4);
        const float dy = h1v != 0 ? acc[0][r] : 0.f;
        q0 = fmaf(p[0], dy, q0); q1 = fmaf(p[1], dy, q1); q2 = fmaf(p[2], dy, q2); q3 += dy;
      }
      pd[0] += (double)q0; pd[1] += (double)q1; pd[2] += (double)q2; pd[3] += (double)q3;   // one tile's fp32 sums folded into fp64
    }
  }
  {   // one slice per cloud: the two half-waves by shuffle, the two row groups of a channel tile through LDS (the consumers read
      // [2B][slices][4][64] doubles with a handful of workgroups: 4 MB at four slices was 17 us per launch)
#pragma unroll
    for (int d = 0; d < 4; ++d) pd[d] += __shfl_xor(pd[d], 32);
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);   // [2 channel tiles][4][32]
    if (rg == 1 && half == 0) {
#pragma unroll
      for (int d = 0; d < 4; ++d) red[(ct * 4 + d) * 32 + (lane & 31)] = pd[d];
    }
    __syncthreads();
    if (rg == 0 && half == 0) {
      double* dst = a.pdy_part + (size_t)vcloud * 4 * C1 + col;
#pragma unroll
      for (int d = 0; d < 4; ++d) dst[(size_t)d * C1] = pd[d] + red[(ct * 4 + d) * 32 + (lane & 31)];
    }
  }
#pragma unroll
  for (int q = 0; q < kAccSlots; ++q) {
    const int item = wave + q * kTW;
    if (item < nblk) {
      const float zero[16] = {};
      if (item < nblk_u) tile_commit(a.u2_part + (size_t)vcloud * C1 * C2, C2, item / CT2, item % CT2, C1, C2, gacc[q], lane, zero);
      else {
        int rem = item - nblk_u, it = 0;
        while (rem >= CT1 - it) { rem -= CT1 - it; ++it; }
        tile_commit(a.g1_part + (size_t)vcloud * C1 * C1, C1, it, it + rem, C1, C1, gacc[q], lane, zero);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// B0: dz1 = k1 (dy1 - dbeta1/M - zhat1 dgamma1/M); per cloud: P[d][c] = sum_n x'[n,d] dz1[n,c], S[c] = sum_n dz1[n,c]
// then  gx[d] = sum_c W1[d,c] S[c],  grot = sum_c (W1[0,c] P[1][c] - W1[1,c] P[0][c])
// grid: 2B workgroups of 256 threads
// ---------------------------------------------------------------------------------
struct BwdB0Args {
  const float* pcs[2]; const float* xform; int B, N, C1;
  const float* w1; const float* b1; const float *mean1, *rstd1, *k1;   // k1 = gamma1*rstd1 [2][C1]
  const float* dbg1;            // [2][C1][2] totals (dbeta1, dgamma1)
  double count;                 // M
  const float* dy1_store;
  float* p_part;                // [2B][3][C1]
  float* gx; float* grot;       // [2B][3], [2B]
};

__global__ __launch_bounds__(256) void train_bwd_b0(const BwdB0Args a)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cloud = blockIdx.x, tower = cloud >= a.B, b = cloud - tower * a.B, tid = threadIdx.x;
  const float* pc = a.pcs[tower] + (size_t)b * a.N * 3;
  const float* xf = a.xform + (size_t)cloud * 12;
  constexpr int kChunk = 1024;
  float* xs = smem;                                              // [kChunk][4]
  double* red = reinterpret_cast<double*>(smem + kChunk * 4);    // [256][4]
  float* fin = reinterpret_cast<float*>(red + 256 * 4);          // [C1][4]
  const int C1 = a.C1;
  float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gr = 0.f;
  for (int c0 = 0; c0 < C1; c0 += 256) {
    const int span = min(C1 - c0, 256), per = 256 / span;
    const int c = c0 + tid % span, rg = tid / span;
    const bool active = rg < per;
    const float w0 = a.w1[c], wa = a.w1[C1 + c], wb = a.w1[2 * C1 + c], bias = a.b1[c];
    const float mu = a.mean1[tower * C1 + c], rs = a.rstd1[tower * C1 + c], k = a.k1[tower * C1 + c];
    const float mb = (float)((double)a.dbg1[(tower * C1 + c) * 2] / a.count), mg = (float)((double)a.dbg1[(tower * C1 + c) * 2 + 1] / a.count);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, sz = 0.0;
    for (int base = 0; base < a.N; base += kChunk) {
      const int cnt = min(kChunk, a.N - base);
      __syncthreads();
      for (int i = tid; i < cnt; i += 256) {
        const float* p = pc + (size_t)(base + i) * 3;
        const float x = p[0] - xf[0], y = p[1] - xf[1], z = p[2] - xf[2];
        xs[i * 4 + 0] = x * xf[3] + y * xf[6] + z * xf[9];
        xs[i * 4 + 1] = x * xf[4] + y * xf[7] + z * xf[10];
        xs[i * 4 + 2] = x * xf[5] + y * xf[8] + z * xf[11];
      }
      __syncthreads();
      if (active)
        for (int i = rg; i < cnt; i += per * 8) {   // 8 independent loads in flight, fp32 partial sums folded into fp64
          float dy[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int iu = i + u * per;
            dy[u] = iu < cnt ? a.dy1_store[((size_t)cloud * a.N + base + iu) * C1 + c] : 0.f;
          }
          float q0 = 0.f, q1 = 0.f, q2 = 0.f, qs = 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int iu = i + u * per;
            if (iu < cnt) {
              const f32x4 x = *reinterpret_cast<const f32x4*>(xs + iu * 4);
              const float zz = fmaf(x[2], wb, fmaf(x[1], wa, x[0] * w0)) + bias;
              const float zh = (zz - mu) * rs;
              const float dz = k * (dy[u] - mb - zh * mg);
              q0 = fmaf(x[0], dz, q0); q1 = fmaf(x[1], dz, q1); q2 = fmaf(x[2], dz, q2); qs += dz;
            }
          }
          p0 += (double)q0; p1 += (double)q1; p2 += (double)q2; sz += (double)qs;
        }
    }
    red[tid * 4 + 0] = active ? p0 : 0.0; red[tid * 4 + 1] = active ? p1 : 0.0;
    red[tid * 4 + 2] = active ? p2 : 0.0; red[tid * 4 + 3] = active ? sz : 0.0;
    __syncthreads();
    if (tid < span) {
      double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      for (int g = 0; g < per; ++g) {
        const double* r = red + (g * span + tid) * 4;
        t0 += r[0]; t1 += r[1]; t2 += r[2]; t3 += r[3];
      }
      const int cc = c0 + tid;
      a.p_part[((size_t)cloud * 3 + 0) * C1 + cc] = (float)t0;
      a.p_part[((size_t)cloud * 3 + 1) * C1 + cc] = (float)t1;
      a.p_part[((size_t)cloud * 3 + 2) * C1 + cc] = (float)t2;
      fin[cc * 4 + 0] = (float)t0; fin[cc * 4 + 1] = (float)t1; fin[cc * 4 + 2] = (float)t2; fin[cc * 4 + 3] = (float)t3;
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int c = 0; c < C1; ++c) {
      const float w0 = a.w1[c], wa = a.w1[C1 + c], wb = a.w1[2 * C1 + c];
      gx0 += w0 * fin[c * 4 + 3]; gx1 += wa * fin[c * 4 + 3]; gx2 += wb * fin[c * 4 + 3];
      gr += w0 * fin[c * 4 + 1] - wa * fin[c * 4 + 0];
    }
    a.gx[cloud * 3 + 0] = gx0; a.gx[cloud * 3 + 1] = gx1; a.gx[cloud * 3 + 2] = gx2;
    a.grot[cloud] = gr;
  }
}

// ---------------------------------------------------------------------------------
// small "prep" kernels between the passes
// ---------------------------------------------------------------------------------
// out[t][i] (+)= alpha * sum_{s < S} part[(t*S + s)*n + i]      (fp64 accumulate, deterministic order)
// grid (ceil(n/32), towers), block 256 = 32 columns x 8 slice groups
template <typename T>
__global__ __launch_bounds__(1024) void reduce_slices_kernel(const T* __restrict__ part, int S, long n, float* __restrict__ out,
                                                             float alpha, int accumulate)
{
  // 32 columns x 32 slice groups per block: the reduction is a chain of dependent L2 round trips per thread, so the
  // slice loop is spread as wide as a block allows
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, t = blockIdx.y;
  const long i = blockIdx.x * 32L + cl;
  double s = 0.0;
  if (i < n)
    for (int k = g; k < S; k += 32 * 8) {   // eight loads in flight per round trip, summed in slice order
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int ku = k + 32 * u; v[u] = ku < S ? part[((size_t)t * S + ku) * n + i] : T(0); }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) tot += red[k][cl];
    const float v = (float)tot * alpha;
    out[(size_t)t * n + i] = accumulate ? out[(size_t)t * n + i] + v : v;
  }
}

// Several slice reductions in one launch (every launch costs >= 4.7 us on the one stream of a step, whatever its work):
// grid (max over jobs of ceil(n/32), 2 towers, jobs).  Same summation order as reduce_slices_kernel.
struct ReduceJob { const void* part; int is_double; int S; long n; float* out; float alpha; int towers;
                   int upper_c = 0;      // upper_c = C > 0: the columns are a C x C matrix of which only the 32 x 32 blocks on / above the block diagonal are summed (Gram partials)
                   double* out64 = nullptr; };   // optional: the same totals unrounded (the statistics derived from a Gram divide by variances that can be 1e-4 of its entries)
//   // upper_c = C > 0: the columns are a C x C matrix of which only the 32 x 32 blocks on / above the block diagonal are summed (Gram partials)
constexpr int kReduceJobs = 16;
struct ReduceJobs { ReduceJob j[kReduceJobs]; };
__device__ __forceinline__ void reduce_multi_body(const ReduceJobs& jobs, int bx, int t, int bz)
{
  __shared__ double red[32][33];
  const ReduceJob jb = jobs.j[bz];
  if (t >= jb.towers) return;
  if (!jb.is_double && jb.n >= 2048 && (jb.n & 3) == 0 && jb.S >= 32) {
    // wide fp32 jobs (the per-cloud Gram / U2 partials: 16 k columns x hundreds of slices, tens of MB): a wave reads one slice row
    // of 256 columns as 64 x 16 bytes (1 KB segments instead of the 128-byte ones of the narrow layout below), sixteen waves take
    // the slices round robin with eight loads in flight each; fp64 accumulation, fixed order.  Block bx <-> columns [256 bx, 256 bx + 256).
    __shared__ double wred[16][256];
    const long c0 = bx * 256L;
    if (c0 >= jb.n) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long col = c0 + lane * 4;
    const float* p = static_cast<const float*>(jb.part) + (size_t)t * jb.S * jb.n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    bool valid = col < jb.n;
    if (jb.upper_c > 0 && valid) { const int i = (int)(col / jb.upper_c), j = (int)(col - (long)i * jb.upper_c); valid = (i >> 5) <= (j >> 5); }
    if (valid)
      for (int k = wv; k < jb.S; k += 16 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ku = k + 16 * u;
          v[u] = ku < jb.S ? *reinterpret_cast<const f32x4*>(p + (size_t)ku * jb.n + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += (double)v[u][0]; a1 += (double)v[u][1]; a2 += (double)v[u][2]; a3 += (double)v[u][3]; }
      }
    wred[wv][lane * 4 + 0] = a0; wred[wv][lane * 4 + 1] = a1; wred[wv][lane * 4 + 2] = a2; wred[wv][lane * 4 + 3] = a3;
    __syncthreads();
    bool wr = threadIdx.x < 256 && c0 + threadIdx.x < jb.n;
    if (jb.upper_c > 0 && wr) { const long cc = c0 + threadIdx.x; const int i = (int)(cc / jb.upper_c), j = (int)(cc - (long)i * jb.upper_c); wr = (i >> 5) <= (j >> 5); }
    if (wr) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += wred[k][threadIdx.x];
      jb.out[(size_t)t * jb.n + c0 + threadIdx.x] = (float)tot * jb.alpha;
      if (jb.out64) jb.out64[(size_t)t * jb.n + c0 + threadIdx.x] = tot * (double)jb.alpha;
    }
    return;
  }
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long i = bx * 32L + cl;
  if (bx * 32L >= jb.n) return;
  double s = 0.0;
  if (i < jb.n) {
    // eight independent loads per round trip (a job of thousands of slices x a few dozen columns -- the column sums of h1: 2048 x 64 --
    // runs on two workgroups per tower and was a 64-deep chain of L2 round trips per thread: 21 us); same summation order as before
    if (jb.is_double) {
      const double* p = static_cast<const double*>(jb.part) + (size_t)t * jb.S * jb.n + i;
      for (int k = g; k < jb.S; k += 32 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int ku = k + 32 * u; v[u] = ku < jb.S ? p[(size_t)ku * jb.n] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
    } else {
      const float* p = static_cast<const float*>(jb.part) + (size_t)t * jb.S * jb.n + i;
      for (int k = g; k < jb.S; k += 32 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int ku = k + 32 * u; v[u] = ku < jb.S ? p[(size_t)ku * jb.n] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
      }
    }
  }
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && i < jb.n) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) tot += red[k][cl];
    jb.out[(size_t)t * jb.n + i] = (float)tot * jb.alpha;
    if (jb.out64) jb.out64[(size_t)t * jb.n + i] = tot * (double)jb.alpha;
  }
}
// [n] doubles -> floats (sync_bn: the all-reduced fp64 Gram / column sums back into the float copies the backward reads)
__global__ void cvt_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, size_t n)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(1024) void reduce_multi_kernel(const ReduceJobs jobs) { reduce_multi_body(jobs, blockIdx.x, blockIdx.y, blockIdx.z); }

// A layer's statistics finish and up to three slice reductions that do not depend on it, in one launch: grid (max of the two
// x extents, 2 towers, njobs + 1), block 1024; z = njobs is the statistics finish.
__global__ __launch_bounds__(1024) void stat_finish_reduce_kernel(const StatFinishArgs f, const ReduceJobs jobs, int njobs)
{
  if ((int)blockIdx.z < njobs) { reduce_multi_body(jobs, blockIdx.x, blockIdx.y, blockIdx.z); return; }
  const int gx = (f.C + kSfC - 1) / kSfC;
  if ((int)blockIdx.x < gx) stat_finish_body(f, blockIdx.x, blockIdx.y, gx);
}

// centred Gram: G[t][i][j] -= s[t][i]*s[t][j]/M ; m[t][i] = s[t][i]/M
__device__ __forceinline__ void centre_gram_elem(float* __restrict__ G, const float* __restrict__ s, int C, double M, float* __restrict__ m, long e, int t);
__device__ __forceinline__ void centre_gram_body(float* __restrict__ G, const float* __restrict__ s, int C, double M, float* __restrict__ m, unsigned bx, int t)
{
  centre_gram_elem(G, s, C, M, m, bx * 256L + threadIdx.x, t);
}
__device__ __forceinline__ void centre_gram_elem(float* __restrict__ G, const float* __restrict__ s, int C, double M, float* __restrict__ m, long e, int t)
{
  // Only the 32 x 32 blocks on or above the block diagonal are read (the bf16 forward accumulates just those); each of
  // their elements is centred and mirrored into the block below the diagonal by the same thread.
  if (e >= (long)C * C) return;
  const int i = e / C, j = e % C;
  if (j == 0) m[t * C + i] = (float)((double)s[t * C + i] / M);
  if ((i >> 5) > (j >> 5)) return;
  float* Gt = G + (size_t)t * C * C;
  const float v = (float)((double)Gt[e] - (double)s[t * C + i] * (double)s[t * C + j] / M);
  Gt[e] = v;
  if ((i >> 5) < (j >> 5)) Gt[(size_t)j * C + i] = v;
}
__global__ __launch_bounds__(256) void centre_gram_kernel(float* __restrict__ G, const float* __restrict__ s, int C, double M, float* __restrict__ m)
{
  centre_gram_body(G, s, C, M, m, blockIdx.x, blockIdx.y);
}
// the forward's last two small launches of a stage in one (independent of each other): blocks [0, 2 nG) centre the Gram of h2,
// the rest finish the pooled features; block 256
__global__ __launch_bounds__(256) void gram_pool_finish_kernel(float* __restrict__ G, const float* __restrict__ s, int C, double M, float* __restrict__ m,
                                                               const PoolFinishArgs pa)
{
  const unsigned nG = (unsigned)((C * C + 255) / 256);
  if (blockIdx.x < 2 * nG) centre_gram_body(G, s, C, M, m, blockIdx.x % nG, blockIdx.x / nG);
  else pool_finish_body(pa, blockIdx.x - 2 * nG);
}

// ---------------------------------------------------------------------------------
// Last conv layer of a stage, after phase 3: batch statistics of z3 = h2 W3 + b3 WITHOUT a pass over z3.  z3 is linear in h2, so
// with the tower's column sums s = sum_n h2[n,:] and Gram G = sum_n h2^T h2 (both reduced over the clouds in fp64 by the launch
// in front of this one) and the centred Gram Ghat = G - s s^T / M:
//     mean_c = (s . w_c) / M + b_c,      var_c = w_c^T Ghat w_c / M        (biased, tf.nn.moments; no E[z^2] - E[z]^2 cancellation)
// -- 128 x 128 multiply-adds per channel in fp64 instead of four VALU operations per element of the [B N, C3] tensor in phase 3's
// epilogue and a [2B][2][C3][2] double slice per cloud.  train_matmul_bf16: h2 and G are those of the rounded h2, W3 is rounded
// here (round_w), so the statistics are those of the bf16 product exactly.
// The same launch finishes the stage's forward: EMA shadows (utils/tf_util.py:476-485), scale / shift / rstd / k for the backward,
// the pooled features relu(bn(extreme)) with the arg-extreme row (pool_finish_body's arithmetic), and the centred Gram + column
// means m2 the layer-3 identities of the backward read (kernels_train_bwd.h header).
// grid (ceil(C3 / 8), 2 towers), block 1024 = 8 channels x 128 row groups; dynamic LDS: Ghat [C2][C2] floats, W3 columns
// [C2][8] doubles, s [C2] doubles.
// ---------------------------------------------------------------------------------
struct Stat3Args {
  const double* G;       // [2][C2*C2] reduced Gram, 32 x 32 blocks on / above the block diagonal valid -- fp64 as the reduction left it: rounded to
  const double* s;       // [2][C2] reduced column sums        float (6e-8 of entries ~1e3 x a small channel variance) the statistics lost three digits
  const float* W;        // [C2][C3]
  int C2, C3; double M; int round_w;
  const float* beta[2]; const float* gamma[2]; float* mov_mean[2]; float* mov_var[2];
  float bn_decay; int update_ema;
  float *mean, *var, *scale, *shift, *rstd, *k;   // [2][C3]
  float* Gc; float* m2;  // out: centred Gram [2][C2*C2] (all blocks) and column means [2][C2]
  PoolFinishArgs pa;     // ext / idx2 / sgn / bias / pooled / zhat_star / idx (scale, shift, mean, var: the arrays above)
};

// Round 4: the quadratic forms on the fp64 matrix pipe.  Rounds 2 - 3 gave a workgroup 8 channels: every one of the 256 workgroups staged
// and centred the whole 64 KB Gram in LDS to evaluate eight 128 x 128 forms on the VALU (28.6 us per launch, three launches per step).
// Now a workgroup owns 16 channels; wave w forms T[16 rows i of tile w][16 channels] = Ghat[i, :] W[:, c] with C2 / 4
// `v_mfma_f64_16x16x4_f64` -- A = the centred Gram row (centred in fp64 on the fly from the reduced upper blocks: no LDS staging, no
// barrier), B = the W3 columns -- and q_c = sum_i W[i, c] T[i, c] is finished by two shuffles and an 8-way sum over the waves.
// C/D layout of the f64 form (cdna_hip_programming.md 3): col = lane & 15, row = (lane >> 4) + 4 reg.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kS3C = 16;   // channels per workgroup; block = 8 waves (one 16-row tile of the Gram each: C2 <= 128)
__global__ __launch_bounds__(512) void stat3_pool_finish_kernel(const Stat3Args a)
{
  __shared__ double red[8][2][kS3C];   // [wave][q | s.w][channel]
  __shared__ float cst[4][kS3C];       // mean, var, scale, shift of the block's channels
  const int C2 = a.C2, C3 = a.C3, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * kS3C, cj = lane & 15, kq = lane >> 4, c = c0 + cj;
  const double* G = a.G + (size_t)t * C2 * C2;
  const double* sv = a.s + (size_t)t * C2;
  const double invM = 1.0 / a.M;
  auto ghat = [&](int i, int j) -> double {   // centred Gram from the reduced upper 32 x 32 blocks
    const double raw = (i >> 5) <= (j >> 5) ? G[(size_t)i * C2 + j] : G[(size_t)j * C2 + i];
    return raw - sv[i] * (sv[j] * invM);
  };
  // the pooled-feature part below does not depend on the statistics until its last step: its first eight clouds per thread are requested
  // now and travel under the Gram loads and the MFMAs
  const PoolFinishArgs& p = a.pa;
  const int g = tid >> 4;
  constexpr int kPre = 8;
  float pe0[kPre], pe1[kPre]; int pi0[kPre], pi1[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int b = min(g + 32 * u, p.B - 1);
    const size_t h0 = ((size_t)(t * p.B + b) * 2) * C3 + min(c, C3 - 1), h1 = h0 + C3;
    pe0[u] = p.ext[h0]; pi0[u] = p.idx2[h0]; pe1[u] = p.ext[h1]; pi1[u] = p.idx2[h1];
  }
  // ... and so are the few values the later phases start from (each was a round trip of its own behind a barrier): the block's share of the
  // centred Gram that goes back to HBM, the finishing threads' parameters, the pooled part's sign and bias
  const int gper = (C2 + (int)gridDim.x - 1) / (int)gridDim.x, gr0 = blockIdx.x * gper, gr1 = min(C2, gr0 + gper), gne = (gr1 - gr0) * C2;
  const int ge_i = gr0 + min(tid, max(gne - 1, 0)) / C2, ge_j = min(tid, max(gne - 1, 0)) % C2;
  const double ge_raw = (ge_i >> 5) <= (ge_j >> 5) ? G[(size_t)ge_i * C2 + ge_j] : G[(size_t)ge_j * C2 + ge_i];
  const double ge_si = sv[ge_i], ge_sj = sv[ge_j];
  const int fcc = min(c0 + (tid & (kS3C - 1)), C3 - 1);
  const float f_bias = a.pa.bias[fcc], f_gamma = a.gamma[t][fcc], f_beta = a.beta[t][fcc];
  const float f_mm = a.update_ema ? a.mov_mean[t][fcc] : 0.f, f_mv = a.update_ema ? a.mov_var[t][fcc] : 0.f;
  const float p_sg = p.sgn[t * C3 + min(c, C3 - 1)], p_bias = p.bias[min(c, C3 - 1)];
  double qp = 0.0, swp = 0.0;
  if (wave * 16 < C2) {
    const int i = wave * 16 + cj;          // A row of this lane
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    // every operand of the wave's C2 / 4 MFMAs is requested before the first one issues (C2 <= 128: 32 Gram + 32 weight + 32 column-sum values per lane)
    double gr[32], sr[32];
    float wr[32];
    const double si = sv[i];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int k = min(4 * m + kq, C2 - 1);
      gr[m] = (i >> 5) <= (k >> 5) ? G[(size_t)i * C2 + k] : G[(size_t)k * C2 + i];
      wr[m] = c < C3 ? a.W[(size_t)k * C3 + c] : 0.f;
      sr[m] = sv[k];
    }
    float wd[4]; double sd[4];   // the D rows' weights and column sums (needed right behind the MFMAs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ir = wave * 16 + kq + 4 * r;
      wd[r] = c < C3 ? a.W[(size_t)ir * C3 + c] : 0.f;
      sd[r] = sv[ir];
    }
#pragma unroll
    for (int m = 0; m < 32; ++m)
      if (4 * m < C2) {
        float w = wr[m];
        if (a.round_w) w = __uint_as_float((unsigned)to_bf16_bits(w) << 16);
        const double av = gr[m] - si * (sr[m] * invM);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (double)w, acc, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // D row of register r: wave * 16 + kq + 4 r
      float wf = wd[r];
      if (a.round_w) wf = __uint_as_float((unsigned)to_bf16_bits(wf) << 16);
      const double w = (double)wf;
      qp += w * acc[r];
      swp += sd[r] * w;
    }
  }
  qp += __shfl_xor(qp, 16); qp += __shfl_xor(qp, 32);
  swp += __shfl_xor(swp, 16); swp += __shfl_xor(swp, 32);
  if (lane < kS3C) { red[wave][0][lane] = qp; red[wave][1][lane] = swp; }
  // rows [r0, r1) of the centred Gram (all blocks) and the column means go to HBM for the backward
  {
    if (tid < gne) a.Gc[(size_t)t * C2 * C2 + (size_t)ge_i * C2 + ge_j] = (float)(ge_raw - ge_si * (ge_sj * invM));   // (= ghat(i, j))
    for (int e = tid + 512; e < gne; e += 512) {
      const int i = gr0 + e / C2, j = e % C2;
      a.Gc[(size_t)t * C2 * C2 + (size_t)i * C2 + j] = (float)ghat(i, j);
    }
    if (blockIdx.x == 0) for (int i = tid; i < C2; i += 512) a.m2[t * C2 + i] = (float)(sv[i] * invM);
  }
  __syncthreads();
  if (tid < kS3C && c0 + tid < C3) {
    const int cc = c0 + tid;
    double Q = 0.0, sw = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { Q += red[w][0][tid]; sw += red[w][1][tid]; }
    const float bias = f_bias;   // (tid < kS3C: fcc == cc)
    const float mf = (float)(sw * invM + (double)bias), vf = (float)fmax(Q * invM, 0.0);
    const float rs = 1.0f / sqrtf(vf + kBnEps), inv = f_gamma * rs;
    a.mean[t * C3 + cc] = mf; a.var[t * C3 + cc] = vf;
    a.scale[t * C3 + cc] = inv; a.shift[t * C3 + cc] = (bias - mf) * inv + f_beta;
    a.rstd[t * C3 + cc] = rs; a.k[t * C3 + cc] = inv;
    if (a.update_ema) {
      a.mov_mean[t][cc] = f_mm - (1.f - a.bn_decay) * (f_mm - mf);
      a.mov_var[t][cc] = f_mv - (1.f - a.bn_decay) * (f_mv - vf);
    }
    cst[0][tid] = mf; cst[1][tid] = vf; cst[2][tid] = inv; cst[3][tid] = (bias - mf) * inv + f_beta;
  }
  __syncthreads();
  if (c >= C3) return;
  // pooled features of the block's channels, all clouds of the tower (ext = extreme of sgn * (z - bias), both half-wave slices):
  // 16 channels x 32 cloud groups; the first eight clouds per thread were requested at the top
  const float sg = p_sg, bias = p_bias, mf = cst[0][cj], rs = 1.0f / sqrtf(cst[1][cj] + kBnEps), sc = cst[2][cj], sh = cst[3][cj];
  auto finish = [&](int b, float e, int bi, float e1, int b1) {
    if (e1 > e || (e1 == e && b1 < bi)) { e = e1; bi = b1; }
    const size_t i = (size_t)(t * p.B + b) * C3 + c;
    p.idx[i] = bi;
    p.pooled[t * p.tower_stride + b * p.row_stride + c] = fmaxf(fmaf(e * sg, sc, sh), 0.f);
    p.zhat_star[i] = (e * sg + bias - mf) * rs;
  };
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int b = g + 32 * u;
    if (b < p.B) finish(b, pe0[u], pi0[u], pe1[u], pi1[u]);
  }
  for (int b0 = g + 32 * kPre; b0 < p.B; b0 += 32 * 4) {
    float e0[4], e1[4]; int i0[4], i1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = min(b0 + 32 * u, p.B - 1);
      const size_t h0 = ((size_t)(t * p.B + b) * 2) * C3 + c, h1 = h0 + C3;
      e0[u] = p.ext[h0]; i0[u] = p.idx2[h0]; e1[u] = p.ext[h1]; i1[u] = p.idx2[h1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 32 * u;
      if (b < p.B) finish(b, e0[u], i0[u], e1[u], i1[u]);
    }
  }
}
inline size_t stat3_lds_bytes(int) { return 0; }   // (static LDS only since round 4)

// last layer: per (tower, channel): dbeta3 = sum_b g0, dgamma3 = sum_b g0 zhat*, E, k*dbeta, gs = k*g0
struct Prep3Args {
  const float* dP; long tower_stride, row_stride;   // dL/dpooled in the pooled layout
  const float* pooled;                              // same layout
  const float* zhat_star;                           // [2B][C]
  const float* gamma[2]; const float* var;          // [2][C]
  int B, C; double M;
  float* dbeta[2]; float* dgamma[2];
  float* E; float* kdb; float* gs;                  // [2][C], [2][C], [2B][C]
  double* totals = nullptr; int mode = 0;           // sync_bn: 1 = local (dbeta, dgamma) -> gradients + totals [2][C][2] and gs; 2 = E / kdb from the all-reduced totals
  // optional (bf16 mode): u[t][c] = (m2[t] . W3[:, c]) E[t][c] + kdb[t][c] / M, from which pass B2's bias row follows without Q3:
  // q3b[j] = -sum_i m2[i] Q3[i][j] - sum_c W3[j][c] kdb[c] / M = -sum_c W3[j][c] u[c]
  float* u = nullptr; const float* m2 = nullptr; const float* W = nullptr; int C2 = 0;
};

__global__ __launch_bounds__(1024) void prep3_kernel(const Prep3Args a)   // grid (ceil(C/32), 2), block 32 channels x 32 cloud groups (the cloud loop is a chain of dependent loads)
{
  __shared__ double red[32][32][2];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, c = blockIdx.x * 32 + cl, t = blockIdx.y;
  float rs = 0.f, k = 0.f;
  double sb = 0.0, sg = 0.0;
  if (c < a.C) { rs = 1.0f / sqrtf(a.var[t * a.C + c] + kBnEps); k = a.gamma[t][c] * rs; }
  if (c < a.C && a.mode != 2) {
    for (int b0 = g; b0 < a.B; b0 += 32 * 8) {   // eight clouds' three loads in flight per thread (B = 256: one round trip instead of eight), summed in cloud order
      float pv[8], dv[8], zv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = min(b0 + 32 * u, a.B - 1);
        const size_t pi = t * a.tower_stride + b * a.row_stride + c;
        pv[u] = a.pooled[pi]; dv[u] = a.dP[pi]; zv[u] = a.zhat_star[(size_t)(t * a.B + b) * a.C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 32 * u;
        if (b < a.B) {
          const float g0 = pv[u] > 0.f ? dv[u] : 0.f;
          sb += g0; sg += (double)g0 * zv[u];
          a.gs[(size_t)(t * a.B + b) * a.C + c] = k * g0;
        }
      }
    }
  }
  red[g][cl][0] = sb; red[g][cl][1] = sg;
  double vp = 0.0;   // this row group's share of m2 . W3[:, c]
  if (a.u && a.mode != 1 && c < a.C)
    for (int i = g; i < a.C2; i += 32) vp += (double)a.m2[t * a.C2 + i] * (double)a.W[(size_t)i * a.C + c];
  __syncthreads();
  if (g == 0 && c < a.C) {
    sb = 0.0; sg = 0.0;
    for (int q = 0; q < 32; ++q) { sb += red[q][cl][0]; sg += red[q][cl][1]; }
  }
  const bool with_u = a.u && a.mode != 1;
  if (with_u) {
    __syncthreads();
    red[g][cl][0] = vp;
    __syncthreads();
  }
  if (g != 0 || c >= a.C) return;
  if (a.mode == 2) { sb = a.totals[((size_t)t * a.C + c) * 2]; sg = a.totals[((size_t)t * a.C + c) * 2 + 1]; }
  else { a.dbeta[t][c] = (float)sb; a.dgamma[t][c] = (float)sg; }
  if (a.mode == 1) { a.totals[((size_t)t * a.C + c) * 2] = sb; a.totals[((size_t)t * a.C + c) * 2 + 1] = sg; return; }
  const float Ef = (float)(-(double)k * rs * sg / a.M), kdbf = (float)((double)k * sb);
  a.E[t * a.C + c] = Ef;
  a.kdb[t * a.C + c] = kdbf;
  if (with_u) {
    double v = 0.0;
    for (int q = 0; q < 32; ++q) v += red[q][cl][0];
    a.u[t * a.C + c] = (float)(v * (double)Ef + (double)kdbf / a.M);
  }
}

// hidden layer (after its pass): E = -k r dgamma/M, kdb = k*dbeta, k, rstd from totals
__global__ void prep_hidden_kernel(const float* __restrict__ dbg /*[2][C][2]*/, const float* __restrict__ var,
                                   const float* g0p, const float* g1p, int C, double M, float* __restrict__ dbeta0,
                                   float* __restrict__ dbeta1, float* __restrict__ dgamma0, float* __restrict__ dgamma1,
                                   float* __restrict__ E, float* __restrict__ kdb, float* __restrict__ kk, float* __restrict__ rstd)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  if (c >= C) return;
  const float gam = t ? g1p[c] : g0p[c];
  const float rs = 1.0f / sqrtf(var[t * C + c] + kBnEps), k = gam * rs;
  const float sb = dbg[(t * C + c) * 2], sg = dbg[(t * C + c) * 2 + 1];
  (t ? dbeta1 : dbeta0)[c] = sb;
  (t ? dgamma1 : dgamma0)[c] = sg;
  if (E) E[t * C + c] = (float)(-(double)k * rs * sg / M);
  if (kdb) kdb[t * C + c] = k * sb;
  if (kk) kk[t * C + c] = k;
  if (rstd) rstd[t * C + c] = rs;
}

// hidden layer, straight from pass B2's per-cloud partials: the slice reduction of (dbeta2, dgamma2) and prep_hidden_kernel's
// arithmetic in one launch, next to up to two independent slice reductions (s1, m1):
// grid (max(ceil(C / 8), x extent of the jobs), 2, 1 + njobs), block 1024; z = 0 is the hidden-layer part.
struct PrepHiddenArgs {
  const double* part; int S;   // [2][S][C][2]
  const float* var; const float* gamma[2]; int C; double M;
  float* dbeta[2]; float* dgamma[2];
  float *E, *kdb, *kk, *rstd;   // [2][C]
  double* totals = nullptr; int mode = 0;   // sync_bn: as Prep3Args
};
constexpr int kPhC = 8;   // channels per workgroup: 128 slice groups x 8 channels (32 channels x 32 groups left the launch on 8 workgroups walking 16-deep load chains: 21 us)
__global__ __launch_bounds__(1024) void prep_hidden_reduce_kernel(const PrepHiddenArgs a, const ReduceJobs jobs)
{
  if (blockIdx.z > 0) { reduce_multi_body(jobs, blockIdx.x, blockIdx.y, blockIdx.z - 1); return; }
  __shared__ double red[1024 / kPhC][kPhC][2];
  const int cl = threadIdx.x % kPhC, g = threadIdx.x / kPhC, c = blockIdx.x * kPhC + cl, t = blockIdx.y;
  constexpr int kG = 1024 / kPhC;
  if ((int)blockIdx.x * kPhC >= a.C) return;
  double sb = 0.0, sg = 0.0;
  if (c < a.C && a.mode != 2)
    for (int k = g; k < a.S; k += kG * 4) {
      double v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ku = min(k + u * kG, a.S - 1);
        const double* p = a.part + (((size_t)t * a.S + ku) * a.C + c) * 2;
        v0[u] = p[0]; v1[u] = p[1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (k + u * kG < a.S) { sb += v0[u]; sg += v1[u]; }
    }
  red[g][cl][0] = sb; red[g][cl][1] = sg;
  __syncthreads();
  if (g != 0 || c >= a.C) return;
  sb = 0.0; sg = 0.0;
  for (int q = 0; q < kG; ++q) { sb += red[q][cl][0]; sg += red[q][cl][1]; }
  if (a.mode == 2) { sb = a.totals[((size_t)t * a.C + c) * 2]; sg = a.totals[((size_t)t * a.C + c) * 2 + 1]; }
  const float sbf = (float)sb, sgf = (float)sg;   // (as the two-launch form: totals rounded to fp32 first)
  const float rs = 1.0f / sqrtf(a.var[t * a.C + c] + kBnEps), k = a.gamma[t][c] * rs;
  if (a.mode != 2) { a.dbeta[t][c] = sbf; a.dgamma[t][c] = sgf; }
  if (a.mode == 1) { a.totals[((size_t)t * a.C + c) * 2] = sb; a.totals[((size_t)t * a.C + c) * 2 + 1] = sg; return; }
  if (a.E) a.E[t * a.C + c] = (float)(-(double)k * rs * sgf / a.M);
  if (a.kdb) a.kdb[t * a.C + c] = k * sbf;
  if (a.kk) a.kk[t * a.C + c] = k;
  if (a.rstd) a.rstd[t * a.C + c] = rs;
}

// Sp[t][k][c] = sum_b gs[b,c] * h2[(cloud, idx[b,c]), k]      grid (C3, 2), block (C2 <= 128) x 4 cloud groups
// h2_bf16: the forward stored h2 as bf16 (train_matmul_bf16).  The (index, weight) pairs of the channel are staged in
// LDS first so that the row gathers are independent loads (4 in flight per thread).
// One workgroup = kSdC consecutive channels of one tower.  Thread (slot = tid / Q, piece = tid % Q) gathers 16-byte pieces (Q = the
// row's 16-byte pieces: C2 / 8 in bf16, C2 / 4 in fp32): a wave-level load fetches 1 KB of rows instead of 128 B, four hits of a
// channel in flight per thread; the slots' partial sums meet in LDS, channel after channel.  (One element per thread -- 2-byte
// loads, 256 of them per thread -- ran at 0.9 TB/s.)  The (index, weight) pairs of the kSdC channels are staged per 128 clouds.
constexpr int kSdC = 8, kSdStage = 128;
__device__ __forceinline__ void sparse_dw_body(const float* __restrict__ gs, const int* __restrict__ idx, const float* __restrict__ h2,
                                               int B, int N, int C2, int C3, float* __restrict__ Sp, int h2_bf16, int cb, int t, unsigned char* lds)
{
  int (*sidx)[kSdC] = reinterpret_cast<int (*)[kSdC]>(lds);                                       // [kSdStage][kSdC]
  float (*sgv)[kSdC] = reinterpret_cast<float (*)[kSdC]>(lds + kSdStage * kSdC * 4);               // [kSdStage][kSdC]
  float (*part)[132] = reinterpret_cast<float (*)[132]>(lds + 2 * kSdStage * kSdC * 4);            // [slot][column]: up to 64 slots x 128 columns (+4: bank spread)
  const int tid = threadIdx.x, nt = blockDim.x, c0 = cb * kSdC;
  const int per = h2_bf16 ? 8 : 4, Q = C2 / per;                  // elements per 16-byte piece, pieces per row
  const int nslot = min(64, nt / Q), slot = tid / Q, piece = tid % Q;
  const bool worker = slot < nslot;
  double tot[kSdC];                                                // (threads tid < C2: the channel sums of column tid)
#pragma unroll
  for (int cc = 0; cc < kSdC; ++cc) tot[cc] = 0.0;
  for (int b0 = 0; b0 < B; b0 += kSdStage) {
    const int nb = min(kSdStage, B - b0);
    __syncthreads();
    for (int i = tid; i < nb * kSdC; i += nt) {
      const int bi = i / kSdC, cc = c0 + i % kSdC;
      const size_t cloud = (size_t)t * B + b0 + bi;
      sidx[bi][i % kSdC] = cc < C3 ? idx[cloud * C3 + cc] : 0;
      sgv[bi][i % kSdC] = cc < C3 ? gs[cloud * C3 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int cc = 0; cc < kSdC; ++cc) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (worker)
        for (int i = slot; i < nb; i += nslot * 4) {
          uint4 v[4]; float g[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int iu = i + u * nslot;
            g[u] = iu < nb ? sgv[min(iu, nb - 1)][cc] : 0.f;
            v[u] = uint4{0u, 0u, 0u, 0u};
            if (g[u] != 0.f) {
              const size_t row = ((size_t)t * B + b0 + iu) * N + sidx[iu][cc];
              v[u] = h2_bf16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(h2) + row * C2 + piece * 8)
                             : *reinterpret_cast<const uint4*>(h2 + row * C2 + piece * 4);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (h2_bf16) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[2 * e] = fmaf(g[u], __uint_as_float(w4[e] << 16), acc[2 * e]);
                acc[2 * e + 1] = fmaf(g[u], __uint_as_float(w4[e] & 0xffff0000u), acc[2 * e + 1]);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[u], __uint_as_float(w4[e]), acc[e]);
            }
          }
        }
      __syncthreads();   // the previous channel's readers of `part` are done
      if (worker) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < per) part[slot][piece * per + e] = acc[e];
      }
      __syncthreads();
      if (tid < C2) {
        double sum = 0.0;
        for (int q = 0; q < nslot; ++q) sum += (double)part[q][tid];
        tot[cc] += sum;
      }
    }
  }
  if (tid < C2) {
#pragma unroll
    for (int cc = 0; cc < kSdC; ++cc)
      if (c0 + cc < C3) Sp[((size_t)t * C2 + tid) * C3 + c0 + cc] = (float)tot[cc];
  }
}
// Round 4 form for row widths whose 16-byte pieces divide a wave (Q = 4, 8, 16, 32: every shipped width).  The loop above walks the
// workgroup's eight channels one after the other -- two row gathers in flight per thread, two barriers and a 64-deep LDS column sum per
// (channel, stage of 128 clouds): 32 barriers around 16 exposed HBM round trips, 115 us for the three stages' 235 MB of 256-byte rows
// (2 TB/s).  Here a thread requests the rows of FOUR channels x its clouds of the stage at once (8 - 16 independent 16-byte loads),
// keeps fp32 partial sums per channel in registers across the stages, and the slots meet once at the end: shuffles across the wave's
// slots, one LDS slab per wave, a 16-way fp64 sum per output element.  Same sums up to the order of the fp32 partials; deterministic.
template <bool BF16>
__device__ __forceinline__ void sparse_dw_wide_body(const float* __restrict__ gs, const int* __restrict__ idx, const float* __restrict__ h2,
                                                    int B, int N, int C2, int C3, float* __restrict__ Sp, int cb, int t, unsigned char* lds)
{
  constexpr int per = BF16 ? 8 : 4, kWavesSd = 16, kH = 4;   // kH channels per pass (register budget of a 1024-thread workgroup: 128 per lane)
  int (*sidx)[kH] = reinterpret_cast<int (*)[kH]>(lds);                                          // [kSdStage][kH]
  float (*sgv)[kH] = reinterpret_cast<float (*)[kH]>(lds + kSdStage * kH * 4);                    // [kSdStage][kH]
  float (*part)[kH][132] = reinterpret_cast<float (*)[kH][132]>(lds + 2 * kSdStage * kH * 4);     // [kWavesSd][kH][132]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c0 = cb * kSdC;
  const int Q = C2 / per, nslot = 1024 / Q, slot = tid / Q, piece = tid % Q;
#pragma unroll 1
  for (int hc = 0; hc < kSdC; hc += kH) {
    float acc[kH][per];
#pragma unroll
    for (int q = 0; q < kH; ++q)
#pragma unroll
      for (int e = 0; e < per; ++e) acc[q][e] = 0.f;
    for (int b0 = 0; b0 < B; b0 += kSdStage) {
      const int nb = min(kSdStage, B - b0);
      __syncthreads();
      for (int i = tid; i < nb * kH; i += 1024) {
        const int bi = i / kH, cc = c0 + hc + i % kH;
        const size_t cloud = (size_t)t * B + b0 + bi;
        sidx[bi][i % kH] = cc < C3 ? idx[cloud * C3 + cc] : 0;
        sgv[bi][i % kH] = cc < C3 ? gs[cloud * C3 + cc] : 0.f;
      }
      __syncthreads();
      for (int i0 = slot; i0 < nb; i0 += 2 * nslot) {   // two clouds x four channels per round: eight rows in flight per thread
        uint4 v[2][kH]; float g[2][kH];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < kH; ++q) {
            const int iu = i0 + u * nslot;
            g[u][q] = iu < nb ? sgv[min(iu, nb - 1)][q] : 0.f;
            v[u][q] = uint4{0u, 0u, 0u, 0u};
            if (g[u][q] != 0.f) {
              const size_t row = ((size_t)t * B + b0 + iu) * N + sidx[iu][q];
              v[u][q] = BF16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(h2) + row * C2 + piece * 8)
                             : *reinterpret_cast<const uint4*>(h2 + row * C2 + piece * 4);
            }
          }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < kH; ++q) {
            const unsigned w4[4] = {v[u][q].x, v[u][q].y, v[u][q].z, v[u][q].w};
            if constexpr (BF16) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[q][2 * e] = fmaf(g[u][q], __uint_as_float(w4[e] << 16), acc[q][2 * e]);
                acc[q][2 * e + 1] = fmaf(g[u][q], __uint_as_float(w4[e] & 0xffff0000u), acc[q][2 * e + 1]);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[q][e] = fmaf(g[u][q], __uint_as_float(w4[e]), acc[q][e]);
            }
          }
      }
    }
    // the wave's 64 / Q slots meet by shuffles (lanes piece, piece + Q, ...), then one slab per wave
#pragma unroll
    for (int q = 0; q < kH; ++q)
#pragma unroll
      for (int e = 0; e < per; ++e) {
        float x = acc[q][e];
        for (int o = Q; o < 64; o <<= 1) x += __shfl_xor(x, o);
        if (lane < Q) part[wave][q][piece * per + e] = x;
      }
    __syncthreads();
    for (int o = tid; o < kH * C2; o += 1024) {
      const int q = o / C2, col = o - q * C2;
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < kWavesSd; ++w) sum += (double)part[w][q][col];
      if (c0 + hc + q < C3) Sp[((size_t)t * C2 + col) * C3 + c0 + hc + q] = (float)sum;
    }
  }
}
__device__ __forceinline__ bool sparse_dw_wide_ok(int C2, int h2_bf16)
{
  const int Q = C2 / (h2_bf16 ? 8 : 4);
  return C2 <= 128 && Q >= 4 && Q <= 64 && (Q & (Q - 1)) == 0 && C2 % (h2_bf16 ? 8 : 4) == 0;
}
__device__ __forceinline__ void sparse_dw_any(const float* __restrict__ gs, const int* __restrict__ idx, const float* __restrict__ h2,
                                              int B, int N, int C2, int C3, float* __restrict__ Sp, int h2_bf16, int cb, int t)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSdStage * kSdC * 4 + 64 * 132 * 4];   // (>= 2 * 128 * 4 * 4 + 16 * 4 * 132 * 4 of the wide form)
  if (sparse_dw_wide_ok(C2, h2_bf16)) {   // (uniform per launch)
    if (h2_bf16) sparse_dw_wide_body<true>(gs, idx, h2, B, N, C2, C3, Sp, cb, t, lds);
    else sparse_dw_wide_body<false>(gs, idx, h2, B, N, C2, C3, Sp, cb, t, lds);
  } else sparse_dw_body(gs, idx, h2, B, N, C2, C3, Sp, h2_bf16, cb, t, lds);
}
__global__ __launch_bounds__(1024) void sparse_dw_kernel(const float* __restrict__ gs, const int* __restrict__ idx, const float* __restrict__ h2,
                                                         int B, int N, int C2, int C3, float* __restrict__ Sp, int h2_bf16)
{
  sparse_dw_any(gs, idx, h2, B, N, C2, C3, Sp, h2_bf16, blockIdx.x, blockIdx.y);
}
// the three stages' Sp in one launch (the weight gradients wait for nothing but the optimiser): grid (max ceil(C3 / kSdC), 2, jobs), block kSdC * C2
struct SparseDwJob { const float* gs; const int* idx; const float* h2; int B, N, C2, C3; float* Sp; int h2_bf16; };
struct SparseDwJobs { SparseDwJob j[3]; };
struct CentreJob { float* G; const float* s; int C; double M; float* m; };
struct CentreJobs { CentreJob j[3]; };
// (+ the centring of the reduced Grams as further z planes of the same grid: they depend on the reductions only, like the gathers, and were a
//  5 us launch of their own between them and the products)
__global__ __launch_bounds__(1024) void sparse_dw_jobs_kernel(const SparseDwJobs jobs, int nsp, const CentreJobs cen)
{
  if ((int)blockIdx.z >= nsp) {
    const CentreJob& c = cen.j[blockIdx.z - nsp];
    if (c.G) centre_gram_elem(c.G, c.s, c.C, c.M, c.m, blockIdx.x * 1024L + threadIdx.x, blockIdx.y);
    return;
  }
  const SparseDwJob& q = jobs.j[blockIdx.z];
  if (!q.gs || (int)blockIdx.x * kSdC >= q.C3) return;
  sparse_dw_any(q.gs, q.idx, q.h2, q.B, q.N, q.C2, q.C3, q.Sp, q.h2_bf16, blockIdx.x, blockIdx.y);
}

// two scaled copies of one matrix in one launch: out_x[t] = W diag(col_x[t]) (transposed if tr_x), towers_x of them; grid (ceil(R*C/256), 2)
__device__ __forceinline__ void scale_cols2_body(const float* __restrict__ W, int R, int C, const float* __restrict__ colA, float* __restrict__ outA, int trA,
                                                 int towersA, const float* __restrict__ colB, float* __restrict__ outB, int trB, int towersB, unsigned bx, int t)
{
  const long e = bx * 256L + threadIdx.x;
  if (e >= (long)R * C) return;
  const int i = e / C, j = e % C;
  const float wv = W[e];
  if (t < towersA) {
    const float v = wv * (colA ? colA[t * C + j] : 1.f);
    outA[(size_t)t * R * C + (trA ? (size_t)j * R + i : (size_t)e)] = v;
  }
  if (t < towersB) {
    const float v = wv * (colB ? colB[t * C + j] : 1.f);
    outB[(size_t)t * R * C + (trB ? (size_t)j * R + i : (size_t)e)] = v;
  }
}
__global__ __launch_bounds__(256) void scale_cols2_kernel(const float* __restrict__ W, int R, int C, const float* __restrict__ colA, float* __restrict__ outA, int trA,
                                                          int towersA, const float* __restrict__ colB, float* __restrict__ outB, int trB, int towersB)
{
  scale_cols2_body(W, R, C, colA, outA, trA, towersA, colB, outB, trB, towersB, blockIdx.x, blockIdx.y);
}

// dW[i][j] (+)= sum_t ( Sp[t][i][j]*spscale[t][j] - m[t][i]*kdb[t][j] + GW[t][i][j]*E[t][j] )
__device__ __forceinline__ void combine_dw_body(const float* __restrict__ Sp, const float* __restrict__ spscale, const float* __restrict__ m,
                                                const float* __restrict__ kdb, const float* __restrict__ GW, const float* __restrict__ E,
                                                int R, int C, float* __restrict__ dW, unsigned bx, float gscale = 1.f)
{
  const long e = bx * 256L + threadIdx.x;
  if (e >= (long)R * C) return;
  const int i = e / C, j = e % C;
  float s = 0.f;
  for (int t = 0; t < 2; ++t) {
    const size_t o = (size_t)t * R * C + e;
    // gscale (sync_bn: 1 / ranks): m, kdb, GW, E are then the GLOBAL batch's, the same on every rank, and the gradient all-reduce adds them ranks times
    s += Sp[o] * (spscale ? spscale[t * C + j] : 1.f) + gscale * (GW[o] * E[t * C + j] - m[t * R + i] * kdb[t * C + j]);
  }
  dW[e] = s;
}
__global__ __launch_bounds__(256) void combine_dw_kernel(const float* __restrict__ Sp, const float* __restrict__ spscale, const float* __restrict__ m,
                                                         const float* __restrict__ kdb, const float* __restrict__ GW, const float* __restrict__ E,
                                                         int R, int C, float* __restrict__ dW, float gscale)
{
  combine_dw_body(Sp, spscale, m, kdb, GW, E, R, C, dW, blockIdx.x, gscale);
}
// layer 3 of a stage: the weight gradient's combine and the two scaled copies of W3 the backward needs next, one launch
// (grid (ceil(R C / 256), 3): y < 2 the copies of tower y, y = 2 the combine)
__global__ __launch_bounds__(256) void combine_scale_kernel(const float* __restrict__ Sp, const float* __restrict__ m, const float* __restrict__ kdb,
                                                            const float* __restrict__ GW, const float* __restrict__ E, int R, int C, float* __restrict__ dW,
                                                            const float* __restrict__ W, float* __restrict__ WE, float* __restrict__ WT)
{
  if (blockIdx.y == 2) combine_dw_body(Sp, nullptr, m, kdb, GW, E, R, C, dW, blockIdx.x);
  else scale_cols2_body(W, R, C, E, WE, 0, 2, nullptr, WT, 1, 1, blockIdx.x, blockIdx.y);
}

// The elementwise tails of the deferred weight-gradient work, all layers of the step in one launch each:
//   centre jobs: G <- G - s s^T / M (+ mirror), m = s / M            grid (max ceil(C^2 / 256), 2 towers, jobs)
//   combine jobs: dW = sum_t (Sp spscale - m kdb^T + GW diag(E))      grid (max ceil(R C / 256), 1, jobs)
__global__ __launch_bounds__(256) void centre_gram_jobs_kernel(const CentreJobs jobs)
{
  const CentreJob& q = jobs.j[blockIdx.z];
  if (q.G) centre_gram_body(q.G, q.s, q.C, q.M, q.m, blockIdx.x, blockIdx.y);
}
struct CombineJob { const float* Sp; const float* spscale; const float* m; const float* kdb; const float* GW; const float* E; int R, C; float* dW; float gscale = 1.f; };
struct CombineJobs { CombineJob j[6]; };
__global__ __launch_bounds__(256) void combine_dw_jobs_kernel(const CombineJobs jobs)
{
  const CombineJob& q = jobs.j[blockIdx.z];
  if (q.Sp) combine_dw_body(q.Sp, q.spscale, q.m, q.kdb, q.GW, q.E, q.R, q.C, q.dW, blockIdx.x, q.gscale);
}

// qb[t][j] = -sum_i m[t][i] Q[t][i][j] - sum_c W[j][c] kdb[t][c] / M      (W: [R=Cin][C=Cout])
// grid (Cin, 2), block 256: the two sums are spread over the block and reduced (fp64)
struct QBiasArgs { const float* Q; const float* m; const float* W; const float* kdb; int Cin, Cout; double M; float* qb; };
__device__ __forceinline__ void qbias_body(const QBiasArgs& a, int j, int t)
{
  __shared__ double red[4];
  const float* __restrict__ Q = a.Q; const float* __restrict__ m = a.m; const float* __restrict__ W = a.W; const float* __restrict__ kdb = a.kdb;
  float* __restrict__ qb = a.qb;
  const int Cin = a.Cin, Cout = a.Cout, tid = threadIdx.x;
  const double M = a.M;
  double s = 0.0;
  for (int i = tid; i < Cin; i += 256) s -= (double)m[t * Cin + i] * Q[((size_t)t * Cin + i) * Cin + j];
  for (int c = tid; c < Cout; c += 256) s -= (double)W[(size_t)j * Cout + c] * kdb[t * Cout + c] / M;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) qb[t * Cin + j] = (float)(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void qbias_kernel(const float* __restrict__ Q, const float* __restrict__ m, const float* __restrict__ W,
                                                    const float* __restrict__ kdb, int Cin, int Cout, double M, float* __restrict__ qb)
{
  qbias_body(QBiasArgs{Q, m, W, kdb, Cin, Cout, M, qb}, blockIdx.x, blockIdx.y);
}

// the bf16 images of a pass's Q / V matrices and the pass's bias row q_b (both read the same fresh Q), one launch:
// grid (pack_gx + Cin, max(jobs, 2)), block 256: x < pack_gx packs job y, the rest is qbias_kernel's grid (Cin, 2)
__global__ __launch_bounds__(256) void pack_qbias_kernel(const PackBf16Jobs j, int njobs, unsigned pack_gx, const QBiasArgs qa)
{
  if (blockIdx.x < pack_gx) { if ((int)blockIdx.y < njobs) pack_bf16_jobs_body(j, blockIdx.x, blockIdx.y, pack_gx); }
  else if (blockIdx.y < 2) qbias_body(qa, blockIdx.x - pack_gx, blockIdx.y);
}

// Round 4: Q = W diag(E) W^T, its bf16 operand image, the pass's bias row and (pass B1) the images of V2, ONE launch -- was
// gemm_small, then pack_qbias_kernel (which re-read the fresh Q), six times per step.  Blocks x < ntile: the product's 32 x 32 tiles with
// the image written from the epilogue (GemmArgs.img);  x < ntile + R: bias row qb[t][j] = -sum_c W[j][c] u[t][c], with
// u[c] = (m . W[:, c]) E[c] + kdb[c] / M given (prep3_kernel) or formed here (small layers);  the rest: pack jobs (V2).  grid (ntile + R + pack_gx, 2).
struct QBias2Args { const float* W; int R, K; const float* u; const float* m; const float* E; const float* kdb; double M; float* qb; };
__device__ __forceinline__ void qbias2_body(const QBias2Args& a, int j, int t)
{
  __shared__ double red2[8];
  __shared__ float us[1024];
  const int tid = threadIdx.x;
  if (!a.u) {   // u of this tower, formed here for small layers (K <= 512 columns x an R-long dot each: the 64 -> 128 hidden layer); P row slices per column
    __shared__ double up[512];
    const int P = 512 / a.K, c = tid % a.K, pp = tid / a.K;
    double v = 0.0;
    if (pp < P)
      for (int i = pp; i < a.R; i += P) v += (double)a.m[t * a.R + i] * (double)a.W[(size_t)i * a.K + c];
    up[tid] = v;
    __syncthreads();
    if (tid < a.K) {
      double tot = 0.0;
      for (int q = 0; q < P; ++q) tot += up[q * a.K + tid];
      us[tid] = (float)(tot * (double)a.E[t * a.K + tid] + (double)a.kdb[t * a.K + tid] / a.M);
    }
    __syncthreads();
  }
  double s = 0.0;
  for (int c = tid; c < a.K; c += 512) s -= (double)a.W[(size_t)j * a.K + c] * (double)(a.u ? a.u[t * a.K + c] : us[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red2[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) { double tot = 0.0; for (int w = 0; w < 8; ++w) tot += red2[w]; a.qb[t * a.R + j] = (float)tot; }
}
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_qimg_kernel(const GemmArgs g, int tx, int ntile, const QBias2Args qa, const PackBf16Jobs pj, int npack,
                                                                    unsigned pack_gx)
{
  const int bx = blockIdx.x, t = blockIdx.y;
  __shared__ float smem[kGemmSmemFloats];
  if (bx < ntile) { gemm_small_tile(g, bx % tx, bx / tx, t, smem); return; }
  if (bx < ntile + qa.R) { qbias2_body(qa, bx - ntile, t); return; }
  if (threadIdx.x < 256)   // (the pack body strides by 256 threads per block)
    for (int q = t; q < npack; q += 2) pack_bf16_jobs_body(pj, bx - ntile - qa.R, q, pack_gx);
}

}  // namespace alignnet
