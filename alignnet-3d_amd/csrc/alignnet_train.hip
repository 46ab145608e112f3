// Training side of the C ABI (include/alignnet_hip.h): the train-mode sess.run of the reference
// (train.py:368) = forward with batch statistics + loss + backward + optimiser + EMA + step++.
#include "engine.h"
#include "kernels_train_fwd.h"
#include "kernels_train_fwd_wide.h"
#include "kernels_train_head.h"
#include "kernels_train_bwd.h"
#include "kernels_train_dgcnn.h"
#include "kernels_train_generic.h"
#include "comm_loopback.h"
#include "kernels_debug.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

using namespace alignnet;

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static int fail(const alignnet_handle* h, const std::string& m) { h->err = m; return 1; }

// sync_bn (alignnet_set_option): every batch sum behind a BatchNorm -- forward moments, backward (dbeta, dgamma) totals, the Gram /
// column-sum matrices the layer identities use -- is added over the data-parallel ranks between the kernel that forms this rank's
// sum and the kernel that uses it (RCCL all-reduce on the compute stream; defined with the RCCL section).
static int sync_world(const alignnet_handle* h) { return h->comm ? h->comm_world : h->sync_emulate_world; }
static bool sync_on(const alignnet_handle* h) { return h->sync_bn && sync_world(h) > 1; }
static int sync_sum(alignnet_handle* h, void* buf, size_t n, bool is_double);
// two float buffers that the workspace carve laid out back to back travel as ONE all-reduce (a step's ~30 per-layer sums are latency-bound)
static int sync_sum2(alignnet_handle* h, float* a, size_t na, float* b, size_t nb)
{
  if (b == a + na) return sync_sum(h, a, na + nb, false);
  if (a == b + nb) return sync_sum(h, b, na + nb, false);
  return sync_sum(h, a, na, false) || sync_sum(h, b, nb, false);
}
static int sync_gather(alignnet_handle* h, const void* src, void* dst, size_t n4);   // n4 four-byte elements per rank; dst = [world][n4]
static bool gloss_on(const alignnet_handle* h) { return h->global_loss && sync_world(h) > 1; }
constexpr size_t kSyncBufDoubles = 4 * 4096;

namespace alignnet {

struct HeadLayerWS { float *z, *y, *mean, *var, *dz, *dyb; };   // per hidden FC layer (with BN); dyb: d(y) from the next layer's dx GEMM

struct StageWS {                 // one backbone stage (T1, T2, embedding)
  float* xform;                  // [2B][12] input frame of this stage
  float *mean[3], *var[3], *scale[3], *shift[3];   // [2][C_l]
  float *rstd[3], *kk[3];                          // [2][C_l]: rsqrt(var+eps), gamma*rsqrt(var+eps)
  float* sgn3;                   // [2][C3]
  float* ext; int* idx2;          // [2B][2 halves][C3] per-half extremes
  float* extp; int* idxp;         // [<= 512 workgroups][2 halves][C3]: the same per WORKGROUP when phase 3 deals a cloud to several (p3_parts)
  int* idx; float* zhat_star;     // [2B][C3] final arg-extreme index, zhat at the extreme
  float* h2;                     // [2B*N][C2]
  float *gram2, *s2, *m2;        // [2][C2*C2] centred Gram of h2, [2][C2] column sums, [2][C2] column means
  float* gram2raw;               // [2][C2*C2] Gram as reduced over the clouds (upper blocks), before centring
  double *gram2raw64, *s264, *g1f64, *s1e64;   // gram2raw | s2 and g1f | s1e as the fp64 reductions left them (each pair back to back): what the statistics kernels read
  float* pooled; long tower_stride, row_stride;    // forward output (layout of the consumer)
  float* dP;                     // dL/dpooled, same layout
  float *gx, *grot;              // [2B][3], [2B]
  // DGCNN branch (kernels_train_dgcnn.h); h2 holds the pooled edge features p = max_k h2
  unsigned char* argk;           // [2B*N][C2] arg-max neighbour slot
  double* mom;                   // [2B][27] moments of the edge feature
  float* s1e;                    // [2][C1] column sums of h1 (DGCNN: over the B*N*k edge rows), kept by the forward
  float* g1f;                    // [2][C1*C1] Gram(h1) of the forward (upper blocks), PointNet: statistics of z2 and the layer-2 weight gradient
  // backward quantities that the stage's WEIGHT gradients read after the stage's backward has moved on (deferred work, see Deferred):
  // one set per stage, so that the three stages' jobs can run together at the end of the step
  float *gs, *E3, *kdb3, *Sp, *GW;                 // [2B][C3], [2][C3], [2][C3], [2][C2*C3], [2][C2*C3]
  float *u2_part, *g1_part, *p_part;               // [2B][C1*C2], [2B][C1*C1], [2B][6*C1]
  float *u2, *g1, *s1, *m1, *E2, *kdb2, *k2, *GW2; // [2][C1*C2], [2][C1*C1], [2][C1] x 2, [2][C2] x 3, [2][C1*C2]
};

// Work that only the optimiser waits for -- the weight gradients of the conv layers (sparse rows + Gram identities), of the head
// layers (dW = x^T dz, bias sums) and the first layer's dW from its per-cloud partials -- is not launched where the backward
// produces its inputs (about ten launches of 5 - 45 us per stage, on the critical path of the one stream) but collected and run
// as five launches after the last stage's backward: all reductions | all sparse gathers, then all Gram centrings, all GEMMs, all
// combines, each as one multi-job launch with three stages' worth of parallelism.
struct Deferred {
  bool on = false;
  std::vector<ReduceJob> red;
  std::vector<SparseDwJob> sp;
  std::vector<CentreJob> cen;
  std::vector<std::pair<GemmArgs, int>> gemm;   // (product, batch entries)
  std::vector<CombineJob> comb;
  std::vector<std::pair<float*, size_t>> sync_after_red;   // sync_bn: reduced matrices that are summed over the ranks before the centrings read them
  void clear() { red.clear(); sp.clear(); cen.clear(); gemm.clear(); comb.clear(); sync_after_red.clear(); }
};

struct TrainWS {
  int cap = 0;
  char* base = nullptr; size_t bytes = 0;
  float* d_pcs[2] = {nullptr, nullptr};
  const float* last_pcs[2] = {nullptr, nullptr};   // the point clouds (device) of the last training forward (alignnet_debug_train_relu_mask)
  const float* last_ang[2] = {nullptr, nullptr};   // its yaw labels pc1_angles / pc2_angles (alignnet_debug_train_decisions: ALIGNNET_DECISION_ANGLE_CLASS)
  float* labels[6];              // device copies of the label tensors
  float* dropout_u;              // host-supplied uniforms (device copy), [sum over heads]
  float* center_mean; float* s1c; float* s2c; float* theta; int* cls;
  StageWS st[3];
  HeadLayerWS hl[3][ALIGNNET_MAX_WIDTHS];
  float* o[3]; float* d_o[3];    // head outputs [2B][3], [2B][3+2nb], [B][3+2nb] and their gradients
  float *d_s1c, *d_s2c;          // [2B][3]
  float* loss_out;               // [17]
  float* loss_scratch;
  float* head_din[3];            // gradient wrt the head input (= dP of the stage)
  // transient backward buffers (shared by the stages)
  double* stat_part; float* gram_part; double* colsum_part;
  bool moments_done[3] = {false, false, false};   // stage s's moments / phase-1 partials were formed by the kernel that wrote its frame (MomentsTail)
  float *dy2, *dy1;
  int* nn;                       // DGCNN: [2B][N][20] neighbour indices
  double* pdy_part;              // DGCNN: [2B][4][7][C1]
  double *dbg2_part, *dbg1_part, *s1_part;
  float *dbg2, *dbg1;
  float *W3E, *W3T, *Q3, *q3b, *q3img;
  float* u3;                     // [2][C3]: prep3_kernel's u, from which pass B2's bias row follows without Q3 (bf16 mode)
  float *rstd2, *W2E, *V2, *Q2, *q2b, *v2img, *q2img;
  float *k1, *rstd1;
  float* outs[8];
  // optimiser
  float *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  bool grad_clean = false;   // the gradient vector is all zeros (freshly allocated, or consumed by the optimiser)
  PackJob* pack_table = nullptr; int n_pack = 0;
  PackJob* stage_pack = nullptr;   // [3 stages][6]: Q3 t0,t1 | Q2 t0,t1, V2 t0,t1 (images rebuilt inside the backward)
  unsigned short* q3imgh = nullptr;   // bf16 images of Q3, both towers (train_bf16)
  unsigned short* b1imgh = nullptr;   // bf16 images of V2 (128 -> 64) and Q2 (64 -> 64), both towers (train_bf16, shipped widths)
  unsigned short* wp2h[3] = {nullptr, nullptr, nullptr};   // bf16 images of the hidden layers (train_bf16, no sign folding)
  unsigned short* wp3h[3] = {nullptr, nullptr, nullptr};   // bf16 images of the three lift layers (train_bf16)
  unsigned short* w2th[3] = {nullptr, nullptr, nullptr};   // bf16 MFMA images of round(W2)^T (K = C2, C = C1) per stage: dense edge backward of the dgcnn branch (train_bf16)
  unsigned short* w3th[3] = {nullptr, nullptr, nullptr};   // bf16 round(W3)^T [C3][C2] per stage: rows gathered by pass B2's sparse part (train_bf16, shipped widths)
  // general-depth PointNet stages (kernels_train_generic.h): pre-BatchNorm activations of every layer stay in HBM
  struct GenStage { float* X0; float* Z[kMaxConv]; float *mean[kMaxConv], *rstd[kMaxConv], *scale[kMaxConv], *shift[kMaxConv]; int* idx;
                    float* P; int* argk; float *one, *zero; } gen[3];   // general dgcnn stages: pooled edge features [2B N][C], their arg-k slot, identity scale / shift
  float *gen_d[2] = {nullptr, nullptr};   // gradient ping-pong buffers [2B*N][widest layer]
  float *gen_part = nullptr, *gen_dwpart = nullptr, *gen_ppart = nullptr, *gen_cA = nullptr, *gen_cB = nullptr, *gen_wt = nullptr;
  int gen_tiles = 0, gen_slabs = 0;
  Deferred defer;
  // global_loss: gathered end points / labels of all ranks, the loss gradient wrt all of them, scratch (carved for gl_cap global pairs)
  float* gl_base = nullptr; int gl_cap = 0;
  float *gl_s1c, *gl_s2c, *gl_o2, *gl_o3, *gl_theta, *gl_lab[6], *gl_d_s1c, *gl_d_s2c, *gl_d_o2, *gl_d_o3, *gl_scratch; int* gl_cls;
  bool glue_folded = false;   // the last backbone backward ran the stage glue inside dg_b0_cloud (fwd_bwd_device skips the glue launches)
};

}  // namespace alignnet

constexpr int kLossGroups = 64;   // workgroups sharing the B x B part of the loss
static size_t loss_scratch_floats(int B)
{
  return 26 * (size_t)B + 64 + (size_t)kLossGroups * 12 + (size_t)kLossGroups * 6 * B + 64 + ((size_t)(3 * B + 63) / 64) * 12 + 16;
}

static TrainWS* tws(alignnet_handle* h)
{
  if (!h->train_ws) h->train_ws = new TrainWS();
  return static_cast<TrainWS*>(h->train_ws);
}

extern "C" void alignnet_train_ws_free(alignnet_handle* h)
{
  if (!h->train_ws) return;
  TrainWS* w = static_cast<TrainWS*>(h->train_ws);
  if (w->base) hipFree(w->base);
  if (w->gl_base) hipFree(w->gl_base);
  for (int t = 0; t < 2; ++t) if (w->d_pcs[t]) hipFree(w->d_pcs[t]);
  if (w->grad) hipFree(w->grad);
  if (h->side_stream) { hipStreamSynchronize(h->side_stream); hipStreamDestroy(h->side_stream); h->side_stream = nullptr; }
  for (auto& e : h->side_ev) if (e) { hipEventDestroy(e); e = nullptr; }
  if (w->adam_m) hipFree(w->adam_m);
  if (w->adam_v) hipFree(w->adam_v);
  if (w->pack_table) hipFree(w->pack_table);
  if (w->stage_pack) hipFree(w->stage_pack);
  if (w->q3imgh) hipFree(w->q3imgh);
  if (w->b1imgh) hipFree(w->b1imgh);
  for (int s = 0; s < 3; ++s) { if (w->wp3h[s]) hipFree(w->wp3h[s]); if (w->wp2h[s]) hipFree(w->wp2h[s]); if (w->w3th[s]) hipFree(w->w3th[s]); if (w->w2th[s]) hipFree(w->w2th[s]); }
  delete w;
  h->train_ws = nullptr;
}

// dgcnn edge kernels (dg_train_fwd, dg_train_bwd_edge*): workgroups per cloud.  One workgroup per cloud fills the chip from 2B = 256 clouds (the backward:
// one eight-wave workgroup per CU) / 512 (the forward: two four-wave workgroups per CU); below that a cloud's tiles are dealt to several workgroups
// whose partial sums meet in the reductions that follow anyway (B = 64, N = 4096: the forward ran on a quarter of the CUs, the backward on half).
constexpr int kDgMaxParts = 8, kDgPartSlices = 512;   // at most 2B * parts <= 512 workgroups when parts > 1: the per-workgroup partial buffers are carved for max(2B, 512) slices
static int dg_parts(const alignnet_handle* h, int B, int per_cu)
{
  const int ntiles = (h->cfg.num_points + kTT - 1) / kTT, slots = 256 * per_cu;
  const int cap = std::max(1, std::min(std::min(kDgMaxParts, ntiles), kDgPartSlices / (2 * B)));
  if (h->dg_parts_opt > 0) return std::min(h->dg_parts_opt, cap);
  int p = 1;
  while (p * 2 <= cap && 2 * B * p * 2 <= slots) p *= 2;
  return p;
}

// PointNet per-cloud kernels that only leave SUMS behind (phase 2 / the first-layer Gram, passes B2 and B1): the same split.  They run two four-wave
// workgroups per CU, so 2B = 512 clouds fill the chip; the reference's shipped configs (batch 128 x 512 points) bring 256.  Phase 3 keeps one workgroup
// per cloud (its per-cloud maxima and Gram would need a merge step).
static int pn_parts(const alignnet_handle* h, int B)
{
  const int ntiles = (h->cfg.num_points + kTT - 1) / kTT;
  const int cap = std::max(1, std::min(std::min(kDgMaxParts, ntiles), kDgPartSlices / (2 * B)));
  if (h->pn_parts_opt > 0) return std::min(h->pn_parts_opt, cap);
  int p = 1;
  while (p * 2 <= cap && 2 * B * p * 2 <= 512) p *= 2;
  return p;
}

// Phase 3 (lift to C3, running arg-extreme, Gram): `slots` workgroups fill the chip (256 for the 128-point-tile kernels: one per CU; 512 for the 64-point
// ones), `tile` = points per tile.  The parts' extremes are folded by merge_ext_parts_kernel (exact), Gram / column sums by the reductions.
static int p3_parts(const alignnet_handle* h, int B, int slots, int tile)
{
  const int ntiles = (h->cfg.num_points + tile - 1) / tile;
  const int cap = std::max(1, std::min(std::min(kDgMaxParts, ntiles), kDgPartSlices / (2 * B)));
  if (h->pn_parts_opt > 0) return std::min(h->pn_parts_opt, cap);
  int p = 1;
  while (p * 2 <= cap && 2 * B * p * 2 <= slots) p *= 2;
  return p;
}

static const Stack& conv_of(const alignnet_handle* h, int s) { return s == 0 ? h->s1_conv : s == 1 ? h->s2_conv : h->emb_conv; }
static const Stack& fc_of(const alignnet_handle* h, int s) { return s == 0 ? h->s1_fc : s == 1 ? h->s2_fc : h->rem_fc; }
static float keep_of(const alignnet_handle* h, int s) { return s == 0 ? h->cfg.s1_keep : s == 1 ? h->cfg.s2_keep : h->cfg.rem_keep; }
static float* P(alignnet_handle* h, int pidx) { return h->d_params + h->params[pidx].offset; }
static float* G(alignnet_handle* h, TrainWS* w, int pidx) { return w->grad + h->params[pidx].offset; }

static size_t img_floats(int K, int C) { return (size_t)((C + 31) / 32) * ((K + 7) / 8) * 256; }

// A PointNet stage outside the shape the specialised kernels are built for (three conv layers, widths multiples of 32, C1, C2 <= 128,
// C3 <= 1024) trains on the general layer-by-layer path -- e.g. the five-layer backbones of the reference's configs/default.json.
// The dgcnn stages the specialised edge kernels are built for: [C1, C2, C3] with C1 in {32, 64}, C2 in {64, 128}.
static bool dg_special_shape(const alignnet_handle* h, int s)
{
  const Stack& st = conv_of(h, s);
  if (st.n != 3) return false;
  const int C1 = h->layers[st.first].cout, C2 = h->layers[st.first + 1].cout, C3 = h->layers[st.first + 2].cout;
  if (C1 % 32 || C2 % 32 || C3 % 32 || C3 > 1024) return false;
  const int CT1 = (C1 + 31) / 32;
  return (C1 == 32 || C1 == 64) && (C2 == 64 || C2 == 128) && CT1 * 2 + CT1 * (CT1 + 1) / 2 <= kBEW && (size_t)C1 * C2 <= 8192 &&
         dg_bwd_edge_lds(C1, C2) <= 160 * 1024;
}

static bool stage_generic(const alignnet_handle* h, int s)
{
  if (h->cfg.backbone == 1) return !dg_special_shape(h, s);   // any other `sizes` list of models/tp8.py:38-41: layer by layer over the edge rows
  const Stack& st = conv_of(h, s);
  if (st.n != 3) return true;
  const int C1 = h->layers[st.first].cout, C2 = h->layers[st.first + 1].cout, C3 = h->layers[st.first + 2].cout;
  return C1 % 32 || C2 % 32 || C3 % 32 || C1 > 128 || C2 > 128 || C3 > 1024;
}

// A general-depth stage whose LAST TWO layers fit the fused kernels (C_{n-2} in multiples of 32 up to 128, C_{n-1} in multiples of 32
// up to 1024, first layer <= 128 wide): the layers up to n - 2 run layer by layer (kernels_train_generic.h), their output
// h = relu(bn(Z_{n-2})) is materialised once ([B N, C_{n-2}]) and the last layer runs on the "given features" kernels of the DGCNN
// branch -- phase 3 / pass B2 with GIVEN = true -- so that the [B N, C_last] tensors of an unfused last layer (2 GB each at
// B = 256, N = 1024, C = 1024: 17 of default.json's 29 ms) never exist.  alignnet_set_option("train_fused_tail", 0) keeps the plain
// layer-by-layer path (tests compare the two).
static bool stage_hybrid(const alignnet_handle* h, int s)
{
  if (!stage_generic(h, s) || !h->fused_tail || h->cfg.backbone == 1) return false;
  const Stack& st = conv_of(h, s);
  if (st.n < 3) return false;
  const int C1 = h->layers[st.first].cout, C2 = h->layers[st.first + st.n - 2].cout, C3 = h->layers[st.first + st.n - 1].cout;
  return C1 <= 128 && C1 % 8 == 0 && C2 % 32 == 0 && C2 <= 128 && C3 % 32 == 0 && C3 <= 1024;
}

static int check_trainable_shape(alignnet_handle* h)
{
  const bool dg = h->cfg.backbone == 1;
  if (h->sync_bn && (h->ab & (AB_B1_LEGACY | AB_PHASE2_LEGACY)))   // those variants keep per-rank (dbeta1, dgamma1) / column sums: not summed over the ranks
    return fail(h, "sync_bn: not available with the ab_b1_legacy / ab_phase2_legacy kernel variants");
  if (h->sync_bn)
    for (int s = 0; s < 3; ++s) {
      if (stage_generic(h, s)) continue;   // layer by layer: the per-layer sums travel through gen_stat_finish / gen_bn_bwd_finish's sync modes
      if (!dg && h->layers[conv_of(h, s).first].cout > 64) return fail(h, "sync_bn: first conv width limited to 64");
    }
  for (int s = 0; s < 3; ++s)
    if (stage_generic(h, s)) {
      if (dg && conv_of(h, s).n < 2) return fail(h, "training: a dgcnn stage needs at least one edge conv and the point conv");
      const Stack& st = conv_of(h, s);
      for (int i = 0; i < st.n; ++i) {
        const Layer& L = h->layers[st.first + i];
        if (L.cout % 8) return fail(h, "training: conv widths must be multiples of 8 (layer " + L.name + ")");
        if (L.cout > (i + 1 < st.n ? kGenMaxK : 4096))
          return fail(h, "training (general-depth path): hidden conv widths are limited to " + std::to_string(kGenMaxK) + " channels, the last to 4096 (layer " + L.name + ")");
      }
    }
  if (dg && h->train_bf16)   // (the bf16 edge conv: dg_train_fwd<C1, true>; the point conv and the backward stay fp32)
    for (int s = 0; s < 3; ++s)
      if (!stage_generic(h, s) && h->layers[conv_of(h, s).first].cout % 16) return fail(h, "training: train_matmul_bf16 with the dgcnn backbone needs a first edge width that is a multiple of 16");
  if (dg && (h->cfg.num_points > 64 * kKnnMaxPerLane || h->cfg.num_points < kDgK))
    return fail(h, "training: dgcnn needs 20 <= num_points <= 4096");
  for (int s = 0; s < 3; ++s) {
    const Stack& st = conv_of(h, s);
    if (stage_generic(h, s)) continue;
    const int C1 = h->layers[st.first].cout, C2 = h->layers[st.first + 1].cout, C3 = h->layers[st.first + 2].cout;
    if (C1 % 32 || C2 % 32 || C3 % 32) return fail(h, "training: conv widths must be multiples of 32");
    if (C1 > 128 || C2 > 128 || C3 > 1024) return fail(h, "training: conv widths limited to C1,C2 <= 128, C3 <= 1024");
    const int CT2 = (C2 + 31) / 32;
    (void)CT2;
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------
static int ensure_train_ws(alignnet_handle* h, int B)
{
  TrainWS* w = tws(h);
  if (!w->grad) {
    HIP_TRY(h, hipMalloc(&w->grad, h->n_trainable * sizeof(float)));
    HIP_TRY(h, hipMalloc(&w->adam_m, h->n_trainable * sizeof(float)));
    HIP_TRY(h, hipMalloc(&w->adam_v, h->n_trainable * sizeof(float)));
    HIP_TRY(h, hipMemset(w->grad, 0, h->n_trainable * sizeof(float)));
    HIP_TRY(h, hipMemset(w->adam_m, 0, h->n_trainable * sizeof(float)));
    HIP_TRY(h, hipMemset(w->adam_v, 0, h->n_trainable * sizeof(float)));
    w->grad_clean = true;
  }
  if (!h->sync_buf) HIP_TRY(h, hipMalloc(&h->sync_buf, kSyncBufDoubles * sizeof(double)));
  if (B <= w->cap && !h->train_ws_stale) return 0;
  B = std::max(B, w->cap);
  h->train_ws_stale = false;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (w->base) { hipFree(w->base); w->base = nullptr; }
  for (int t = 0; t < 2; ++t) if (w->d_pcs[t]) { hipFree(w->d_pcs[t]); w->d_pcs[t] = nullptr; }
  const int N = h->cfg.num_points, nb2 = 2 * h->cfg.num_bins;
  const size_t B2 = 2 * (size_t)B, MN = B2 * N;
  int maxC = 8, maxC1 = 8, maxC2 = 8, maxC3 = 8, maxH = 8;
  int genC = 0, genK = 8;   // widest layer / widest MFMA-layer input of the general-depth stages
  size_t genRC = 0, genPart = 0, genDW = 0;   // per tower, the largest over the layer-by-layer tensors of: rows x width (gradient ping-pong buffers), tiles x width (statistics partials), slabs x K x C (dW partials)
  for (int s = 0; s < 3; ++s) {
    const Stack& st = conv_of(h, s);
    {
      const Stack& fs = fc_of(h, s);
      for (int j = 0; j < fs.n; ++j) maxH = std::max(maxH, h->layers[fs.first + j].cout);
    }
    if (stage_generic(h, s)) {
      const bool hyb = stage_hybrid(h, s);   // the last layer belongs to the fused kernels: no [B N, C_last] buffers
      for (int i = 0; i < st.n - (hyb ? 1 : 0); ++i) {
        genC = std::max(genC, h->layers[st.first + i].cout); if (i) genK = std::max(genK, h->layers[st.first + i].cin);
        const size_t rows = (size_t)B * N * ((h->cfg.backbone == 1 && i < st.n - 1) ? kDgK : 1);   // per tower
        genRC = std::max(genRC, rows * h->layers[st.first + i].cout);
        genPart = std::max(genPart, ((rows + kGenTile - 1) / kGenTile) * h->layers[st.first + i].cout);
        if (i) genDW = std::max(genDW, ((rows + kGenSlab - 1) / kGenSlab) * h->layers[st.first + i].cin * h->layers[st.first + i].cout);
      }
      maxC3 = std::max(maxC3, h->layers[st.first + st.n - 1].cout);
      if (hyb) maxC2 = std::max(maxC2, h->layers[st.first + st.n - 2].cout);
      continue;
    }
    maxC1 = std::max(maxC1, h->layers[st.first].cout);
    maxC2 = std::max(maxC2, h->layers[st.first + 1].cout);
    maxC3 = std::max(maxC3, h->layers[st.first + 2].cout);
    const Stack& fs = fc_of(h, s);
    for (int j = 0; j < fs.n; ++j) maxH = std::max(maxH, h->layers[fs.first + j].cout);
  }
  maxC = std::max(maxC1, std::max(maxC2, maxC3));
  // two passes over the same carve code: size, then assign
  for (int pass = 0; pass < 2; ++pass) {
    size_t off = 0;
    auto take = [&](size_t nbytes) -> char* {
      char* p = pass ? w->base + off : nullptr;
      off += (nbytes + 255) & ~(size_t)255;
      return p;
    };
    auto F = [&](size_t n) { return reinterpret_cast<float*>(take(n * sizeof(float))); };
    auto D = [&](size_t n) { return reinterpret_cast<double*>(take(n * sizeof(double))); };
    auto I = [&](size_t n) { return reinterpret_cast<int*>(take(n * sizeof(int))); };
    const int lw[6] = {3, 1, 3, 3, 1, 1};
    for (int i = 0; i < 6; ++i) w->labels[i] = F((size_t)B * lw[i]);
    w->dropout_u = F(5 * (size_t)B * maxH);
    w->center_mean = F(B2 * 3); w->s1c = F(B2 * 3); w->s2c = F(B2 * 3); w->theta = F(B2); w->cls = I(B2);
    for (int s = 0; s < 3; ++s) {
      const Stack& st = conv_of(h, s);
      const bool gen = stage_generic(h, s), hyb = stage_hybrid(h, s);
      // specialised stages: [C1, C2, C3]; general-depth stages only need the stage interface (frame, pooled output and its gradient,
      // frame gradients), sized by the LAST layer's width; hybrid stages also the fused tail's buffers for [C_{n-2}, C_{n-1}]
      const int Cl = h->layers[st.first + st.n - 1].cout;
      const int C[3] = {gen ? 8 : h->layers[st.first].cout, hyb ? h->layers[st.first + st.n - 2].cout : gen ? 8 : h->layers[st.first + 1].cout, Cl};
      StageWS& S = w->st[s];
      S.xform = F(B2 * 12);
      if (gen) {
        TrainWS::GenStage& Gs = w->gen[s];
        const bool dgg = h->cfg.backbone == 1;            // general dgcnn stage: the layers in front of the last one live on the B N k edge rows
        const size_t ME = dgg ? MN * kDgK : MN;
        Gs.X0 = F(dgg ? ME * 8 : MN * 4);
        for (int i = 0; i < st.n; ++i) {
          const int c = h->layers[st.first + i].cout;
          Gs.Z[i] = F((hyb && i == st.n - 1) ? 8 : (i < st.n - 1 ? ME : MN) * c); Gs.mean[i] = F(2 * c); Gs.rstd[i] = F(2 * c); Gs.scale[i] = F(2 * c); Gs.shift[i] = F(2 * c);
        }
        Gs.idx = I(B2 * Cl);
        const int Cp = dgg ? h->layers[st.first + st.n - 2].cout : 8;
        Gs.P = F(dgg ? MN * Cp : 8); Gs.argk = I(dgg ? MN * Cp : 8); Gs.one = F(2 * Cp); Gs.zero = F(2 * Cp);
      }
      for (int l = 0; l < 3; ++l) { S.mean[l] = F(2 * C[l]); S.var[l] = F(2 * C[l]); S.scale[l] = F(2 * C[l]); S.shift[l] = F(2 * C[l]); S.rstd[l] = F(2 * C[l]); S.kk[l] = F(2 * C[l]); }
      S.sgn3 = F(2 * C[2]);
      S.ext = F(B2 * 2 * C[2]); S.idx2 = I(B2 * 2 * C[2]);
      S.idx = I(B2 * C[2]); S.zhat_star = F(B2 * C[2]);
      S.extp = F((size_t)kDgPartSlices * 2 * C[2]); S.idxp = I((size_t)kDgPartSlices * 2 * C[2]);   // phase 3 split over workgroups (p3_parts: only below 512 of them)
      S.h2 = F((gen && !hyb) ? 8 : MN * C[1]);
      S.gram2 = F(2 * (size_t)C[1] * C[1]); S.gram2raw = F(2 * (size_t)C[1] * C[1]); S.s2 = F(2 * C[1]); S.m2 = F(2 * C[1]);
      S.pooled = F(B2 * C[2]); S.dP = F(B2 * C[2]);
      if (s < 2) { S.tower_stride = (long)B * C[2]; S.row_stride = C[2]; }
      else { S.tower_stride = C[2]; S.row_stride = 2L * C[2]; }
      S.gx = F(B2 * 3); S.grot = F(B2);
      const bool dgb = h->cfg.backbone == 1;
      S.argk = reinterpret_cast<unsigned char*>(take(dgb ? MN * C[1] : 0));
      (void)gen;
      S.mom = D(B2 * kDgMom); S.s1e = F(2 * C[0]); S.g1f = F(2 * (size_t)C[0] * C[0]);
      S.gram2raw64 = D(2 * (size_t)C[1] * C[1] + 2 * C[1]); S.s264 = S.gram2raw64 ? S.gram2raw64 + 2 * (size_t)C[1] * C[1] : nullptr;
      S.g1f64 = D(2 * (size_t)C[0] * C[0] + 2 * C[0]); S.s1e64 = S.g1f64 ? S.g1f64 + 2 * (size_t)C[0] * C[0] : nullptr;
      S.gs = F(B2 * C[2]); S.E3 = F(2 * C[2]); S.kdb3 = F(2 * C[2]); S.Sp = F(2 * (size_t)C[1] * C[2]); S.GW = F(2 * (size_t)C[1] * C[2]);
      const size_t dgs = std::max(B2, (size_t)kDgPartSlices);   // one partial per WORKGROUP of the split per-cloud kernels (dg_parts, pn_parts)
      S.u2_part = F(dgs * (size_t)C[0] * C[1]); S.g1_part = F(dgs * (size_t)C[0] * C[0]); S.p_part = F(B2 * 6 * C[0]);
      S.u2 = F(2 * (size_t)C[0] * C[1]); S.g1 = F(2 * (size_t)C[0] * C[0]); S.s1 = F(2 * C[0]); S.m1 = F(2 * C[0]);
      S.E2 = F(2 * C[1]); S.kdb2 = F(2 * C[1]); S.k2 = F(2 * C[1]); S.GW2 = F(2 * (size_t)C[0] * C[1]);
      if (hyb && pass) {   // the fused tail reads the statistics of the layers in front of it where the fused stages keep theirs
        TrainWS::GenStage& Gs = w->gen[s];
        S.scale[0] = Gs.scale[0]; S.shift[0] = Gs.shift[0];
        S.mean[1] = Gs.mean[st.n - 2]; S.rstd[1] = Gs.rstd[st.n - 2]; S.scale[1] = Gs.scale[st.n - 2]; S.shift[1] = Gs.shift[st.n - 2];
      }
      const Stack& fs = fc_of(h, s);
      const size_t M = s < 2 ? B2 : (size_t)B;
      for (int j = 0; j < fs.n - 1; ++j) {
        const int wd = h->layers[fs.first + j].cout;
        HeadLayerWS& L = w->hl[s][j];
        L.z = F(M * wd); L.y = F(M * wd); L.dz = F(M * wd); L.dyb = F(M * wd); L.mean = F(2 * wd); L.var = F(2 * wd);
      }
      const int ow = h->layers[fs.first + fs.n - 1].cout;
      w->o[s] = F(M * ow); w->d_o[s] = F(M * ow);
      w->head_din[s] = S.dP;
    }
    if (genC) {
      const size_t M1 = (size_t)B * N * (h->cfg.backbone == 1 ? kDgK : 1);   // rows per tower of the widest layer-by-layer tensors (dgcnn: edge rows)
      w->gen_tiles = (int)((M1 + kGenTile - 1) / kGenTile); w->gen_slabs = (int)((M1 + kGenSlab - 1) / kGenSlab);
      w->gen_d[0] = F(2 * genRC); w->gen_d[1] = F(2 * genRC);
      w->gen_part = F(2 * genPart * 2);
      w->gen_dwpart = F(2 * std::max(genDW, (size_t)8));
      w->gen_ppart = F(B2 * 6 * (size_t)genC);
      w->gen_cA = F(2 * genC); w->gen_cB = F(2 * genC);
      w->gen_wt = F(img_floats(genC, genK) + 1024);
    }
    w->d_s1c = F(B2 * 3); w->d_s2c = F(B2 * 3);
    w->loss_out = F(32); w->loss_scratch = F(loss_scratch_floats(B));
    const size_t dgs = std::max(B2, (size_t)kDgPartSlices);
    w->stat_part = D(std::max(B2 * 4 * maxC, dgs * 4 * maxC2) * 2); w->gram_part = F(dgs * (size_t)maxC2 * maxC2); w->colsum_part = D(dgs * std::max((size_t)4 * maxC2, (size_t)1024));   // [2B][slices][C2]: 4 slices (PointNet), 1024 / C2 (bf16 point conv of the dgcnn branch)
    w->dy2 = F(MN * maxC2); w->dy1 = F(h->cfg.backbone == 1 ? 0 : MN * maxC1);
    w->nn = I(h->cfg.backbone == 1 ? MN * kDgK : 0); w->pdy_part = D(dgs * 4 * 7 * maxC1);
    w->dbg2_part = D(dgs * 4 * maxC2 * 2); w->dbg1_part = D(dgs * 4 * maxC1 * 2); w->s1_part = D(dgs * 1024);   // [2B][row groups][C1]: 256 / C1 groups (PointNet kernels), 1024 / C1 (dg_train_fwd)
    w->dbg2 = F(4 * maxC2); w->dbg1 = F(4 * maxC1);
    w->W3E = F(2 * (size_t)maxC2 * maxC3);
    w->W3T = F((size_t)maxC2 * maxC3); w->Q3 = F(2 * (size_t)maxC2 * maxC2); w->q3b = F(2 * maxC2);
    w->q3img = F(2 * (size_t)maxC2 * maxC2 + 1024);
    w->u3 = F(2 * (size_t)maxC3);
    w->rstd2 = F(2 * maxC2);
    w->W2E = F(2 * (size_t)maxC1 * maxC2); w->V2 = F(2 * (size_t)maxC1 * maxC2); w->Q2 = F(2 * (size_t)maxC1 * maxC1);
    w->q2b = F(2 * maxC1); w->v2img = F(2 * (size_t)maxC1 * maxC2 + 1024); w->q2img = F(2 * (size_t)maxC1 * maxC1 + 1024);
    w->k1 = F(2 * maxC1); w->rstd1 = F(2 * maxC1);
    const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
    for (int i = 0; i < 8; ++i) w->outs[i] = F((size_t)B * widths[i]);
    if (!pass) {
      w->bytes = off;
      HIP_TRY(h, hipMalloc(&w->base, off));
      HIP_TRY(h, hipMemset(w->base, 0, off));
    }
  }
  for (int t = 0; t < 2; ++t) HIP_TRY(h, hipMalloc(&w->d_pcs[t], (size_t)B * N * 3 * sizeof(float)));
  {   // pack jobs of the backward (pointers into the carve above)
    std::vector<PackJob> jobs;
    for (int s = 0; s < 3; ++s) {
      const Stack& st = conv_of(h, s);
      const bool gen = stage_generic(h, s), hyb = stage_hybrid(h, s);   // (general-depth stages do not use these images: harmless 8 x 8 jobs keep the table's layout)
      const int C1 = gen ? 8 : h->layers[st.first].cout, C2 = hyb ? h->layers[st.first + st.n - 2].cout : gen ? 8 : h->layers[st.first + 1].cout;
      const size_t qimg = img_floats(C2, C2), vimg = img_floats(C2, C1), q2img = img_floats(C1, C1);
      for (int t = 0; t < 2; ++t) jobs.push_back(PackJob{w->Q3 + (size_t)t * C2 * C2, w->q3img + t * qimg, C2, C2});
      for (int t = 0; t < 2; ++t) jobs.push_back(PackJob{w->Q2 + (size_t)t * C1 * C1, w->q2img + t * q2img, C1, C1});
      for (int t = 0; t < 2; ++t) jobs.push_back(PackJob{w->V2 + (size_t)t * C1 * C2, w->v2img + t * vimg, C2, C1});
    }
    if (!w->stage_pack) HIP_TRY(h, hipMalloc(&w->stage_pack, jobs.size() * sizeof(PackJob)));
    HIP_TRY(h, hipMemcpy(w->stage_pack, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
  }
  w->cap = B;
  return 0;
}

// ---------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------
static void launch_gemm(alignnet_handle* h, const float* A, long sai, long sak, const float* Bm, long sbk, long sbj, float* C,
                        long sci, long scj, int M, int N, int K, const float* bias = nullptr, float alpha = 1.f, int acc = 0,
                        int batch = 1, long ba = 0, long bb = 0, long bc = 0)
{
  GemmArgs g{A, sai, sak, Bm, sbk, sbj, C, sci, scj, M, N, K, bias, alpha, acc, ba, bb, bc};
  hipLaunchKernelGGL(gemm_small, dim3((N + 31) / 32, (M + 31) / 32, batch), dim3(kGemmWaves * 64), 0, h->stream, g);
}

static GemmArgs gemm_args(const float* A, long sai, long sak, const float* Bm, long sbk, long sbj, float* C, long sci, long scj, int M, int N, int K)
{
  return GemmArgs{A, sai, sak, Bm, sbk, sbj, C, sci, scj, M, N, K, nullptr, 1.f, 0, 0, 0, 0};
}
static void launch_gemm_pair(alignnet_handle* h, const GemmArgs& g0, const GemmArgs& g1)
{
  GemmPair p;
  p.g[0] = g0; p.g[1] = g1;
  p.tx[0] = (g0.N + 31) / 32; p.tx[1] = (g1.N + 31) / 32;
  p.n0 = p.tx[0] * ((g0.M + 31) / 32);
  const int n1 = p.tx[1] * ((g1.M + 31) / 32);
  hipLaunchKernelGGL(gemm_small2, dim3(p.n0 + n1), dim3(kGemmWaves * 64), 0, h->stream, p);
}

template <typename T>
static void launch_reduce(alignnet_handle* h, const T* part, int S, long n, float* out, int towers = 2, float alpha = 1.f, int acc = 0)
{
  hipLaunchKernelGGL((reduce_slices_kernel<T>), dim3((unsigned)((n + 31) / 32), towers), dim3(1024), 0, h->stream, part, S, n, out, alpha, acc);
}

static ReduceJob rjob(const float* part, int S, long n, float* out, float alpha = 1.f, int towers = 2) { return ReduceJob{part, 0, S, n, out, alpha, towers}; }
static ReduceJob rjob(const double* part, int S, long n, float* out, float alpha = 1.f, int towers = 2) { return ReduceJob{part, 1, S, n, out, alpha, towers}; }
static void launch_reduce_multi(alignnet_handle* h, int njobs, ReduceJob a, ReduceJob b, ReduceJob c = ReduceJob{nullptr, 0, 0, 0, nullptr, 1.f, 0})
{
  ReduceJobs J{{a, b, c}};
  long nmax = 0;
  for (int i = 0; i < njobs; ++i) nmax = std::max(nmax, J.j[i].n);
  hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)((nmax + 31) / 32), 2, njobs), dim3(1024), 0, h->stream, J);
}

static void launch_loss(alignnet_handle* h, const LossArgs& la)
{
  const int nprep = (3 * la.B + 63) / 64;   // softmax-row workgroups of loss_prep_kernel (+1 for the Huber terms)
  hipLaunchKernelGGL(loss_prep_kernel, dim3(nprep + 1 + kLossGroups), dim3(1024), 0, h->stream, la, kLossGroups, nprep);
  hipLaunchKernelGGL(loss_final_kernel, dim3((la.B + kLossCols - 1) / kLossCols), dim3(256), 0, h->stream, la, kLossGroups, nprep);
}


// ---- deferred weight-gradient work (struct Deferred): each helper either records the job or, with deferral off, launches it in place
static void def_reduce(alignnet_handle* h, TrainWS* w, const ReduceJob& j)
{
  if (w->defer.on) { w->defer.red.push_back(j); return; }
  ReduceJobs J{}; J.j[0] = j;
  hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)((j.n + 31) / 32), 2, 1), dim3(1024), 0, h->stream, J);
}
static void def_sparse(alignnet_handle* h, TrainWS* w, const SparseDwJob& j)
{
  if (w->defer.on) { w->defer.sp.push_back(j); return; }
  hipLaunchKernelGGL(sparse_dw_kernel, dim3((j.C3 + kSdC - 1) / kSdC, 2), dim3(1024), 0, h->stream, j.gs, j.idx, j.h2, j.B, j.N, j.C2, j.C3, j.Sp, j.h2_bf16);
}
static void def_centre(alignnet_handle* h, TrainWS* w, const CentreJob& j)
{
  if (w->defer.on) { w->defer.cen.push_back(j); return; }
  hipLaunchKernelGGL(centre_gram_kernel, dim3((unsigned)(((size_t)j.C * j.C + 255) / 256), 2), dim3(256), 0, h->stream, j.G, j.s, j.C, j.M, j.m);
}
static void def_gemm(alignnet_handle* h, TrainWS* w, const GemmArgs& g, int batch = 1)
{
  if (w->defer.on) { w->defer.gemm.push_back({g, batch}); return; }
  hipLaunchKernelGGL(gemm_small, dim3((g.N + 31) / 32, (g.M + 31) / 32, batch), dim3(kGemmWaves * 64), 0, h->stream, g);
}
static void def_combine(alignnet_handle* h, TrainWS* w, const CombineJob& j)
{
  if (w->defer.on) { w->defer.comb.push_back(j); return; }
  hipLaunchKernelGGL(combine_dw_kernel, dim3((unsigned)(((size_t)j.R * j.C + 255) / 256)), dim3(256), 0, h->stream, j.Sp, j.spscale, j.m, j.kdb, j.GW, j.E, j.R, j.C, j.dW, j.gscale);
}
// the recorded jobs, five launches: reductions | sparse gathers -> Gram centrings -> GEMMs -> combines.  A group that outgrows its
// job table (deep heads: one dW product per FC layer, ALIGNNET_MAX_WIDTHS = 8 layers per head) goes out in several launches of the same kernel.
static int flush_deferred(alignnet_handle* h, hipStream_t stream)
{
  TrainWS* w = tws(h);
  Deferred& d = w->defer;
  for (size_t i0 = 0; i0 < d.red.size(); i0 += kReduceJobs) {
    const size_t nj = std::min(d.red.size() - i0, (size_t)kReduceJobs);
    ReduceJobs J{}; long nmax = 0;
    for (size_t i = 0; i < nj; ++i) { J.j[i] = d.red[i0 + i]; nmax = std::max(nmax, d.red[i0 + i].n); }
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)((nmax + 31) / 32), 2, (unsigned)nj), dim3(1024), 0, stream, J);
  }
  for (auto& sr : d.sync_after_red) if (sync_sum(h, sr.first, sr.second, false)) return 1;
  constexpr size_t kSp = sizeof(SparseDwJobs::j) / sizeof(SparseDwJob), kCen = sizeof(CentreJobs::j) / sizeof(CentreJob), kComb = sizeof(CombineJobs::j) / sizeof(CombineJob);
  // gathers and Gram centrings in one launch where both exist (chunk by chunk; whatever is left of either goes alone)
  {
    size_t is = 0, ic = 0;
    while (is < d.sp.size() || ic < d.cen.size()) {
      const size_t ns = std::min(d.sp.size() - is, kSp), nc = std::min(d.cen.size() - ic, kCen);
      SparseDwJobs J{}; CentreJobs Cj{}; int cmax = 0; size_t emax = 0;
      for (size_t i = 0; i < ns; ++i) { J.j[i] = d.sp[is + i]; cmax = std::max(cmax, d.sp[is + i].C3); }
      for (size_t i = 0; i < nc; ++i) { Cj.j[i] = d.cen[ic + i]; emax = std::max(emax, (size_t)d.cen[ic + i].C * d.cen[ic + i].C); }
      if (ns) {
        const unsigned gx = (unsigned)std::max<size_t>((cmax + kSdC - 1) / kSdC, (emax + 1023) / 1024);
        hipLaunchKernelGGL(sparse_dw_jobs_kernel, dim3(gx, 2, (unsigned)(ns + nc)), dim3(1024), 0, stream, J, (int)ns, Cj);
      } else
        hipLaunchKernelGGL(centre_gram_jobs_kernel, dim3((unsigned)((emax + 255) / 256), 2, (unsigned)nc), dim3(256), 0, stream, Cj);
      is += ns; ic += nc;
    }
  }
  // the throughput-shaped products on 64 x 64 tiles (gemm_tile64_jobs), the rest on the K-split tiles; tiles numbered job after job
  for (int big = 1; big >= 0; --big) {
    std::vector<size_t> sel;
    for (size_t i = 0; i < d.gemm.size(); ++i) if ((int)(gemm_job_is_big(d.gemm[i].first) && !(h->ab & AB_GEMM_JOBS_KSPLIT)) == big) sel.push_back(i);
    // longest first: a workgroup's time is its K; with the deep products at the end of the grid they would run alone in a second round
    std::stable_sort(sel.begin(), sel.end(), [&](size_t x, size_t y) { return d.gemm[x].first.K > d.gemm[y].first.K; });
    const int ts = big ? kGemmT : 32;
    for (size_t i0 = 0; i0 < sel.size(); i0 += kGemmJobs) {
      const size_t nj = std::min(sel.size() - i0, (size_t)kGemmJobs);
      GemmJobs J{}; int tot = 0;
      for (size_t i = 0; i < nj; ++i) {
        const GemmArgs& g = d.gemm[sel[i0 + i]].first;
        J.g[i] = g; J.tx[i] = (g.N + ts - 1) / ts; J.ty[i] = (g.M + ts - 1) / ts; J.start[i] = tot;
        tot += J.tx[i] * J.ty[i] * d.gemm[sel[i0 + i]].second;
      }
      J.start[nj] = tot; J.n = (int)nj;
      if (big) hipLaunchKernelGGL(gemm_tile64_jobs, dim3(tot), dim3(kGemmWaves * 64), 0, stream, J);
      else hipLaunchKernelGGL(gemm_small_jobs, dim3(tot), dim3(kGemmWaves * 64), 0, stream, J);
    }
  }
  for (size_t i0 = 0; i0 < d.comb.size(); i0 += kComb) {
    const size_t nj = std::min(d.comb.size() - i0, kComb);
    CombineJobs J{}; size_t emax = 0;
    for (size_t i = 0; i < nj; ++i) { J.j[i] = d.comb[i0 + i]; emax = std::max(emax, (size_t)d.comb[i0 + i].R * d.comb[i0 + i].C); }
    hipLaunchKernelGGL(combine_dw_jobs_kernel, dim3((unsigned)((emax + 255) / 256), 1, (unsigned)nj), dim3(256), 0, stream, J);
  }
  d.clear();
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// stage s is a specialised PointNet stage: its first layer's statistics come from the cloud's moments (backbone_fwd_train's last branch), which the
// kernel writing the stage's frame can form right away
static MomentsTail moments_tail_of(alignnet_handle* h, int s)
{
  MomentsTail mt;
  if (h->cfg.backbone != 0 || stage_generic(h, s) || (h->ab & AB_NO_GLUE_FOLD)) return mt;
  TrainWS* w = tws(h);
  const Layer& L0 = h->layers[conv_of(h, s).first];
  mt.mom = w->st[s].mom; mt.w1 = P(h, L0.p_w); mt.b1 = P(h, L0.p_b); mt.C1 = L0.cout; mt.stat_part = w->stat_part;
  return mt;
}

// the step's weight images: queued into `start` (train_start_kernel launches them with the centroids) or, start == nullptr, launched here
struct StartJobs { const PackJob* f32 = nullptr; int nf32 = 0; PackBf16Jobs pj{}; int nbf16 = 0; };
static int pack_all_weights(alignnet_handle* h, StartJobs* start = nullptr)
{
  TrainWS* w = tws(h);
  if (!w->pack_table) {   // (src, dst, K, C) per MFMA conv layer, built once.  (The head layers' images are eval-only: fold_for_eval rebuilds every
                          //  image when an eval forward follows a training step; packing them here too was 1.9 M of the step's 2.1 M elements.)
    std::vector<PackJob> jobs;
    for (const Layer& L : h->layers)
      if (L.conv && !L.first_conv) jobs.push_back(PackJob{P(h, L.p_w), h->d_wp + L.off_wp, L.cin, L.cout});
    w->n_pack = (int)jobs.size();
    HIP_TRY(h, hipMalloc(&w->pack_table, jobs.size() * sizeof(PackJob)));
    HIP_TRY(h, hipMemcpy(w->pack_table, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
  }
  // the fp32 MFMA images of the conv layers are read by the fp32 kernels, the dgcnn branch and the general-depth stages; in bf16 mode
  // with only specialised PointNet stages nothing reads them (the heads use the raw matrices, the backbones the bf16 images below)
  bool need_f32 = !h->train_bf16 || h->cfg.backbone == 1;
  for (int s = 0; s < 3; ++s) need_f32 = need_f32 || stage_generic(h, s);
  if (need_f32 && w->n_pack) {
    if (start) { start->f32 = w->pack_table; start->nf32 = w->n_pack; }
    else hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(256, w->n_pack), dim3(256), 0, h->stream, w->pack_table);   // (<= 2 elements per thread)
  }
  if (h->train_bf16) {
    // bf16 images of every specialised stage's lift (one per tower: the sign of that tower's gamma folded in) and hidden layer, one launch
    PackBf16Jobs pj{};
    int nj = 0;
    for (int s = 0; s < 3; ++s) {
      if (stage_generic(h, s) && !stage_hybrid(h, s)) continue;   // the layer-by-layer path computes in fp32 whatever the option says
      const Stack& cst = conv_of(h, s);
      const Layer& L = h->layers[cst.first + cst.n - 1];   // (hybrid stages: only the last layer, their fused tail, takes bf16 operands)
      const size_t n = (size_t)((L.cout + 31) / 32) * ((L.cin + 15) / 16) * 512;   // per tower
      if (!w->wp3h[s]) HIP_TRY(h, hipMalloc(&w->wp3h[s], 2 * n * sizeof(unsigned short)));
      for (int t = 0; t < 2; ++t) {
        pj.src[nj] = P(h, L.p_w); pj.gamma[nj] = P(h, L.p_bn[t][1]); pj.dst[nj] = w->wp3h[s] + t * n; pj.K[nj] = L.cin; pj.C[nj] = L.cout; ++nj;
      }
      if (stage_hybrid(h, s)) continue;
      const Layer& L2 = h->layers[conv_of(h, s).first + 1];   // hidden layer: one image, no sign folding
      const size_t n2 = (size_t)((L2.cout + 31) / 32) * ((L2.cin + 15) / 16) * 512;
      if (!w->wp2h[s]) HIP_TRY(h, hipMalloc(&w->wp2h[s], n2 * sizeof(unsigned short)));
      pj.src[nj] = P(h, L2.p_w); pj.gamma[nj] = nullptr; pj.dst[nj] = w->wp2h[s]; pj.K[nj] = L2.cin; pj.C[nj] = L2.cout; ++nj;
      if (h->cfg.backbone == 1) {   // dense edge backward (dg_train_bwd_edge_dense): image of W2^T, element (k = c2, c = c1) = W2[c1][c2]
        const size_t nt = (size_t)((L2.cin + 31) / 32) * ((L2.cout + 15) / 16) * 512;
        if (!w->w2th[s]) HIP_TRY(h, hipMalloc(&w->w2th[s], nt * sizeof(unsigned short)));
        pj.src[nj] = P(h, L2.p_w); pj.gamma[nj] = nullptr; pj.tr[nj] = 1; pj.dst[nj] = w->w2th[s]; pj.K[nj] = L2.cout; pj.C[nj] = L2.cin; ++nj;
      }
      if (h->cfg.backbone == 0 && L2.cin == 64 && L2.cout == 128) {   // pass B2's sparse part on the matrix pipe (train_bwd_b2: SPM)
        if (!w->w3th[s]) HIP_TRY(h, hipMalloc(&w->w3th[s], (size_t)L.cin * L.cout * sizeof(unsigned short)));
        pj.src[nj] = P(h, L.p_w); pj.gamma[nj] = nullptr; pj.dst[nj] = w->w3th[s]; pj.K[nj] = -L.cin; pj.C[nj] = L.cout; ++nj;
      }
    }
    if (nj > kPackBf16Jobs) return fail(h, "pack_all_weights: job table overflow");
    if (nj && start) { start->pj = pj; start->nbf16 = nj; }
    else if (nj) hipLaunchKernelGGL(pack_bf16_jobs_kernel, dim3(128, nj), dim3(256), 0, h->stream, pj);   // (128 blocks per image: four elements per thread; with 32 a thread walked sixteen dependent-latency trips)
  }
  h->folded = false;   // eval-mode scale/shift are rebuilt lazily by the next eval forward
  return 0;
}

static size_t lds_train(int ld0, int ldb_or_ld1) { return ((size_t)kTT * 4 + (size_t)kTT * (ld0 + ldb_or_ld1)) * sizeof(float); }

static int set_lds_attrs(alignnet_handle* h)
{
  static PerDeviceOnce done;
  if (!done.need(h->cfg.device)) return 0;
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, true, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b1<>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b1<0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b1<64, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b1_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<2, false, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, false, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase3_wide<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase3_wide<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase3_wide<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase3_wide<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase3_wide_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<2, true, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, false, true, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_fwd_phase23<3, true, true, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, false, true, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, true, true, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, true, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_bwd_b2<false, false, false, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_fwd<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_fwd<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_fwd<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_fwd<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<32, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<32, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<32, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<64, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge<64, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge_dense<32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge_dense<32, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge_dense<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dg_train_bwd_edge_dense<64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  done.mark(h->cfg.device);
  return 0;
}

// ---------------------------------------------------------------------------------
// general-depth PointNet stage (kernels_train_generic.h): forward with batch statistics, then backward
// ---------------------------------------------------------------------------------
// nl: layers to run (hybrid stages stop in front of the last one and do not pool: backbone_fwd_train takes over)
static int backbone_fwd_generic(alignnet_handle* h, int s, const float* p1, const float* p2, int B, float bn_decay, int update_ema, int nl = -1)
{
  TrainWS* w = tws(h);
  StageWS& S = w->st[s];
  TrainWS::GenStage& Gs = w->gen[s];
  const Stack& st = conv_of(h, s);
  const int N = h->cfg.num_points, M = B * N, tiles = (M + kGenTile - 1) / kGenTile;   // of THIS call's batch (gen_part is sized for the workspace capacity, w->gen_tiles)
  static PerDeviceOnce attr;
  if (attr.need(h->cfg.device)) {
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(gen_gemm_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(gen_gemm_dw), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(gen_gemm_dx), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr.mark(h->cfg.device);
  }
  // dgcnn (models/tp8.py:30-46): the layers in front of the last one are edge convs over the B N k edge rows, the last one is the point
  // conv over the B N rows of max-over-k features
  const bool dg = h->cfg.backbone == 1;
  const int ME = dg ? M * kDgK : M, tiles_e = (ME + kGenTile - 1) / kGenTile;
  if (dg) hipLaunchKernelGGL(gen_edge_kernel, dim3((unsigned)(((size_t)2 * ME + 255) / 256)), dim3(256), 0, h->stream, p1, p2, S.xform, w->nn, B, N, kDgK, Gs.X0);
  else hipLaunchKernelGGL(gen_xform_kernel, dim3((unsigned)(((size_t)2 * M + 255) / 256)), dim3(256), 0, h->stream, p1, p2, S.xform, B, N, Gs.X0);
  const bool partial = nl >= 0;
  if (!partial) nl = st.n;
  for (int l = 0; l < nl; ++l) {
    const Layer& L = h->layers[st.first + l];
    const bool edge = dg && l < st.n - 1;
    const int rows = edge ? ME : M, lt = edge ? tiles_e : tiles;
    if (l == 0) {
      GenL1Args a{Gs.X0, P(h, L.p_w), P(h, L.p_b), Gs.Z[0], w->gen_part, rows, L.cout, lt};
      if (dg) hipLaunchKernelGGL(gen_layer1e_fwd, dim3(lt, 2), dim3(256), 0, h->stream, a);
      else hipLaunchKernelGGL(gen_layer1_fwd, dim3(lt, 2), dim3(256), 0, h->stream, a);
    } else if (dg && !edge) {
      // max over the k neighbours of relu(bn(Z_{n-2})) (utils/tf_util_dgcnn.py: reduce_max over the k axis) -> P [2 B N][K] and the slot
      // it came from; the point conv reads P through identity scale / shift (P >= 0: the relu of gen_gemm_fwd's staging is a no-op)
      GenPoolArgs pk{Gs.Z[l - 1], Gs.scale[l - 1], Gs.shift[l - 1], M, kDgK, L.cin, Gs.P, (long)M * L.cin, (long)L.cin, Gs.argk};
      hipLaunchKernelGGL(gen_pool_fwd, dim3(2 * M, (L.cin + 63) / 64), dim3(256), 0, h->stream, pk);
      hipLaunchKernelGGL(gen_fill_kernel, dim3((2 * L.cin + 255) / 256), dim3(256), 0, h->stream, Gs.one, (size_t)2 * L.cin, 1.f);
      hipLaunchKernelGGL(gen_fill_kernel, dim3((2 * L.cin + 255) / 256), dim3(256), 0, h->stream, Gs.zero, (size_t)2 * L.cin, 0.f);
      GenGemmArgs a{Gs.P, Gs.one, Gs.zero, h->d_wp + L.off_wp, P(h, L.p_b), Gs.Z[l], w->gen_part, M, L.cin, L.cout, tiles};
      hipLaunchKernelGGL(gen_gemm_fwd, dim3(tiles, 2), dim3(kGenWaves * 64), (size_t)kGenTile * (L.cin + 4) * sizeof(float), h->stream, a);
    } else {
      GenGemmArgs a{Gs.Z[l - 1], Gs.scale[l - 1], Gs.shift[l - 1], h->d_wp + L.off_wp, P(h, L.p_b), Gs.Z[l], w->gen_part, rows, L.cin, L.cout, lt};
      hipLaunchKernelGGL(gen_gemm_fwd, dim3(lt, 2), dim3(kGenWaves * 64), (size_t)kGenTile * (L.cin + 4) * sizeof(float), h->stream, a);
    }
    GenStatArgs f;
    f.part = w->gen_part; f.tiles = lt; f.M = rows; f.C = L.cout;
    for (int t = 0; t < 2; ++t) {
      f.beta[t] = P(h, L.p_bn[t][0]); f.gamma[t] = P(h, L.p_bn[t][1]); f.mov_mean[t] = P(h, L.p_bn[t][2]); f.mov_var[t] = P(h, L.p_bn[t][3]);
    }
    f.bn_decay = bn_decay; f.update_ema = update_ema;
    f.mean = Gs.mean[l]; f.rstd = Gs.rstd[l]; f.scale = Gs.scale[l]; f.shift = Gs.shift[l];
    if (sync_on(h)) {   // this rank's sum -> all ranks' -> squared differences from the global mean -> all ranks' -> finish (global count)
      f.tot = h->sync_buf; f.world = (double)sync_world(h);
      f.mode = 1; hipLaunchKernelGGL(gen_stat_finish, dim3((L.cout + 63) / 64, 2), dim3(1024), 0, h->stream, f);
      if (sync_sum(h, h->sync_buf, (size_t)2 * L.cout, true)) return 1;
      f.mode = 2; hipLaunchKernelGGL(gen_stat_finish, dim3((L.cout + 63) / 64, 2), dim3(1024), 0, h->stream, f);
      if (sync_sum(h, h->sync_buf + (size_t)2 * L.cout, (size_t)2 * L.cout, true)) return 1;
      f.mode = 3;
    }
    hipLaunchKernelGGL(gen_stat_finish, dim3((L.cout + 63) / 64, 2), dim3(1024), 0, h->stream, f);
  }
  if (partial) { HIP_TRY(h, hipGetLastError()); return 0; }
  const int Ll = st.n - 1, Cl = h->layers[st.first + Ll].cout;
  GenPoolArgs pa{Gs.Z[Ll], Gs.scale[Ll], Gs.shift[Ll], B, N, Cl, S.pooled, S.tower_stride, S.row_stride, Gs.idx};
  hipLaunchKernelGGL(gen_pool_fwd, dim3(2 * B, (Cl + 63) / 64), dim3(256), 0, h->stream, pa);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// given_dY: hybrid stages -- the fused tail has produced dY of layer n - 2 (w->dy2, already masked by [h > 0]): start there
static int backbone_bwd_generic(alignnet_handle* h, int s, int B, float* given_dY = nullptr)
{
  TrainWS* w = tws(h);
  StageWS& S = w->st[s];
  TrainWS::GenStage& Gs = w->gen[s];
  const Stack& st = conv_of(h, s);
  const int N = h->cfg.num_points, M = B * N, tiles = (M + kGenTile - 1) / kGenTile;   // of THIS call's batch (gen_part is sized for the workspace capacity, w->gen_tiles)
  const size_t R = (size_t)2 * M;
  const int Ll = st.n - 1, Cl = h->layers[st.first + Ll].cout;
  float* dY = given_dY ? given_dY : w->gen_d[0];
  float* dYprev = given_dY ? w->gen_d[0] : w->gen_d[1];
  if (!given_dY) {
    HIP_TRY(h, hipMemsetAsync(dY, 0, R * Cl * sizeof(float), h->stream));
    hipLaunchKernelGGL(gen_pool_bwd, dim3((unsigned)(((size_t)2 * B * Cl + 255) / 256)), dim3(256), 0, h->stream, S.dP, S.tower_stride, S.row_stride, Gs.idx, B, N, Cl, dY);
  }
  const bool dg = h->cfg.backbone == 1;   // (never with given_dY: dgcnn stages are not hybrid)
  const int ME = dg ? M * kDgK : M, tiles_e = (ME + kGenTile - 1) / kGenTile;
  for (int l = given_dY ? Ll - 1 : Ll; l >= 0; --l) {
    const Layer& L = h->layers[st.first + l];
    const int C = L.cout, K = L.cin;
    const bool edge = dg && l < Ll, pointconv = dg && l == Ll;
    const int rows = edge ? ME : M, lt = edge ? tiles_e : tiles;
    GenBnBwdArgs b{Gs.Z[l], dY, Gs.mean[l], Gs.rstd[l], Gs.scale[l], Gs.shift[l], w->gen_part, w->gen_cA, w->gen_cB, rows, C, lt};
    hipLaunchKernelGGL(gen_bn_bwd_reduce, dim3(lt, 2, (C + 63) / 64), dim3(256), 0, h->stream, b);
    GenBnFinArgs f{w->gen_part, lt, rows, C, {G(h, w, L.p_bn[0][0]), G(h, w, L.p_bn[1][0])}, {G(h, w, L.p_bn[0][1]), G(h, w, L.p_bn[1][1])}, w->gen_cA, w->gen_cB};
    if (sync_on(h)) {   // the coefficients of dZ = k (g - dbeta / M - zhat dgamma / M) from all ranks' totals; the gradient keeps this rank's sums
      f.mode = 1; f.tot = h->sync_buf; f.world = (double)sync_world(h);
      hipLaunchKernelGGL(gen_bn_bwd_finish, dim3((C + 63) / 64, 2), dim3(1024), 0, h->stream, f);
      if (sync_sum(h, h->sync_buf, (size_t)2 * C * 2, true)) return 1;
      f.mode = 2;
    }
    hipLaunchKernelGGL(gen_bn_bwd_finish, dim3((C + 63) / 64, 2), dim3(1024), 0, h->stream, f);
    hipLaunchKernelGGL(gen_bn_bwd_apply, dim3(lt, 2, (C + 63) / 64), dim3(256), 0, h->stream, b);   // dY is dZ_l from here on
    if (l == 0) {
      GenL1BwdArgs a{Gs.X0, dY, P(h, L.p_w), B, dg ? N * kDgK : N, C, w->gen_ppart, S.gx, S.grot};
      if (dg) hipLaunchKernelGGL(gen_layer1e_bwd, dim3(2 * B), dim3(256), 0, h->stream, a);
      else hipLaunchKernelGGL(gen_layer1_bwd, dim3(2 * B), dim3(256), 0, h->stream, a);
      launch_reduce<float>(h, w->gen_ppart, 2 * B, (long)(dg ? 6 : 3) * C, G(h, w, L.p_w), 1);   // (32 slice groups per column block: one thread per element walking all partials was 0.12 ms per layer)
      break;
    }
    const int slab_rows = gen_slab_rows(C), lslabs = (rows + slab_rows - 1) / slab_rows;   // <= slabs (the buffer's extent, 1024-row slabs)
    GenDwArgs dw{pointconv ? Gs.P : Gs.Z[l - 1], pointconv ? Gs.one : Gs.scale[l - 1], pointconv ? Gs.zero : Gs.shift[l - 1], dY, w->gen_dwpart, rows, K, C, lslabs, slab_rows};
    hipLaunchKernelGGL(gen_gemm_dw, dim3(lslabs, 2, (C + 63) / 64), dim3(kGenWaves * 64), ((size_t)kGenTile * K + kGenTile * 64) * sizeof(float), h->stream, dw);
    launch_reduce<float>(h, w->gen_dwpart, 2 * lslabs, (long)K * C, G(h, w, L.p_w), 1);
    hipLaunchKernelGGL(gen_pack_transposed, dim3(64), dim3(256), 0, h->stream, P(h, L.p_w), K, C, w->gen_wt);
    GenDxArgs dx{dY, w->gen_wt, dYprev, rows, K, C};
    hipLaunchKernelGGL(gen_gemm_dx, dim3(lt, 2), dim3(kGenWaves * 64), (size_t)kGenTile * 132 * sizeof(float), h->stream, dx);
    if (pointconv) {
      // dYprev is dP [2 B N][K]: scatter it to the edge row each maximum came from (dZ_L in dY is spent: the zeroed edge gradient takes
      // its place, so dY stays the current gradient and there is no swap)
      HIP_TRY(h, hipMemsetAsync(dY, 0, (size_t)2 * ME * K * sizeof(float), h->stream));
      hipLaunchKernelGGL(gen_pool_bwd, dim3((unsigned)(((size_t)2 * M * K + 255) / 256)), dim3(256), 0, h->stream, dYprev, (long)M * K, (long)K, Gs.argk, M, kDgK, K, dY);
      continue;
    }
    if (dY == given_dY) { dY = dYprev; dYprev = w->gen_d[1]; }   // (the tail's buffer is only as wide as its own layer: not a scratch for the others)
    else std::swap(dY, dYprev);
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// backbone forward (training mode) for stage s
// ---------------------------------------------------------------------------------
static int backbone_fwd_train(alignnet_handle* h, int s, const float* p1, const float* p2, int B, float bn_decay, int update_ema)
{
  const bool hyb = stage_hybrid(h, s);
  if (stage_generic(h, s) && !hyb) return backbone_fwd_generic(h, s, p1, p2, B, bn_decay, update_ema);
  TrainWS* w = tws(h);
  StageWS& S = w->st[s];
  const Stack& st = conv_of(h, s);
  // (hybrid stages: "layer 2" / "layer 3" of the fused tail are the stage's last two layers; C1 only sizes LDS regions the tail does not use)
  const Layer* L[3] = {&h->layers[st.first], &h->layers[st.first + st.n - 2], &h->layers[st.first + st.n - 1]};
  const int N = h->cfg.num_points, C1 = L[0]->cout, C2 = L[1]->cout, C3 = L[2]->cout;
  TrainFwdArgs a;
  a.pcs[0] = p1; a.pcs[1] = p2; a.xform = S.xform; a.B = B; a.N = N; a.C1 = C1; a.C2 = C2; a.C3 = C3;
  a.ld[0] = ((C1 + 7) & ~7) + 4; a.ld[1] = ((C2 + 7) & ~7) + 4;
  a.w1 = P(h, L[0]->p_w); a.wp2 = h->d_wp + L[1]->off_wp; a.wp3 = h->d_wp + L[2]->off_wp;
  a.b1 = P(h, L[0]->p_b); a.b2 = P(h, L[1]->p_b); a.b3 = P(h, L[2]->p_b);
  a.sc1 = S.scale[0]; a.sh1 = S.shift[0]; a.sc2 = S.scale[1]; a.sh2 = S.shift[1]; a.sgn3 = S.sgn3;
  a.stat_part = w->stat_part; a.ext = S.ext; a.idx = S.idx2; a.gram_part = w->gram_part; a.colsum_part = w->colsum_part;
  a.h2_store = S.h2; a.wp3h = nullptr;
  a.gram_inline = ((C2 + 31) / 32) * ((C2 + 31) / 32 + 1) / 2 > 3 * kTW;   // never for C2 <= 128
  a.dbg = h->ablate_dbg;
  a.stamps = (a.dbg & 32) ? reinterpret_cast<long long*>(w->loss_scratch) + 32 : nullptr;
  const double count = (double)B * N;
  const bool dg = h->cfg.backbone == 1;
  const bool std_w = C1 == 64 && C2 == 128;   // every shipped config: instantiations with compile-time widths
  // fp32 phase 3 of the shipped widths: 128-point tiles (kernels_train_fwd_wide.h), with the Gram of the hidden features accumulated in the same pass
  const bool wide = std_w && C3 <= 32 * kWW * kWSlots && !h->p3_tile64;
  const bool wide_gram = wide && !(h->ab & AB_P3_NOGRAM);
  const bool sync = sync_on(h);
  double W = sync ? (double)sync_world(h) : 1.0;   // sync_bn: batch counts are the global batch's
#ifdef ALIGNNET_ABLATE
  if (h->ablate_mutation == 3 && s == 1) W = 1.0;   // (mutation: stage 2 divides all ranks' sums by this rank's count)
#endif
  // global_part: the partials are already sums over all ranks (statistics derived from all-reduced Gram / column sums)
  auto fin_args = [&](int l, int C, int slices, double cnt, int nb = -1) -> StatFinishArgs {
    StatFinishArgs f;
    f.part = w->stat_part; f.B = nb < 0 ? B : nb; f.C = C; f.slices = slices; f.count = cnt * W; f.bias = P(h, L[l]->p_b);
    for (int t = 0; t < 2; ++t) {
      f.beta[t] = P(h, L[l]->p_bn[t][0]); f.gamma[t] = P(h, L[l]->p_bn[t][1]);
      f.mov_mean[t] = P(h, L[l]->p_bn[t][2]); f.mov_var[t] = P(h, L[l]->p_bn[t][3]);
    }
    f.bn_decay = bn_decay; f.update_ema = update_ema;
    f.mean = S.mean[l]; f.var = S.var[l]; f.scale = S.scale[l]; f.shift = S.shift[l];
    f.sgn = nullptr; f.next_gamma[0] = f.next_gamma[1] = nullptr;
    if (l == 1) { f.sgn = S.sgn3; f.next_gamma[0] = P(h, L[2]->p_bn[0][1]); f.next_gamma[1] = P(h, L[2]->p_bn[1][1]); f.next_C = C3; }
    f.rstd = S.rstd[l]; f.k = S.kk[l];
    return f;
  };
  auto finish = [&](int l, int C, int slices, double cnt, int nb = -1, bool global_part = false) -> int {
    StatFinishArgs f = fin_args(l, C, slices, cnt, nb);
    const dim3 grid((C + kSfC - 1) / kSfC, 2);
    if (sync && !global_part) {   // this rank's (sum, sum of squares) -> all ranks' -> finish
      f.totals_out = h->sync_buf;
      hipLaunchKernelGGL(stat_finish_kernel, grid, dim3(1024), 0, h->stream, f);
      if (sync_sum(h, h->sync_buf, (size_t)2 * C * 2, true)) return 1;
      f.totals_out = nullptr; f.part = h->sync_buf; f.B = 1; f.slices = 1;
    }
    hipLaunchKernelGGL(stat_finish_kernel, grid, dim3(1024), 0, h->stream, f);
    return 0;
  };
  // sync_bn: a reduced (Gram | column sums) pair is added over the ranks in fp64 and its float copy (what the backward reads) rewritten from the sum
  // (d64 = [na | nb] back to back; the float copies fa, fb live wherever the carve put them)
  auto sync_gram = [&](double* d64, float* fa, size_t na, float* fb, size_t nb) -> int {
    if (sync_sum(h, d64, na + nb, true)) return 1;
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, h->stream, d64, fa, na);
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, h->stream, d64 + na, fb, nb);
    return 0;
  };
  // the reductions of the Gram / column-sum partials of h2 over the clouds (the last layer's statistics follow from them: stat3 below)
  auto finish_and_reduce = [&](ReduceJob ja, ReduceJob jb) -> int {
    ja.out = S.gram2raw; ja.upper_c = C2;   // (only the upper 32 x 32 blocks of the per-cloud Grams are valid -- and read)
    ja.out64 = S.gram2raw64; jb.out64 = S.s264;   // the statistics read the unrounded totals
    launch_reduce_multi(h, 2, ja, jb);
    if (sync && sync_gram(S.gram2raw64, S.gram2raw, (size_t)2 * C2 * C2, S.s2, (size_t)2 * C2)) return 1;
    return 0;
  };
  // phase 3 over p3 workgroups per cloud: the kernels write per-workgroup extremes to S.extp / S.idxp, folded into S.ext / S.idx2 right behind them
  auto p3_split = [&](int slots, int tile) -> int {
    const int p3 = p3_parts(h, B, slots, tile);
    a.parts3 = p3;
    if (p3 > 1) { a.ext = S.extp; a.idx = S.idxp; }
    return p3;
  };
  auto p3_merge = [&](int p3) {
    if (p3 <= 1) return;
    const size_t n = (size_t)2 * B * 2 * C3;
    hipLaunchKernelGGL(merge_ext_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, S.extp, S.idxp, p3, C3, n, S.ext, S.idx2);
  };
  int pool_parts = 1;   // dg_pool_finish: workgroups per cloud (its column-sum slices)
  if (dg) {
    // edge part (kernels_train_dgcnn.h): statistics over the B*N*k edge rows, then p = max_k h2 -> S.h2, arg-k -> S.argk
    const double ecount = count * kDgK;
    DgTrainArgs d;
    d.pcs[0] = p1; d.pcs[1] = p2; d.xform = S.xform; d.nn = w->nn; d.B = B; d.N = N; d.k = kDgK; d.C1 = C1; d.C2 = C2; d.ld0 = a.ld[0];
    d.w1 = a.w1; d.b1 = a.b1; d.wp2 = a.wp2; d.b2 = a.b2; d.sc1 = a.sc1; d.sh1 = a.sh1; d.sc2 = a.sc2; d.sh2 = a.sh2;
    d.mom = S.mom; d.stat_part = w->stat_part; d.p_store = S.h2; d.argk = S.argk; d.colsum_part = w->colsum_part; d.s1_part = w->s1_part; d.g1_part = S.g1_part;
    d.wp2h = h->train_bf16 ? w->wp2h[s] : nullptr;
    d.parts = dg_parts(h, B, 2);
    const int nG1 = ((C1 + 31) / 32) * ((C1 + 31) / 32 + 1) / 2;   // upper 32 x 32 blocks of Gram(h1), one fp64 slab [16][64] each behind the tiles
    const size_t dlds = (h->train_bf16 ? (size_t)2 * kTT * 8 * sizeof(float) + 2 * ((size_t)kTT * (C1 + 8) + (size_t)C1 * (kTT + 8)) * sizeof(unsigned short)
                                       : ((size_t)2 * kTT * 8 + 2 * (size_t)kTT * d.ld0) * sizeof(float))   // two edge-feature buffers | two lift buffers
                        + (size_t)nG1 * 16 * 64 * sizeof(double);
    hipLaunchKernelGGL(dg_train_phase1, dim3(2 * B), dim3(kP1T), 0, h->stream, d);
    if (finish(0, C1, 1, ecount)) return 1;
    for (int t = 0; t < 2; ++t) d.gamma2[t] = P(h, L[1]->p_bn[t][1]);
    d.stamps = (a.dbg & 32) ? reinterpret_cast<long long*>(w->loss_scratch) + 48 : nullptr;
    d.dbg = a.dbg;
    if (d.stamps) hipMemsetAsync(d.stamps + 8, 0, 3 * sizeof(long long), h->stream);
  { ProfScope prof_scope(h, PK_DG_FWD, true);
    if (h->train_bf16 && C1 == 64) TIMED_LAUNCH((dg_train_fwd<64, true>), dim3(2 * B * d.parts), dim3(kTW * 64), dlds, d);
    else if (h->train_bf16) TIMED_LAUNCH((dg_train_fwd<32, true>), dim3(2 * B * d.parts), dim3(kTW * 64), dlds, d);
    else if (C1 == 64) TIMED_LAUNCH(dg_train_fwd<64>, dim3(2 * B * d.parts), dim3(kTW * 64), dlds, d);
    else TIMED_LAUNCH(dg_train_fwd<32>, dim3(2 * B * d.parts), dim3(kTW * 64), dlds, d);
  }
    if (d.stamps) {
      long long sv[11];
      hipStreamSynchronize(h->stream);
      hipMemcpy(sv, d.stamps, sizeof(sv), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "FE stage %d whole-kernel cycles per workgroup: longest %lld shortest %lld workgroup 0 %lld (%d slots)\n", s, sv[8], (1ll << 40) - sv[9], sv[10],
                   ((N + kTT - 1) / kTT) * kDgK);
      std::fprintf(stderr, "FE stage %d it-25 cycles: gather + mfma %lld epilogue %lld colsum %lld es %lld barrier %lld lift %lld barrier %lld\n", s,
                   sv[1] - sv[0], sv[2] - sv[1], sv[3] - sv[2], sv[4] - sv[3], sv[5] - sv[4], sv[6] - sv[5], sv[7] - sv[6]);
    }
    {   // statistics of z2 from s1 = sum h1 and G1 = sum h1^T h1 over the edge rows (both kept for the backward)
      const int sGe = 1024 / C1;   // row groups of dg_train_fwd's column sums
      ReduceJob jg = rjob(S.g1_part, B * d.parts, (long)(C1 * C1), S.g1f), js = rjob(w->s1_part, B * d.parts * sGe, (long)(C1), S.s1e);   // (one slice per workgroup)
      jg.out64 = S.g1f64; js.out64 = S.s1e64;
      launch_reduce_multi(h, 2, jg, js);
      if (sync && sync_gram(S.g1f64, S.g1f, (size_t)2 * C1 * C1, S.s1e, (size_t)2 * C1)) return 1;
      hipLaunchKernelGGL(stat2_from_gram_kernel, dim3(C2, 2), dim3(256), 0, h->stream, S.g1f64, S.s1e64, P(h, L[1]->p_w), a.b2, C1, C2, ecount * W,
                         h->train_bf16 ? 1 : 0, w->stat_part);   // bf16 mode: Gram and sums are those of the rounded h1, W2 is rounded here
    }
    if (finish(1, C2, 1, ecount, 1, true)) return 1;
    // (a stream over the cloud's rows: four workgroups per CU fill the chip; 2B * parts slices of 2 C2 doubles fit the max(2B, 512) x 1024 carve up to C2 = 256)
    pool_parts = std::max(1, std::min(std::min(8, 1024 / (2 * B)), N / 64));
    if ((size_t)2 * B * pool_parts * 2 * C2 > std::max((size_t)2 * B, (size_t)kDgPartSlices) * 1024) pool_parts = 1;
    hipLaunchKernelGGL(dg_pool_finish, dim3(2 * B * pool_parts), dim3(256), 0, h->stream, S.h2, B, N, C2, S.scale[1], S.shift[1], w->colsum_part, pool_parts);
  } else if (hyb) {
    // layers 1 .. n - 1 layer by layer, then h = relu(bn(Z_{n-1})) once, with its column sums; sign(gamma) of the last layer
    if (backbone_fwd_generic(h, s, p1, p2, B, bn_decay, update_ema, st.n - 1)) return 1;
    const TrainWS::GenStage& Gs = w->gen[s];
    const int c4 = C2 / 4;
    hipLaunchKernelGGL(gen_apply_colsum, dim3(2 * B), dim3(c4 * (256 / c4)), 0, h->stream, Gs.Z[st.n - 2], Gs.scale[st.n - 2], Gs.shift[st.n - 2], B, N, C2, S.h2,
                       w->colsum_part);
    for (int t = 0; t < 2; ++t)
      hipLaunchKernelGGL(sign_kernel, dim3((C3 + 255) / 256), dim3(256), 0, h->stream, P(h, L[2]->p_bn[t][1]), C3, S.sgn3 + (size_t)t * C3);
  }
  // bf16 operands for the fused tail on given features (its staging loop needs 256 % (C2 / 4) == 0: C2 = 32, 64, 128)
  const bool tail_bf16 = h->train_bf16 && (dg || (hyb && 256 % (C2 / 4) == 0));
  if (dg || hyb) {
    // point conv on the stored features (DGCNN: p = max_k h2; hybrid: the output of the layer-by-layer part)
    if (tail_bf16) {
      // bf16 operands (p rounded while it is staged, the sign-folded bf16 image of W3); Gram(p) and the column sums of the rounded p
      // come out of the same pass (1024 / C2 row-group slices per cloud)
      a.wp3h = w->wp3h[s];
      const size_t ldsh = ((size_t)kTT * 4 + (size_t)kTT * a.ld[0]) * sizeof(float) +
                          ((size_t)kTT * (C2 + 8) + (size_t)C2 * (kTT + 8)) * sizeof(unsigned short);
      const int p3 = p3_split(512, kTT);
      { ProfScope prof_scope(h, PK_TRAIN_PHASE3, true);
      if (std_w) TIMED_LAUNCH((train_fwd_phase23<3, true, true, 64, 128>), dim3(2 * B * p3), dim3(kTW * 64), ldsh, a);
      else TIMED_LAUNCH((train_fwd_phase23<3, true, true>), dim3(2 * B * p3), dim3(kTW * 64), ldsh, a);
      }
      p3_merge(p3);
      if (finish_and_reduce(rjob(w->gram_part, B * p3, (long)(C2 * C2), S.gram2), rjob(w->colsum_part, (1024 / C2) * B * p3, (long)(C2), S.s2))) return 1;
    } else {
    const int p3 = wide_gram ? p3_split(256, kWT) : 1;
  { ProfScope prof_scope(h, PK_TRAIN_PHASE3, true);
    if (wide_gram) TIMED_LAUNCH((train_fwd_phase3_wide<true, true>), dim3(2 * B * p3), dim3(kWW * 64), lds_p3_wide_f32(), a);
    else if (wide) TIMED_LAUNCH((train_fwd_phase3_wide<true, false>), dim3(2 * B), dim3(kWW * 64), lds_p3_wide_f32(), a);
    else if (std_w) TIMED_LAUNCH((train_fwd_phase23<3, false, true, 64, 128>), dim3(2 * B), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
    else TIMED_LAUNCH((train_fwd_phase23<3, false, true>), dim3(2 * B), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  }
    p3_merge(p3);
    #ifdef ALIGNNET_ABLATE
    if (wide_gram && p3 == 1 && getenv("ALIGNNET_P3_EXTCHECK")) {   // debug: ext / idx of the GRAM variant against the plain one
      std::vector<float> ea((size_t)2 * B * 2 * C3), eb(ea.size());
      std::vector<int> ia(ea.size()), ib(ea.size());
      hipStreamSynchronize(h->stream);
      hipMemcpy(ea.data(), S.ext, ea.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ia.data(), S.idx2, ia.size() * 4, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL((train_fwd_phase3_wide<true, false>), dim3(2 * B), dim3(kWW * 64), lds_p3_wide_f32(), h->stream, a);
      hipStreamSynchronize(h->stream);
      hipMemcpy(eb.data(), S.ext, eb.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ib.data(), S.idx2, ib.size() * 4, hipMemcpyDeviceToHost);
      size_t nde = 0, ndi = 0, first = (size_t)-1;
      for (size_t i = 0; i < ea.size(); ++i) { if (ea[i] != eb[i]) { ++nde; if (first == (size_t)-1) first = i; } if (ia[i] != ib[i]) ++ndi; }
      std::fprintf(stderr, "EXTCHECK stage %d: %zu of %zu ext differ, %zu idx differ; first at %zu (cloud %zu half %zu col %zu): %g vs %g\n", s, nde, ea.size(), ndi, first,
                   first == (size_t)-1 ? 0 : first / (2 * C3), first == (size_t)-1 ? 0 : (first / C3) & 1, first == (size_t)-1 ? 0 : first % C3,
                   first == (size_t)-1 ? 0.f : ea[first], first == (size_t)-1 ? 0.f : eb[first]);
    }
    if (wide_gram && p3 == 1 && getenv("ALIGNNET_P3_GRAMCHECK")) {   // debug: the in-kernel Gram against gram_h2_kernel's, block by block
      std::vector<float> ga((size_t)2 * B * C2 * C2), gb(ga.size());
      hipStreamSynchronize(h->stream);
      hipMemcpy(ga.data(), w->gram_part, ga.size() * 4, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(gram_h2_kernel<128>, dim3(2 * B), dim3(kTW * 64), (size_t)2 * kTT * (C2 + 4) * sizeof(float), h->stream, S.h2, N, C2, w->gram_part);
      hipStreamSynchronize(h->stream);
      hipMemcpy(gb.data(), w->gram_part, gb.size() * 4, hipMemcpyDeviceToHost);
      for (int it = 0; it < 4; ++it)
        for (int jt = it; jt < 4; ++jt) {
          double worst = 0, mag = 0; int wc = -1;
          for (int c = 0; c < 2 * B; ++c)
            for (int i = 0; i < 32; ++i)
              for (int j = 0; j < 32; ++j) {
                const size_t o = (size_t)c * C2 * C2 + (size_t)(it * 32 + i) * C2 + jt * 32 + j;
                const double e = std::fabs((double)ga[o] - gb[o]);
                if (e > worst) { worst = e; wc = c; }
                mag = std::max(mag, (double)std::fabs(gb[o]));
              }
          std::fprintf(stderr, "GRAMCHECK stage %d block (%d,%d): worst abs diff %.3e at cloud %d (max |G| %.3e)\n", s, it, jt, worst, wc, mag);
        }
    }
#endif   // ALIGNNET_ABLATE
    if (!wide_gram) { ProfScope prof_scope(h, PK_TRAIN_GRAM, true);
    TIMED_LAUNCH(C2 == 128 ? gram_h2_kernel<128> : gram_h2_kernel<0>, dim3(2 * B), dim3(kTW * 64), (size_t)2 * kTT * (C2 + 4) * sizeof(float), S.h2, N, C2, w->gram_part);
    }
    if (finish_and_reduce(rjob(w->gram_part, B * p3, (long)(C2 * C2), S.gram2), rjob(w->colsum_part, (dg ? 2 * pool_parts : 1) * B, (long)(C2), S.s2))) return 1;   // (column sums: the producer's)
    }
  } else {
  // phase 1 from the cloud's moments of x' (kept in S.mom for the first-layer backward)
  if (!w->moments_done[s])
    hipLaunchKernelGGL(pn_moments_kernel, dim3(2 * B), dim3(256), 0, h->stream, p1, p2, S.xform, B, N, S.mom, a.w1, a.b1, C1, w->stat_part);
  w->moments_done[s] = false;
  if (finish(0, C1, 1, count)) return 1;
  a.wp2h = h->train_bf16 ? w->wp2h[s] : nullptr;
  const int CT1f = (C1 + 31) / 32, CT2f = (C2 + 31) / 32;
  // fp32 only: the bf16 phase 2 is light on the matrix pipe (49 us) and a fp32 Gram there costs more than passes B1 / B2 save
  const bool fwd_gram = !h->train_bf16 && CT1f * CT2f + CT1f * (CT1f + 1) / 2 <= 3 * kTW && !(h->ab & AB_PHASE2_LEGACY);   // cf. acc_in_b1 below
  if (fwd_gram) {
    // phase 2 from s1 = sum h1 and G1 = sum h1^T h1 (kept for the backward)
    Gram1Args g;
    g.pcs[0] = p1; g.pcs[1] = p2; g.xform = S.xform; g.B = B; g.N = N; g.C1 = C1; g.ld0 = a.ld[0];
    g.w1 = a.w1; g.sc1 = a.sc1; g.sh1 = a.sh1; g.g1_part = S.g1_part; g.s1_part = w->s1_part;
    const int pp = pn_parts(h, B);
    g.parts = pp;
    const size_t glds = ((size_t)kTT * 4 + (size_t)kTT * g.ld0) * sizeof(float);
    if (h->train_bf16) hipLaunchKernelGGL(train_fwd_gram1<true>, dim3(2 * B * pp), dim3(kTW * 64), glds, h->stream, g);
    else if (C1 == 64) hipLaunchKernelGGL((train_fwd_gram1<false, 64>), dim3(2 * B * pp), dim3(kTW * 64), glds, h->stream, g);
    else hipLaunchKernelGGL(train_fwd_gram1<false>, dim3(2 * B * pp), dim3(kTW * 64), glds, h->stream, g);
    const int sG1 = std::max(1, 256 / C1);
    ReduceJob jg = rjob(S.g1_part, B * pp, (long)(C1 * C1), S.g1f), js = rjob(w->s1_part, B * pp * sG1, (long)(C1), S.s1e);
    jg.out64 = S.g1f64; js.out64 = S.s1e64;
    launch_reduce_multi(h, 2, jg, js);
    if (sync && sync_gram(S.g1f64, S.g1f, (size_t)2 * C1 * C1, S.s1e, (size_t)2 * C1)) return 1;
    // (statistics from the Gram and their finish in one launch; ab_phase2_legacy keeps the pass, ab_no_glue_fold the two launches)
    if (!(h->ab & AB_NO_GLUE_FOLD))
      hipLaunchKernelGGL(stat2_from_gram_finish_kernel, dim3(C2, 2), dim3(256), 0, h->stream, S.g1f64, S.s1e64, P(h, L[1]->p_w), a.b2, C1, C2, count * W,
                         h->train_bf16 ? 1 : 0, fin_args(1, C2, 1, count, 1));
    else {
    hipLaunchKernelGGL(stat2_from_gram_kernel, dim3(C2, 2), dim3(256), 0, h->stream, S.g1f64, S.s1e64, P(h, L[1]->p_w), a.b2, C1, C2, count * W,
                       h->train_bf16 ? 1 : 0, w->stat_part);
    if (finish(1, C2, 1, count, 1, true)) return 1;
    }
  } else {
  const int pp = pn_parts(h, B);
  a.parts = pp;
  { ProfScope prof_scope(h, PK_TRAIN_PHASE2, true);
  if (h->train_bf16 && std_w) TIMED_LAUNCH((train_fwd_phase23<2, true, false, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  else if (h->train_bf16) TIMED_LAUNCH((train_fwd_phase23<2, true>), dim3(2 * B * pp), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  else if (std_w) TIMED_LAUNCH((train_fwd_phase23<2, false, false, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  else TIMED_LAUNCH(train_fwd_phase23<2>, dim3(2 * B * pp), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  }
  if (finish(1, C2, 4 * pp, count)) return 1;
  a.parts = 1;
  }
  int p3 = 1;   // workgroups per cloud of phase 3 (the 128-point-tile kernels with the Gram in the same pass)
  if (h->train_bf16) {
    a.wp3h = w->wp3h[s];
    const size_t ldsh = ((size_t)kTT * 4 + (size_t)kTT * a.ld[0]) * sizeof(float) +
                        ((size_t)kTT * (C2 + 8) + (size_t)C2 * (kTT + 8)) * sizeof(unsigned short);
    const bool wide_bf16 = std_w && C3 <= 32 * kWW * kWSlots && !h->p3_tile64;
    if (wide_bf16) p3 = p3_split(256, kWT);
  { ProfScope prof_scope(h, PK_TRAIN_PHASE3, true);
    // shipped widths: 128-point tiles, the next tile's prologue under the lift (kernels_train_fwd_wide.h)
    if (wide_bf16)
      TIMED_LAUNCH(train_fwd_phase3_wide_bf16, dim3(2 * B * p3), dim3(kWW * 64), lds_p3_wide_bf16(), a);
    else if (std_w && !(h->ab & AB_P3BF16_GENERIC)) TIMED_LAUNCH((train_fwd_phase23<3, true, false, 64, 128>), dim3(2 * B), dim3(kTW * 64), ldsh, a);
    else TIMED_LAUNCH((train_fwd_phase23<3, true>), dim3(2 * B), dim3(kTW * 64), ldsh, a);
  }
    if (a.stamps) {
      long long sv[8];
      hipStreamSynchronize(h->stream);
      hipMemcpy(sv, a.stamps, sizeof(sv), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "P3 stage %d tile-3 cycles: xform %lld lift %lld layer2 %lld barrier %lld store %lld gram %lld layer3 %lld\n", s, sv[1] - sv[0],
                   sv[2] - sv[1], sv[3] - sv[2], sv[4] - sv[3], sv[5] - sv[4], sv[6] - sv[5], sv[7] - sv[6]);
    }
  } else {
    if (wide_gram) p3 = p3_split(256, kWT);
  { ProfScope prof_scope(h, PK_TRAIN_PHASE3, true);
    // shipped widths: 128-point tiles (a weight fragment of the lift feeds four row tiles; kernels_train_fwd_wide.h)
    if (wide_gram) TIMED_LAUNCH((train_fwd_phase3_wide<false, true>), dim3(2 * B * p3), dim3(kWW * 64), lds_p3_wide_f32(), a);
    else if (wide) TIMED_LAUNCH((train_fwd_phase3_wide<false, false>), dim3(2 * B), dim3(kWW * 64), lds_p3_wide_f32(), a);
    else if (std_w) TIMED_LAUNCH((train_fwd_phase23<3, false, false, 64, 128>), dim3(2 * B), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
    else TIMED_LAUNCH(train_fwd_phase23<3>, dim3(2 * B), dim3(kTW * 64), lds_train(a.ld[0], a.ld[1]), a);
  }
    { ProfScope prof_scope(h, PK_TRAIN_GRAM, true);
    if (!a.gram_inline && !wide_gram)
      TIMED_LAUNCH(C2 == 128 ? gram_h2_kernel<128> : gram_h2_kernel<0>, dim3(2 * B), dim3(kTW * 64), (size_t)2 * kTT * (C2 + 4) * sizeof(float), S.h2, N, C2,
                         w->gram_part);
    }
  }
  p3_merge(p3);
  if (finish_and_reduce(rjob(w->gram_part, B * p3, (long)(C2 * C2), S.gram2), rjob(w->colsum_part, 4 * B * p3, (long)(C2), S.s2))) return 1;
  }
  {   // statistics of the last layer from the Gram, EMA, pooled features, centred Gram + column means for the backward: one launch
    Stat3Args f;
    f.G = S.gram2raw64; f.s = S.s264; f.W = P(h, L[2]->p_w); f.C2 = C2; f.C3 = C3; f.M = count * W;
    f.round_w = (h->train_bf16 && (!(dg || hyb) || tail_bf16)) ? 1 : 0;   // the lift ran on bf16 operands
    for (int t = 0; t < 2; ++t) {
      f.beta[t] = P(h, L[2]->p_bn[t][0]); f.gamma[t] = P(h, L[2]->p_bn[t][1]);
      f.mov_mean[t] = P(h, L[2]->p_bn[t][2]); f.mov_var[t] = P(h, L[2]->p_bn[t][3]);
    }
    f.bn_decay = bn_decay; f.update_ema = update_ema;
    f.mean = S.mean[2]; f.var = S.var[2]; f.scale = S.scale[2]; f.shift = S.shift[2]; f.rstd = S.rstd[2]; f.k = S.kk[2];
    f.Gc = S.gram2; f.m2 = S.m2;
    f.pa = PoolFinishArgs{S.ext, S.idx2, S.sgn3, P(h, L[2]->p_b), S.scale[2], S.shift[2], S.mean[2], S.var[2], B, C3, S.pooled,
                          S.tower_stride, S.row_stride, S.zhat_star, S.idx, 1};
    hipLaunchKernelGGL(stat3_pool_finish_kernel, dim3((C3 + kS3C - 1) / kS3C, 2), dim3(512), stat3_lds_bytes(C2), h->stream, f);
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// head MLP forward / backward (training mode)
// ---------------------------------------------------------------------------------
static BnRowsArgs bn_args(alignnet_handle* h, TrainWS* w, int s, int j, int M, int rows_per_set, float bn_decay, int update_ema,
                          const float* u_dev)
{
  const Stack& fs = fc_of(h, s);
  const Layer& L = h->layers[fs.first + j];
  HeadLayerWS& HL = w->hl[s][j];
  BnRowsArgs a{};
  a.z = HL.z; a.y = HL.y; a.M = M; a.C = L.cout; a.rows_per_set = rows_per_set;
  for (int t = 0; t < 2; ++t) {
    const int src = L.p_bn[t][0] >= 0 ? t : 0;
    a.beta[t] = P(h, L.p_bn[src][0]); a.gamma[t] = P(h, L.p_bn[src][1]);
    a.mov_mean[t] = P(h, L.p_bn[src][2]); a.mov_var[t] = P(h, L.p_bn[src][3]);
  }
  a.bn_decay = bn_decay; a.update_ema = update_ema; a.mean = HL.mean; a.var = HL.var;
  const bool last_hidden = j == fs.n - 2;
  a.keep = last_hidden ? keep_of(h, s) : -1.f;
  // dropout uniforms layout (include/alignnet_hip.h): [s1 t0 | s2 t0 | s1 t1 | s2 t1 | pair], each B x width
  const size_t blk = (size_t)rows_per_set * L.cout;
  if (u_dev && last_hidden) {
    if (s < 2) { a.u = u_dev + (size_t)s * blk; a.u_set_stride = 2 * (long)blk; }
    else { a.u = u_dev + 4 * (size_t)rows_per_set * h->layers[fc_of(h, 0).first + fc_of(h, 0).n - 2].cout; a.u_set_stride = 0; }
  }
  a.seed = h->dropout_seed_base() + s * 4;
  return a;
}

// The uniforms of the device-side dropout stream for the current step counter, in the host layout of `dropout_u`
// (include/alignnet_hip.h).  Evaluates the very expression dropout_scale() evaluates (seed of bn_args + set, element index).
__global__ void dropout_uniforms_kernel(float* __restrict__ out, int B, int w12, int w3, uint64_t seed_base)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t blk = (size_t)B * w12, n = 4 * blk + (size_t)B * w3;
  if (i >= n) return;
  int s, set; size_t e;
  if (i < 4 * blk) { const int q = (int)(i / blk); s = q & 1; set = q >> 1; e = i - (size_t)q * blk; }   // [s1 t0 | s2 t0 | s1 t1 | s2 t1]
  else { s = 2; set = 0; e = i - 4 * blk; }
  out[i] = hash_uniform(seed_base + (uint64_t)s * 4 + set, (uint64_t)e);
}

extern "C" int alignnet_debug_dropout_uniforms(alignnet_handle* h, int32_t B, float* dst, size_t count)
{
  if (!h || !dst) return 1;
  const int w1 = h->layers[h->s1_fc.first + h->s1_fc.n - 2].cout, w2 = h->layers[h->s2_fc.first + h->s2_fc.n - 2].cout;
  const int w3 = h->layers[h->rem_fc.first + h->rem_fc.n - 2].cout;
  if (w1 != w2) return fail(h, "alignnet_debug_dropout_uniforms: needs equal last-hidden widths in the s1/s2 heads");
  const size_t n = (size_t)B * (4 * (size_t)w1 + w3);
  if (B < 1 || count != n) return fail(h, "alignnet_debug_dropout_uniforms: count must be B * (4 * w_hidden + w_pair_hidden)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  float* d = nullptr;
  HIP_TRY(h, hipMalloc(&d, n * sizeof(float)));
  hipLaunchKernelGGL(dropout_uniforms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d, B, w1, w3,
                     h->dropout_seed_base());
  hipError_t e = hipMemcpyAsync(dst, d, n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d);
  HIP_TRY(h, e);
  return 0;
}

static int head_fwd_train(alignnet_handle* h, int s, const float* in, long ldin, int M, int rows_per_set, float bn_decay,
                          int update_ema, const float* u_dev)
{
  TrainWS* w = tws(h);
  const Stack& fs = fc_of(h, s);
  const float* cur = in; long ldc = ldin;
  const int nsets = M > rows_per_set ? 2 : 1;
  for (int j = 0; j < fs.n; ++j) {
    const Layer& L = h->layers[fs.first + j];
    if (j < fs.n - 1) {
      HeadLayerWS& HL = w->hl[s][j];
      // few tiles and a deep K (the pair head's first layer: 128 tiles, K = 2048 = eight dependent chunks per wave, 20 us): the two K halves as
      // two batch entries into (z, dz -- free until the backward), summed with the bias by the statistics pass that reads z anyway
      const bool splitk = !sync_on(h) && L.cin >= 1024 && (L.cin & 63) == 0 && (long)((L.cout + 31) / 32) * ((M + 31) / 32) <= 128;
      if (splitk) launch_gemm(h, cur, ldc, 1, P(h, L.p_w), L.cout, 1, HL.z, L.cout, 1, M, L.cout, L.cin / 2, nullptr, 1.f, 0, 2, L.cin / 2,
                              (long)(L.cin / 2) * L.cout, HL.dz - HL.z);
      else
      launch_gemm(h, cur, ldc, 1, P(h, L.p_w), L.cout, 1, HL.z, L.cout, 1, M, L.cout, L.cin, P(h, L.p_b));
      BnRowsArgs a = bn_args(h, w, s, j, M, rows_per_set, bn_decay, update_ema, u_dev);
      if (splitk) { a.z2 = HL.dz; a.zbias = P(h, L.p_b); a.zw = HL.z; }
      const dim3 bgrid((L.cout + kBnCols - 1) / kBnCols, nsets);
      if (sync_on(h)) {   // this rank's column sums -> all ranks' -> normalise with the global batch's moments
        a.mode = 1; a.totals = h->sync_buf;
        hipLaunchKernelGGL(bn_rows_fwd_kernel, bgrid, dim3(kBnCols * kBnGroups), 0, h->stream, a);
        if (sync_sum(h, h->sync_buf, (size_t)nsets * L.cout * 2, true)) return 1;
        a.mode = 2; a.world = sync_world(h);
      }
      hipLaunchKernelGGL(bn_rows_fwd_kernel, bgrid, dim3(kBnCols * kBnGroups), 0, h->stream, a);
      cur = HL.y; ldc = L.cout;
    } else {
      launch_gemm(h, cur, ldc, 1, P(h, L.p_w), L.cout, 1, w->o[s], L.cout, 1, M, L.cout, L.cin, P(h, L.p_b));
    }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// d_out = w->d_o[s]  ->  parameter gradients + gradient wrt the head input (din, same layout as the input)
static int head_bwd_train(alignnet_handle* h, int s, const float* in, long ldin, float* din, int M, int rows_per_set, const float* u_dev)
{
  TrainWS* w = tws(h);
  const Stack& fs = fc_of(h, s);
  const int nsets = M > rows_per_set ? 2 : 1;
  const float* dcur = w->d_o[s];
  for (int j = fs.n - 1; j >= 0; --j) {
    const Layer& L = h->layers[fs.first + j];
    const float* xin = j == 0 ? in : w->hl[s][j - 1].y;
    const long ldx = j == 0 ? ldin : L.cin;
    if (j < fs.n - 1) {
      // dcur is d(y_j): through dropout / relu / BN
      HeadLayerWS& HL = w->hl[s][j];
      BnRowsBwdArgs b{};
      b.f = bn_args(h, w, s, j, M, rows_per_set, 0.f, 0, u_dev);
      b.dy = dcur; b.dz = HL.dz;
      for (int t = 0; t < 2; ++t) {
        const int src = L.p_bn[t][0] >= 0 ? t : 0;
        b.dbeta[t] = G(h, w, L.p_bn[src][0]); b.dgamma[t] = G(h, w, L.p_bn[src][1]);
      }
      const dim3 bgrid((L.cout + kBnCols - 1) / kBnCols, nsets);
      if (sync_on(h)) {
        b.f.mode = 1; b.f.totals = h->sync_buf;
        hipLaunchKernelGGL(bn_rows_bwd_kernel, bgrid, dim3(kBnCols * kBnGroups), 0, h->stream, b);
        if (sync_sum(h, h->sync_buf, (size_t)nsets * L.cout * 2, true)) return 1;
        b.f.mode = 2; b.f.world = sync_world(h);
      }
      hipLaunchKernelGGL(bn_rows_bwd_kernel, bgrid, dim3(kBnCols * kBnGroups), 0, h->stream, b);
      dcur = HL.dz;
      // bias feeds a BatchNorm: its gradient is identically zero (TF computes rounding noise here); the whole gradient
      // vector is zeroed once per step, so nothing to do
    } else {
      def_reduce(h, w, rjob(dcur, M, (long)L.cout, G(h, w, L.p_b), 1.f, 1));   // bias of the last (linear) layer
    }
    // dx = dz W^T (NT) is on the backward's critical path; dW = x^T dz (TN) waits for nothing but the optimiser (deferred)
    float* dx = j == 0 ? din : w->hl[s][j - 1].dyb;
    def_gemm(h, w, gemm_args(xin, 1, ldx, dcur, L.cout, 1, G(h, w, L.p_w), L.cout, 1, L.cin, L.cout, M));
    launch_gemm(h, dcur, L.cout, 1, P(h, L.p_w), 1, L.cout, dx, j == 0 ? ldin : L.cin, 1, M, L.cin, L.cout);
    dcur = dx;
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// backbone backward for stage s: consumes S.dP, produces parameter gradients + S.gx / S.grot
// ---------------------------------------------------------------------------------
static int backbone_bwd_train(alignnet_handle* h, int s, const float* p1, const float* p2, int B)
{
  const bool hyb = stage_hybrid(h, s);
  tws(h)->glue_folded = false;   // (before the layer-by-layer return: a folded glue of the stage behind this one must not carry over)
  if (stage_generic(h, s) && !hyb) return backbone_bwd_generic(h, s, B);
  TrainWS* w = tws(h);
  StageWS& S = w->st[s];
  const Stack& st = conv_of(h, s);
  // (hybrid stages: the fused tail's "layers 2 and 3" are the stage's last two; the layers in front of them follow layer by layer)
  const Layer* L[3] = {&h->layers[st.first], &h->layers[st.first + st.n - 2], &h->layers[st.first + st.n - 1]};
  const int N = h->cfg.num_points, C1 = L[0]->cout, C2 = L[1]->cout, C3 = L[2]->cout;
  const double M0 = (double)B * N;
  const bool dg = h->cfg.backbone == 1;
  const bool given = dg || hyb;   // pass B2 on stored features
  const bool given_bf16 = given && h->train_bf16 && 256 % (C2 / 4) == 0;   // ... with h2 Q3 on bf16 MFMA (same rule as the forward tail)
  const bool sync = sync_on(h);
  const double Wn = sync ? (double)sync_world(h) : 1.0;
  const double Me = (dg ? M0 * kDgK : M0) * Wn;   // rows behind the statistics of layers 1 and 2 (DGCNN: the B*N*k edge rows); sync_bn: of all ranks
  const double M = M0 * Wn;
  const float* W2 = P(h, L[1]->p_w); const float* W3 = P(h, L[2]->p_w);
  auto g256 = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  auto g256t = [](size_t n) { return dim3((unsigned)((n + 255) / 256), 2); };
  w->glue_folded = false;
  auto set_glue = [&](DgB0Args& z) {   // the glue towards the previous stage rides on dg_b0_cloud (its thread 0 per cloud)
    if ((h->ab & AB_NO_GLUE_FOLD) || s == 0) return;
    const int nb = h->cfg.num_bins;
    z.glue = s == 2 ? 3 : 2; z.xform = S.xform; z.pcls = w->cls; z.nb = nb;
    z.d_s2c = w->d_s2c; z.d_o2 = w->d_o[1]; z.ldo2 = 3 + 2 * nb; z.d_s1c = w->d_s1c; z.d_o0 = w->d_o[0];
    w->glue_folded = true;
  };

  // ---- layer 3 (sparse + Gram identities) ----
  Prep3Args p3;
  p3.dP = S.dP; p3.tower_stride = S.tower_stride; p3.row_stride = S.row_stride; p3.pooled = S.pooled; p3.zhat_star = S.zhat_star;
  for (int t = 0; t < 2; ++t) { p3.gamma[t] = P(h, L[2]->p_bn[t][1]); p3.dbeta[t] = G(h, w, L[2]->p_bn[t][0]); p3.dgamma[t] = G(h, w, L[2]->p_bn[t][1]); }
  p3.var = S.var[2]; p3.B = B; p3.C = C3; p3.M = M; p3.E = S.E3; p3.kdb = S.kdb3; p3.gs = S.gs;
  const size_t qimg = img_floats(C2, C2);
  const bool std_w = C1 == 64 && C2 == 128;   // every shipped config: instantiations with compile-time widths
  const bool b2_accum = ((C1 + 31) / 32) * ((C2 + 31) / 32) + ((C1 + 31) / 32) * ((C1 + 31) / 32 + 1) / 2 > 3 * kTW;   // else pass B1 does it
  const bool spm = !given && std_w && !b2_accum && h->train_bf16;   // sparse rows on the matrix pipe: the X region holds h1 | R^T | S lo instead of an fp32 tile
  const bool q3_bf16 = h->train_bf16 && (!given || given_bf16);   // pass B2 reads only the bf16 images of Q3 in this mode
  if (q3_bf16) { p3.u = w->u3; p3.m2 = S.m2; p3.W = P(h, L[2]->p_w); p3.C2 = C2; }
  if (sync) {
    p3.mode = 1; p3.totals = h->sync_buf;
    hipLaunchKernelGGL(prep3_kernel, dim3((C3 + 31) / 32, 2), dim3(1024), 0, h->stream, p3);
    if (sync_sum(h, h->sync_buf, (size_t)2 * C3 * 2, true)) return 1;
    p3.mode = 2;
  }
  hipLaunchKernelGGL(prep3_kernel, dim3((C3 + 31) / 32, 2), dim3(1024), 0, h->stream, p3);
  // weight gradient of layer 3 (deferred: only the optimiser waits for it):  dW3 = Sp - m2 (k db)^T + (Ghat2 W3) diag(E)
  def_sparse(h, w, SparseDwJob{S.gs, S.idx, S.h2, B, N, C2, C3, S.Sp, (h->train_bf16 && !given) ? 1 : 0});
  {
    GemmArgs g = gemm_args(S.gram2, C2, 1, W3, C3, 1, S.GW, C3, 1, C2, C3, C2);   // GW[t] = Ghat2[t] W3, both towers
    g.batch_a = (long)C2 * C2; g.batch_b = 0; g.batch_c = (long)C2 * C3;
    def_gemm(h, w, g, 2);
  }
  float gscale3 = (float)(1.0 / Wn);
#ifdef ALIGNNET_ABLATE
  if (h->ablate_mutation == 4) gscale3 = 1.f;   // (mutation: the global-sum terms of dW3 are added `world` times by the gradient all-reduce)
#endif
  def_combine(h, w, CombineJob{S.Sp, nullptr, S.m2, S.kdb3, S.GW, S.E3, C2, C3, G(h, w, L[2]->p_w), gscale3});
  // W3^T for the VALU form of pass B2's sparse rows (the matrix-pipe form gathers from the bf16 table packed at the start of the step)
  if (!spm) hipLaunchKernelGGL(scale_cols2_kernel, g256t((size_t)C2 * C3), dim3(256), 0, h->stream, W3, C2, C3, nullptr, w->W3E, 0, 0, nullptr, w->W3T, 1, 1);
  const size_t qimgh = (size_t)((C2 + 31) / 32) * ((C2 + 15) / 16) * 512;   // bf16 image elements per tower
  GemmArgs gq3 = gemm_args(W3, C3, 1, W3, 1, C3, w->Q3, C2, 1, C2, C2, C3);   // Q3[t] = W3 diag(E3[t]) W3^T: the per-k scale rides on the product's A operand (no W3 diag(E) copy)
  gq3.batch_a = 0; gq3.batch_b = 0; gq3.batch_c = (long)C2 * C2; gq3.kscale = S.E3; gq3.batch_k = C3;
  if (q3_bf16) {
    // one launch: the product with its bf16 operand image written from the epilogue + pass B2's bias row from prep3's u (no Q3 read-back, no pack launch)
    if (!w->q3imgh) HIP_TRY(h, hipMalloc(&w->q3imgh, 2 * (size_t)4 * 8 * 512 * sizeof(unsigned short)));   // C2 <= 128
    gq3.img = w->q3imgh; gq3.batch_img = (long)qimgh;
    const int tx = (C2 + 31) / 32, ntile = tx * tx;
    hipLaunchKernelGGL(gemm_qimg_kernel, dim3(ntile + C2, 2), dim3(kGemmWaves * 64), 0, h->stream, gq3, tx, ntile,
                       QBias2Args{W3, C2, C3, w->u3, nullptr, nullptr, nullptr, M, w->q3b}, PackBf16Jobs{}, 0, 0u);
  } else {
    hipLaunchKernelGGL(gemm_small, dim3((C2 + 31) / 32, (C2 + 31) / 32, 2), dim3(kGemmWaves * 64), 0, h->stream, gq3);
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(16, 2), dim3(256), 0, h->stream, w->stage_pack + s * 6);
    hipLaunchKernelGGL(qbias_kernel, dim3(C2, 2), dim3(256), 0, h->stream, w->Q3, S.m2, W3, S.kdb3, C2, C3, M, w->q3b);
  }
  // ---- pass B2 ----
  BwdB2Args b2;
  b2.pcs[0] = p1; b2.pcs[1] = p2; b2.xform = S.xform; b2.B = B; b2.N = N; b2.C1 = C1; b2.C2 = C2; b2.C3 = C3;
  b2.ld0 = ((C1 + 7) & ~7) + 4; b2.ldb = ((std::max(C1, C2) + 7) & ~7) + 4;
  b2.w1 = P(h, L[0]->p_w); b2.wp2 = h->d_wp + L[1]->off_wp;
  b2.sc1 = S.scale[0]; b2.sh1 = S.shift[0]; b2.sc2 = S.scale[1]; b2.sh2 = S.shift[1];
  b2.b2 = P(h, L[1]->p_b); b2.mean2 = S.mean[1]; b2.rstd2 = S.rstd[1];
  b2.q3img = w->q3img; b2.q3img_stride = (long)qimg; b2.q3b = w->q3b; b2.gs = S.gs; b2.idx = S.idx; b2.w3t = w->W3T;
  b2.dbg = h->ablate_dbg;
  b2.stamps = (b2.dbg & 32) ? reinterpret_cast<long long*>(w->loss_scratch) : nullptr;   // scratch is free during the backward
  b2.dy2_store = w->dy2; b2.dbg2_part = w->dbg2_part; b2.u2_part = S.u2_part; b2.g1_part = S.g1_part;
  b2.s1_part = (!given && !h->train_bf16 && !b2_accum && !(h->ab & AB_PHASE2_LEGACY)) ? nullptr : w->s1_part;   // null: the forward kept the column sums of h1
  const size_t b2_extra = (size_t)C3 * 8 + (size_t)(kTW * ((N + kTT - 1) / kTT + 1) + kTW) * 4;
  b2.wp2h = h->train_bf16 ? w->wp2h[s] : nullptr; b2.q3imgh = w->q3imgh; b2.q3imgh_stride = (long)qimgh;
  b2.h2_given = S.h2;
  b2.w3th = spm ? w->w3th[s] : nullptr;
  const int pp = b2_accum ? 1 : pn_parts(h, B);   // workgroups per cloud of passes B2 and B1 (their outputs are sums the reductions below fold anyway)
  b2.parts = pp;
  const size_t b2_lds = lds_train(b2.ldb, b2.ldb) + b2_extra + (spm ? (size_t)(kTT * 72 + 128 * 72 + kTT * 72) * 2 - (size_t)kTT * b2.ldb * sizeof(float) + (5 * 64 + 16) * sizeof(float) : 0) +
                        ((std_w && !b2_accum && !h->train_bf16 && !given) ? (size_t)(5 * 64 + 16 + 6 * 128) * sizeof(float) : 0);   // STDF: the LDS parameter tables
  if (b2_lds > 160 * 1024) return fail(h, "training: num_points too large for the B2 hit-list LDS budget");   // (on the final size: the parameter tables count)
  { ProfScope prof_scope(h, PK_TRAIN_B2, true);
  if (given_bf16 && std_w) TIMED_LAUNCH((train_bwd_b2<false, true, true, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (given_bf16) TIMED_LAUNCH((train_bwd_b2<false, true, true>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (given && std_w) TIMED_LAUNCH((train_bwd_b2<false, false, true, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (given) TIMED_LAUNCH((train_bwd_b2<false, false, true>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (std_w && !b2_accum && h->train_bf16) TIMED_LAUNCH((train_bwd_b2<false, true, false, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (std_w && !b2_accum) TIMED_LAUNCH((train_bwd_b2<false, false, false, 64, 128>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (h->train_bf16 && b2_accum) TIMED_LAUNCH((train_bwd_b2<true, true>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (h->train_bf16) TIMED_LAUNCH((train_bwd_b2<false, true>), dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else if (b2_accum) TIMED_LAUNCH(train_bwd_b2<true>, dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  else TIMED_LAUNCH(train_bwd_b2<false>, dim3(2 * B * pp), dim3(kTW * 64), b2_lds, b2);
  }
  if (b2.stamps) {
    long long st[11];
    hipStreamSynchronize(h->stream);
    hipMemcpy(st, b2.stamps, sizeof(st), hipMemcpyDeviceToHost);
    std::fprintf(stderr, "B2 stage %d tile-3 phase cycles:", s);
    for (int i = 1; i < 11; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
    std::fprintf(stderr, "  total %lld\n", st[10] - st[0]);
  }

  if (hyb) {   // w->dy2 = dL/dh [h > 0] of layer n - 1's output: the layers in front of the tail take it from here
    HIP_TRY(h, hipGetLastError());
    return backbone_bwd_generic(h, s, B, w->dy2);
  }
  // ---- operators for B1 (the layer-2 weight gradient follows B1 when B1 accumulates U2 / Gram(h1)) ----
  const int CT1 = (C1 + 31) / 32, CT2 = (C2 + 31) / 32;
  const bool acc_in_b1 = CT1 * CT2 + CT1 * (CT1 + 1) / 2 <= 3 * kTW;   // register-resident blocks in B1 (every shipped config)
  const bool fwd_gram = !dg && !h->train_bf16 && acc_in_b1 && !(h->ab & AB_PHASE2_LEGACY);   // the forward kept s1 and Gram(h1) (backbone_fwd_train)
  bool u2_prescaled = false;
  int u2_slices = B * pp;   // per-workgroup partials of U2 per tower (dgcnn: B x workgroups per cloud, set where the edge pass is launched)
  auto layer2_weight_grad = [&]() {   // (deferred: dW2 = U2 diag(k2) - m1 (k db)^T + (Ghat1 W2) diag(E2))
    def_reduce(h, w, rjob(S.u2_part, u2_slices, (long)(C1 * C2), S.u2));
    if (fwd_gram || dg) def_reduce(h, w, rjob(S.g1f, 1, (long)(C1 * C1), S.g1));
    else {
      def_reduce(h, w, rjob(S.g1_part, B * pp, (long)(C1 * C1), S.g1));
      if (sync) {   // Gram(h1) of all ranks (the forward's, in the other two cases, already is)
        w->defer.sync_after_red.push_back({S.g1, (size_t)2 * C1 * C1});
      }
    }
    def_centre(h, w, CentreJob{S.g1, S.s1, C1, Me, S.m1});
    {
      GemmArgs g = gemm_args(S.g1, C1, 1, W2, C2, 1, S.GW2, C2, 1, C1, C2, C1);   // GW2[t] = Ghat1[t] W2
      g.batch_a = (long)C1 * C1; g.batch_b = 0; g.batch_c = (long)C1 * C2;
      def_gemm(h, w, g, 2);
    }
    def_combine(h, w, CombineJob{S.u2, u2_prescaled ? nullptr : S.k2, S.m1, S.kdb2, S.GW2, S.E2, C1, C2, G(h, w, L[1]->p_w), (float)(1.0 / Wn)});
  };
  const int sG = (h->train_bf16 && !b2_accum && !given) ? 8 : std::max(1, 256 / C1);   // row-group slices of B2's column sums of h1 (bf16: the lift's eight)
  {   // totals of (dbeta2, dgamma2) over the clouds + the hidden layer's backward coefficients, and s1 / m1 = s1 / M (qbias needs it before B1): one launch
    PrepHiddenArgs ph;
    ph.part = w->dbg2_part; ph.S = 2 * B * pp; ph.var = S.var[1]; ph.gamma[0] = P(h, L[1]->p_bn[0][1]); ph.gamma[1] = P(h, L[1]->p_bn[1][1]); ph.C = C2; ph.M = Me;
    for (int t = 0; t < 2; ++t) { ph.dbeta[t] = G(h, w, L[1]->p_bn[t][0]); ph.dgamma[t] = G(h, w, L[1]->p_bn[t][1]); }
    ph.E = S.E2; ph.kdb = S.kdb2; ph.kk = S.k2; ph.rstd = w->rstd2;
    ReduceJobs J{};
    const dim3 pgrid(std::max((C2 + kPhC - 1) / kPhC, (C1 + 31) / 32), 2, 3);
    if (sync) {
      // local (dbeta2, dgamma2) -> gradients + totals; s1 of this rank; then both over all ranks; then the coefficients and m1 = s1 / M
      if (dg || fwd_gram) J.j[0] = rjob(S.s1e, 1, (long)(C1), S.s1);   // (already the global sums: the forward all-reduced them)
      else J.j[0] = rjob(w->s1_part, B * pp * sG, (long)(C1), S.s1);
      ph.mode = 1; ph.totals = h->sync_buf;
      hipLaunchKernelGGL(prep_hidden_reduce_kernel, pgrid, dim3(1024), 0, h->stream, ph, J);
      if (sync_sum(h, h->sync_buf, (size_t)2 * C2 * 2, true)) return 1;
      if (!(dg || fwd_gram) && sync_sum(h, S.s1, (size_t)2 * C1, false)) return 1;
      ph.mode = 2;
      J.j[0] = rjob(S.s1, 1, (long)(C1), S.m1, (float)(1.0 / Me));
      hipLaunchKernelGGL(prep_hidden_reduce_kernel, pgrid, dim3(1024), 0, h->stream, ph, J);
    } else {
    if (dg || fwd_gram) {   // the forward kept the column sums of h1 (DGCNN: over all edge rows)
      J.j[0] = rjob(S.s1e, 1, (long)(C1), S.s1); J.j[1] = rjob(S.s1e, 1, (long)(C1), S.m1, (float)(1.0 / Me));
    } else {
      J.j[0] = rjob(w->s1_part, B * pp * sG, (long)(C1), S.s1); J.j[1] = rjob(w->s1_part, B * pp * sG, (long)(C1), S.m1, (float)(1.0 / Me));
    }
    hipLaunchKernelGGL(prep_hidden_reduce_kernel, pgrid, dim3(1024), 0, h->stream, ph, J);
    }
  }
  if (!acc_in_b1) layer2_weight_grad();
  const bool b1_bf16 = !dg && h->train_bf16 && C1 == 64 && C2 == 128 && !(h->ab & AB_B1_FP32) && !(h->ab & AB_B1_LEGACY);   // (packs its own bf16 images of V2 / Q2 below)
  // V2[t] = (W2 diag(k2))^T  [C2][C1]  (the bf16 pass B1 packs its image straight from W2 and k2: no fp32 copy)
  if (!b1_bf16) hipLaunchKernelGGL(scale_cols2_kernel, g256t((size_t)C1 * C2), dim3(256), 0, h->stream, W2, C1, C2, nullptr, w->W2E, 0, 0, S.k2, w->V2, 1, 2);
  const size_t vimg = img_floats(C2, C1), q2img = img_floats(C1, C1);
  const bool dg_bf16 = dg && h->train_bf16;   // edge pass with h1 Q2 on bf16 MFMA: bf16 images of Q2, packed with the bias row
  constexpr size_t kQ2hMax = 2 * 4 * 512;     // [CT1 <= 2][KG16 <= 4][64 lanes][8]
  constexpr size_t kV2h = 2 * 8 * 512, kQ2h = 2 * 4 * 512;   // bf16 pass B1 (64 / 128): [CT = 2][KG][64 lanes][8]
  {   // Q2[t] = W2 diag(E2[t]) W2^T
    GemmArgs g = gemm_args(W2, C2, 1, W2, 1, C2, w->Q2, C1, 1, C1, C1, C2);
    g.batch_a = 0; g.batch_b = 0; g.batch_c = (long)C1 * C1; g.kscale = S.E2; g.batch_k = C2;
    const int tx = (C1 + 31) / 32, ntile = tx * tx;
    const QBias2Args qa{W2, C1, C2, nullptr, S.m1, S.E2, S.kdb2, Me, w->q2b};   // (u formed in the bias-row blocks: C2 columns x a C1-long dot)
    if (b1_bf16) {
      // one launch: the product with its bf16 image, the bias row q2b, and the images of V2 = (W2 diag(k2))^T packed straight from W2 and k2
      if (!w->b1imgh) HIP_TRY(h, hipMalloc(&w->b1imgh, 2 * (kV2h + kQ2h) * sizeof(unsigned short)));
      g.img = w->b1imgh + 2 * kV2h; g.batch_img = (long)kQ2h;
      PackBf16Jobs pj{};
      for (int t = 0; t < 2; ++t) { pj.src[t] = W2; pj.tr[t] = 1; pj.rowscale[t] = S.k2 + (size_t)t * C2; pj.dst[t] = w->b1imgh + t * kV2h; pj.K[t] = C2; pj.C[t] = C1; }
      hipLaunchKernelGGL(gemm_qimg_kernel, dim3(ntile + C1 + 8, 2), dim3(kGemmWaves * 64), 0, h->stream, g, tx, ntile, qa, pj, 2, 8u);
    } else if (dg_bf16 && C2 <= 512) {
      if (!w->b1imgh) HIP_TRY(h, hipMalloc(&w->b1imgh, 2 * (2 * 8 * 512 + kQ2hMax) * sizeof(unsigned short)));   // (the PointNet bf16 B1's allocation)
      g.img = w->b1imgh; g.batch_img = (long)kQ2hMax;
      hipLaunchKernelGGL(gemm_qimg_kernel, dim3(ntile + C1, 2), dim3(kGemmWaves * 64), 0, h->stream, g, tx, ntile, qa, PackBf16Jobs{}, 0, 0u);
    } else {
      hipLaunchKernelGGL(gemm_small, dim3(tx, tx, 2), dim3(kGemmWaves * 64), 0, h->stream, g);
      hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(16, 4), dim3(256), 0, h->stream, w->stage_pack + s * 6 + 2);
      hipLaunchKernelGGL(qbias_kernel, dim3(C1, 2), dim3(256), 0, h->stream, w->Q2, S.m1, W2, S.kdb2, C1, C2, Me, w->q2b);
    }
  }
  if (dg) {
    // ---- edge pass + first layer from the reduced quantities (kernels_train_dgcnn.h) ----
    DgBwdArgs e;
    e.pcs[0] = p1; e.pcs[1] = p2; e.xform = S.xform; e.nn = w->nn; e.B = B; e.N = N; e.k = kDgK; e.C1 = C1; e.C2 = C2;
    e.ld0 = ((C1 + 7) & ~7) + 4;
    e.w1 = P(h, L[0]->p_w); e.sc1 = S.scale[0]; e.sh1 = S.shift[0];
    e.v2 = w->V2; e.v2_stride = (long)C1 * C2; e.q2img = w->q2img; e.q2img_stride = (long)q2img; e.q2b = w->q2b;
    e.q2imgh = w->b1imgh; e.q2imgh_stride = (long)kQ2hMax;
    e.dyp = w->dy2; e.argk = S.argk; e.u2_part = S.u2_part; e.g1_part = nullptr; e.pdy_part = w->pdy_part;   // Gram(h1): the forward's
    e.stamps = (b2.dbg & 32) ? reinterpret_cast<long long*>(w->loss_scratch) : nullptr;
    e.parts = dg_parts(h, B, 1);
    u2_slices = B * e.parts;
    const dim3 eg(2 * B * e.parts), eb(kBEW * 64);
    const size_t el = dg_bwd_edge_lds(C1, C2, dg_bf16);
    const bool dense = dg_bf16 && !(h->ab & AB_DG_SPARSE);   // bf16 mode: both dy2_s products as dense bf16 MFMAs on per-slot tiles
    e.k2 = S.k2; e.w2th = w->w2th[s];
    const size_t eld = dg_bwd_edge_dense_lds(C1, C2);
  { ProfScope prof_scope(h, PK_DG_BWD_EDGE, true);
    if (dense && C1 == 64 && C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge_dense<64, 128>), eg, eb, eld, e);
    else if (dense && C1 == 64) TIMED_LAUNCH((dg_train_bwd_edge_dense<64, 64>), eg, eb, eld, e);
    else if (dense && C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge_dense<32, 128>), eg, eb, eld, e);
    else if (dense) TIMED_LAUNCH((dg_train_bwd_edge_dense<32, 64>), eg, eb, eld, e);
    else if (dg_bf16 && C1 == 64 && C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge<64, 128, true>), eg, eb, el, e);
    else if (dg_bf16 && C1 == 64) TIMED_LAUNCH((dg_train_bwd_edge<64, 64, true>), eg, eb, el, e);
    else if (dg_bf16 && C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge<32, 128, true>), eg, eb, el, e);
    else if (dg_bf16) TIMED_LAUNCH((dg_train_bwd_edge<32, 64, true>), eg, eb, el, e);
    else if (C1 == 64 && C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge<64, 128>), eg, eb, el, e);
    else if (C1 == 64) TIMED_LAUNCH((dg_train_bwd_edge<64, 64>), eg, eb, el, e);
    else if (C2 == 128) TIMED_LAUNCH((dg_train_bwd_edge<32, 128>), eg, eb, el, e);
    else TIMED_LAUNCH((dg_train_bwd_edge<32, 64>), eg, eb, el, e);
  }
    u2_prescaled = dense;   // the dense form accumulates U2 diag(k2)
    if (e.stamps) {
      long long st[19];
      hipStreamSynchronize(h->stream);
      hipMemcpy(st, e.stamps, sizeof(st), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "BE stage %d it-25 phase cycles:", s);
      for (int i = 1; i < 8; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "  | dh1: init %lld mfma %lld epilogue %lld | P2 (waves 4-7, from the dh1 start) %lld .. %lld", st[8] - st[6], st[9] - st[8], st[7] - st[9],
                   st[17] - st[6], st[18] - st[6]);
      std::fprintf(stderr, "  | tile start (it 40): barrier %lld staging %lld barrier %lld row lists %lld column lists %lld | 20 slots + 1 tile start %lld\n",
                   st[11] - st[10], st[12] - st[11], st[13] - st[12], st[14] - st[13], st[15] - st[14], st[16] - st[0]);
    }
    layer2_weight_grad();
    DgB0Args z;
    z.pdy_part = w->pdy_part; z.slices = 4 * e.parts; z.mom = S.mom; z.w1 = P(h, L[0]->p_w); z.b1 = P(h, L[0]->p_b);   // ([cloud][part][4] partials: 4 parts contiguous slices per cloud)
    z.mean1 = S.mean[0]; z.rstd1 = S.rstd[0]; z.k1 = S.kk[0]; z.B = B; z.C1 = C1; z.rows = N * kDgK; z.count = Me;
    for (int t = 0; t < 2; ++t) { z.dbeta[t] = G(h, w, L[0]->p_bn[t][0]); z.dgamma[t] = G(h, w, L[0]->p_bn[t][1]); }
    z.dbg1 = w->dbg1; z.p_part = S.p_part; z.gx = S.gx; z.grot = S.grot;
    set_glue(z);
    hipLaunchKernelGGL(dg_b0_totals<6>, dim3((C1 + kB0C - 1) / kB0C, 2), dim3(1024), 0, h->stream, z);
    if (sync && sync_sum(h, w->dbg1, (size_t)2 * C1 * 2, false)) return 1;   // (dbeta1, dgamma1) of all ranks: the per-cloud part divides them by the global count
    hipLaunchKernelGGL(dg_b0_cloud<6>, dim3(2 * B), dim3(128), 0, h->stream, z);
    def_reduce(h, w, rjob(S.p_part, 2 * B, (long)6 * C1, G(h, w, L[0]->p_w), 1.f, 1));
    HIP_TRY(h, hipGetLastError());
    return 0;
  }
  // ---- pass B1 ----
  BwdB1Args b1;
  b1.pcs[0] = p1; b1.pcs[1] = p2; b1.xform = S.xform; b1.B = B; b1.N = N; b1.C1 = C1; b1.C2 = C2;
  b1.ld0 = ((C1 + 7) & ~7) + 4; b1.ldb = ((C2 + 7) & ~7) + 4;
  b1.w1 = P(h, L[0]->p_w); b1.sc1 = S.scale[0]; b1.sh1 = S.shift[0]; b1.b1 = P(h, L[0]->p_b); b1.mean1 = S.mean[0]; b1.rstd1 = S.rstd[0];
  b1.v2img = w->v2img; b1.q2img = w->q2img; b1.v2img_stride = (long)vimg; b1.q2img_stride = (long)q2img; b1.q2b = w->q2b;
  b1.dy2_store = w->dy2; b1.dy1_store = w->dy1; b1.dbg1_part = w->dbg1_part; b1.dy2_bf16 = h->train_bf16 ? 1 : 0;
  b1.u2_part = acc_in_b1 ? S.u2_part : nullptr; b1.g1_part = (acc_in_b1 && !fwd_gram) ? S.g1_part : nullptr;
  // (the legacy train_bwd_b1<64, 128> with compile-time widths unrolls further and spills 67 registers -- the generic one is kept)
  const bool pdy = C1 <= 64 && acc_in_b1 && !(h->ab & AB_B1_LEGACY);   // one dh1 item per wave: no stored dy1, no pass B0
  b1.pdy_part = w->pdy_part; b1.parts = pp;
  const bool b1h = pdy && std_w && h->train_bf16 && !(h->ab & AB_B1_FP32);
  if (b1h) {
    // bf16 pass B1 (kernels_train_bwd.h: train_bwd_b1_bf16); the bf16 images of V2 / Q2 and the bias row came out of the Q2 launch above
    BwdB1hArgs bh;
    bh.pcs[0] = p1; bh.pcs[1] = p2; bh.xform = S.xform; bh.B = B; bh.N = N;
    bh.w1 = P(h, L[0]->p_w); bh.sc1 = S.scale[0]; bh.sh1 = S.shift[0];
    bh.v2imgh = w->b1imgh; bh.q2imgh = w->b1imgh + 2 * kV2h; bh.v2_stride = (long)kV2h; bh.q2_stride = (long)kQ2h; bh.q2b = w->q2b;
    bh.dy2_store = reinterpret_cast<const unsigned short*>(w->dy2);
    bh.u2_part = b1.u2_part; bh.g1_part = b1.g1_part; bh.pdy_part = w->pdy_part; bh.parts = pp;
    const size_t ldsh = (size_t)kTT * 4 * sizeof(float) + ((size_t)kTT * 72 * 2 + (size_t)kTT * 136 + (size_t)128 * 72) * sizeof(unsigned short) +
                        (size_t)(2 * 8 + 2 * 4) * 64 * 16;   // + the bf16 operand images of V2 and Q2 (kernels_train_bwd.h: Vl, Ql)
    ProfScope prof_scope(h, PK_TRAIN_B1, true);
    TIMED_LAUNCH(train_bwd_b1_bf16, dim3(2 * B * pp), dim3(kTW * 64), ldsh, bh);
  } else
  { ProfScope prof_scope(h, PK_TRAIN_B1, true);
  if (pdy && std_w) TIMED_LAUNCH((train_bwd_b1<64, 128, true>), dim3(2 * B * pp), dim3(kTW * 64), lds_train(b1.ld0, b1.ldb), b1);
  else if (pdy) TIMED_LAUNCH((train_bwd_b1<0, 0, true>), dim3(2 * B * pp), dim3(kTW * 64), lds_train(b1.ld0, b1.ldb), b1);
  else TIMED_LAUNCH(train_bwd_b1<>, dim3(2 * B * pp), dim3(kTW * 64), lds_train(b1.ld0, b1.ldb), b1);
  }
  if (acc_in_b1) layer2_weight_grad();
  if (pdy) {
    // first layer from the reduced quantities (kernels_train_dgcnn.h, D = 3)
    DgB0Args z;
    z.pdy_part = w->pdy_part; z.slices = 4 * pp; z.mom = S.mom; z.w1 = P(h, L[0]->p_w); z.b1 = P(h, L[0]->p_b);
    z.mean1 = S.mean[0]; z.rstd1 = S.rstd[0]; z.k1 = S.kk[0]; z.B = B; z.C1 = C1; z.rows = N; z.count = M;
    if (b1h) z.slices = pp;   // train_bwd_b1_bf16 reduces the four partials itself
    for (int t = 0; t < 2; ++t) { z.dbeta[t] = G(h, w, L[0]->p_bn[t][0]); z.dgamma[t] = G(h, w, L[0]->p_bn[t][1]); }
    z.dbg1 = w->dbg1; z.p_part = S.p_part; z.gx = S.gx; z.grot = S.grot;
    set_glue(z);
    hipLaunchKernelGGL(dg_b0_totals<3>, dim3((C1 + kB0C - 1) / kB0C, 2), dim3(1024), 0, h->stream, z);
    if (sync && sync_sum(h, w->dbg1, (size_t)2 * C1 * 2, false)) return 1;   // (dbeta1, dgamma1) of all ranks: the per-cloud part divides them by the global count
    hipLaunchKernelGGL(dg_b0_cloud<3>, dim3(2 * B), dim3(128), 0, h->stream, z);
    def_reduce(h, w, rjob(S.p_part, 2 * B, (long)3 * C1, G(h, w, L[0]->p_w), 1.f, 1));
    HIP_TRY(h, hipGetLastError());
    return 0;
  }
  launch_reduce<double>(h, w->dbg1_part, 4 * B * pp, (long)(C1 * 2), w->dbg1);
  hipLaunchKernelGGL(prep_hidden_kernel, dim3((C1 + 127) / 128, 2), dim3(128), 0, h->stream, w->dbg1, S.var[0], P(h, L[0]->p_bn[0][1]),
                     P(h, L[0]->p_bn[1][1]), C1, M, G(h, w, L[0]->p_bn[0][0]), G(h, w, L[0]->p_bn[1][0]), G(h, w, L[0]->p_bn[0][1]),
                     G(h, w, L[0]->p_bn[1][1]), (float*)nullptr, (float*)nullptr, w->k1, w->rstd1);

  // ---- pass B0 ----
  BwdB0Args b0;
  b0.pcs[0] = p1; b0.pcs[1] = p2; b0.xform = S.xform; b0.B = B; b0.N = N; b0.C1 = C1;
  b0.w1 = P(h, L[0]->p_w); b0.b1 = P(h, L[0]->p_b); b0.mean1 = S.mean[0]; b0.rstd1 = S.rstd[0]; b0.k1 = S.kk[0]; b0.dbg1 = w->dbg1;
  b0.count = M; b0.dy1_store = w->dy1; b0.p_part = S.p_part; b0.gx = S.gx; b0.grot = S.grot;
  hipLaunchKernelGGL(train_bwd_b0, dim3(2 * B), dim3(256), 1024 * 4 * sizeof(float) + 256 * 4 * sizeof(double) + (size_t)C1 * 4 * sizeof(float),
                     h->stream, b0);
  def_reduce(h, w, rjob(S.p_part, 2 * B, (long)3 * C1, G(h, w, L[0]->p_w), 1.f, 1));
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// full forward + backward on device buffers
// ---------------------------------------------------------------------------------
static int comm_bucket(alignnet_handle* h, int stage, hipStream_t after = nullptr);   // defined with the RCCL section below
static int comm_join(alignnet_handle* h);
static void comm_poison(alignnet_handle* h);   // loopback group: a failed rank must not leave the others waiting at the next rendezvous

static int fwd_bwd_device(alignnet_handle* h, const float* p1, const float* p2, const float* const lab[6], int B, const float* u_dev,
                          int do_backward, int update_ema, int comm_overlap = 0)
{
  TrainWS* w = tws(h);
  const int N = h->cfg.num_points, nb = h->cfg.num_bins, nb2 = 2 * nb, B2 = 2 * B;
  alignnet_state stt;
  alignnet_get_state(h, &stt);
  const float bn_decay = stt.bn_decay;
  if (set_lds_attrs(h)) return 1;
  h->sync_collectives = 0;
  h->last_train_B = B;
  w->last_pcs[0] = p1; w->last_pcs[1] = p2;
  w->last_ang[0] = lab[4]; w->last_ang[1] = lab[5];
  if (h->prof_pending.size() > 4096 && alignnet_drain_profile(h)) return 1;   // (as the eval forward does: ~15 event pairs per profiled step)
  {
    bool std_all = true;   // all three backbones on the instantiations with the widths (64, 128) compiled in
    for (int s = 0; s < 3; ++s) std_all = std_all && h->layers[conv_of(h, s).first].cout == 64 && h->layers[conv_of(h, s).first + 1].cout == 128;
    bool any_gen = false;
    for (int s = 0; s < 3; ++s) any_gen = any_gen || stage_generic(h, s);
    bool any_hyb = false;
    for (int s = 0; s < 3; ++s) any_hyb = any_hyb || stage_hybrid(h, s);
    h->last_train_kernel = (std_all ? 1 : 0) | (h->train_bf16 ? 2 : 0) | (h->cfg.backbone == 1 ? 4 : 0) | (any_gen ? 8 : 0) | (any_hyb ? 16 : 0);
  }
  for (int s = 0; s < 2; ++s) {   // pooled layout [tower][B rows][C]: the tower stride is THIS call's B (the workspace may be carved for a larger one)
    const Stack& cs = conv_of(h, s);
    w->st[s].tower_stride = (long)B * h->layers[cs.first + cs.n - 1].cout;
  }
  if (do_backward && !w->grad_clean) HIP_TRY(h, hipMemsetAsync(w->grad, 0, h->n_trainable * sizeof(float), h->stream));   // incl. the BN-fed biases (exact zero)
  if (do_backward) w->grad_clean = false;   // (clean = the optimiser zeroed it behind its read: alignnet_apply_gradients)
  {
    StartJobs sj;
    if (pack_all_weights(h, &sj)) return 1;
    const MomentsTail mt = moments_tail_of(h, 0);
    hipLaunchKernelGGL(train_start_kernel, dim3(B2 + sj.nf32 * kStartF32Blocks + sj.nbf16 * kStartBf16Blocks), dim3(256), 0, h->stream, p1, p2, B, N,
                       w->st[0].xform, w->center_mean, sj.f32, sj.nf32, sj.pj, sj.nbf16, mt);
    w->moments_done[0] = mt.mom != nullptr;
  }
  if (h->cfg.backbone == 1) {   // static kNN graph, once per cloud in the mean-centred frame (as the eval path: alignnet_api.hip)
    ProfScope prof_scope(h, PK_KNN, true);
    prof_scope.used = true;
    HIP_TRY(h, launch_knn(h->cfg.device, h->stream, p1, p2, w->center_mean, B, N, kDgK, w->nn, prof_scope.a, prof_scope.b));
  }
  // stage 1
  if (backbone_fwd_train(h, 0, p1, p2, B, bn_decay, update_ema)) return 1;
  if (head_fwd_train(h, 0, w->st[0].pooled, w->st[0].row_stride, B2, B, bn_decay, update_ema, u_dev)) return 1;
  if (const MomentsTail mt = moments_tail_of(h, 1); mt.mom) {
    hipLaunchKernelGGL(stage1_finish_moments_kernel, dim3(B2), dim3(256), 0, h->stream, w->o[0], w->center_mean, B, N, w->s1c, w->st[1].xform, w->outs[2],
                       w->outs[3], p1, p2, mt);
    w->moments_done[1] = true;
  } else
  hipLaunchKernelGGL(stage1_finish_kernel, dim3((B2 + 127) / 128), dim3(128), 0, h->stream, w->o[0], w->center_mean, B, w->s1c,
                     w->st[1].xform, w->outs[2], w->outs[3]);
  // stage 2
  if (backbone_fwd_train(h, 1, p1, p2, B, bn_decay, update_ema)) return 1;
  if (head_fwd_train(h, 1, w->st[1].pooled, w->st[1].row_stride, B2, B, bn_decay, update_ema, u_dev)) return 1;
  if (const MomentsTail mt = moments_tail_of(h, 2); mt.mom) {
    hipLaunchKernelGGL(stage2_finish_moments_kernel, dim3(B2), dim3(256), 0, h->stream, w->o[1], 3 + nb2, w->s1c, B, N, nb, w->s2c, w->st[2].xform, w->theta,
                       w->cls, w->outs[4], w->outs[5], w->outs[6], w->outs[7], p1, p2, mt);
    w->moments_done[2] = true;
  } else
  hipLaunchKernelGGL(stage2_finish_kernel, dim3((B2 + 3) / 4), dim3(256), 0, h->stream, w->o[1], 3 + nb2, w->s1c, B, nb, w->s2c,
                     w->st[2].xform, w->theta, w->cls, w->outs[4], w->outs[5], w->outs[6], w->outs[7]);
  // stage 3
  if (backbone_fwd_train(h, 2, p1, p2, B, bn_decay, update_ema)) return 1;
  if (head_fwd_train(h, 2, w->st[2].pooled, w->st[2].row_stride, B, B, bn_decay, update_ema, u_dev)) return 1;
  const bool ff_fold = !gloss_on(h) && !(h->ab & AB_NO_GLUE_FOLD);   // (global loss: the loss kernels run on the gathered batch, the glue on this rank's rows)
  if (!ff_fold)
    hipLaunchKernelGGL(final_finish_kernel, dim3((B * (3 + nb2) + 255) / 256), dim3(256), 0, h->stream, w->o[2], 3 + nb2, w->s2c, B, nb, w->outs[0], w->outs[1]);
  // loss (+ gradient wrt the end points)
  LossArgs la;
  if (ff_fold) { la.ff_net = w->o[2]; la.ff_ldn = 3 + nb2; la.ff_s2c = w->s2c; la.ff_B = B; la.ff_out_t = w->outs[0]; la.ff_out_l = w->outs[1]; }
  la.B = B; la.nb = nb; la.esf = h->cfg.early_stage_factor; la.af = h->cfg.angle_factor; la.accept_inverted = h->cfg.accept_inverted_angle;
  la.s1c = w->s1c; la.s2c = w->s2c; la.o2 = w->o[1]; la.ldo2 = 3 + nb2; la.o3 = w->o[2]; la.ldo3 = 3 + nb2; la.theta = w->theta; la.pcls = w->cls;
  la.tr = lab[0]; la.c1 = lab[2]; la.c2 = lab[3]; la.a1 = lab[4]; la.a2 = lab[5];
  la.out = w->loss_out; la.d_s1c = w->d_s1c; la.d_s2c = w->d_s2c; la.d_o2 = w->d_o[1]; la.d_o3 = w->d_o[2]; la.scratch = w->loss_scratch;
  la.want_grad = do_backward;
  if (gloss_on(h)) {
    // The reference's loss couples every sample of the (global) batch: all-gather what the loss reads -- per tower, so that the gathered
    // arrays keep the [tower 0 rows | tower 1 rows] layout -- run the same three kernels on the global batch, take this rank's rows of
    // the gradient.  The loss is already divided by the global B: the gradient all-reduce sums (reduce_and_apply).
    const int Wg = sync_world(h), Bg = B * Wg, ld = 3 + nb2;
    int rk = h->comm ? h->comm_rank : 0;
#ifdef ALIGNNET_ABLATE
    if (h->ablate_mutation == 2) rk = 0;            // (mutation: every rank keeps rank 0's rows of the loss gradient)
    const bool swap_towers = h->ablate_mutation == 1;   // (mutation: the stage-2 centres are gathered with the towers swapped)
#else
    const bool swap_towers = false;
#endif
    if (Bg > w->gl_cap) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      if (w->gl_base) hipFree(w->gl_base);
      const size_t B2g = 2 * (size_t)Bg;
      size_t off = 0;
      auto take = [&](size_t n) { const size_t o = off; off += (n + 63) & ~(size_t)63; return o; };
      const size_t o_s1c = take(B2g * 3), o_s2c = take(B2g * 3), o_o2 = take(B2g * ld), o_o3 = take((size_t)Bg * ld), o_th = take(B2g), o_cls = take(B2g);
      const int lw[6] = {3, 1, 3, 3, 1, 1};
      size_t o_lab[6];
      for (int i = 0; i < 6; ++i) o_lab[i] = take((size_t)Bg * lw[i]);
      const size_t o_ds1 = take(B2g * 3), o_ds2 = take(B2g * 3), o_do2 = take(B2g * ld), o_do3 = take((size_t)Bg * ld), o_scr = take(loss_scratch_floats(Bg));
      HIP_TRY(h, hipMalloc(&w->gl_base, off * sizeof(float)));
      float* b0 = w->gl_base;
      w->gl_s1c = b0 + o_s1c; w->gl_s2c = b0 + o_s2c; w->gl_o2 = b0 + o_o2; w->gl_o3 = b0 + o_o3; w->gl_theta = b0 + o_th; w->gl_cls = reinterpret_cast<int*>(b0 + o_cls);
      for (int i = 0; i < 6; ++i) w->gl_lab[i] = b0 + o_lab[i];
      w->gl_d_s1c = b0 + o_ds1; w->gl_d_s2c = b0 + o_ds2; w->gl_d_o2 = b0 + o_do2; w->gl_d_o3 = b0 + o_do3; w->gl_scratch = b0 + o_scr;
      w->gl_cap = Bg;
    }
    for (int t = 0; t < 2; ++t) {
      if (sync_gather(h, w->s1c + (size_t)t * B * 3, w->gl_s1c + (size_t)t * Bg * 3, (size_t)B * 3)) return 1;
      if (sync_gather(h, w->s2c + (size_t)t * B * 3, w->gl_s2c + (size_t)(swap_towers ? 1 - t : t) * Bg * 3, (size_t)B * 3)) return 1;
      if (sync_gather(h, w->o[1] + (size_t)t * B * ld, w->gl_o2 + (size_t)t * Bg * ld, (size_t)B * ld)) return 1;
      if (sync_gather(h, w->theta + (size_t)t * B, w->gl_theta + (size_t)t * Bg, (size_t)B)) return 1;
      if (sync_gather(h, w->cls + (size_t)t * B, w->gl_cls + (size_t)t * Bg, (size_t)B)) return 1;
    }
    if (sync_gather(h, w->o[2], w->gl_o3, (size_t)B * ld)) return 1;
    const int lw[6] = {3, 1, 3, 3, 1, 1};
    for (int i = 0; i < 6; ++i) if (sync_gather(h, lab[i], w->gl_lab[i], (size_t)B * lw[i])) return 1;
    la.B = Bg; la.s1c = w->gl_s1c; la.s2c = w->gl_s2c; la.o2 = w->gl_o2; la.o3 = w->gl_o3; la.theta = w->gl_theta; la.pcls = w->gl_cls;
    la.tr = w->gl_lab[0]; la.c1 = w->gl_lab[2]; la.c2 = w->gl_lab[3]; la.a1 = w->gl_lab[4]; la.a2 = w->gl_lab[5];
    la.d_s1c = w->gl_d_s1c; la.d_s2c = w->gl_d_s2c; la.d_o2 = w->gl_d_o2; la.d_o3 = w->gl_d_o3; la.scratch = w->gl_scratch;
    launch_loss(h, la);
    if (do_backward) {
      auto cp = [&](float* dst, const float* src, size_t n) { return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream); };
      for (int t = 0; t < 2; ++t) {
        HIP_TRY(h, cp(w->d_s1c + (size_t)t * B * 3, w->gl_d_s1c + ((size_t)t * Bg + (size_t)rk * B) * 3, (size_t)B * 3));
        HIP_TRY(h, cp(w->d_s2c + (size_t)t * B * 3, w->gl_d_s2c + ((size_t)t * Bg + (size_t)rk * B) * 3, (size_t)B * 3));
        HIP_TRY(h, cp(w->d_o[1] + (size_t)t * B * ld, w->gl_d_o2 + ((size_t)t * Bg + (size_t)rk * B) * ld, (size_t)B * ld));
      }
      HIP_TRY(h, cp(w->d_o[2], w->gl_d_o3 + (size_t)rk * B * ld, (size_t)B * ld));
    }
  } else
  launch_loss(h, la);
  HIP_TRY(h, hipGetLastError());
  if (!do_backward) return 0;

  // ---- backward ----
  const int CE = h->layers[h->emb_conv.first + h->emb_conv.n - 1].cout;
  h->comm_buckets = 0;
  w->defer.clear();
  w->defer.on = !(h->ab & AB_NO_DEFER) || sync_on(h);   // (ablation switch: every weight-gradient job launched where its inputs appear, as before round 3)
  // The deferred weight-gradient jobs of a stage go to a side stream as soon as that stage's backward is queued: they run under the next
  // stage's head / prep chains (a few workgroups each, most of the chip idle), and in data-parallel steps the stage's gradient segment
  // is final -- and its all-reduce bucket on its way -- two stages earlier than with one flush at the end.  sync_bn keeps the single
  // flush on the compute stream (its reduced matrices are summed over the ranks between the job groups).
  // "train_dw_side_stream": 1 = every stage's deferred jobs on a second (low-priority) stream right behind that stage's backward; 2 = only
  // stage 3's (55 % of the deferred work: its sparse gather reads 134 MB of h2 rows) -- they then run under the head / prep chains of
  // stages 2 and 1, where a handful of workgroups leave most of the chip idle; the rest stays one group after the backward
  const int side_mode = (w->defer.on && !sync_on(h)) ? h->dw_side : 0;
  // Data-parallel steps (communicator + "allreduce_overlap"): a stage's deferred jobs are flushed right behind that stage's backward, so
  // that its segment of the flat gradient is final and its all-reduce bucket leaves on the comm stream UNDER the next stage's backward:
  // stage 3's bucket (64 % of the vector for the shipped widths) travels under stages 2 and 1, stage 2's (23 %) under stage 1, only the
  // last and smallest one (14 %) is exposed.  Without a communicator the three stages' jobs stay one group after the whole backward
  // (five launches instead of fifteen: the one-GPU step does not pay for an overlap it does not need).
  h->comm_order = 0;
  auto stage_flush = [&](int sg) -> int {
    h->comm_order = h->comm_order * 10 + 1 + sg;   // "comm_order": digit 1..3 = that stage's backward is queued, 4..6 = that stage's bucket is issued
    if (side_mode == 1 || (side_mode == 2 && sg == 2)) {
      if (!h->side_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least priority: the passes on the compute stream go first wherever both want a CU
        HIP_TRY(h, hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, lo));
        for (auto& e : h->side_ev) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      HIP_TRY(h, hipEventRecord(h->side_ev[sg], h->stream));
      HIP_TRY(h, hipStreamWaitEvent(h->side_stream, h->side_ev[sg], 0));
      if (flush_deferred(h, h->side_stream)) return 1;
      if (comm_overlap && comm_bucket(h, sg, h->side_stream)) return 1;
      return 0;
    }
    if (!comm_overlap) return 0;
    if (w->defer.on && flush_deferred(h, h->stream)) return 1;
    return comm_bucket(h, sg);
  };
  if (head_bwd_train(h, 2, w->st[2].pooled, 2L * CE, w->st[2].dP, B, B, u_dev)) return 1;
  if (backbone_bwd_train(h, 2, p1, p2, B)) return 1;
  if (stage_flush(2)) return 1;
  if (!w->glue_folded) {
    hipLaunchKernelGGL(stage3_glue_bwd_kernel, dim3((B2 + 127) / 128), dim3(128), 0, h->stream, w->st[2].gx, w->st[2].grot, w->st[2].xform, w->cls,
                       B, nb, w->d_s2c, w->d_o[1], 3 + nb2);
    hipLaunchKernelGGL(stage2_glue_bwd_kernel, dim3((B2 * 3 + 127) / 128), dim3(128), 0, h->stream, w->d_s2c, B, w->d_o[1], 3 + nb2, w->d_s1c);
  }
  const int C2l = h->layers[h->s2_conv.first + h->s2_conv.n - 1].cout, C1l = h->layers[h->s1_conv.first + h->s1_conv.n - 1].cout;
  if (head_bwd_train(h, 1, w->st[1].pooled, C2l, w->st[1].dP, B2, B, u_dev)) return 1;
  if (backbone_bwd_train(h, 1, p1, p2, B)) return 1;
  if (stage_flush(1)) return 1;
  if (!w->glue_folded) {
    hipLaunchKernelGGL(stage1_glue_bwd_kernel, dim3((B2 * 3 + 127) / 128), dim3(128), 0, h->stream, w->st[1].gx, B, w->d_s1c);
    // s1c = o1 + center_mean  ->  d_o1 = d_s1c
    HIP_TRY(h, hipMemcpyAsync(w->d_o[0], w->d_s1c, (size_t)B2 * 3 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  }
  if (head_bwd_train(h, 0, w->st[0].pooled, C1l, w->st[0].dP, B2, B, u_dev)) return 1;
  if (backbone_bwd_train(h, 0, p1, p2, B)) return 1;
  if (stage_flush(0)) return 1;
  if (w->defer.on && flush_deferred(h, h->stream)) return 1;   // whatever is still recorded (no communicator: all three stages' weight gradients, five multi-job launches)
  if (side_mode) {
    HIP_TRY(h, hipEventRecord(h->side_ev[3], h->side_stream));   // the optimiser (compute stream) reads the whole gradient
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->side_ev[3], 0));
  }
  w->defer.on = false;
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// optimiser (train.py:211-217): tf.train.AdamOptimizer / MomentumOptimizer
// ---------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float gscale, float lr_t, float b1, float b2, float eps)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  g[i] = 0.f;   // the gradient is consumed: the next step's backward starts from zeros without a memset launch of its own (8.7 MB, two fill kernels)
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  w[i] -= lr_t * mi / (sqrtf(vi) + eps);   // epsilon outside the bias correction (TF form)
}

__global__ void momentum_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ acc, size_t n, float gscale,
                                float lr, float mom)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = mom * acc[i] + g[i] * gscale;
  g[i] = 0.f;
  acc[i] = a;
  w[i] -= lr * a;
}

extern "C" int alignnet_apply_gradients(alignnet_handle* h, float grad_scale)
{
  if (!h) return 1;
  TrainWS* w = tws(h);
  if (!w->grad) return fail(h, "alignnet_apply_gradients: no gradients computed yet");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  alignnet_state st;
  alignnet_get_state(h, &st);   // schedules use the pre-increment step (train.py:145-150,172)
  const size_t n = h->n_trainable;
  const dim3 grid((unsigned)((n + 255) / 256));
  ProfScope prof_scope(h, PK_OPTIMIZER, true);
  if (h->cfg.optimizer == 0) {
    // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) exactly as TF evaluates it: in float32, with the beta powers kept as float32
    // products (one multiplication per step).  With the kernel's float32 (1 - beta) factors this makes the first step lr * sign(g)
    // to the last bit of the arithmetic; double-precision powers here were 6.4e-6 off (1 - 0.999f != 0.001).
    const int64_t t = h->step + 1;
    if (h->adam_power_t > t || h->adam_power_t < 0) { h->adam_b1p = h->adam_b2p = 1.f; h->adam_power_t = 0; }
    while (h->adam_power_t < t) {
      h->adam_b1p *= 0.9f; h->adam_b2p *= 0.999f; h->adam_power_t++;
      if (h->adam_b2p == 0.f) { h->adam_power_t = t; break; }   // both powers have underflowed: further products stay 0
    }
    const float lr_t = st.learning_rate * std::sqrt(1.f - h->adam_b2p) / (1.f - h->adam_b1p);
    TIMED_LAUNCH(adam_kernel, grid, dim3(256), 0, h->d_params, w->grad, w->adam_m, w->adam_v, n, grad_scale, lr_t, 0.9f, 0.999f, 1e-8f);
  } else {
    TIMED_LAUNCH(momentum_kernel, grid, dim3(256), 0, h->d_params, w->grad, w->adam_m, n, grad_scale, st.learning_rate, h->cfg.momentum);
  }
  HIP_TRY(h, hipGetLastError());
  w->grad_clean = true;
  h->step += 1;
  h->folded = false;
  return 0;
}

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
static int stage_inputs(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels, int B,
                        const float* dropout_u, const float** u_dev)
{
  TrainWS* w = tws(h);
  const size_t nin = (size_t)B * h->cfg.num_points * 3 * sizeof(float);
  HIP_TRY(h, hipMemcpyAsync(w->d_pcs[0], pcs1, nin, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(w->d_pcs[1], pcs2, nin, hipMemcpyHostToDevice, h->stream));
  const float* src[6] = {labels->translations, labels->rel_angles, labels->pc1_centers, labels->pc2_centers, labels->pc1_angles, labels->pc2_angles};
  const int lw[6] = {3, 1, 3, 3, 1, 1};
  for (int i = 0; i < 6; ++i) {
    if (!src[i]) return fail(h, "training: all six label tensors are required");
    HIP_TRY(h, hipMemcpyAsync(w->labels[i], src[i], (size_t)B * lw[i] * sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  *u_dev = nullptr;
  if (dropout_u) {
    const int w1 = h->layers[h->s1_fc.first + h->s1_fc.n - 2].cout, w2 = h->layers[h->s2_fc.first + h->s2_fc.n - 2].cout;
    const int w3 = h->layers[h->rem_fc.first + h->rem_fc.n - 2].cout;
    if (w1 != w2) return fail(h, "explicit dropout uniforms need equal last-hidden widths in the s1/s2 heads");
    const size_t n = (size_t)B * (4 * (size_t)w1 + w3);
    HIP_TRY(h, hipMemcpyAsync(w->dropout_u, dropout_u, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    *u_dev = w->dropout_u;
  }
  return 0;
}

static int fetch_result(alignnet_handle* h, alignnet_step_result* result, const alignnet_outputs* out, int B, const alignnet_state& pre)
{
  TrainWS* w = tws(h);
  float lo[17];
  HIP_TRY(h, hipMemcpyAsync(lo, w->loss_out, sizeof(lo), hipMemcpyDeviceToHost, h->stream));
  if (out) {
    float* host[8] = {out->pred_translations, out->pred_remaining_angle_logits, out->pred_s1_pc1centers, out->pred_s1_pc2centers,
                      out->pred_s2_pc1centers, out->pred_s2_pc2centers, out->pred_pc1angle_logits, out->pred_pc2angle_logits};
    const int nb2 = 2 * h->cfg.num_bins;
    const int widths[8] = {3, nb2, 3, 3, 3, 3, nb2, nb2};
    for (int i = 0; i < 8; ++i)
      if (host[i]) HIP_TRY(h, hipMemcpyAsync(host[i], w->outs[i], (size_t)B * widths[i] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (result) {
    result->step = h->step;
    result->loss = lo[0];
    result->learning_rate = pre.learning_rate;
    result->bn_decay = pre.bn_decay;
    for (int i = 0; i < 16; ++i) result->summaries[i] = lo[1 + i];
  }
  return 0;
}

static int train_fb_host(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels, int32_t B,
                         const float* dropout_u, int comm_overlap)
{
  if (!pcs1 || !pcs2 || !labels) return fail(h, "alignnet_train_forward_backward: null argument");
  if (B < 2) return fail(h, "training needs B >= 2 (batch statistics)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (check_trainable_shape(h)) return 1;
  if (ensure_train_ws(h, B)) return 1;
  TrainWS* w = tws(h);
  const float* u_dev = nullptr;
  if (stage_inputs(h, pcs1, pcs2, labels, B, dropout_u, &u_dev)) return 1;
  return fwd_bwd_device(h, w->d_pcs[0], w->d_pcs[1], w->labels, B, u_dev, 1, 1, comm_overlap);
}

static int alignnet_train_forward_backward_impl(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels,
                                               int32_t B, const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out);
extern "C" int alignnet_train_forward_backward(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels,
                                               int32_t B, const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out)
{
  const int rc = alignnet_train_forward_backward_impl(h, pcs1, pcs2, labels, B, dropout_u, result, out);
  if (rc) comm_poison(h);
  return rc;
}
static int alignnet_train_forward_backward_impl(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels,
                                               int32_t B, const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out)
{
  if (!h) return 1;
  alignnet_state pre;
  alignnet_get_state(h, &pre);
  if (train_fb_host(h, pcs1, pcs2, labels, B, dropout_u, 0)) return 1;
  return fetch_result(h, result, out, B, pre);
}

// all-reduce (bucketed next to the backward, or one call after it) + optimiser: the tail of both train_step entry points
static int reduce_and_apply(alignnet_handle* h, int overlapped)
{
  float scale = 1.f;
  if (h->comm) {
    ProfScope prof_scope(h, PK_ALLREDUCE);   // what the compute stream waits for: the part of the all-reduce the backward did not hide
    if (overlapped ? comm_join(h) : alignnet_comm_allreduce_grads(h)) return 1;
    scale = h->global_loss ? 1.f : 1.f / (float)h->comm_world;   // global_loss: the loss is already that of the global batch (divided by its B): sum
  }
  return alignnet_apply_gradients(h, scale);
}

static int alignnet_train_step_impl(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels, int32_t B,
                                   const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out);
extern "C" int alignnet_train_step(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels, int32_t B,
                                   const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out)
{
  const int rc = alignnet_train_step_impl(h, pcs1, pcs2, labels, B, dropout_u, result, out);
  if (rc) comm_poison(h);
  return rc;
}
static int alignnet_train_step_impl(alignnet_handle* h, const float* pcs1, const float* pcs2, const alignnet_labels* labels, int32_t B,
                                   const float* dropout_u, alignnet_step_result* result, const alignnet_outputs* out)
{
  if (!h) return 1;
  alignnet_state pre;
  alignnet_get_state(h, &pre);
  const int overlap = h->comm && h->comm_overlap;
  if (train_fb_host(h, pcs1, pcs2, labels, B, dropout_u, overlap)) return 1;
  if (reduce_and_apply(h, overlap)) return 1;
  return fetch_result(h, result, out, B, pre);
}

static int alignnet_train_step_device_impl(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2, const alignnet_labels* d_labels,
                                          int32_t B, alignnet_step_result* result);
extern "C" int alignnet_train_step_device(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2, const alignnet_labels* d_labels,
                                          int32_t B, alignnet_step_result* result)
{
  const int rc = alignnet_train_step_device_impl(h, d_pcs1, d_pcs2, d_labels, B, result);
  if (rc) comm_poison(h);
  return rc;
}
static int alignnet_train_step_device_impl(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2, const alignnet_labels* d_labels,
                                          int32_t B, alignnet_step_result* result)
{
  if (!h) return 1;
  if (!d_pcs1 || !d_pcs2 || !d_labels) return fail(h, "alignnet_train_step_device: null argument");
  if (B < 2) return fail(h, "training needs B >= 2 (batch statistics)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (check_trainable_shape(h)) return 1;
  if (ensure_train_ws(h, B)) return 1;
  const float* lab[6] = {d_labels->translations, d_labels->rel_angles, d_labels->pc1_centers, d_labels->pc2_centers, d_labels->pc1_angles,
                         d_labels->pc2_angles};
  alignnet_state pre;
  alignnet_get_state(h, &pre);
  const int overlap = h->comm && h->comm_overlap;
  if (fwd_bwd_device(h, d_pcs1, d_pcs2, lab, B, nullptr, 1, 1, overlap)) return 1;
  if (reduce_and_apply(h, overlap)) return 1;
  if (result) return fetch_result(h, result, nullptr, B, pre);
  return 0;
}

extern "C" int alignnet_eval_loss(alignnet_handle* h, const alignnet_labels* labels, int32_t B, float* loss, float* summaries)
{
  if (!h) return 1;
  if (!labels) return fail(h, "alignnet_eval_loss: null labels");
  if (B != h->last_B || B < 1) return fail(h, "alignnet_eval_loss: B must equal the batch of the preceding alignnet_forward");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  Workspace& ws = h->ws;
  const int nb = h->cfg.num_bins, nb2 = 2 * nb;
  // label + scratch staging
  float* stage = nullptr;
  const size_t nlab = ((size_t)B * 12 + 63) & ~(size_t)63, nscr = loss_scratch_floats(B) + 32;
  HIP_TRY(h, hipMalloc(&stage, (nlab + nscr) * sizeof(float)));
  const float* src[6] = {labels->translations, labels->rel_angles, labels->pc1_centers, labels->pc2_centers, labels->pc1_angles, labels->pc2_angles};
  const int lw[6] = {3, 1, 3, 3, 1, 1};
  float* dl[6]; size_t off = 0;
  for (int i = 0; i < 6; ++i) {
    dl[i] = stage + off; off += (size_t)B * lw[i];
    if (!src[i]) { hipFree(stage); return fail(h, "alignnet_eval_loss: all six label tensors are required"); }
    hipMemcpyAsync(dl[i], src[i], (size_t)B * lw[i] * sizeof(float), hipMemcpyHostToDevice, h->stream);
  }
  LossArgs la{};
  la.B = B; la.nb = nb; la.esf = h->cfg.early_stage_factor; la.af = h->cfg.angle_factor; la.accept_inverted = h->cfg.accept_inverted_angle;
  la.s1c = ws.s1c; la.s2c = ws.s2c; la.o2 = ws.o2; la.ldo2 = 3 + nb2; la.o3 = ws.o3; la.ldo3 = 3 + nb2; la.theta = ws.theta; la.pcls = ws.cls;
  la.tr = dl[0]; la.c1 = dl[2]; la.c2 = dl[3]; la.a1 = dl[4]; la.a2 = dl[5];
  la.out = stage + nlab; la.scratch = stage + nlab + 32; la.want_grad = 0;
  launch_loss(h, la);
  float lo[17];
  hipMemcpyAsync(lo, la.out, sizeof(lo), hipMemcpyDeviceToHost, h->stream);
  hipError_t e = hipStreamSynchronize(h->stream);
  hipFree(stage);
  if (e != hipSuccess) return fail(h, std::string("alignnet_eval_loss: ") + hipGetErrorString(e));
  if (loss) *loss = lo[0];
  if (summaries) for (int i = 0; i < 16; ++i) summaries[i] = lo[1 + i];
  return 0;
}

extern "C" int alignnet_grad_buffer(alignnet_handle* h, float** d_grad, size_t* count)
{
  if (!h) return 1;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  TrainWS* w = tws(h);
  if (!w->grad && ensure_train_ws(h, 0)) return 1;
  // The caller may write through this pointer (an external all-reduce, a hand-made gradient): from here on the engine no longer knows the
  // buffer to be all zeros, so the next backward clears it itself instead of trusting the optimiser's "consumed" state.
  if (d_grad) { *d_grad = w->grad; w->grad_clean = false; }
  if (count) *count = h->n_trainable;
  return 0;
}

extern "C" int alignnet_get_grad(alignnet_handle* h, const char* name, float* dst, size_t count)
{
  if (!h) return 1;
  if (!name || !dst) return fail(h, "alignnet_get_grad: null argument");
  auto it = h->by_name.find(name);
  if (it == h->by_name.end()) return fail(h, std::string("alignnet_get_grad: unknown variable '") + name + "'");
  const ParamInfo& p = h->params[it->second];
  if (!p.trainable) return fail(h, "alignnet_get_grad: variable is not trainable");
  if (p.count() != count) return fail(h, "alignnet_get_grad: element count mismatch");
  TrainWS* w = tws(h);
  if (!w->grad) return fail(h, "alignnet_get_grad: no gradients computed yet");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(dst, w->grad + p.offset, count * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int alignnet_train_set_option(alignnet_handle* h, const std::string& k, int64_t value)
{
  if (k == "loopback_timeout_s") {   // rendezvous timeout of the in-process loopback groups (process-wide; csrc/comm_loopback.h)
    if (value < 1) return fail(h, "loopback_timeout_s must be >= 1");
    loop_timeout() = std::chrono::seconds(value);
    return 0;
  }
  return -1;
}

int alignnet_train_get_option(alignnet_handle* h, const std::string& k, int64_t* value)
{
  (void)h;
  if (k == "loopback_timeout_s") { *value = (int64_t)loop_timeout().count(); return 0; }
  if (k == "grad_communicator") { *value = h->comm_grad ? 1 : 0; return 0; }   // 1: the gradient buckets have a communicator of their own (alignnet_comm_init_grad)
  return -1;
}

// test hook (include/alignnet_hip.h): the discontinuous choices of the last training forward -- decoded yaw classes, the max-pool's
// arg-max points, the dgcnn branch's arg-max neighbour slots and the neighbour table itself -- read back from the workspace
extern "C" int alignnet_debug_train_decisions(alignnet_handle* h, int32_t kind, int32_t stage, int32_t* dst, size_t count)
{
  if (!h) return 1;
  if (!dst) return fail(h, "alignnet_debug_train_decisions: null argument");
  TrainWS* w = static_cast<TrainWS*>(h->train_ws);
  const int B = h->last_train_B, N = h->cfg.num_points;
  if (!w || !w->base || B < 1) return fail(h, "alignnet_debug_train_decisions: no training forward has run on this handle");
  const bool dg = h->cfg.backbone == 1;
  const size_t B2 = 2 * (size_t)B;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  auto copy_i32 = [&](const int* src, size_t n) -> int {
    if (count != n) return fail(h, "alignnet_debug_train_decisions: count does not match the requested array (" + std::to_string(n) + " elements)");
    HIP_TRY(h, hipMemcpy(dst, src, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
  };
  if (kind == ALIGNNET_DECISION_YAW_CLASS) return copy_i32(w->cls, B2);
  if (kind == ALIGNNET_DECISION_ANGLE_CLASS) {
    // the class tf_angle2class (models/tp8.py:193-199) put every target angle of the loss into -- for the pair term a [B, B] matrix (:327) --
    // recomputed by the loss kernels' own device function from the step's labels and decoded yaws
    if (stage < 0 || stage > 2) return fail(h, "alignnet_debug_train_decisions: the loss has angle terms 0 (tower 1), 1 (tower 2), 2 (pair)");
    const size_t W = stage == 2 ? (size_t)B : 1, n = 2 * (size_t)B * W;
    if (count != n) return fail(h, "alignnet_debug_train_decisions: count does not match the requested array (" + std::to_string(n) + " elements)");
    if (!w->last_ang[0] || !w->last_ang[1]) return fail(h, "alignnet_debug_train_decisions: no labels seen yet");
    int* d = nullptr;
    HIP_TRY(h, hipMalloc(&d, n * sizeof(int)));
    hipLaunchKernelGGL(dbg_angle_class_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, w->last_ang[0], w->last_ang[1], w->theta, B, h->cfg.num_bins, stage, d);
    hipError_t e = hipMemcpyAsync(dst, d, n * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIP_TRY(h, e);
    return 0;
  }
  if (kind == ALIGNNET_DECISION_KNN_GRAPH) {
    if (!dg) return fail(h, "alignnet_debug_train_decisions: the neighbour table exists for the dgcnn backbone only");
    return copy_i32(w->nn, B2 * N * kDgK);
  }
  if (stage < 0 || stage > 2) return fail(h, "alignnet_debug_train_decisions: stage must be 0, 1 or 2");
  const Stack& st = conv_of(h, stage);
  const bool gen = stage_generic(h, stage), hyb = stage_hybrid(h, stage);
  if (kind == ALIGNNET_DECISION_POOL_POINT) {
    const int Cl = h->layers[st.first + st.n - 1].cout;
    return copy_i32((gen && !hyb) ? w->gen[stage].idx : w->st[stage].idx, B2 * Cl);
  }
  if (kind == ALIGNNET_DECISION_EDGE_SLOT) {
    if (!dg) return fail(h, "alignnet_debug_train_decisions: neighbour slots exist for the dgcnn backbone only");
    const int Ce = h->layers[st.first + st.n - 2].cout;   // width of the last edge conv = input width of the point conv
    const size_t n = B2 * N * Ce;
    if (gen) return copy_i32(w->gen[stage].argk, n);
    if (count != n) return fail(h, "alignnet_debug_train_decisions: count does not match the requested array (" + std::to_string(n) + " elements)");
    std::vector<unsigned char> bytes(n);   // the fused edge kernels keep one byte per (point, channel)
    HIP_TRY(h, hipMemcpy(bytes.data(), w->st[stage].argk, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) dst[i] = bytes[i];
    return 0;
  }
  return fail(h, "alignnet_debug_train_decisions: unknown kind");
}

// test hook (include/alignnet_hip.h): the sign every relu of the last training forward / backward saw, one byte per element (kernels_debug.h)
extern "C" int alignnet_debug_train_relu_mask(alignnet_handle* h, int32_t kind, int32_t stage, int32_t layer, uint8_t* dst, size_t count)
{
  if (!h) return 1;
  if (!dst) return fail(h, "alignnet_debug_train_relu_mask: null argument");
  TrainWS* w = static_cast<TrainWS*>(h->train_ws);
  const int B = h->last_train_B, N = h->cfg.num_points;
  if (!w || !w->base || B < 1) return fail(h, "alignnet_debug_train_relu_mask: no training forward has run on this handle");
  if (stage < 0 || stage > 2) return fail(h, "alignnet_debug_train_relu_mask: stage must be 0, 1 or 2");
  const bool dg = h->cfg.backbone == 1, was_bf16 = (h->last_train_kernel & 2) != 0;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  size_t n = 0;
  unsigned char* d = nullptr;
  auto begin = [&](size_t elems) -> int {
    n = elems;
    if (count != n) return fail(h, "alignnet_debug_train_relu_mask: count does not match the requested array (" + std::to_string(n) + " elements)");
    HIP_TRY(h, hipMalloc(&d, n));
    return 0;
  };
  auto finish = [&]() -> int {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(dst, d, n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIP_TRY(h, e);
    return 0;
  };
  const dim3 blk(256);
  auto grid_of = [](size_t elems) { return dim3((unsigned)((elems + 255) / 256)); };
  if (kind == ALIGNNET_RELU_HEAD) {
    const Stack& fs = fc_of(h, stage);
    if (layer < 0 || layer >= fs.n - 1) return fail(h, "alignnet_debug_train_relu_mask: the head has no hidden layer of that index");
    const Layer& L = h->layers[fs.first + layer];
    const HeadLayerWS& HL = w->hl[stage][layer];
    const int M = stage < 2 ? 2 * B : B, rows_per_set = B;
    if (begin((size_t)M * L.cout)) return 1;
    const int s1 = L.p_bn[1][0] >= 0 ? 1 : 0;
    hipLaunchKernelGGL(dbg_mask_head_kernel, grid_of(n), blk, 0, h->stream, HL.z, HL.mean, HL.var, P(h, L.p_bn[0][1]), P(h, L.p_bn[s1][1]),
                       P(h, L.p_bn[0][0]), P(h, L.p_bn[s1][0]), M, L.cout, rows_per_set, d);
    return finish();
  }
  if (kind != ALIGNNET_RELU_CONV) return fail(h, "alignnet_debug_train_relu_mask: unknown kind");
  const Stack& st = conv_of(h, stage);
  if (layer < 0 || layer >= st.n) return fail(h, "alignnet_debug_train_relu_mask: the stage has no conv layer of that index");
  const bool gen = stage_generic(h, stage), hyb = stage_hybrid(h, stage);
  const StageWS& S = w->st[stage];
  const TrainWS::GenStage& Gs = w->gen[stage];
  const Layer& L = h->layers[st.first + layer];
  const size_t B2 = 2 * (size_t)B, rowsN = (size_t)B * N;   // rows per tower over the points
  if (layer == st.n - 1) {   // the conv in front of the max over the points: the sign at the winner = the sign of the pooled feature
    if (begin(B2 * L.cout)) return 1;
    hipLaunchKernelGGL(dbg_mask_pooled_kernel, grid_of(n), blk, 0, h->stream, S.pooled, S.tower_stride, S.row_stride, B, L.cout, d);
    return finish();
  }
  if (dg && layer == st.n - 2) {   // the last edge conv, in front of the max over the k neighbours: the sign of the pooled edge feature
    if (begin(2 * rowsN * L.cout)) return 1;
    hipLaunchKernelGGL(dbg_mask_rows_kernel<float>, grid_of(n), blk, 0, h->stream, gen ? Gs.P : S.h2, nullptr, nullptr, rowsN, L.cout, d);
    return finish();
  }
  const size_t rows = dg ? rowsN * kDgK : rowsN;
  if (begin(2 * rows * L.cout)) return 1;
  if (gen) {   // layer by layer (also the front of a hybrid stage): the pre-BatchNorm activations and the batch-statistics scale / shift are in the workspace
    hipLaunchKernelGGL(dbg_mask_rows_kernel<float>, grid_of(n), blk, 0, h->stream, Gs.Z[layer], Gs.scale[layer], Gs.shift[layer], rows, L.cout, d);
    return finish();
  }
  (void)hyb;
  if (layer == 1) {   // fused PointNet stage: the stored h2 (bf16 in the bf16 step)
    if (was_bf16) hipLaunchKernelGGL(dbg_mask_rows_kernel<unsigned short>, grid_of(n), blk, 0, h->stream, reinterpret_cast<const unsigned short*>(S.h2), nullptr, nullptr, rows, L.cout, d);
    else hipLaunchKernelGGL(dbg_mask_rows_kernel<float>, grid_of(n), blk, 0, h->stream, S.h2, nullptr, nullptr, rows, L.cout, d);
    return finish();
  }
  // layer 0 of a fused stage: recomputed from xyz by every pass -- here by the passes' own device functions
  const float* p1 = w->last_pcs[0]; const float* p2 = w->last_pcs[1];   // (a device-pointer step: the caller's buffers, which must still hold that batch)
  const int C1 = L.cout;
  if (dg) {
    DbgEdge1Args a{{p1, p2}, S.xform, w->nn, B, N, kDgK, P(h, L.p_w), S.scale[0], S.shift[0], d};
    const size_t lds = ((size_t)kTT * 8 + (size_t)kTT * (C1 + 4)) * sizeof(float);
    if (C1 == 64) hipLaunchKernelGGL(dbg_mask_edge1_kernel<64>, dim3(2 * B), dim3(kTW * 64), lds, h->stream, a);
    else hipLaunchKernelGGL(dbg_mask_edge1_kernel<32>, dim3(2 * B), dim3(kTW * 64), lds, h->stream, a);
  } else {
    const int ld0 = ((C1 + 7) & ~7) + 4;
    DbgLayer1Args a{{p1, p2}, S.xform, B, N, C1, ld0, P(h, L.p_w), S.scale[0], S.shift[0], d};
    hipLaunchKernelGGL(dbg_mask_layer1_kernel, dim3(2 * B), dim3(kTW * 64), ((size_t)kTT * 4 + (size_t)kTT * ld0) * sizeof(float), h->stream, a);
  }
  return finish();
}

// test hook (include/alignnet_hip.h): the bf16-rounded activations the last bf16 training step multiplied (fused PointNet stages)
extern "C" int alignnet_debug_train_rounded(alignnet_handle* h, int32_t stage, int32_t layer, uint16_t* dst, size_t count)
{
  if (!h) return 1;
  if (!dst) return fail(h, "alignnet_debug_train_rounded: null argument");
  TrainWS* w = static_cast<TrainWS*>(h->train_ws);
  const int B = h->last_train_B, N = h->cfg.num_points;
  if (!w || !w->base || B < 1) return fail(h, "alignnet_debug_train_rounded: no training forward has run on this handle");
  if (!(h->last_train_kernel & 2)) return fail(h, "alignnet_debug_train_rounded: the last training step did not run with train_matmul_bf16");
  if (stage < 0 || stage > 2 || stage_generic(h, stage)) return fail(h, "alignnet_debug_train_rounded: fused stages 0..2 only");
  if (layer < 0 || layer > 1) return fail(h, "alignnet_debug_train_rounded: layer 0 (h1, input of the hidden conv) or 1 (h2, input of the lift)");
  const bool dg = h->cfg.backbone == 1;
  const Stack& st = conv_of(h, stage);
  const Layer& L = h->layers[st.first + layer];
  const StageWS& S = w->st[stage];
  const size_t n = 2 * (size_t)B * N * L.cout * ((dg && layer == 0) ? kDgK : 1);
  if (count != n) return fail(h, "alignnet_debug_train_rounded: count does not match the requested array (" + std::to_string(n) + " elements)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (layer == 1 && !dg) {   // the stored h2 IS the rounded tile
    HIP_TRY(h, hipMemcpy(dst, S.h2, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return 0;
  }
  unsigned short* d = nullptr;
  HIP_TRY(h, hipMalloc(&d, n * sizeof(uint16_t)));
  const int C1 = L.cout, ld0 = ((C1 + 7) & ~7) + 4;
  if (dg && layer == 1)   // the pooled edge features p (fp32 in the workspace), rounded as the bf16 point conv rounds them while staging
    hipLaunchKernelGGL(dbg_round_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, S.h2, d, n);
  else if (dg) {
    DbgEdgeR1Args a{{w->last_pcs[0], w->last_pcs[1]}, S.xform, w->nn, B, N, kDgK, P(h, L.p_w), S.scale[0], S.shift[0], d};
    const size_t lds = ((size_t)kTT * 8 + (size_t)kTT * (C1 + 4)) * sizeof(float);
    if (C1 == 64) hipLaunchKernelGGL(dbg_rounded_edge1_kernel<64>, dim3(2 * B), dim3(kTW * 64), lds, h->stream, a);
    else hipLaunchKernelGGL(dbg_rounded_edge1_kernel<32>, dim3(2 * B), dim3(kTW * 64), lds, h->stream, a);
  } else {
  DbgRound1Args a{{w->last_pcs[0], w->last_pcs[1]}, S.xform, B, N, C1, ld0, P(h, L.p_w), S.scale[0], S.shift[0], d};
  hipLaunchKernelGGL(dbg_rounded_layer1_kernel, dim3(2 * B), dim3(kTW * 64), ((size_t)kTT * 4 + (size_t)kTT * ld0) * sizeof(float), h->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(dst, d, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d);
  HIP_TRY(h, e);
  return 0;
}

// ---------------------------------------------------------------------------------
// RCCL (loaded lazily so that single-GPU use does not depend on librccl)
// ---------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // ncclUniqueId passed by value = 128-byte struct
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId { char internal[128]; };
Rccl g_rccl;
bool load_rccl(std::string& err)
{
  if (g_rccl.lib) return true;
  void* l = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!l) l = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!l) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
  g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(l, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, const void*, int)>(dlsym(l, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(l, "ncclAllReduce"));
  g_rccl.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(l, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(l, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(l, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) { err = "librccl.so lacks the expected nccl* symbols"; return false; }
  g_rccl.lib = l;
  return true;
}
}  // namespace

// ncclCommInitRank takes ncclUniqueId BY VALUE; declare the real prototype for the call
typedef int (*InitRankByValue)(void**, int, UniqueId, int);

// What h->comm points to: an RCCL communicator (one process per GPU, xGMI) or a rank of an in-process loopback group (comm_loopback.h:
// W handles of one process on one device -- the multi-rank code paths with distinct shards on a 1-GPU box).
struct CommImpl { void* nccl = nullptr; LoopComm* loop = nullptr; };
static CommImpl* comm_of(const alignnet_handle* h) { return static_cast<CommImpl*>(h->comm); }
// the communicator the gradient buckets travel on: the handle's second one when alignnet_comm_init_grad created it, else the first
static CommImpl* grad_comm_of(const alignnet_handle* h) { return static_cast<CommImpl*>(h->comm_grad ? h->comm_grad : h->comm); }

// sum of `n` floats / doubles over the ranks, in place, in stream order on `s`
static int comm_allreduce(alignnet_handle* h, void* buf, size_t n, bool is_double, hipStream_t s, const char* what, CommImpl* c = nullptr)
{
  if (!c) c = comm_of(h);
  if (c->loop) {
    std::string err;
    if (loop_collective(c->loop, is_double ? kLoopSumF64 : kLoopSumF32, buf, nullptr, n, s, err)) return fail(h, std::string(what) + ": " + err);
    return 0;
  }
  const int rc = g_rccl.AllReduce(buf, buf, n, is_double ? 8 : 7, 0, c->nccl, s);   // ncclFloat64 = 8, ncclFloat32 = 7, ncclSum = 0
  if (rc != 0) return fail(h, std::string("ncclAllReduce (") + what + "): " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
  return 0;
}

// dst[rank][n4] = every rank's n4 four-byte elements, in stream order on `s`
static int comm_allgather(alignnet_handle* h, const void* src, void* dst, size_t n4, hipStream_t s)
{
  CommImpl* c = comm_of(h);
  if (c->loop) {
    std::string err;
    if (loop_collective(c->loop, kLoopGather, const_cast<void*>(src), dst, n4, s, err)) return fail(h, "all-gather: " + err);
    return 0;
  }
  if (!g_rccl.AllGather) return fail(h, "librccl.so lacks ncclAllGather");
  const int rc = g_rccl.AllGather(src, dst, n4, 7, c->nccl, s);   // ncclFloat32 = 7 (four-byte elements; class ids travel as bits)
  if (rc != 0) return fail(h, std::string("ncclAllGather (global_loss): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
  return 0;
}

// A rank that fails between two collectives would leave the other ranks' threads waiting at the next rendezvous: break the group.
static void comm_poison(alignnet_handle* h)
{
  if (h && h->comm && comm_of(h)->loop) comm_of(h)->loop->g->poison();
  if (h && h->comm_grad && static_cast<CommImpl*>(h->comm_grad)->loop) static_cast<CommImpl*>(h->comm_grad)->loop->g->poison();
}

extern "C" int alignnet_comm_unique_id(uint8_t id[128])
{
  std::string err;
  if (!id || !load_rccl(err)) return 1;
  UniqueId u;
  if (g_rccl.GetUniqueId(&u) != 0) return 1;
  std::memcpy(id, u.internal, 128);
  return 0;
}

extern "C" int alignnet_comm_loopback_id(uint8_t id[128])
{
  if (!id) return 1;
  loop_make_id(id);
  return 0;
}

static CommImpl* comm_create(alignnet_handle* h, int rank, int world, const uint8_t id[128])
{
  if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return nullptr; }
  std::string err;
  if (loop_is_id(id)) {
    LoopComm* lc = loop_join(id, rank, world, h->cfg.device, err);
    if (!lc) { h->err = err; return nullptr; }
    CommImpl* c = new CommImpl(); c->loop = lc;
    return c;
  }
  if (!load_rccl(err)) { h->err = err; return nullptr; }
  UniqueId u;
  std::memcpy(u.internal, id, 128);
  void* comm = nullptr;
  const int rc = reinterpret_cast<InitRankByValue>(g_rccl.CommInitRank)(&comm, world, u, rank);
  if (rc != 0) { h->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return nullptr; }
  CommImpl* c = new CommImpl(); c->nccl = comm;
  return c;
}

extern "C" int alignnet_comm_init(alignnet_handle* h, int32_t rank, int32_t world, const uint8_t id[128])
{
  if (!h) return 1;
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(h, "alignnet_comm_init: bad arguments");
  if (h->comm) return fail(h, "alignnet_comm_init: this handle already has a communicator");
  CommImpl* c = comm_create(h, rank, world, id);
  if (!c) return 1;
  h->comm = c; h->comm_world = world; h->comm_rank = rank;
  return 0;
}

// A SECOND communicator of the same ranks for the gradient buckets alone.  NCCL-style communicators serialise the collectives issued
// on them whatever stream they are issued on: with "sync_bn" a per-layer sum of the NEXT stage's backward (compute stream) queues
// behind the gradient bucket of the previous stage (comm stream) when both use one communicator, and the bucket no longer travels
// under the backward but in front of it.  With their own communicator the buckets only meet the compute stream at the optimiser.
extern "C" int alignnet_comm_init_grad(alignnet_handle* h, int32_t rank, int32_t world, const uint8_t id[128])
{
  if (!h) return 1;
  if (!id) return fail(h, "alignnet_comm_init_grad: bad arguments");
  if (!h->comm) return fail(h, "alignnet_comm_init_grad: create the handle's first communicator (alignnet_comm_init) before the gradient one");
  if (h->comm_grad) return fail(h, "alignnet_comm_init_grad: this handle already has a gradient communicator");
  if (world != h->comm_world || rank != h->comm_rank) return fail(h, "alignnet_comm_init_grad: rank / world differ from the first communicator's");
  CommImpl* c = comm_create(h, rank, world, id);
  if (!c) return 1;
  h->comm_grad = c;
  return 0;
}

extern "C" void alignnet_comm_free(alignnet_handle* h)
{
  if (!h) return;
  if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
  for (void** slot : {&h->comm_grad, &h->comm}) {
    if (!*slot) continue;
    CommImpl* c = static_cast<CommImpl*>(*slot);
    if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
    if (c->loop) { if (h->stream) hipStreamSynchronize(h->stream); loop_leave(c->loop); }
    delete c;
    *slot = nullptr;
  }
  h->comm_world = 1; h->comm_rank = 0;
  for (auto& e : h->comm_ev) if (e) { hipEventDestroy(e); e = nullptr; }
  if (h->comm_stream) { hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
}

// Bucket `stage` of the gradient all-reduce: the flat gradient is laid out [stage 1 | stage 2 | stage 3] (graph-construction order,
// alignnet_api.hip), and the backward runs stage 3 -> 2 -> 1, writing only that stage's segment.  As soon as a segment is final
// the side stream waits for it and all-reduces it over xGMI while the compute stream goes on with the next stage's backward.
static int comm_bucket(alignnet_handle* h, int stage, hipStream_t after)   // after: the stream whose work completes the segment (default: the compute stream)
{
  if (!h->comm) return 0;
  TrainWS* w = tws(h);
  if (!h->comm_stream) {
    HIP_TRY(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    for (auto& e : h->comm_ev) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const size_t lo = h->params[h->layers[conv_of(h, stage).first].p_w].offset;
  const size_t hi = stage == 2 ? h->n_trainable : h->params[h->layers[conv_of(h, stage + 1).first].p_w].offset;
  if (hi <= lo || hi > h->n_trainable) return fail(h, "comm_bucket: gradient segments are not in stage order");
  HIP_TRY(h, hipEventRecord(h->comm_ev[stage], after ? after : h->stream));
  HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->comm_ev[stage], 0));
  if (comm_allreduce(h, w->grad + lo, hi - lo, false, h->comm_stream, "gradient bucket", grad_comm_of(h))) return 1;
  h->comm_buckets++;
  h->comm_order = h->comm_order * 10 + 4 + stage;
  return 0;
}

// The optimiser (compute stream) must not start before the last bucket has landed.
static int comm_join(alignnet_handle* h)
{
  if (!h->comm || !h->comm_stream || h->comm_buckets != 3) return fail(h, "comm_join: the three gradient buckets were not issued");
  HIP_TRY(h, hipEventRecord(h->comm_ev[3], h->comm_stream));
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->comm_ev[3], 0));
  return 0;
}

template <typename T>
__global__ void scale_buf_kernel(T* __restrict__ p, size_t n, T f)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] *= f;
}

// sum of one small buffer over the data-parallel ranks, in stream order on the compute stream.  Without a communicator the test hook
// "sync_bn_emulate_world" = w stands for w ranks holding identical shards: every sum is w times this rank's.
static int sync_sum(alignnet_handle* h, void* buf, size_t n, bool is_double)
{
  h->sync_collectives++;
  if (h->comm) return comm_allreduce(h, buf, n, is_double, h->stream, "sync_bn");
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (is_double) hipLaunchKernelGGL(scale_buf_kernel<double>, dim3(grid), dim3(256), 0, h->stream, static_cast<double*>(buf), n, (double)h->sync_emulate_world);
  else hipLaunchKernelGGL(scale_buf_kernel<float>, dim3(grid), dim3(256), 0, h->stream, static_cast<float*>(buf), n, (float)h->sync_emulate_world);
  return 0;
}

// all-gather of one small array (four-byte elements) over the data-parallel ranks, rank-major, in stream order on the compute stream;
// without a communicator "sync_bn_emulate_world" = w stands for w ranks holding identical shards: w copies
static int sync_gather(alignnet_handle* h, const void* src, void* dst, size_t n4)
{
  h->sync_collectives++;
  if (h->comm) return comm_allgather(h, src, dst, n4, h->stream);
  for (int r = 0; r < h->sync_emulate_world; ++r)
    HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(dst) + (size_t)r * n4 * 4, src, n4 * 4, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

// Local-BN data parallelism updates each rank's BatchNorm EMA shadows from its own shard's statistics; averaging them across the ranks
// (the mean of the per-rank EMAs is the EMA of the per-rank means) keeps every rank's eval-mode model -- and the checkpoint rank 0
// writes -- identical.  The shadows are the non-trainable tail of the flat variable vector: one all-reduce + a scale, on the device.
extern "C" int alignnet_comm_average_shadows(alignnet_handle* h)
{
  if (!h) return 1;
  if (!h->comm) return fail(h, "alignnet_comm_average_shadows: communicator not initialised");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t n = h->n_total - h->n_trainable;
  if (!n) return 0;
  float* p = h->d_params + h->n_trainable;
  if (comm_allreduce(h, p, n, false, h->stream, "EMA shadows")) { comm_poison(h); return 1; }
  hipLaunchKernelGGL(scale_buf_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p, n, 1.f / (float)h->comm_world);
  HIP_TRY(h, hipGetLastError());
  h->folded = false;   // the eval-mode scale / shift are derived from the shadows
  return 0;
}

extern "C" int alignnet_comm_allreduce_grads(alignnet_handle* h)
{
  if (!h) return 1;
  if (!h->comm) return fail(h, "alignnet_comm_allreduce_grads: communicator not initialised");
  TrainWS* w = tws(h);
  if (!w->grad) return fail(h, "alignnet_comm_allreduce_grads: no gradients computed yet");
  // one bucket: the whole trainable vector (8.66 MB fp32 for the SynthCars widths), ncclFloat = 7, ncclSum = 0
  return comm_allreduce(h, w->grad, h->n_trainable, false, h->stream, "gradient", grad_comm_of(h));
}

// ---------------------------------------------------------------------------------
// checkpoints: own container ("ALN3" + version, step, then name/shape/data records for every variable and
// the optimiser slots) -- replaces tf.train.Saver (train.py:220,252,268,281,317,321)
// ---------------------------------------------------------------------------------
extern "C" int alignnet_save(alignnet_handle* h, const char* path)
{
  if (!h) return 1;
  if (!path) return fail(h, "alignnet_save: null path");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<float> host(h->n_total), m(h->n_trainable, 0.f), v(h->n_trainable, 0.f);
  HIP_TRY(h, hipMemcpy(host.data(), h->d_params, h->n_total * sizeof(float), hipMemcpyDeviceToHost));
  TrainWS* w = tws(h);
  if (w->adam_m) {
    HIP_TRY(h, hipMemcpy(m.data(), w->adam_m, h->n_trainable * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(v.data(), w->adam_v, h->n_trainable * sizeof(float), hipMemcpyDeviceToHost));
  }
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(h, std::string("alignnet_save: cannot open ") + tmp);
  const char magic[8] = {'A', 'L', 'N', '3', 'C', 'K', 'P', '1'};
  const int64_t step = h->step;
  const int32_t nvars = (int32_t)h->params.size();
  bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(&step, 8, 1, f) == 1 && std::fwrite(&nvars, 4, 1, f) == 1;
  for (const ParamInfo& p : h->params) {
    const int32_t len = (int32_t)p.name.size(), dims[3] = {p.rows, p.cols, p.trainable ? 1 : 0};
    ok = ok && std::fwrite(&len, 4, 1, f) == 1 && std::fwrite(p.name.data(), 1, len, f) == (size_t)len && std::fwrite(dims, 4, 3, f) == 3;
    ok = ok && std::fwrite(host.data() + p.offset, 4, p.count(), f) == p.count();
    if (p.trainable) {
      ok = ok && std::fwrite(m.data() + p.offset, 4, p.count(), f) == p.count();
      ok = ok && std::fwrite(v.data() + p.offset, 4, p.count(), f) == p.count();
    }
  }
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) return fail(h, std::string("alignnet_save: write failed for ") + path);
  return 0;
}

extern "C" int alignnet_load(alignnet_handle* h, const char* path, int32_t skip_step)
{
  if (!h) return 1;
  if (!path) return fail(h, "alignnet_load: null path");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(h, std::string("alignnet_load: cannot open ") + path);
  char magic[8]; int64_t step = 0; int32_t nvars = 0;
  bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "ALN3CKP1", 8) == 0 && std::fread(&step, 8, 1, f) == 1 &&
            std::fread(&nvars, 4, 1, f) == 1;
  if (!ok || nvars < 0 || nvars > 1 << 20) { std::fclose(f); return fail(h, "alignnet_load: not an ALN3CKP1 checkpoint"); }
  if (ensure_train_ws(h, 0)) { std::fclose(f); return 1; }
  TrainWS* w = tws(h);
  std::vector<float> host(h->n_total), m(h->n_trainable, 0.f), v(h->n_trainable, 0.f);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(host.data(), h->d_params, h->n_total * sizeof(float), hipMemcpyDeviceToHost));
  int matched = 0;
  for (int i = 0; i < nvars && ok; ++i) {
    int32_t len = 0, dims[3];
    ok = std::fread(&len, 4, 1, f) == 1 && len > 0 && len < 4096;
    std::string name(ok ? len : 0, '\0');
    ok = ok && std::fread(&name[0], 1, len, f) == (size_t)len && std::fread(dims, 4, 3, f) == 3;
    if (!ok) break;
    // a corrupt header must not become a bad_alloc / abort across the C ABI: dims are checked against the graph before allocating
    if (dims[0] <= 0 || dims[1] <= 0 || (int64_t)dims[0] * dims[1] > (int64_t)h->n_total || (dims[2] != 0 && dims[2] != 1)) {
      std::fclose(f);
      return fail(h, "alignnet_load: corrupt header for variable '" + name + "' (dims " + std::to_string(dims[0]) + " x " + std::to_string(dims[1]) + ")");
    }
    const size_t cnt = (size_t)dims[0] * dims[1];
    std::vector<float> buf(cnt), bm, bv;
    ok = std::fread(buf.data(), 4, cnt, f) == cnt;
    if (ok && dims[2]) { bm.resize(cnt); bv.resize(cnt); ok = std::fread(bm.data(), 4, cnt, f) == cnt && std::fread(bv.data(), 4, cnt, f) == cnt; }
    auto it = h->by_name.find(name);
    if (!ok || it == h->by_name.end()) continue;
    const ParamInfo& p = h->params[it->second];
    if (p.count() != cnt) { std::fclose(f); return fail(h, "alignnet_load: shape mismatch for " + name); }
    std::memcpy(host.data() + p.offset, buf.data(), cnt * 4);
    if (p.trainable && dims[2]) { std::memcpy(m.data() + p.offset, bm.data(), cnt * 4); std::memcpy(v.data() + p.offset, bv.data(), cnt * 4); }
    ++matched;
  }
  std::fclose(f);
  if (!ok) return fail(h, "alignnet_load: truncated checkpoint");
  if (matched != (int)h->params.size()) return fail(h, "alignnet_load: checkpoint does not hold every variable of this graph");
  HIP_TRY(h, hipMemcpy(h->d_params, host.data(), h->n_total * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(w->adam_m, m.data(), h->n_trainable * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(w->adam_v, v.data(), h->n_trainable * sizeof(float), hipMemcpyHostToDevice));
  if (!skip_step) h->step = step;   // the pre-training restore excludes `batch` (train.py:278-281)
  h->folded = false;
  return 0;
}
