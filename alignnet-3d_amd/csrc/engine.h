// Host-side engine state behind the opaque C handle (include/alignnet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <atomic>

#include "../../include/alignnet_hip.h"

namespace alignnet {

struct ParamInfo {
  std::string name;
  int rows, cols;
  bool trainable;
  size_t offset;   // floats into d_params
  size_t count() const { return (size_t)rows * cols; }
};

// One conv / fc layer of the graph (models/tp8.py:101-158 construction order).
struct Layer {
  std::string name;      // scope path below the tower prefix, e.g. "transformer1/embedding/conv1"
  int cin, cout;
  bool bn, siamese, conv, first_conv;
  int fan_in, fan_out;
  int p_w, p_b;          // indices into params
  int p_bn[2][4];        // [set][beta,gamma,moving_mean,moving_var]; -1 when absent
  size_t off_wp;         // packed MFMA weight image (floats into d_wp); first conv: unused
  size_t off_ss;         // scale/shift [2][cout] (floats into d_scale/d_shift)
};

struct Stack { int first, n; };   // range of layers

// Kernels whose launches are bracketed with HIP events on the handle's stream while profiling is enabled
// (alignnet_profile_enable; read back by name through alignnet_profile_read_kernel -- bench.py's roofline legs).
enum ProfKernel { PK_BACKBONE = 0, PK_KNN, PK_TRAIN_PHASE2, PK_TRAIN_PHASE3, PK_TRAIN_GRAM, PK_TRAIN_B2, PK_TRAIN_B1, PK_DG_FWD, PK_DG_BWD_EDGE,
                  PK_ALLREDUCE, PK_OPTIMIZER, PK_COUNT };
static const char* const kProfKernelNames[PK_COUNT] = {"backbone", "knn", "train_fwd_phase2", "train_fwd_phase3", "train_gram_h2", "train_bwd_b2",
                                                       "train_bwd_b1", "dg_train_fwd", "dg_train_bwd_edge", "allreduce", "optimizer"};

struct Workspace {
  int cap = 0;                     // pairs
  float* d_pcs[2] = {nullptr, nullptr};
  int* d_nn = nullptr;             // DGCNN neighbour indices [2*cap][N][20]
  float* d_all = nullptr;          // everything else, carved below
  float *xform, *center_mean, *s1c, *s2c, *theta;
  int* cls;
  float *pool1, *pool2, *emb;      // [2B][C1], [2B][C2], [B][2*Cemb]
  float *hid_a, *hid_b;            // head hidden activations
  float* hid_s;                    // raw (pre-BatchNorm) output of a head layer run as two K halves (atomic adds onto zeros: cleared with the pooled buffers)
  bool hid_s_zeroed = false;       // hid_s has been cleared by this forward's centroid kernel and not been added onto yet (alignnet_api.hip: run_head)
  float *o1, *o2, *o3;             // head outputs [2B][3], [2B][3+2nb], [B][3+2nb]
  float* outs[8];                  // device copies of the 8 prediction tensors
};

}  // namespace alignnet

namespace alignnet {
struct DatasetTables { const float* pts[2]; const long long* off; long long n; };   // device pointers of the uploaded dataset
}
// A/B dispatch overrides (alignnet_set_option "ab_<name>", default 0): each selects another kernel variant of the SAME arithmetic (an
// earlier instantiation kept for same-box comparisons and as a test hook); results agree up to summation order.  The library reads
// no environment variable: these replace the ALIGNNET_* switches of rounds 1 - 3.
enum AbBit : unsigned {
  AB_NO_LD_CONST = 1u << 0,       // eval backbones: run-time LDS strides instead of the instantiations with the shipped widths compiled in
  AB_INFER_TILE64 = 1u << 1,      // eval PointNet backbone on 64-point tiles
  AB_PHASE2_LEGACY = 1u << 2,     // fp32 training: phase 2 computes z2 and its statistics instead of deriving them from Gram(h1)
  AB_B1_LEGACY = 1u << 3,         // pass B1 stores dy1 and pass B0 re-reads it (no Pdy accumulation)
  AB_B1_FP32 = 1u << 4,           // bf16 training: pass B1 on fp32 MFMA
  AB_P3BF16_GENERIC = 1u << 5,    // bf16 phase 3 (64-point tiles) with run-time widths
  AB_P3_NOGRAM = 1u << 6,         // fp32 phase 3 without the fused Gram (gram_h2_kernel runs instead)
  AB_NO_DEFER = 1u << 7,          // weight-gradient jobs launched where their inputs appear instead of as deferred multi-job launches
  AB_DG_SPARSE = 1u << 8,         // bf16 dgcnn training: the index-list edge backward instead of the dense form
  AB_NO_GLUE_FOLD = 1u << 9,      // the glue between the stages' backwards as separate launches
  AB_FC_DIRECT = 1u << 11,        // eval head layers: the A operand read straight into the MFMA layout instead of through the per-wave LDS tile
  AB_FC_NO_SPLITK = 1u << 12,     // eval pair head: the first layer (K = 2048) in one piece on 128 workgroups instead of two K halves on 256
  AB_SPLIT_TILEWISE = 1u << 13,   // split-bf16 eval PointNet backbone: one workgroup per 128-point tile (pointnet_split<64, 128>) instead of the persistent one
  AB_GEMM_JOBS_KSPLIT = 1u << 10, // the deferred weight-gradient products on the K-split 32 x 32 tiles (gemm_small_jobs) instead of the 64 x 64 ones
};
static const struct { const char* key; unsigned bit; } kAbKeys[] = {
  {"ab_no_ld_const", AB_NO_LD_CONST}, {"ab_infer_tile64", AB_INFER_TILE64}, {"ab_phase2_legacy", AB_PHASE2_LEGACY}, {"ab_b1_legacy", AB_B1_LEGACY},
  {"ab_b1_fp32", AB_B1_FP32}, {"ab_p3_bf16_generic", AB_P3BF16_GENERIC}, {"ab_p3_nogram", AB_P3_NOGRAM}, {"ab_no_defer", AB_NO_DEFER},
  {"ab_dg_sparse", AB_DG_SPARSE}, {"ab_no_glue_fold", AB_NO_GLUE_FOLD}, {"ab_gemm_jobs_ksplit", AB_GEMM_JOBS_KSPLIT}, {"ab_fc_direct", AB_FC_DIRECT}, {"ab_fc_no_splitk", AB_FC_NO_SPLITK}, {"ab_split_tilewise", AB_SPLIT_TILEWISE}};
struct alignnet_handle;
bool alignnet_dataset_tables(alignnet_handle* h, alignnet::DatasetTables* out);   // alignnet_dataset.hip; false when none uploaded
int alignnet_drain_profile(alignnet_handle* h);
// options that live with the training / communicator code (alignnet_train.hip); -1 = not one of its keys
int alignnet_train_set_option(alignnet_handle* h, const std::string& key, int64_t value);
int alignnet_train_get_option(alignnet_handle* h, const std::string& key, int64_t* value);   // alignnet_api.hip: read back the pending profiling event pairs (synchronises the stream)

struct alignnet_handle {
  alignnet_config cfg;
  std::vector<alignnet::Layer> layers;
  alignnet::Stack s1_conv, s1_fc, s2_conv, s2_fc, emb_conv, rem_fc;
  std::vector<alignnet::ParamInfo> params;
  std::map<std::string, int> by_name;
  size_t n_trainable = 0, n_total = 0;
  float* d_params = nullptr;       // [trainable | EMA shadows]
  float* d_wp = nullptr;
  float* d_scale = nullptr;
  float* d_shift = nullptr;
  size_t n_wp = 0, n_ss = 0;
  bool infer_split = false;        // alignnet_set_option("infer_matmul_bf16x3"): split-bf16 backbone (kernels_infer_split.h)
  unsigned short* d_wps = nullptr; // split (hi, lo) bf16 weight images of the MFMA conv layers, built on demand
  std::vector<size_t> off_wps;     // per layer, in elements
  bool train_bf16 = false;         // alignnet_set_option("train_matmul_bf16"): bf16 operands for the dominant training GEMMs
  bool fused_tail = true;          // alignnet_set_option("train_fused_tail"): general-depth stages run their last layer on the fused kernels
  bool train_ws_stale = false;     // the training workspace was carved for another setting of fused_tail: re-carve on the next step
  bool folded = false;             // eval-mode scale/shift + packed weights are current
  alignnet::Workspace ws;
  hipStream_t stream = nullptr;
  int64_t step = 0;
  uint64_t dropout_stream = 0;     // alignnet_set_option("dropout_stream"): data-parallel ranks draw different dropout masks from one cfg.seed
  // tf.train.AdamOptimizer's beta1_power / beta2_power: float32 variables multiplied by beta once per step (adam.py _finish);
  // cached for `adam_power_t` applied steps so that a step costs one multiplication (alignnet_apply_gradients)
  float adam_b1p = 1.f, adam_b2p = 1.f;
  int64_t adam_power_t = 0;
  int last_B = 0;
  int last_kernel = 0;             // which backbone instantiation the last eval forward launched (alignnet_get_option "last_backbone_kernel")
  int last_train_B = 0;            // pairs of the last training forward (alignnet_debug_train_decisions)
  int last_train_kernel = 0;       // same for the training step: bit 0 = compile-time widths (64, 128), bit 1 = bf16 operands, bit 2 = dgcnn, bit 3 = general depth, bit 4 = fused tail
  // profiling
  bool prof = false;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double prof_backbone_ms = 0, prof_total_ms = 0;
  int64_t prof_backbone_launches = 0;
  struct ProfEv { int id; hipEvent_t a, b; };
  std::vector<ProfEv> prof_pending;                                  // recorded, not yet read back
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;          // event pairs ready for reuse
  double prof_ms[alignnet::PK_COUNT] = {0};                          // accumulated HIP-event time per timed kernel (PK_*)
  int64_t prof_launches[alignnet::PK_COUNT] = {0};
  // training / multi-GPU state (alignnet_train.hip)
  void* train_ws = nullptr;
  void* dataset_ws = nullptr;      // HBM-resident dataset + batch buffers (alignnet_dataset.hip)
  void* pipe = nullptr;            // pipelined host path: two staging slots, copy streams, events (alignnet_api.hip: alignnet_forward_submit / _wait)
  // seed base of the device-side dropout stream at the current step counter (alignnet_train.hip: bn_args, dropout_uniforms_kernel)
  uint64_t dropout_seed_base() const { return (cfg.seed + dropout_stream * 0xD1B54A32D192ED03ull) * 0x9E3779B97F4A7C15ull + (uint64_t)step * 16; }
  void* comm = nullptr;            // alignnet_train.hip: CommImpl (RCCL communicator or a rank of an in-process loopback group)
  void* comm_grad = nullptr;       // optional second communicator of the same ranks, used by the gradient buckets only (alignnet_comm_init_grad)
  int comm_world = 1, comm_rank = 0;
  // gradient all-reduce in three buckets (stage 3 | stage 2 | stage 1 segment of the flat gradient) on a side stream, each issued as
  // soon as that stage's backward has produced its segment; the optimiser waits for the last one (alignnet_train.hip: comm_bucket)
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [stage] = segment ready on the compute stream; [3] = buckets done
  bool comm_overlap = true;        // alignnet_set_option("allreduce_overlap")
  // weight-gradient side stream: the deferred dW jobs of a stage (only the optimiser and the all-reduce wait for them) run here under the
  // next stage's backward (alignnet_train.hip: flush_deferred); [stage] = that stage's backward done on the compute stream, [3] = all flushed
  hipStream_t side_stream = nullptr;
  hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int dw_side = 0;                 // alignnet_set_option("train_dw_side_stream"): 0 = off, 1 = every stage, 2 = stage 3 only
  unsigned ab = 0;                 // AbBit mask (alignnet_set_option "ab_*")
  int infer_tile_opt = 0;          // "infer_tile_points": eval PointNet backbone tile (0 = from the batch: alignnet_api.hip infer_tile_pts; 64 / 128 fixed)
  int dg_parts_opt = 0;            // "dg_cloud_parts": dgcnn training, workgroups per cloud of the edge kernels (0 = chosen from the batch: alignnet_train.hip dg_parts)
  int pn_parts_opt = 0;            // "pn_cloud_parts": PointNet training, workgroups per cloud of phase 2 / passes B2, B1 (0 = chosen from the batch: alignnet_train.hip pn_parts)
  int ab_tiles_per_wg = 0;         // "ab_tiles_per_wg": eval PointNet backbone, point tiles per workgroup (0 = chosen from the grid size)
  int ablate_mutation = 0;         // ablation build only ("ablate_mutation"): deliberately wrong multi-rank arithmetic, to show that tests/test_loopback_gpu.py catches it
  int ablate_dbg = 0;              // ablation build only (-DALIGNNET_ABLATE): the kernels' result-changing timing switches, from ALIGNNET_DBG
  bool p3_tile64 = false;          // alignnet_set_option("train_phase3_tile64"): the forward's phase 3 on 64-point tiles (default: 128-point tiles, kernels_train_fwd_wide.h)
  // alignnet_set_option("sync_bn"): training-mode BatchNorm statistics (and the backward's batch sums) over ALL data-parallel ranks --
  // the reference's single-device semantics at the global batch (utils/tf_util.py:474) -- instead of per rank.
  bool sync_bn = false;
  // alignnet_set_option("global_loss"): the loss (and its gradient) over the GLOBAL batch -- the [B, B] broadcast terms of models/tp8.py:279,327
  // and the whole-batch tf.cond (:288) couple all samples -- from the all-gathered end points and labels; gradients are then summed, not averaged
  bool global_loss = false;
  int sync_emulate_world = 1;      // test hook ("sync_bn_emulate_world"): without a communicator, every BN sum is multiplied by this many
                                   // identical virtual ranks (a step must then reproduce the plain local-BN step on the same shard)
  double* sync_buf = nullptr;      // staging for the per-layer totals that travel through the all-reduce
  int comm_buckets = 0;            // bucket all-reduces issued by the last training step (0: one all-reduce after the backward)
  int sync_collectives = 0;        // sync_bn / global_loss collectives (all-reduces of per-layer sums, gathers) the last training step issued ("sync_collectives")
  long long comm_order = 0;        // last training step, one decimal digit per event in issue order: 1..3 = backward of stage 1..3 queued, 4..6 = bucket of stage 1..3 issued
  mutable std::string err;
};

namespace alignnet {
// hipFuncSetAttribute (dynamic LDS size) applies to the CURRENT device only: every call site remembers per device ordinal whether it
// has been done, so that a second handle on another GPU of the same process sets its own attributes (a process-global flag skipped them).
// Usage: `if (once.need(dev)) { ...attribute calls (HIP_TRY returns on failure)...; once.mark(dev); }` -- the bit is set only after
// every call succeeded, so a failed attempt is retried (and reported) by the next call instead of surfacing later as an opaque
// LDS-size launch failure.  Atomic: handles on different GPUs may be driven from different threads.  Ordinals >= 64 are never
// remembered (the attributes are then set on every call, which is harmless).
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  bool need(int device) const { return device < 0 || device >= 64 || !(done.load(std::memory_order_acquire) & (1ull << device)); }
  void mark(int device) { if (device >= 0 && device < 64) done.fetch_or(1ull << device, std::memory_order_release); }
};

// Kernel timers (only while h->prof is set), two forms:
//  - bind = false: brackets everything launched on h->stream during its lifetime with one recorded event pair.  Each record is a marker
//    packet of its own in the queue (measured: ~5 us of bubble per event between dependent kernels);
//  - bind = true: the scope only holds the pair, and the ONE kernel launched inside it with TIMED_LAUNCH carries the events on its own
//    dispatch (hipExtLaunchKernelGGL start / stop events: the timestamps of the dispatch packet's completion signal, no extra packet).
//    What is measured is then the kernel's own duration, as rocprofv3 --kernel-trace reports it.
struct ProfScope {
  alignnet_handle* h; int id; hipEvent_t a = nullptr, b = nullptr; bool bind, used = false;
  ProfScope(alignnet_handle* h_, int id_, bool bind_ = false) : h(h_), id(id_), bind(bind_)
  {
    if (!h->prof) return;
    if (!h->prof_pool.empty()) { a = h->prof_pool.back().first; b = h->prof_pool.back().second; h->prof_pool.pop_back(); }
    else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    if (!bind) hipEventRecord(a, h->stream);
  }
  ~ProfScope()
  {
    if (!a) return;
    if (bind && !used) { h->prof_pool.push_back({a, b}); return; }   // nothing was launched inside: the pair holds no (or stale) timestamps
    if (!bind) hipEventRecord(b, h->stream);
    h->prof_pending.push_back(alignnet_handle::ProfEv{id, a, b});
  }
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
};
// the one kernel of a bind-mode ProfScope named `prof_scope` (with profiling off the events are null: a plain launch)
#define TIMED_LAUNCH(kernel, grid, block, lds, ...) \
  do { prof_scope.used = true; hipExtLaunchKernelGGL(kernel, grid, block, lds, h->stream, prof_scope.a, prof_scope.b, 0, __VA_ARGS__); } while (0)
}  // namespace alignnet
