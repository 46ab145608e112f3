/*
 * alignnet_hip.h -- C ABI of libalignnet_hip.so, the MI355X (gfx950) engine for the
 * AlignNet-3D `tp8` network.
 *
 * The reference (grossjohannes/AlignNet-3D) has no FFI: the seam its hot path sits
 * behind is the Python module API consumed by train.py plus tf.Session.run.  Each
 * entry point below names the reference call site it replaces (paths relative to the
 * reference repository root).  Plain C types only; row-major float32 everywhere.
 *
 * Ownership: the caller owns every buffer it passes in; the library owns device
 * memory, parameters and optimiser state behind the opaque handle.
 * Errors: every call returns 0 on success, non-zero on failure; the message is
 * available from alignnet_last_error().  (The reference asserts / raises and aborts:
 * train.py:55,143,216,251 -- the Python wrapper raises on a non-zero status.)
 * Threading: one handle per GPU/process, calls on a handle are serialised by the
 * caller (the reference issues one blocking sess.run at a time).
 */
#ifndef ALIGNNET_HIP_H
#define ALIGNNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALIGNNET_MAX_WIDTHS 8
#define ALIGNNET_ABI_VERSION 1

typedef struct alignnet_handle alignnet_handle;

/* A conv / fc width list as written in the JSON configs (model.options). */
typedef struct {
  int32_t n;
  int32_t w[ALIGNNET_MAX_WIDTHS];
} alignnet_widths;

/* Everything models/tp8.py reads from `cfg` at graph-build time (tp8.py:10,14,98,154,
 * 307-308,318) plus what train.py:133-174,211-217 reads for the schedules/optimiser. */
typedef struct {
  int32_t abi_version;   /* must be ALIGNNET_ABI_VERSION */
  int32_t device;        /* HIP device ordinal (train.py:189 cfg.gpu_index) */
  int32_t num_points;    /* cfg.model.num_points */
  int32_t num_channels;  /* cfg.data.num_channels (3) */
  int32_t num_bins;      /* cfg.model.angles.num_bins */
  int32_t backbone;      /* 0 = "pointnet", 1 = "dgcnn" (cfg.model.backbone) */
  alignnet_widths s1_conv, s1_fc;   /* options.s1transformer = [conv, [fc, keep]] */
  alignnet_widths s2_conv, s2_fc;   /* options.s2transformer */
  alignnet_widths emb_conv;         /* options.embedding */
  alignnet_widths rem_fc;           /* options.remaining_transform_prediction[0] */
  float s1_keep, s2_keep, rem_keep; /* dropout keep_prob; <= 0 means "None" (tp8.py:80) */
  float angle_factor;               /* options.angle_factor */
  float early_stage_factor;         /* options.early_stage_factor */
  int32_t accept_inverted_angle;    /* cfg.model.angles.accept_inverted_angle */
  /* training (train.py:133-174,211-217) */
  int32_t batch_size;               /* cfg.training.batch_size (global, drives schedules) */
  int32_t ntrain;                   /* cfg.data.ntrain */
  float learning_rate;              /* cfg.training.learning_rate */
  int32_t lr_step;                  /* lr_extension.step */
  float lr_rate;                    /* lr_extension.rate */
  int32_t lr_per_epoch;             /* lr_extension.per == "epoch" */
  float bn_init, bn_rate, bn_clip;  /* bn_extension.{init,rate,clip} */
  int32_t bn_step;                  /* bn_extension.step */
  int32_t bn_per_epoch;             /* bn_extension.per == "epoch" */
  int32_t optimizer;                /* 0 = adam, 1 = momentum (train.py:211-216) */
  float momentum;                   /* cfg.training.optimizer.momentum */
  uint64_t seed;                    /* dropout / init RNG stream */
} alignnet_config;

/* The nine prediction tensors tf.Session.run returns in eval (train.py:448) minus the
 * summary: all float32, caller-allocated, any of them may be NULL to skip it.
 * B rows each; widths: centres/translations 3, logits 2*num_bins. */
typedef struct {
  float* pred_translations;            /* [B,3]   tp8.py:155 */
  float* pred_remaining_angle_logits;  /* [B,2nb] tp8.py:156 */
  float* pred_s1_pc1centers;           /* [B,3]   tp8.py:146 */
  float* pred_s1_pc2centers;           /* [B,3]   tp8.py:147 */
  float* pred_s2_pc1centers;           /* [B,3]   tp8.py:148 */
  float* pred_s2_pc2centers;           /* [B,3]   tp8.py:149 */
  float* pred_pc1angle_logits;         /* [B,2nb] tp8.py:150 */
  float* pred_pc2angle_logits;         /* [B,2nb] tp8.py:151 */
} alignnet_outputs;

/* The six label tensors of placeholder_inputs (tp8.py:16-22). */
typedef struct {
  const float* translations;  /* [B,3] */
  const float* rel_angles;    /* [B,1] */
  const float* pc1_centers;   /* [B,3] */
  const float* pc2_centers;   /* [B,3] */
  const float* pc1_angles;    /* [B,1] */
  const float* pc2_angles;    /* [B,1] */
} alignnet_labels;

/* The scalars a train-mode sess.run returns (train.py:368) + the 16 summaries of
 * tp8.py:336-353 in declaration order + the two schedule scalars (train.py:197,210). */
typedef struct {
  int64_t step;          /* value of `batch` after the update */
  float loss;            /* per_transform_loss, tp8.py:334 */
  float learning_rate;   /* train.py:155 */
  float bn_decay;        /* train.py:173 */
  float summaries[16];   /* tp8.py:336-353 */
} alignnet_step_result;

typedef struct {
  int64_t step;
  float learning_rate;
  float bn_decay;
} alignnet_state;

/* ---- lifetime: replaces graph construction, train.py:190-227 ------------------ */
int alignnet_create(const alignnet_config* cfg, alignnet_handle** out);
void alignnet_destroy(alignnet_handle* h);
/* Message of the most recent failure on this handle (h may be NULL for create()). */
const char* alignnet_last_error(const alignnet_handle* h);
int alignnet_abi_version(void);

/* ---- parameters: replaces sess.run(init) train.py:241 and tf.train.Saver
 *      get/set of individual variables (train.py:220,252,268,281) ----------------- */
int alignnet_init_params(alignnet_handle* h, uint64_t seed);   /* utils/tf_util.py:10-49 */
int alignnet_num_params(const alignnet_handle* h);
/* name: variable name (DESIGN.md section "variables"); rows*cols floats; trainable flag. */
int alignnet_param_info(const alignnet_handle* h, int index, const char** name,
                        int32_t* rows, int32_t* cols, int32_t* trainable);
int alignnet_get_param(alignnet_handle* h, const char* name, float* dst, size_t count);
int alignnet_set_param(alignnet_handle* h, const char* name, const float* src, size_t count);

/* ---- inference: replaces the eval sess.run, train.py:447-449 (is_training=False) -- */
/* Host buffers; any B >= 1 (no padding to a static batch needed, cf. train.py:451-452). */
int alignnet_forward(alignnet_handle* h, const float* pcs1, const float* pcs2, int32_t B,
                     const alignnet_outputs* out);
/* Device buffers already resident in HBM; asynchronous on the handle's stream.
 * Follow with alignnet_synchronize() before reading the outputs. */
int alignnet_forward_device(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2,
                            int32_t B, const alignnet_outputs* d_out);
/* Pipelined form of alignnet_forward for a stream of batches (the evaluation loop of train.py:432-462 issues one blocking
 * sess.run per batch): submit() stages the batch in pinned memory and queues copy-in (copy stream), forward (compute stream) and
 * copy-out (a third stream), then returns; wait() blocks until the OLDEST submitted batch is complete and its outputs are in the
 * `out` buffers that were passed to its submit() (they must stay valid until then; pcs1 / pcs2 may be reused as soon as submit()
 * returns).  At most two batches in flight: the copy-in of batch i + 1 runs under the forward of batch i.  Same results as
 * alignnet_forward, bit for bit.  (alignnet_eval_loss refers to the most recently SUBMITTED batch.) */
int alignnet_forward_submit(alignnet_handle* h, const float* pcs1, const float* pcs2, int32_t B,
                            const alignnet_outputs* out);
int alignnet_forward_wait(alignnet_handle* h);
/* Eval-mode loss on the last forward's predictions (train.py:448 `loss` fetch). */
int alignnet_eval_loss(alignnet_handle* h, const alignnet_labels* labels, int32_t B,
                       float* loss, float summaries[16]);
int alignnet_synchronize(alignnet_handle* h);

/* ---- training: replaces the train sess.run, train.py:368 (is_training=True):
 *      forward with batch statistics, loss, backward, (all-reduce), Adam, EMA, step++ */
/* dropout_u: optional host array of uniforms [0,1) used for the dropout masks, laid out
 * [s1 tower0 | s2 tower0 | s1 tower1 | s2 tower1 | pair head], each B x last-hidden-width;
 * NULL = draw on the device from cfg.seed and the step counter.
 * Shapes: PointNet backbones of any depth (2 .. 6 conv layers, models/tp8.py:49-59), widths multiples of 8.  Three-layer stages with
 * widths multiples of 32, C1, C2 <= 128, C3 <= 1024 (every shipped dataset config) run the fused recompute kernels; any other stage
 * (e.g. the five-layer backbones of configs/default.json) runs the layer-by-layer path (fp32, hidden widths <= 256, last <= 4096);
 * backbone "dgcnn" (models/tp8.py:30-46, k = 20, 20 <= num_points <= 4096): [C1, C2, C3] with C1 in {32, 64}, C2 in {64, 128} (every
 * shipped config) runs the fused edge kernels; any other list of 2 .. 6 widths (multiples of 8, edge convs <= 256) runs the same
 * layer-by-layer path over the B N k edge rows (fp32).
 * Anything else fails with a message (alignnet_last_error), it is never run on a fallback. */
int alignnet_train_step(alignnet_handle* h, const float* pcs1, const float* pcs2,
                        const alignnet_labels* labels, int32_t B, const float* dropout_u,
                        alignnet_step_result* result, const alignnet_outputs* out);
/* Same step with inputs and labels already resident in HBM (device pointers inside d_labels); dropout
 * masks are drawn on the device.  result may be NULL to avoid the host synchronisation. */
int alignnet_train_step_device(alignnet_handle* h, const float* d_pcs1, const float* d_pcs2,
                               const alignnet_labels* d_labels, int32_t B, alignnet_step_result* result);
/* Split form used for data-parallel training: forward+backward only (gradients stay on
 * the device), then the optimiser.  alignnet_grad_buffer exposes the flat gradient
 * (device pointer, float count) for an external all-reduce when the built-in RCCL
 * communicator is not used. */
int alignnet_train_forward_backward(alignnet_handle* h, const float* pcs1, const float* pcs2,
                                    const alignnet_labels* labels, int32_t B,
                                    const float* dropout_u, alignnet_step_result* result,
                                    const alignnet_outputs* out);
int alignnet_grad_buffer(alignnet_handle* h, float** d_grad, size_t* count);
int alignnet_apply_gradients(alignnet_handle* h, float grad_scale);   /* consumes the gradient: the buffer reads zero afterwards */
/* Debug/parity: copy the flat gradient of one variable to the host. */
int alignnet_get_grad(alignnet_handle* h, const char* name, float* dst, size_t count);
/* Debug/parity: the uniforms the device-side dropout stream (tf.nn.dropout's random_uniform, utils/tf_util.py:554-575) draws at
 * the current step counter when dropout_u is NULL, in the layout of dropout_u; count = B * (4 * w_hidden + w_pair_hidden).
 * A step run with these uniforms passed explicitly is bit-identical to the step that draws them itself. */
int alignnet_debug_dropout_uniforms(alignnet_handle* h, int32_t B, float* dst, size_t count);
/* Test hook: the k-nearest-neighbour graph (k = 20, self included; the SET tf.nn.top_k selects, ties at the k-th distance to the
 * lower index; utils/tf_util_dgcnn.py:638-676) the last eval-mode forward of a dgcnn engine built: int32 [2B][num_points][20],
 * tower 1's B clouds first.  count = 2 * B * num_points * 20.  Order within a row: nearest first when the query's candidate list
 * has <= 64 survivors (the usual case); queries with more survivors (clustered / duplicated points) emit their k entries in
 * point-index order -- the max over the k neighbours is order-invariant, so compare rows as sets (sorted) where that can occur. */
int alignnet_debug_knn_graph(alignnet_handle* h, int32_t* dst, size_t count);
/* Test hook: what the last TRAINING forward on this handle DECIDED -- the discontinuous choices of the graph.  A parity test pins
 * the oracle to them (after checking that each one is a maximum of the oracle's own values to within rounding), so that the rest of
 * the comparison is continuous (tests/test_fullsize_gpu.py, oracle/alignnet_torch.py `pinned`).  All int32, towers outermost
 * (tower 1's B clouds, then tower 2's), B = the batch of that call:
 *   ALIGNNET_DECISION_YAW_CLASS   (stage ignored)  [2][B]              decoded yaw class, models/tp8.py:296
 *   ALIGNNET_DECISION_POOL_POINT  stage 0..2       [2][B][C_last]      arg-max point of the max over points, utils/tf_util.py:350-373 / models/tp8.py:58
 *   ALIGNNET_DECISION_EDGE_SLOT   stage 0..2, dgcnn [2][B][N][C_edge]  arg-max neighbour slot of the max over k, models/tp8.py:42
 *   ALIGNNET_DECISION_KNN_GRAPH   (stage ignored), dgcnn [2][B][N][20] the neighbour table the step used (slot = position in the row)
 *   ALIGNNET_DECISION_ANGLE_CLASS stage = loss term 0 (tower 1), 1 (tower 2): [2][B]; 2 (pair): [2][B][B] -- the class tf_angle2class
 *                                 (models/tp8.py:193-199) put each target angle of the loss into, for the target and for target + pi (:286);
 *                                 the pair term's target is the [B, B] matrix of :327, entry (i, j) = labels of row i against the decoded yaws of
 *                                 column j.  The residual label res = sh - centre(class) is a sawtooth in the angle: an entry within a rounding of
 *                                 a class boundary lands on the other tooth in another evaluation (label off by 2 in units of pi / num_bins).
 * count must equal the element count of the requested array. */
#define ALIGNNET_DECISION_YAW_CLASS 0
#define ALIGNNET_DECISION_POOL_POINT 1
#define ALIGNNET_DECISION_EDGE_SLOT 2
#define ALIGNNET_DECISION_KNN_GRAPH 3
#define ALIGNNET_DECISION_ANGLE_CLASS 4
int alignnet_debug_train_decisions(alignnet_handle* h, int32_t kind, int32_t stage, int32_t* dst, size_t count);
/* Test hook: the SIGN every relu of the last training step saw (utils/tf_util.py:167-168,345-346), one byte (0 / 1) per element.  What the
 * decisions above leave undecided are these signs: a pre-activation within one rounding of zero is "on" in one evaluation and "off" in another,
 * and on a row that wins many max-pool channels that re-routes per cents of a weight column's gradient.  A parity test pins the oracle to the
 * masks too (y = bn(z) * mask, after checking that every disagreeing pre-activation is rounding-sized) and the whole step becomes a smooth
 * function of its inputs (oracle/alignnet_torch.py `pinned["relu"]`, tests/test_fullsize_gpu.py).  Towers outermost, as above:
 *   ALIGNNET_RELU_CONV  stage 0..2, layer l of the stage's conv stack (models/tp8.py:30-59):
 *        hidden layers                       [2][B][N][C_l]        (dgcnn edge convs: [2][B][N][20][C_l])
 *        dgcnn, last edge conv (l = n - 2)   [2][B][N][C_l]        at the winning neighbour slot (max and relu commute)
 *        last conv (l = n - 1)               [2][B][C_l]           at the winning point
 *   ALIGNNET_RELU_HEAD  stage 0..2 = s1 head, s2 head, pair head (models/tp8.py:75-82), hidden layer j: [2][B][C_j] ([B][C_j] for the pair head)
 * Stored activations are read back; activations the passes recompute from xyz (conv1 of the fused stages) are recomputed by the same
 * device functions from the frame, weights and batch statistics of that step and its point clouds (after alignnet_train_step_device:
 * the caller's device buffers, which must still hold that batch).  Call it after alignnet_train_forward_backward, before the optimiser
 * changes the parameters (alignnet_apply_gradients) and before the next step.  count must equal the element count
 * of the requested array. */
#define ALIGNNET_RELU_CONV 0
#define ALIGNNET_RELU_HEAD 1
int alignnet_debug_train_relu_mask(alignnet_handle* h, int32_t kind, int32_t stage, int32_t layer, uint8_t* dst, size_t count);
/* Test hook, bf16 step only (train_matmul_bf16, fused stages): the bf16-ROUNDED activations the last training step fed to its MFMA convs, as bf16
 * bits, towers outermost: layer 0 = h1 [2][B][N][C1] (input of the hidden conv; dgcnn: the lifted edge features [2][B][N][20][C1], input of the second edge
 * conv), layer 1 = h2 [2][B][N][C2] (input of the lift; dgcnn: the pooled edge features, input of the point conv).  Every rounding of an
 * operand to bf16 is a small decision of its own (2^-8 of the value, a billion per step): an oracle that models the operand rounding (oracle/alignnet_torch.py
 * `bf16_lift`) and takes THESE rounded values -- after checking each is one of the two bf16 neighbours of its own value -- is a smooth function of its inputs,
 * as with the decisions and signs above.  Same calling window as alignnet_debug_train_relu_mask. */
int alignnet_debug_train_rounded(alignnet_handle* h, int32_t stage, int32_t layer, uint16_t* dst, size_t count);

/* ---- multi-GPU (not in the reference, which is single-device: train.py:189).
 *      One process per GPU; RCCL communicator over xGMI for the gradient all-reduce. */
int alignnet_comm_unique_id(uint8_t id[128]);
/* Id of a new IN-PROCESS loopback group (first bytes "ALN3LOOP"): alignnet_comm_init with such an id joins `world` handles of ONE
 * process on ONE device, each driven by its own host thread, into a communicator whose collectives are stream-ordered device copies /
 * sums with a host rendezvous (csrc/comm_loopback.h) instead of RCCL.  Every multi-rank code path of the engine -- sync_bn's per-layer
 * sums, global_loss's gathers, the bucketed gradient all-reduce -- then runs with DIFFERENT shards on a 1-GPU box (tests/test_loopback_gpu.py:
 * W = 2 / 8 ranks against one engine at the concatenated batch, BASELINE.json configs[3]).  All ranks must issue the same collectives in
 * the same order (as with RCCL); a rank that fails breaks the group, the others return an error instead of waiting. */
int alignnet_comm_loopback_id(uint8_t id[128]);
int alignnet_comm_init(alignnet_handle* h, int32_t rank, int32_t world, const uint8_t id[128]);
/* Optional: a SECOND communicator of the same ranks (its own id, same rank / world as the first) that carries the gradient buckets
 * only.  NCCL-style communicators serialise the collectives issued on them across streams; with sync_bn a per-layer sum of the next
 * stage's backward would otherwise queue behind the previous stage's gradient bucket.  Results are identical with and without it
 * (tests/test_loopback_gpu.py); alignnet_get_option("grad_communicator") reads 1 when it exists.
 * HAZARD (why it is opt-in and off in bench.py): two RCCL communicators then have collectives in flight on two streams of one device at the
 * same time.  NCCL / RCCL document that as deadlock-prone when the device-side execution order of the two collectives differs between ranks
 * (each needs all ranks' kernels resident to progress).  It has run on the in-process loopback backend only, never on real links: validate it
 * on a multi-GPU node before relying on it. */
int alignnet_comm_init_grad(alignnet_handle* h, int32_t rank, int32_t world, const uint8_t id[128]);
int alignnet_comm_allreduce_grads(alignnet_handle* h);
/* Average the BatchNorm EMA shadows (the non-trainable variables) over the ranks, on the device: local-BN data parallelism updates them
 * from each rank's own shard (utils/tf_util.py:476-485); the drop-in train.py calls this once per epoch so that every rank's eval-mode
 * model and rank 0's checkpoint agree.  (Identical already under "sync_bn".) */
int alignnet_comm_average_shadows(alignnet_handle* h);

/* ---- schedule read-back: sess.run([learning_rate, bn_decay]) train.py:298,
 *      sess.run(batch) train.py:261,272 */
int alignnet_get_state(alignnet_handle* h, alignnet_state* st);
int alignnet_set_step(alignnet_handle* h, int64_t step);

/* ---- checkpoints: saver.save / saver.restore, train.py:252,268,281,317,321.
 *      Own container format; skip_step mirrors the pre-training restore
 *      that excludes `batch` (train.py:278-281). */
int alignnet_save(alignnet_handle* h, const char* path);
int alignnet_load(alignnet_handle* h, const char* path, int32_t skip_step);

/* ---- measurement hook for bench.py: HIP-event time (ms) of the dominant kernel
 *      (the fused shared-MLP backbone) accumulated since the last reset, and the
 *      number of launches, measured on the handle's own stream. */
int alignnet_profile_enable(alignnet_handle* h, int32_t on);

/* ---- HBM-resident dataset + device-side batch sampler ------------------------------
 * Replaces provider.load_batch (provider.py:85-136: per-example file opens, np.random.choice(n, N, replace=True)
 * resampling :97-98) and provider.jitter_point_cloud (provider.py:60-71, called at train.py:354-356) for runs that do
 * not need np.random's stream: the packed dataset is uploaded once and a batch is drawn on the device.
 *   points1/points2: all clouds concatenated, [offsets[n][t], 3] float32;  offsets: [n_examples + 1][2] row offsets
 *   (int64, start at 0, non-decreasing; an empty cloud samples as zeros like provider.py:97-98);
 *   labels: [n_examples][12] float32 = translation(3) rel_angle start_position(3) end_position(3) start_angle end_angle
 *   (the meta/<id>.json fields read at provider.py:86-89).  Host pointers; copied, not retained.
 * sample(): rows = example rows (not ids) of the batch; point n of cloud t of example r is a function of
 *   (seed, r, t, n) only.  jitter_sigma <= 0 disables the jitter (evaluation, train.py:434); clip must be > 0 otherwise.
 * batch(): device pointers of the last sampled batch (valid until the next sample()/upload()/destroy()).
 * train_step_dataset = sample + alignnet_train_step_device;  forward_dataset = sample (no jitter) + forward to host. */
int alignnet_dataset_upload(alignnet_handle* h, const float* points1, const float* points2, const int64_t* offsets,
                            const float* labels, int64_t n_examples);
int alignnet_dataset_free(alignnet_handle* h);
int alignnet_dataset_sample(alignnet_handle* h, const int32_t* rows, int32_t B, uint64_t seed, float jitter_sigma,
                            float jitter_clip);
int alignnet_dataset_batch(alignnet_handle* h, const float** d_pcs1, const float** d_pcs2, alignnet_labels* d_labels);
int alignnet_train_step_dataset(alignnet_handle* h, const int32_t* rows, int32_t B, uint64_t seed, float jitter_sigma,
                                float jitter_clip, alignnet_step_result* result);
int alignnet_forward_dataset(alignnet_handle* h, const int32_t* rows, int32_t B, uint64_t seed, const alignnet_outputs* out);

/* ---- ICP refinement of the prediction on the full clouds -------------------------------
 * Replaces icp.icp_p2point (icp.py:69-78: o3.registration_icp, point-to-point, with_constraint=True = rotation about z
 * only, with_scaling=False) as called by the evaluation loop (train.py:463-484: radius 0.1, --its iterations,
 * init = get_mat_angle(pred_translation, pred_angle, pred_s2_pc1center), tp_utils/pointcloud.py:279-289).
 * points1 = source clouds, points2 = target clouds, concatenated; offsets [B + 1][2] as in alignnet_dataset_upload;
 * init / out: [B][16] row-major 4x4 float64 (out maps source onto target: q ~ out * p); fitness / rmse / iterations:
 * [B] each, may be NULL (Open3D's RegistrationResult fields + the number of estimate steps taken).
 * Stops like Open3D: after `its` estimates or when fitness and inlier rmse both change by < 1e-6.
 * _dataset: the clouds of the uploaded dataset (alignnet_dataset_upload) addressed by example rows -- "Careful: Pass
 * full point cloud, not subsampled one" (train.py:469). */
int alignnet_icp_refine(alignnet_handle* h, const float* points1, const float* points2, const int64_t* offsets, int32_t B,
                        const double* init, double radius, int32_t its, double* out, double* fitness, double* rmse,
                        int32_t* iterations);
int alignnet_icp_refine_dataset(alignnet_handle* h, const int32_t* rows, int32_t B, const double* init, double radius,
                                int32_t its, double* out, double* fitness, double* rmse, int32_t* iterations);

/* ---- run-time options with no counterpart in the reference's config surface --------
 * "train_matmul_bf16" (0/1, default 0): training only -- the two MFMA convs of every backbone (the hidden 1x1 conv
 *   and the -> C3 feature lift, 96 % of the step's FLOPs, models/tp8.py:55-57), in the forward and in the backward's
 *   recompute, run on bf16 MFMA with fp32 accumulation (BASELINE.json configs[2]); statistics, pooling, the K = 3 lift,
 *   the heads, gradient accumulation, optimiser state and the eval-mode forward stay fp32.  Backbones outside the fused shape: only
 *   the last conv, when it runs as the fused tail ("train_fused_tail").  With the dgcnn backbone: the edge
 *   conv behind the K = 6 lift and the point conv (forward), and the dense product h1 Q2 of the backward edge pass.
 * "train_fused_tail" (0/1, default 1): training of backbones outside the fused kernels' shape (depth != 3, odd widths; e.g. the
 *   five-layer backbones of configs/default.json): the layers up to the second-to-last run layer by layer, their output is kept once,
 *   and the last layer -- 90 % of such a backbone's FLOPs -- runs on the fused phase-3 / pass-B2 kernels on the stored features
 *   (when the last two widths fit: multiples of 32, <= 128 and <= 1024), so that the [B N, C_last] tensors never exist.  0 = every
 *   layer layer by layer.  Same arithmetic up to summation order.
 * "infer_matmul_bf16x3" (0/1, default 0): eval-mode forward of 3-layer PointNet backbones (every shipped config) -- every
 *   fp32 operand of the two MFMA layers is written x = bf16(x) + bf16(x - bf16(x)) and each product is formed as
 *   x_hi w_hi + x_hi w_lo + x_lo w_hi with fp32 accumulation: three bf16 MFMAs instead of one fp32 MFMA (16x slower on
 *   gfx950).  Outputs stay within the 1e-4 parity bar (measured 3e-6 against the fp64 oracle, like the exact path).
 *   Also covers the DGCNN branch with widths [<= 64, <= 128, C3]; other backbone shapes keep the exact-fp32 kernels.
 * "allreduce_overlap" (0/1, default 1): data-parallel training steps (alignnet_train_step*, communicator initialised) all-reduce the
 *   gradient in three buckets on a side stream -- the stage-3, stage-2 and stage-1 segment of the flat gradient.  A stage's deferred
 *   weight-gradient jobs are flushed right behind that stage's backward, its bucket is issued at once and travels under the next stage's
 *   backward, so that only the last (smallest, 14 % of the vector for the shipped widths) bucket is exposed; 0 = one all-reduce of the
 *   whole vector after the backward.  Same sums either way.  (Without a communicator the three stages' jobs stay one group of five
 *   launches after the whole backward.)
 * "train_dw_side_stream" (0/1, default 0): the weight-gradient jobs only the optimiser waits for are queued per stage on a second stream
 *   (under the next stage's backward; data-parallel: that stage's all-reduce bucket leaves right behind them) instead of as one group
 *   after the whole backward.  Same arithmetic, same summation order.  Slower on one GPU (measured).
 * "train_phase3_tile64" (0/1, default 0): the training forward's last conv layer + max-pool (phase 3) of the shipped widths 64 / 128 on
 *   64-point tiles, two workgroups per CU (rounds 1 - 2) instead of 128-point tiles, one workgroup of eight waves per CU
 *   (csrc/kernels_train_fwd_wide.h: a weight fragment of the lift feeds four row tiles instead of two).  Same lift values bit for bit; the
 *   column sums of h2 are grouped differently.  A/B switch and test hook.
 * "sync_bn" (0/1, default 0): data-parallel training with the reference's single-device BatchNorm semantics at the GLOBAL batch
 *   (utils/tf_util.py:474): every batch sum behind a BatchNorm -- forward moments, the Gram / column-sum matrices of the layer
 *   identities, the backward's (dbeta, dgamma) totals, the heads' row statistics -- is all-reduced over the ranks (RCCL, about thirty
 *   small all-reduces per step) between the kernel that forms this rank's sum and the one that uses it.  Every backbone shape: fused
 *   three-layer stages, the dgcnn branch, and stages that train layer by layer (any depth / widths: two all-reduces per layer forward --
 *   the sum, then the squared differences from the global mean -- and one backward).  0: every rank normalises with its own shard's
 *   statistics ("local BN").
 * "global_loss" (0/1, default 0): the loss and its gradient over the global batch -- the [B, B] broadcast terms (models/tp8.py:279,327)
 *   and the whole-batch tf.cond (:288) couple all samples: end points and labels are all-gathered, every rank evaluates the same
 *   global loss and keeps its rows of the gradient; the gradient all-reduce then sums instead of averaging.  With "sync_bn" a
 *   data-parallel step is the reference's step at batch B x ranks.
 * "sync_bn_emulate_world" (test hook, default 1): without a communicator, stand for this many ranks holding identical shards.
 * "dropout_stream" (default 0): selects one of 2^64 independent device-side dropout streams under the same cfg.seed; data-parallel
 *   ranks set it to their rank so that they do not draw identical masks for their local rows (initialisation stays cfg.seed's).
 * "ab_*" (0/1, default 0): A/B dispatch overrides -- each selects an earlier kernel variant of the SAME arithmetic for same-box comparisons
 *   (results agree up to summation order; tests/test_train_gpu.py runs them against the default): "ab_no_ld_const", "ab_infer_tile64",
 *   "ab_phase2_legacy", "ab_b1_legacy", "ab_b1_fp32", "ab_p3_bf16_generic", "ab_p3_nogram", "ab_no_defer", "ab_dg_sparse",
 *   "ab_no_glue_fold", "ab_gemm_jobs_ksplit", "ab_fc_direct", "ab_fc_no_splitk", "ab_split_tilewise" (csrc/engine.h: AbBit), and "ab_tiles_per_wg" (eval PointNet backbone: point tiles per workgroup, 0 = automatic);
 *   "dg_cloud_parts" (dgcnn training: workgroups per cloud of the edge kernels, 0 = as many as fill the chip at this batch, 1 .. 8 = fixed; results agree up to the grouping of partial sums);
 *   "infer_tile_points" (eval PointNet backbone: points per workgroup tile; 0 = automatic -- 64-point tiles while the 128-point tiling would cover at most half of the
 *   256 CUs (2B x ceil(N / 128) <= 128: B <= 8 at N = 1024; 0.19 -> 0.134 ms per step at B = 1, 0.198 -> 0.142 at B = 8), 128 otherwise; 64 / 128 = fixed; bit-identical outputs);
 *   "pn_cloud_parts" (the same split for the per-cloud kernels both backbones share: phase 2 / the first-layer Gram, phase 3 -- the lift to C3 with its running
 *   arg-extreme and Gram, the parts' extremes folded exactly in tile order -- and passes B2, B1; the chip is full from 2B = 256 (128-point-tile kernels) or 512
 *   clouds, the reference's shipped batch of 128 brings 256; 0 = automatic, 1 .. 8 = fixed).
 *   "ab_mask" (read-only) = the bits that are set; bench.py prints it.  The library reads NO environment variable; result-changing
 *   ablation switches exist only in the separate ablation build (csrc/ablate.h, `make ablate`).
 * Read-only keys (alignnet_get_option; parity tests use them to assert which kernel instantiation ran, since the shipped
 * widths 64 / 128 dispatch to kernels with the widths compiled in):
 * "last_backbone_kernel": ALIGNNET_KERNEL_* of the most recent eval-mode backbone launch;
 * "comm_world": number of ranks of the communicator (0 = none); "comm_buckets": bucket all-reduces the last step issued;
 * "sync_collectives": sync_bn / global_loss collectives (per-layer all-reduces, gathers) the last training step issued;
 * "comm_order": issue order of the last training step's backward, one decimal digit per event: 1..3 = backward of stage 1..3 queued,
 *   4..6 = all-reduce bucket of stage 1..3 issued (data-parallel step with "allreduce_overlap": 362514 -- every bucket leaves before the
 *   next stage's backward is queued);
 * "last_train_kernel": bit mask of the most recent training step -- 1 = compile-time widths (64, 128), 2 = bf16 operands,
 *   4 = dgcnn backbone, 8 = at least one stage ran the general-depth (layer-by-layer) path, 16 = with its last layer on the fused
 *   kernels ("train_fused_tail").
 * Unknown keys fail. */
#define ALIGNNET_KERNEL_POINTNET_FUSED 1            /* pointnet_fused<128>: run-time widths */
#define ALIGNNET_KERNEL_POINTNET_FUSED_64_128 2     /* pointnet_fused<128, 68, 132> */
#define ALIGNNET_KERNEL_POINTNET_FUSED_64_128_K16 3 /* pointnet_fused<128, 68, 132, 16>: the shipped 3-layer shape */
#define ALIGNNET_KERNEL_POINTNET_FUSED_TP64 4       /* pointnet_fused<64> (ALIGNNET_TILE=64) */
#define ALIGNNET_KERNEL_POINTNET_FUSED_64_128_K16_TP64 8 /* pointnet_fused<64, 68, 132, 16>: the shipped shape on 64-point tiles (small batches: "infer_tile_points") */
#define ALIGNNET_KERNEL_POINTNET_SPLIT 5            /* pointnet_split<> */
#define ALIGNNET_KERNEL_POINTNET_SPLIT_64_128 6     /* pointnet_split<64, 128> */
#define ALIGNNET_KERNEL_POINTNET_SPLIT_PERSIST 7    /* pointnet_split_persist: shipped widths, persistent workgroups */
#define ALIGNNET_KERNEL_DGCNN_FUSED 10              /* dgcnn_fused<> */
#define ALIGNNET_KERNEL_DGCNN_FUSED_64_128 11       /* dgcnn_fused<68, 132> */
#define ALIGNNET_KERNEL_DGCNN_SPLIT 12              /* dgcnn_split<> */
#define ALIGNNET_KERNEL_DGCNN_SPLIT_64_128 13       /* dgcnn_split<64, 128> */
int alignnet_set_option(alignnet_handle* h, const char* key, int64_t value);
int alignnet_get_option(alignnet_handle* h, const char* key, int64_t* value);
int alignnet_profile_read(alignnet_handle* h, double* backbone_ms, int64_t* backbone_launches,
                          double* total_ms, int32_t reset);
/* HIP-event time (ms, summed) and launch count of one timed kernel since the last reset (alignnet_profile_read(..., reset = 1)), measured
 * on the stream it is launched on while profiling is enabled.  name: "backbone" (eval-mode fused backbone), "knn",
 * "train_fwd_phase2", "train_fwd_phase3", "train_gram_h2", "train_bwd_b2", "train_bwd_b1", "dg_train_fwd", "dg_train_bwd_edge",
 * "allreduce" (what the compute stream waits for), "optimizer". */
int alignnet_profile_read_kernel(alignnet_handle* h, const char* name, double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNNET_HIP_H */
