"""Engine: the session-like object the drop-in train.py drives.

It plays the role of the TF graph + Session of the reference: construction =
train.py:190-227 (placeholders, get_model, get_loss, optimizer), `forward` = the eval
`sess.run` (train.py:447-449), `train_step` = the train `sess.run` (train.py:368).
All arithmetic happens in libalignnet_hip.so on the GPU; this file only marshals."""
import ctypes as C

import numpy as np

from . import _capi

OUTPUT_NAMES = ("pred_translations", "pred_remaining_angle_logits", "pred_s1_pc1centers", "pred_s1_pc2centers",
                "pred_s2_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits")

SUMMARY_NAMES = (
    "losses/translation", "losses/angle",
    "losses_stages/stage1_pc1_transl_loss", "losses_stages/stage1_pc2_transl_loss",
    "losses_stages/stage2_pc1_transl_loss", "losses_stages/stage2_pc2_transl_loss",
    "losses_stages/stage3_transl_loss",
    "losses_stages/stage2_pc1_angle_loss", "losses_stages/stage2_pc1_angle_class_loss",
    "losses_stages/stage2_pc1_angle_residual_loss",
    "losses_stages/stage2_pc2_angle_loss", "losses_stages/stage2_pc2_angle_class_loss",
    "losses_stages/stage2_pc2_angle_residual_loss",
    "losses_stages/stage3_angle_loss", "losses_stages/stage3_angle_class_loss",
    "losses_stages/stage3_angle_residual_loss")  # models/tp8.py:336-353 order


# include/alignnet_hip.h ALIGNNET_KERNEL_*: which backbone instantiation an eval forward launched (get_option("last_backbone_kernel"))
KERNEL_IDS = {"pointnet_fused": 1, "pointnet_fused<64,128>": 2, "pointnet_fused<64,128,k16>": 3, "pointnet_fused<tp64>": 4,
              "pointnet_split": 5, "pointnet_split<64,128>": 6, "pointnet_split_persist": 7, "pointnet_fused<64,128,k16,tp64>": 8, "dgcnn_fused": 10, "dgcnn_fused<64,128>": 11,
              "dgcnn_split": 12, "dgcnn_split<64,128>": 13}
KERNEL_NAMES = {v: k for k, v in KERNEL_IDS.items()}


class EngineError(RuntimeError):
    pass


def default_model_config():
    """The SynthCars layer widths (reference configs/SynthCars.json:12-15) at N=1024 -- the
    operating point of BASELINE.json.  Same nesting as the merged reference config."""
    return {
        "data": {"num_channels": 3, "ntrain": 0},
        "model": {
            "backbone": "pointnet", "num_points": 1024,
            "options": {
                "angle_factor": 1.0, "early_stage_factor": 0.5,
                "s1transformer": [[64, 128, 256], [[512, 256], 0.7]],
                "s2transformer": [[64, 128, 512], [[512, 256], 0.7]],
                "embedding": [64, 128, 1024],
                "remaining_transform_prediction": [[512, 256], 0.7],
            },
            "angles": {"num_bins": 50, "accept_inverted_angle": True},
        },
        "training": {
            "batch_size": 128, "learning_rate": 0.005,
            "optimizer": {"optimizer": "adam"},
            "lr_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5},
            "bn_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5, "init": 0.5, "clip": 0.99},
        },
        "gpu_index": 0,
    }


def _to_dict(cfg):
    """Accept a plain dict or the config NameSpace (config.py:9-29)."""
    if isinstance(cfg, dict):
        return cfg
    out = {}
    for k, v in cfg.__dict__.items():
        out[k] = _to_dict(v) if hasattr(v, "__dict__") and not isinstance(v, (list, tuple, str)) else v
    return out


def make_c_config(cfg, device=None, seed=0):
    d = _to_dict(cfg)
    m, o, t = d["model"], d["model"]["options"], d.get("training", {})
    c = _capi.Config()
    c.abi_version = _capi.ABI_VERSION
    c.device = int(d.get("gpu_index", 0) if device is None else device)
    c.num_points = int(m["num_points"])
    c.num_channels = int(d["data"]["num_channels"])
    c.num_bins = int(m["angles"]["num_bins"])
    bk = m["backbone"]
    if bk not in ("pointnet", "dgcnn"):
        raise AssertionError(bk)  # models/tp8.py:68
    c.backbone = 0 if bk == "pointnet" else 1
    c.s1_conv, c.s1_fc = _capi.Widths.of(o["s1transformer"][0]), _capi.Widths.of(o["s1transformer"][1][0])
    c.s2_conv, c.s2_fc = _capi.Widths.of(o["s2transformer"][0]), _capi.Widths.of(o["s2transformer"][1][0])
    c.emb_conv = _capi.Widths.of(o["embedding"])
    c.rem_fc = _capi.Widths.of(o["remaining_transform_prediction"][0])
    keep = lambda v: -1.0 if v is None else float(v)
    c.s1_keep, c.s2_keep = keep(o["s1transformer"][1][1]), keep(o["s2transformer"][1][1])
    c.rem_keep = keep(o["remaining_transform_prediction"][1])
    c.angle_factor, c.early_stage_factor = float(o["angle_factor"]), float(o["early_stage_factor"])
    c.accept_inverted_angle = int(bool(m["angles"]["accept_inverted_angle"]))
    c.batch_size = int(t.get("batch_size", 1))
    c.ntrain = int(d["data"].get("ntrain", 0))
    c.learning_rate = float(t.get("learning_rate", 0.001))
    lr, bn = t.get("lr_extension", {}), t.get("bn_extension", {})
    c.lr_step, c.lr_rate = int(lr.get("step", 0)), float(lr.get("rate", 1.0))
    c.lr_per_epoch = int(lr.get("per", "epoch") == "epoch")
    c.bn_init, c.bn_rate, c.bn_clip = float(bn.get("init", 0.5)), float(bn.get("rate", 0.5)), float(bn.get("clip", 0.99))
    c.bn_step, c.bn_per_epoch = int(bn.get("step", 0)), int(bn.get("per", "epoch") == "epoch")
    opt = t.get("optimizer", {}).get("optimizer", "adam")
    if opt not in ("adam", "momentum"):
        raise AssertionError("Invalid optimizer")  # train.py:216
    c.optimizer = 0 if opt == "adam" else 1
    c.momentum = float(t.get("optimizer", {}).get("momentum", 0.9))
    # the engine implements loss "separate" with hard angle classes (every shipped config); anything else must not train silently
    loss = t.get("loss", {})
    if isinstance(loss, dict):
        if loss.get("loss", "separate") != "separate":
            raise AssertionError("training.loss.loss=%r: only the 'separate' loss is built (models/tp8.py:401-407; 'p2p' is never "
                                 "enabled by a shipped config, SURVEY 8.A4)" % loss.get("loss"))
        if (loss.get("options", {}) or {}).get("soft_angle_classes", False):
            raise AssertionError("training.loss.options.soft_angle_classes is not built (models/tp8.py:253-274; no shipped config enables it)")
    c.seed = int(seed)
    return c


def _fp(a):
    return a.ctypes.data_as(_capi.FP)


class Engine:
    def __init__(self, cfg=None, device=None, seed=0):
        self._lib = _capi.load_library()
        self._h = _capi.H()
        self.cfg = _to_dict(cfg) if cfg is not None else default_model_config()
        self._c = make_c_config(self.cfg, device, seed)
        if self._lib.alignnet_create(C.byref(self._c), C.byref(self._h)) != 0:
            raise EngineError(self._lib.alignnet_last_error(None).decode())
        self.num_points = self._c.num_points
        self.num_bins = self._c.num_bins
        self._inflight = []   # output arrays of submitted, not yet waited-for batches (forward_submit / forward_wait)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.alignnet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(self._lib.alignnet_last_error(self._h).decode())

    # ---- variables ---------------------------------------------------------
    def variables(self):
        """[(name, (rows, cols), trainable)] in graph-construction order."""
        out = []
        name, r, c, t = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
        for i in range(self._lib.alignnet_num_params(self._h)):
            self._check(self._lib.alignnet_param_info(self._h, i, C.byref(name), C.byref(r), C.byref(c), C.byref(t)))
            out.append((name.value.decode(), (r.value, c.value), bool(t.value)))
        return out

    def get_variable(self, name):
        shp = dict((n, s) for n, s, _ in self.variables())[name]
        a = np.empty(shp[0] * shp[1], np.float32)
        self._check(self._lib.alignnet_get_param(self._h, name.encode(), _fp(a), a.size))
        return a.reshape(shp) if shp[0] > 1 else a

    def set_variable(self, name, value):
        a = np.ascontiguousarray(value, np.float32).ravel()
        self._check(self._lib.alignnet_set_param(self._h, name.encode(), _fp(a), a.size))

    def set_variables(self, d):
        for k, v in d.items():
            self.set_variable(k, v)

    def init_variables(self, seed=0):
        self._check(self._lib.alignnet_init_params(self._h, seed))

    # ---- eval sess.run (train.py:447-449) -----------------------------------
    def _alloc_outputs(self, B):
        nb2 = 2 * self.num_bins
        widths = dict(pred_translations=3, pred_remaining_angle_logits=nb2, pred_s1_pc1centers=3, pred_s1_pc2centers=3,
                      pred_s2_pc1centers=3, pred_s2_pc2centers=3, pred_pc1angle_logits=nb2, pred_pc2angle_logits=nb2)
        arrs = {k: np.empty((B, widths[k]), np.float32) for k in OUTPUT_NAMES}
        o = _capi.Outputs()
        for k in OUTPUT_NAMES:
            setattr(o, k, _fp(arrs[k]))
        return arrs, o

    def _check_pcs(self, pcs1, pcs2):
        # the feed is float64 in the reference (provider.py:110-119) and cast to the float32 placeholder
        p1 = np.ascontiguousarray(pcs1, np.float32)
        p2 = np.ascontiguousarray(pcs2, np.float32)
        if p1.ndim != 3 or p1.shape != p2.shape or p1.shape[1] != self.num_points or p1.shape[2] != 3:
            raise ValueError(f"pcs must be [B,{self.num_points},3] and equal-shaped, got {p1.shape} and {p2.shape}")
        return p1, p2

    def forward(self, pcs1, pcs2):
        p1, p2 = self._check_pcs(pcs1, pcs2)
        arrs, o = self._alloc_outputs(p1.shape[0])
        self._check(self._lib.alignnet_forward(self._h, _fp(p1), _fp(p2), p1.shape[0], C.byref(o)))
        return arrs

    # ---- pipelined host path: up to two batches in flight, the copy-in of batch i + 1 under the forward of batch i ----
    def forward_submit(self, pcs1, pcs2):
        """Queue one batch (returns at once); the matching forward_wait() returns its outputs.  At most two submits without a wait."""
        p1, p2 = self._check_pcs(pcs1, pcs2)
        arrs, o = self._alloc_outputs(p1.shape[0])
        self._check(self._lib.alignnet_forward_submit(self._h, _fp(p1), _fp(p2), p1.shape[0], C.byref(o)))
        self._inflight.append(arrs)   # (the C side writes into these arrays at wait(): keep them alive)

    def forward_wait(self):
        if not self._inflight:
            raise EngineError("forward_wait: no batch in flight")
        self._check(self._lib.alignnet_forward_wait(self._h))
        return self._inflight.pop(0)

    def forward_stream(self, batches):
        """Generator over an iterable of (pcs1, pcs2): yields each batch's outputs in order, with two batches in flight."""
        pending = 0
        for pcs1, pcs2 in batches:
            if pending == 2:
                yield self.forward_wait()
                pending -= 1
            self.forward_submit(pcs1, pcs2)
            pending += 1
        while pending:
            yield self.forward_wait()
            pending -= 1

    def forward_device(self, d_pcs1, d_pcs2, B, d_out_ptrs=None):
        """Device pointers (ints); d_out_ptrs: dict name -> device pointer, or None to skip copies out."""
        o = _capi.Outputs()
        for k in OUTPUT_NAMES:
            p = (d_out_ptrs or {}).get(k)
            setattr(o, k, C.cast(C.c_void_p(p), _capi.FP) if p else None)
        self._check(self._lib.alignnet_forward_device(self._h, C.c_void_p(d_pcs1), C.c_void_p(d_pcs2), B, C.byref(o)))

    def synchronize(self):
        self._check(self._lib.alignnet_synchronize(self._h))

    # ---- loss on the last eval forward (train.py:448 `loss` fetch) --------------
    def _labels(self, labels, B):
        keys = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
        widths = (3, 1, 3, 3, 1, 1)
        arrs, L = [], _capi.Labels()
        for k, w in zip(keys, widths):
            a = np.ascontiguousarray(labels[k], np.float32).reshape(B, w)
            arrs.append(a)
            setattr(L, k, _fp(a))
        return arrs, L

    def eval_loss(self, labels, B):
        keep, L = self._labels(labels, B)
        loss, summ = C.c_float(), (C.c_float * 16)()
        self._check(self._lib.alignnet_eval_loss(self._h, C.byref(L), B, C.byref(loss), summ))
        return loss.value, dict(zip(SUMMARY_NAMES, list(summ)))

    # ---- train sess.run (train.py:368) -------------------------------------------
    def _train_call(self, fn, pcs1, pcs2, labels, dropout_u):
        p1, p2 = self._check_pcs(pcs1, pcs2)
        B = p1.shape[0]
        keep, L = self._labels(labels, B)
        arrs, o = self._alloc_outputs(B)
        res = _capi.StepResult()
        u = None
        if dropout_u is not None:
            u = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).ravel() for x in dropout_u]), np.float32)
        self._check(fn(self._h, _fp(p1), _fp(p2), C.byref(L), B, _fp(u) if u is not None else None, C.byref(res), C.byref(o)))
        out = dict(step=res.step, loss=res.loss, learning_rate=res.learning_rate, bn_decay=res.bn_decay,
                   summaries=dict(zip(SUMMARY_NAMES, list(res.summaries))))
        out.update(arrs)
        return out

    def train_forward_backward(self, pcs1, pcs2, labels, dropout_u=None):
        """Forward (batch statistics, EMA update), loss and backward; gradients stay on the device.
        dropout_u: optional list [s1 tower0, s2 tower0, s1 tower1, s2 tower1, pair head] of uniforms."""
        return self._train_call(self._lib.alignnet_train_forward_backward, pcs1, pcs2, labels, dropout_u)

    def train_step(self, pcs1, pcs2, labels, dropout_u=None):
        return self._train_call(self._lib.alignnet_train_step, pcs1, pcs2, labels, dropout_u)

    def train_step_device(self, d_pcs1, d_pcs2, d_labels, B, want_result=False):
        """d_labels: dict name -> device pointer (int)."""
        L = _capi.Labels()
        for k in ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles"):
            setattr(L, k, C.cast(C.c_void_p(d_labels[k]), _capi.FP))
        res = _capi.StepResult()
        self._check(self._lib.alignnet_train_step_device(self._h, C.c_void_p(d_pcs1), C.c_void_p(d_pcs2), C.byref(L), B,
                                                         C.byref(res) if want_result else None))
        return dict(step=res.step, loss=res.loss) if want_result else None

    # ---- HBM-resident dataset + device sampler (include/alignnet_hip.h; replaces provider.load_batch + jitter) ----
    def upload_dataset(self, points1, points2, offsets, labels):
        """points*: [sum n, 3]; offsets: [n_examples + 1, 2] int64 row offsets; labels: [n_examples, 12]
        (alignnet3d/packed.py layout).  Copied to HBM once."""
        p1 = np.ascontiguousarray(points1, np.float32).reshape(-1, 3)
        p2 = np.ascontiguousarray(points2, np.float32).reshape(-1, 3)
        off = np.ascontiguousarray(offsets, np.int64).reshape(-1, 2)
        lab = np.ascontiguousarray(labels, np.float32).reshape(off.shape[0] - 1, 12)
        assert off[-1, 0] == p1.shape[0] and off[-1, 1] == p2.shape[0], "offsets do not cover the point blobs"
        self._check(self._lib.alignnet_dataset_upload(self._h, _fp(p1), _fp(p2), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                      _fp(lab), off.shape[0] - 1))

    @staticmethod
    def _rows(rows):
        r = np.ascontiguousarray(rows, np.int32).ravel()
        return r, r.ctypes.data_as(C.POINTER(C.c_int32))

    def sample_batch(self, rows, seed, jitter_sigma=0.0, jitter_clip=0.05):
        """Draw the batch on the device; returns (d_pcs1, d_pcs2, {label name: device pointer})."""
        r, rp = self._rows(rows)
        self._check(self._lib.alignnet_dataset_sample(self._h, rp, r.size, int(seed), float(jitter_sigma), float(jitter_clip)))
        p1, p2, L = C.c_void_p(), C.c_void_p(), _capi.Labels()
        self._check(self._lib.alignnet_dataset_batch(self._h, C.byref(p1), C.byref(p2), C.byref(L)))
        names = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
        return p1.value, p2.value, {k: C.cast(getattr(L, k), C.c_void_p).value for k in names}

    def train_step_rows(self, rows, seed, jitter_sigma=0.01, jitter_clip=0.05):
        """sample (with the reference's jitter defaults, provider.py:60) + full training step, no host batch."""
        r, rp = self._rows(rows)
        res = _capi.StepResult()
        self._check(self._lib.alignnet_train_step_dataset(self._h, rp, r.size, int(seed), float(jitter_sigma), float(jitter_clip),
                                                          C.byref(res)))
        return dict(step=res.step, loss=res.loss, learning_rate=res.learning_rate, bn_decay=res.bn_decay,
                    summaries=dict(zip(SUMMARY_NAMES, list(res.summaries))))

    def forward_rows(self, rows, seed):
        r, rp = self._rows(rows)
        arrs, o = self._alloc_outputs(r.size)
        self._check(self._lib.alignnet_forward_dataset(self._h, rp, r.size, int(seed), C.byref(o)))
        return arrs

    # ---- ICP refinement on the full clouds (icp.py:69-78 / train.py:463-484) ------
    @staticmethod
    def _icp_bufs(inits, B):
        init = np.ascontiguousarray(inits, np.float64).reshape(B, 16)
        out = np.empty((B, 16), np.float64)
        fit, rmse, its = np.empty(B, np.float64), np.empty(B, np.float64), np.empty(B, np.int32)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        return init, out, fit, rmse, its, dp

    def icp_refine(self, sources, targets, inits, radius=0.1, its=30):
        """sources / targets: lists of [n, 3] arrays; inits: [B, 4, 4].  Returns dict(transforms, fitness, rmse, iterations)."""
        B = len(sources)
        off = np.zeros((B + 1, 2), np.int64)
        off[1:, 0] = np.cumsum([len(s) for s in sources]); off[1:, 1] = np.cumsum([len(t) for t in targets])
        cat = lambda L: np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).reshape(-1, 3) for x in L], 0)) if L else np.zeros((0, 3), np.float32)
        p1, p2 = cat(sources), cat(targets)
        init, out, fit, rmse, it, dp = self._icp_bufs(inits, B)
        self._check(self._lib.alignnet_icp_refine(self._h, _fp(p1), _fp(p2), off.ctypes.data_as(C.POINTER(C.c_int64)), B, dp(init),
                                                  float(radius), int(its), dp(out), dp(fit), dp(rmse), it.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(transforms=out.reshape(B, 4, 4), fitness=fit, rmse=rmse, iterations=it)

    def icp_refine_rows(self, rows, inits, radius=0.1, its=30):
        """Same on the clouds of the uploaded dataset (upload_dataset), addressed by example rows."""
        r, rp = self._rows(rows)
        init, out, fit, rmse, it, dp = self._icp_bufs(inits, r.size)
        self._check(self._lib.alignnet_icp_refine_dataset(self._h, rp, r.size, dp(init), float(radius), int(its), dp(out), dp(fit), dp(rmse),
                                                          it.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(transforms=out.reshape(r.size, 4, 4), fitness=fit, rmse=rmse, iterations=it)

    @staticmethod
    def read_device(ptr, count, dtype=np.float32):
        """Debug / test helper: copy `count` elements from a device pointer (synchronous hipMemcpy)."""
        hip = C.CDLL("libamdhip64.so")
        out = np.empty(count, dtype)
        rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2))
        if rc != 0:
            raise EngineError("hipMemcpy device->host failed (%d)" % rc)
        return out

    def apply_gradients(self, grad_scale=1.0):
        self._check(self._lib.alignnet_apply_gradients(self._h, float(grad_scale)))

    def get_gradient(self, name):
        shp = dict((n, s) for n, s, _ in self.variables())[name]
        a = np.empty(shp[0] * shp[1], np.float32)
        self._check(self._lib.alignnet_get_grad(self._h, name.encode(), _fp(a), a.size))
        return a.reshape(shp) if shp[0] > 1 else a

    def debug_dropout_uniforms(self, B):
        """The uniforms the device-side dropout stream draws at the current step (layout of `dropout_u`): list of five
        [B, width] arrays [s1 tower0, s2 tower0, s1 tower1, s2 tower1, pair head]."""
        o = self.cfg["model"]["options"]
        w12, w3 = int(o["s1transformer"][1][0][-1]), int(o["remaining_transform_prediction"][0][-1])
        a = np.empty(B * (4 * w12 + w3), np.float32)
        self._check(self._lib.alignnet_debug_dropout_uniforms(self._h, B, _fp(a), a.size))
        return [a[i * B * w12:(i + 1) * B * w12].reshape(B, w12) for i in range(4)] + [a[4 * B * w12:].reshape(B, w3)]

    def debug_knn_graph(self, B):
        """The k = 20 neighbour graph the last eval-mode forward (of B pairs) built, dgcnn engines only: int32
        [2, B, num_points, 20] (tower, pair, point, neighbour rank; utils/tf_util_dgcnn.py:638-676)."""
        n = self.num_points
        a = np.empty((2, B, n, 20), np.int32)
        self._check(self._lib.alignnet_debug_knn_graph(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), a.size))
        return a

    def debug_train_decisions(self, B, relu=False):
        """What the last training forward (of B pairs) decided -- the graph's discontinuous choices, for decision-pinned parity tests
        (include/alignnet_hip.h: alignnet_debug_train_decisions).  dict: "yaw" int32 [2, B]; "pool" list over the three stages of
        [2, B, C_last]; dgcnn engines also "slot" list of [2, B, N, C_edge] and "knn" [2, B, N, 20].  Tower outermost.
        relu=True adds "relu": the sign every relu of that step saw (alignnet_debug_train_relu_mask), keyed as oracle/alignnet_torch.py keys
        its layers -- "<tower>:<scope>/conv<l>" bool [B*N(*k), C] (the conv in front of a max: at the winner), "<tower>:<scope>/fc<j>" [B, C],
        "p:fc<j>" for the pair head.  Call it right after train_forward_backward (the masks of recomputed layers are rebuilt from the
        step's parameters and statistics)."""
        o, n = self.cfg["model"]["options"], self.num_points
        convs = [list(o["s1transformer"][0]), list(o["s2transformer"][0]), list(o["embedding"])]
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))

        def get(kind, stage, shape):
            a = np.empty(shape, np.int32)
            self._check(self._lib.alignnet_debug_train_decisions(self._h, kind, stage, ip(a), a.size))
            return a
        out = {"yaw": get(0, 0, (2, B)), "pool": [get(1, s, (2, B, convs[s][-1])) for s in range(3)]}
        if self.cfg["model"]["backbone"] == "dgcnn":
            out["slot"] = [get(2, s, (2, B, n, convs[s][-2])) for s in range(3)]
            out["knn"] = get(3, 0, (2, B, n, 20))
        if relu:
            out["relu"] = self.debug_train_relu_masks(B)
            # (with the signs: the classes of the loss's target angles, models/tp8.py:193-199 -- [2, B] per tower term, [2, B, B] for the pair term)
            out["loss_cls"] = [get(4, 0, (2, B)), get(4, 1, (2, B)), get(4, 2, (2, B, B))]
        return out

    def debug_train_rounded(self, B):
        """bf16 step, fused PointNet stages: the bf16-rounded h1 / h2 the step's MFMA convs multiplied (alignnet_debug_train_rounded), keyed like the relu masks
        by the conv layer that CONSUMES them: "<tower>:<scope>/conv2" -> h1 as float32 [B*N, C1], "<tower>:<scope>/conv3" -> h2 [B*N, C2]."""
        o, n = self.cfg["model"]["options"], self.num_points
        convs = [list(o["s1transformer"][0]), list(o["s2transformer"][0]), list(o["embedding"])]
        scopes = ["transformer1/embedding", "transformer2/embedding", "embedding"]
        out = {}
        dg = self.cfg["model"]["backbone"] == "dgcnn"
        for s in range(3):
            for l in range(2):
                a = np.empty((2, B * n * (20 if dg and l == 0 else 1), convs[s][l]), np.uint16)
                self._check(self._lib.alignnet_debug_train_rounded(self._h, s, l, a.ctypes.data_as(C.POINTER(C.c_uint16)), a.size))
                f = (a.astype(np.uint32) << 16).view(np.float32)
                for t in range(2):
                    out[f"{t}:{scopes[s]}/conv{l + 2}"] = f[t]
        return out

    def debug_train_relu_masks(self, B):
        o, n = self.cfg["model"]["options"], self.num_points
        dg = self.cfg["model"]["backbone"] == "dgcnn"
        convs = [list(o["s1transformer"][0]), list(o["s2transformer"][0]), list(o["embedding"])]
        fcs = [list(o["s1transformer"][1][0]), list(o["s2transformer"][1][0]), list(o["remaining_transform_prediction"][0])]
        scopes = ["transformer1/embedding", "transformer2/embedding", "embedding"]
        heads = ["transformer1/mlp/", "transformer2/mlp/", ""]

        def get(kind, stage, layer, shape):
            a = np.empty(shape, np.uint8)
            self._check(self._lib.alignnet_debug_train_relu_mask(self._h, kind, stage, layer, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size))
            return a.astype(bool)
        out = {}
        for s in range(3):
            nl = len(convs[s])
            for l, c in enumerate(convs[s]):
                if l == nl - 1:
                    shape = (2, B, c)                 # at the winning point
                elif dg and l == nl - 2:
                    shape = (2, B * n, c)             # at the winning neighbour slot
                else:
                    shape = (2, B * n * (20 if dg else 1), c)
                m = get(0, s, l, shape)
                for t in range(2):
                    out[f"{t}:{scopes[s]}/conv{l + 1}"] = m[t]
            for j, c in enumerate(fcs[s]):
                if s < 2:
                    m = get(1, s, j, (2, B, c))
                    for t in range(2):
                        out[f"{t}:{heads[s]}fc{j + 1}"] = m[t]
                else:
                    out[f"p:fc{j + 1}"] = get(1, s, j, (B, c))
        return out

    def grad_buffer(self):
        ptr, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.alignnet_grad_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    # ---- multi-GPU (RCCL over xGMI) ---------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        if _capi.load_library().alignnet_comm_unique_id(buf) != 0:
            raise EngineError("alignnet_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    @staticmethod
    def comm_loopback_id():
        """Id of a new in-process loopback group: `world` engines of this process on one device, one host thread each
        (include/alignnet_hip.h: alignnet_comm_loopback_id) -- the multi-rank paths on a 1-GPU box."""
        buf = (C.c_uint8 * 128)()
        if _capi.load_library().alignnet_comm_loopback_id(buf) != 0:
            raise EngineError("alignnet_comm_loopback_id failed")
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.alignnet_comm_init(self._h, rank, world, buf))

    def comm_init_grad(self, rank, world, unique_id):
        """A second communicator of the same ranks for the gradient buckets alone (include/alignnet_hip.h: alignnet_comm_init_grad)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.alignnet_comm_init_grad(self._h, rank, world, buf))

    def comm_allreduce_grads(self):
        self._check(self._lib.alignnet_comm_allreduce_grads(self._h))

    def comm_average_shadows(self):
        """Average the BatchNorm EMA shadows over the ranks of the engine's communicator (one all-reduce on the device)."""
        self._check(self._lib.alignnet_comm_average_shadows(self._h))

    # ---- checkpoints (tf.train.Saver, train.py:220,252,268,281,317,321) -----------
    def save(self, path):
        self._check(self._lib.alignnet_save(self._h, str(path).encode()))

    def load(self, path, skip_step=False):
        self._check(self._lib.alignnet_load(self._h, str(path).encode(), int(skip_step)))

    # ---- state -----------------------------------------------------------------
    def state(self):
        st = _capi.State()
        self._check(self._lib.alignnet_get_state(self._h, C.byref(st)))
        return dict(step=st.step, learning_rate=st.learning_rate, bn_decay=st.bn_decay)

    def set_step(self, step):
        self._check(self._lib.alignnet_set_step(self._h, int(step)))

    # ---- profiling hook ----------------------------------------------------------
    def set_option(self, key, value):
        """Run-time options outside the reference's config surface (include/alignnet_hip.h: alignnet_set_option)."""
        self._check(self._lib.alignnet_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._check(self._lib.alignnet_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def last_backbone_kernel(self):
        """Name of the backbone kernel instantiation the most recent eval forward launched."""
        return KERNEL_NAMES.get(self.get_option("last_backbone_kernel"), "none")

    def profile_enable(self, on=True):
        self._check(self._lib.alignnet_profile_enable(self._h, int(on)))

    PROFILED_KERNELS = ("backbone", "knn", "train_fwd_phase2", "train_fwd_phase3", "train_gram_h2", "train_bwd_b2", "train_bwd_b1",
                        "dg_train_fwd", "dg_train_bwd_edge", "allreduce", "optimizer")

    def profile_kernels(self):
        """{kernel: (ms, launches)} accumulated since the last profile_read(reset=True); call BEFORE that reset."""
        out = {}
        for k in self.PROFILED_KERNELS:
            ms, n = C.c_double(), C.c_int64()
            self._check(self._lib.alignnet_profile_read_kernel(self._h, k.encode(), C.byref(ms), C.byref(n)))
            if n.value:
                out[k] = (ms.value, n.value)
        return out

    def profile_read(self, reset=True):
        ms, n, tot = C.c_double(), C.c_int64(), C.c_double()
        self._check(self._lib.alignnet_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(tot), int(reset)))
        return dict(backbone_ms=ms.value, backbone_launches=n.value, total_ms=tot.value)
