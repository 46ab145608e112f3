"""CPU oracle: NumPy restatement of the AlignNet-3D `tp8` hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  The product path (`alignnet-3d_amd/`) never routes through this file and
fails loudly when the HIP library is missing.

PARITY STATUS: **parity unpinned** for the TensorFlow part of the path.  The
reference arithmetic lives in TensorFlow 1.8 (reference `README.md:31`), which
is neither vendored in the reference nor installable here, and the reference
ships no tests / golden vectors.  What pins this
file instead:
  * a second, independently written restatement (`oracle/alignnet_torch.py`,
    torch-CPU autograd) that must agree with this one (tests/test_oracle.py);
  * the NumPy-only pieces of the reference that CAN be imported here
    (`models/tp8.py:229-244` classLogits2angle) are checked against golden
    vectors generated from the reference itself (tests/golden/).

Every function cites the reference file:line it follows (paths relative to the
reference repository root).

Conventions
-----------
* point clouds are [B, N, 3] row-major, xyz innermost (models/tp8.py:13-15)
* conv weights are stored 2-D: conv1 `[1,3,1,C]` HWIO -> `[3, C]`; 1x1 convs
  `[1,1,Cin,Cout]` -> `[Cin, Cout]`; FC `[Cin, Cout]` (utils/tf_util.py:148-152,
  333-337)
* shared `weights`/`biases` (tf.get_variable + AUTO_REUSE, models/tp8.py:140-143,
  utils/tf_util.py:21), but one BatchNorm parameter set PER TOWER because
  beta/gamma are `tf.Variable` (utils/tf_util.py:470-473): tower 0 lives under
  `siamese/`, tower 1 under `siamese_1/`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

BN_EPS = float(np.float32(1e-3))  # utils/tf_util.py:491; the Python float 1e-3 enters the graph as a float32 constant (1.0000000475e-3)


# --------------------------------------------------------------------------
# network description (mirrors cfg.model.* of config.py / configs/*.json)
# --------------------------------------------------------------------------
@dataclass
class NetSpec:
    """Hyper-parameters the graph builder reads from `cfg` (models/tp8.py:10,98,154)."""

    num_points: int = 1024
    num_channels: int = 3
    num_bins: int = 50
    s1_conv: Sequence[int] = (64, 128, 256)
    s1_fc: Sequence[int] = (512, 256)
    s1_keep: Optional[float] = 0.7
    s2_conv: Sequence[int] = (64, 128, 512)
    s2_fc: Sequence[int] = (512, 256)
    s2_keep: Optional[float] = 0.7
    emb_conv: Sequence[int] = (64, 128, 1024)
    rem_fc: Sequence[int] = (512, 256)
    rem_keep: Optional[float] = 0.7
    backbone: str = "pointnet"
    angle_factor: float = 1.0
    early_stage_factor: float = 0.5
    accept_inverted_angle: bool = True
    knn_k: int = 20  # models/tp8.py:33 (hard-coded)

    @property
    def out_s1(self) -> int:
        return 3  # models/tp8.py:98 with_angles=False

    @property
    def out_s2(self) -> int:
        return 3 + 2 * self.num_bins  # models/tp8.py:98 with_angles=True

    @staticmethod
    def from_cfg(cfg: dict) -> "NetSpec":
        """Build from a merged config dict (config.py:66-82 layout)."""
        m = cfg["model"]
        o = m["options"]
        return NetSpec(
            num_points=m["num_points"],
            num_channels=cfg["data"]["num_channels"],
            num_bins=m["angles"]["num_bins"],
            s1_conv=tuple(o["s1transformer"][0]),
            s1_fc=tuple(o["s1transformer"][1][0]),
            s1_keep=o["s1transformer"][1][1],
            s2_conv=tuple(o["s2transformer"][0]),
            s2_fc=tuple(o["s2transformer"][1][0]),
            s2_keep=o["s2transformer"][1][1],
            emb_conv=tuple(o["embedding"]),
            rem_fc=tuple(o["remaining_transform_prediction"][0]),
            rem_keep=o["remaining_transform_prediction"][1],
            backbone=m["backbone"],
            angle_factor=o["angle_factor"],
            early_stage_factor=o["early_stage_factor"],
            accept_inverted_angle=m["angles"]["accept_inverted_angle"],
        )


@dataclass
class LayerDesc:
    name: str  # variable-scope path below the tower prefix, e.g. "transformer1/embedding/conv1"
    cin: int
    cout: int
    bn: bool
    siamese: bool  # True: lives under siamese/ (one BN set per tower)
    fan_in: int
    fan_out: int


def layer_table(spec: NetSpec) -> List[LayerDesc]:
    """Every trainable layer in graph-construction order (models/tp8.py:101-158)."""
    out: List[LayerDesc] = []

    def conv_stack(prefix: str, widths: Sequence[int], siamese: bool = True):
        if spec.backbone == "pointnet":
            cin = spec.num_channels
            for i, c in enumerate(widths):
                if i == 0:
                    # kernel [1, num_channel] on a 1-channel image: models/tp8.py:55
                    fi, fo = spec.num_channels * 1, spec.num_channels * c
                else:
                    fi, fo = cin, c
                out.append(LayerDesc(f"{prefix}/conv{i+1}", cin, c, True, siamese, fi, fo))
                cin = c
        elif spec.backbone == "dgcnn":
            # edge feature has 2*num_dims channels: utils/tf_util_dgcnn.py:705
            cin = 2 * spec.num_channels
            for i, c in enumerate(widths):
                out.append(LayerDesc(f"{prefix}/conv{i+1}", cin, c, True, siamese, cin, c))
                cin = c
        else:
            raise AssertionError("backbone")  # models/tp8.py:68

    def fc_stack(prefix: str, cin: int, widths: Sequence[int], siamese: bool):
        for j, c in enumerate(widths):
            last = j == len(widths) - 1
            nm = f"{prefix}/fc{j+1}" if prefix else f"fc{j+1}"
            out.append(LayerDesc(nm, cin, c, not last, siamese, cin, c))
            cin = c

    conv_stack("transformer1/embedding", spec.s1_conv)
    fc_stack("transformer1/mlp", spec.s1_conv[-1], list(spec.s1_fc) + [spec.out_s1], True)
    conv_stack("transformer2/embedding", spec.s2_conv)
    fc_stack("transformer2/mlp", spec.s2_conv[-1], list(spec.s2_fc) + [spec.out_s2], True)
    # scope_name 'final_embedding' is ignored -> 'embedding' (models/tp8.py:62-66,130)
    conv_stack("embedding", spec.emb_conv)
    fc_stack("", 2 * spec.emb_conv[-1], list(spec.rem_fc) + [spec.out_s2], False)
    return out


TOWER_PREFIX = ("siamese", "siamese_1")  # models/tp8.py:140-143 name scopes


def param_names(spec: NetSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat list of (name, shape) for every persistent tensor (trainable + EMA)."""
    names: List[Tuple[str, Tuple[int, ...]]] = []
    for L in layer_table(spec):
        base = f"siamese/{L.name}" if L.siamese else L.name
        names.append((f"{base}/weights", (L.cin, L.cout)))
        names.append((f"{base}/biases", (L.cout,)))
        if L.bn:
            towers = TOWER_PREFIX if L.siamese else ("",)
            for t in towers:
                b = f"{t}/{L.name}" if t else L.name
                for leaf in ("beta", "gamma", "moving_mean", "moving_var"):
                    names.append((f"{b}/bn/{leaf}", (L.cout,)))
    return names


def trainable_names(spec: NetSpec) -> List[str]:
    return [n for n, _ in param_names(spec) if not n.endswith(("moving_mean", "moving_var"))]


def init_params(spec: NetSpec, seed: int = 0, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Xavier-uniform weights, zero biases, beta 0, gamma 1, EMA shadows 0
    (utils/tf_util.py:10-49,470-480; SURVEY 8.A3: TF EMA slots start at zero)."""
    rng = np.random.default_rng(seed)
    P: Dict[str, np.ndarray] = {}
    for L in layer_table(spec):
        base = f"siamese/{L.name}" if L.siamese else L.name
        limit = math.sqrt(6.0 / (L.fan_in + L.fan_out))
        P[f"{base}/weights"] = rng.uniform(-limit, limit, size=(L.cin, L.cout)).astype(dtype)
        P[f"{base}/biases"] = np.zeros((L.cout,), dtype)
        if L.bn:
            for t in (TOWER_PREFIX if L.siamese else ("",)):
                b = f"{t}/{L.name}" if t else L.name
                P[f"{b}/bn/beta"] = np.zeros((L.cout,), dtype)
                P[f"{b}/bn/gamma"] = np.ones((L.cout,), dtype)
                P[f"{b}/bn/moving_mean"] = np.zeros((L.cout,), dtype)
                P[f"{b}/bn/moving_var"] = np.zeros((L.cout,), dtype)
    return P


def randomize_bn(P: Dict[str, np.ndarray], seed: int = 1) -> None:
    """Give BN tensors non-trivial values so parity tests exercise them
    (a trained checkpoint has arbitrary beta/gamma/shadows; each tower differs)."""
    rng = np.random.default_rng(seed)
    for k in sorted(P):
        dt = P[k].dtype
        if k.endswith("/beta"):
            P[k] = rng.normal(0, 0.1, P[k].shape).astype(dt)
        elif k.endswith("/gamma"):
            P[k] = rng.uniform(0.5, 1.5, P[k].shape).astype(dt)
        elif k.endswith("/moving_mean"):
            P[k] = rng.normal(0, 0.2, P[k].shape).astype(dt)
        elif k.endswith("/moving_var"):
            P[k] = rng.uniform(0.5, 2.0, P[k].shape).astype(dt)
        elif k.endswith("/biases"):
            P[k] = rng.normal(0, 0.05, P[k].shape).astype(dt)


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def batch_norm(z, P, bn_base, is_training, bn_decay, updates):
    """utils/tf_util.py:455-492 (+ tf.nn.moments two-pass biased variance and
    tf.nn.batch_normalization `inv = rsqrt(var+eps)*gamma`)."""
    beta, gamma = P[f"{bn_base}/beta"], P[f"{bn_base}/gamma"]
    if is_training:
        mean = z.mean(axis=0)
        var = ((z - mean) ** 2).mean(axis=0)
        if updates is not None:
            d = 0.9 if bn_decay is None else bn_decay  # tf_util.py:475
            for leaf, val in (("moving_mean", mean), ("moving_var", var)):
                s = P[f"{bn_base}/{leaf}"]
                # ExponentialMovingAverage.apply: s -= (1-d)*(s-v)
                updates[f"{bn_base}/{leaf}"] = (d * s + (1 - d) * val).astype(s.dtype)
    else:
        mean, var = P[f"{bn_base}/moving_mean"], P[f"{bn_base}/moving_var"]
    inv = gamma / np.sqrt(var + z.dtype.type(BN_EPS))
    return z * inv + (beta - mean * inv)


def dense(x, P, base, bn_base, relu, is_training, bn_decay, updates):
    """`act(BN(x W + b))`: utils/tf_util.py:156-168 (conv2d), :337-346 (fully_connected)."""
    z = x @ P[f"{base}/weights"] + P[f"{base}/biases"]
    if bn_base is not None:
        z = batch_norm(z, P, bn_base, is_training, bn_decay, updates)
    if relu:
        z = np.maximum(z, 0)
    return z


def pointnet_backbone(x, P, spec, scope, widths, tower, is_training, bn_decay, updates):
    """models/tp8.py:49-59: shared per-point MLP then max over points.
    x: [B, N, 3] -> [B, C_last]."""
    B, N, _ = x.shape
    h = x.reshape(B * N, -1)
    for i in range(len(widths)):
        nm = f"{scope}/conv{i+1}"
        h = dense(h, P, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", True, is_training, bn_decay, updates)
    return h.reshape(B, N, -1).max(axis=1)  # tf_util.py:350-373, window = all points


def knn_indices(x, k):
    """utils/tf_util_dgcnn.py:638-671.  top_k on the negated distance, self included.
    Ties: tf.nn.top_k returns the lower index first; a stable argsort does the same."""
    inner = -2.0 * (x @ x.transpose(0, 2, 1))
    sq = (x * x).sum(-1, keepdims=True)
    adj = sq + inner + sq.transpose(0, 2, 1)
    return np.argsort(adj, axis=-1, kind="stable")[..., :k]


def dgcnn_backbone(x, P, spec, scope, widths, tower, is_training, bn_decay, updates):
    """models/tp8.py:30-46: static kNN graph on xyz, edge feature [x_i, x_j - x_i],
    1x1 convs widths[:-1] on [B,N,k,.], max over k, conv widths[-1], max over N."""
    B, N, D = x.shape
    k = spec.knn_k
    idx = knn_indices(x, k)  # [B,N,k]
    nbr = x[np.arange(B)[:, None, None], idx]  # [B,N,k,D]  tf_util_dgcnn.py:699-700
    cen = np.broadcast_to(x[:, :, None, :], nbr.shape)
    h = np.concatenate([cen, nbr - cen], axis=-1).reshape(B * N * k, 2 * D)  # tf_util_dgcnn.py:705
    for i in range(len(widths) - 1):
        nm = f"{scope}/conv{i+1}"
        h = dense(h, P, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", True, is_training, bn_decay, updates)
    h = h.reshape(B * N, k, -1).max(axis=1)  # tp8.py:42
    nm = f"{scope}/conv{len(widths)}"
    h = dense(h, P, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", True, is_training, bn_decay, updates)
    return h.reshape(B, N, -1).max(axis=1)


def backbone(x, P, spec, scope, widths, tower, is_training, bn_decay, updates):
    fn = pointnet_backbone if spec.backbone == "pointnet" else dgcnn_backbone
    return fn(x, P, spec, scope, widths, tower, is_training, bn_decay, updates)


def head_mlp(x, P, scope, widths, tower, keep, is_training, bn_decay, updates, dropout_u):
    """models/tp8.py:75-82.  `tower=None` means the top-level pair head (single BN set).
    dropout_u: uniform[0,1) array of the hidden shape, or None -> no dropout mask
    applied (only legal when not training).  tf.nn.dropout: x/keep*floor(keep+u)."""
    h = x
    for j in range(len(widths) - 1):
        nm = f"{scope}/fc{j+1}" if scope else f"fc{j+1}"
        if tower is None:
            base, bnb = nm, nm + "/bn"
        else:
            base, bnb = f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn"
        h = dense(h, P, base, bnb, True, is_training, bn_decay, updates)
    if keep is not None and is_training:
        assert dropout_u is not None, "training-mode dropout needs explicit uniforms"
        kp = h.dtype.type(np.float32(keep))   # keep_prob enters the graph as a float32 constant (0.7 -> 0.69999999)
        mask = np.floor(kp + dropout_u.astype(h.dtype))
        h = h / kp * mask
    nm = f"{scope}/fc{len(widths)}" if scope else f"fc{len(widths)}"
    base = nm if tower is None else f"siamese/{nm}"
    return dense(h, P, base, None, False, is_training, bn_decay, updates)


def floor_mod(x, y):
    """tf.mod is floor-mod."""
    return x - np.floor(x / y) * y


def get_angles(logits, nb):
    """models/tp8.py:294-301 + :202-212.  In-graph yaw decode (residual scaled by pi/nb)."""
    dt = logits.dtype.type
    cls = np.argmax(logits[:, :nb], axis=1)
    res = logits[:, nb:] * (dt(np.float32(np.pi)) / dt(nb))
    per = res[np.arange(logits.shape[0]), cls]
    pi = dt(np.float32(np.pi))
    apc = dt(2.0) * pi / dt(nb)
    ang = cls.astype(logits.dtype) * apc + per
    return floor_mod(ang + pi, dt(2.0) * pi) - pi, cls


def rot_z(a):
    """models/tp8.py:26-27, batched: [B] -> [B,3,3]."""
    c, s = np.cos(a), np.sin(a)
    z, o = np.zeros_like(a), np.ones_like(a)
    return np.stack([c, -s, z, s, c, z, z, z, o], axis=-1).reshape(-1, 3, 3)


def embedding_net(pcs, P, spec, tower, is_training, bn_decay, updates, dropout_u):
    """models/tp8.py:101-132 (one tower)."""
    center_mean = pcs.mean(axis=1)  # :104
    x1 = pcs - center_mean[:, None, :]  # :106
    f1 = backbone(x1, P, spec, "transformer1/embedding", spec.s1_conv, tower, is_training, bn_decay, updates)
    o1 = head_mlp(f1, P, "transformer1/mlp", list(spec.s1_fc) + [spec.out_s1], tower, spec.s1_keep,
                  is_training, bn_decay, updates, None if dropout_u is None else dropout_u[f"s1_{tower}"])
    s1c = o1 + center_mean  # :109
    x2 = pcs - s1c[:, None, :]  # :113
    f2 = backbone(x2, P, spec, "transformer2/embedding", spec.s2_conv, tower, is_training, bn_decay, updates)
    o2 = head_mlp(f2, P, "transformer2/mlp", list(spec.s2_fc) + [spec.out_s2], tower, spec.s2_keep,
                  is_training, bn_decay, updates, None if dropout_u is None else dropout_u[f"s2_{tower}"])
    s2c = o2[:, :3] + s1c  # :117
    logits = o2[:, 3:]  # :118
    ang, cls = get_angles(logits, spec.num_bins)  # :123
    R = rot_z(-ang)  # :125
    x3 = np.einsum("bnc,bcd->bnd", pcs - s2c[:, None, :], R)  # :122,127
    emb = backbone(x3, P, spec, "embedding", spec.emb_conv, tower, is_training, bn_decay, updates)  # :130
    aux = dict(f1=f1, f2=f2, cls=cls, ang=ang, x3=x3)
    return emb, center_mean, s1c, s2c, logits, aux


def get_model(P, spec: NetSpec, pcs1, pcs2, is_training=False, bn_decay=None,
              dropout_u: Optional[Dict[str, np.ndarray]] = None, collect_updates: bool = True):
    """models/tp8.py:135-158.  Returns (end_points, ema_updates, aux)."""
    updates: Optional[Dict[str, np.ndarray]] = {} if (is_training and collect_updates) else None
    e1, cm1, s1c1, s2c1, lg1, aux1 = embedding_net(pcs1, P, spec, 0, is_training, bn_decay, updates, dropout_u)
    e2, cm2, s1c2, s2c2, lg2, aux2 = embedding_net(pcs2, P, spec, 1, is_training, bn_decay, updates, dropout_u)
    comb = np.concatenate([e1, e2], axis=1)  # :144,153
    net = head_mlp(comb, P, "", list(spec.rem_fc) + [spec.out_s2], None, spec.rem_keep, is_training, bn_decay,
                   updates, None if dropout_u is None else dropout_u["rem"])
    ep = {
        "pred_s1_pc1centers": s1c1, "pred_s1_pc2centers": s1c2,
        "pred_s2_pc1centers": s2c1, "pred_s2_pc2centers": s2c2,
        "pred_pc1angle_logits": lg1, "pred_pc2angle_logits": lg2,
        "pred_translations": net[:, :3] + (s2c2 - s2c1),  # :155
        "pred_remaining_angle_logits": net[:, 3:],  # :156
    }
    aux = dict(emb1=e1, emb2=e2, t1=aux1, t2=aux2, center_mean1=cm1, center_mean2=cm2)
    return ep, updates, aux


# --------------------------------------------------------------------------
# host-side decode used for pred_angles.npy (NumPy in the reference as well)
# --------------------------------------------------------------------------
def class_logits_to_angle(logits: np.ndarray, nb: int) -> np.ndarray:
    """models/tp8.py:229-244.  NOTE the residual is NOT de-normalised here
    (quirk A6(i)); arithmetic is float64 because `angle_per_class` is a Python float."""
    cls = np.argmax(logits[:, :nb], axis=1)
    res = logits[:, nb:]
    apc = 2 * np.pi / float(nb)
    out = []
    for c, r in zip(cls, res):
        a = c * apc + r[c]
        if a > np.pi:
            a = a - 2 * np.pi
        out.append(a)
    return np.array(out)


def pred_angles(ep: Dict[str, np.ndarray], nb: int) -> np.ndarray:
    """train.py:453-456."""
    a1 = class_logits_to_angle(ep["pred_pc1angle_logits"], nb)
    a2 = class_logits_to_angle(ep["pred_pc2angle_logits"], nb)
    ar = class_logits_to_angle(ep["pred_remaining_angle_logits"], nb)
    return a2 - a1 + ar


# --------------------------------------------------------------------------
# loss `separate` (models/tp8.py:304-354) -- shapes follow the reference
# EXACTLY, including its [B] vs [B,1] broadcasts (see DESIGN.md quirks ix, x).
# --------------------------------------------------------------------------
def huber(err, delta):
    """models/tp8.py:173-178: mean over ALL elements."""
    a = np.abs(err)
    q = np.minimum(a, delta)
    return (0.5 * q ** 2 + delta * (a - q)).mean()


def angle2class(angle, nb):
    """models/tp8.py:181-199.  angle [B,1] (or [B,B]) -> (class_id[:,0] [B], residual same shape)."""
    dt = angle.dtype.type
    twopi = dt(np.float32(2.0 * np.pi))
    a = floor_mod(angle, twopi)
    apc = twopi / dt(nb)
    sh = floor_mod(a + apc / dt(2.0), twopi)
    cls = (sh / apc).astype(np.int32)  # tf.to_int32 truncates; sh/apc >= 0
    res = sh - (cls.astype(angle.dtype) * apc + apc / dt(2.0))
    return cls[:, 0], res


def softmax_ce(logits, labels):
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    return lse - logits[np.arange(logits.shape[0]), labels]


def angle_loss(logits, target_angles, nb):
    """models/tp8.py:266-281.  target_angles is [B,1] (stage 2) or [B,B] (stage 3):
    `reduce_sum(...)` is [B] and the label is [B,1]/[B,B], so the Huber error
    broadcasts to [B,B] -- reference behaviour, reproduced as is."""
    dt = logits.dtype.type
    cls, res = angle2class(target_angles, nb)
    # an out-of-range class id would make TF's sparse CE return NaN; clamp never triggers
    # for finite angles except sh/apc == nb by rounding, kept as-is (index error if so).
    ce = softmax_ce(logits[:, :nb], cls).mean()
    onehot_pick = logits[:, nb:][np.arange(logits.shape[0]), cls]  # [B]
    label = res / dt(np.float32(np.pi / nb))  # [B,1] or [B,B]; the Python float np.pi / nb enters the graph as a float32 constant
    rl = huber(onehot_pick - label, dt(1.0))  # [B] - [B,1] -> [B,B]
    return np.array([ce + dt(20.0) * rl, ce, rl])


def angle_losses(logits, target_angles, nb, accept_inverted):
    """models/tp8.py:284-291: tf.cond picks the LARGER of (theta, theta+pi) (quirk A6(ii))."""
    a = angle_loss(logits, target_angles, nb)
    if accept_inverted:
        b = angle_loss(logits, target_angles + target_angles.dtype.type(np.float32(np.pi)), nb)   # `+ np.pi`: a float32 constant in the graph
        return a if a[0] > b[0] else b
    return a


def get_loss(spec: NetSpec, ep, translations, rel_angles, pc1_centers, pc2_centers, pc1_angles, pc2_angles):
    """models/tp8.py:304-354.  Returns (per_transform_loss, dict of the 16 summary scalars)."""
    nb = spec.num_bins
    dt = ep["pred_translations"].dtype.type
    B = translations.shape[0]
    s1a = huber(ep["pred_s1_pc1centers"] - pc1_centers, dt(1.0))
    s1b = huber(ep["pred_s1_pc2centers"] - pc2_centers, dt(1.0))
    s1 = (s1a + s1b) / dt(2.0)
    s2a = huber(ep["pred_s2_pc1centers"] - pc1_centers, dt(1.0))
    s2b = huber(ep["pred_s2_pc2centers"] - pc2_centers, dt(1.0))
    a1 = angle_losses(ep["pred_pc1angle_logits"], pc1_angles, nb, spec.accept_inverted_angle)
    a2 = angle_losses(ep["pred_pc2angle_logits"], pc2_angles, nb, spec.accept_inverted_angle)
    s2t = (s2a + s2b) / dt(2.0)
    s2ang = (a1[0] + a2[0]) / dt(2.0)
    s3t = huber(ep["pred_translations"] - translations, dt(2.0))
    p1, _ = get_angles(ep["pred_pc1angle_logits"], nb)  # [B]
    p2, _ = get_angles(ep["pred_pc2angle_logits"], nb)
    # [B,1] - [B] -> [B,B]  (models/tp8.py:327) -- reference broadcast, kept
    rem_target = (pc2_angles - pc1_angles) - (p2 - p1)
    a3 = angle_losses(ep["pred_remaining_angle_logits"], rem_target, nb, spec.accept_inverted_angle)
    esf, af = dt(spec.early_stage_factor), dt(spec.angle_factor)
    lt = esf * (s1 + s2t) + s3t
    la = esf * s2ang + a3[0]
    loss = lt + af * la
    summ = {
        "losses/translation": lt, "losses/angle": la,
        "losses_stages/stage1_pc1_transl_loss": s1a, "losses_stages/stage1_pc2_transl_loss": s1b,
        "losses_stages/stage2_pc1_transl_loss": s2a, "losses_stages/stage2_pc2_transl_loss": s2b,
        "losses_stages/stage3_transl_loss": s3t,
        "losses_stages/stage2_pc1_angle_loss": a1[0], "losses_stages/stage2_pc1_angle_class_loss": a1[1],
        "losses_stages/stage2_pc1_angle_residual_loss": a1[2],
        "losses_stages/stage2_pc2_angle_loss": a2[0], "losses_stages/stage2_pc2_angle_class_loss": a2[1],
        "losses_stages/stage2_pc2_angle_residual_loss": a2[2],
        "losses_stages/stage3_angle_loss": a3[0], "losses_stages/stage3_angle_class_loss": a3[1],
        "losses_stages/stage3_angle_residual_loss": a3[2],
    }
    return loss / dt(B), summ


# --------------------------------------------------------------------------
# schedules + optimiser (train.py:133-174, 211-217)
# --------------------------------------------------------------------------
def exponential_decay_staircase(base, global_step, decay_steps, rate):
    return base * rate ** math.floor(global_step / decay_steps)


def learning_rate(step, batch_size, ntrain, lr0, decay_step_epochs, rate, per="epoch"):
    """train.py:133-156."""
    nb_per_epoch = ntrain // batch_size
    ds = decay_step_epochs * (batch_size * nb_per_epoch if per == "epoch" else 1)
    return max(exponential_decay_staircase(lr0, step * batch_size, ds, rate), 1e-5)


def bn_decay_schedule(step, batch_size, ntrain, init, decay_step_epochs, rate, clip, per="epoch"):
    """train.py:159-174."""
    nb_per_epoch = ntrain // batch_size
    ds = decay_step_epochs * (batch_size * nb_per_epoch if per == "epoch" else 1)
    return min(clip, 1 - exponential_decay_staircase(init, step * batch_size, ds, rate))


def adam_step(w, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: eps OUTSIDE the bias correction. t = 1,2,...
    TF evaluates the update in the variable's dtype, float32: beta, (1 - beta), epsilon and the running beta powers (one float32
    multiplication per step) are float32 numbers -- 1 - float32(0.999) is 0.00099998713, not 0.001 -- and this restatement uses
    exactly those constants (in whatever precision w, g, m, v come) so that step 1 is lr * g / (|g| + eps') as in TF."""
    f = np.float32
    b1f, b2f = f(b1), f(b2)
    b1p, b2p = f(1), f(1)
    for _ in range(int(t)):
        b1p, b2p = f(b1p * b1f), f(b2p * b2f)
    lr_t = float(f(lr)) * math.sqrt(float(f(1) - b2p)) / float(f(1) - b1p)
    m = float(b1f) * m + float(f(1) - b1f) * g
    v = float(b2f) * v + float(f(1) - b2f) * g * g
    return w - lr_t * m / (np.sqrt(v) + float(f(eps))), m, v


# --------------------------------------------------------------------------
# synthetic pairs
# --------------------------------------------------------------------------
def synth_pairs(B, N, seed=1234, dtype=np.float32):
    """Synthetic pairs for the tests (the recipe lives in alignnet3d/synth.py; same function, same stream)."""
    import os, sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alignnet-3d_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from alignnet3d.synth import synth_pairs as _sp
    return _sp(B, N, seed, dtype)


def cast_params(P, dtype):
    return {k: v.astype(dtype) for k, v in P.items()}
