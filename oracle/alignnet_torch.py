"""CPU oracle #2: torch-CPU autograd restatement of the `tp8` hot path.

TEST INFRASTRUCTURE ONLY (same rules as `oracle/alignnet_ref.py`): never
imported by the product path.  **parity unpinned** against TensorFlow (absent,
SURVEY.md 8c); this file exists so that two restatements written separately --
this one from the reference sources with torch's own primitives
(`F.batch_norm`, `F.huber_loss`, `F.cross_entropy`, `torch.remainder`), the
other in NumPy from SURVEY 8.A -- must agree, and so that the analytic backward
of the HIP path has an autograd reference.

Reference lines followed: models/tp8.py:26-27,49-59,75-158 (forward),
:173-354 (loss), utils/tf_util.py:112-169,311-373,455-575 (layers),
train.py:211-217 (Adam).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .alignnet_ref import NetSpec, TOWER_PREFIX, BN_EPS


def to_torch(P: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=False) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not k.endswith(("moving_mean", "moving_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


class TorchTp8:
    """Eager re-statement; `P` is a name->tensor dict using the oracle's names."""

    STAGES = {"transformer1/embedding": 0, "transformer2/embedding": 1, "embedding": 2}

    def __init__(self, spec: NetSpec, P: Dict[str, torch.Tensor], bf16_lift: bool = False, checkpoint: bool = False, sync=None, pinned=None):
        self.spec, self.P = spec, P
        # pinned: the DECISIONS of another evaluation of the same step (the HIP engine's, alignnet_debug_train_decisions) -- dict with
        # "yaw" [2, B] decoded classes (models/tp8.py:296), "pool" three arrays [2, B, C_last] of arg-max points (utils/tf_util.py:350-373,
        # models/tp8.py:58), and for dgcnn "slot" three arrays [2, B, N, C_edge] of arg-max neighbour slots (models/tp8.py:42) and "knn"
        # [2, B, N, k] the neighbour table (utils/tf_util_dgcnn.py:638-676).  The oracle then GATHERS at those indices instead of
        # max / argmax / top_k, which makes the rest of the graph continuous in its inputs: a comparison of two evaluations no longer
        # sits on the noise floor of re-decided near-ties.  Each pinned decision is first checked against the oracle's own values:
        # pin_report collects (what, worst gap between the true extreme and the value at the pinned index, scale of the values,
        # number of entries whose index differs from the oracle's own first maximum) -- the caller asserts that the gaps are rounding.
        # "relu" (optional): {key: bool array} -- the SIGNS the other evaluation used at every relu (utils/tf_util.py:167-168,345-346): key =
        # "<tower>:<layer scope>" ("p:<scope>" for the pair head), hidden conv layers [rows, C], the conv in front of a max-pool [R, C] AT THE
        # WINNER (max and relu commute), head layers [B, C].  The oracle then evaluates y = bn(z) * mask instead of relu(bn(z)): with winners,
        # classes and signs pinned the whole step is a smooth function of its inputs.  Each mask is checked first: where it disagrees with
        # the oracle's own sign, |bn(z)| must be rounding-sized (pin_report entry "relu:<key>": worst such |bn(z)|, scale, disagreements).
        self.pinned = pinned
        self.pin_report = []
        # record_decisions = True: the unpinned evaluation writes its own choices into self.decisions in the layout of `pinned`
        # (tests/test_oracle.py: pinning the oracle to its own decisions must change nothing)
        self.record_decisions = False
        self.decisions = {"yaw": [None, None], "pool": [[None, None] for _ in range(3)], "slot": [[None, None] for _ in range(3)], "knn": [None, None], "relu": {}}
        # sync: data-parallel protocol of the engine's "sync_bn" / "global_loss" options restated (tests/test_parallel_cpu.py): an object with
        # .world, .rank, .allreduce(t) (differentiable sum over the ranks) and .gather(t) (list of every rank's tensor, no gradient).
        # BatchNorm moments are then those of the global batch; loss_global() evaluates the loss on the gathered batch, with the
        # gradient flowing into this rank's rows only -- summed over the ranks, the parameter gradients are the single-device ones.
        self.sync = sync
        # checkpoint: recompute each backbone in the backward instead of keeping its [B*N, C] activations (full-size batches:
        # 256 x 1024 points x 1024 channels in fp64 is 2 GB per tensor).  Same arithmetic, same gradients.
        self.checkpoint = checkpoint
        self.ema_updates: Dict[str, torch.Tensor] = {}
        # Model of the engine's "train_matmul_bf16" option (not a reference feature): the operands of the MFMA convs of
        # every PointNet backbone (all but the K = 3 lift) -- with the dgcnn backbone: of the edge convs behind the K = 6 lift --
        # are rounded to bf16 (round-to-nearest-even), products are accumulated exactly, and the backward treats the rounding
        # as identity (straight-through).
        self.bf16_lift = bf16_lift

    def _round_bf16_st(self, x, key=None):
        """x rounded to bf16, straight-through.  pinned["round"][key] (optional): ANOTHER evaluation's rounded values of this tensor -- every rounding to bf16 is a
        decision of its own, 2^-8 of the value; the other evaluation's value is taken after checking that it is one of the two bf16 neighbours of x
        (pin_report "round:<key>": worst |pinned - x| in bf16 ulps of x, entries that differ from this evaluation's own rounding)."""
        own = x.detach().to(torch.bfloat16).to(x.dtype)
        r = None
        if key is not None and self.pinned is not None and "round" in self.pinned:
            r = self.pinned["round"].get(key)
        if r is None:
            return x + (own - x.detach())
        r = torch.as_tensor(np.asarray(r)).to(x.dtype).reshape(x.shape)
        with torch.no_grad():
            xd = x.detach()
            # bf16 spacing is 2^-7 .. 2^-8 of the value; near zero (a relu output that is 1e-9 here and 1e-6 there) the two evaluations differ by
            # their own rounding of the pre-activation, not by a bf16 step: floor of 1e-5 of the tensor's scale
            ulp = torch.maximum(torch.abs(xd), torch.abs(r)) * 2.0 ** -7 + 1e-5 * float(xd.abs().max())
            d = torch.abs(r - xd) / ulp
            differ = r != own
            nd = int(differ.sum())
            self.pin_report.append((f"round:{key}", float(d[differ].max()) if nd else 0.0, 1.0, nd, r.numel()))
            # ... and how many sit further than 1.5 steps away (the other evaluation's OWN passes recompute h1 from xyz in separately compiled copies of
            # one expression; where two copies differ by an fp32 rounding that crosses a bf16 boundary, the h2 row behind it moves by a bf16 step of one input)
            far = int((d > 1.5).sum())
            self.pin_report.append((f"roundtail:{key}", far / r.numel(), 1.0, far, r.numel()))
        return x + (r - x.detach())

    # -- layers ---------------------------------------------------------
    def _bn(self, z, base, training, decay):
        P = self.P
        g, b = P[base + "/gamma"], P[base + "/beta"]
        if training and self.sync is not None:
            n = z.shape[0] * self.sync.world
            m = self.sync.allreduce(z.sum(0)) / n
            v = self.sync.allreduce(((z - m) ** 2).sum(0)) / n
            with torch.no_grad():
                d = 0.9 if decay is None else decay
                self.ema_updates[base + "/moving_mean"] = d * P[base + "/moving_mean"] + (1 - d) * m
                self.ema_updates[base + "/moving_var"] = d * P[base + "/moving_var"] + (1 - d) * v
            return (z - m) * torch.rsqrt(v + BN_EPS) * g + b
        if training:
            # F.batch_norm(training=True) normalises with the biased batch variance,
            # which is tf.nn.moments' variance (utils/tf_util.py:474).
            y = F.batch_norm(z, None, None, g, b, True, 0.0, BN_EPS)
            with torch.no_grad():
                d = 0.9 if decay is None else decay
                m = z.mean(0)
                v = z.var(0, unbiased=False)
                self.ema_updates[base + "/moving_mean"] = d * P[base + "/moving_mean"] + (1 - d) * m
                self.ema_updates[base + "/moving_var"] = d * P[base + "/moving_var"] + (1 - d) * v
            return y
        return F.batch_norm(z, P[base + "/moving_mean"], P[base + "/moving_var"], g, b, False, 0.0, BN_EPS)

    def _relu_mask(self, key):
        if self.pinned is None or "relu" not in self.pinned or key is None:
            return None
        m = self.pinned["relu"].get(key)
        return None if m is None else torch.as_tensor(np.asarray(m)).to(torch.bool)

    def _relu(self, z, key):
        """relu(z) -- or, with the signs pinned, z * mask (after checking the mask against z)."""
        m = self._relu_mask(key)
        if m is None:
            if self.record_decisions and key is not None:
                with torch.no_grad():
                    self.decisions["relu"][key] = (z > 0).numpy()
            return torch.relu(z)
        m = m.reshape(z.shape)
        with torch.no_grad():
            differ = (z > 0) != m
            nd = int(differ.sum())
            self.pin_report.append((f"relu:{key}", float(z[differ].abs().max()) if nd else 0.0, float(z.abs().max()), nd, m.numel()))
        return z * m.to(z.dtype)

    def _layer(self, x, wbase, bnbase, training, decay, act=True, round_operands=False, relu_key=None):
        w = self.P[wbase + "/weights"]
        if round_operands:
            x, w = self._round_bf16_st(x, relu_key), self._round_bf16_st(w)   # (relu_key = "<tower>:<layer scope>": the layer that consumes x)
        z = F.linear(x, w.t(), self.P[wbase + "/biases"])
        if bnbase is not None:
            z = self._bn(z, bnbase, training, decay)
        return self._relu(z, relu_key) if act else z

    @staticmethod
    def bf16_conv_layers(widths):
        """Which conv layers of a PointNet backbone the engine runs on bf16 MFMA under "train_matmul_bf16" (alignnet_train.hip:
        stage_generic / stage_hybrid): every layer behind the K = 3 lift of a stage in the fused kernels' shape (three layers, widths in
        multiples of 32, C1, C2 <= 128, C3 <= 1024); of any other stage only the last layer, when it runs as the fused tail
        (last two widths in multiples of 32, <= 128 / <= 1024, first width <= 128) and its input width is 32, 64 or 128; else none."""
        n = len(widths)
        if n == 3 and all(c % 32 == 0 for c in widths) and widths[0] <= 128 and widths[1] <= 128 and widths[2] <= 1024:
            return {1, 2}
        if n >= 3 and widths[0] <= 128 and widths[0] % 8 == 0 and widths[-2] in (32, 64, 128) and widths[-1] % 32 == 0 and widths[-1] <= 1024:
            return {n - 1}
        return set()

    def _pointnet(self, x, scope, widths, tower, training, decay):
        B, N, _ = x.shape
        h = x.reshape(B * N, -1)
        rounded = self.bf16_conv_layers(list(widths)) if (self.bf16_lift and training) else set()
        for i in range(len(widths)):
            nm = f"{scope}/conv{i+1}"
            # the last conv's relu is applied AFTER the max-pool (max_n relu(v_n) = relu(max_n v_n)): one sign per (cloud, channel)
            h = self._layer(h, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", training, decay, round_operands=i in rounded,
                            act=i < len(widths) - 1, relu_key=f"{tower}:{nm}")
        return self._relu(self._max_over(h.reshape(B, N, -1), "pool", scope, tower), f"{tower}:{scope}/conv{len(widths)}")

    def _pin(self, kind, scope, tower):
        if self.pinned is None or kind not in self.pinned:
            return None
        a = self.pinned[kind]
        a = a[self.STAGES[scope]] if scope is not None else a
        return torch.as_tensor(np.asarray(a[tower]), dtype=torch.long)

    def _max_over(self, hh, kind, scope, tower):
        """max over dim 1 of hh [R, n, C] (utils/tf_util.py:350-373) -- or, pinned, the gather at the given winners [R, C].  The callers apply
        the layer's relu BEHIND this max (max_n relu(v_n) = relu(max_n v_n)), so a pinned winner is checked on the relu'd values: in a channel
        whose values are all negative every point is a maximum of relu(v) = 0."""
        idx = self._pin(kind, scope, tower)
        if idx is None:
            if self.record_decisions:
                with torch.no_grad():
                    self.decisions[kind][self.STAGES[scope]][tower] = torch.relu(hh).argmax(dim=1).numpy()
            return hh.amax(dim=1)
        idx = idx.reshape(hh.shape[0], hh.shape[2])
        out = hh.gather(1, idx[:, None, :])[:, 0]
        with torch.no_grad():
            top, own = torch.relu(hh).max(dim=1)
            self.pin_report.append((f"{kind}:{scope}:{tower}", float((top - torch.relu(out)).max()), float(top.abs().max()), int((own != idx).sum()), idx.numel()))
        return out

    def _dgcnn(self, x, scope, widths, tower, training, decay):
        B, N, D = x.shape
        k = self.spec.knn_k
        with torch.no_grad():
            xx = (x * x).sum(-1, keepdim=True)
            adj = xx - 2 * x @ x.transpose(1, 2) + xx.transpose(1, 2)
            idx = torch.sort(adj, dim=-1, stable=True).indices[..., :k]
            pin = self._pin("knn", None, tower)
            if pin is not None:
                # the pinned table must be a k-nearest SET of every query up to rounding of the distances: its farthest member may not be
                # farther than the true k-th distance by more than the reported gap (scale: the largest k-th distance)
                kth = torch.sort(adj, dim=-1).values[..., k - 1]
                far = adj.gather(2, pin).amax(-1)
                self.pin_report.append((f"knn:{scope}:{tower}", float((far - kth).max()), float(kth.abs().max()),
                                        int((torch.sort(pin, -1).values != torch.sort(idx, -1).values).any(-1).sum()), pin.shape[0] * pin.shape[1]))
                assert int((torch.sort(pin, -1).values[..., 1:] == torch.sort(pin, -1).values[..., :-1]).sum()) == 0, "pinned neighbour table repeats an index"
                idx = pin
            elif self.record_decisions:
                self.decisions["knn"][tower] = idx.numpy()
        nbr = torch.gather(x[:, None].expand(B, N, N, D), 2, idx[..., None].expand(B, N, k, D))
        cen = x[:, :, None, :].expand_as(nbr)
        h = torch.cat([cen, nbr - cen], -1).reshape(B * N * k, 2 * D)
        for i in range(len(widths) - 1):
            nm = f"{scope}/conv{i+1}"
            # bf16 option with the dgcnn backbone: the edge convs behind the K = 6 lift and the point conv below take rounded
            # operands (the whole backward stays fp32 in the engine: DESIGN.md 4.5b)
            # (the relu in front of a max -- last edge conv, point conv -- is applied behind it: one sign per winner)
            h = self._layer(h, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", training, decay,
                            round_operands=self.bf16_lift and training and i >= 1, act=i < len(widths) - 2, relu_key=f"{tower}:{nm}")
        h = self._relu(self._max_over(h.reshape(B * N, k, -1), "slot", scope, tower), f"{tower}:{scope}/conv{len(widths) - 1}")
        nm = f"{scope}/conv{len(widths)}"
        h = self._layer(h, f"siamese/{nm}", f"{TOWER_PREFIX[tower]}/{nm}/bn", training, decay,
                        round_operands=self.bf16_lift and training, act=False, relu_key=f"{tower}:{nm}")   # (act=False: the key only names the rounded input)
        return self._relu(self._max_over(h.reshape(B, N, -1), "pool", scope, tower), f"{tower}:{nm}")

    def _backbone(self, *a):
        fn = self._pointnet if self.spec.backbone == "pointnet" else self._dgcnn
        if self.checkpoint and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            return checkpoint(fn, *a, use_reentrant=False)
        return fn(*a)

    def _mlp(self, x, scope, widths, tower, keep, training, decay, u):
        h = x
        for j in range(len(widths) - 1):
            nm = f"{scope}/fc{j+1}" if scope else f"fc{j+1}"
            if tower is None:
                h = self._layer(h, nm, nm + "/bn", training, decay, relu_key=f"p:{nm}")
            else:
                h = self._layer(h, "siamese/" + nm, f"{TOWER_PREFIX[tower]}/{nm}/bn", training, decay, relu_key=f"{tower}:{nm}")
        if keep is not None and training:
            kp = float(np.float32(keep))   # keep_prob enters the TF graph as a float32 constant
            h = h / kp * torch.floor(kp + u)
        nm = f"{scope}/fc{len(widths)}" if scope else f"fc{len(widths)}"
        return self._layer(h, nm if tower is None else "siamese/" + nm, None, training, decay, act=False)

    # -- decode -----------------------------------------------------------
    def angles(self, logits, tower=None):
        nb = self.spec.num_bins
        pi = torch.tensor(np.float32(np.pi), dtype=logits.dtype)
        cls = torch.argmax(logits[:, :nb], dim=1)
        pin = self._pin("yaw", None, tower) if tower is not None else None
        if pin is not None:
            with torch.no_grad():
                lg = logits[:, :nb]
                self.pin_report.append((f"yaw:{tower}", float((lg.amax(1) - lg.gather(1, pin[:, None])[:, 0]).max()), float(lg.abs().max()),
                                        int((cls != pin).sum()), pin.numel()))
            cls = pin
        elif tower is not None and self.record_decisions:
            self.decisions["yaw"][tower] = cls.numpy()
        res = (logits[:, nb:] * (pi / nb)).gather(1, cls[:, None])[:, 0]
        a = cls.to(logits.dtype) * (2.0 * pi / nb) + res
        return torch.remainder(a + pi, 2.0 * pi) - pi

    def _tower(self, pcs, tower, training, decay, u):
        s = self.spec
        cm = pcs.mean(1)
        f1 = self._backbone(pcs - cm[:, None], "transformer1/embedding", s.s1_conv, tower, training, decay)
        s1c = self._mlp(f1, "transformer1/mlp", list(s.s1_fc) + [3], tower, s.s1_keep, training, decay,
                        None if u is None else u[f"s1_{tower}"]) + cm
        f2 = self._backbone(pcs - s1c[:, None], "transformer2/embedding", s.s2_conv, tower, training, decay)
        o2 = self._mlp(f2, "transformer2/mlp", list(s.s2_fc) + [s.out_s2], tower, s.s2_keep, training, decay,
                       None if u is None else u[f"s2_{tower}"])
        s2c, lg = o2[:, :3] + s1c, o2[:, 3:]
        a = -self.angles(lg, tower)
        c, sn, z, o = torch.cos(a), torch.sin(a), torch.zeros_like(a), torch.ones_like(a)
        R = torch.stack([c, -sn, z, sn, c, z, z, z, o], -1).reshape(-1, 3, 3)
        emb = self._backbone(torch.bmm(pcs - s2c[:, None], R), "embedding", s.emb_conv, tower, training, decay)
        return emb, s1c, s2c, lg

    def forward(self, pcs1, pcs2, training=False, decay=None, u=None):
        s = self.spec
        self.ema_updates = {}
        e1, a1, b1, l1 = self._tower(pcs1, 0, training, decay, u)
        e2, a2, b2, l2 = self._tower(pcs2, 1, training, decay, u)
        net = self._mlp(torch.cat([e1, e2], 1), "", list(s.rem_fc) + [s.out_s2], None, s.rem_keep, training, decay,
                        None if u is None else u["rem"])
        return {
            "pred_s1_pc1centers": a1, "pred_s1_pc2centers": a2,
            "pred_s2_pc1centers": b1, "pred_s2_pc2centers": b2,
            "pred_pc1angle_logits": l1, "pred_pc2angle_logits": l2,
            "pred_translations": net[:, :3] + (b2 - b1),
            "pred_remaining_angle_logits": net[:, 3:],
        }

    # -- loss ---------------------------------------------------------------
    def _angle_loss(self, logits, target, pin=None, what=""):
        nb = self.spec.num_bins
        twopi = torch.tensor(np.float32(2 * np.pi), dtype=logits.dtype)
        apc = twopi / nb
        sh = torch.remainder(torch.remainder(target, twopi) + apc / 2, twopi)
        cls = (sh / apc).to(torch.int32)
        if pin is not None:
            # the class each target angle falls into is a DECISION (models/tp8.py:197: int32(shifted / angle_per_class)), taken for every entry
            # of the pair term's [B, B] target (:327): the residual label is a sawtooth in the angle, and an entry within a rounding of a class
            # boundary sits on the other tooth in another evaluation (label off by 2 pi / nb / (pi / nb) = 2).  Pinned: the other evaluation's
            # class, after checking that wherever it differs from this one's the angle is within rounding of the boundary between the two;
            # the residual then follows continuously (wrapped, since the last class's upper boundary is the wrap of the shifted angle).
            pc = torch.as_tensor(np.asarray(pin), dtype=torch.int32).reshape(cls.shape)
            with torch.no_grad():
                differ = pc != cls
                nd = int(differ.sum())
                frac = sh / apc - torch.round(sh / apc)   # distance to the nearest class boundary, in classes
                self.pin_report.append((f"losscls:{what}", float(frac[differ].abs().max()) if nd else 0.0, 1.0, nd, pc.numel()))
            cls = pc
            res = sh - (cls.to(logits.dtype) * apc + apc / 2)
            res = res - twopi * torch.round(res / twopi)
        else:
            res = sh - (cls.to(logits.dtype) * apc + apc / 2)
        cls0 = cls[:, 0].long()
        ce = F.cross_entropy(logits[:, :nb], cls0)
        pick = (logits[:, nb:] * F.one_hot(cls0, nb).to(logits.dtype)).sum(1)  # [B]
        lab = res / float(np.float32(np.pi / nb))  # [B,1] or [B,B]; Python floats enter the TF graph as float32 constants
        err = pick - lab  # broadcast to [B,B], as in the reference
        rl = F.huber_loss(err, torch.zeros_like(err), delta=1.0)
        return torch.stack([ce + 20.0 * rl, ce, rl])

    def _angle_losses(self, logits, target, term=None):
        pin = None
        if term is not None and self.pinned is not None and "loss_cls" in self.pinned:
            pin = self.pinned["loss_cls"][term]   # [2 variants: target, target + pi][B] or [2][B][B]
        a = self._angle_loss(logits, target, None if pin is None else pin[0], f"{term}:0")
        if self.spec.accept_inverted_angle:
            b = self._angle_loss(logits, target + float(np.float32(np.pi)), None if pin is None else pin[1], f"{term}:1")
            return a if bool(a[0] > b[0]) else b
        return a

    def loss(self, ep, translations, rel_angles, c1, c2, ang1, ang2):
        s = self.spec
        hub = lambda e, d: F.huber_loss(e, torch.zeros_like(e), delta=d)
        s1 = (hub(ep["pred_s1_pc1centers"] - c1, 1.0) + hub(ep["pred_s1_pc2centers"] - c2, 1.0)) / 2
        s2 = (hub(ep["pred_s2_pc1centers"] - c1, 1.0) + hub(ep["pred_s2_pc2centers"] - c2, 1.0)) / 2
        la1 = self._angle_losses(ep["pred_pc1angle_logits"], ang1, 0)
        la2 = self._angle_losses(ep["pred_pc2angle_logits"], ang2, 1)
        s3t = hub(ep["pred_translations"] - translations, 2.0)
        p1, p2 = self.angles(ep["pred_pc1angle_logits"], 0), self.angles(ep["pred_pc2angle_logits"], 1)
        tgt = (ang2 - ang1) - (p2 - p1)  # [B,1]-[B] -> [B,B]
        la3 = self._angle_losses(ep["pred_remaining_angle_logits"], tgt, 2)
        lt = s.early_stage_factor * (s1 + s2) + s3t
        la = s.early_stage_factor * ((la1[0] + la2[0]) / 2) + la3[0]
        return (lt + s.angle_factor * la) / translations.shape[0]


def _loss_global(self, ep, labels):
    """The loss on the all-gathered batch ("global_loss"): other ranks' rows enter as constants, this rank's with their graph."""
    def glob(t):
        parts = [x.detach() for x in self.sync.gather(t)]
        parts[self.sync.rank] = t
        return torch.cat(parts, 0)
    return self.loss({k: glob(v) for k, v in ep.items()}, *[glob(x) for x in labels])


TorchTp8.loss_global = _loss_global


def tf_adam(w, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """Same float32 constants as oracle/alignnet_ref.py adam_step (TF evaluates Adam in the variable's dtype)."""
    f = np.float32
    b1f, b2f = f(b1), f(b2)
    b1p, b2p = f(1), f(1)
    for _ in range(int(t)):
        b1p, b2p = f(b1p * b1f), f(b2p * b2f)
    lr_t = float(f(lr)) * math.sqrt(float(f(1) - b2p)) / float(f(1) - b1p)
    m = float(b1f) * m + float(f(1) - b1f) * g
    v = float(b2f) * v + float(f(1) - b2f) * g * g
    return w - lr_t * m / (v.sqrt() + float(f(eps))), m, v
