"""Model module of the drop-in (counterpart of the reference's models/tp8.py).  The TF graph-building
functions become eager calls into the MI355X engine (alignnet3d.Engine -> libalignnet_hip.so); the NumPy
helpers keep the reference's arithmetic, including quirk A6(i) (`classLogits2angle` does not de-normalise
the residual, tp8.py:229-244).  There is no CPU path: get_model raises when the engine cannot be created."""
from collections import namedtuple

import numpy as np

from config import configGlobal as cfg

Placeholder = namedtuple("Placeholder", "name shape dtype")
_engine = None


def engine():
    """The process-wide engine bound to `configGlobal` (created on first use; train.py:190-227)."""
    global _engine
    if _engine is None:
        import alignnet3d
        _engine = alignnet3d.Engine(cfg)
    return _engine


def bind_engine(e):
    global _engine
    _engine = e


def placeholder_inputs(batch_size, num_point):
    """Shapes/dtypes of the 8 feeds (tp8.py:13-23).  The engine needs no static batch dimension."""
    c = cfg.data.num_channels
    spec = [("pcs1", (batch_size, num_point, c)), ("pcs2", (batch_size, num_point, c)), ("translations", (batch_size, 3)),
            ("rel_angles", (batch_size, 1)), ("pc1_centers", (batch_size, 3)), ("pc2_centers", (batch_size, 3)),
            ("pc1_angles", (batch_size, 1)), ("pc2_angles", (batch_size, 1))]
    return tuple(Placeholder(n, s, np.float32) for n, s in spec)


def get_model(pcs1, pcs2, is_training=False, bn_decay=None):
    """Eval-mode forward -> end_points dict (tp8.py:135-158).  Training goes through Engine.train_step,
    which fuses forward, loss, backward and the optimiser like the reference's single sess.run."""
    if is_training:
        raise ValueError("training-mode forward is part of Engine.train_step (one fused step, train.py:368)")
    return engine().forward(pcs1, pcs2)


def get_loss(pcs1, pcs2, translations, rel_angles, pc1_centers, pc2_centers, pc1_angles, pc2_angles, end_points):
    """per_transform_loss of the LAST get_model call (tp8.py:401-407; only loss == 'separate' is shipped)."""
    assert cfg.training.loss.loss == "separate"
    labels = dict(translations=translations, rel_angles=rel_angles, pc1_centers=pc1_centers, pc2_centers=pc2_centers,
                  pc1_angles=pc1_angles, pc2_angles=pc2_angles)
    loss, _ = engine().eval_loss(labels, len(np.asarray(translations)))
    return loss


def class2angle(pred_cls, residual, to_label_format=True):
    angle = pred_cls * (2 * np.pi / float(cfg.model.angles.num_bins)) + residual
    if to_label_format and angle > np.pi:
        angle = angle - 2 * np.pi
    return angle


def classLogits2angle(logits, to_label_format=True):
    nb = cfg.model.angles.num_bins
    classes = np.argmax(logits[:, :nb], axis=1)
    return np.array([class2angle(c, r[c]) for c, r in zip(classes, logits[:, nb:])])
