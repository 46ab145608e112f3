"""Config surface of the drop-in (counterpart of the reference's config.py:9-91): a process-wide mutable
`configGlobal` namespace, defaults deep-merged with the user's JSON, plus the derived keys the rest of the
code reads (`name`, `data.basename`, `logging.logdir`, `data.ntrain`, `data.nval`).  The defaults are the
values of the reference's configs/default.json, kept here as a literal so that the package is self-contained;
any reference config file (configs/SynthCars.json, ...) can be passed to --config unchanged."""
import json
import os

_DEFAULTS = {
    "data": {"basepath": "/home/gross/data/SynthCars", "num_channels": 3},
    "gpu_index": 0,
    "model": {
        "model": "tp8", "backbone": "pointnet",
        "options": {
            "angle_factor": 1.0, "early_stage_factor": 0.1,
            "s1transformer": [[128, 128, 256], [[512, 256], 0.7]],
            "s2transformer": [[64, 64, 64, 128, 1024], [[512, 256], 0.7]],
            "embedding": [64, 64, 64, 128, 1024],
            "remaining_transform_prediction": [[512, 256], 0.7],
        },
        "num_points": 1024,
        "angles": {"num_bins": 36, "accept_inverted_angle": False},
    },
    "logging": {"basedir": "/home/gross/models/alignnet"},
    "evaluation": {"save_every_epoch": True},
    "training": {
        "batch_size": 64, "num_epochs": 100,
        "optimizer": {"optimizer": "adam"},
        "learning_rate": 0.01,
        "lr_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5},
        "bn_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5, "init": 0.5, "clip": 0.99},
        "loss": {"loss": "separate", "options": {"soft_angle_classes": False, "soft_angle_classes_sigma_in_degree": 5.0}},
        "pretraining": {"model": ""},
    },
}


class NameSpace(object):
    """Attribute view of a nested dict; `.has(key)` as in the reference (config.py:28-29)."""

    def has(self, key):
        return key in vars(self)

    def reset(self):
        vars(self).clear()

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, NameSpace) else v) for k, v in vars(self).items()}

    def merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                cur = vars(self).get(k)
                if not isinstance(cur, NameSpace):   # a scalar default replaced by a sub-tree
                    cur = NameSpace()
                    vars(self)[k] = cur
                cur.merge(v)
            else:
                vars(self)[k] = v
        return self

    def _lines(self, indent):
        out = []
        for k, v in vars(self).items():
            if isinstance(v, NameSpace):
                out.append("%s%s:" % (" " * indent, k))
                out.extend(v._lines(indent + 4))
            else:
                out.append("%s%s: %s" % (" " * indent, k, v))
        return out

    def __repr__(self):
        return "config:\n" + "\n".join(self._lines(4))


configGlobal = NameSpace()


def reset_config():
    configGlobal.reset()
    configGlobal.merge(json.loads(json.dumps(_DEFAULTS)))


reset_config()


def _read_split(path):
    return [int(line.rstrip()) for line in open(path)]


def load_config(filename):
    assert filename.endswith(".json")
    name = os.path.basename(filename)[:-5]
    with open(filename) as fh:
        configGlobal.merge(json.load(fh))
    cfg = configGlobal
    vars(cfg)["name"] = name
    vars(cfg.data)["basename"] = os.path.basename(cfg.data.basepath)
    vars(cfg.logging)["logdir"] = cfg.logging.basedir + "/" + name
    if cfg.evaluation.has("special") and cfg.evaluation.special.mode == "icp":
        vars(cfg.logging)["logdir"] = cfg.logging.basedir + "/icp_%s/%s" % (cfg.data.basename, name)
    vars(cfg.data)["ntrain"] = len(_read_split(cfg.data.basepath + "/split/train.txt"))
    vars(cfg.data)["nval"] = len(_read_split(cfg.data.basepath + "/split/val.txt"))
    return cfg


def save_config(filename):
    assert filename.endswith(".json")
    with open(filename, "w") as fh:
        json.dump(configGlobal.to_dict(), fh)
