"""TensorFlow tensor-bundle reader/writer (alignnet3d/tf_bundle.py).  Unpinned against TensorFlow itself (absent
here): round trips, known-answer CRC32C, table structure, and the SURVEY 8.A2 variable-name mapping."""
import os
import struct

import numpy as np
import pytest

from alignnet3d import tf_bundle as tb
from oracle import alignnet_ref as R


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283          # standard CRC-32C check value
    assert tb.crc32c(b"") == 0
    assert tb.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    assert tb.mask_crc(0) == 0xA282EAD8


def test_roundtrip_and_structure(tmp_path):
    rng = np.random.default_rng(0)
    spec = R.NetSpec()
    tensors = {}
    for name, shp in R.param_names(spec):      # 148 real variable names: exercises key prefix compression + several blocks
        tensors[name] = rng.normal(size=shp).astype(np.float32)
    tensors["Variable"] = np.asarray(31200, np.int32)
    tensors["beta1_power"] = np.asarray(0.5, np.float32)
    tensors["some/int64"] = np.arange(6, dtype=np.int64).reshape(2, 3)
    prefix = str(tmp_path / "model-199")
    tb.write_bundle(prefix, tensors, block_entries=16)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in tensors.values())
    entries, header = tb.read_index(prefix)
    assert header[1] == [1] and set(entries) == set(tensors)
    back = tb.read_bundle(prefix)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape
        np.testing.assert_array_equal(back[k], v)
    # corruption is detected
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
    idx = bytearray(raw)
    idx[10] ^= 0xFF
    open(prefix + ".index", "wb").write(idx)
    with pytest.raises(ValueError):
        tb.read_index(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError):
        tb.read_index(prefix)


def test_snappy_block_decoding():
    # literal "abcd", copy(offset 4, len 4) x2, literal "xyz": hand-assembled raw snappy stream
    payload = b"abcdabcdabcdxyz"
    stream = tb._put_varint(len(payload)) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((4 - 4) << 2) | 1 | (0 << 5), 4]) + \
        bytes([((4 - 1) << 2) | 2]) + struct.pack("<H", 4) + bytes([(3 - 1) << 2]) + b"xyz"
    assert tb._snappy_decompress(stream) == payload


def test_variable_mapping_follows_tf_names():
    """Checkpoint names as TF 1.x writes them (derivation in tf_bundle.tf_shadow_name): the EMA slot is named by the VARIABLE
    scope (`siamese/...` for both towers) followed by the moments op's NAME-scope path (`siamese/...` or `siamese_1/...`)."""
    assert tb.tf_shadow_name("siamese/transformer1/embedding/conv1/bn/moving_mean") == \
        "siamese/transformer1/embedding/conv1/bn/siamese/transformer1/embedding/conv1/bn/moments/Squeeze/ExponentialMovingAverage"
    assert tb.tf_shadow_name("siamese_1/transformer1/embedding/conv1/bn/moving_var") == \
        "siamese/transformer1/embedding/conv1/bn/siamese_1/transformer1/embedding/conv1/bn/moments/Squeeze_1/ExponentialMovingAverage"
    assert tb.tf_shadow_name("siamese_1/transformer2/mlp/fc2/bn/moving_mean") == \
        "siamese/transformer2/mlp/fc2/bn/siamese_1/transformer2/mlp/fc2/bn/moments/Squeeze/ExponentialMovingAverage"
    assert tb.tf_shadow_name("fc1/bn/moving_var") == "fc1/bn/fc1/bn/moments/Squeeze_1/ExponentialMovingAverage"
    assert tb.tf_shadow_name("fc1/weights") is None
    spec = R.NetSpec()
    engine_vars = [(n, (int(np.prod(s[:-1])) if len(s) > 1 else 1, s[-1]), not n.endswith(("moving_mean", "moving_var")))
                   for n, s in R.param_names(spec)]
    ckpt = []
    for n, _, _ in engine_vars:
        if n.endswith(("/bn/moving_mean", "/bn/moving_var")):
            ckpt.append(tb.tf_shadow_name(n))
        else:
            ckpt.append(n)
            ckpt.append(n + "/Adam")
            ckpt.append(n + "/Adam_1")
    ckpt += ["Variable", "beta1_power", "beta2_power"]
    assert len(set(ckpt)) == len(ckpt)
    # every tower-1 shadow shares the `siamese/<layer>/bn/` prefix with the tower-0 shadow of the same layer
    assert sum(c.startswith("siamese/embedding/conv3/bn/") and c.endswith("ExponentialMovingAverage") for c in ckpt) == 4
    assert not any(c.startswith("siamese_1/") and c.endswith("ExponentialMovingAverage") for c in ckpt)
    mapping, missing = tb.map_variables(engine_vars, ckpt)
    assert not missing and len(mapping) == len(engine_vars)
    assert len(set(mapping.values())) == len(mapping)
    assert mapping["siamese_1/embedding/conv3/bn/moving_var"] == \
        "siamese/embedding/conv3/bn/siamese_1/embedding/conv3/bn/moments/Squeeze_1/ExponentialMovingAverage"
    assert mapping["siamese/embedding/conv3/bn/moving_mean"] == \
        "siamese/embedding/conv3/bn/siamese/embedding/conv3/bn/moments/Squeeze/ExponentialMovingAverage"
    assert mapping["fc1/bn/moving_mean"] == "fc1/bn/fc1/bn/moments/Squeeze/ExponentialMovingAverage"
    # a checkpoint without tower-1 statistics: exactly those are reported missing
    mapping2, missing2 = tb.map_variables(engine_vars, [c for c in ckpt if "siamese_1/" not in c])
    assert missing2 and all(m.startswith("siamese_1/") for m in missing2)
    # another outer spelling with the same unambiguous tail is still accepted
    alt = [c.replace("siamese/embedding/conv3/bn/siamese_1/", "x/siamese_1/") for c in ckpt]
    mapping3, missing3 = tb.map_variables(engine_vars, alt)
    assert not missing3 and mapping3["siamese_1/embedding/conv3/bn/moving_var"].startswith("x/siamese_1/")


@pytest.mark.gpu
def test_engine_export_import_roundtrip(gpu_required, tmp_path):
    import alignnet3d
    from tests.helpers import small_cfg, oracle_params
    cfg = small_cfg(N=128)
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    eng.set_step(77)
    prefix = str(tmp_path / "model-3")
    names = tb.export_from_engine(eng, prefix)
    assert "siamese/transformer1/embedding/conv1/weights" in names
    t = tb.read_bundle(prefix)
    assert t["siamese/transformer1/embedding/conv1/weights"].shape == (1, 3, 1, 32)     # HWIO, utils/tf_util.py:148-152
    assert t["siamese/embedding/conv2/weights"].shape == (1, 1, 32, 64)
    assert "siamese/embedding/conv2/bn/siamese_1/embedding/conv2/bn/moments/Squeeze_1/ExponentialMovingAverage" in t   # TF's slot name
    assert not any(k.endswith(("moving_mean", "moving_var")) for k in t)
    eng2 = alignnet3d.Engine(cfg)
    mapping, missing = tb.load_into_engine(eng2, prefix)
    assert not missing and eng2.state()["step"] == 77
    d = R.synth_pairs(4, 128, dtype=np.float32)
    a, b = eng.forward(d["pcs1"], d["pcs2"]), eng2.forward(d["pcs1"], d["pcs2"])
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    eng.close(); eng2.close()
