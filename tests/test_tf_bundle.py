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


# ---- a checkpoint assembled byte by byte from the published formats, independently of tf_bundle.write_bundle -------------------------
def _crc32c_bitwise(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time: deliberately not the table code of tf_bundle."""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c_bitwise(data)
    return struct.pack("<I", (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)   # leveldb/TF crc32c::Mask


def _vi(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def _entry(shared, key_suffix, value):
    return _vi(shared) + _vi(len(key_suffix)) + _vi(len(value)) + key_suffix + value


def _snappy_with_copies(raw, needle):
    """Raw snappy stream of `raw`: literals, except that every later occurrence of `needle` becomes a copy with a 2-byte offset
    (tag kind 2) of its first occurrence -- so the reader's copy path is exercised, not only literals."""
    out = _vi(len(raw))
    first = raw.index(needle)
    pos = lit_start = 0

    def literal(chunk):
        o = b""
        while chunk:
            part, chunk = chunk[:60], chunk[60:]
            o += bytes([(len(part) - 1) << 2]) + part
        return o
    nxt = raw.find(needle, first + len(needle))
    while nxt != -1:
        out += literal(raw[lit_start:nxt])
        out += bytes([((len(needle) - 1) << 2) | 2]) + struct.pack("<H", nxt - first)
        lit_start = nxt + len(needle)
        nxt = raw.find(needle, lit_start)
    return out + literal(raw[lit_start:])


def test_reads_a_checkpoint_assembled_by_hand(tmp_path):
    """tensorflow/core/util/tensor_bundle + tensorflow/core/lib/io/table (LevelDB table format), written out byte by byte here:
    BundleHeaderProto / BundleEntryProto protobuf bytes, key-prefix-compressed entries, restart arrays, block trailers (type +
    masked CRC-32C from an independent bitwise implementation), one block stored snappy-compressed with real back-references, an
    index block of block handles, the empty metaindex block and the 48-byte footer with the table magic."""
    w = np.arange(6, dtype="<f4").reshape(2, 3) / 7
    adam = (np.arange(6, dtype="<f4").reshape(2, 3) + 1) / 13
    b64 = np.array([1.5, -2.25], "<f8")
    step = np.array(31200, "<i4")
    blobs = [("Variable", step, 3), ("a/weights", w, 1), ("a/weights/Adam", adam, 1), ("b", b64, 2)]   # (name, array, DataType enum)
    data, entries = b"", {}
    for name, arr, dt in blobs:
        raw = arr.tobytes()
        dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in arr.shape))      # repeated TensorShapeProto.Dim{size}
        e = b"\x08" + _vi(dt)                                       # 1: dtype
        e += b"\x12" + _vi(len(dims)) + dims                        # 2: shape
        if len(data):
            e += b"\x20" + _vi(len(data))                           # 4: offset (0 is the proto3 default and is omitted)
        e += b"\x28" + _vi(len(raw))                                # 5: size
        e += b"\x35" + _masked(raw)                                 # 6: crc32c, fixed32 (masked)
        entries[name] = e
        data += raw
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                      # num_shards = 1; version { producer = 1 }; endianness LITTLE = 0 omitted
    one_restart = struct.pack("<II", 0, 1)                          # restart offsets [0], count 1
    block1 = _entry(0, b"", header) + _entry(0, b"Variable", entries["Variable"]) + one_restart
    block2 = (_entry(0, b"a/weights", entries["a/weights"]) + _entry(9, b"/Adam", entries["a/weights/Adam"]) +   # shares "a/weights"
              _entry(0, b"b", entries["b"]) + one_restart)
    comp2 = _snappy_with_copies(block2, b"\x12\x02\x08")            # the Dim sub-message prefix occurs in every entry
    assert len(comp2) < len(block2) + 8 and tb._snappy_decompress(comp2) == block2
    out = b""

    def emit(body, ctype):
        nonlocal out
        off = len(out)
        out += body + bytes([ctype]) + _masked(body + bytes([ctype]))
        return _vi(off) + _vi(len(body))
    h1 = emit(block1, 0)
    h2 = emit(comp2, 1)
    hmeta = emit(one_restart, 0)
    index = _entry(0, b"Variable", h1) + _entry(0, b"c", h2) + struct.pack("<III", 0, len(_entry(0, b"Variable", h1)), 2)   # "c" >= every key of block 2
    hidx = emit(index, 0)
    footer = hmeta + hidx
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "model-199")
    open(prefix + ".index", "wb").write(out)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    ents, hdr = tb.read_index(prefix)
    assert sorted(ents) == ["Variable", "a/weights", "a/weights/Adam", "b"] and hdr[1] == [1]
    assert ents["a/weights/Adam"]["shape"] == (2, 3) and ents["a/weights/Adam"]["offset"] == 4 + 24
    got = tb.read_bundle(prefix)
    assert got["Variable"].dtype == np.int32 and got["Variable"].shape == () and int(got["Variable"]) == 31200
    np.testing.assert_array_equal(got["a/weights"], w)
    np.testing.assert_array_equal(got["a/weights/Adam"], adam)
    np.testing.assert_array_equal(got["b"], b64)
    assert got["b"].dtype == np.float64
    # the writer must produce what this independent reader-side knowledge expects as well: same CRCs, same entry bytes
    assert tb.mask_crc(tb.crc32c(data)) == struct.unpack("<I", _masked(data))[0]
    assert tb._make_entry(1, (2, 3), 4, 24, struct.unpack("<I", _masked(w.tobytes()))[0]) == entries["a/weights"]
    # one flipped byte inside the compressed block is caught by the block checksum
    bad = bytearray(out)
    bad[len(block1) + 5 + 3] ^= 0x40
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_index(prefix)
