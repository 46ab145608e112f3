"""TensorFlow tensor-bundle (V2 checkpoint) reader / writer in pure Python -- SURVEY.md 8(f) row 2.

The reference saves `model-<epoch>.{index,data-00000-of-00001}` with tf.train.Saver (train.py:220,317-322) and the
released models (README.md:92) are in that format.  Format as published in tensorflow/core/util/tensor_bundle and
tensorflow/core/lib/io/table (a LevelDB table):

  <prefix>.index   table file: data blocks + index block + 48-byte footer (metaindex handle, index handle, padding,
                   magic 0xdb4775248b80fb57).  Block = entries [shared|unshared|value_len varint32, key delta, value]
                   + restart array + trailer (1-byte compression type, 4-byte masked CRC32C).  Key "" holds the
                   BundleHeaderProto, every other key is a tensor name with a BundleEntryProto value
                   (1 dtype, 2 shape, 3 shard_id, 4 offset, 5 size, 6 crc32c).
  <prefix>.data-00000-of-00001   raw little-endian tensor bytes.

PARITY STATUS: unpinned -- there is no TensorFlow in the build environment and the reference ships no checkpoint, so
this module is checked by round trips through its own writer, by structural tests (footer magic, CRCs, prefix
compression) and against the variable naming of SURVEY 8.A2.  `load_into_engine` reports every variable it could not
match instead of guessing silently."""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 4: np.uint8, 6: np.int8, 5: np.int16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- CRC32C (Castagnoli), masked as in leveldb/tensorflow --------------------------------------------------------
def _make_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _make_table()


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / protobuf ---------------------------------------------------------------------------------------------
def _get_varint(buf, pos):
    shift = out = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """Minimal protobuf wire parser: {field: [values]} with varints as int, length-delimited as bytes, fixed as int."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_entry(buf):
    p = _parse_proto(buf)
    shape = []
    for sh in p.get(2, []):
        for dim in _parse_proto(sh).get(2, []):
            shape.append(_signed(_parse_proto(dim).get(1, [0])[0]))
    return dict(dtype=p.get(1, [0])[0], shape=tuple(shape), shard=p.get(3, [0])[0], offset=p.get(4, [0])[0], size=p.get(5, [0])[0],
                crc=p.get(6, [None])[0], sliced=7 in p)


def _make_entry(dtype_id, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# ---- snappy (raw format) decompressor: only needed if a table was written with compression ---------------------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
        else:
            if kind == 1:
                ln = ((tag >> 2) & 7) + 4
                off = ((tag >> 5) << 8) | buf[pos]
                pos += 1
            elif kind == 2:
                ln = (tag >> 2) + 1
                off = struct.unpack_from("<H", buf, pos)[0]
                pos += 2
            else:
                ln = (tag >> 2) + 1
                off = struct.unpack_from("<I", buf, pos)[0]
                pos += 4
            if off == 0 or off > len(out):
                raise ValueError("corrupt snappy stream")
            for _ in range(ln):
                out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---- table reader ------------------------------------------------------------------------------------------------
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    body, ctype, crc = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if verify and mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return body


def _block_entries(body):
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(body, pos)
        unshared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + body[pos:pos + unshared]
        pos += unshared
        yield key, body[pos:pos + vlen]
        pos += vlen


def read_index(prefix, verify=True):
    """{tensor name: entry dict} + header, from <prefix>.index."""
    path = prefix + ".index"
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        fsize = f.tell()
        if fsize < 48:
            raise ValueError("%s is too small to be a table file" % path)
        f.seek(fsize - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
            raise ValueError("%s: bad table magic (not a TensorFlow V2 checkpoint index)" % path)
        _, p = _get_varint(footer, 0)
        _, p = _get_varint(footer, p)
        ioff, p = _get_varint(footer, p)
        isize, p = _get_varint(footer, p)
        entries, header = {}, None
        for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
            boff, q = _get_varint(handle, 0)
            bsize, q = _get_varint(handle, q)
            for key, val in _block_entries(_read_block(f, boff, bsize, verify)):
                if key == b"":
                    header = _parse_proto(val)
                else:
                    entries[key.decode()] = _parse_entry(val)
    return entries, header


def read_bundle(prefix, verify=True):
    """{tensor name: ndarray} for every non-sliced tensor of the checkpoint."""
    entries, header = read_index(prefix, verify)
    nshards = (header or {}).get(1, [1])[0]
    out = {}
    files = {}
    try:
        for name, e in entries.items():
            if e["sliced"]:
                continue   # partitioned variables do not occur in the reference graph
            if e["dtype"] not in _DTYPES:
                continue
            if e["shard"] not in files:
                files[e["shard"]] = open("%s.data-%05d-of-%05d" % (prefix, e["shard"], nshards), "rb")
            f = files[e["shard"]]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if verify and e["crc"] is not None and mask_crc(crc32c(raw)) != e["crc"]:
                raise ValueError("tensor %s: data checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ---- table writer (export, and the round-trip tests) -----------------------------------------------------------------
def _build_block(items, restart_interval=16):
    body, restarts, last = bytearray(), [], b""
    for i, (key, val) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_bundle(prefix, tensors, block_entries=64):
    """Write {name: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001 (single shard, no compression)."""
    names = sorted(tensors)
    items = [(b"", b"\x08\x01\x10\x00\x1a\x02\x08\x01")]   # BundleHeaderProto{num_shards:1, endianness:LITTLE, version{producer:1}}
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as df:
        for n in names:
            a = np.asarray(tensors[n])
            if a.ndim and not a.flags.c_contiguous:   # (ascontiguousarray would turn a 0-d scalar into shape (1,))
                a = np.ascontiguousarray(a)
            dt = np.dtype(a.dtype).newbyteorder("=")
            if dt not in _DTYPE_IDS:
                raise ValueError("unsupported dtype %s for %s" % (a.dtype, n))
            raw = a.astype(dt.newbyteorder("<"), copy=False).tobytes()
            df.write(raw)
            items.append((n.encode(), _make_entry(_DTYPE_IDS[dt], a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        index_items = []

        def emit(body):
            off = f.tell()
            trailer = b"\x00"
            f.write(body + trailer + struct.pack("<I", mask_crc(crc32c(body + trailer))))
            return off, len(body)

        for i in range(0, len(items), block_entries):
            chunk = items[i:i + block_entries]
            off, size = emit(_build_block(chunk))
            index_items.append((chunk[-1][0], _put_varint(off) + _put_varint(size)))
        moff, msize = emit(_build_block([]))
        ioff, isize = emit(_build_block(index_items, restart_interval=1))
        footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))


# ---- variable-name mapping -----------------------------------------------------------------------------
def tf_shadow_name(name):
    """TF 1.x checkpoint name of a BatchNorm EMA shadow, given the engine's `<tower>/<layer>/bn/moving_{mean,var}`.

    batch_norm_template (utils/tf_util.py:455-492) calls `ema.apply([batch_mean, batch_var])` on the two output TENSORS of
    `tf.nn.moments(..., name='moments')`, whose op names live in the NAME scope: `<tower>/<layer>/bn/moments/Squeeze` (mean)
    and `.../Squeeze_1` (variance), tower = `siamese` or `siamese_1` (the second `tf.variable_scope("siamese", reuse=AUTO_REUSE)`
    of models/tp8.py:140-143 re-enters the same VARIABLE scope under a uniquified name scope).  ExponentialMovingAverage.apply
    creates the shadow with slot_creator.create_zeros_slot -> `variable_scope(None, primary.op.name + "/ExponentialMovingAverage")`
    + `get_variable("")`, i.e. the slot is named by the current VARIABLE scope -- `siamese/<layer>/bn` for BOTH towers -- followed
    by the primary's full op name:
        siamese/<layer>/bn/ + <tower>/<layer>/bn/moments/Squeeze[_1]/ExponentialMovingAverage
    The pair head sits at the top level (scope '' at models/tp8.py:154): `fc1/bn/fc1/bn/moments/Squeeze/ExponentialMovingAverage`."""
    for leaf, op in (("/bn/moving_mean", "Squeeze"), ("/bn/moving_var", "Squeeze_1")):
        if name.endswith(leaf):
            inner = name[: -len(leaf)] + "/bn"                 # name scope of the moments op
            parts = inner.split("/", 1)
            var_scope = ("siamese/" + parts[1]) if parts[0] in ("siamese", "siamese_1") else inner
            return "%s/%s/moments/%s/ExponentialMovingAverage" % (var_scope, inner, op)
    return None


def map_variables(engine_vars, ckpt_names):
    """engine variable name -> checkpoint tensor name.  weights / biases / beta / gamma keep their graph names; the EMA shadows
    follow tf_shadow_name().  Should another TF version spell the slot's outer scope differently, a shadow is also accepted by
    its unambiguous tail `/<tower>/<layer>/bn/moments/Squeeze[_1]/ExponentialMovingAverage`.  Returns (mapping, missing)."""
    ck = set(ckpt_names)
    mapping, missing = {}, []
    for name, _shape, _tr in engine_vars:
        if name in ck:
            mapping[name] = name
            continue
        want = tf_shadow_name(name)
        if want is not None:
            if want in ck:
                mapping[name] = want
                continue
            tail = want[want.index("/bn/") + len("/bn/"):]    # <tower>/<layer>/bn/moments/...
            pick = [c for c in ck if c == tail or c.endswith("/" + tail)]
            if len(pick) == 1:
                mapping[name] = pick[0]
                continue
        missing.append(name)
    return mapping, missing


def load_into_engine(engine, prefix, load_step=True, strict=True):
    """Restore an engine from a TF checkpoint prefix (train.py:252,268,281).  Conv kernels are stored HWIO
    ([1,3,1,C] / [1,1,Cin,Cout]) and are flattened to the engine's 2-D layout."""
    tensors = read_bundle(prefix)
    mapping, missing = map_variables(engine.variables(), tensors.keys())
    if missing and strict:
        raise KeyError("checkpoint %s lacks %d variables, e.g. %s" % (prefix, len(missing), missing[:3]))
    shapes = {n: s for n, s, _ in engine.variables()}
    for name, src in mapping.items():
        a = np.asarray(tensors[src], np.float32)
        r, c = shapes[name]
        if a.size != r * c:
            raise ValueError("%s: checkpoint shape %s does not match %s" % (name, a.shape, (r, c)))
        engine.set_variable(name, a.reshape(r, c))
    if load_step:
        for key in ("Variable", "global_step"):   # train.py:195 `batch = tf.Variable(0)` is saved as "Variable"
            if key in tensors and tensors[key].size == 1:
                engine.set_step(int(np.ravel(tensors[key])[0]))
                break
    return mapping, missing


def export_from_engine(engine, prefix):
    """Write the engine's variables under the names the reference's tf.train.Saver uses (conv kernels back in HWIO, EMA
    shadows under tf_shadow_name()) plus the global step `Variable` (train.py:195)."""
    out = {}
    for name, (r, c), _ in engine.variables():
        a = engine.get_variable(name).reshape(r, c)
        name = tf_shadow_name(name) or name
        if name.endswith("/weights") and "/conv" in name:
            a = a.reshape(1, 3, 1, c) if (r == 3 and name.endswith("conv1/weights")) else a.reshape(1, 1, r, c)
        elif r == 1:
            a = a.reshape(c)
        out[name] = a.astype(np.float32)
    out["Variable"] = np.asarray(engine.state()["step"], np.int32)
    write_bundle(prefix, out)
    return sorted(out)
