"""A NumPy stand-in for the ~70 `tf.*` symbols that the reference's graph-building code touches
(models/tp8.py, utils/tf_util.py, utils/tf_util_dgcnn.py), so that THAT CODE -- unmodified, imported from
/root/reference in the build container only -- can be executed and its outputs captured as fixtures
(tests/golden/make_graph_golden.py).

WHAT THIS IS AND IS NOT.  TensorFlow 1.8 is not installable here.  This file restates the published
semantics of the TF1 ops and of TF1's graph-mode naming rules; it is a stand-in for a LIBRARY, written by the same hands
as the oracles, so fixtures made with it do NOT pin parity formally (DESIGN.md 2 keeps saying "parity unpinned").  What it
does buy: the reference's own wiring -- layer order, scopes, `reuse=tf.AUTO_REUSE`, the `[B]` vs `[B,1]` broadcasts in the loss,
`tf.cond` on whole-batch scalars, which tensors feed which -- is executed by the machine from the reference's text instead of
being read by a person twice.  Nothing of this file is used by the product path or travels with a GPU run's results.

Model: a lazy dataflow graph like TF1's.  Building an op creates a Node (and evaluates it once on "build values" --
zeros for placeholders, initial values for variables -- only to know its static shape); `Session.run(fetches, feed_dict)`
evaluates the fetched nodes recursively with memoisation.  `tf.cond` builds both branches and evaluates only the taken one.
Names follow Graph.unique_name / name_scope / variable_scope / get_variable / tf.Variable / slot_creator as published for TF 1.x.

dtype policy: every float tensor is held as FLOAT (np.float32 as in the reference graph, or np.float64 for a "wide" run whose
only rounding is that of constants: a Python scalar meeting a tensor is first rounded to float32, as TF converts it)."""
import builtins
import contextlib
import math
import sys
import types

import numpy as np

FLOAT = np.float32          # set_float(np.float64) for the wide mode


def set_float(dt):
    global FLOAT
    FLOAT = dt


class DType:
    def __init__(self, name, kind):
        self.name, self.kind = name, kind   # kind: 'f', 'i', 'b'

    def np(self):
        return {"f": FLOAT, "b": np.bool_}.get(self.kind) or {"int32": np.int32, "int64": np.int64}[self.name]

    def __repr__(self):
        return "tf." + self.name


float32, float16, int32, int64, bool_ = DType("float32", "f"), DType("float16", "f"), DType("int32", "i"), DType("int64", "i"), DType("bool", "b")


class Dimension(int):
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    def as_list(self):
        return [int(d) for d in self]

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r


# ----------------------------------------------------------------------------------------------------------------- graph
class VarScope:
    def __init__(self, name, reuse):
        self.name, self.reuse = name, reuse


class _GraphImpl:
    def __init__(self):
        self.name_stack = ""
        self.names_in_use = {}
        self.var_scope = VarScope("", None)
        self.var_scope_counts = {}
        self.var_store = {}            # full name -> Variable (get_variable's store)
        self.variables = []            # every Variable in creation order (global variables collection)
        self.collections = {}
        self.summaries = []            # (tag, tensor)
        self.control_stack = []
        self.building_value = True

    def unique_name(self, name, mark_as_used=True):
        """Graph.unique_name: prefix with the current name stack, append _N on collision."""
        if self.name_stack:
            name = self.name_stack + "/" + name
        i = self.names_in_use.get(name, 0)
        if mark_as_used:
            self.names_in_use[name] = i + 1
        if i > 0:
            base = name
            while name in self.names_in_use:
                name = "%s_%d" % (base, i)
                i += 1
            if mark_as_used:
                self.names_in_use[name] = 1
        return name

    @contextlib.contextmanager
    def name_scope(self, name):
        old = self.name_stack
        if not name:
            new = ""                                   # name=None and name="" reset to the root scope
        elif name[-1] == "/":
            new = name[:-1]                            # an absolute, already-unique scope
        else:
            new = self.unique_name(name)
        self.name_stack = new
        try:
            yield (new + "/") if new else ""
        finally:
            self.name_stack = old


_graph = _GraphImpl()


def get_default_graph():
    return _graph


def reset_default_graph():
    global _graph
    _graph = _GraphImpl()


class _GraphCtx:
    def as_default(self):
        return contextlib.nullcontext()


def Graph():
    """tf.Graph(): the reference only ever does `with tf.Graph().as_default():`."""
    reset_default_graph()
    return _GraphCtx()


# ----------------------------------------------------------------------------------------------------------------- nodes
class OpInfo:
    def __init__(self, name, type_):
        self.name, self.type = name, type_


class Tensor:
    """A graph node producing one value."""
    __array_priority__ = 1000

    def __init__(self, fn, inputs, name=None, op_type="Op", dtype=None, build=True):
        g = _graph
        self.fn, self.inputs = fn, list(inputs)
        self.control_inputs = list(g.control_stack[-1]) if g.control_stack else []
        self.op = OpInfo(g.unique_name(name) if name else None, op_type)
        self.build_value = None
        if build:
            self.build_value = self._compute([_bv(i) for i in self.inputs])
        self._dtype = dtype

    def _compute(self, vals):
        return self.fn(*vals)

    # -- static info
    @property
    def name(self):
        return (self.op.name or "anon") + ":0"

    @property
    def dtype(self):
        if self._dtype is not None:
            return self._dtype
        k = np.asarray(self.build_value).dtype.kind
        return float32 if k == "f" else bool_ if k == "b" else (int64 if np.asarray(self.build_value).dtype == np.int64 else int32)

    def get_shape(self):
        return TensorShape(Dimension(d) for d in np.shape(self.build_value))

    shape = property(get_shape)

    # -- evaluation
    def eval(self, cache, sess):
        if self in cache:
            return cache[self]
        for c in self.control_inputs:
            c.eval(cache, sess)
        v = self._run([i.eval(cache, sess) if isinstance(i, Tensor) else i for i in self.inputs], cache, sess)
        cache[self] = v
        return v

    def _run(self, vals, cache, sess):
        return self.fn(*vals)

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    # -- operators
    def __add__(self, o): return _binary(np.add, self, o)
    def __radd__(self, o): return _binary(np.add, o, self)
    def __sub__(self, o): return _binary(np.subtract, self, o)
    def __rsub__(self, o): return _binary(np.subtract, o, self)
    def __mul__(self, o): return _binary(np.multiply, self, o)
    def __rmul__(self, o): return _binary(np.multiply, o, self)
    def __truediv__(self, o): return _binary(_div, self, o)
    def __rtruediv__(self, o): return _binary(_div, o, self)
    def __pow__(self, o): return _binary(np.power, self, o)
    def __neg__(self): return Tensor(np.negative, [self])
    def __gt__(self, o): return _binary(np.greater, self, o)
    def __lt__(self, o): return _binary(np.less, self, o)
    def __ge__(self, o): return _binary(np.greater_equal, self, o)

    def __getitem__(self, idx):
        return Tensor(lambda x: x[idx], [self])

    def __iter__(self):
        raise TypeError("Tensor objects are not iterable in graph mode")

    def __bool__(self):
        raise TypeError("Using a tf.Tensor as a Python bool is not allowed in graph mode")


def _div(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in "iu" and b.dtype.kind in "iu":
        return a // b
    return a / b


def _bv(x):
    return x.build_value if isinstance(x, Tensor) else x


def convert_to_tensor(v, like=None, dtype=None):
    """Python / NumPy value -> constant node.  Floats are rounded to float32 first (TF converts a Python scalar to the
    other operand's dtype, float32 everywhere in this graph), then held as FLOAT."""
    if isinstance(v, Tensor):
        return v
    a = np.asarray(v)
    if isinstance(v, (list, tuple)) and any(isinstance(e, Tensor) for e in v):
        return stack(list(v))
    if dtype is not None:
        a = a.astype(np.float32).astype(FLOAT) if dtype.kind == "f" else a.astype(dtype.np())
    elif a.dtype.kind == "f":
        a = a.astype(np.float32).astype(FLOAT)
    elif a.dtype.kind in "iu" and like is not None and like.dtype.kind == "f":
        a = a.astype(np.float32).astype(FLOAT)
    elif a.dtype.kind in "iu":
        a = a.astype(np.int32)
    return Tensor(lambda a=a: a, [], op_type="Const")


def _binary(f, a, b):
    ta, tb = isinstance(a, Tensor), isinstance(b, Tensor)
    a = a if ta else convert_to_tensor(a, like=b)
    b = b if tb else convert_to_tensor(b, like=a)
    return Tensor(lambda x, y: f(x, y), [a, b])


def _unary(f, name=None):
    def op(x, name=None):
        return Tensor(f, [convert_to_tensor(x)])
    return op


class Placeholder(Tensor):
    def __init__(self, dtype, shape):
        self._shape = tuple(int(s) for s in shape)
        self._pdtype = dtype
        super().__init__(None, [], op_type="Placeholder", dtype=dtype, build=False)
        self.build_value = np.zeros(self._shape, dtype.np())

    def _run(self, vals, cache, sess):
        raise KeyError("placeholder was not fed")


def placeholder(dtype, shape=()):
    return Placeholder(dtype, shape)


class Variable(Tensor):
    """tf.Variable: named by the NAME scope (`with ops.name_scope(name, "Variable")`), always a new variable.
    get_variable() builds the same object but names it by the VARIABLE scope (see below)."""

    def __init__(self, initial_value=None, name=None, trainable=True, dtype=None, _full_name=None):
        g = _graph
        if _full_name is not None:
            vname = _full_name[:-1] if _full_name.endswith("/") else _full_name   # name_scope("x/") uses the absolute name as is
            g.names_in_use[vname] = g.names_in_use.get(vname, 0) + 1
        else:
            with g.name_scope(name or "Variable") as sc:
                vname = sc[:-1]
        init = initial_value() if callable(initial_value) else initial_value
        self.initial = np.array(_bv(convert_to_tensor(init, dtype=dtype)))
        self.var_name, self.trainable = vname, trainable
        Tensor.__init__(self, None, [], op_type="VariableV2", build=False)
        self.op = OpInfo(vname, "VariableV2")
        self.build_value = self.initial
        g.variables.append(self)

    def _run(self, vals, cache, sess):
        return sess.values[self]

    def initialized_value(self):
        return self


AUTO_REUSE = "AUTO_REUSE"


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    """tf.variable_scope(str): variable scope = parent/name (shared by re-entry), name scope = unique_name(name) (NOT shared:
    the second `variable_scope("siamese", reuse=AUTO_REUSE)` is name scope `siamese_1`).  reuse is inherited.  With
    name_or_scope=None the default_name is uniquified among VARIABLE scopes (slot_creator's use)."""
    g = _graph
    old = g.var_scope
    if isinstance(name_or_scope, VarScope):
        raise NotImplementedError("re-entering a captured scope object is not used by the reference")
    if name_or_scope is None:
        prefix = default_name
        full = (old.name + "/" + prefix) if old.name else prefix
        if g.var_scope_counts.get(full, 0) > 0:
            idx = 1
            while g.var_scope_counts.get(full + "_%d" % idx, 0) > 0:
                idx += 1
            prefix += "_%d" % idx
        name = prefix
        ns_name = default_name
    else:
        name = ns_name = name_or_scope
    new_name = (old.name + "/" + name) if (old.name and name) else (name or old.name if not name else name)
    if name == "":
        new_name = old.name        # '' adds nothing to the variable scope ...
    g.var_scope_counts[new_name] = g.var_scope_counts.get(new_name, 0) + 1
    g.var_scope = VarScope(new_name, reuse if reuse is not None else old.reuse)
    try:
        with g.name_scope(ns_name):   # ... and resets the NAME scope to the root (Graph.name_scope(""))
            yield g.var_scope
    finally:
        g.var_scope = old


def get_variable(name, shape=None, initializer=None, dtype=None, trainable=True, **_):
    g = _graph
    sc = g.var_scope
    full = (sc.name + "/" + name) if sc.name else name
    if full in g.var_store:
        if not sc.reuse:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True or reuse=tf.AUTO_REUSE in VarScope?" % full)
        return g.var_store[full]
    if sc.reuse is True:
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % full)
    if isinstance(initializer, Tensor) or isinstance(initializer, np.ndarray):
        init = initializer
    else:
        init = initializer(tuple(int(s) for s in shape))
    v = Variable(init, trainable=trainable, dtype=dtype, _full_name=full)
    g.var_store[full] = v
    return v


def global_variables():
    return list(_graph.variables)


def trainable_variables():
    return [v for v in _graph.variables if v.trainable]


# ------------------------------------------------------------------------------------------------------------ initializers
_init_rng = np.random.default_rng(0)


def seed_initializers(seed):
    global _init_rng
    _init_rng = np.random.default_rng(seed)


def xavier_initializer(uniform=True, seed=None, dtype=None):
    """tf.contrib.layers.xavier_initializer = variance_scaling_initializer(factor=1.0, mode='FAN_AVG', uniform=True):
    fan_in = shape[-2] * receptive field, fan_out = shape[-1] * receptive field; limit = sqrt(3 * factor / ((fan_in + fan_out) / 2))."""
    def init(shape):
        rf = 1
        for d in shape[:-2]:
            rf *= d
        fan_in, fan_out = (shape[-2] * rf, shape[-1] * rf) if len(shape) >= 2 else (shape[-1], shape[-1])
        limit = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
        return _init_rng.uniform(-limit, limit, size=shape).astype(np.float32)
    return init


def constant_initializer(value=0.0):
    return lambda shape: np.full(shape, value, np.float32)


def zeros_initializer():
    return constant_initializer(0.0)


def ones_initializer():
    return constant_initializer(1.0)


def truncated_normal_initializer(stddev=1.0, **_):
    return lambda shape: np.clip(_init_rng.normal(0, stddev, size=shape), -2 * stddev, 2 * stddev).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------- ops
def constant(value, dtype=None, shape=None, name=None):
    a = np.asarray(value)
    if shape is not None:
        a = np.broadcast_to(a, tuple(int(s) for s in shape)).copy()
    if dtype is None:
        dtype = float32 if a.dtype.kind == "f" else (bool_ if a.dtype.kind == "b" else int32)
    return convert_to_tensor(a, dtype=dtype)


def zeros(shape, dtype=float32):
    return constant(0.0, dtype=dtype, shape=shape)


def identity(x, name=None):
    return Tensor(lambda v: v, [convert_to_tensor(x)])


class NoOp(Tensor):
    def __init__(self, deps=()):
        Tensor.__init__(self, lambda *a: None, list(deps), op_type="NoOp", build=False)


def no_op(name=None):
    return NoOp()


def group(*ops):
    return NoOp(ops)


@contextlib.contextmanager
def control_dependencies(ops):
    _graph.control_stack.append(list(ops or []))
    try:
        yield
    finally:
        _graph.control_stack.pop()


@contextlib.contextmanager
def device(_):
    yield


class CondOut(Tensor):
    def __init__(self, pred, t, f):
        Tensor.__init__(self, None, [], op_type="Merge", build=False)
        self.pred, self.t, self.f = pred, t, f
        self.build_value = _bv(t) if t is not None else None

    def eval(self, cache, sess):
        if self in cache:
            return cache[self]
        for c in self.control_inputs:
            c.eval(cache, sess)
        branch = self.t if builtins.bool(self.pred.eval(cache, sess)) else self.f
        v = branch.eval(cache, sess) if isinstance(branch, Tensor) else branch
        cache[self] = v
        return v


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None):
    """Graph-mode tf.cond: BOTH branch functions run at build time (so both create their variables); only the taken
    branch's ops are evaluated by Session.run."""
    true_fn, false_fn = true_fn or fn1, false_fn or fn2
    pred = convert_to_tensor(pred)
    with _graph.name_scope(name or "cond"):
        t, f = true_fn(), false_fn()
    if isinstance(t, (tuple, list)):
        return tuple(CondOut(pred, convert_to_tensor(a), convert_to_tensor(b)) for a, b in zip(t, f))
    wrap = lambda x: x if isinstance(x, Tensor) or x is None else convert_to_tensor(x)
    return CondOut(pred, wrap(t), wrap(f))


def expand_dims(x, axis=None, name=None, dim=None):
    axis = dim if axis is None else axis
    return Tensor(lambda v: np.expand_dims(v, axis), [convert_to_tensor(x)])


def squeeze(x, axis=None, name=None):
    ax = tuple(axis) if isinstance(axis, (list, tuple)) else axis
    return Tensor(lambda v: np.squeeze(v, axis=ax), [convert_to_tensor(x)], name=name or "Squeeze")


def tile(x, multiples, name=None):
    return Tensor(lambda v: np.tile(v, [int(m) for m in multiples]), [convert_to_tensor(x)])


def reshape(x, shape, name=None):
    return Tensor(lambda v: np.reshape(v, [int(s) for s in shape]), [convert_to_tensor(x)])


def transpose(x, perm=None, name=None):
    return Tensor(lambda v: np.transpose(v, perm), [convert_to_tensor(x)])


def concat(values, axis, name=None):
    vs = [convert_to_tensor(v) for v in values]
    return Tensor(lambda *a: np.concatenate(a, axis=axis), vs)


def stack(values, axis=0, name=None):
    if isinstance(values, Tensor):           # tf.stack(tensor) unpacks along axis 0 and packs again
        return identity(values)
    vs = [convert_to_tensor(v, like=next((u for u in values if isinstance(u, Tensor)), None)) for v in values]
    return Tensor(lambda *a: np.stack([np.asarray(x) for x in a], axis=axis), vs)


def _reduce(f):
    def op(x, axis=None, keepdims=False, name=None, keep_dims=None, reduction_indices=None):
        axis = reduction_indices if axis is None else axis
        kd = keepdims if keep_dims is None else keep_dims
        ax = tuple(axis) if isinstance(axis, (list, tuple)) else axis
        return Tensor(lambda v: f(v, axis=ax, keepdims=kd), [convert_to_tensor(x)])
    return op


reduce_mean, reduce_sum, reduce_max = _reduce(np.mean), _reduce(np.sum), _reduce(np.max)


def matmul(a, b, name=None):
    return Tensor(np.matmul, [convert_to_tensor(a), convert_to_tensor(b)])


def multiply(a, b, name=None):
    return _binary(np.multiply, a, b)


def minimum(a, b, name=None):
    return _binary(np.minimum, a, b)


def maximum(a, b, name=None):
    return _binary(np.maximum, a, b)


def mod(a, b, name=None):
    return _binary(np.mod, a, b)            # floor-mod: the result has the sign of the divisor (tf.mod = FloorMod)


def equal(a, b, name=None):
    return _binary(np.equal, a, b)


def where(c, x, y, name=None):
    return Tensor(np.where, [convert_to_tensor(c), convert_to_tensor(x), convert_to_tensor(y)])


abs = _unary(np.abs)
square = _unary(np.square)
cos, sin, acos = _unary(np.cos), _unary(np.sin), _unary(np.arccos)


def to_float(x, name=None):
    return Tensor(lambda v: np.asarray(v).astype(np.float32).astype(FLOAT) if np.asarray(v).dtype.kind != "f" else np.asarray(v).astype(FLOAT),
                  [convert_to_tensor(x)])


def to_int32(x, name=None):
    return Tensor(lambda v: np.trunc(v).astype(np.int32), [convert_to_tensor(x)])     # float -> int casts truncate toward zero


def cast(x, dtype, name=None):
    if dtype.kind == "f":
        return to_float(x)
    return Tensor(lambda v: np.trunc(v).astype(dtype.np()) if np.asarray(v).dtype.kind == "f" else np.asarray(v).astype(dtype.np()), [convert_to_tensor(x)])


def range(start, limit=None, delta=1, dtype=None, name=None):
    lo, hi = (0, start) if limit is None else (start, limit)
    a = np.arange(int(lo), int(hi), int(delta)).astype((dtype or int32).np())
    return convert_to_tensor(a, dtype=dtype or int32)


def argmax(x, axis=None, output_type=int64, name=None, dimension=None):
    axis = dimension if axis is None else axis
    return Tensor(lambda v: np.argmax(v, axis=axis).astype(output_type.np()), [convert_to_tensor(x)])   # first maximum, as TF


def gather_nd(params, indices, name=None):
    return Tensor(lambda p, i: p[tuple(np.moveaxis(np.asarray(i), -1, 0))], [convert_to_tensor(params), convert_to_tensor(indices)])


def gather(params, indices, name=None):
    return Tensor(lambda p, i: p[np.asarray(i)], [convert_to_tensor(params), convert_to_tensor(indices)])


def one_hot(indices, depth, on_value=1.0, off_value=0.0, axis=-1, dtype=None, name=None):
    assert axis == -1
    isint = isinstance(on_value, int) and isinstance(off_value, int)

    def f(i):
        out = np.where(np.asarray(i)[..., None] == np.arange(depth), on_value, off_value)
        return out.astype(np.int32) if isint else out.astype(FLOAT)
    return Tensor(f, [convert_to_tensor(indices)])


class MapFn(Tensor):
    def __init__(self, fn, elems):
        elems = convert_to_tensor(elems)
        self.elem = Placeholder(elems.dtype, np.shape(_bv(elems))[1:])
        self.elem.build_value = np.asarray(_bv(elems))[0]
        self.body = fn(self.elem)
        Tensor.__init__(self, None, [elems], op_type="map", build=False)
        n = np.shape(_bv(elems))[0]
        self.build_value = np.stack([np.asarray(_bv(self.body))] * n)

    def _run(self, vals, cache, sess):
        outs = []
        for e in np.asarray(vals[0]):
            local = dict(cache)
            local[self.elem] = e
            outs.append(np.asarray(self.body.eval(local, sess)))
        return np.stack(outs)


def map_fn(fn, elems, dtype=None, name=None, **_):
    return MapFn(fn, elems)


# ---- tf.nn
def _conv2d(x, k, strides, padding):
    """NHWC x HWIO, VALID, the only shapes the reference builds: kernel height 1, stride 1, kernel width = 1 or the full width."""
    assert padding == "VALID" and list(strides) == [1, 1, 1, 1] and k.shape[0] == 1
    kw = k.shape[1]
    w_out = x.shape[2] - kw + 1
    cols = np.stack([x[:, :, j:j + kw, :] for j in np.arange(w_out)], axis=2)          # [B,H,Wout,kw,Cin]
    return np.einsum("bhwkc,kco->bhwo", cols, k[0])


def _max_pool(x, ksize, strides, padding):
    assert padding == "VALID"
    _, kh, kw, _ = ksize
    _, sh, sw, _ = strides
    H, W = x.shape[1], x.shape[2]
    oh, ow = (H - kh) // sh + 1, (W - kw) // sw + 1
    out = np.empty((x.shape[0], oh, ow, x.shape[3]), x.dtype)
    for i in np.arange(oh):
        for j in np.arange(ow):
            out[:, i, j, :] = x[:, i * sh:i * sh + kh, j * sw:j * sw + kw, :].max(axis=(1, 2))
    return out


def _moments(x, axes, shift=None, name=None, keep_dims=False):
    """nn_impl.moments: mean, then the mean of the squared difference from that mean (two-pass, biased); the two outputs are the
    ops `<scope>/moments/Squeeze` and `<scope>/moments/Squeeze_1` -- the names ExponentialMovingAverage keys its slots on."""
    x = convert_to_tensor(x)
    ax = tuple(int(a) for a in axes)
    with _graph.name_scope(name or "moments"):
        mean_k = Tensor(lambda v: np.mean(v, axis=ax, keepdims=True), [x], name="mean")
        var_k = Tensor(lambda v, m: np.mean(np.square(v - m), axis=ax, keepdims=True), [x, mean_k], name="variance")
        return squeeze(mean_k, ax), squeeze(var_k, ax)


def _batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
    """nn_impl.batch_normalization: inv = rsqrt(var + eps) * scale;  x * inv + (offset - mean * inv)."""
    def f(x, m, v, b, g):
        inv = (1.0 / np.sqrt(v + np.asarray(variance_epsilon, np.float32).astype(FLOAT))) * g
        return x * inv + (b - m * inv)
    return Tensor(f, [convert_to_tensor(a) for a in (x, mean, variance, offset, scale)])


class Dropout(Tensor):
    """nn_ops.dropout: x / keep_prob * floor(keep_prob + random_uniform(noise_shape)).  The uniforms come from the session
    (Session.dropout_uniforms: consumed in graph-construction order, recorded for the fixtures)."""

    def __init__(self, x, keep_prob, noise_shape):
        assert noise_shape is None
        self.keep = keep_prob
        Tensor.__init__(self, None, [convert_to_tensor(x)], op_type="dropout", build=False)
        self.build_value = _bv(self.inputs[0])
        _graph.collections.setdefault("_dropout_nodes", []).append(self)   # graph-construction order

    def _run(self, vals, cache, sess):
        x = vals[0]
        u = sess.next_uniform(self, np.shape(x))
        keep = np.asarray(self.keep, np.float32).astype(FLOAT)
        return x / keep * np.floor(keep + u)


def _sparse_ce(_sentinel=None, labels=None, logits=None, name=None):
    def f(lg, lb):
        z = lg - lg.max(axis=-1, keepdims=True)
        lse = np.log(np.exp(z).sum(axis=-1))
        return lse - np.take_along_axis(z, np.asarray(lb)[..., None].astype(np.int64), axis=-1)[..., 0]
    return Tensor(f, [convert_to_tensor(logits), convert_to_tensor(labels)])


def _top_k(x, k=1, sorted=True, name=None):
    """values and indices of the k largest entries of the last axis; ties: the lower index first (TopKV2)."""
    x = convert_to_tensor(x)
    idx = Tensor(lambda v: np.argsort(-v, axis=-1, kind="stable")[..., :k].astype(np.int32), [x])
    vals = Tensor(lambda v, i: np.take_along_axis(v, i.astype(np.int64), axis=-1), [x, idx])
    return vals, idx


nn = types.SimpleNamespace(
    relu=_unary(lambda v: np.maximum(v, 0)),
    conv2d=lambda x, k, strides, padding, **_: Tensor(lambda a, b: _conv2d(a, b, strides, padding), [convert_to_tensor(x), convert_to_tensor(k)]),
    bias_add=lambda x, b, **_: _binary(np.add, x, b),
    max_pool=lambda x, ksize, strides, padding, name=None: Tensor(lambda v: _max_pool(v, ksize, strides, padding), [convert_to_tensor(x)]),
    avg_pool=None, conv1d=None, conv3d=None, conv2d_transpose=None, max_pool3d=None, avg_pool3d=None,
    moments=_moments, batch_normalization=_batch_normalization,
    dropout=lambda x, keep_prob, noise_shape=None, **_: Dropout(x, keep_prob, noise_shape),
    l2_loss=lambda x, name=None: Tensor(lambda v: np.sum(np.square(v)) / 2, [convert_to_tensor(x)]),
    sparse_softmax_cross_entropy_with_logits=_sparse_ce, top_k=_top_k,
)


# ---- tf.train
class ExponentialMovingAverage:
    """moving_averages.ExponentialMovingAverage(decay) with the defaults the reference uses (num_updates=None, zero_debias=False).
    apply([tensors]): a shadow per tensor via slot_creator.create_zeros_slot -> `variable_scope(None, primary.op.name + "/" + name)`
    + `get_variable("")`: named by the current VARIABLE scope + the primary's op name; update s -= (s - x) * (1 - decay)."""

    def __init__(self, decay, num_updates=None, zero_debias=False, name="ExponentialMovingAverage"):
        self.decay, self.name, self.averages = decay, name, {}

    def apply(self, var_list=None):
        updates = []
        for var in var_list:
            if var not in self.averages:
                init = np.zeros(np.shape(_bv(var)), np.float32)
                with variable_scope(None, default_name=var.op.name + "/" + self.name):
                    avg = get_variable("", initializer=init, trainable=False)
                self.averages[var] = avg
            updates.append(AssignSub(self.averages[var], var, convert_to_tensor(self.decay)))
        return group(*updates)

    def average(self, var):
        return self.averages.get(var)


class AssignSub(Tensor):
    def __init__(self, avg, value, decay):
        Tensor.__init__(self, None, [avg, value, decay], op_type="AssignSub", build=False)
        self.avg = avg

    def _run(self, vals, cache, sess):
        s, x, d = vals
        one = np.asarray(1.0, np.float32).astype(FLOAT)
        new = s - (s - x) * (one - d)
        sess.values[self.avg] = np.asarray(new, FLOAT)
        return new


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
    """learning_rate * decay_rate ^ (global_step / decay_steps), floor()ed exponent with staircase=True; computed in float32."""
    gs = to_float(global_step)
    p = gs / convert_to_tensor(float(decay_steps))
    if staircase:
        p = Tensor(np.floor, [p])
    return convert_to_tensor(float(learning_rate)) * Tensor(np.power, [convert_to_tensor(float(decay_rate)), p])


train = types.SimpleNamespace(ExponentialMovingAverage=ExponentialMovingAverage, exponential_decay=exponential_decay,
                              AdamOptimizer=None, MomentumOptimizer=None, Saver=None)


def _summary_scalar(tag, tensor, **_):
    _graph.summaries.append((tag, tensor))
    return tensor


summary = types.SimpleNamespace(scalar=_summary_scalar, merge_all=lambda: None, FileWriter=None, Summary=None)


def add_to_collection(name, value):
    _graph.collections.setdefault(name, []).append(value)


def get_collection(name, scope=None):
    return list(_graph.collections.get(name, []))


contrib = types.SimpleNamespace(layers=types.SimpleNamespace(xavier_initializer=xavier_initializer))


# ------------------------------------------------------------------------------------------------------------- session
class Session:
    def __init__(self, config=None):
        self.values = {}
        self.dropout_uniforms = None     # {dropout node: uniforms} supplied by the caller, else drawn and recorded
        self.drawn = []
        self.rng = np.random.default_rng(12345)

    def init_variables(self):
        for v in _graph.variables:
            self.values[v] = np.asarray(v.initial).astype(FLOAT if np.asarray(v.initial).dtype.kind == "f" else np.asarray(v.initial).dtype)

    def next_uniform(self, node, shape):
        if self.dropout_uniforms is not None and node in self.dropout_uniforms:
            u = np.asarray(self.dropout_uniforms[node])
        else:
            u = self.rng.uniform(size=shape).astype(np.float32)
        self.drawn.append((node, u))
        return u.astype(FLOAT)

    def run(self, fetches, feed_dict=None):
        cache = {}
        for k, v in (feed_dict or {}).items():
            a = np.asarray(v)
            cache[k] = a.astype(FLOAT) if a.dtype.kind == "f" else a
        self.drawn = []

        def ev(f):
            if isinstance(f, (list, tuple)):
                return type(f)(ev(x) for x in f)
            if isinstance(f, dict):
                return {k: ev(x) for k, x in f.items()}
            return f.eval(cache, self)
        return ev(fetches)


def global_variables_initializer():
    return NoOp()


def install():
    """Register this module as `tensorflow` (and the submodules the reference imports)."""
    me = sys.modules[__name__]
    me.bool = bool_      # tf.bool (this module uses builtins.bool itself)
    sys.modules["tensorflow"] = me
    for sub in ("tensorflow.python", "tensorflow.python.util", "tensorflow.python.util.nest"):
        sys.modules[sub] = types.ModuleType(sub)
    return me
