"""A small deferred-execution graph with the slice of the TF 1.x surface the
reference's model functions use (placeholders, variable scopes, elementwise
ops, reshape/concat/slice, resize_images).

The reference builds a TensorFlow graph through ``tools_wscale/GAN.py`` and
evaluates it with ``sess.run(sampler, feed_dict=...)`` (e.g.
GAN/multipassGAN-out.py:446).  Here the same calls record ``Node`` objects;
``session.Session.run`` fuses them into HIP kernel launches.
Nothing in this module touches the GPU.
"""
import contextlib
import math

import numpy as np


class GraphError(Exception):
    pass


_default_graph = []


class Node(object):
    """A symbolic NHWC tensor.  ``shape`` uses None for the unknown batch size."""

    _counter = [0]

    def __init__(self, op, inputs=(), shape=None, **attrs):
        self.op = op
        self.inputs = list(inputs)
        self.shape = tuple(shape) if shape is not None else None
        self.attrs = attrs
        Node._counter[0] += 1
        self.id = Node._counter[0]
        self.name = attrs.get("name") or "%s_%d" % (op, self.id)
        self.scope = _default_graph[0].scope_name() if _default_graph else ""     # variable scope at creation

    def get_shape(self):
        return _Shape(self.shape)

    def __repr__(self):
        return "<Node %s %s %s>" % (self.name, self.op, self.shape)

    # the reference writes `_dens + x` on tensors (multipassGAN-out.py:330-332)
    def __add__(self, other):
        return add(self, other)

    __radd__ = __add__


class _Shape(tuple):
    def as_list(self):
        return list(self)


# ----------------------------------------------------------------------------
# variables and scopes (tf.variable_scope / tf.get_variable)
# ----------------------------------------------------------------------------
class VariableSpec(object):
    def __init__(self, name, shape, kind):
        self.name, self.shape, self.kind = name, tuple(int(s) for s in shape), kind


class Graph(object):
    def __init__(self):
        self.scope = []
        self.variables = {}      # name -> VariableSpec, in creation order

    def scope_name(self):
        return "/".join(self.scope)

    def get_variable(self, name, shape, kind):
        full = "/".join(self.scope + [name])
        if full in self.variables:
            spec = self.variables[full]
            if spec.shape != tuple(int(s) for s in shape):
                raise GraphError("variable %s reused with shape %s, was %s" % (full, tuple(shape), spec.shape))
        else:
            spec = VariableSpec(full, shape, kind)
            self.variables[full] = spec
        return Node("variable", shape=spec.shape, var=spec.name, name=spec.name)


_default_graph.append(Graph())


def get_default_graph():
    return _default_graph[0]


def reset_default_graph():
    _default_graph[0] = Graph()
    return _default_graph[0]


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    g = get_default_graph()
    parts = [p for p in str(name).split("/") if p]
    g.scope.extend(parts)
    try:
        yield g.scope_name()
    finally:
        del g.scope[len(g.scope) - len(parts):]


def get_variable_scope():
    return get_default_graph().scope_name()


def get_variable(name, shape, kind="weight"):
    return get_default_graph().get_variable(name, shape, kind)


# ----------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------
def placeholder(shape, name=None):
    """tf.placeholder(tf.float32, shape): a flat feed buffer; the batch dimension is None."""
    return Node("placeholder", shape=shape, name=name)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def reshape(x, shape):
    shape = [(-1 if s is None else int(s)) for s in shape]
    known = _numel([s for s in shape if s != -1])
    if shape.count(-1) > 1:
        raise GraphError("reshape: more than one -1 in %r" % (shape,))
    out = []
    for s in shape:
        if s != -1:
            out.append(s)
        elif x.shape is not None and None not in x.shape:
            out.append(_numel(x.shape) // known)
        else:
            out.append(None)
    return Node("reshape", [x], shape=out, target=tuple(shape))


def concat(values, axis=-1):
    values = list(values)
    rank = len(values[0].shape)
    axis = axis % rank
    if axis != rank - 1:
        raise GraphError("concat: only the channel axis is supported")
    shape = list(values[0].shape)
    shape[-1] = sum(v.shape[-1] for v in values)
    return Node("concat", values, shape=shape)


def random_normal_like(x, channels, stddev, seed=0):
    """tf.random_normal(shape of x with `channels` channels, mean 0, stddev) (GAN.noise, GAN.py:624-631): fresh values
    every run; `seed` selects the stream (the reference draws from TensorFlow's global generator)"""
    shape = list(x.shape)
    shape[-1] = int(channels)
    return Node("random_normal", [x], shape=shape, stddev=float(stddev), seed=None if seed is None else int(seed))


def slice_channels(x, begin, size):
    """tf.slice(x, [0,0,0,begin], [-1,h,w,size]) (multipassGAN-out.py:330)."""
    shape = list(x.shape)
    shape[-1] = size
    return Node("slice", [x], shape=shape, begin=int(begin), size=int(size))


def slice_flat(x, count):
    """tf.slice(x, [0,0], [N, count]) on a flat [N, n] tensor (multipassGAN-4x.py:582-583)."""
    return Node("slice_flat", [x], shape=(x.shape[0], int(count)), count=int(count))


def add(a, b):
    if not isinstance(a, Node) or not isinstance(b, Node):
        raise GraphError("add: both operands must be graph tensors")
    return Node("add", [a, b], shape=a.shape)


def relu(x):
    return Node("act", [x], shape=x.shape, act="relu")


relu.__name__ = "relu"


def tanh(x):
    return Node("act", [x], shape=x.shape, act="tanh")


def lrelu(x, leak=0.2, name="lrelu"):
    """module-level lrelu of tools_wscale/GAN.py:733-737."""
    return Node("act", [x], shape=x.shape, act="lrelu", leak=leak)


def activation_name(fn):
    """Map an activation callable (as passed to GAN.convolutional_layer) to a kernel id."""
    if fn is None:
        return None
    if isinstance(fn, str):
        return fn
    nm = getattr(fn, "__name__", "")
    if nm in ("relu", "lrelu", "tanh"):
        return nm
    raise GraphError("unsupported activation function %r" % (fn,))


class Scalar(object):
    """A fed scalar plus a constant (``percentage - (j - 1)`` of the growing nets, multipassGAN-8x.py:828-833):
    the blend factors are read from the feed at run time, so one graph serves every training iteration."""

    def __init__(self, node, offset=0.0):
        self.node, self.offset = node, float(offset)

    def __sub__(self, k):
        return Scalar(self.node, self.offset - float(k))

    def __add__(self, k):
        return Scalar(self.node, self.offset + float(k))

    def value(self, feeds):
        for k, v in feeds.items():
            if k is self.node:
                return float(v) + self.offset
        raise GraphError("scalar %r was not fed" % (self.node,))


def scalar_placeholder(name=None):
    """tf.placeholder(tf.float32) for a scalar such as `percentage` (multipassGAN-8x.py:1018)"""
    return Scalar(Node("scalar", shape=(), name=name))


def lerp(x, y, t):
    """lerp(x, y, t) = x + (y - x) * clip(t, 0, 1) (multipassGAN-8x.py:598-599).  x None stands for
    tf.zeros_like(y); t is a float or a ``Scalar``."""
    if x is not None and tuple(x.shape[1:]) != tuple(y.shape[1:]):
        raise GraphError("lerp: shapes %s and %s differ" % (x.shape, y.shape))
    return Node("lerp", [y] if x is None else [x, y], shape=y.shape, t=t, zero_x=x is None)


def pixel_norm(x, epsilon=1e-8):
    return Node("pixel_norm", [x], shape=x.shape, eps=float(epsilon))


def minibatch_stddev(x, group_size=4):
    """GAN.minibatch_stddev_layer (GAN.py:476-488)"""
    return Node("minibatch_stddev", [x], shape=(x.shape[0], x.shape[1], x.shape[2], x.shape[3] + 1), group_size=int(group_size))


def advect(source, vel, flags, dt, order, strength=0.0, start_bz=15):
    """GAN.advect (GAN.py:347-418): `source` [n,h,w,c] carried by the (MAC) velocity channels (x, y) of `vel`
    [n,hv,wv,>=2] over dt * (+1, 0, -1) per frame triple; order 1 semi-Lagrangian, order 2 MacCormack (`flags`: cells
    below 0.2 are fluid)"""
    if len(source.shape) != 4 or len(vel.shape) != 4 or vel.shape[3] < 2:
        raise GraphError("advect: source %s / velocity %s" % (source.shape, vel.shape))
    ins = [source, vel] + ([flags] if flags is not None else [])
    return Node("advect", ins, shape=source.shape, dt=float(dt), order=int(order), strength=float(strength),
                start_bz=int(start_bz))


def resize_images(x, size, method=0):
    """tf.image.resize_images(x, [oh, ow], method) with TF1 legacy coordinates."""
    oh, ow = int(size[0]), int(size[1])
    return Node("resize", [x], shape=(x.shape[0], oh, ow, x.shape[3]), oh=oh, ow=ow, method=int(method))


def max_pool(x, k=2, s=2):
    """tf.nn.max_pool(x, [1,k,k,1], [1,s,s,1], VALID) (GAN.py:152-159)"""
    if x.shape[1] < k or x.shape[2] < k:
        raise GraphError("max_pool: window %d does not fit %s" % (k, x.shape))
    return Node("max_pool", [x], shape=(x.shape[0], (x.shape[1] - k) // s + 1, (x.shape[2] - k) // s + 1, x.shape[3]),
                k=int(k), s=int(s))


def avg_pool(x, k=2, s=2):
    if k != 2 or s != 2:
        raise GraphError("avg_pool: only 2x2 stride 2 is implemented")
    return Node("avg_pool", [x], shape=(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]))


def conv2d(x, w, stride, wscale):
    """tf.nn.conv2d(x, W * wscale, SAME) (GAN.py:664-668,686-691)."""
    kh, kw, cin, cout = w.shape
    if x.shape[3] != cin:
        raise GraphError("conv2d: input has %s channels, weights expect %d" % (x.shape[3], cin))
    sh, sw = stride
    oh = -(-x.shape[1] // sh)
    ow = -(-x.shape[2] // sw)
    return Node("conv2d", [x, w], shape=(x.shape[0], oh, ow, cout), stride=(sh, sw), wscale=float(wscale))


def conv2d_transpose(x, w, stride, wscale):
    """tf.nn.conv2d_transpose(x, W * wscale, [N, H*sh, W*sw, Cout], strides, SAME) with W[kh,kw,Cout,Cin] (GAN.py:703-708)"""
    kh, kw, cout, cin = w.shape
    if x.shape[3] != cin:
        raise GraphError("conv2d_transpose: input has %s channels, weights expect %d" % (x.shape[3], cin))
    sh, sw = stride
    return Node("conv2d_transpose", [x, w], shape=(x.shape[0], x.shape[1] * sh, x.shape[2] * sw, cout), stride=(sh, sw),
                wscale=float(wscale))


def depth_to_space(x, r):
    """tf.depth_to_space (GAN.py:559)"""
    r = int(r)
    if x.shape[3] % (r * r):
        raise GraphError("depth_to_space: %s channels, block size %d" % (x.shape[3], r))
    return Node("depth_to_space", [x], shape=(x.shape[0], x.shape[1] * r, x.shape[2] * r, x.shape[3] // (r * r)), r=r)


def bias_add(x, b):
    return Node("bias_add", [x, b], shape=x.shape)


def batch_norm(x, gamma, beta, mean, var, eps=1e-3, training=False):
    return Node("batch_norm", [x, gamma, beta, mean, var], shape=x.shape, eps=eps, training=training)


def flatten(x):
    n = x.shape[1] * x.shape[2] * x.shape[3]
    return Node("reshape", [x], shape=(x.shape[0], n), target=(-1, n))


def matmul(x, w, wscale):
    return Node("matmul", [x, w], shape=(x.shape[0], w.shape[1]), wscale=float(wscale))


def he_wscale(shape, gain=math.sqrt(2.0)):
    """np.float32(gain / sqrt(prod(shape[:-1]))) (GAN.py:664-667)."""
    return float(np.float32(gain / np.sqrt(np.prod(shape[:-1]))))
