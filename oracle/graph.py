"""Eager numpy tape with the reference's tf_utils op names -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Lets oracle/models.py read like the
reference's buildModel() bodies: each method below stands in for the tf_utils.py
function of the same name (file:line in each docstring) and records a closure for
the reverse pass, i.e. what tf.train.AdamOptimizer.minimize() derives by autodiff.
"""
from collections import OrderedDict
import math
import numpy as np
from . import ops


class Node:
    __slots__ = ('v', 'g', 'needs_grad')

    def __init__(self, v, needs_grad=True):
        self.v = v
        self.g = None
        self.needs_grad = needs_grad

    def acc(self, g):
        if not self.needs_grad:
            return
        self.g = g.copy() if self.g is None else self.g + g


class Tape:
    """One forward/backward evaluation.  `variables` maps TF variable names ('e0/w',
    'fc1/Matrix', 'pre_image0/e0/b', ...) to arrays; missing ones are created with the
    reference initialisers from `rng` (tf_utils.py:58-65, 74-80, 91-95)."""

    def __init__(self, variables=None, rng=None, dtype=np.float32, sign_override=None, warp_override=None):
        # sign_override: optional list (one entry per activation call, in call order) of sign arrays to
        # use in the activation's backward instead of sign(x).  lrelu'/relu' are discontinuous at 0,
        # so a pre-activation within rounding noise of 0 may legitimately get either slope depending
        # on summation order; parity tests pass the device's pattern for exactly those elements.
        self.sign_override = sign_override
        # warp_override: optional list (one entry per resample_layer call) of sampling coordinates to use instead
        # of the computed ones.  The sampler's gradient w.r.t. the coordinates jumps where a coordinate crosses
        # an integer (floor picks another cell); parity tests substitute the device's coordinates for exactly
        # the samples whose cell differs, after checking that the coordinates themselves agree to rounding.
        self.warp_override = warp_override
        self.warp_inputs = []     # sampling coordinates, in resample_layer call order
        self.act_inputs = []      # pre-activation arrays, in activation call order
        self.vars = OrderedDict() if variables is None else variables
        self.rng = rng
        self.dtype = dtype
        self.grads = OrderedDict()
        self.used = []            # variable names in creation/use order (TF trainable_variables order)
        self._scope = []
        self._back = []

    # ----- scopes / variables
    def variable_scope(self, name):
        tape = self

        class _S:
            def __enter__(self_s):
                tape._scope.append(name)

            def __exit__(self_s, *a):
                tape._scope.pop()
        return _S()

    def _var(self, name, shape, init):
        full = '/'.join(self._scope + [name])
        if full not in self.vars:
            if self.rng is None:
                raise KeyError('variable %s missing and no rng given' % full)
            self.vars[full] = init(shape).astype(self.dtype)
        v = self.vars[full]
        assert tuple(v.shape) == tuple(shape), (full, v.shape, shape)
        if full not in self.used:
            self.used.append(full)
        return full, v.astype(self.dtype, copy=False)

    def _acc_var(self, full, g):
        self.grads[full] = g if full not in self.grads else self.grads[full] + g

    def const(self, v):
        return Node(np.asarray(v, dtype=self.dtype), needs_grad=False)

    # ----- tf_utils ops
    def conv2d_msra(self, x, output_dim, k_h, k_w, d_h, d_w, name):
        """tf_utils.py:70-84."""
        cin = x.v.shape[-1]
        std = math.sqrt(2. / float(k_h * k_w * cin))
        with self.variable_scope(name):
            wn, w = self._var('w', (k_h, k_w, cin, output_dim), lambda s: ops.truncated_normal(self.rng, s, std))
            bn, b = self._var('b', (output_dim,), lambda s: np.zeros(s, np.float32))
        y = Node(ops.conv2d_fwd(x.v, w, b, d_h, d_w))

        def back():
            dx, dw, db = ops.conv2d_bwd(x.v, w, y.g, d_h, d_w, need_dx=x.needs_grad)
            self._acc_var(wn, dw)
            self._acc_var(bn, db)
            if x.needs_grad:
                x.acc(dx)
        self._rec([y], back)
        return y

    def deconv2d_msra(self, x, output_shape, k_h, k_w, d_h, d_w, name):
        """tf_utils.py:87-98 (no bias)."""
        cin = x.v.shape[-1]
        std = math.sqrt(2.0 / float(k_h * k_w * cin) * float(d_h) * float(d_w))
        with self.variable_scope(name):
            wn, w = self._var('w', (k_h, k_w, output_shape[-1], cin), lambda s: ops.random_normal(self.rng, s, std))
        y = Node(ops.deconv2d_fwd(x.v, w, (output_shape[1], output_shape[2]), d_h, d_w))

        def back():
            dx, dw = ops.deconv2d_bwd(x.v, w, y.g, d_h, d_w)
            self._acc_var(wn, dw)
            x.acc(dx)
        self._rec([y], back)
        return y

    def linear_msra(self, x, output_size, name):
        """tf_utils.py:54-67."""
        fan_in = x.v.shape[-1]
        std = math.sqrt(2. / float(fan_in))
        with self.variable_scope(name):
            mn, m = self._var('Matrix', (fan_in, output_size), lambda s: ops.random_normal(self.rng, s, std))
            bn, b = self._var('b', (output_size,), lambda s: np.zeros(s, np.float32))
        y = Node(ops.linear_fwd(x.v, m, b))

        def back():
            dx, dm, db = ops.linear_bwd(x.v, m, y.g)
            self._acc_var(mn, dm)
            self._acc_var(bn, db)
            x.acc(dx)
        self._rec([y], back)
        return y

    def _absact(self, x, kind, leak=0.2):
        y = Node(ops.absact_fwd(x.v, kind, leak))
        idx = len(self.act_inputs)
        self.act_inputs.append(x.v)
        if self.sign_override is not None and self.sign_override[idx] is not None:
            f1, f2 = ops.act_coeffs(kind, leak)
            sgn = self.sign_override[idx].astype(x.v.dtype)
            self._rec([y], lambda: x.acc(y.g * (x.v.dtype.type(f1) + x.v.dtype.type(f2) * sgn)))
        else:
            self._rec([y], lambda: x.acc(ops.absact_bwd(x.v, y.g, kind, leak)))
        return y

    def lrelu(self, x, leak=0.2, name='lrelu'):
        """tf_utils.py:29-33."""
        return self._absact(x, 'lrelu', leak)

    def relu(self, x, name='relu'):
        """tf_utils.py:25-27."""
        return self._absact(x, 'relu')

    def tanh(self, x):
        """tf.nn.tanh (main_model.py:79)."""
        y = Node(ops.tanh_fwd(x.v))
        self._rec([y], lambda: x.acc(ops.tanh_bwd(y.v, y.g)))
        return y

    def warp_pts_layer(self, flow, name='warp_pts'):
        """tf_utils.py:35-38."""
        y = Node(ops.warp_pts_layer(flow.v))
        self._rec([y], lambda: flow.acc(y.g))
        return y

    def resample_layer(self, src, warp, name='tgt_img'):
        """tf_utils.py:40-42."""
        idx = len(self.warp_inputs)
        self.warp_inputs.append(warp.v)
        wv = warp.v
        if self.warp_override is not None and self.warp_override[idx] is not None:
            wv = self.warp_override[idx].astype(warp.v.dtype)
        y = Node(ops.resampler_fwd(src.v, wv))

        def back():
            dd, dw = ops.resampler_bwd(src.v, wv, y.g, need_ddata=src.needs_grad)
            if src.needs_grad:
                src.acc(dd)
            warp.acc(dw)
        self._rec([y], back)
        return y

    def euclidean_loss(self, a, b):
        """tf_utils.py:18-19."""
        y = Node(ops.euclidean_loss_fwd(a.v, b.v))

        def back():
            g = ops.euclidean_loss_bwd(a.v, b.v, float(y.g))
            a.acc(g)
            b.acc(-g)
        self._rec([y], back)
        return y

    def l1_loss(self, a, b):
        """tf_utils.py:22-23."""
        y = Node(ops.l1_loss_fwd(a.v, b.v))

        def back():
            g = ops.l1_loss_bwd(a.v, b.v, float(y.g))
            a.acc(g)
            b.acc(-g)
        self._rec([y], back)
        return y

    def masked_euclidean_loss(self, a, b, mask):
        """reduce_mean(reduce_sum(pow((a-b)*mask, 2), 3))  (multiobject_appflow.py:239-242)."""
        d = (a.v - b.v) * mask.v
        n, h, w, _ = d.shape
        y = Node((d * d).sum(axis=3).mean(dtype=np.float64).astype(self.dtype))
        self._rec([y], lambda: a.acc(d * mask.v * self.dtype(2.0 * float(y.g) / (n * h * w))))
        return y

    # ----- glue (tf.concat / tf.split / tf.reshape / tf.tile and scalar arithmetic on losses)
    def concat(self, axis, values):
        y = Node(np.concatenate([t.v for t in values], axis=axis))
        sizes = np.cumsum([t.v.shape[axis] for t in values])[:-1]

        def back():
            for t, g in zip(values, np.split(y.g, sizes, axis=axis)):
                t.acc(g)
        self._rec([y], back)
        return y

    def split(self, x, num, axis):
        """tf.split: `num` equal parts, or a list of sizes (stands in for the tf.slice pairs of mv3d/nobg_dm.py:85-89)."""
        sections = num if not isinstance(num, (list, tuple)) else list(np.cumsum(num)[:-1])
        outs = [Node(p.copy(), needs_grad=x.needs_grad) for p in np.split(x.v, sections, axis=axis)]
        keep = list(outs)          # callers pop() from the returned list (main_model.py:131-137)

        def back():
            x.acc(np.concatenate([o.g if o.g is not None else np.zeros_like(o.v) for o in keep], axis=axis))
        self._rec(keep, back)
        return outs

    def reshape(self, x, shape):
        y = Node(x.v.reshape(shape))
        self._rec([y], lambda: x.acc(y.g.reshape(x.v.shape)))
        return y

    def tile(self, x, reps):
        y = Node(np.tile(x.v, reps))

        def back():
            g = y.g
            shp = []
            for r, s in zip(reps, x.v.shape):
                shp += [r, s]
            g = g.reshape(shp).sum(axis=tuple(range(0, 2 * len(reps), 2)))
            x.acc(g)
        self._rec([y], back)
        return y

    def scale(self, x, c):
        y = Node(x.v * self.dtype(c))
        self._rec([y], lambda: x.acc(y.g * self.dtype(c)))
        return y

    def multiply(self, x, m):
        """tf.multiply(x, mask) with a broadcast one-channel constant mask (mv3d/bg_nodm.py:91)."""
        y = Node(x.v * m.v, needs_grad=x.needs_grad)
        self._rec([y], lambda: x.acc(y.g * m.v) if x.needs_grad else None)
        return y

    def add(self, a, b):
        y = Node(a.v + b.v)

        def back():
            a.acc(y.g)
            b.acc(y.g)
        self._rec([y], back)
        return y

    # ----- reverse pass
    def _rec(self, outs, fn):
        self._back.append((outs, fn))

    def backward(self, loss):
        """Reverse pass; ops whose outputs received no gradient are skipped (TF prunes them:
        e.g. the dead a0/a1 of highdim_angle.py:8-9 get no gradient and no Adam slots)."""
        loss.g = np.asarray(1.0, dtype=self.dtype)
        for outs, fn in reversed(self._back):
            if any(o.g is not None for o in outs):
                fn()
        return self.grads
