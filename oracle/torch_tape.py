"""A second, independent CPU evaluator with oracle.graph.Tape's interface: torch CPU ops + autograd -- TEST INFRASTRUCTURE ONLY.

Two uses: (1) float64: running the oracle.models builders on it checks the numpy oracle's hand-written reverse pass
(closures in oracle/graph.py) against autograd over whole model graphs (tests/test_oracle_models.py, tests/golden/make_golden.py);
(2) float32: bench.py's `cpu_baseline` times it on the host cores -- the same restated TF-1.3 graph (SAME padding,
conv2d_transpose as input-gradient, tf.contrib.resampler, abs-based lrelu, TF-Adam) on a multi-threaded CPU convolution
library (oneDNN behind torch), which is what "the reference graph on the node's host cores" looks like when TensorFlow 1.3
itself cannot run offline.  PARITY UNPINNED like the rest of oracle/."""
from collections import OrderedDict
import numpy as np
import torch
import torch.nn.functional as F
from oracle import ops


class TNode:
    def __init__(self, t, needs_grad=True):
        self.t = t
        self.needs_grad = needs_grad

    @property
    def v(self):            # shape queries used by the builders
        return self.t


def _conv_same(x, w, b, s):
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = ops.same_pads(x.shape[1], kh, s)
    _, pl, pr = ops.same_pads(x.shape[2], kw, s)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=s).permute(0, 2, 3, 1)


def _deconv_same(x, w, out_hw, s):
    kh, kw = w.shape[0], w.shape[1]
    _, pt, _ = ops.same_pads(out_hw[0], kh, s)
    _, pl, _ = ops.same_pads(out_hw[1], kw, s)
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=s)
    return y[:, :, pt:pt + out_hw[0], pl:pl + out_hw[1]].permute(0, 2, 3, 1)


class TorchTape:
    def __init__(self, variables, dtype=np.float64, params=None):
        self.dtype = dtype
        self.params = params if params is not None else OrderedDict(
            (k, torch.tensor(np.asarray(v, dtype), requires_grad=True)) for k, v in variables.items())
        self._scope = []

    def variable_scope(self, name):
        tape = self

        class _S:
            def __enter__(s):
                tape._scope.append(name)

            def __exit__(s, *a):
                tape._scope.pop()
        return _S()

    def _p(self, name):
        return self.params['/'.join(self._scope + [name])]

    def const(self, v):
        return TNode(torch.tensor(np.asarray(v, self.dtype)), False)

    def conv2d_msra(self, x, output_dim, k_h, k_w, d_h, d_w, name):
        with self.variable_scope(name):
            return TNode(_conv_same(x.t, self._p('w'), self._p('b'), d_h))

    def deconv2d_msra(self, x, output_shape, k_h, k_w, d_h, d_w, name):
        with self.variable_scope(name):
            return TNode(_deconv_same(x.t, self._p('w'), (output_shape[1], output_shape[2]), d_h))

    def linear_msra(self, x, output_size, name):
        with self.variable_scope(name):
            return TNode(x.t @ self._p('Matrix') + self._p('b'))

    def lrelu(self, x, leak=0.2, name='lrelu'):
        return TNode(0.5 * (1 + leak) * x.t + 0.5 * (1 - leak) * x.t.abs())

    def relu(self, x, name='relu'):
        return TNode(0.5 * x.t + 0.5 * x.t.abs())

    def tanh(self, x):
        return TNode(torch.tanh(x.t))

    def warp_pts_layer(self, flow, name='warp_pts'):
        n, h, w, _ = flow.t.shape
        return TNode(flow.t + torch.tensor(ops.coords(h, w, n, self.dtype)))

    def resample_layer(self, src, warp, name='tgt_img'):
        n, h, w, c = src.t.shape
        gx = 2 * warp.t[..., 0] / (w - 1) - 1
        gy = 2 * warp.t[..., 1] / (h - 1) - 1
        out = F.grid_sample(src.t.permute(0, 3, 1, 2), torch.stack((gx, gy), -1), mode='bilinear',
                            padding_mode='zeros', align_corners=True)
        return TNode(out.permute(0, 2, 3, 1))

    def euclidean_loss(self, a, b):
        return TNode(((a.t - b.t) ** 2).sum(3).mean())

    def l1_loss(self, a, b):
        return TNode((a.t - b.t).abs().sum(3).mean())

    def masked_euclidean_loss(self, a, b, mask):
        return TNode((((a.t - b.t) * mask.t) ** 2).sum(3).mean())

    def concat(self, axis, values):
        return TNode(torch.cat([v.t for v in values], dim=axis))

    def split(self, x, num, axis):
        if isinstance(num, (list, tuple)):
            return [TNode(p) for p in torch.split(x.t, list(num), dim=axis)]
        return [TNode(p) for p in torch.chunk(x.t, num, dim=axis)]

    def multiply(self, x, m):
        return TNode(x.t * m.t)

    def reshape(self, x, shape):
        return TNode(x.t.reshape(shape))

    def tile(self, x, reps):
        return TNode(x.t.repeat(*reps))

    def scale(self, x, c):
        return TNode(x.t * c)

    def add(self, a, b):
        return TNode(a.t + b.t)


def run_torch(builder, variables, feeds):
    t = TorchTape(variables)
    nodes = {k: t.const(v) for k, v in feeds.items()}
    out = builder(t, nodes)
    out['loss'].t.backward()
    outs = OrderedDict((k, n.t.detach().numpy()) for k, n in out.items())
    grads = OrderedDict((k, p.grad.numpy()) for k, p in t.params.items() if p.grad is not None)
    return outs, grads


class TorchTrainer:
    """fp32 train step of a builder's graph on torch CPU ops: forward, autograd reverse pass, TF ApplyAdam
    (epsilon outside the bias correction) over every variable with a gradient.  Used by bench.py's
    cpu_baseline only."""

    def __init__(self, builder, variables, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
        self.builder = builder
        self.tape = TorchTape(variables, dtype=np.float32)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m = {k: torch.zeros_like(p) for k, p in self.tape.params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in self.tape.params.items()}
        self.b1p, self.b2p = beta1, beta2

    def step(self, feeds):
        t = self.tape
        for p in t.params.values():
            p.grad = None
        out = self.builder(t, {k: t.const(v) for k, v in feeds.items()})
        loss = out['loss'].t
        loss.backward()
        alpha = self.lr * (1.0 - self.b2p) ** 0.5 / (1.0 - self.b1p)
        with torch.no_grad():
            for k, p in t.params.items():
                g = p.grad
                if g is None:
                    continue
                m, v = self.m[k], self.v[k]
                m.add_(g - m, alpha=1.0 - self.beta1)
                v.add_(g * g - v, alpha=1.0 - self.beta2)
                p.sub_(alpha * m / (v.sqrt() + self.eps))
        self.b1p *= self.beta1
        self.b2p *= self.beta2
        return float(loss)
