"""A second, independent evaluator with oracle.graph.Tape's interface: torch CPU ops +
autograd in float64.  Running oracle.models builders on it checks the oracle's hand-written
reverse pass (closures in oracle/graph.py) against autograd over whole model graphs."""
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
    def __init__(self, variables):
        self.params = OrderedDict((k, torch.tensor(np.asarray(v, np.float64), requires_grad=True))
                                  for k, v in variables.items())
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
        return TNode(torch.tensor(np.asarray(v, np.float64)), False)

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
        return TNode(flow.t + torch.tensor(ops.coords(h, w, n, np.float64)))

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
