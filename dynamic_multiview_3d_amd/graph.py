"""Static graph of the appearance-flow train step, executed by libmv3d_hip.so.

The reference builds a TensorFlow-1.3 graph once (buildModel/build_loss) and then calls
sess.run([loss, train_op]) per iteration (multi_view_model/train.py:122).  This module is the
MI355X-side counterpart: the tf_utils-named functions (tf_utils.py of this package) append nodes
to the current Graph; `finalize()` lays every activation, gradient and variable out in HBM once;
`compile()` records the forward and the hand-scheduled reverse launch sequences into native plans
(mv3d_plan_*), so a step is two native calls plus one fused Adam kernel -- no autograd, no
per-op Python in the steady state.

Data layout in HBM
  * activations: NHWC fp32, one allocation per root Storage [rows, channels]; tf.concat /
    tf.split on the channel axis and tf.reshape are views (channel offset + pixel stride `ld`).
  * gradients: one buffer per root Storage with identical layout.  The gradient buffer of a
    tensor that is an activation output holds dL/d(pre-activation): the consumer's dgrad kernel
    multiplies by act'(output) in its epilogue (slope at 0 is f1).
  * variables: ONE flat fp32 buffer (TF variable order), plus flat grad / Adam m / Adam v buffers
    of the same layout -> Adam is one kernel launch and the data-parallel all-reduce runs on
    contiguous bucket views.
PyTorch is used for device memory (torch.empty), streams and torch.distributed only.
"""
import ctypes as C
import os
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH

_current = []


def current_graph():
    if not _current:
        raise RuntimeError("no Graph is active: build models inside `with Graph(...) as g:`")
    return _current[-1]


# =============================================================================================== storage / tensors
class Storage:
    """[rows, ch] fp32 region.  Either a root allocation, a channel slice of `parent`
    (set when tf.concat adopts it), or a reshaped alias of a dense root (`alias_of`)."""

    def __init__(self, rows, ch, alias_of=None):
        self.rows, self.ch = rows, ch
        self.parent, self.ch_off = None, 0
        self.alias_of = alias_of
        self.has_alias = False
        self.data = None
        self.grad = None
        self.needs_grad = False
        self.external = False
        if alias_of is not None:
            alias_of.has_alias = True

    def resolve(self):
        s, off = self, 0
        while s.parent is not None:
            off += s.ch_off
            s = s.parent
        return s, off

    def can_be_adopted(self):
        return self.parent is None and self.alias_of is None and not self.has_alias and not self.external


class Tensor:
    def __init__(self, graph, shape, storage=None, ch_off=0, producer=None, act=ACT_NONE, leak=0.2,
                 requires_grad=False, name=None):
        self.graph = graph
        self.shape = tuple(int(s) for s in shape)
        self.C = self.shape[-1]
        self.rows = int(np.prod(self.shape[:-1])) if len(self.shape) > 1 else 1
        self.storage = storage if storage is not None else Storage(self.rows, self.C)
        self.ch_off = ch_off
        self.producer = producer
        self.act, self.leak = act, leak          # activation these VALUES are the output of
        self.requires_grad = requires_grad
        self.name = name
        self.fused_into = None                   # set on a pre-activation tensor once lrelu()/tanh() fused it
        self.grad_consumers = 0
        # backward-emission state
        self.grad_written = False
        self.grad_masked = False
        if requires_grad:
            self.storage.needs_grad = True

    # ---- TF-like surface
    def get_shape(self):
        return list(self.shape)

    # ---- resolved addressing (valid after Graph.finalize())
    def _root(self):
        s = self.storage
        if s.alias_of is not None:
            root, off = s.alias_of.resolve()
            assert off == 0 and root.ch == s.alias_of.ch
            return root, 0, s.ch
        root, off = s.resolve()
        return root, off, root.ch

    @property
    def ld(self):
        return self._root()[2]

    @property
    def ptr(self):
        root, off, _ = self._root()
        return root.data.data_ptr() + 4 * (off + self.ch_off)

    @property
    def grad_ptr(self):
        root, off, _ = self._root()
        return root.grad.data_ptr() + 4 * (off + self.ch_off)

    def _view(self, buf):
        root, off, ld = self._root()
        v = buf.view(-1, ld)[:, off + self.ch_off: off + self.ch_off + self.C]
        return v.reshape(self.shape) if v.is_contiguous() else v.unflatten(0, self.shape[:-1])

    def value(self):
        """torch view of the activation (device)."""
        return self._view(self._root()[0].data)

    def grad_value(self):
        return self._view(self._root()[0].grad)

    def numpy(self):
        return self.value().detach().cpu().numpy().copy()

    def set(self, array):
        t = torch.as_tensor(np.ascontiguousarray(array, dtype=np.float32)) if not torch.is_tensor(array) else array
        self.value().copy_(t.reshape(self.shape), non_blocking=True)


class Variable:
    def __init__(self, name, shape, init):
        self.name, self.shape, self.init = name, tuple(shape), init
        self.size = int(np.prod(shape))
        self.offset = None
        self.graph = None
        self.has_grad = False

    @property
    def ptr(self):
        return self.graph.params.data_ptr() + 4 * self.offset

    @property
    def grad_ptr(self):
        return self.graph.grads.data_ptr() + 4 * self.offset

    def value(self):
        self.graph._settle()
        return self.graph.params[self.offset:self.offset + self.size].view(self.shape)

    def grad_value(self):
        self.graph._settle()
        return self.graph.grads[self.offset:self.offset + self.size].view(self.shape)


class ScalarExpr:
    """Weighted sum of loss terms; supports the arithmetic the reference applies to losses
    (main_model.py:144-152: `self.loss = 0.; self.loss += euclidean_loss(...) * factor`)."""

    def __init__(self, terms=()):
        self.terms = list(terms)          # [(weight, LossTerm)]

    def __add__(self, o):
        if isinstance(o, (int, float)):
            if o != 0:
                raise NotImplementedError("adding a non-zero constant to a loss")
            return ScalarExpr(self.terms)
        return ScalarExpr(self.terms + o.terms)

    __radd__ = __add__

    def __mul__(self, c):
        return ScalarExpr([(w * float(c), t) for w, t in self.terms])

    __rmul__ = __mul__


class LossTerm:
    def __init__(self, a, b, kind, mask=None, b_scale=1.0):
        self.a, self.b, self.kind, self.mask, self.b_scale = a, b, kind, mask, float(b_scale)


# =============================================================================================== nodes
class Node:
    def forward(self, g):
        raise NotImplementedError

    def backward(self, g):
        pass


def _epi(bias=None, act=ACT_NONE, leak=0.2, mask_of=None):
    """Epilogue for a kernel; mask_of = tensor whose activation derivative multiplies the result."""
    if mask_of is not None and mask_of.act != ACT_NONE:
        return _lib.epilogue(bias, act, leak, mask_of.act, mask_of.leak, mask_of.ptr, mask_of.ld)
    return _lib.epilogue(bias, act, leak)


def _ensure_premasked(g, t):
    """Make t's gradient buffer hold dL/d(pre-activation) if t is an activation output."""
    if t.act != ACT_NONE and not t.grad_masked:
        g.lib.act_bwd(t.rows, t.C, t.grad_ptr, t.ld, t.ptr, t.ld, t.grad_ptr, t.ld, t.act, t.leak, g.stream)
        t.grad_masked = True


def _note_grad_written(x, masked):
    x.grad_written = True
    x.grad_masked = masked


class ConvNode(Node):
    """conv2d_msra (tf_utils.py:70-84) / deconv2d_msra (tf_utils.py:87-98) with the following
    lrelu/relu/tanh fused into the epilogue."""

    def __init__(self, x, y, w, b, kh, kw, sh, sw, transposed):
        self.x, self.y, self.w, self.b = x, y, w, b
        self.k = (kh, kw, sh, sw)
        self.transposed = transposed
        self.act, self.leak = ACT_NONE, 0.2

    def geom(self):
        kh, kw, sh, sw = self.k
        img, feat = (self.y, self.x) if self.transposed else (self.x, self.y)
        n, h, w, c = img.shape
        return _lib.conv_geom(n, h, w, c, feat.shape[3], kh, kw, sh, sw, img.ld, feat.ld)

    wg_cus = 0          # CUs of this layer's filter-gradient launch (0 = the library's default, mv3d_set_wgrad_cus)

    def workspace_bytes(self, g):
        old = g.lib.set_wgrad_cus(self.wg_cus)
        try:
            return g.lib.conv_workspace_bytes(C.byref(self.geom()))
        finally:
            g.lib.set_wgrad_cus(old)

    def wgrad_partial_bytes(self, g):
        old = g.lib.set_wgrad_cus(self.wg_cus)
        try:
            return int(g.lib.conv_wgrad_workspace_bytes(C.byref(self.geom())))
        finally:
            g.lib.set_wgrad_cus(old)

    def forward(self, g):
        geom = self.geom()
        epi = _epi(self.b.ptr if self.b is not None else None, self.act, self.leak)
        fn = g.lib.deconv2d_fwd if self.transposed else g.lib.conv2d_fwd
        fn(C.byref(geom), self.x.ptr, self.w.ptr, self.y.ptr, C.byref(epi), g.ws_ptr, g.ws_bytes, g.stream)

    def backward(self, g):
        y, x = self.y, self.x
        if not y.grad_written:
            return
        _ensure_premasked(g, y)
        geom = self.geom()
        # filter / bias gradients are side work (only Adam reads them): own scratch, may run on the side stream
        kh, kw = self.k[0], self.k[1]
        us = 12.0 + 2.0 * geom.N * geom.Ho * geom.Wo * kh * kw * geom.C * geom.K / 120e6      # rough kernel time, microseconds
        ws_side = g.begin_side(us + 10.0, us if x.requires_grad else 0.0)
        ws_len = g.ws_bytes
        if g._finalizing:
            # mv3d_grad_finalize_*: the per-slab partial sums stay in THIS layer's region of the arena until the one batched
            # reduction (+ optimiser) at the end of the pass has read them
            ws_side, ws_len = g._part_arena.data_ptr() + self._part_off, self._part_bytes
        old_cus = g.lib.set_wgrad_cus(self.wg_cus)
        try:
            if self.transposed:
                g.lib.deconv2d_wgrad(C.byref(geom), x.ptr, y.grad_ptr, self.w.grad_ptr, ws_side, ws_len, g.stream)
            else:
                g.lib.conv2d_wgrad(C.byref(geom), x.ptr, y.grad_ptr, self.w.grad_ptr,
                                   self.b.grad_ptr if self.b is not None else None, ws_side, ws_len, g.stream)
        finally:
            g.lib.set_wgrad_cus(old_cus)      # process-global override: never leave it set behind a failed call
        g.end_side()
        self.w.has_grad = True
        if self.b is not None:
            self.b.has_grad = True
        if x.requires_grad:
            epi = _epi(mask_of=x)
            fn = g.lib.deconv2d_dgrad if self.transposed else g.lib.conv2d_dgrad
            fn(C.byref(geom), y.grad_ptr, self.w.ptr, x.grad_ptr, C.byref(epi), g.ws_ptr, g.ws_bytes, g.stream)
            _note_grad_written(x, x.act != ACT_NONE)


class LinearNode(Node):
    """linear_msra (tf_utils.py:54-67)."""

    def __init__(self, x, y, m, b):
        self.x, self.y, self.m, self.b = x, y, m, b
        self.act, self.leak = ACT_NONE, 0.2

    def workspace_bytes(self, g):
        return g.lib.fc_workspace_bytes(self.x.shape[0], self.x.C, self.y.C)

    chain = None        # Graph._find_fc_chains: the list of small layers this one is part of (one launch for all of them)

    def forward(self, g):
        x, y = self.x, self.y
        if self.chain is not None:
            # a chain of small layers (the angle MLP a0 -> a1 -> a2): ONE launch, recorded at the position of the LAST layer (the
            # first launch that touches its output: the fc hazard of the pipelined optimiser looks at that position)
            if self is not self.chain[-1]:
                return
            c = _lib.FcChain()
            first = self.chain[0]
            c.B, c.nlayers, c.in_, c.x_ld, c.x = first.x.shape[0], len(self.chain), first.x.C, first.x.ld, first.x.ptr
            for k, n in enumerate(self.chain):
                c.l[k].M, c.l[k].bias, c.l[k].y, c.l[k].y_ld, c.l[k].out = n.m.ptr, n.b.ptr, n.y.ptr, n.y.ld, n.y.C
                c.l[k].act, c.l[k].leak = n.act, n.leak
            g.lib.fc_chain_fwd(C.byref(c), g.stream)
            return
        epi = _epi(self.b.ptr, self.act, self.leak)
        g.lib.fc_fwd(x.shape[0], x.C, y.C, x.ptr, x.ld, self.m.ptr, y.ptr, y.ld, C.byref(epi), g.ws_ptr, g.ws_bytes, g.stream)

    def backward(self, g):
        x, y = self.x, self.y
        if not y.grad_written:
            return
        _ensure_premasked(g, y)
        B, fin, fout = x.shape[0], x.C, y.C
        # the fused kernel reads x and dy with 16-byte loads: a channel slice at an odd offset takes fc_wgrad + adam_step_dev
        if g._fusing and g.lib.fc_wgrad_adam_supported(B, fin, fout, x.ld, y.ld) and x.ptr % 16 == 0 and y.grad_ptr % 16 == 0:
            # Single-GPU step: the matrix gradient never goes to HBM -- ApplyAdam runs in the epilogue of the filter-gradient
            # kernel (mv3d_fc_wgrad_adam).  It rewrites the matrix, so it is recorded BEHIND the layer's data gradient (the
            # last reader of the old weights; the side stream forks after it); the bias gradient takes the ordinary path.
            if x.requires_grad:
                epi = _epi(mask_of=x)
                g.lib.fc_dgrad(B, fin, fout, y.grad_ptr, y.ld, self.m.ptr, x.grad_ptr, x.ld, C.byref(epi), g.ws_ptr, g.ws_bytes, g.stream)
                _note_grad_written(x, x.act != ACT_NONE)
            def emit(g=g, self=self, x=x, y=y, B=B, fin=fin, fout=fout):
                g.begin_side(60.0, 0.0, cls=2)          # its own stream: 400 MB of HBM traffic must not hold up the conv filter gradients
                off = 4 * self.m.offset
                g.lib.fc_wgrad_adam(B, fin, fout, x.ptr, x.ld, y.grad_ptr, y.ld, self.m.ptr, g.adam_m.data_ptr() + off,
                                    g.adam_v.data_ptr() + off, self.b.grad_ptr, g.adam_state.data_ptr() + 32, g.stream)
                g.end_side()
            g._deferred.append([g.fcadam_delay, emit])
            g._fused_vars.append(self.m)
            g._fused_nodes.append(self)
            self.m.has_grad = self.b.has_grad = True
            return
        # The angle MLP (a0 -> a1 -> a2, appearance_flow_model.py:101-103): nothing on the main stream reads the input gradient of a
        # small fc layer whose input comes from another small fc layer, so its data gradient rides the filter-gradient stream too
        # (two 5 us launches off the dependent chain of the reverse pass)
        side_dgrad = (x.requires_grad and fin <= 128 and fout <= 128 and x.grad_consumers <= 1 and
                      any(isinstance(n, LinearNode) and n.y.storage is x.storage and n.y.ch_off == x.ch_off and n.x.C <= 128
                          for n in g.nodes))
        ws_side = g.begin_side(25.0, 32.0 if x.requires_grad else 0.0)
        if side_dgrad:
            epi = _epi(mask_of=x)       # both gradients read dy: one launch (mv3d_fc_wgrad_dgrad)
            g.lib.fc_wgrad_dgrad(B, fin, fout, x.ptr, x.ld, y.grad_ptr, y.ld, self.m.ptr, self.m.grad_ptr, self.b.grad_ptr,
                                 x.grad_ptr, x.ld, C.byref(epi), ws_side, g.ws_bytes, g.stream)
            _note_grad_written(x, x.act != ACT_NONE)
        else:
            g.lib.fc_wgrad(B, fin, fout, x.ptr, x.ld, y.grad_ptr, y.ld, self.m.grad_ptr, self.b.grad_ptr,
                           ws_side, g.ws_bytes, g.stream)
        g.end_side()
        self.m.has_grad = self.b.has_grad = True
        if x.requires_grad and not side_dgrad:
            epi = _epi(mask_of=x)
            g.lib.fc_dgrad(B, fin, fout, y.grad_ptr, y.ld, self.m.ptr, x.grad_ptr, x.ld, C.byref(epi),
                           g.ws_ptr, g.ws_bytes, g.stream)
            _note_grad_written(x, x.act != ACT_NONE)


class ActNode(Node):
    """Stand-alone activation (only when it could not be fused into its producer)."""

    def __init__(self, x, y, act, leak):
        self.x, self.y, self.act, self.leak = x, y, act, leak

    def forward(self, g):
        x, y = self.x, self.y
        g.lib.act_fwd(x.rows, x.C, x.ptr, x.ld, y.ptr, y.ld, self.act, self.leak, g.stream)

    def backward(self, g):
        x, y = self.x, self.y
        if not y.grad_written or not x.requires_grad:
            return
        _ensure_premasked(g, y)          # y.grad now holds dL/dx
        g.lib.copy2d(x.rows, x.C, y.grad_ptr, y.ld, 1, x.grad_ptr, x.ld, 0, g.stream)
        _note_grad_written(x, False)


class ViewNode(Node):
    """tf.reshape / tf.split / zero-copy tf.concat: no kernels; only gradient bookkeeping."""

    def __init__(self, ins, outs):
        self.ins, self.outs = ins, outs

    def forward(self, g):
        pass

    def backward(self, g):
        written = [o for o in self.outs if o.grad_written]
        if not written:
            return
        if len(written) != len(self.outs):
            # a slice nobody differentiated through: its gradient is zero
            for o in self.outs:
                if not o.grad_written and o.requires_grad:
                    g.lib.copy2d(o.rows, o.C, g.zero_ptr, 0, max(o.rows, 1) * 4, o.grad_ptr, o.ld, 0, g.stream)
                    _note_grad_written(o, o.act != ACT_NONE)
        masked = [o.grad_masked for o in self.outs if o.requires_grad]
        want_mask = any(masked)
        if want_mask and not all(masked):
            for o in self.outs:
                if o.requires_grad:
                    _ensure_premasked(g, o)
        for i in self.ins:
            if i.requires_grad:
                if want_mask and i.act == ACT_NONE:
                    raise RuntimeError("view of mixed activation / linear tensors cannot carry a masked gradient")
                _note_grad_written(i, want_mask)


class CopyConcatNode(Node):
    """tf.concat fallback when an input cannot be adopted as a slice (copies)."""

    def __init__(self, ins, out):
        self.ins, self.out = ins, out

    def forward(self, g):
        off = 0
        for t in self.ins:
            g.lib.copy2d(t.rows, t.C, t.ptr, t.ld, 1, self.out.ptr + 4 * off, self.out.ld, 0, g.stream)
            off += t.C

    def backward(self, g):
        o = self.out
        if not o.grad_written:
            return
        off = 0
        for t in self.ins:
            if t.requires_grad:
                g.lib.copy2d(t.rows, t.C, o.grad_ptr + 4 * off, o.ld, 1, t.grad_ptr, t.ld, 0, g.stream)
                _note_grad_written(t, o.grad_masked and t.act != ACT_NONE)
                if o.grad_masked and t.act == ACT_NONE:
                    raise RuntimeError("masked gradient reached a linear tensor through concat")
            off += t.C


class TileNode(Node):
    """tf.tile of a [B,1,1,C] code over [B,h,w,C] (multiobject_appflow.py:148-149)."""

    def __init__(self, x, y, reps):
        self.x, self.y, self.reps = x, y, reps

    def forward(self, g):
        x, y = self.x, self.y
        g.lib.copy2d(y.rows, y.C, x.ptr, x.ld, self.reps, y.ptr, y.ld, 0, g.stream)

    def backward(self, g):
        x, y = self.x, self.y
        if not y.grad_written or not x.requires_grad:
            return
        g.lib.group_sum(x.rows, self.reps, x.C, y.grad_ptr, y.ld, x.grad_ptr, x.ld, g.stream)
        _note_grad_written(x, y.grad_masked)


class ResampleNode(Node):
    """warp_pts_layer + resample_layer (tf_utils.py:35-42) fused: flow -> (warp_pts, gen)."""

    def __init__(self, src, flow, warp, gen):
        self.src, self.flow, self.warp, self.gen = src, flow, warp, gen
        self.fused_loss = None      # (weight, LossTerm) when gen feeds exactly one plain pixel loss (Graph._fuse_resample_losses)

    def forward(self, g):
        n, h, w, _ = self.flow.shape
        _, hs, ws, c = self.src.shape
        if self.fused_loss is not None:
            # sampler + loss + sampler gradient in one pass: the loss gradient d(gen) never goes to HBM
            wgt, term = self.fused_loss
            want_grad = self.flow.requires_grad
            g.lib.warp_resample_loss(n, h, w, hs, ws, c, self.src.ptr, self.flow.ptr, self.flow.ld, term.b.ptr, term.b.ld,
                                     term.kind, float(wgt), self.warp.ptr, self.gen.ptr,
                                     self.flow.grad_ptr if want_grad else None, self.flow.ld, g.loss_buf.data_ptr(), g.stream)
            if want_grad:
                _note_grad_written(self.flow, False)
            return
        g.lib.warp_resample_fwd(n, h, w, hs, ws, c, self.src.ptr, self.flow.ptr, self.flow.ld,
                                self.warp.ptr, self.gen.ptr, g.stream)

    def backward(self, g):
        if self.fused_loss is not None or not self.gen.grad_written or not self.flow.requires_grad:
            return
        n, h, w, _ = self.flow.shape
        _, hs, ws, c = self.src.shape
        g.lib.warp_resample_bwd(n, h, w, hs, ws, c, self.src.ptr, self.flow.ptr, self.flow.ld,
                                self.gen.grad_ptr, self.flow.grad_ptr, self.flow.ld, g.stream)
        _note_grad_written(self.flow, False)


# =============================================================================================== graph
class Graph:
    """Build with the tf_utils functions inside `with Graph(...)`, then finalize() + compile()."""

    ALIGN = 64          # variables start on 256-byte boundaries of the flat buffer

    def __init__(self, device=None, seed=1234):
        self.device = torch.device(device if device is not None else 'cuda')
        self.rng = np.random.default_rng(seed)
        self.nodes = []
        self.tensors = []
        self.inputs = OrderedDict()
        self.variables = OrderedDict()
        self._scope = []
        self.loss_expr = None
        self.lr = None
        self.finalized = False
        self.lib = None
        self.stream = None
        self.ws = None
        self.ws_ptr, self.ws_bytes = None, 0
        self.n_side = max(0, min(4, int(os.environ.get('MV3D_SIDE_STREAMS', '2'))))      # 0: single-stream reverse pass; class 1 conv filter gradients, class 2 fused fc optimiser
        # fused fc optimiser launches are recorded this many graph nodes after their layer's data gradient (0 = right behind it;
        # default: at the end of the reverse pass).  They stream 400 MB each: next to the other fc layers' weight streams and the
        # latency-bound 8x8 / 4x4 convolutions they only fight for HBM, next to the MFMA-bound tail of the filter-gradient chain
        # they are free (measured at B = 64: 0 -> 26.56k, 6 -> 26.65k, end -> 26.83k images/s; unfused bucketed Adam 26.03k)
        self.fcadam_delay = int(os.environ.get('MV3D_FCADAM_DELAY', '1000000'))
        self._deferred = []
        self.ws_side = []
        self.side_streams = None
        self._side_rr = 0
        self._clk_main = self._clk_side = 0.0
        self.balance_streams = os.environ.get('MV3D_BALANCE', '0') != '0'     # measured: level clocks do not pay (concurrent kernels share the CUs)
        self.adam_stream = None
        self.adam_timing = None         # list of (start, end) events per optimiser launch when a bench wants them
        self.overlap_adam = os.environ.get('MV3D_OVERLAP_ADAM', '1') != '0'
        self.fuse_fc_adam = os.environ.get('MV3D_FUSE_FC_ADAM', '1') != '0'      # single-GPU step: Adam of the fc matrices inside their filter-gradient kernels
        self._fusing = False
        # single-GPU step: the slab reductions of all conv filter gradients and the optimiser of everything that is not a fused fc
        # matrix in ONE launch at the end of the reverse pass (mv3d_grad_finalize_*) instead of one reduction per layer + Adam
        self.fuse_finalize = os.environ.get('MV3D_FUSE_FINALIZE', '1') != '0'
        self.finalize_chunk_bytes = int(float(os.environ.get('MV3D_FINALIZE_CHUNK_MB', '48')) * 1e6)
        self._finalizing = False
        self._finalized_in_plan = False
        self._part_arena = self._fin_table = None
        self._fused_vars = []
        # The fused fc optimiser (4 x 400 MB of HBM streaming) is not joined at the end of the step: it keeps running on its side
        # stream under the NEXT step's encoder, and the forward pass waits for it in front of the first launch that touches an fc
        # matrix or an fc layer's saved input (Graph._fwd_wait_idx).  Every other reader of the weights settles first (_settle()).
        self.pipeline_fc = os.environ.get('MV3D_PIPELINE_FCADAM', '1') != '0'
        # data parallel, sharded optimiser: all-gathers of buckets first read at forward launch >= pipeline_dp_min_idx are deferred
        self.pipeline_dp = os.environ.get('MV3D_PIPELINE_DP', '1') != '0'
        self.dp_join = os.environ.get('MV3D_DP_JOIN', '0') != '0'      # 1: the main stream joins the side streams at every bucket boundary (round-1 behaviour)
        self.pipeline_dp_min_idx = 8
        self.fc_after_wgrads = os.environ.get('MV3D_FC_AFTER_WGRADS', '0') != '0'      # hold the fused fc optimiser until the conv filter gradients are done
        self._fused_nodes = []
        self._fwd_wait_idx = 0
        self._fc_event = None
        self._fc_pending = False
        self._pending_idx = 0           # forward launch index the pending event is waited for in front of
        self.plan_bwd_fused = None
        self.adam_state = None
        self.plan_fwd = self.plan_bwd = None
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.beta1_power = np.float32(self.beta1)
        self.beta2_power = np.float32(self.beta2)
        self.world_size, self.dist_group = 1, None
        self.comm = None                # parallel.TorchComm / parallel.RcclComm once data parallel is enabled
        self.dp_mode = 'sharded'        # 'sharded': reduce-scatter -> Adam on 1/world of every bucket -> all-gather; 'allreduce': SUM + redundant Adam
        self.comm_stream = None
        self.bucket_elems = 16 * 1024 * 1024        # 64 MB of fp32 gradients per all-reduce bucket

    def __enter__(self):
        _current.append(self)
        return self

    def __exit__(self, *a):
        _current.pop()

    # ---------------------------------------------------------------- construction helpers
    def scope_name(self, name):
        return '/'.join(self._scope + [name])

    def variable(self, name, shape, init):
        full = self.scope_name(name)
        if full in self.variables:
            raise ValueError("variable %s already exists" % full)
        v = Variable(full, shape, init)
        v.graph = self
        self.variables[full] = v
        return v

    def placeholder(self, shape, name):
        t = Tensor(self, shape, name=name)
        t.storage.external = True
        self.inputs[name] = t
        self.tensors.append(t)
        return t

    def new_tensor(self, shape, **kw):
        t = Tensor(self, shape, **kw)
        self.tensors.append(t)
        return t

    def add(self, node):
        self.nodes.append(node)
        return node

    # ---------------------------------------------------------------- memory layout
    def finalize(self):
        if self.finalized:
            return
        self.lib = _lib.lib()
        dev = self.device
        roots = []
        for t in self.tensors:
            s = t.storage
            base = s.alias_of if s.alias_of is not None else s
            root, _ = base.resolve()
            if s.needs_grad:
                root.needs_grad = True
            if root not in roots:
                roots.append(root)
        act_bytes = 0
        for r in roots:
            r.data = torch.zeros(r.rows * r.ch, dtype=torch.float32, device=dev)
            act_bytes += r.rows * r.ch * 4
            if r.needs_grad:
                r.grad = torch.zeros(r.rows * r.ch, dtype=torch.float32, device=dev)
                act_bytes += r.rows * r.ch * 4
        self.activation_bytes = act_bytes
        # variables: flat buffer in creation order (= TF trainable_variables order)
        off = 0
        for v in self.variables.values():
            v.offset = off
            off += -(-v.size // self.ALIGN) * self.ALIGN
        self.flat_size = off
        host = np.zeros(off, dtype=np.float32)
        for v in self.variables.values():
            host[v.offset:v.offset + v.size] = v.init(self.rng, v.shape).reshape(-1)
        self.params = torch.from_numpy(host).to(dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.adam_state = torch.zeros(16, dtype=torch.float32, device=dev)     # two records (include/mv3d_hip.h MV3D_ADAM_*): [0:8] main stream, [8:16] the fused fc optimiser's stream
        self.loss_buf = torch.zeros(4, dtype=torch.float32, device=dev)
        self.zero_buf = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.zero_ptr = self.zero_buf.data_ptr()
        self.finalized = True       # pointers are valid from here on (workspace queries need them)
        # (Measured and dropped, round 2 and again round 3: spreading the LAST filter gradients of the pass -- the first layers' --
        # over all 256 CUs, or issuing them on the main stream once the data-gradient chain has ended: +-0.3 % on the step.)
        need = 0
        for n in self.nodes:
            if hasattr(n, 'workspace_bytes'):
                need = max(need, int(n.workspace_bytes(self)))
        self.ws_bytes = need
        self.ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device=dev)
        self.ws_ptr = self.ws.data_ptr()
        # one scratch buffer per side-work class (classes may run concurrently)
        self.ws_side = [torch.empty(max(need // 4, 4), dtype=torch.float32, device=dev) for _ in range(self.n_side)]

    # ---------------------------------------------------------------- plans
    def _fuse_resample_losses(self):
        """A resampler output that is consumed by exactly one unmasked, unscaled pixel loss and by no other node (the
        appearance-flow head, appearance_flow_model.py:127-130 + build_loss) is computed together with that loss and
        the flow gradient by mv3d_warp_resample_loss.  MV3D_FUSE_RESAMPLE=0 keeps the three separate launches."""
        self.fused_terms = set()
        enabled = os.environ.get('MV3D_FUSE_RESAMPLE', '1') != '0'
        for n in self.nodes:
            if not isinstance(n, ResampleNode):
                continue
            n.fused_loss = None
            if not enabled or self.loss_expr is None:
                continue
            gen = n.gen
            uses = [(w, t) for w, t in self.loss_expr.terms if t.a is gen or t.b is gen or t.mask is gen]
            if len(uses) != 1:
                continue
            w, t = uses[0]
            if t.a is not gen or t.mask is not None or t.b_scale != 1.0 or t.b.requires_grad or t.b.rows != gen.rows or t.b.C != gen.C:
                continue
            if gen.C > 4 or gen.ld != gen.C or gen.storage.has_alias or gen.storage.alias_of is not None:
                continue
            read_elsewhere = False
            for m in self.nodes:
                if m is n:
                    continue
                for v in vars(m).values():
                    vs = v if isinstance(v, (list, tuple)) else (v,)
                    if any(x is gen for x in vs):
                        read_elsewhere = True
            if read_elsewhere:
                continue
            n.fused_loss = (w, t)
            self.fused_terms.add(id(t))

    def _find_fc_chains(self):
        """Chains of small linear layers (every width <= 64) whose intermediate outputs nobody else reads: one forward launch each
        (mv3d_fc_chain_fwd).  MV3D_FC_CHAINS=0 keeps one launch per layer."""
        for n in self.nodes:
            if isinstance(n, LinearNode):
                n.chain = None
        cur = self.lib.set_diagnostics(0)          # the live mask (mv3d_set_diagnostics returns the previous one)
        self.lib.set_diagnostics(cur)
        if os.environ.get('MV3D_FC_CHAINS', '1') == '0' or (cur & 4):       # bit 4: no small-fc kernels
            return
        def tensors_of(node):
            for val in vars(node).values():
                for t in (val if isinstance(val, (list, tuple)) else (val,)):
                    if isinstance(t, Tensor):
                        yield t
        users = {}          # root storage -> nodes that touch it
        for n in self.nodes:
            for t in tensors_of(n):
                users.setdefault(id(t._root()[0]), set()).add(id(n))
        small = lambda n: isinstance(n, LinearNode) and n.x.C <= 64 and n.y.C <= 64 and n.b is not None
        nxt = {}
        for n in self.nodes:
            if not small(n):
                continue
            root = id(n.y._root()[0])
            if len(users.get(root, ())) != 2:
                continue
            for m in self.nodes:
                if m is not n and small(m) and id(m.x._root()[0]) == root and m.x.ptr == n.y.ptr and m.x.C == n.y.C and m.x.ld == n.y.ld:
                    nxt[id(n)] = m
        heads = [n for n in self.nodes if id(n) in nxt and not any(v is n for v in nxt.values())]
        for h in heads:
            chain = [h]
            while id(chain[-1]) in nxt and len(chain) < 4:
                chain.append(nxt[id(chain[-1])])
            if len(chain) >= 2:
                for n in chain:
                    n.chain = chain

    def _emit_losses(self, with_grad):
        if self.loss_expr is None:
            return
        for w, term in self.loss_expr.terms:
            if id(term) in self.fused_terms:
                continue
            a, b, m = term.a, term.b, term.mask
            if a.C != b.C or a.rows != b.rows:
                raise ValueError("loss operands of different shapes")
            grad = a.grad_ptr if (with_grad and a.requires_grad) else None
            self.lib.pixel_loss_strided(a.rows, a.C, a.ptr, a.ld, b.ptr, b.ld, term.b_scale,
                                        m.ptr if m is not None else None, m.ld if m is not None else 1,
                                        term.kind, float(w), self.loss_buf.data_ptr(), grad, a.ld, self.stream)
            if grad is not None:
                _note_grad_written(a, False)

    def compile(self, stream=None):
        """Record the forward (+loss, +loss gradient) and backward launch sequences."""
        self.finalize()
        self.stream = stream        # None = the null stream; plans take the stream at run time
        lib = self.lib
        for t in self.tensors:
            t.grad_written = t.grad_masked = False
        self._bind_prepared_filters()
        self._fuse_resample_losses()
        self._find_fc_chains()
        self.plan_fwd = lib.plan_create()
        lib.plan_begin(self.plan_fwd)
        try:
            lib.filter_cache_refresh(None)      # first launch of the step: convert every conv filter once
            # loss terms accumulate into loss_buf[0]; the first one recorded stores instead (no launch to clear the accumulator)
            if self.loss_expr is not None and self.loss_expr.terms:
                lib.loss_overwrite_next()
            else:
                lib.fill(self.loss_buf.data_ptr(), 1, 0.0, self.stream)
            self._fwd_first_op = {}
            for n in self.nodes:
                self._fwd_first_op[id(n)] = lib.plan_size(self.plan_fwd)
                n.forward(self)
            self._emit_losses(with_grad=True)
        finally:
            lib.plan_end()
        flags_after_forward = [(t.grad_written, t.grad_masked) for t in self.tensors]
        self.plan_bwd = lib.plan_create()
        lib.plan_begin(self.plan_bwd)
        self.grad_buckets = []          # [(plan op end index, lo, hi)]: grads[lo:hi] are final once ops [.., end) ran
        try:
            done = set()
            order = list(self.variables.values())
            suffix = len(order)         # variables[suffix:] are complete
            cut_hi = self.flat_size
            gate = 0
            self._clk_main = self._clk_side = 0.0
            for n in reversed(self.nodes):
                n.backward(self)
                if isinstance(n, LinearNode):
                    gate = lib.plan_size(self.plan_bwd)
                for v in (getattr(n, 'w', None), getattr(n, 'b', None), getattr(n, 'm', None)):
                    if isinstance(v, Variable):
                        done.add(v.name)
                # the completed variables form a growing suffix of the flat buffer when the backward
                # order mirrors the creation order (true for every sequential model of the reference)
                while suffix > 0 and order[suffix - 1].name in done:
                    suffix -= 1
                lo = order[suffix].offset if suffix < len(order) else self.flat_size
                # also cut where the reverse pass moves from convolutions to an fc layer: conv filters are read by the first
                # launch of the next step (filter conversion), fc matrices much later -- their buckets' all-gathers can run
                # under the next step's encoder (run_backward_overlapped), a mixed bucket could not
                boundary = False
                if isinstance(n, ConvNode) and cut_hi > lo:
                    for prev in reversed(self.nodes[:self.nodes.index(n)]):      # the next node of the reverse pass that owns parameters
                        if any(isinstance(v, Variable) for v in vars(prev).values()):
                            boundary = isinstance(prev, LinearNode)
                            break
                if cut_hi - lo >= self.bucket_elems or boundary:
                    self.grad_buckets.append((lib.plan_size(self.plan_bwd), lo, cut_hi))
                    cut_hi = lo
        finally:
            lib.plan_end()
        nbwd = lib.plan_size(self.plan_bwd)
        if cut_hi > 0 or not self.grad_buckets:
            self.grad_buckets.append((nbwd, 0, cut_hi))
        else:
            end, lo, hi = self.grad_buckets[-1]
            self.grad_buckets[-1] = (nbwd, lo, hi)
        # Extra (empty) segment boundary: the optimiser launches of the early (fc) buckets are held back until the main
        # stream is past the fc layers and the small-spatial convolutions -- kernels that stream HBM or wait on it like
        # Adam does and slow down 3-10x next to it -- and overlap the encoder's large MFMA-bound layers instead
        # (measured: 0.85-0.92 of the backward launch list beats 0.5-0.8 by ~1.5 %: the optimiser then runs next to the two
        # largest data-gradient kernels and the tail of the filter-gradient chain; 'fc' = right behind the last fc layer).
        mode = os.environ.get('MV3D_ADAM_GATE', '0.9')      # 'fc', 'none' or a fraction of the backward launch list
        if mode == 'none':
            gate = 0
        elif mode != 'fc':
            gate = int(float(mode) * nbwd)
        self.adam_gate = gate if 0 < gate < nbwd else 0
        if self.adam_gate and all(b[0] != self.adam_gate for b in self.grad_buckets):
            i = next(j for j, b in enumerate(self.grad_buckets) if b[0] > self.adam_gate)
            hi = self.grad_buckets[i][2]
            self.grad_buckets.insert(i, (self.adam_gate, hi, hi))
        # Adam runs over the prefix of the flat buffer that holds variables with a gradient path;
        # variables without one (highdim_angle.py:8-9) keep zero gradients and are never touched.
        self.n_launch_fwd = lib.plan_size(self.plan_fwd)
        self.n_launch_bwd = lib.plan_size(self.plan_bwd)
        self._bucket_first_use = [self._first_param_use(lo, hi) for _, lo, hi in self.grad_buckets]
        # Second recording of the reverse pass for the single-GPU step, with the optimiser of the large fc matrices fused into
        # their filter-gradient kernels (LinearNode.backward); the plain plan above stays for run_backward() (tests read the
        # gradients) and for the data-parallel step (the all-reduce needs them).
        self.plan_bwd_fused = None
        self._fused_vars = []
        self._fused_nodes = []
        if self.fuse_fc_adam and self.lr is not None and any(isinstance(n, LinearNode) for n in self.nodes):
            for t, (gw, gm) in zip(self.tensors, flags_after_forward):
                t.grad_written, t.grad_masked = gw, gm
            self._clk_main = self._clk_side = 0.0
            self._side_rr = 0
            plan = lib.plan_create()
            on_gpu = torch.device(self.device).type == 'cuda'
            self._finalized_in_plan = False
            if self.fuse_finalize:
                # every conv layer gets its own region for its per-slab partial filters (they live until the end of the pass)
                off = 0
                for n in self.nodes:
                    if isinstance(n, ConvNode):
                        n._part_off, n._part_bytes = off, n.wgrad_partial_bytes(self)
                        off += -(-n._part_bytes // 256) * 256
                self._part_arena = torch.empty(max(off // 4, 4), dtype=torch.float32, device=self.device)
                lib.grad_finalize_begin()
                self._finalizing = True
                self._fin_pending, self._fin_vars, self._fin_done, self._fin_tables = 0, [], set(), []
            lib.plan_begin(plan)
            self._fusing = True
            self._deferred = []
            try:
                for n in reversed(self.nodes):
                    n.backward(self)
                    if self._finalizing and isinstance(n, ConvNode) and n.w.has_grad:
                        # the reduction (+ optimiser) of what has piled up goes out as soon as it is worth a launch: it then runs
                        # beside the rest of the pass instead of in its tail
                        self._fin_pending += n._part_bytes
                        self._fin_vars += [n.w] + ([n.b] if n.b is not None else [])
                        if self._fin_pending >= self.finalize_chunk_bytes:
                            self._finalize_commit()
                            lib.grad_finalize_begin()
                    for d in self._deferred:
                        d[0] -= 1
                    for d in [d for d in self._deferred if d[0] < 0]:
                        d[1]()
                        self._deferred.remove(d)
                for d in self._deferred:
                    d[1]()
                self._deferred = []
                if self._finalizing and self._fused_vars:
                    # everything else the optimiser owns: gradients that are already final in the flat buffer (the angle MLP, conv
                    # layers whose filter gradient is a single slab); the fused fc matrices and their biases are updated on the
                    # fused kernels' own stream (run_backward_fused).  Ranges a slab segment produces are dropped by the library.
                    fused = {id(v) for v in self._fused_vars} | {id(n.b) for n in self._fused_nodes}
                    for v in self.variables.values():
                        if v.has_grad and id(v) not in fused and id(v) not in self._fin_done:
                            lib.grad_finalize_add(v.grad_ptr, -(-v.size // 4) * 4)
                    self._finalize_commit()
                    self._finalizing = False
                    self._finalized_in_plan = True
            finally:
                self._fusing = False
                if self._finalizing:
                    lib.grad_finalize_abort()
                    self._finalizing = False
                lib.plan_end()
            if self._fused_vars:
                self.plan_bwd_fused = plan
                self.n_launch_bwd_fused = lib.plan_size(plan)
                # ranges of the flat buffers the main stream's optimiser launch leaves alone: the fused matrices, and their layers'
                # biases (whose gradients the fused kernels produce: their Adam runs behind those kernels, on their stream)
                def merged(ranges):
                    out = []
                    for lo, hi in sorted(ranges):
                        if out and lo <= out[-1][1]:
                            out[-1][1] = max(out[-1][1], hi)
                        else:
                            out.append([lo, hi])
                    return out
                pad4 = lambda v: (v.offset, v.offset + -(-v.size // 4) * 4)
                biases = merged(pad4(n.b) for n in self._fused_nodes)
                skips = merged([pad4(v) for v in self._fused_vars] + [tuple(b) for b in biases])
                assert len(skips) <= 8, "mv3d_adam_step_dev leaves at most 8 ranges untouched"
                self._skip_lo = (C.c_int64 * len(skips))(*[a for a, _ in skips])
                self._skip_hi = (C.c_int64 * len(skips))(*[b for _, b in skips])
                # the biases: ONE launch over [first bias, last bias end) that skips what lies between them
                self._bias_span = (biases[0][0], biases[-1][1])
                gaps = [(biases[i][1] - biases[0][0], biases[i + 1][0] - biases[0][0]) for i in range(len(biases) - 1)]
                assert len(gaps) <= 8
                self._bias_skip = (len(gaps), (C.c_int64 * max(1, len(gaps)))(*[a for a, _ in gaps] or [0]),
                                   (C.c_int64 * max(1, len(gaps)))(*[b for _, b in gaps] or [0]))
                self._fwd_wait_idx = self._first_fc_hazard()
            else:
                lib.plan_destroy(plan)
        self.upload_adam_state()
        return self

    def _finalize_commit(self):
        """Close the open mv3d_grad_finalize collection: one launch on the filter-gradient stream that sums the collected layers'
        slabs and applies their optimiser update (recorded into the plan being recorded)."""
        lib = self.lib
        on_gpu = torch.device(self.device).type == 'cuda'
        tb = int(lib.grad_finalize_table_bytes())
        table = torch.empty(max(tb, 16), dtype=torch.uint8, device=self.device) if on_gpu else None
        self._fin_tables.append(table)
        self.begin_side(20.0, 0.0)
        try:
            lib.grad_finalize_commit(table.data_ptr() if on_gpu else None, tb, self.grads.data_ptr(), self.params.data_ptr(),
                                     self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.adam_state.data_ptr(), self.stream)
        finally:
            self.end_side()
        self._fin_done |= {id(v) for v in self._fin_vars}
        self._fin_pending, self._fin_vars = 0, []

    def _bind_prepared_filters(self):
        """Weights only change in apply_adam(), so each conv filter is converted to the kernels' operand format
        once per step (mv3d_filter_cache_*, include/mv3d_hip.h) instead of once per call.  The library's cache is
        process-global: it is reset here, the plans recorded below keep their own copy of the job table."""
        lib = self.lib
        lib.filter_cache_clear()
        self._prepared = []
        if torch.device(self.device).type != 'cuda':      # plan recording without a GPU (host-logic tests): nothing to bind
            return
        for n in self.nodes:
            if not isinstance(n, ConvNode):
                continue
            geom = n.geom()
            ops = [2 if n.transposed else 0]
            if n.x.requires_grad:
                ops.append(3 if n.transposed else 1)
            for op in ops:
                nbytes = int(lib.filter_prepared_bytes(C.byref(geom), op))
                if nbytes > 0:
                    buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                    self._prepared.append(buf)
                    lib.filter_cache_bind(C.byref(geom), op, n.w.ptr, buf.data_ptr(), nbytes)
        tb = int(lib.filter_cache_table_bytes())
        self._prepared_table = torch.empty(max(tb, 16), dtype=torch.uint8, device=self.device)
        lib.filter_cache_commit(self._prepared_table.data_ptr(), tb, None)

    # ---------------------------------------------------------------- execution
    def _stream_ptr(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _first_param_use(self, lo, hi):
        """Index of the first forward launch that reads a parameter of the flat range [lo, hi): a conv filter is read by launch 0
        (the conversion of all filters, mv3d_filter_cache_refresh), everything else by its layer's first launch."""
        first = self.n_launch_fwd
        for n in self.nodes:
            for name, val in vars(n).items():
                if isinstance(val, Variable) and val.offset is not None and lo <= val.offset < hi:
                    first = min(first, 0 if (isinstance(n, ConvNode) and name == 'w') else self._fwd_first_op[id(n)])
        return first

    def _first_fc_hazard(self):
        """Index of the first forward launch that must not run before the previous step's fused fc optimiser finished: the
        first node that reads one of those matrices or touches the storage of a fused layer's saved input (the deferred
        kernels still read it).  None (no pipelining) when such an input is fed from outside the graph."""
        roots = set()
        for n in self._fused_nodes:
            root = n.x._root()[0]
            if root.external or n.x.storage.external:
                return None
            roots.add(id(root))
        def tensors_of(node):
            for val in vars(node).values():
                for t in (val if isinstance(val, (list, tuple)) else (val,)):
                    if isinstance(t, Tensor):
                        yield t
        fused = {id(n) for n in self._fused_nodes}
        idx = self.n_launch_fwd
        for n in self.nodes:
            if id(n) in fused or any(id(t._root()[0]) in roots for t in tensors_of(n)):
                idx = min(idx, self._fwd_first_op[id(n)])
        return idx

    def _settle(self):
        """Order the current stream behind a fused fc optimiser still in flight from the last train step."""
        if self._fc_pending:
            torch.cuda.current_stream(self.device).wait_event(self._fc_event)
            self._fc_pending = False

    settle = _settle        # public name: call before touching Graph.params / adam_m / adam_v / grads directly

    def run_forward(self):
        st = self._stream_ptr()
        if self._fc_pending and 0 < self._pending_idx < self.n_launch_fwd:
            self.lib.plan_run_range(self.plan_fwd, 0, self._pending_idx, st)
            self._settle()
            self.lib.plan_run_range(self.plan_fwd, self._pending_idx, self.n_launch_fwd, st)
            return
        self._settle()
        self.lib.plan_run(self.plan_fwd, st)

    def begin_side(self, cost_side=0.0, cost_main=0.0, cls=1):
        """Tag the calls recorded until end_side() as side work of class `cls` (1: conv / fc filter gradients, 2: the fused fc
        optimiser; a class maps to side stream (cls - 1) % n_side); returns the scratch pointer reserved for that class.
        cost_side / cost_main: estimated microseconds of the side work and of the main-stream work recorded next (only
        used by the optional list schedule MV3D_BALANCE)."""
        fork = self._clk_main
        if self.n_side == 0 or (self.balance_streams and self._clk_side > self._clk_main + cost_main):
            self._clk_main += cost_side + cost_main
            return self.ws_ptr
        self._clk_side = max(self._clk_side, fork) + cost_side
        self._clk_main += cost_main
        k = (cls - 1) % self.n_side
        self.lib.plan_side(k + 1)
        return self.ws_side[k].data_ptr()

    def end_side(self):
        if self.n_side:
            self.lib.plan_side(0)

    def _side_ptrs(self):
        """HIP streams for the filter-gradient kernels of the reverse pass (empty on CPU / when disabled)."""
        if self.n_side == 0 or torch.device(self.device).type != 'cuda':
            return None, 0
        if self.side_streams is None:
            # MV3D_SIDE_PRIORITY: comma-separated stream priorities of the side streams (0 = default, -1 = high); measured: no gain
            prio = [int(x) for x in os.environ.get('MV3D_SIDE_PRIORITY', '').split(',') if x.strip()]
            self.side_streams = [torch.cuda.Stream(device=self.device, priority=(prio[k] if k < len(prio) else 0)) for k in range(self.n_side)]
            self._side_arr = (C.c_void_p * self.n_side)(*[st.cuda_stream for st in self.side_streams])
        return self._side_arr, self.n_side

    def run_backward(self):
        self._settle()
        sides, ns = self._side_ptrs()
        self.lib.plan_run_range_multi(self.plan_bwd, 0, self.n_launch_bwd, self._stream_ptr(), sides, ns, 0)

    def allreduce_grads(self):
        self._settle()
        if self.world_size > 1:
            self.comm.allreduce_sum_(self.grads, 0, self.flat_size, self._stream_ptr())

    def upload_adam_state(self):
        """lr, betas, epsilon, the two beta powers and the gradient scale (1 / world size) to the device Adam state."""
        if self.adam_state is None or self.lr is None:
            return
        vals = np.array([self.lr, self.beta1, self.beta2, self.eps, self.beta1_power, self.beta2_power, 1.0 / self.world_size, 0.0], np.float32)
        self._settle()
        self.adam_state.copy_(torch.from_numpy(np.concatenate([vals, vals])))

    def _adam_range(self, lo, hi, stream, skips=None, state=None):
        off = lo * 4
        n, slo, shi = (0, None, None) if skips is None else skips
        self.lib.adam_step_dev(hi - lo, self.params.data_ptr() + off, self.grads.data_ptr() + off, self.adam_m.data_ptr() + off,
                               self.adam_v.data_ptr() + off, self.adam_state.data_ptr() if state is None else state, n, slo, shi, stream)

    def _adam_advance(self, stream=None, both=True):
        """beta powers *= betas: on the device (behind every optimiser launch of this step) and in the host mirror the
        checkpoints read"""
        st = self._stream_ptr() if stream is None else stream
        self.lib.adam_advance(self.adam_state.data_ptr(), st)
        if both:                                             # the fused fc optimiser's record (advanced on its own stream by run_backward_fused)
            self.lib.adam_advance(self.adam_state.data_ptr() + 32, st)
        self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))

    def apply_adam(self):
        self._settle()
        self._adam_range(0, self.flat_size, self._stream_ptr())
        self._adam_advance()

    def run_backward_fused(self):
        """Single-GPU reverse pass with the fc matrices' optimiser inside their filter-gradient kernels; one small launch
        updates everything else (conv filters, biases, the angle MLP: 3 % of the parameters) once the side streams joined."""
        sides, ns = self._side_ptrs()
        st = self._stream_ptr()
        pipelined = self.pipeline_fc and ns >= 2 and self._fwd_wait_idx is not None
        self.lib.plan_run_range_multi(self.plan_bwd_fused, 0, self.n_launch_bwd_fused, st, sides, ns,
                                      (1 if pipelined else 0) | (2 if pipelined and self.fc_after_wgrads else 0))
        fc_state = self.adam_state.data_ptr() + 32
        if pipelined and self.fc_after_wgrads:
            # flags bit 2 made the plan hold the class-2 launches back: issue them now, behind the conv filter gradients
            for k, q in enumerate(self.side_streams):
                if k != 1 % ns:
                    self.side_streams[1 % ns].wait_stream(q)
            self.lib.plan_run_side(self.plan_bwd_fused, 2, self.side_streams[1 % ns].cuda_stream)
        if pipelined:
            main = torch.cuda.current_stream(self.device)
            for k, q in enumerate(self.side_streams):
                if k != 1 % ns:
                    main.wait_stream(q)                      # the conv filter gradients: the launch below reads them
            fcq = self.side_streams[1 % ns]                  # class 2: the fused fc optimiser
            fc_stream = fcq.cuda_stream
        else:
            fc_stream = st
        lo, hi = self._bias_span
        self._adam_range(lo, hi, fc_stream, self._bias_skip, state=fc_state)
        self.lib.adam_advance(fc_state, fc_stream)
        if pipelined:
            if self._fc_event is None:
                self._fc_event = torch.cuda.Event()
            self._fc_event.record(fcq)
            self._fc_pending = True
            self._pending_idx = self._fwd_wait_idx
        if not self._finalized_in_plan:          # otherwise the plan's last launch (grad_finalize_adam) was the optimiser of everything else
            self._adam_range(0, self.flat_size, st, (len(self._skip_lo), self._skip_lo, self._skip_hi))
        self._adam_advance(st, both=False)

    def run_backward_with_adam(self):
        """Single-GPU reverse pass with the optimiser folded in: the backward plan is issued bucket by bucket
        (the same >= 64 MB suffix buckets the data-parallel path all-reduces); as soon as a bucket's gradients are
        final -- its filter-gradient kernels sit on the side streams -- Adam for that slice of the flat buffers
        starts on its own stream while the main stream continues with the data gradients of the layers below.
        Adam is pure HBM streaming (28 B per parameter), the convolution kernels it overlaps are MFMA / latency bound."""
        self._settle()
        main = torch.cuda.current_stream(self.device)
        sides, ns = self._side_ptrs()
        if self.adam_stream is None:
            self.adam_stream = torch.cuda.Stream(device=self.device)
        begin, pending = 0, []
        for end, lo, hi in self.grad_buckets:
            self.lib.plan_run_range_multi(self.plan_bwd, begin, end, main.cuda_stream, sides, ns, 1)      # no join
            begin = end
            if hi > lo:
                pending.append((lo, hi))
            if end < self.adam_gate or not pending:
                continue
            # the slices' gradients come from side-stream kernels; their weights were last read by main-stream kernels
            self.adam_stream.wait_stream(main)
            for st in (self.side_streams or []):
                self.adam_stream.wait_stream(st)
            for plo, phi in pending:
                if self.adam_timing is not None:      # bench: HIP events around the optimiser launches, on their stream
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.adam_stream)
                self._adam_range(plo, phi, self.adam_stream.cuda_stream)
                if self.adam_timing is not None:
                    e1.record(self.adam_stream)
                    self.adam_timing.append((e0, e1))
            pending = []
        main.wait_stream(self.adam_stream)
        for st in (self.side_streams or []):
            main.wait_stream(st)
        self._adam_advance(main.cuda_stream)

    def run_backward_overlapped(self, with_adam=False):
        """Data-parallel reverse pass.  The recorded backward sequence is issued in segments; after each segment the gradients
        it completed -- a contiguous suffix range of the flat buffer, >= 64 MB: the fc matrices, 97 % of the bytes, are final
        after the decoder's half of the reverse pass -- go to the communicator on its own stream while the next segment computes:
          'sharded'    reduce-scatter(SUM) of the bucket -> TF-Adam on this rank's 1/world slice of it (grad scale 1/world) ->
                       all-gather of the updated parameters, all on the communication stream.  The optimiser's 28 B/param are
                       paid once per node instead of once per GPU, and the all-gather of a bucket only overwrites weights no
                       later kernel of this step reads (a layer's data gradient precedes its bucket).
          'allreduce'  SUM all-reduce of the bucket, then Adam on all of it on every rank (identical weights by construction).
        Both leave bit-identical weights on every rank (tests/test_dist_cpu.py)."""
        self._settle()
        main = self._stream_ptr()
        sides, ns = self._side_ptrs()
        on_gpu = torch.device(self.device).type == 'cuda'
        if on_gpu and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)
        cs = self.comm_stream.cuda_stream if on_gpu else None
        comm, W = self.comm, self.world_size
        begin = 0
        late = []                       # [(first forward launch that reads the bucket, lo, slice length)]
        for end, lo, hi in self.grad_buckets:
            # the segment's filter-gradient launches stay on their side streams: only the COMMUNICATION stream waits for them (the
            # main stream goes straight on with the next layers' data gradients instead of idling behind an 80 us fc filter gradient
            # at every bucket boundary)
            self.lib.plan_run_range_multi(self.plan_bwd, begin, end, main, sides, ns, 1 if (on_gpu and not self.dp_join) else 0)
            begin = end
            if hi <= lo:
                continue
            if on_gpu:
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                for q in (self.side_streams or []):
                    self.comm_stream.wait_stream(q)
                ctx = torch.cuda.stream(self.comm_stream)      # a torch.distributed communicator takes the current stream
                ctx.__enter__()
            if self.dp_mode == 'sharded' and with_adam:
                n = (hi - lo) // W
                assert n * W == hi - lo and n % 4 == 0, "bucket not divisible by 4 * world"
                comm.reduce_scatter_sum_(self.grads, lo, n, cs)
                a = lo + comm.rank * n
                self._adam_range(a, a + n, cs)
                self._slots_sharded = True      # this rank's Adam slots are now current on its own slices only
                # the updated slices of a bucket whose parameters the next forward pass reads late (the fc matrices: 97 % of the
                # bytes) are gathered AFTER every bucket has been reduced and, on the GPU, under the next step's encoder
                use = self._bucket_first_use[self.grad_buckets.index((end, lo, hi))]
                if self.pipeline_dp and use >= self.pipeline_dp_min_idx:
                    late.append((use, lo, n))
                else:
                    comm.allgather_(self.params, lo, n, cs)
            else:
                comm.allreduce_sum_(self.grads, lo, hi - lo, cs)
                if with_adam:
                    self._adam_range(lo, hi, cs)
            if on_gpu:
                ctx.__exit__(None, None, None)
        if on_gpu:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.comm_stream)                # every reduce-scatter, Adam slice and early all-gather
            for q in (self.side_streams or []):              # (and any filter gradient behind the last bucket's boundary)
                cur.wait_stream(q)
        if late:
            late.sort()
            if on_gpu:
                ctx = torch.cuda.stream(self.comm_stream)
                ctx.__enter__()
            for _, lo, n in late:
                comm.allgather_(self.params, lo, n, cs)
            if on_gpu:
                ctx.__exit__(None, None, None)
                if self._fc_event is None:
                    self._fc_event = torch.cuda.Event()
                self._fc_event.record(self.comm_stream)
                self._fc_pending = True
                self._pending_idx = late[0][0]
        if with_adam:
            self._adam_advance()

    def train_step(self):
        """forward + loss + reverse pass + (all-reduce) + Adam; returns the device loss scalar."""
        self.run_forward()
        if self.world_size > 1:
            self.run_backward_overlapped(with_adam=True)
        elif self.plan_bwd_fused is not None:
            self.run_backward_fused()
        elif self.overlap_adam and torch.device(self.device).type == 'cuda':
            self.run_backward_with_adam()
        else:
            self.run_backward()
            self.apply_adam()
        return self.loss_buf[0]

    # ---------------------------------------------------------------- variables I/O
    def get_variables(self):
        self._settle()
        return OrderedDict((k, v.value().detach().cpu().numpy().copy()) for k, v in self.variables.items())

    def set_variables(self, values):
        self._settle()
        for k, a in values.items():
            self.variables[k].value().copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)).reshape(self.variables[k].shape))

    def get_gradients(self):
        self._settle()
        return OrderedDict((k, v.grad_value().detach().cpu().numpy().copy()) for k, v in self.variables.items() if v.has_grad)

    def gather_optimizer_state(self):
        """COLLECTIVE (every rank calls it): after sharded data-parallel steps a rank's Adam slots are current only on its own
        1/world slice of every bucket; all-gather them (same lo / n layout as the parameters) so that any rank can write a
        complete checkpoint.  No-op on one GPU and in 'allreduce' mode."""
        self._settle()
        if self.world_size <= 1 or not getattr(self, '_slots_sharded', False):
            return
        on_gpu = torch.device(self.device).type == 'cuda'
        if on_gpu:
            torch.cuda.synchronize(self.device)
        cs = self._stream_ptr() if on_gpu else None
        W = self.world_size
        for _, lo, hi in self.grad_buckets:
            if hi > lo:
                n = (hi - lo) // W
                self.comm.allgather_(self.adam_m, lo, n, cs)
                self.comm.allgather_(self.adam_v, lo, n, cs)
        if on_gpu:
            torch.cuda.synchronize(self.device)
        self._slots_sharded = False

    def state_dict(self):
        """TF-Saver-style names: <var>, <var>/Adam, <var>/Adam_1, beta1_power, beta2_power
        (train.py:70-71 saves GLOBAL_VARIABLES)."""
        self._settle()
        if self.world_size > 1 and getattr(self, '_slots_sharded', False):
            raise RuntimeError("sharded data-parallel step: this rank holds 1/%d of the Adam slots; call "
                               "Graph.gather_optimizer_state() on EVERY rank before state_dict() / Saver.save()" % self.world_size)
        sd = OrderedDict()
        for k, v in self.variables.items():
            sd[k] = v.value().detach().cpu().clone()
            if v.has_grad:
                sd[k + '/Adam'] = self.adam_m[v.offset:v.offset + v.size].view(v.shape).detach().cpu().clone()
                sd[k + '/Adam_1'] = self.adam_v[v.offset:v.offset + v.size].view(v.shape).detach().cpu().clone()
        sd['beta1_power'] = torch.tensor(float(self.beta1_power))
        sd['beta2_power'] = torch.tensor(float(self.beta2_power))
        return sd

    def load_state_dict(self, sd):
        self._settle()
        missing = [k for k in self.variables if k not in sd]
        slots = {'beta1_power', 'beta2_power'}
        unexpected = [k for k in sd if k not in slots and k not in self.variables and
                      not (k.rsplit('/', 1)[0] in self.variables and k.rsplit('/', 1)[-1] in ('Adam', 'Adam_1'))]
        bad = [k for k, v in self.variables.items() if k in sd and tuple(sd[k].shape) != tuple(v.shape)]
        if missing or unexpected or bad or not slots <= set(sd):
            raise KeyError("checkpoint does not match the model: missing %s; unexpected %s; shape mismatch %s%s"
                           % (missing[:8], unexpected[:8], [(k, tuple(sd[k].shape), self.variables[k].shape) for k in bad[:8]],
                              '' if slots <= set(sd) else '; no beta1_power / beta2_power'))
        for k, v in self.variables.items():
            v.value().copy_(sd[k])
            if k + '/Adam' in sd:
                self.adam_m[v.offset:v.offset + v.size].view(v.shape).copy_(sd[k + '/Adam'])
                self.adam_v[v.offset:v.offset + v.size].view(v.shape).copy_(sd[k + '/Adam_1'])
        self.beta1_power = np.float32(float(sd['beta1_power']))
        self.beta2_power = np.float32(float(sd['beta2_power']))
        self.upload_adam_state()


# =============================================================================================== initialisers
def truncated_normal_init(stddev):
    """tf.truncated_normal_initializer(stddev) (tf_utils.py:77): redraw beyond 2 sigma."""
    def init(rng, shape):
        out = rng.standard_normal(shape)
        bad = np.abs(out) > 2
        while bad.any():
            out[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(out) > 2
        return (out * stddev).astype(np.float32)
    return init


def random_normal_init(stddev):
    """tf.random_normal_initializer(stddev) (tf_utils.py:63,95)."""
    return lambda rng, shape: (rng.standard_normal(shape) * stddev).astype(np.float32)


def zeros_init():
    """tf.constant_initializer(0.) (tf_utils.py:65,80)."""
    return lambda rng, shape: np.zeros(shape, np.float32)
