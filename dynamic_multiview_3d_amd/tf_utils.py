"""Host-side mirror of the reference op library dyn_mult_view/mv3d/utils/tf_utils.py:18-98.

Same function names, argument order and variable naming ('<scope>/w', '<scope>/b',
'<scope>/Matrix'); instead of TensorFlow graph nodes the functions append launch nodes of
libmv3d_hip.so to the current Graph (graph.py).  Activations that directly follow a
conv / deconv / linear are fused into that kernel's epilogue.
"""
import math

from .graph import (current_graph, Tensor, Storage, ScalarExpr, LossTerm, ConvNode, LinearNode, ActNode, ViewNode,
                    CopyConcatNode, TileNode, ResampleNode, truncated_normal_init, random_normal_init, zeros_init)
from ._lib import ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH


class variable_scope:
    """tf.variable_scope(name): prefixes variable names (main_model.py:58,69)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        current_graph()._scope.append(self.name)
        return self

    def __exit__(self, *a):
        current_graph()._scope.pop()


def _check_usable(t):
    if t.fused_into is not None:
        raise RuntimeError("pre-activation tensor was fused into its producer's epilogue; use the activation output")


def _same_out(size, s):
    return -(-size // s)


# ------------------------------------------------------------------------------------------------ losses
class _Scaled:
    """x * c for a constant c: a loss target only (mv3d/bg_nodm.py:88 `gt_sm = gt_sm * 0.75`); folded into the loss kernel."""

    def __init__(self, x, c):
        self.x, self.c = x, float(c)


class _Masked:
    """tf.multiply(x, mask) with a one-channel mask: a loss operand only (mv3d/bg_nodm.py:91); (a*m - b*m) is evaluated
    as (a - b)*m inside the loss kernel."""

    def __init__(self, x, mask):
        self.x, self.mask = x, mask


def scale(x, c):
    return _Scaled(x, c)


def multiply(x, mask):
    if mask.C != 1:
        raise NotImplementedError("multiply: only a one-channel mask is supported")
    return _Masked(x, mask)


def _loss_term(input1, input2, kind, mask=None):
    for v in (input1, input2):
        if isinstance(v, _Masked):
            if mask is not None and mask is not v.mask:
                raise NotImplementedError("loss operands multiplied by different masks")
            mask = v.mask
    a, b = (v.x if isinstance(v, _Masked) else v for v in (input1, input2))
    # the differentiated operand goes first; both reference losses are symmetric in their arguments
    a_grad = (a.x if isinstance(a, _Scaled) else a).requires_grad
    b_grad = (b.x if isinstance(b, _Scaled) else b).requires_grad
    if b_grad and not a_grad:
        a, b = b, a
    elif a_grad and b_grad:
        raise NotImplementedError("loss between two differentiated tensors")
    if isinstance(a, _Scaled):
        raise NotImplementedError("scaling the differentiated operand of a loss")
    b_scale = 1.0
    if isinstance(b, _Scaled):
        b, b_scale = b.x, b.c
    return ScalarExpr([(1.0, LossTerm(a, b, kind, mask, b_scale))])


def euclidean_loss(input1, input2):
    """tf_utils.py:18-19: reduce_mean(reduce_sum(pow(a-b, 2), 3))."""
    return _loss_term(input1, input2, 2)


def l1_loss(input1, input2):
    """tf_utils.py:22-23: reduce_mean(reduce_sum(abs(a-b), 3))."""
    return _loss_term(input1, input2, 1)


def masked_euclidean_loss(input1, input2, mask):
    """reduce_mean(reduce_sum(pow((a-b)*mask, 2), 3)) -- the inline expression of
    multi_view_model/multiobject_appflow.py:239-242 (mask is [B,H,W,1])."""
    return _loss_term(input1, input2, 2, mask)


# ------------------------------------------------------------------------------------------------ activations
def _activation(x, act, leak=0.2):
    g = current_graph()
    _check_usable(x)
    p = x.producer
    if isinstance(p, (ConvNode, LinearNode)) and p.y is x and p.act == ACT_NONE and x.grad_consumers == 0:
        p.act, p.leak = act, leak
        y = g.new_tensor(x.shape, storage=x.storage, ch_off=x.ch_off, producer=p, act=act, leak=leak,
                         requires_grad=x.requires_grad)
        p.y = y
        x.fused_into = p
        return y
    y = g.new_tensor(x.shape, act=act, leak=leak, requires_grad=x.requires_grad)
    x.grad_consumers += 1
    y.producer = g.add(ActNode(x, y, act, leak))
    return y


def relu(x, name="relu"):
    """tf_utils.py:25-27: 0.5*x + 0.5*abs(x)."""
    return _activation(x, ACT_RELU)


def lrelu(x, leak=0.2, name="lrelu"):
    """tf_utils.py:29-33: f1*x + f2*abs(x), f1 = 0.5(1+leak), f2 = 0.5(1-leak)."""
    return _activation(x, ACT_LRELU, leak)


def tanh(x):
    """tf.nn.tanh (main_model.py:79)."""
    return _activation(x, ACT_TANH)


# ------------------------------------------------------------------------------------------------ warp / resample
def coords(h, w, batch_size):
    """tf_utils.py:44-52.  The grid is generated inside the resampler kernels: channel 0 = row
    index, channel 1 = column index; returned here as a description."""
    return ('coords', int(h), int(w), int(batch_size))


class _WarpPts(Tensor):
    pass


def warp_pts_layer(flow_field, name="warp_pts"):
    """tf_utils.py:35-38: flow_field + coords(...).  Materialised by resample_layer's kernel."""
    g = current_graph()
    _check_usable(flow_field)
    w = _WarpPts(g, flow_field.shape, requires_grad=False, name=name)
    w.flow = flow_field
    g.tensors.append(w)
    return w


def resample_layer(src_img, warp_pts, name="tgt_img"):
    """tf_utils.py:40-42: tf.contrib.resampler.resampler(src_img, warp_pts)."""
    g = current_graph()
    if not isinstance(warp_pts, _WarpPts):
        raise NotImplementedError("resample_layer expects the output of warp_pts_layer")
    flow = warp_pts.flow
    if src_img.requires_grad:
        raise NotImplementedError("gradient w.r.t. the resampled image (every reference model warps an input image)")
    if src_img.ld != src_img.C:
        raise NotImplementedError("resampling a channel-sliced source")
    n, h, w, _ = flow.shape
    gen = g.new_tensor((n, h, w, src_img.C), requires_grad=flow.requires_grad, name=name)
    flow.grad_consumers += 1
    gen.producer = g.add(ResampleNode(src_img, flow, warp_pts, gen))
    return gen


# ------------------------------------------------------------------------------------------------ layers
def linear_msra(input_, output_size, name):
    """tf_utils.py:54-67: Matrix ~ N(0, sqrt(2/fan_in)), b = 0; matmul + b."""
    g = current_graph()
    _check_usable(input_)
    fan_in = int(input_.get_shape()[-1])
    stddev = 1.0 * math.sqrt(2. / float(fan_in))
    with variable_scope(name):
        matrix = g.variable("Matrix", [fan_in, output_size], random_normal_init(stddev))
        b = g.variable("b", [output_size], zeros_init())
    y = g.new_tensor((input_.shape[0], output_size), requires_grad=True)
    input_.grad_consumers += 1
    y.producer = g.add(LinearNode(input_, y, matrix, b))
    return y


def conv2d_msra(input_, output_dim, k_h, k_w, d_h, d_w, name):
    """tf_utils.py:70-84: w ~ truncated N(0, sqrt(2/(k_h*k_w*Cin))), b = 0; SAME conv + b."""
    g = current_graph()
    _check_usable(input_)
    n, h, w, cin = input_.shape
    stddev = 1.0 * math.sqrt(2. / float(k_h * k_w * cin))
    with variable_scope(name):
        wv = g.variable('w', [k_h, k_w, cin, output_dim], truncated_normal_init(stddev))
        b = g.variable('b', [output_dim], zeros_init())
    y = g.new_tensor((n, _same_out(h, d_h), _same_out(w, d_w), output_dim), requires_grad=True)
    input_.grad_consumers += 1
    y.producer = g.add(ConvNode(input_, y, wv, b, k_h, k_w, d_h, d_w, transposed=False))
    return y


def deconv2d_msra(input_, output_shape, k_h, k_w, d_h, d_w, name):
    """tf_utils.py:87-98: w[k_h,k_w,Cout,Cin] ~ N(0, sqrt(2/(k_h*k_w*Cin)*d_h*d_w)), no bias;
    conv2d_transpose with default SAME padding."""
    g = current_graph()
    _check_usable(input_)
    n, h, w, cin = input_.shape
    out = tuple(int(s) for s in output_shape)
    if out[0] != n or _same_out(out[1], d_h) != h or _same_out(out[2], d_w) != w:
        raise ValueError("output_shape %s inconsistent with input %s and stride (%d,%d)" % (out, input_.shape, d_h, d_w))
    stddev = 1.0 * math.sqrt(2.0 / float(k_h * k_w * cin) * float(d_h) * float(d_w))
    with variable_scope(name):
        wv = g.variable('w', [k_h, k_w, out[-1], cin], random_normal_init(stddev))
    y = g.new_tensor(out, requires_grad=True)
    input_.grad_consumers += 1
    y.producer = g.add(ConvNode(input_, y, wv, None, k_h, k_w, d_h, d_w, transposed=True))
    return y


# ------------------------------------------------------------------------------------------------ glue
def _common_act(ts):
    acts = {(t.act, t.leak) for t in ts}
    return acts.pop() if len(acts) == 1 else (ACT_NONE, 0.2)


def concat(values=None, axis=None):
    """tf.concat(axis=..., values=[...]) on the channel (last) axis.  Inputs that own their storage are adopted as channel
    slices of one wider buffer, so their producers write straight into it (no copy)."""
    g = current_graph()
    values = list(values)
    for t in values:
        _check_usable(t)
    nd = len(values[0].shape)
    if axis not in (nd - 1, -1):
        raise NotImplementedError("concat on a non-channel axis")
    rows = values[0].rows
    total = sum(t.C for t in values)
    shape = values[0].shape[:-1] + (total,)
    rg = any(t.requires_grad for t in values)
    act, leak = _common_act(values)
    adoptable = all(t.storage.can_be_adopted() and t.ch_off == 0 and t.C == t.storage.ch and t.rows == rows
                    for t in values) and len({id(t.storage) for t in values}) == len(values)
    if adoptable:
        st = Storage(rows, total)
        off = 0
        for t in values:
            t.storage.parent, t.storage.ch_off = st, off
            if t.storage.needs_grad:
                st.needs_grad = True
            off += t.C
        out = g.new_tensor(shape, storage=st, act=act, leak=leak, requires_grad=rg)
        out.producer = g.add(ViewNode(values, [out]))
    else:
        out = g.new_tensor(shape, act=act, leak=leak, requires_grad=rg)
        out.producer = g.add(CopyConcatNode(values, out))
    for t in values:
        t.grad_consumers += 1
    return out


def split(value, num_or_size_splits, axis):
    """tf.split into channel slices (views): a count (equal slices) or a list of sizes -- the latter also stands in for
    the tf.slice pairs of mv3d/nobg_dm.py:85-89 (colour = channels 0..2, depth / mask = channel 3)."""
    g = current_graph()
    _check_usable(value)
    if axis not in (len(value.shape) - 1, -1):
        raise NotImplementedError("split on a non-channel axis")
    if isinstance(num_or_size_splits, (list, tuple)):
        sizes = [int(c) for c in num_or_size_splits]
        if sum(sizes) != value.C or min(sizes) < 1:
            raise ValueError("split sizes %s do not add up to %d channels" % (sizes, value.C))
    else:
        num = int(num_or_size_splits)
        if num < 1 or value.C % num:
            raise ValueError("channels not divisible")
        sizes = [value.C // num] * num
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    outs = [g.new_tensor(value.shape[:-1] + (c,), storage=value.storage, ch_off=value.ch_off + o, act=value.act,
                         leak=value.leak, requires_grad=value.requires_grad) for c, o in zip(sizes, offs)]
    node = g.add(ViewNode([value], list(outs)))
    for o in outs:
        o.producer = node
    value.grad_consumers += 1
    return outs


def reshape(tensor, shape):
    """tf.reshape between [B,h,w,c] and [B,h*w*c] (NHWC flatten order h,w,c)."""
    g = current_graph()
    _check_usable(tensor)
    shape = tuple(int(s) for s in shape)
    st = tensor.storage
    dense = st.parent is None and tensor.ch_off == 0 and tensor.C == st.ch
    if not dense:
        raise NotImplementedError("reshape of a channel-sliced tensor")
    base = st.alias_of if st.alias_of is not None else st
    rows = 1
    for s in shape[:-1]:
        rows *= s
    if rows * shape[-1] != tensor.rows * tensor.C:
        raise ValueError("reshape size mismatch")
    alias = Storage(rows, shape[-1], alias_of=base)
    alias.external = base.external
    if st.needs_grad:
        alias.needs_grad = True
    out = g.new_tensor(shape, storage=alias, act=tensor.act, leak=tensor.leak, requires_grad=tensor.requires_grad)
    out.producer = g.add(ViewNode([tensor], [out]))
    tensor.grad_consumers += 1
    return out


def tile_spatial(code, h, w):
    """reshape [B,C] -> [B,1,1,C] + tf.tile over [1,h,w,1] (multiobject_appflow.py:148-149)."""
    g = current_graph()
    _check_usable(code)
    b, c = code.shape
    out = g.new_tensor((b, h, w, c), act=code.act, leak=code.leak, requires_grad=code.requires_grad)
    code.grad_consumers += 1
    out.producer = g.add(TileNode(code, out, h * w))
    return out


# ---------------------------------------------------------------------------- snapshots (mv3d/utils/tf_utils.py:199-212)
def save_snapshot(saver, session, path, step):
    """saver.save(session, path/snapshot<step>, global_step=step): files `snapshot<step>-<step>.index|.data-*`."""
    import os
    return saver.save(session, os.path.join(path, "snapshot" + str(step)), global_step=step)


def load_snapshot(saver, session, path):
    """Restore the checkpoint the `checkpoint` state file of `path` names; returns its iteration (the digits after
    the last '-'), or None when the directory holds no checkpoint state."""
    import re
    from . import tf_checkpoint
    ckpt = tf_checkpoint.get_checkpoint_state(path)
    if ckpt is None:
        return None
    print("loading " + ckpt['model_checkpoint_path'] + "...")
    saver.restore(session, ckpt['model_checkpoint_path'])
    num_iter = int(re.match(r'.*-(\d*)$', ckpt['model_checkpoint_path']).group(1))
    print("done.")
    return num_iter


from .visualize import save_images, rescale_image, rescale_dm      # noqa: E402,F401  (tf_utils.py:101-147)
