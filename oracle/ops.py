"""numpy restatement of the TF-1.3 ops behind tf_utils.py -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Every function cites the reference
call site it stands in for; the TF kernel semantics are SURVEY.md Appendix A.

Layouts (reference): activations NHWC, conv filters HWIO [kh,kw,Cin,Cout],
deconv filters [kh,kw,Cout,Cin], fc matrices [in,out].
All functions are dtype-generic (float32 for parity, float64 for gradient checks).
"""
import numpy as np
from numpy.lib.stride_tricks import as_strided


# --------------------------------------------------------------------------- padding
def same_pads(size, k, s):
    """TF 'SAME' geometry (Appendix A.1): returns (out, pad_before, pad_after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _im2col(x, kh, kw, sh, sw):
    """x [N,H,W,C] -> (cols [N*Ho*Wo, kh*kw*C], geometry)."""
    n, h, w, c = x.shape
    ho, pt, pb = same_pads(h, kh, sh)
    wo, pl, pr = same_pads(w, kw, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    sn, s_h, s_w, sc = xp.strides
    view = as_strided(xp, shape=(n, ho, wo, kh, kw, c),
                      strides=(sn, s_h * sh, s_w * sw, s_h, s_w, sc), writeable=False)
    return view.reshape(n * ho * wo, kh * kw * c), (n, h, w, c, ho, wo, pt, pb, pl, pr)


def _col2im(dcol, geom, kh, kw, sh, sw):
    n, h, w, c, ho, wo, pt, pb, pl, pr = geom
    dxp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=dcol.dtype)
    d6 = dcol.reshape(n, ho, wo, kh, kw, c)
    for p in range(kh):
        for q in range(kw):
            dxp[:, p:p + (ho - 1) * sh + 1:sh, q:q + (wo - 1) * sw + 1:sw, :] += d6[:, :, :, p, q, :]
    return dxp[:, pt:pt + h, pl:pl + w, :]


# --------------------------------------------------------------------------- conv2d (tf_utils.py:81-82)
def conv2d_fwd(x, w, b, sh, sw):
    """tf.nn.conv2d(x, w, [1,sh,sw,1], 'SAME') + b   (tf_utils.py:81-82, Appendix A.1)."""
    kh, kw, ci, co = w.shape
    cols, g = _im2col(x, kh, kw, sh, sw)
    y = cols @ w.reshape(kh * kw * ci, co)
    if b is not None:
        y = y + b
    return y.reshape(g[0], g[4], g[5], co)


def conv2d_bwd(x, w, dy, sh, sw, need_dx=True):
    """Gradients of conv2d_fwd: returns (dx, dw, db)  (Conv2DBackpropInput/Filter, BiasAddGrad)."""
    kh, kw, ci, co = w.shape
    cols, g = _im2col(x, kh, kw, sh, sw)
    dy2 = dy.reshape(-1, co)
    dw = (cols.T @ dy2).reshape(kh, kw, ci, co)
    db = dy2.sum(axis=0)
    dx = None
    if need_dx:
        dcol = dy2 @ w.reshape(kh * kw * ci, co).T
        dx = _col2im(dcol, g, kh, kw, sh, sw)
    return dx, dw, db


# --------------------------------------------------------------------------- conv2d_transpose (tf_utils.py:96-97)
def deconv2d_fwd(x, w, out_hw, sh, sw):
    """tf.nn.conv2d_transpose(x, w, output_shape, [1,sh,sw,1]) with default SAME, no bias
    (tf_utils.py:96-97).  Defined by TF as conv2d_backprop_input(input_sizes=output_shape,
    filter=w, out_backprop=x) (Appendix A.2).  w is [kh,kw,Cout,Cin]."""
    kh, kw, co, ci = w.shape
    n, hi, wi, _ = x.shape
    ho, wo = out_hw
    hi2, pt, pb = same_pads(ho, kh, sh)
    wi2, pl, pr = same_pads(wo, kw, sw)
    assert (hi2, wi2) == (hi, wi), "output_shape inconsistent with SAME/stride"
    geom = (n, ho, wo, co, hi, wi, pt, pb, pl, pr)
    dcol = x.reshape(-1, ci) @ w.reshape(kh * kw * co, ci).T
    return _col2im(dcol, geom, kh, kw, sh, sw)


def deconv2d_bwd(x, w, dy, sh, sw):
    """Gradients of deconv2d_fwd: returns (dx, dw).  dx is the SAME conv of dy with w
    (as an HWIO filter with I=Cout, O=Cin); dw is the conv filter-gradient with the roles of
    x and dy swapped (Appendix A.2)."""
    kh, kw, co, ci = w.shape
    cols, _ = _im2col(dy, kh, kw, sh, sw)           # [N*hi*wi, kh*kw*co]
    dx = (cols @ w.reshape(kh * kw * co, ci)).reshape(x.shape)
    dw = (cols.T @ x.reshape(-1, ci)).reshape(kh, kw, co, ci)
    return dx, dw


# --------------------------------------------------------------------------- linear (tf_utils.py:67)
def linear_fwd(x, m, b):
    """tf.matmul(input_, matrix) + b  (tf_utils.py:67)."""
    return x @ m + b


def linear_bwd(x, m, dy):
    return dy @ m.T, x.T @ dy, dy.sum(axis=0)


# --------------------------------------------------------------------------- activations (tf_utils.py:25-33)
def act_coeffs(kind, leak=0.2):
    """(f1, f2) of  f1*x + f2*abs(x):  lrelu tf_utils.py:29-33, relu tf_utils.py:25-27."""
    if kind == 'lrelu':
        return 0.5 * (1 + leak), 0.5 * (1 - leak)
    if kind == 'relu':
        return 0.5, 0.5
    raise ValueError(kind)


def absact_fwd(x, kind, leak=0.2):
    f1, f2 = act_coeffs(kind, leak)
    return x.dtype.type(f1) * x + x.dtype.type(f2) * np.abs(x)


def absact_bwd(x, dy, kind, leak=0.2):
    """TF: d abs(x)/dx = sign(x), sign(0) = 0  => slope at 0 is f1 (Appendix A.5)."""
    f1, f2 = act_coeffs(kind, leak)
    return dy * (x.dtype.type(f1) + x.dtype.type(f2) * np.sign(x))


def tanh_fwd(x):
    return np.tanh(x)


def tanh_bwd(y, dy):
    return dy * (1 - y * y)


# --------------------------------------------------------------------------- coords / warp (tf_utils.py:35-52)
def coords(h, w, batch, dtype=np.float32):
    """tf_utils.py:44-52: X,Y = meshgrid(x,y); stack((Y,X),2) -> channel 0 = row index i,
    channel 1 = column index j (Appendix A.4)."""
    y = np.arange(h, dtype=dtype)
    x = np.arange(w, dtype=dtype)
    X, Y = np.meshgrid(x, y)
    c = np.stack((Y, X), axis=2)[None]
    return np.tile(c, (batch, 1, 1, 1))


def warp_pts_layer(flow):
    """tf_utils.py:35-38."""
    n, h, w, _ = flow.shape
    return flow + coords(h, w, n, flow.dtype)


# --------------------------------------------------------------------------- resampler (tf_utils.py:40-42)
def _gather(data, b, yy, xx):
    n, h, w, c = data.shape
    ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
    xs = np.clip(xx, 0, w - 1)
    ys = np.clip(yy, 0, h - 1)
    v = data[b, ys, xs]                      # [...,C]
    return np.where(ok[..., None], v, data.dtype.type(0)), ok


def resampler_fwd(data, warp):
    """tf.contrib.resampler.resampler(data, warp)  (tf_utils.py:42, Appendix A.3).
    warp[...,0] is x (column), warp[...,1] is y (row); zero outside."""
    n, h, w, c = data.shape
    x = warp[..., 0]
    y = warp[..., 1]
    valid = (x > -1) & (y > -1) & (x < w) & (y < h)
    fx = np.floor(x).astype(np.int64)
    fy = np.floor(y).astype(np.int64)
    cx = fx + 1
    cy = fy + 1
    dx = (cx - x).astype(data.dtype)[..., None]
    dy = (cy - y).astype(data.dtype)[..., None]
    b = np.arange(n).reshape((n,) + (1,) * (x.ndim - 1))
    b = np.broadcast_to(b, x.shape)
    i_ff, _ = _gather(data, b, fy, fx)
    i_cc, _ = _gather(data, b, cy, cx)
    i_fc, _ = _gather(data, b, cy, fx)       # (fx, cy)
    i_cf, _ = _gather(data, b, fy, cx)       # (cx, fy)
    one = data.dtype.type(1)
    out = dx * dy * i_ff + (one - dx) * (one - dy) * i_cc + dx * (one - dy) * i_fc + (one - dx) * dy * i_cf
    return np.where(valid[..., None], out, data.dtype.type(0))


def resampler_bwd(data, warp, g, need_ddata=True):
    """Gradients of resampler_fwd w.r.t. (data, warp)  (Appendix A.3)."""
    n, h, w, c = data.shape
    x = warp[..., 0]
    y = warp[..., 1]
    valid = (x > -1) & (y > -1) & (x < w) & (y < h)
    fx = np.floor(x).astype(np.int64)
    fy = np.floor(y).astype(np.int64)
    cx = fx + 1
    cy = fy + 1
    dx = (cx - x).astype(data.dtype)[..., None]
    dy = (cy - y).astype(data.dtype)[..., None]
    b = np.broadcast_to(np.arange(n).reshape((n,) + (1,) * (x.ndim - 1)), x.shape)
    i_ff, ok_ff = _gather(data, b, fy, fx)
    i_cc, ok_cc = _gather(data, b, cy, cx)
    i_fc, ok_fc = _gather(data, b, cy, fx)
    i_cf, ok_cf = _gather(data, b, fy, cx)
    one = data.dtype.type(1)
    gx = (g * (dy * (i_cf - i_ff) + (one - dy) * (i_cc - i_fc))).sum(-1)
    gy = (g * (dx * (i_fc - i_ff) + (one - dx) * (i_cc - i_cf))).sum(-1)
    dwarp = np.stack((gx, gy), axis=-1)
    dwarp = np.where(valid[..., None], dwarp, data.dtype.type(0))
    ddata = None
    if need_ddata:
        ddata = np.zeros_like(data)
        gv = np.where(valid[..., None], g, data.dtype.type(0))
        for (yy, xx, wt, ok) in ((fy, fx, dx * dy, ok_ff), (cy, cx, (one - dx) * (one - dy), ok_cc),
                                 (cy, fx, dx * (one - dy), ok_fc), (fy, cx, (one - dx) * dy, ok_cf)):
            m = ok & valid
            np.add.at(ddata, (b[m], yy[m], xx[m]), (gv * wt)[m])
    return ddata, dwarp


# --------------------------------------------------------------------------- losses (tf_utils.py:18-23)
def euclidean_loss_fwd(a, b):
    """tf.reduce_mean(tf.reduce_sum(tf.pow(a-b, 2), 3))  (tf_utils.py:18-19)."""
    d = a - b
    return (d * d).sum(axis=3).mean(dtype=np.float64).astype(a.dtype)


def euclidean_loss_bwd(a, b, gl=1.0):
    n, h, w, _ = a.shape
    return (a - b) * a.dtype.type(2.0 * gl / (n * h * w))


def l1_loss_fwd(a, b):
    """tf.reduce_mean(tf.reduce_sum(tf.abs(a-b), 3))  (tf_utils.py:22-23)."""
    return np.abs(a - b).sum(axis=3).mean(dtype=np.float64).astype(a.dtype)


def l1_loss_bwd(a, b, gl=1.0):
    n, h, w, _ = a.shape
    return np.sign(a - b) * a.dtype.type(gl / (n * h * w))


# --------------------------------------------------------------------------- Adam (appearance_flow_model.py:77)
def adam_alpha(lr, beta1_power, beta2_power, dtype=np.float32):
    """lr_t of TF's ApplyAdam: lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in the
    variable's dtype from the running beta-power accumulators (Appendix A.7)."""
    dt = dtype
    return dt(dt(lr) * np.sqrt(dt(1) - dt(beta2_power)) / (dt(1) - dt(beta1_power)))


def adam_step(p, g, m, v, beta1_power, beta2_power, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer's ApplyAdam kernel (appearance_flow_model.py:77, Appendix A.7):
        alpha = lr*sqrt(1-beta2_power)/(1-beta1_power)
        m += (g - m)*(1-beta1);  v += (g*g - v)*(1-beta2);  p -= m*alpha/(sqrt(v)+eps)
    epsilon sits OUTSIDE the bias correction.  beta*_power are the accumulators BEFORE this
    step (beta^t for the t-th step, t 1-based); the caller multiplies them by beta afterwards.
    Updates p, m, v in place in p's dtype."""
    dt = p.dtype.type
    alpha = adam_alpha(lr, beta1_power, beta2_power, dt)
    m += (g - m) * (dt(1) - dt(beta1))
    v += (g * g - v) * (dt(1) - dt(beta2))
    p -= (m * alpha) / (np.sqrt(v) + dt(eps))


# --------------------------------------------------------------------------- initialisers (tf_utils.py:58-95)
def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal_initializer: redraw samples beyond 2 sigma (Appendix A.8).
    Matches TF in distribution only (TF's Philox stream is not reproducible here)."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2
    return (out * stddev).astype(np.float32)


def random_normal(rng, shape, stddev):
    return (rng.standard_normal(shape) * stddev).astype(np.float32)
