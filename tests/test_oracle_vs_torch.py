"""Pins the numpy oracle against an INDEPENDENT implementation of the same maths
(torch CPU ops + autograd), float64 finite differences and analytic known answers.
The reference has no golden vectors for this path (parity unpinned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops

RNG = np.random.default_rng(7)


def t64(a):
    return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)


def torch_conv_same(x, w, b, s):
    """TF SAME conv via explicit asymmetric pad + F.conv2d (NHWC/HWIO in, NHWC out)."""
    n, h, wd, c = x.shape
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = ops.same_pads(h, kh, s)
    _, pl, pr = ops.same_pads(wd, kw, s)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=s)
    return y.permute(0, 2, 3, 1)


def torch_deconv_same(x, w, out_hw, s):
    """conv2d_transpose(SAME) = full conv_transpose2d sliced [pad_top : pad_top + out]."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, _ = ops.same_pads(out_hw[0], kh, s)
    _, pl, _ = ops.same_pads(out_hw[1], kw, s)
    # torch weight [Cin, Cout, kh, kw]; ours [kh,kw,Cout,Cin]
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=s)
    # pad in case the full output is smaller than pad+out (k < s never happens here)
    y = y[:, :, pt:pt + out_hw[0], pl:pl + out_hw[1]]
    assert y.shape[2] == out_hw[0] and y.shape[3] == out_hw[1]
    return y.permute(0, 2, 3, 1)


CONV_CASES = [  # n,h,w,cin,cout,k,s
    (2, 8, 8, 3, 4, 5, 2), (2, 8, 8, 4, 5, 5, 1), (1, 7, 9, 2, 3, 3, 2), (2, 6, 6, 1, 2, 3, 1),
    (1, 16, 16, 32, 8, 5, 2), (1, 4, 4, 6, 6, 3, 1), (1, 5, 5, 2, 2, 5, 2),
]


@pytest.mark.parametrize("n,h,w,ci,co,k,s", CONV_CASES)
def test_conv2d_fwd_bwd(n, h, w, ci, co, k, s):
    x = RNG.standard_normal((n, h, w, ci))
    wt = RNG.standard_normal((k, k, ci, co))
    b = RNG.standard_normal(co)
    y = ops.conv2d_fwd(x, wt, b, s, s)
    tx, tw, tb = t64(x), t64(wt), t64(b)
    ty = torch_conv_same(tx, tw, tb, s)
    assert y.shape == tuple(ty.shape)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = RNG.standard_normal(y.shape)
    ty.backward(torch.tensor(dy))
    dx, dw, db = ops.conv2d_bwd(x, wt, dy, s, s)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dw, tw.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-10, atol=1e-10)


DECONV_CASES = [  # n,hi,wi,cin,cout,k,s
    (2, 4, 4, 3, 5, 3, 2), (1, 8, 8, 4, 2, 5, 2), (2, 4, 4, 6, 3, 3, 1), (1, 16, 16, 8, 2, 5, 2),
    (1, 3, 5, 2, 2, 5, 2), (1, 6, 6, 4, 16, 3, 1),
]


@pytest.mark.parametrize("n,hi,wi,ci,co,k,s", DECONV_CASES)
def test_deconv2d_fwd_bwd(n, hi, wi, ci, co, k, s):
    x = RNG.standard_normal((n, hi, wi, ci))
    wt = RNG.standard_normal((k, k, co, ci))
    out_hw = (hi * s, wi * s)
    y = ops.deconv2d_fwd(x, wt, out_hw, s, s)
    tx, tw = t64(x), t64(wt)
    ty = torch_deconv_same(tx, tw, out_hw, s)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = RNG.standard_normal(y.shape)
    ty.backward(torch.tensor(dy))
    dx, dw = ops.deconv2d_bwd(x, wt, dy, s, s)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dw, tw.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_deconv_is_conv_backprop_input():
    """Appendix A.2: conv2d_transpose(x) == d/d(input) of SAME conv, applied to x."""
    x = RNG.standard_normal((2, 4, 4, 3))          # small side
    wt = RNG.standard_normal((5, 5, 6, 3))         # [kh,kw,Cout(big side),Cin(small side)]
    y = ops.deconv2d_fwd(x, wt, (8, 8), 2, 2)
    big = np.zeros((2, 8, 8, 6))
    dx, _, _ = ops.conv2d_bwd(big, wt, x, 2, 2)    # wt as HWIO with I=6,O=3
    np.testing.assert_allclose(y, dx, rtol=1e-12, atol=1e-12)


def test_same_pads_table():
    assert ops.same_pads(128, 5, 2) == (64, 1, 2)
    assert ops.same_pads(128, 3, 2) == (64, 0, 1)
    assert ops.same_pads(64, 5, 1) == (64, 2, 2)
    assert ops.same_pads(7, 3, 2) == (4, 1, 1)


def test_absact():
    x = np.array([-2.0, -0.0, 0.0, 1.5, 1e-30], dtype=np.float32)
    y = ops.absact_fwd(x, 'lrelu')
    np.testing.assert_allclose(y, np.where(x > 0, x, 0.2 * x), rtol=1e-6)
    g = ops.absact_bwd(x, np.ones_like(x), 'lrelu')
    np.testing.assert_allclose(g, [0.2, 0.6, 0.6, 1.0, 1.0], rtol=1e-6)   # slope 0.6 at exactly 0
    g = ops.absact_bwd(x, np.ones_like(x), 'relu')
    np.testing.assert_allclose(g, [0.0, 0.5, 0.5, 1.0, 1.0], rtol=1e-6)
    tx = torch.tensor(x.astype(np.float64), requires_grad=True)
    (0.6 * tx + 0.4 * tx.abs()).sum().backward()
    np.testing.assert_allclose(ops.absact_bwd(x, np.ones_like(x), 'lrelu'), tx.grad.numpy(), rtol=1e-6)


def torch_resampler(data, warp):
    n, h, w, c = data.shape
    gx = 2 * warp[..., 0] / (w - 1) - 1
    gy = 2 * warp[..., 1] / (h - 1) - 1
    grid = torch.stack((gx, gy), -1)
    out = F.grid_sample(data.permute(0, 3, 1, 2), grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return out.permute(0, 2, 3, 1)


def test_resampler_vs_grid_sample():
    n, h, w, c = 2, 9, 9, 3
    data = RNG.standard_normal((n, h, w, c))
    warp = RNG.uniform(-2.5, 11.0, size=(n, 6, 7, 2))
    # some exactly-integer, some exactly on the -1 / W validity edges
    warp[0, 0, 0] = (3.0, 4.0)
    warp[0, 0, 1] = (0.0, 0.0)
    warp[0, 0, 2] = (8.0, 8.0)
    warp[0, 0, 3] = (-0.5, 2.25)
    warp[0, 0, 4] = (8.5, 2.25)
    out = ops.resampler_fwd(data, warp)
    td, tw = t64(data), t64(warp)
    tout = torch_resampler(td, tw)
    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=1e-9, atol=1e-9)
    g = RNG.standard_normal(out.shape)
    tout.backward(torch.tensor(g))
    ddata, dwarp = ops.resampler_bwd(data, warp, g)
    np.testing.assert_allclose(ddata, td.grad.numpy(), rtol=1e-9, atol=1e-9)
    # grid_sample's warp-gradient agrees wherever the sample is not exactly on an integer
    # lattice line (there the one-sided derivative conventions may differ)
    frac_ok = (np.abs(warp - np.round(warp)) > 1e-9).all(-1)
    np.testing.assert_allclose(dwarp[frac_ok], tw.grad.numpy()[frac_ok], rtol=1e-8, atol=1e-8)


def test_resampler_validity_window():
    """Appendix A.3: zero unless x>-1, y>-1, x<W, y<H."""
    data = np.ones((1, 4, 4, 1))
    warp = np.array([[[[-1.0, 0.0], [-0.999, 0.0], [3.999, 3.0], [4.0, 3.0], [1.0, -1.0], [1.0, 4.0]]]])
    out = ops.resampler_fwd(data, warp)[0, 0, :, 0]
    np.testing.assert_allclose(out, [0.0, 0.001, 0.001, 0.0, 0.0, 0.0], atol=1e-12)


def test_zero_flow_gives_transposed_image():
    """Appendix A.4: coords() feeds (row, col) as (x, y) => zero flow transposes the image."""
    img = RNG.standard_normal((2, 8, 8, 3)).astype(np.float32)
    warp = ops.warp_pts_layer(np.zeros((2, 8, 8, 2), np.float32))
    assert warp[0, 3, 5, 0] == 3 and warp[0, 3, 5, 1] == 5
    gen = ops.resampler_fwd(img, warp)
    np.testing.assert_array_equal(gen, img.transpose(0, 2, 1, 3))


def test_unit_flow_shifts_source_rows():
    img = RNG.standard_normal((1, 8, 8, 1)).astype(np.float32)
    flow = np.zeros((1, 8, 8, 2), np.float32)
    flow[..., 1] = 1.0     # +1 in y (source row)
    gen = ops.resampler_fwd(img, ops.warp_pts_layer(flow))
    exp = np.zeros_like(img)
    # gen[i,j] = img[row=j+1, col=i]
    exp[0, :, :7, 0] = img[0, 1:, :, 0].T
    np.testing.assert_array_equal(gen, exp)


def test_resampler_rotation_demo_like_reference():
    """tests/test_resampler.py of the reference rotates a rectangle by 10 degrees; here on a
    synthetic 40x30 rectangle, checked against grid_sample."""
    h, w = 30, 40
    img = np.zeros((1, h, w, 3))
    img[0, 8:22, 10:30] = (1.0, 0.5, 0.25)
    X, Y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    th = np.deg2rad(10)
    xr = np.cos(th) * X - np.sin(th) * Y
    yr = np.sin(th) * X + np.cos(th) * Y
    warp = np.stack([np.clip(xr, 0, w - 1), np.clip(yr, 0, h - 1)], -1)[None]
    out = ops.resampler_fwd(img, warp)
    tout = torch_resampler(torch.tensor(img), torch.tensor(warp)).numpy()
    np.testing.assert_allclose(out, tout, atol=1e-12)
    assert 0.3 < out[..., 0].mean() / img[..., 0].mean() < 1.2


def test_losses():
    a = RNG.standard_normal((2, 4, 4, 3))
    b = RNG.standard_normal((2, 4, 4, 3))
    ta = t64(a)
    l = ((ta - torch.tensor(b)) ** 2).sum(3).mean()
    np.testing.assert_allclose(ops.euclidean_loss_fwd(a, b), l.item(), rtol=1e-12)
    l.backward()
    np.testing.assert_allclose(ops.euclidean_loss_bwd(a, b), ta.grad.numpy(), rtol=1e-12)
    ta = t64(a)
    l = (ta - torch.tensor(b)).abs().sum(3).mean()
    np.testing.assert_allclose(ops.l1_loss_fwd(a, b), l.item(), rtol=1e-12)
    l.backward()
    np.testing.assert_allclose(ops.l1_loss_bwd(a, b), ta.grad.numpy(), rtol=1e-12)


def test_adam_matches_tf_formula_not_torch():
    """Appendix A.7: epsilon outside the bias correction ('epsilon hat')."""
    p = RNG.standard_normal(50).astype(np.float32)
    p0 = p.copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for t in range(1, 4):
        g = RNG.standard_normal(50).astype(np.float32) * 1e-3
        p_before = p.astype(np.float64)
        m64 = 0.9 * m.astype(np.float64) + 0.1 * g
        v64 = 0.999 * v.astype(np.float64) + 0.001 * g.astype(np.float64) ** 2
        lr_t = 1e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        expect = p_before - lr_t * m64 / (np.sqrt(v64) + 1e-8)
        ops.adam_step(p, g, m, v, b1p, b2p, 1e-4)
        b1p = np.float32(b1p * np.float32(0.9))
        b2p = np.float32(b2p * np.float32(0.999))
        np.testing.assert_allclose(p, expect, rtol=2e-6, atol=1e-9)
    assert np.abs(p - p0).max() > 1e-5


def test_linear():
    x = RNG.standard_normal((3, 5))
    m = RNG.standard_normal((5, 4))
    b = RNG.standard_normal(4)
    dy = RNG.standard_normal((3, 4))
    tx, tm, tb = t64(x), t64(m), t64(b)
    ty = tx @ tm + tb
    np.testing.assert_allclose(ops.linear_fwd(x, m, b), ty.detach().numpy(), rtol=1e-12)
    ty.backward(torch.tensor(dy))
    dx, dm, db = ops.linear_bwd(x, m, dy)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(dm, tm.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-12)
