#!/usr/bin/env python
"""Generates tests/golden/appflow_golden.npz -- small input/output vectors for every operation of the hot path.

The reference (Python 2 + TensorFlow 1.3) cannot run in this environment and ships no golden vectors of its own
(SURVEY.md section 8c), so these vectors are produced by the oracle (oracle/ops.py, the CPU restatement of the TF-1.3
semantics) and, before being written, cross-checked against an independent float64 torch implementation
(oracle/torch_tape.py helpers / torch.nn.functional + autograd); the script aborts if the two disagree by more than 2e-6
of the tensor maximum.  They pin the oracle against regressions and give the HIP library fixed vectors to hit; they do not
replace a run of the reference ("parity unpinned").

    python tests/golden/make_golden.py          # rewrites appflow_golden.npz (deterministic: seeded)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'appflow_golden.npz')


def t64(a, grad=False):
    return torch.tensor(np.asarray(a, np.float64), requires_grad=grad)


def same_pad_torch(x, k, s):
    """NHWC float64 -> padded NCHW per TF 'SAME' (pad_before = total // 2)."""
    n, h, w, c = x.shape
    ph = max((-(-h // s) - 1) * s + k - h, 0)
    pw = max((-(-w // s) - 1) * s + k - w, 0)
    return F.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def check(name, a, b, tol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err < tol, (name, err)


def conv_case(rng, tag, n, h, w, c, k, ksz, s, out):
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ksz, ksz, c, k)) / np.sqrt(ksz * ksz * c)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32)
    y = ops.conv2d_fwd(x, wt, b, s, s)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    dx, dw, db = ops.conv2d_bwd(x, wt, dy, s, s)
    # independent: torch conv2d on explicitly SAME-padded input, autograd
    tx, tw, tb = t64(x, True), t64(wt, True), t64(b, True)
    ty = F.conv2d(same_pad_torch(tx, ksz, s), tw.permute(3, 2, 0, 1), tb, stride=s).permute(0, 2, 3, 1)
    ty.backward(t64(dy))
    check(tag + '/y', y, ty.detach().numpy()); check(tag + '/dx', dx, tx.grad.numpy())
    check(tag + '/dw', dw, tw.grad.numpy()); check(tag + '/db', db, tb.grad.numpy())
    out.update({tag + '/x': x, tag + '/w': wt, tag + '/b': b, tag + '/stride': np.int32(s), tag + '/y': y, tag + '/dy': dy,
                tag + '/dx': dx, tag + '/dw': dw, tag + '/db': db})


def deconv_case(rng, tag, n, hi, wi, ci, co, ksz, s, out):
    x = rng.standard_normal((n, hi, wi, ci)).astype(np.float32)
    wt = (rng.standard_normal((ksz, ksz, co, ci)) / np.sqrt(ksz * ksz * ci)).astype(np.float32)
    H, W = hi * s, wi * s
    y = ops.deconv2d_fwd(x, wt, (H, W), s, s)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    dx, dw = ops.deconv2d_bwd(x, wt, dy, s, s)
    # independent: conv2d_transpose == gradient of the SAME-padded conv w.r.t. its input
    tx, tw = t64(x, True), t64(wt, True)
    img = torch.zeros((n, H, W, co), dtype=torch.float64, requires_grad=True)
    feat = F.conv2d(same_pad_torch(img, ksz, s), tw.permute(3, 2, 0, 1), None, stride=s).permute(0, 2, 3, 1)
    ty, = torch.autograd.grad(feat, img, tx, create_graph=True)
    ty.backward(t64(dy))
    check(tag + '/y', y, ty.detach().numpy()); check(tag + '/dx', dx, tx.grad.numpy()); check(tag + '/dw', dw, tw.grad.numpy())
    out.update({tag + '/x': x, tag + '/w': wt, tag + '/stride': np.int32(s), tag + '/y': y, tag + '/dy': dy, tag + '/dx': dx, tag + '/dw': dw})


def main():
    rng = np.random.default_rng(20261003)
    out = {}
    conv_case(rng, 'conv5s2', 1, 9, 11, 4, 6, 5, 2, out)          # ragged sizes, SAME pad (1, 2)
    conv_case(rng, 'conv3s1', 2, 8, 8, 16, 16, 3, 1, out)          # matrix-core halo path on the device
    deconv_case(rng, 'deconv5s2', 1, 4, 5, 8, 3, 5, 2, out)
    deconv_case(rng, 'deconv3s2c16', 1, 8, 8, 16, 16, 3, 2, out)   # 4-phase matrix-core path on the device

    # linear
    x = rng.standard_normal((3, 10)).astype(np.float32); m = rng.standard_normal((10, 7)).astype(np.float32)
    b = rng.standard_normal(7).astype(np.float32); dy = rng.standard_normal((3, 7)).astype(np.float32)
    y = ops.linear_fwd(x, m, b); dx, dm, db = ops.linear_bwd(x, m, dy)
    check('fc/y', y, x.astype(np.float64) @ m.astype(np.float64) + b); check('fc/dm', dm, x.astype(np.float64).T @ dy.astype(np.float64))
    out.update({'fc/x': x, 'fc/m': m, 'fc/b': b, 'fc/dy': dy, 'fc/y': y, 'fc/dx': dx, 'fc/dm': dm, 'fc/db': db})

    # abs-based activations at and around zero (tf_utils.py:25-33): slope f1 at exactly 0
    a = np.array([-2.0, -1e-9, -0.0, 0.0, 1e-9, 0.5, 3.0], np.float32); g = np.ones_like(a)
    for kind in ('lrelu', 'relu'):
        out['act/' + kind + '/y'] = ops.absact_fwd(a, kind); out['act/' + kind + '/dx'] = ops.absact_bwd(a, g, kind)
    out['act/x'] = a
    assert out['act/lrelu/dx'][3] == np.float32(0.6) and out['act/relu/dx'][3] == np.float32(0.5)

    # warp + resampler, incl. integer coordinates, the validity window and far-outside points
    src = rng.standard_normal((2, 8, 8, 3)).astype(np.float32)
    flow = rng.uniform(-3, 3, (2, 8, 8, 2)).astype(np.float32)
    flow[0, 0, :4] = 0.0; flow[0, 1, :2, 0] = -2.0; flow[1, 7, :, 1] = 0.5; flow[1, 3, 3] = (40.0, -40.0)
    warp = ops.warp_pts_layer(flow); gen = ops.resampler_fwd(src, warp)
    gg = rng.standard_normal(gen.shape).astype(np.float32)
    dsrc, dwarp = ops.resampler_bwd(src, warp, gg)
    np.testing.assert_array_equal(ops.resampler_fwd(src, ops.warp_pts_layer(np.zeros_like(flow))), src.transpose(0, 2, 1, 3))
    out.update({'warp/src': src, 'warp/flow': flow, 'warp/pts': warp, 'warp/gen': gen, 'warp/dgen': gg, 'warp/dsrc': dsrc, 'warp/dpts': dwarp})

    # losses (mean over B*H*W of the channel sum, tf_utils.py:18-23)
    p = rng.uniform(0, 1, (2, 4, 4, 3)).astype(np.float32); q = rng.uniform(0, 1, (2, 4, 4, 3)).astype(np.float32)
    out.update({'loss/a': p, 'loss/b': q, 'loss/l2': np.float32(ops.euclidean_loss_fwd(p, q)), 'loss/l2_da': ops.euclidean_loss_bwd(p, q),
                'loss/l1': np.float32(ops.l1_loss_fwd(p, q)), 'loss/l1_da': ops.l1_loss_bwd(p, q)})
    check('loss/l2', out['loss/l2'], ((p.astype(np.float64) - q) ** 2).sum(3).mean())

    # TF ApplyAdam, three steps (epsilon outside the bias correction)
    w = rng.standard_normal(37).astype(np.float32); mm = np.zeros_like(w); vv = np.zeros_like(w)
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    out['adam/p0'] = w.copy()
    for i in range(3):
        gr = rng.standard_normal(37).astype(np.float32) * np.float32(10.0 ** (i - 2))
        out['adam/g%d' % i] = gr
        ops.adam_step(w, gr, mm, vv, b1p, b2p, 1e-4)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
        out['adam/p%d' % (i + 1)] = w.copy(); out['adam/m%d' % (i + 1)] = mm.copy(); out['adam/v%d' % (i + 1)] = vv.copy()

    np.savez_compressed(OUT, **out)
    print('wrote', OUT, '%d arrays, %d bytes' % (len(out), os.path.getsize(OUT)))


if __name__ == '__main__':
    main()
