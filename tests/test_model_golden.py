"""Whole-graph golden vectors (tests/golden/model_golden.npz, written by tests/golden/make_model_golden.py): fingerprints --
L2 norm, sum and 16 seeded elements -- of the regenerated weights, every output, every gradient and, for AppearanceFlowModel,
the loss sequence and the weights after three TF-Adam steps.  CPU: the oracle reproduces them.  GPU: the HIP graph (reference-
named model classes, C ABI underneath) hits them.

Still "parity unpinned": the vectors come from this repo's oracle, cross-checked against float64 torch autograd when they were
written; the reference holds no expected outputs (its only fixture, multi_view_model/tests/rectangle.png, is an input)."""
import os

import numpy as np
import pytest

from oracle import models as omodels
from tests.golden.make_model_golden import fingerprint, init_variables, model_cases

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'model_golden.npz'))
CASES = model_cases()


def _fp_close(got, want, tol, what):
    """norm and sum relative to the norm; sampled elements relative to the largest of them"""
    scale = max(abs(want[0]), 1e-30)
    assert abs(got[0] - want[0]) <= tol * scale, (what, 'norm', got[0], want[0])
    assert abs(got[1] - want[1]) <= 40 * tol * scale, (what, 'sum', got[1], want[1])       # a sum of n terms: sqrt(n) x the per-element noise
    assert np.abs(got[2:] - want[2:]).max() <= 3 * tol * max(np.abs(want[2:]).max(), scale * 1e-3), (what, 'samples')      # single elements: 3 x the norm's bar


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_reproduces_model_golden(name):
    builder, feeds, _ = CASES[name]
    variables = init_variables(builder, feeds)
    for k, v in variables.items():
        np.testing.assert_array_equal(fingerprint(v, k), G['%s/w0/%s' % (name, k)])          # same seeds, same initialisers, same order
    out, grads, _ = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds)
    np.testing.assert_allclose(float(out['loss']), float(G[name + '/loss']), rtol=1e-6)
    for k, v in out.items():
        if np.ndim(v) > 0:
            _fp_close(fingerprint(v, k), G['%s/out/%s' % (name, k)], 2e-6, k)
    assert {k.split('/grad/', 1)[1] for k in G.files if k.startswith(name + '/grad/')} == set(grads)
    for k, g in grads.items():
        _fp_close(fingerprint(g, k), G['%s/grad/%s' % (name, k)], 2e-5, k)


def _build(name, conf_extra):
    if name == 'appflow':
        from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel as M
        kw = dict(load_tfrec=False, build_loss=True)
    elif name == 'basepred':
        from dynamic_multiview_3d_amd.main_model import Base_Prediction_Model as M
        kw = dict(load_tfrec=False)
    else:
        from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow as M
        kw = dict(load_tfrec=False)
    return M(dict(conf_extra, batch_size=2, learning_rate=1e-4), device='cuda', **kw)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_hip_graph_hits_model_golden(name):
    """Loss to 1e-5 and output fingerprints to 1e-3 of the tensor norm (north_star's bar; measured ~1e-5).  Gradient
    fingerprints to 1e-2: a fixed vector cannot follow the device across the kinks of the reference's own function -- about
    1e-4 of the samples sit within rounding of an integer coordinate, where the bilinear sampler's floor picks the other cell
    and the image gradient (noise-dominated renders) is uncorrelated: that alone moves the norm of the earliest layers'
    gradients by ~2e-3 (measured on e0/w).  The 1e-3 gradient bar is held where the kinks can be followed: against the live
    oracle evaluated at the device's decisions (tests/test_gpu_model.py, tests/test_gpu_layers.py).

    Round 3 measured what the bar has to absorb (tools/golden_probe.py on the GPU box; worst gradient tensor per model, relative to
    the tensor's norm): norm 1.7e-3 / 1.1e-3 / 3.0e-3 / 0.7e-3, sampled elements 9.8e-3 / 6.9e-3 / 1.5e-2 / 9.1e-3 of their largest
    for appflow / basepred / multiobj_fc / multiobj_conv.  basepred has no sampler at all, so its 1.1e-3 is the activation kinks
    alone (lrelu' at pre-activations within rounding of 0: ~20 of 10^7 units take the other slope on the device); a 1e-4 pixel
    perturbation of every sampling coordinate moves the oracle's own gradient norms by 2.7e-4 (noisy renders) or 0.8e-4 (the
    same renders blurred) -- smoother inputs would not remove the activation part, so the fixture keeps the renders of SURVEY 8d."""
    import torch
    builder, feeds, conf = CASES[name]
    model = _build(name, conf)
    g = model.graph
    variables = init_variables(builder, feeds)
    assert list(variables) == list(g.variables)
    g.set_variables(variables)
    model.feed(**feeds)
    g.run_forward()
    g.run_backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(g.loss_buf[0]), float(G[name + '/loss']), rtol=1e-5)
    outs = [k.split('/out/', 1)[1] for k in G.files if k.startswith(name + '/out/')]
    checked = 0
    for k in outs:
        t = getattr(model, k, None)
        if t is not None and hasattr(t, 'numpy'):
            _fp_close(fingerprint(t.numpy(), k), G['%s/out/%s' % (name, k)], 1e-3, k)
            checked += 1
    assert checked >= 1
    grads = g.get_gradients()
    assert {k.split('/grad/', 1)[1] for k in G.files if k.startswith(name + '/grad/')} == set(grads)
    for k, gr in grads.items():
        _fp_close(fingerprint(gr, k), G['%s/grad/%s' % (name, k)], 1e-2, k)
    if name == 'appflow':
        losses = [float(model.train_step()) for _ in range(3)]
        np.testing.assert_allclose(losses, G[name + '/losses'], rtol=1e-4)
        got = g.get_variables()
        for k, v in got.items():
            # three Adam steps move every element by <= 3 lr whatever its gradient's size (Adam normalises): an element whose
            # gradient is within rounding of zero may move the other way, so the norm is pinned to 1e-5 of itself plus 2 % of
            # the largest move the three steps can make, 3 lr sqrt(n).  Bit-exactness of the update itself is held elsewhere
            # (tests/test_gpu_ops.py::test_adam_bit_exact_vs_oracle, the fused / pipelined / unfused equalities).
            want = G['%s/w3/%s' % (name, k)]
            slack = 0.02 * 3 * 1e-4 * np.sqrt(v.size)
            assert abs(fingerprint(v, k)[0] - want[0]) <= 1e-5 * max(want[0], 1e-30) + slack, k


@pytest.mark.gpu
def test_hip_fc_hits_op_golden():
    """the fc vectors of tests/golden/appflow_golden.npz through the C ABI (small_fc kernels)"""
    import ctypes as C
    import torch
    from dynamic_multiview_3d_amd import _lib
    from tests.gpu_utils import dev, host, stream, Ws
    g0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'appflow_golden.npz'))
    L = _lib.lib()
    x, m, b, dy = g0['fc/x'], g0['fc/m'], g0['fc/b'], g0['fc/dy']
    B, fin = x.shape
    fout = m.shape[1]
    ws = Ws(int(L.fc_workspace_bytes(B, fin, fout)))
    dx_, dm, db, ddy = dev(x), dev(m), dev(b), dev(dy)
    y = torch.full((B, fout), float('nan'), device='cuda')
    epi = _lib.epilogue(db.data_ptr())
    L.fc_fwd(B, fin, fout, dx_.data_ptr(), fin, dm.data_ptr(), y.data_ptr(), fout, C.byref(epi), ws.ptr, ws.bytes, stream())
    np.testing.assert_allclose(host(y), g0['fc/y'], rtol=0, atol=3e-6 * np.abs(g0['fc/y']).max())
    gx = torch.full((B, fin), float('nan'), device='cuda')
    epi0 = _lib.epilogue()
    L.fc_dgrad(B, fin, fout, ddy.data_ptr(), fout, dm.data_ptr(), gx.data_ptr(), fin, C.byref(epi0), ws.ptr, ws.bytes, stream())
    np.testing.assert_allclose(host(gx), g0['fc/dx'], rtol=0, atol=3e-6 * np.abs(g0['fc/dx']).max())
    gm = torch.full((fin, fout), float('nan'), device='cuda'); gb = torch.full((fout,), float('nan'), device='cuda')
    L.fc_wgrad(B, fin, fout, dx_.data_ptr(), fin, ddy.data_ptr(), fout, gm.data_ptr(), gb.data_ptr(), ws.ptr, ws.bytes, stream())
    np.testing.assert_allclose(host(gm), g0['fc/dm'], rtol=0, atol=3e-6 * np.abs(g0['fc/dm']).max())
    np.testing.assert_allclose(host(gb), g0['fc/db'], rtol=0, atol=3e-6 * np.abs(g0['fc/db']).max())
