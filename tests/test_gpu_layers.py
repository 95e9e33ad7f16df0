"""Per-layer parity AT THE BENCHMARKED SHAPES: every conv / deconv / fc layer of the timed configurations, at the batch
they are timed at, through the C ABI against oracle/ops.py -- forward (bias + activation epilogue), data gradient (with the
fused act'(saved output) mask the model's reverse pass uses) and filter / bias gradient, on the pixel strides of the concat
buffers the model really reads and writes.  tests/test_label_coverage.py proves (on the CPU) that these cases dispatch every
kernel instance the recorded steps launch.

Tolerance: 2e-5 of the tensor maximum per op (split-bf16 products drop terms below 2^-16 relative; fp32 accumulation order
differs from BLAS); north_star's bar is 1e-3."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ops
from oracle import models as omodels
from dynamic_multiview_3d_amd import _lib
from tests import layer_cases as LC
from tests.gpu_utils import dev, host, stream, Ws, rel_err, conv_ws

pytestmark = pytest.mark.gpu
TOL = 2e-5


def L():
    return _lib.lib()


def _wide(rng, shape, ld):
    """dense random tensor `shape` embedded at channel 0 of a buffer with pixel stride ld"""
    buf = rng.standard_normal(shape[:-1] + (ld,)).astype(np.float32)
    return buf, np.ascontiguousarray(buf[..., :shape[-1]])


def _slopes(saved):
    return np.where(saved > 0, 1.0, np.where(saved < 0, 0.2, 0.6)).astype(np.float32)


def run_conv_case(case, seed=0):
    kind, n, h, w, c, k, ksz, s, img_ld, feat_ld, need_dx = case
    rng = np.random.default_rng(seed)
    g = _lib.conv_geom(n, h, w, c, k, ksz, ksz, s, s, img_ld, feat_ld)
    ho, wo = g.Ho, g.Wo
    ws = conv_ws(g)
    if kind == LC.CONV:
        xbuf, x = _wide(rng, (n, h, w, c), img_ld)
        wt = (rng.standard_normal((ksz, ksz, c, k)) / np.sqrt(ksz * ksz * c)).astype(np.float32)
        b = rng.standard_normal(k).astype(np.float32)
        dxb, dw_, db_ = dev(xbuf), dev(wt), dev(b)
        ybuf = torch.full((n, ho, wo, feat_ld), 7.0, device='cuda')
        epi = _lib.epilogue(db_.data_ptr(), _lib.ACT_LRELU, 0.2)
        L().conv2d_fwd(C.byref(g), dxb.data_ptr(), dw_.data_ptr(), ybuf.data_ptr(), C.byref(epi), ws.ptr, ws.bytes, stream())
        ref = ops.absact_fwd(ops.conv2d_fwd(x, wt, b, s, s), 'lrelu')
        got = host(ybuf)
        assert rel_err(got[..., :k], ref) < TOL
        assert np.all(got[..., k:] == 7.0)                                  # channels outside the slice are untouched
        dybuf, dy = _wide(rng, ref.shape, feat_ld)
        rdx, rdw, rdb = ops.conv2d_bwd(x, wt, dy, s, s, need_dx=need_dx)
        ddy = dev(dybuf)
        if need_dx:
            saved = rng.standard_normal(xbuf.shape).astype(np.float32)
            saved[:, ::3, ::2, :] = 0.0                                       # exact zeros: slope 0.6 (sign(0) = 0)
            dsaved = dev(saved)
            gxb = torch.full((n, h, w, img_ld), 7.0, device='cuda')
            epi = _lib.epilogue(None, 0, 0.2, _lib.ACT_LRELU, 0.2, dsaved.data_ptr(), img_ld)
            L().conv2d_dgrad(C.byref(g), ddy.data_ptr(), dw_.data_ptr(), gxb.data_ptr(), C.byref(epi), ws.ptr, ws.bytes, stream())
            got = host(gxb)
            assert rel_err(got[..., :c], rdx * _slopes(saved[..., :c])) < TOL
            assert np.all(got[..., c:] == 7.0)
        gw = torch.full(wt.shape, float('nan'), device='cuda')
        gb = torch.full((k,), float('nan'), device='cuda')
        L().conv2d_wgrad(C.byref(g), dxb.data_ptr(), ddy.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.ptr, ws.bytes, stream())
        assert rel_err(host(gw), rdw) < TOL
        assert rel_err(host(gb), rdb) < TOL
    else:
        xbuf, x = _wide(rng, (n, ho, wo, k), feat_ld)                        # feature side in
        wt = (rng.standard_normal((ksz, ksz, c, k)) / np.sqrt(ksz * ksz * k)).astype(np.float32)
        dxb, dw_ = dev(xbuf), dev(wt)
        ybuf = torch.full((n, h, w, img_ld), 7.0, device='cuda')
        act = _lib.ACT_LRELU if c > 4 else _lib.ACT_NONE
        epi = _lib.epilogue(None, act, 0.2)
        L().deconv2d_fwd(C.byref(g), dxb.data_ptr(), dw_.data_ptr(), ybuf.data_ptr(), C.byref(epi), ws.ptr, ws.bytes, stream())
        ref = ops.deconv2d_fwd(x, wt, (h, w), s, s)
        if c > 4:
            ref = ops.absact_fwd(ref, 'lrelu')
        got = host(ybuf)
        assert rel_err(got[..., :c], ref) < TOL
        assert np.all(got[..., c:] == 7.0)
        dybuf, dy = _wide(rng, ref.shape, img_ld)
        rdx, rdw = ops.deconv2d_bwd(x, wt, dy, s, s)
        ddy = dev(dybuf)
        saved = rng.standard_normal(xbuf.shape).astype(np.float32)
        saved[:, ::3, ::2, :] = 0.0
        dsaved = dev(saved)
        gxb = torch.full((n, ho, wo, feat_ld), 7.0, device='cuda')
        epi = _lib.epilogue(None, 0, 0.2, _lib.ACT_LRELU, 0.2, dsaved.data_ptr(), feat_ld)
        L().deconv2d_dgrad(C.byref(g), ddy.data_ptr(), dw_.data_ptr(), gxb.data_ptr(), C.byref(epi), ws.ptr, ws.bytes, stream())
        got = host(gxb)
        assert rel_err(got[..., :k], rdx * _slopes(saved[..., :k])) < TOL
        assert np.all(got[..., k:] == 7.0)
        gw = torch.full(wt.shape, float('nan'), device='cuda')
        L().deconv2d_wgrad(C.byref(g), dxb.data_ptr(), ddy.data_ptr(), gw.data_ptr(), ws.ptr, ws.bytes, stream())
        assert rel_err(host(gw), rdw) < TOL


def run_fc_case(case, seed=0):
    B, fin, fout, x_ld, y_ld = case
    rng = np.random.default_rng(seed)
    xbuf, x = _wide(rng, (B, fin), x_ld)
    m = (rng.standard_normal((fin, fout)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    ws = Ws(int(L().fc_workspace_bytes(B, fin, fout)))
    dxb, dm, db = dev(xbuf), dev(m), dev(b)
    ybuf = torch.full((B, y_ld), 7.0, device='cuda')
    epi = _lib.epilogue(db.data_ptr(), _lib.ACT_LRELU, 0.2)
    L().fc_fwd(B, fin, fout, dxb.data_ptr(), x_ld, dm.data_ptr(), ybuf.data_ptr(), y_ld, C.byref(epi), ws.ptr, ws.bytes, stream())
    ref = ops.absact_fwd(ops.linear_fwd(x, m, b), 'lrelu')
    got = host(ybuf)
    assert rel_err(got[:, :fout], ref) < TOL
    assert np.all(got[:, fout:] == 7.0)
    dybuf, dy = _wide(rng, ref.shape, y_ld)
    rdx, rdm, rdb = ops.linear_bwd(x, m, dy)
    ddy = dev(dybuf)
    saved = rng.standard_normal(xbuf.shape).astype(np.float32)
    saved[::3, ::2] = 0.0
    dsaved = dev(saved)
    gxb = torch.full((B, x_ld), 7.0, device='cuda')
    epi = _lib.epilogue(None, 0, 0.2, _lib.ACT_LRELU, 0.2, dsaved.data_ptr(), x_ld)
    L().fc_dgrad(B, fin, fout, ddy.data_ptr(), y_ld, dm.data_ptr(), gxb.data_ptr(), x_ld, C.byref(epi), ws.ptr, ws.bytes, stream())
    got = host(gxb)
    assert rel_err(got[:, :fin], rdx * _slopes(saved[:, :fin])) < TOL
    assert np.all(got[:, fin:] == 7.0)
    gm = torch.full((fin, fout), float('nan'), device='cuda')
    gb = torch.full((fout,), float('nan'), device='cuda')
    L().fc_wgrad(B, fin, fout, dxb.data_ptr(), x_ld, ddy.data_ptr(), y_ld, gm.data_ptr(), gb.data_ptr(), ws.ptr, ws.bytes, stream())
    assert rel_err(host(gm), rdm) < TOL
    assert rel_err(host(gb), rdb) < TOL


@pytest.mark.parametrize("case", LC.APPFLOW_B64, ids=LC.case_id)
def test_appflow_layers_at_batch_64(case):
    """appearance_flow_model.py:88-125 at conf.py:21's batch: the kernels bench.py times."""
    run_conv_case(case)


@pytest.mark.parametrize("case", LC.BASEPRED_B128, ids=LC.case_id)
def test_base_prediction_layers_at_batch_128(case):
    """main_model.py:96-137 at BASELINE config 3's batch."""
    run_conv_case(case)


@pytest.mark.parametrize("case", LC.MULTIOBJ_256_B32, ids=LC.case_id)
def test_multiobject_256_layers_at_batch_32(case):
    """multiobject_appflow.py:93-153 at 256 x 256 (BASELINE config 5), batch 32 per GPU."""
    run_conv_case(case)


@pytest.mark.parametrize("case", LC.EXTRA_KERNEL_CASES, ids=LC.case_id)
def test_kernel_instances_no_shipped_configuration_reaches(case):
    """Kernel instances the planner can pick but none of the three timed configurations does (LC.EXTRA_KERNEL_CASES says which
    and why): held to the same per-layer bar."""
    run_conv_case(case)


@pytest.mark.parametrize("case", LC.FC_B64 + LC.FC_HIGHDIM + LC.FC_MULTIOBJ, ids=LC.case_id)
def test_fc_layers_at_benchmark_batch(case):
    run_fc_case(case)


@pytest.mark.parametrize("case", [c for c in LC.FC_B64 + LC.FC_HIGHDIM if min(c[1], c[2]) >= 64], ids=LC.case_id)
def test_fused_fc_wgrad_adam_equals_wgrad_then_adam(case):
    """mv3d_fc_wgrad_adam (the optimiser of an fc matrix inside its filter-gradient kernel) against the two calls it replaces,
    mv3d_fc_wgrad (held to the oracle above) followed by mv3d_adam_step_dev (held to the oracle in test_gpu_ops.py): parameter
    and both Adam slots bit-identical after two steps with a changing bias correction; bias gradient identical."""
    B, fin, fout, x_ld, y_ld = case
    if not L().fc_wgrad_adam_supported(B, fin, fout, x_ld, y_ld):
        pytest.skip("layer is not one of the fused kernel's")
    rng = np.random.default_rng(1)
    ws = Ws(int(L().fc_workspace_bytes(B, fin, fout)))
    p0 = (rng.standard_normal((fin, fout)) / np.sqrt(fin)).astype(np.float32)
    state = np.array([1e-4, 0.9, 0.999, 1e-8, 0.9, 0.999, 1.0, 0.0], np.float32)
    st_a, st_b = dev(state), dev(state)
    pa, ma, va = dev(p0), torch.zeros(fin, fout, device='cuda'), torch.zeros(fin, fout, device='cuda')
    pb, mb, vb = dev(p0), torch.zeros(fin, fout, device='cuda'), torch.zeros(fin, fout, device='cuda')
    for step in range(2):
        xbuf, _ = _wide(rng, (B, fin), x_ld)
        dybuf, _ = _wide(rng, (B, fout), y_ld)
        dybuf *= np.float32(10.0 ** rng.integers(-5, 1))
        dxb, ddy = dev(xbuf), dev(dybuf)
        gm = torch.empty(fin, fout, device='cuda')
        gb_a, gb_b = torch.empty(fout, device='cuda'), torch.empty(fout, device='cuda')
        L().fc_wgrad(B, fin, fout, dxb.data_ptr(), x_ld, ddy.data_ptr(), y_ld, gm.data_ptr(), gb_a.data_ptr(), ws.ptr, ws.bytes, stream())
        L().adam_step_dev(fin * fout, pa.data_ptr(), gm.data_ptr(), ma.data_ptr(), va.data_ptr(), st_a.data_ptr(), 0, None, None, stream())
        L().adam_advance(st_a.data_ptr(), stream())
        L().fc_wgrad_adam(B, fin, fout, dxb.data_ptr(), x_ld, ddy.data_ptr(), y_ld, pb.data_ptr(), mb.data_ptr(), vb.data_ptr(),
                          gb_b.data_ptr(), st_b.data_ptr(), stream())
        L().adam_advance(st_b.data_ptr(), stream())
        np.testing.assert_array_equal(host(gb_a), host(gb_b))
        np.testing.assert_array_equal(host(ma), host(mb))
        np.testing.assert_array_equal(host(va), host(vb))
        np.testing.assert_array_equal(host(pa), host(pb))
    assert np.abs(host(pb) - p0).max() > 0
    np.testing.assert_allclose(host(st_b)[4:6], [0.9 ** 3, 0.999 ** 3], rtol=1e-6)


@pytest.mark.parametrize("batch", [4, 64])
def test_fused_step_equals_unfused_step(monkeypatch, batch):
    """Graph.train_step with the fc optimiser fused (default) and with MV3D_FUSE_FC_ADAM=0 (bucketed Adam launches behind the
    plain reverse pass): every loss, parameter and Adam slot bit-identical after three steps -- at batch 4 and at the
    benchmarked batch 64 (the fused reverse plan bench.py times, against the plain plan the gradient tests check)."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from tests.synth import appflow_feeds
    feeds = appflow_feeds(np.random.default_rng(3), batch)
    res = []
    for fuse in ('1', '0'):
        monkeypatch.setenv('MV3D_FUSE_FC_ADAM', fuse)
        model = AppearanceFlowModel({'batch_size': batch, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
        g = model.graph
        assert (g.plan_bwd_fused is not None) == (fuse == '1')
        losses = [float(model.train_step(**feeds)) for _ in range(3)]
        torch.cuda.synchronize()
        g.settle()
        res.append((losses, g.params.cpu().numpy().copy(), g.adam_m.cpu().numpy().copy(), g.adam_v.cpu().numpy().copy(), float(g.beta1_power)))
    (l1, p1, m1, v1, b1), (l0, p0, m0, v0, b0) = res
    assert l1 == l0                                         # the scalar loss is a fixed-order sum (elem.hip loss_combine): same bits
    assert b1 == b0
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    np.testing.assert_array_equal(p1, p0)


def test_pipelined_fc_optimiser_equals_joined_step(monkeypatch):
    """The fused fc optimiser of step N runs under the encoder of step N+1 (Graph.pipeline_fc): at the benchmarked batch, four
    steps with fresh feeds leave every parameter and Adam slot bit-identical to the schedule that joins all streams at the end
    of each step (MV3D_PIPELINE_FCADAM=0) -- a missed dependency (a weight read early, a saved input overwritten) would show."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from tests.synth import appflow_feeds
    rng = np.random.default_rng(5)
    feeds = [appflow_feeds(rng, 64) for _ in range(2)]
    res = []
    for pipe in ('1', '0'):
        monkeypatch.setenv('MV3D_PIPELINE_FCADAM', pipe)
        model = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
        g = model.graph
        assert g.pipeline_fc == (pipe == '1') and g.plan_bwd_fused is not None
        assert 0 < g._fwd_wait_idx < g.n_launch_fwd
        for step in range(4):
            model.feed(**feeds[step % 2])
            g.train_step()
            assert g._fc_pending == (pipe == '1')
        torch.cuda.synchronize()
        res.append((g.params.cpu().numpy().copy(), g.adam_m.cpu().numpy().copy(), g.adam_v.cpu().numpy().copy(),
                    g.adam_state.cpu().numpy().copy()))
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[0][3][:8], res[0][3][8:])          # both device Adam records advanced alike


def test_held_fc_optimiser_schedule_equals_joined_step(monkeypatch):
    """MV3D_FC_AFTER_WGRADS=1: the fused fc optimiser launches are left out of the reverse plan's run (MV3D_RUN_HOLD_CLASS2) and issued
    with mv3d_plan_run_side behind the conv filter gradients -- an opt-in schedule, held bit-identical to the joined one."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from tests.synth import appflow_feeds
    feeds = appflow_feeds(np.random.default_rng(6), 8)
    res = []
    for held in ('1', '0'):
        monkeypatch.setenv('MV3D_FC_AFTER_WGRADS', held)
        monkeypatch.setenv('MV3D_PIPELINE_FCADAM', held)
        model = AppearanceFlowModel({'batch_size': 8, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
        g = model.graph
        for _ in range(3):
            model.feed(**feeds)
            g.train_step()
        torch.cuda.synchronize()
        g.settle()
        res.append((g.params.cpu().numpy().copy(), g.adam_m.cpu().numpy().copy(), g.adam_v.cpu().numpy().copy()))
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_wgrad_cu_override_changes_only_the_summation_order():
    """mv3d_set_wgrad_cus: the partial-filter slabs of a conv filter gradient spread over 64 / 256 CUs instead of the default 128 --
    same result up to fp32 summation order, and the setter returns the previous value."""
    case = LC.APPFLOW_B64[1]
    assert L().set_wgrad_cus(64) == 0
    try:
        run_conv_case(case)
        assert L().set_wgrad_cus(256) == 64
        run_conv_case(case)
    finally:
        L().set_wgrad_cus(0)
    assert L().set_wgrad_cus(0) == 0


def test_exact_fp32_rung_at_batch_64():
    """the MV3D_DISABLE=4096 twins (exact fp32 MFMA) of the two layers that carry the step, at the benchmarked batch"""
    old = L().set_diagnostics(4096)
    try:
        run_conv_case(LC.APPFLOW_B64[1])
        run_conv_case(LC.APPFLOW_B64[14])
    finally:
        L().set_diagnostics(old)


def test_round1_kernels_at_batch_64():
    """diagnostics 1048576 | 2097152: the kernels the round-2 ones replaced (bconvu for cconv, wgrad_b3 for cwgrad) stay fallback
    rungs and stay correct at the benchmarked shapes"""
    old = L().set_diagnostics(1048576 | 2097152)
    try:
        run_conv_case(LC.APPFLOW_B64[1])
        run_conv_case(LC.APPFLOW_B64[13])
    finally:
        L().set_diagnostics(old)


def test_small_fc_chain_and_fused_backward_equal_per_layer_calls():
    """The angle MLP (appearance_flow_model.py:101-103: 2 -> 64 -> 64 -> 64, the last output a slice of the [fc1, a2] buffer) as ONE
    forward launch (mv3d_fc_chain_fwd) and one backward launch per layer (mv3d_fc_wgrad_dgrad): bit-identical to the per-layer
    calls they replace, and within the per-op bar of the oracle."""
    B, widths, y_lds = 64, [2, 64, 64, 64], [64, 64, 96]
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, widths[0])).astype(np.float32)
    Ms = [(rng.standard_normal((widths[k], widths[k + 1])) / np.sqrt(widths[k])).astype(np.float32) for k in range(3)]
    bs = [rng.standard_normal(widths[k + 1]).astype(np.float32) for k in range(3)]
    dx_, dMs, dbs = dev(x), [dev(m) for m in Ms], [dev(b) for b in bs]
    ws = Ws(int(L().fc_workspace_bytes(B, 64, 64)))
    # per-layer reference on the device
    ys_ref, cur, cur_ld = [], dx_, widths[0]
    for k in range(3):
        y = torch.full((B, y_lds[k]), 7.0, device='cuda')
        epi = _lib.epilogue(dbs[k].data_ptr(), _lib.ACT_LRELU, 0.2)
        L().fc_fwd(B, widths[k], widths[k + 1], cur.data_ptr(), cur_ld, dMs[k].data_ptr(), y.data_ptr(), y_lds[k], C.byref(epi), ws.ptr, ws.bytes, stream())
        ys_ref.append(y); cur, cur_ld = y, y_lds[k]
    # the chain
    ys = [torch.full((B, y_lds[k]), 7.0, device='cuda') for k in range(3)]
    c = _lib.FcChain()
    c.B, c.nlayers, c.in_, c.x_ld, c.x = B, 3, widths[0], widths[0], dx_.data_ptr()
    for k in range(3):
        c.l[k].M, c.l[k].bias, c.l[k].y, c.l[k].y_ld, c.l[k].out = dMs[k].data_ptr(), dbs[k].data_ptr(), ys[k].data_ptr(), y_lds[k], widths[k + 1]
        c.l[k].act, c.l[k].leak = _lib.ACT_LRELU, 0.2
    L().fc_chain_fwd(C.byref(c), stream())
    h = x
    for k in range(3):
        assert np.array_equal(host(ys[k]), host(ys_ref[k]))
        h = ops.absact_fwd(ops.linear_fwd(h, Ms[k], bs[k]), 'lrelu')
        assert rel_err(host(ys[k])[:, :widths[k + 1]], h) < TOL
    # backward of the middle layer: filter + data gradient in one launch vs the two calls
    dy = rng.standard_normal((B, y_lds[1])).astype(np.float32)
    ddy = dev(dy)
    gm1, gb1, gx1 = (torch.full(sh, float('nan'), device='cuda') for sh in ((64, 64), (64,), (B, y_lds[0])))
    gm2, gb2, gx2 = (torch.full(sh, float('nan'), device='cuda') for sh in ((64, 64), (64,), (B, y_lds[0])))
    epi = _lib.epilogue(None, 0, 0.2, _lib.ACT_LRELU, 0.2, ys[0].data_ptr(), y_lds[0])
    L().fc_wgrad(B, 64, 64, ys[0].data_ptr(), y_lds[0], ddy.data_ptr(), y_lds[1], gm1.data_ptr(), gb1.data_ptr(), ws.ptr, ws.bytes, stream())
    L().fc_dgrad(B, 64, 64, ddy.data_ptr(), y_lds[1], dMs[1].data_ptr(), gx1.data_ptr(), y_lds[0], C.byref(epi), ws.ptr, ws.bytes, stream())
    L().fc_wgrad_dgrad(B, 64, 64, ys[0].data_ptr(), y_lds[0], ddy.data_ptr(), y_lds[1], dMs[1].data_ptr(), gm2.data_ptr(), gb2.data_ptr(),
                       gx2.data_ptr(), y_lds[0], C.byref(epi), ws.ptr, ws.bytes, stream())
    assert np.array_equal(host(gm1), host(gm2)) and np.array_equal(host(gb1), host(gb2)) and np.array_equal(host(gx1), host(gx2))
    rdx, rdm, rdb = ops.linear_bwd(host(ys[0])[:, :64], Ms[1], dy[:, :64])
    assert rel_err(host(gm2), rdm) < TOL and rel_err(host(gb2), rdb) < TOL
    assert rel_err(host(gx2)[:, :64], rdx * _slopes(host(ys[0])[:, :64])) < TOL


def test_split_k_small_image_rung_at_batch_64():
    """diagnostics 16777216: the 8 x 8 / 4 x 4 layers on the kernels sconv replaced (chunks split over workgroups + split-K
    epilogue launch) stay a fallback rung and stay correct at the benchmarked shapes"""
    old = L().set_diagnostics(16777216)
    try:
        for i in (7, 8, 9, 10):          # e3_0 / d4_0, e4, e4_0, d4
            run_conv_case(LC.APPFLOW_B64[i])
    finally:
        L().set_diagnostics(old)


def test_stride2_transposed_conv_rung_at_batch_64():
    """diagnostics 134217728: the stride-2 transposed convolutions d1 / d2 (and the data gradients of e1 / e2) on the fused
    4-phase bconv kernel of round 1, the rung behind sconv4"""
    old = L().set_diagnostics(134217728)
    try:
        run_conv_case(LC.APPFLOW_B64[14])          # d1
        run_conv_case(LC.APPFLOW_B64[12])          # d2
        run_conv_case(LC.APPFLOW_B64[2])           # e1 (its data gradient is the 4-phase direction)
    finally:
        L().set_diagnostics(old)


def test_stride2_forward_conv_rung_at_batch_64():
    """diagnostics 268435456: the stride-2 convolutions e1 / e2 (and the data gradients of d1 / d2) on the unrolled 64-pixel
    kernel (bconvu) and the stride-1 5 x 5 layers on the pipelined kernel (cconv, both tile sizes), the rungs behind s2conv"""
    old = L().set_diagnostics(268435456)
    try:
        for i in (2, 4, 12, 14, 1, 3, 5, 13):          # e1, e2, d2, d1; e0_0 / d1_0, e1_0, e2_0 / d3_0, d2_0
            run_conv_case(LC.APPFLOW_B64[i])
    finally:
        L().set_diagnostics(old)


def test_vector_alu_head_rung():
    """diagnostics 536870912: the 32 -> 1..3-channel stride-2 heads on the tiled vector-ALU kernel (thin_deconv_s2_tile), the rung
    behind the matrix-core form (thin_head)"""
    old = L().set_diagnostics(536870912)
    try:
        run_conv_case(LC.APPFLOW_B64[15])          # flow_field
        run_conv_case(LC.BASEPRED_B128[16])        # dec_image1/d0 (3 channels)
    finally:
        L().set_diagnostics(old)


def test_per_item_thin_input_rung_at_batch_64():
    """diagnostics 33554432 | 67108864: the 3- / 2-channel input layers on the kernels the row-band kernels of thin.hip
    replaced (smallc_b3s / smallc_b3 per-item forward, thin_filtgrad on the vector ALUs) stay fallback rungs and stay correct
    at the benchmarked shapes"""
    old = L().set_diagnostics(33554432 | 67108864)
    try:
        run_conv_case(LC.APPFLOW_B64[0])           # e0
        run_conv_case(LC.APPFLOW_B64[15])          # flow_field (its data gradient is the thin image -> feature direction)
    finally:
        L().set_diagnostics(old)


def test_model_step_at_benchmark_batch():
    """Whole AppearanceFlowModel at batch 64 (BASELINE config 2, what bench.py times): forward outputs, loss and all 47
    gradients against the oracle graph on the same inputs and weights -- the product's own plans (prepared-filter cache,
    side streams), not per-op calls."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from tests.synth import appflow_feeds
    from tests.test_gpu_model import _perturb_biases, _oracle_at_device_kinks, _rel
    B = 64
    model = AppearanceFlowModel({'batch_size': B, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
    g = model.graph
    names = {o[0] for plan in (g.plan_fwd, g.plan_bwd) for o in _lib.plan_ops(plan)}
    assert 'bconv_split_all' in names and any(n.startswith('s2conv<5x5,s1,C32,N32,128px') for n in names)
    variables = _perturb_biases(g)
    feeds = appflow_feeds(np.random.default_rng(3), B)
    builder = omodels.appearance_flow_builder('base')
    out, grads, tape = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds)
    model.feed(**feeds)
    g.run_forward()
    g.run_backward()
    torch.cuda.synchronize()
    out, grads, tape = _oracle_at_device_kinks(model, builder, variables, feeds, out, grads, tape)
    assert _rel(model.flow_field.numpy(), out['flow_field']) < 1e-4
    assert _rel(model.gen.numpy(), out['gen']) < 1e-4
    np.testing.assert_allclose(float(g.loss_buf[0]), float(out['loss']), rtol=1e-5)
    got = g.get_gradients()
    assert set(got) == set(grads)
    worst = max(_rel(got[k], grads[k]) for k in grads)
    assert worst < 1e-3, worst


def test_grad_finalize_equals_per_layer_reduction_and_adam():
    """mv3d_grad_finalize_*: three filter gradients (tiled MFMA kernel, transposed 3x3, thin 3-channel kernel) collected and
    finished by ONE launch -- gradients-only mode bit-identical to the per-layer reduce_slabs launches, optimiser mode
    bit-identical to those followed by mv3d_adam_step_dev over the flat buffer (parameters and both slots), with a plain
    (already final) range in the middle of the buffer covered too."""
    rng = np.random.default_rng(11)
    lib = L()
    cases = [(LC.CONV, 8, 32, 32, 32, 64, 5, 1), (LC.DECONV, 8, 16, 16, 64, 128, 3, 2), (LC.CONV, 8, 64, 64, 3, 32, 5, 2)]
    layers, off = [], 0
    for kind, n, h, w, c, k, ksz, s in cases:
        g = _lib.conv_geom(n, h, w, c, k, ksz, ksz, s, s)
        img = dev(rng.standard_normal((n, h, w, c)).astype(np.float32))
        feat = dev(rng.standard_normal((n, g.Ho, g.Wo, k)).astype(np.float32))
        wn = ksz * ksz * c * k
        lay = dict(kind=kind, g=g, img=img, feat=feat, wn=wn, k=k, w_off=off)
        off += -(-wn // 64) * 64
        if kind == LC.CONV:
            lay['b_off'] = off
            off += -(-k // 64) * 64
        layers.append(lay)
    plain_off, plain_n = off, 192
    off += 256
    flat = off

    def wgrads(grads, ws_of):
        for i, lay in enumerate(layers):
            ws, wsb = ws_of(i, lay)
            gw = grads.data_ptr() + 4 * lay['w_off']
            if lay['kind'] == LC.CONV:
                lib.conv2d_wgrad(C.byref(lay['g']), lay['img'].data_ptr(), lay['feat'].data_ptr(), gw, grads.data_ptr() + 4 * lay['b_off'], ws, wsb, stream())
            else:
                lib.deconv2d_wgrad(C.byref(lay['g']), lay['feat'].data_ptr(), lay['img'].data_ptr(), gw, ws, wsb, stream())

    plain = rng.standard_normal(plain_n).astype(np.float32)
    ref = torch.zeros(flat, device='cuda')
    ref[plain_off:plain_off + plain_n] = dev(plain)
    shared = [conv_ws(lay['g']) for lay in layers]
    wgrads(ref, lambda i, lay: (shared[i].ptr, shared[i].bytes))
    torch.cuda.synchronize()
    assert float(ref.abs().sum()) > 0

    own = [Ws(max(int(lib.conv_wgrad_workspace_bytes(C.byref(lay['g']))), 16)) for lay in layers]
    assert sum(int(lib.conv_wgrad_workspace_bytes(C.byref(lay['g']))) > 0 for lay in layers) >= 2      # the case really has slabs
    # gradients only
    got = torch.zeros(flat, device='cuda')
    got[plain_off:plain_off + plain_n] = dev(plain)
    lib.grad_finalize_begin()
    wgrads(got, lambda i, lay: (own[i].ptr, own[i].bytes))
    tb = int(lib.grad_finalize_table_bytes())
    table = torch.empty(tb, dtype=torch.uint8, device='cuda')
    lib.grad_finalize_commit(table.data_ptr(), tb, None, None, None, None, None, stream())
    np.testing.assert_array_equal(host(got), host(ref))
    # with the optimiser: two steps with a changing bias correction
    state = np.array([1e-3, 0.9, 0.999, 1e-8, 0.9, 0.999, 1.0, 0.0], np.float32)
    p0 = rng.standard_normal(flat).astype(np.float32)
    pa, ma, va, sa = dev(p0), torch.zeros(flat, device='cuda'), torch.zeros(flat, device='cuda'), dev(state)
    pb, mb, vb, sb = dev(p0), torch.zeros(flat, device='cuda'), torch.zeros(flat, device='cuda'), dev(state)
    for step in range(2):
        lib.adam_step_dev(flat, pa.data_ptr(), ref.data_ptr(), ma.data_ptr(), va.data_ptr(), sa.data_ptr(), 0, None, None, stream())
        lib.adam_advance(sa.data_ptr(), stream())
        gbuf = torch.zeros(flat, device='cuda')
        gbuf[plain_off:plain_off + plain_n] = dev(plain)
        lib.grad_finalize_begin()
        wgrads(gbuf, lambda i, lay: (own[i].ptr, own[i].bytes))
        lib.grad_finalize_add(gbuf.data_ptr() + 4 * plain_off, plain_n)
        tb = int(lib.grad_finalize_table_bytes())
        table = torch.empty(tb, dtype=torch.uint8, device='cuda')
        lib.grad_finalize_commit(table.data_ptr(), tb, gbuf.data_ptr(), pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), sb.data_ptr(), stream())
        lib.adam_advance(sb.data_ptr(), stream())
        torch.cuda.synchronize()
    # the reference optimiser ran over the whole flat buffer (zero gradients in the padding move nothing); compare everywhere
    np.testing.assert_array_equal(host(ma), host(mb))
    np.testing.assert_array_equal(host(va), host(vb))
    np.testing.assert_array_equal(host(pa), host(pb))
    assert np.abs(host(pb) - p0).max() > 0
