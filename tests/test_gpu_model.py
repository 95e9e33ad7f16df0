"""Whole-model parity on the GPU: the HIP graph (through the reference-named model classes)
against the numpy oracle on the same seeded inputs and identical weights.
Bar from north_star: outputs within 1e-3 relative fp32; measured agreement is ~1e-5."""
import os

import numpy as np
import pytest
import torch

from dynamic_multiview_3d_amd import _lib
from oracle import models as omodels
from tests.synth import appflow_feeds

pytestmark = pytest.mark.gpu


def _perturb_biases(g, seed=5):
    rng = np.random.default_rng(seed)
    vals = g.get_variables()
    for k, v in vals.items():
        if k.endswith('/b'):
            vals[k] = v + rng.normal(0, 0.05, v.shape).astype(np.float32)
    g.set_variables(vals)
    return vals


def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def _activation_pattern_override(model, tape):
    """lrelu'/relu' jump at 0: a pre-activation that is zero to within summation-order noise can get
    either slope.  For exactly those elements (|pre| <= 1e-5 of the layer's max) the oracle's reverse
    pass is told to use the device's activation sign; everywhere else the signs must agree."""
    from dynamic_multiview_3d_amd.graph import ConvNode, LinearNode
    acts = [n for n in model.graph.nodes if isinstance(n, (ConvNode, LinearNode)) and n.act in (1, 2)]
    assert len(acts) == len(tape.act_inputs)
    override, flips, total = [], 0, 0
    for n, pre in zip(acts, tape.act_inputs):
        total += pre.size
        out = n.y.value().detach().cpu().numpy().reshape(pre.shape)
        if n.act == 2:      # relu keeps -0.0 for negative inputs
            dev_sign = np.where(out > 0, 1.0, np.where(np.signbit(out), -1.0, 0.0))
        else:
            dev_sign = np.sign(out)
        diff = dev_sign != np.sign(pre)
        if diff.any():
            assert np.abs(pre[diff]).max() <= 1e-5 * np.abs(pre).max(), "activation sign differs away from zero"
            flips += int(diff.sum())
            override.append(np.where(diff, dev_sign, np.sign(pre)))
        else:
            override.append(None)
    # how many elements may sit that close to zero: summation-order / split-bf16 noise is a few 1e-6 of the
    # layer maximum, the density of pre-activations at 0 is ~0.4/sigma, max ~ 5 sigma => ~1e-5 of the elements
    assert flips <= 8 + 2e-5 * total, (flips, total)
    return override, flips


def _sampling_cell_override(model, tape):
    """The bilinear sampler's gradient w.r.t. a coordinate jumps where the coordinate crosses an integer.  For
    samples whose cell (floor) differs between device and oracle -- the coordinates themselves must agree to
    1e-4 of the coordinate range -- the oracle is re-evaluated at the device's coordinates."""
    from dynamic_multiview_3d_amd.graph import ResampleNode
    rs = [n for n in model.graph.nodes if isinstance(n, ResampleNode)]
    assert len(rs) == len(tape.warp_inputs)
    override, moved = [], 0
    for n, w in zip(rs, tape.warp_inputs):
        dev = n.warp.value().detach().cpu().numpy().reshape(w.shape)
        assert np.abs(dev - w).max() <= 1e-4 * max(np.abs(w).max(), 1.0)
        diff = (np.floor(dev) != np.floor(w)).any(axis=-1, keepdims=True)
        moved += int(diff.sum())
        override.append(np.where(diff, dev, w) if diff.any() else None)
    assert moved <= 4 + 1e-3 * sum(w[..., 0].size for w in tape.warp_inputs), moved
    return override, moved


def _oracle_at_device_kinks(model, builder, variables, feeds, out, grads, tape):
    """Re-run the oracle with the device's decisions at the kinks of lrelu / relu / floor (see above)."""
    override, flips = _activation_pattern_override(model, tape)
    woverride, moved = _sampling_cell_override(model, tape)
    if flips or moved:
        out, grads, tape = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds,
                                       sign_override=override, warp_override=woverride)
    return out, grads, tape


def _check_model(cls, variant, dead=()):
    conf = {'batch_size': 2, 'learning_rate': 1e-4}
    model = cls(conf, load_tfrec=False, build_loss=True, device='cuda')
    g = model.graph
    variables = _perturb_biases(g)
    feeds = appflow_feeds(np.random.default_rng(3), 2)
    builder = omodels.appearance_flow_builder(variant)
    ov = {k: v.copy() for k, v in variables.items()}
    out, grads, tape = omodels.run(builder, ov, feeds)
    assert list(variables.keys()) == tape.used                     # TF variable creation order

    model.feed(**feeds)
    g.run_forward()
    g.run_backward()
    torch.cuda.synchronize()
    out, grads, tape = _oracle_at_device_kinks(model, builder, variables, feeds, out, grads, tape)
    assert _rel(model.flow_field.numpy(), out['flow_field']) < 1e-4
    assert _rel(model.warp_pts.numpy(), out['warp_pts']) < 1e-5
    assert _rel(model.gen.numpy(), out['gen']) < 1e-4
    np.testing.assert_allclose(float(g.loss_buf[0]), float(out['loss']), rtol=1e-5)
    got = g.get_gradients()
    assert set(variables) - set(got) == set(dead)
    assert set(got) == set(grads)
    worst = max(_rel(got[k], grads[k]) for k in grads)
    assert worst < 1e-3, worst
    return model, variables, feeds, builder


def test_appearance_flow_model_forward_backward():
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    model, variables, feeds, builder = _check_model(AppearanceFlowModel, 'base')
    assert sum(v.size for v in model.graph.variables.values()) == 69535232
    # three Adam steps: loss sequence and updated weights
    g = model.graph
    ov = {k: v.copy() for k, v in variables.items()}
    adam = omodels.AdamState(1e-4)
    ref_losses = [omodels.step(builder, ov, adam, feeds)[0] for _ in range(3)]
    losses = [float(model.train_step()) for _ in range(3)]
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4)
    got = g.get_variables()
    # Adam's first steps move every weight by ~lr * g/|g|: a sign function of tiny gradients, so the
    # element-wise update is not a stable quantity (bit-exactness of the Adam kernel itself is pinned in
    # test_gpu_ops.py::test_adam_bit_exact_vs_oracle).  Compare the DIRECTION of the 3-step update.
    for k in ('e0/w', 'fc1/Matrix', 'a3/Matrix', 'd1_0/w', 'flow_field/w', 'a5/b'):
        upd_ref = (ov[k] - variables[k]).ravel().astype(np.float64)
        upd = (got[k] - variables[k]).ravel().astype(np.float64)
        cos = float(upd @ upd_ref / (np.linalg.norm(upd) * np.linalg.norm(upd_ref)))
        assert cos > 0.98, (k, cos)
        assert abs(np.abs(upd).max() / np.abs(upd_ref).max() - 1) < 0.05, k


def test_highdim_lowdim_tinghui_variants():
    from dynamic_multiview_3d_amd.highdim_angle import AppFlowHighDimAngle
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from dynamic_multiview_3d_amd.appearance_flow_tinghui import AppearanceFlowTinghui
    _check_model(AppFlowHighDimAngle, 'highdim', dead=('a0/Matrix', 'a0/b', 'a1/Matrix', 'a1/b'))
    _check_model(AppFlowLowDimAngle, 'lowdim')
    _check_model(AppearanceFlowTinghui, 'tinghui')


def test_zero_flow_head_reproduces_transposed_input():
    """Known answer through the whole model: with flow_field/w = 0 the network output is the
    transposed source image, bit-exact."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    model = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cuda')
    model.graph.set_variables({'flow_field/w': np.zeros((5, 5, 2, 32), np.float32)})
    feeds = appflow_feeds(np.random.default_rng(9), 2)
    model.forward(**feeds)
    np.testing.assert_array_equal(model.gen.numpy(), feeds['image0'].transpose(0, 2, 1, 3))


def _check_generic(model, builder, feeds, out_names, out_tol=1e-4):
    """forward outputs, loss and every variable gradient of a model vs the oracle graph."""
    g = model.graph
    variables = _perturb_biases(g)
    out, grads, tape = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds)
    assert list(variables.keys()) == tape.used
    model.feed(**feeds)
    g.run_forward()
    g.run_backward()
    torch.cuda.synchronize()
    out, grads, tape = _oracle_at_device_kinks(model, builder, variables, feeds, out, grads, tape)
    errs = {attr: _rel(getattr(model, attr).numpy(), out[key]) for attr, key in out_names.items()}
    for attr, e in errs.items():
        assert e < out_tol, (attr, e)
    np.testing.assert_allclose(float(g.loss_buf[0]), float(out['loss']), rtol=2e-5)
    got = g.get_gradients()
    assert set(got) == set(grads)
    worst = max(_rel(got[k], grads[k]) for k in grads)
    assert worst < 1e-3, worst
    return errs


def test_base_prediction_model_color_and_depth():
    """SURVEY 8a row a11 (tensorflowdata/cars_colordepth): RGB + depth towers, two tanh decoders."""
    from dynamic_multiview_3d_amd.main_model import Base_Prediction_Model
    conf = {'batch_size': 2, 'learning_rate': 1e-4, 'use_color': '', 'use_depth': '', 'depth_lr_factor': 0.1}
    model = Base_Prediction_Model(conf, load_tfrec=False, device='cuda')
    assert sum(v.size for v in model.graph.variables.values()) == 70049984
    f = appflow_feeds(np.random.default_rng(4), 2)
    f['dimage0'] = f['image0'][..., :1].copy()
    f['dimage1'] = f['image1'][..., :1].copy()
    _check_generic(model, omodels.base_prediction_builder(conf), f, {'gen_image1': 'gen_image1', 'gen_dimage1': 'gen_dimage1'})


def test_base_prediction_model_at_batch_128():
    """BASELINE config 3 as a GRAPH at its batch (the B = 128 graph had only been plan-recorded on the CPU;
    tests/test_gpu_layers.py covers its layers one by one): forward outputs, loss and every gradient of the recorded plans
    against the oracle graph on the same 128 images, then five train steps on that batch -- finite, and the loss falls."""
    from dynamic_multiview_3d_amd.main_model import Base_Prediction_Model
    B = 128
    conf = {'batch_size': B, 'learning_rate': 1e-4, 'use_color': '', 'use_depth': '', 'depth_lr_factor': 0.1}
    model = Base_Prediction_Model(conf, load_tfrec=False, device='cuda')
    f = appflow_feeds(np.random.default_rng(4), B)
    f['dimage0'] = f['image0'][..., :1].copy()
    f['dimage1'] = f['image1'][..., :1].copy()
    _check_generic(model, omodels.base_prediction_builder(conf), f, {'gen_image1': 'gen_image1', 'gen_dimage1': 'gen_dimage1'})
    losses = [float(model.train_step(**f))] + [float(model.train_step()) for _ in range(5)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert np.isfinite(model.gen_image1.numpy()).all() and np.isfinite(model.gen_dimage1.numpy()).all()


@pytest.mark.parametrize("extra", [
    {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'fully_conv': ''},
    {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'predict_target_masks': 0.5},
    {'use_color': '', 'gen_sep_images': '', 'masked_image_loss': '', 'fully_conv': ''},
])
def test_multiobject_appflow(extra):
    """fc and fully-convolutional bottlenecks, flow + direct decoders, masked loss."""
    from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow
    from tests.synth import multiobj_feeds
    conf = dict(extra, batch_size=2, learning_rate=1e-4)
    model = MultiObjectAppFlow(conf, load_tfrec=False, device='cuda')
    f = multiobj_feeds(np.random.default_rng(5), 2)
    names = {a: a for a in ('gen_image1', 'gen_image1_only0', 'gen_image1_only1', 'gen_depth1', 'gen_depth1_only0',
                            'gen_depth1_only1', 'gen_image1_mask0', 'gen_image1_mask1') if getattr(model, a) is not None}
    _check_generic(model, omodels.multiobject_builder(conf), f, names)


def test_multiobject_appflow_256x256():
    """BASELINE config 5 (SURVEY 8a note 2: every spatial constant of multiobject_appflow.py:21-22,93-100,155 doubled, fully
    convolutional bottleneck): all outputs, loss and every gradient vs the oracle graph at 256 x 256, batch 2."""
    from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow
    from tests.synth import multiobj_feeds
    from tests.layer_cases import MULTIOBJ_256_CONF
    conf = dict(MULTIOBJ_256_CONF, batch_size=2, learning_rate=1e-4)
    model = MultiObjectAppFlow(conf, load_tfrec=False, device='cuda')
    assert model.gen_image1.shape == (2, 256, 256, 3) and model.graph.variables['e4_1/w'].shape == (3, 3, 320, 256)
    f = multiobj_feeds(np.random.default_rng(5), 2, 256)
    names = {a: a for a in ('gen_image1', 'gen_image1_only0', 'gen_image1_only1', 'gen_depth1', 'gen_depth1_only0',
                            'gen_depth1_only1') if getattr(model, a) is not None}
    assert len(names) == 6
    # sampled outputs: a flow error of 1e-5 pixel times the contrast of the noisy 256 x 256 source (measured 1.5e-4 of the image
    # maximum, against the 1e-3 bar of north_star; the 128 x 128 graphs hold 1e-4).  That this is operand rounding and not
    # indexing is checked, not assumed: the same graph on the exact-fp32 rung (mv3d_set_diagnostics(4096): fp32 MFMA, same tiles,
    # same index arithmetic) must hold the 128 x 128 bar of 1e-4 on every output and be no worse than the split-bf16 kernels.
    errs = _check_generic(model, omodels.multiobject_builder(conf), f, names, out_tol=5e-4)
    old = _lib.lib().set_diagnostics(4096)
    try:
        exact = MultiObjectAppFlow(conf, load_tfrec=False, device='cuda')
        errs_exact = _check_generic(exact, omodels.multiobject_builder(conf), f, names, out_tol=1e-4)
    finally:
        _lib.lib().set_diagnostics(old)
    print('256 x 256 output errors: split-bf16 %s  exact fp32 %s' % ({k: '%.1e' % v for k, v in errs.items()}, {k: '%.1e' % v for k, v in errs_exact.items()}))
    assert max(errs_exact.values()) <= max(max(errs.values()), 2e-5)


def test_multiobject_appflow_256x256_full_batch_properties():
    """The same configuration at its full per-GPU batch (32 x 256 x 256: too large for the oracle in a test) through properties
    that do not depend on the size: finite outputs, a loss that decreases over five Adam steps on a fixed batch, and the
    known answer of the sampler -- with every flow head zeroed the flow decoders return the transposed
    source image bit for bit."""
    from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow
    from tests.synth import multiobj_feeds
    from tests.layer_cases import MULTIOBJ_256_CONF
    conf = dict(MULTIOBJ_256_CONF, batch_size=32, learning_rate=1e-4)
    model = MultiObjectAppFlow(conf, load_tfrec=False, device='cuda')
    g = model.graph
    f = multiobj_feeds(np.random.default_rng(6), 32, 256)
    losses = [float(model.train_step(**f))] + [float(model.train_step()) for _ in range(5)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    for a in ('gen_image1', 'gen_depth1', 'gen_image1_only0'):
        assert np.isfinite(getattr(model, a).numpy()).all()
    heads = {k: np.zeros(v.shape, np.float32) for k, v in g.variables.items() if k.endswith('/d0/w') and v.shape[2] == 2}
    assert len(heads) == 3                                             # gen_image1, gen_image1_only0, gen_image1_only1
    g.set_variables(heads)
    model.forward(**f)
    np.testing.assert_array_equal(model.gen_image1.numpy(), f['image0'].transpose(0, 2, 1, 3))


@pytest.mark.parametrize("extra", [
    {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'masked_image_loss': '', 'fully_conv': ''},
    {'use_color': '', 'combination_image': '', 'predict_target_masks': 0.5},
])
def test_multiobject_main_model(extra):
    """multiobject_main_model.Base_Prediction_Model -- every output from a direct tanh decoder."""
    from dynamic_multiview_3d_amd.multiobject_main_model import Base_Prediction_Model
    from tests.synth import multiobj_feeds
    conf = dict(extra, batch_size=2, learning_rate=1e-4)
    model = Base_Prediction_Model(conf, load_tfrec=False, device='cuda')
    assert model.graph.variables['dec_image1/d0/w'].shape == (5, 5, 3, 32)
    f = multiobj_feeds(np.random.default_rng(5), 2)
    names = {a: a for a in ('gen_image1', 'gen_image1_only0', 'gen_image1_only1', 'gen_depth1', 'gen_depth1_only0',
                            'gen_depth1_only1', 'gen_image1_mask0', 'gen_image1_mask1') if getattr(model, a) is not None}
    _check_generic(model, omodels.multiobject_builder(conf, direct_color=True), f, names)


def test_train_driver_runs_saves_and_resumes(tmp_path):
    """the train.py loop (inclusive iteration range, final checkpoint, resume
    iteration parsed from the file name) on a tiny batch."""
    from dynamic_multiview_3d_amd import train
    conf_py = tmp_path / 'conf.py'
    conf_py.write_text(
        "import os\nfrom lowdim_angle import AppFlowLowDimAngle\n"
        "configuration = {'experiment_name': 't', 'data_dir': '', 'output_dir': os.path.dirname(os.path.realpath(__file__)) + '/modeldata',\n"
        "  'num_iterations': 30, 'batch_size': 2, 'learning_rate': 1e-4, 'train_val_split': 0.95, 'model': AppFlowLowDimAngle}\n")
    model = train.main(['--hyper', str(conf_py)])
    out = tmp_path / 'modeldata'
    assert (out / 'model.index').exists() and (out / 'model.data-00000-of-00001').exists() and (out / 'checkpoint').exists()
    import json
    rows = [json.loads(l) for l in open(out / 'train_log.jsonl')]
    its = [r['itr'] for r in rows if 'training_loss' in r]
    assert its == [0, 10, 20, 30]                                                  # range(itr_0, num_iterations + 1)
    losses = [r['training_loss'] for r in rows if 'training_loss' in r]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    from dynamic_multiview_3d_amd import tf_checkpoint
    sd = tf_checkpoint.read_checkpoint(str(out / 'model'))
    assert 'a0/Matrix/Adam_1' in sd and abs(float(sd['beta1_power']) - 0.9 ** 32) < 1e-6      # 31 steps taken
    assert tf_checkpoint.get_checkpoint_state(str(out))['model_checkpoint_path'] == str(out / 'model')
    for ext in ('.index', '.data-00000-of-00001'):
        os.replace(str(out / 'model') + ext, str(out / 'model30') + ext)
    model2 = train.main(['--hyper', str(conf_py), '--pretrained', str(out / 'model30'), '--num_iterations', '33'])
    rows = [json.loads(l) for l in open(out / 'train_log.jsonl')]
    assert [r['itr'] for r in rows if 'training_loss' in r][-1] == 30              # resumed at 30: logs itr 30 again


def test_train_driver_reads_tfrecord_shards(tmp_path):
    """conf['data_dir'] with TFRecord shards in the reference's format feeds the train loop
    (background reader thread, pinned upload on its own stream); the first batch the model sees is the first two
    records of the only training file."""
    from dynamic_multiview_3d_amd import train, read_tf_records as R
    data = tmp_path / 'data'
    data.mkdir()
    rng = np.random.default_rng(0)
    first = []
    for f in range(2):
        with R.TFRecordWriter(str(data / ('%d.tfrecords' % f))) as w:
            for i in range(6):
                img0 = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
                if f == 0 and i < 2:
                    first.append(img0)
                w.write(R.serialize_example({'image0': img0.tobytes(), 'image1': rng.integers(0, 256, (128, 128, 3), dtype=np.uint8).tobytes(),
                                             'depth0': bytes(128 * 128), 'depth1': bytes(128 * 128),
                                             'displacement': rng.uniform(-1, 1, 2).astype(np.float32)}))
    conf_py = tmp_path / 'conf.py'
    conf_py.write_text(
        "import os\nfrom appearance_flow_model import AppearanceFlowModel\n"
        "configuration = {'experiment_name': 't', 'data_dir': %r, 'output_dir': os.path.dirname(os.path.realpath(__file__)) + '/modeldata',\n"
        "  'num_iterations': 12, 'batch_size': 2, 'learning_rate': 1e-4, 'train_val_split': 0.5, 'model': AppearanceFlowModel}\n" % str(data))
    model = train.main(['--hyper', str(conf_py)])
    import json
    rows = [json.loads(l) for l in open(tmp_path / 'modeldata' / 'train_log.jsonl')]
    losses = [r['training_loss'] for r in rows if 'training_loss' in r]
    assert len(losses) == 2 and all(np.isfinite(losses))
    # 13 steps x 2 records over a 6-record training file: the 13th batch wraps to records 0, 1 of file 0 again
    np.testing.assert_array_equal(model.image0.numpy(), np.stack(first).astype(np.float32) / np.float32(255))


@pytest.mark.parametrize("variant", ['nobg_nodm', 'nobg_dm', 'bg_nodm'])
def test_mv3d_models(variant):
    """mv3d direct-prediction networks -- forward image, loss and every gradient vs the oracle
    (channel-slice losses, masked colour loss and the 0.75 target scale run inside mv3d_pixel_loss_strided)."""
    from dynamic_multiview_3d_amd import mv3d
    from tests.test_oracle_models import mv3d_feeds
    cls = {'nobg_nodm': mv3d.mv3d_nobg_nodm, 'nobg_dm': mv3d.mv3d_nobg_dm, 'bg_nodm': mv3d.mv3d_bg_nodm}[variant]
    model = cls({'batch_size': 2}, device='cuda')
    f = mv3d_feeds(np.random.default_rng(8), 2, variant)
    _check_generic(model, omodels.mv3d_builder(variant), f, {'gen': 'gen'})
    l0 = float(model.train_step(**f))
    for _ in range(5):
        l1 = float(model.train_step())
    assert np.isfinite(l1) and l1 < l0


def test_kink_flips_are_rounding_not_indexing():
    """The activation-sign override (above) tolerates 8 + 2e-5 * elements flipped signs.  If those flips were an indexing
    slip rather than rounding at pre-activations of ~0, the exact-fp32 rung (mv3d_set_diagnostics(4096): an fma chain that
    differs from BLAS only in summation order) would flip as many; it must flip no more than the split-bf16 kernels do,
    every flipped element must sit within 1e-5 of the layer maximum of zero (asserted inside the override), and with the
    flips overridden both rungs must agree with the oracle.  Model: the multi-object graph that showed 20 flips in round 1."""
    from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow
    from tests.synth import multiobj_feeds
    conf = {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'fully_conv': '',
            'batch_size': 2, 'learning_rate': 1e-4}
    f = multiobj_feeds(np.random.default_rng(5), 2)
    builder = omodels.multiobject_builder(conf)
    counts = {}
    for rung, mask in (('split_bf16', 0), ('exact_fp32', 4096)):
        old = _lib.lib().set_diagnostics(mask)
        try:
            model = MultiObjectAppFlow(conf, load_tfrec=False, device='cuda')
            g = model.graph
            variables = _perturb_biases(g)
            out, grads, tape = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, f)
            model.feed(**f)
            g.run_forward()
            g.run_backward()
            torch.cuda.synchronize()
            _, flips = _activation_pattern_override(model, tape)
            counts[rung] = flips
            out, grads, tape = _oracle_at_device_kinks(model, builder, variables, f, out, grads, tape)
            got = g.get_gradients()
            worst = max(_rel(got[k], grads[k]) for k in grads)
            assert worst < (1e-3 if mask == 0 else 5e-5), (rung, worst)
        finally:
            _lib.lib().set_diagnostics(old)
    print("flipped activation signs:", counts)
    assert counts['exact_fp32'] <= max(counts['split_bf16'], 8), counts


@pytest.mark.gpu
def test_fused_head_equals_the_three_launch_head(monkeypatch):
    """Graph._fuse_resample_losses: the appearance-flow head as one launch (sampler + loss + sampler gradient) and as three
    (MV3D_FUSE_RESAMPLE=0) give the same outputs, loss and parameter gradients; and the exact-fp32 rungs
    (mv3d_set_diagnostics(4096)) stay within 1e-5 of the oracle on the whole model."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from dynamic_multiview_3d_amd.graph import ResampleNode
    feeds = appflow_feeds(np.random.default_rng(3), 2)
    res = []
    for fuse in ('1', '0'):
        monkeypatch.setenv('MV3D_FUSE_RESAMPLE', fuse)
        model = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
        g = model.graph
        _perturb_biases(g)
        node = [n for n in g.nodes if isinstance(n, ResampleNode)][0]
        assert (node.fused_loss is not None) == (fuse == '1')
        names = [o[0] for o in _lib.plan_ops(g.plan_fwd)]
        assert ('resample_loss' in names) == (fuse == '1') and ('pixel_loss' in names) == (fuse == '0')
        model.feed(**feeds)
        g.run_forward()
        g.run_backward()
        torch.cuda.synchronize()
        res.append((model.gen.numpy(), float(g.loss_buf[0]), g.get_gradients()))
    (gen1, loss1, gr1), (gen0, loss0, gr0) = res
    np.testing.assert_array_equal(gen1, gen0)
    np.testing.assert_allclose(loss1, loss0, rtol=2e-6)
    for k in gr0:
        assert _rel(gr1[k], gr0[k]) < 1e-6, k

    old = _lib.lib().set_diagnostics(4096)
    try:
        monkeypatch.setenv('MV3D_FUSE_RESAMPLE', '1')
        model = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda')
        g = model.graph
        variables = _perturb_biases(g)
        builder = omodels.appearance_flow_builder('base')
        out, grads, tape = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds)
        model.feed(**feeds)
        g.run_forward()
        g.run_backward()
        torch.cuda.synchronize()
        out, grads, tape = _oracle_at_device_kinks(model, builder, variables, feeds, out, grads, tape)
        got = g.get_gradients()
        worst = max(_rel(got[k], grads[k]) for k in grads)
        assert worst < 2e-5, worst
        assert _rel(model.gen.numpy(), out['gen']) < 1e-5
    finally:
        _lib.lib().set_diagnostics(old)


@pytest.mark.gpu
def test_graft_entry_smoke_runs():
    """the driver's round-end check (`__graft_entry__.smoke()`) stays green with the rest of the suite"""
    import __graft_entry__
    __graft_entry__.smoke()
