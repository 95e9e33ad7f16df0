"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of
include/mv3d_hip.h, argument validation works without a device, the model classes build and
record their launch plans, and the host bookkeeping (variable order, views, fusion) is right."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dynamic_multiview_3d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from dynamic_multiview_3d_amd import build
        build.build()
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'mv3d_hip.h')).read()
    declared = set(re.findall(r'\b(mv3d_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'mv3d_plan'}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib.dll, name), "libmv3d_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b'gfx950' in lib.version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.ConvGeom) == 14 * 4
    assert _lib.Epilogue.bias.offset == 0 and _lib.Epilogue.act.offset == 8
    assert _lib.Epilogue.gmask_ref.offset == 24 and C.sizeof(_lib.Epilogue) == 40
    # mv3d_fc_chain_layer: three pointers + four 32-bit fields; mv3d_fc_chain: four 32-bit fields, a pointer, four layers
    assert C.sizeof(_lib.FcChainLayer) == 40 and _lib.FcChainLayer.y_ld.offset == 24 and _lib.FcChainLayer.leak.offset == 36
    assert _lib.FcChain.x.offset == 16 and _lib.FcChain.l.offset == 24 and C.sizeof(_lib.FcChain) == 24 + 4 * 40


def test_small_fc_chains_are_found_and_can_be_switched_off(lib, monkeypatch):
    """The angle MLP a0 -> a1 -> a2 (appearance_flow_model.py:101-103) is one forward launch, recorded at the position of a2 (the
    launch that writes into the [fc1, a2] buffer: the fc hazard of the pipelined optimiser points at or in front of it); its data
    gradients ride the filter-gradient stream with the filter gradients (one launch per layer); MV3D_FC_CHAINS=0 keeps one
    launch per layer."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    m = AppearanceFlowModel({'batch_size': 8, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cpu')
    g = m.graph
    fwd = [o[0] for o in _lib.plan_ops(g.plan_fwd)]
    assert fwd.count('small_fc_chain_fwd') == 1 and 'small_fc_fwd' not in fwd
    chain = [n for n in g.nodes if getattr(n, 'chain', None)]
    assert [n.m.name.split('/')[0] for n in chain] == ['a0', 'a1', 'a2'] and chain[0].chain[-1] is chain[2]
    assert g._fwd_wait_idx is not None and g._fwd_wait_idx <= fwd.index('small_fc_chain_fwd')
    bwd = [o[0] for o in _lib.plan_ops(g.plan_bwd_fused)]
    assert bwd.count('small_fc_bwd') == 2 and bwd.count('small_fc_wgrad') == 1 and 'small_fc_dgrad' not in bwd
    monkeypatch.setenv('MV3D_FC_CHAINS', '0')
    m2 = AppearanceFlowModel({'batch_size': 8, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cpu')
    fwd2 = [o[0] for o in _lib.plan_ops(m2.graph.plan_fwd)]
    assert fwd2.count('small_fc_fwd') == 3 and 'small_fc_chain_fwd' not in fwd2
    assert len(fwd2) == len(fwd) + 2


def test_validation_without_device(lib):
    g = _lib.conv_geom(1, 8, 8, 4, 4, 3, 3, 3, 3)
    assert lib.raw_conv2d_fwd(C.byref(g), 1, 1, 1, None, None, 0, None) == -4        # MV3D_E_UNSUPPORTED
    assert 'stride' in lib.last_error()
    g = _lib.conv_geom(1, 8, 8, 4, 4, 3, 3, 1, 1)
    g.Ho = 7
    assert lib.raw_conv2d_fwd(C.byref(g), 1, 1, 1, None, None, 0, None) == -1        # MV3D_E_INVAL
    assert lib.raw_adam_step(0, 1, 1, 1, 1, 1e-4, .9, .999, 1e-8, .9, .999, 1.0, None) == -1
    assert lib.raw_plan_end() == -1


def test_plan_records_launch_metadata(lib):
    plan = lib.plan_create()
    lib.plan_begin(plan)
    try:
        g = _lib.conv_geom(2, 64, 64, 32, 32, 5, 5, 1, 1)
        epi = _lib.epilogue()
        lib.conv2d_fwd(C.byref(g), 0x1000, 0x2000, 0x3000, C.byref(epi), None, 0, None)
        lib.adam_step(1024, 0x1000, 0x2000, 0x3000, 0x4000, 1e-4, .9, .999, 1e-8, .9, .999, 1.0, None)
    finally:
        lib.plan_end()
    ops = _lib.plan_ops(plan)
    assert [o[0] for o in ops][-1] == 'adam' and ops[0][0].startswith('hconv')
    assert ops[0][1] == 2.0 * 2 * 64 * 64 * 25 * 32 * 32                  # algorithmic FLOPs of the conv
    assert ops[1][2] == 28.0 * 1024                                       # Adam: 28 B/param
    lib.plan_destroy(plan)


@pytest.mark.parametrize("modname,clsname,nvars,dead", [
    ("appearance_flow_model", "AppearanceFlowModel", 47, 0),
    ("highdim_angle", "AppFlowHighDimAngle", 47, 4),
    ("lowdim_angle", "AppFlowLowDimAngle", 43, 0),
    ("appearance_flow_tinghui", "AppearanceFlowTinghui", 29, 0),
])
def test_models_build_and_record_plans_on_cpu(lib, modname, clsname, nvars, dead):
    import importlib
    cls = getattr(importlib.import_module('dynamic_multiview_3d_amd.' + modname), clsname)
    m = cls({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cpu')
    g = m.graph
    assert len(g.variables) == nvars
    assert sum(1 for v in g.variables.values() if not v.has_grad) == dead
    assert g.n_launch_fwd > 15 and g.n_launch_bwd > 30
    assert m.gen.shape == (2, 128, 128, 3) and m.flow_field.shape == (2, 128, 128, 2)
    # variable names / order follow the TF scopes of the reference
    names = list(g.variables)
    assert names[0] == 'e0/w' and names[1] == 'e0/b' and names[-1] == 'flow_field/w'
    # initialisers: biases zero, conv weights truncated at 2 sigma (tf_utils.py:74-80)
    assert float(g.variables['e0/b'].value().abs().max()) == 0.0
    w = g.variables['e0/w'].value().numpy()
    std = np.sqrt(2.0 / (w.shape[0] * w.shape[1] * 3))
    assert np.abs(w).max() <= 2 * std + 1e-6 and 0.7 * std < w.std() < 1.0 * std


@pytest.mark.parametrize("clsname,nvars,nout,last", [("mv3d_nobg_nodm", 47, 3, 'd0/w'), ("mv3d_nobg_dm", 47, 4, 'd0/w'), ("mv3d_bg_nodm", 49, 4, 'd0_1/b')])
def test_mv3d_models_build_on_cpu(lib, clsname, nvars, nout, last):
    """the mv3d classes (mv3d/nobg_nodm.py, nobg_dm.py, bg_nodm.py) build and record their plans;
    the tf.slice pairs are views, the loss terms run on channel slices."""
    from dynamic_multiview_3d_amd import mv3d
    m = getattr(mv3d, clsname)({'batch_size': 2}, device='cpu')
    g = m.graph
    assert len(g.variables) == nvars and list(g.variables)[-1] == last
    assert m.gen.shape == (2, 128, 128, nout) and m.images2.shape == (2, 128, 128, nout) and m.labels.shape == (2, 5)
    assert all(v.has_grad for v in g.variables.values())
    nterms = len(g.loss_expr.terms)
    assert nterms == (1 if nout == 3 else 2)
    if clsname == "mv3d_bg_nodm":
        (w0, t0), (w1, t1) = g.loss_expr.terms
        assert t0.mask is not None and t0.mask.C == 1 and t0.mask.ld == 4 and t0.a.C == 3 and t0.a.ld == 4
        assert abs(w1 - 0.1) < 1e-12 and t1.b_scale == 0.75 and t1.a.requires_grad and not t1.b.requires_grad


def test_concat_is_a_view_and_checkpoint_names(lib, tmp_path):
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    m = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cpu')
    g = m.graph
    from dynamic_multiview_3d_amd.graph import ViewNode, CopyConcatNode
    assert not any(isinstance(n, CopyConcatNode) for n in g.nodes)       # tf.concat([e5, a2]) costs no copy
    assert sum(isinstance(n, ViewNode) for n in g.nodes) == 3            # 2 reshapes + 1 concat
    sd = g.state_dict()
    for k in ('fc1/Matrix', 'fc1/Matrix/Adam', 'fc1/Matrix/Adam_1', 'beta1_power', 'beta2_power'):
        assert k in sd
    path = str(tmp_path / 'model12000')
    m.saver.save(None, path)
    m.saver.restore(None, path)
    from dynamic_multiview_3d_amd.model_base import iteration_from_checkpoint_name
    assert iteration_from_checkpoint_name(path) == 12000                 # train.py:99-101


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.Mv3dError, match='no fallback'):
        _lib.lib()


def test_train_driver_loads_reference_style_conf(lib, tmp_path):
    """train.py:44-60: conf module with a `configuration` dict; reference confs import the model
    class by its bare module name and read dyn_mult_view.__file__."""
    from dynamic_multiview_3d_amd import train
    conf_py = tmp_path / 'conf.py'
    conf_py.write_text(
        "import os\ncurrent_dir = os.path.dirname(os.path.realpath(__file__))\n"
        "import sys\nsys.path.append('/home/nobody/dynamic_multiview_3d/dyn_multi_view/multi_view_model')\n"
        "from highdim_angle import AppFlowHighDimAngle\nimport dyn_mult_view\n"
        "DATA_DIR = '/'.join(str.split(dyn_mult_view.__file__, '/')[:-2]) + '/trainingdata/cardataset/train'\n"
        "configuration = {'experiment_name': 't', 'data_dir': DATA_DIR, 'output_dir': current_dir + '/modeldata',\n"
        "  'num_iterations': 20, 'batch_size': 2, 'learning_rate': 1e-4, 'train_val_split': 0.95, 'model': AppFlowHighDimAngle}\n")
    conf = train.load_conf(str(conf_py))
    from dynamic_multiview_3d_amd.highdim_angle import AppFlowHighDimAngle
    assert train.select_model(conf) is AppFlowHighDimAngle and conf['data_dir'].endswith('/trainingdata/cardataset/train')
    del conf['model']
    from dynamic_multiview_3d_amd.main_model import Base_Prediction_Model
    assert train.select_model(conf) is Base_Prediction_Model                     # train.py:57-60
    assert (train.VAL_INTERVAL, train.SAVE_INTERVAL) == (500, 10000)


def test_reference_module_paths_resolve_to_this_package():
    """`dyn_mult_view` with the reference's module paths and class names (appearance_flow_model.py:5,
    train.py:9, the conf files' bare imports)."""
    import importlib
    import dyn_mult_view
    from dyn_mult_view.mv3d.utils.tf_utils import conv2d_msra, deconv2d_msra, linear_msra, lrelu, resample_layer, warp_pts_layer  # noqa: F401
    from dyn_mult_view.multi_view_model.appearance_flow_model import AppearanceFlowModel
    from dyn_mult_view.multi_view_model.main_model import Base_Prediction_Model
    from dyn_mult_view.multi_view_model.multiobject_appflow import MultiObjectAppFlow  # noqa: F401
    from dyn_mult_view.multi_view_model.utils.read_tf_records import build_tfrecord_input  # noqa: F401
    import dynamic_multiview_3d_amd.appearance_flow_model as ours
    assert AppearanceFlowModel is ours.AppearanceFlowModel
    assert importlib.import_module('appearance_flow_model') is ours                    # conf files: `from appearance_flow_model import ...`
    assert importlib.import_module('dyn_mult_view.multi_view_model.train').main
    assert dyn_mult_view.__file__.endswith('dyn_mult_view/__init__.py')              # confs derive data_dir from it
    assert Base_Prediction_Model.__module__ == 'dynamic_multiview_3d_amd.main_model'
    with pytest.raises(ImportError):
        importlib.import_module('dyn_mult_view.collect_data')                          # out of scope stays unresolved
