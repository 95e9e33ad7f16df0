"""Whole-graph checks of the oracle's reverse pass against torch autograd (float64)."""
import numpy as np
import pytest

from oracle import models
from oracle.graph import Tape
from oracle.torch_tape import run_torch
from tests.synth import appflow_feeds, multiobj_feeds


def _init_vars(builder, feeds, seed=1234):
    """Create variables with the reference initialisers by running the builder once."""
    rng = np.random.default_rng(seed)
    t = Tape(None, rng=rng, dtype=np.float32)
    nodes = {k: t.const(v) for k, v in feeds.items()}
    builder(t, nodes)
    # biases are zero at init (tf_utils.py:65,80); perturb them so their gradients are exercised
    for k, v in t.vars.items():
        if k.endswith('/b'):
            v += rng.normal(0, 0.05, v.shape).astype(np.float32)
    return t.vars, t.used


def _compare(builder, feeds, expect_dead=()):
    variables, used = _init_vars(builder, feeds)
    v64 = {k: v.astype(np.float64) for k, v in variables.items()}
    f64 = {k: v.astype(np.float64) for k, v in feeds.items()}
    out_o, g_o, _ = models.run(builder, dict(v64), f64, dtype=np.float64)
    out_t, g_t = run_torch(builder, v64, f64)
    np.testing.assert_allclose(out_o['loss'], out_t['loss'], rtol=1e-10)
    for k in out_o:
        np.testing.assert_allclose(out_o[k], out_t[k], rtol=1e-8, atol=1e-10, err_msg=k)
    dead = set(variables) - set(g_o)
    assert dead == set(expect_dead), dead
    assert set(g_o) == set(g_t)
    for k in g_o:
        scale = max(np.abs(g_t[k]).max(), 1e-30)
        assert np.abs(g_o[k] - g_t[k]).max() <= 1e-8 * scale, k
    return variables, used


@pytest.mark.parametrize("variant,dead", [
    ('base', ()), ('lowdim', ()), ('tinghui', ()),
    ('highdim', ('a0/Matrix', 'a0/b', 'a1/Matrix', 'a1/b')),
])
def test_appearance_flow_graph(variant, dead):
    feeds = appflow_feeds(np.random.default_rng(3), 2)
    variables, used = _compare(models.appearance_flow_builder(variant), feeds, dead)
    if variant == 'base':
        assert sum(v.size for v in variables.values()) == 69535232
        assert used[:4] == ['e0/w', 'e0/b', 'e0_0/w', 'e0_0/b'] and used[-1] == 'flow_field/w'
        assert len(used) == 47


def test_base_prediction_graph():
    rng = np.random.default_rng(4)
    f = appflow_feeds(rng, 2)
    f['dimage0'] = f['image0'][..., :1].copy()
    f['dimage1'] = f['image1'][..., :1].copy()
    conf = {'use_color': '', 'use_depth': '', 'depth_lr_factor': 0.1}
    variables, used = _compare(models.base_prediction_builder(conf), f)
    assert 'pre_dimage0/e0/w' in variables and variables['pre_dimage0/e0/w'].shape == (5, 5, 1, 32)
    assert variables['d3_0/w'].shape == (5, 5, 64, 128)
    assert variables['dec_image1/d0/w'].shape == (5, 5, 3, 32)


def mv3d_feeds(rng, b, variant):
    f = appflow_feeds(rng, b)
    images2 = f['image1'] * 2.0 - 1.0                       # tanh range
    if variant != 'nobg_nodm':
        extra = (f['image0'][..., :1] > 0.5).astype(np.float32) if variant == 'bg_nodm' else f['image0'][..., :1] * 2.0 - 1.0
        images2 = np.concatenate([images2, extra], axis=3)
    labels = rng.uniform(-1, 1, (b, 5)).astype(np.float32)
    return {'images1': f['image0'], 'images2': images2.astype(np.float32), 'labels': labels}


@pytest.mark.parametrize("variant,nvars,nparams", [('nobg_nodm', 47, 69536224), ('nobg_dm', 47, 69537024), ('bg_nodm', 49, 69260772)])
def test_mv3d_graphs(variant, nvars, nparams):
    """the direct-prediction mv3d networks (mv3d/nobg_nodm.py, nobg_dm.py, bg_nodm.py)."""
    feeds = mv3d_feeds(np.random.default_rng(6), 2, variant)
    variables, used = _compare(models.mv3d_builder(variant), feeds)
    assert len(used) == nvars and sum(v.size for v in variables.values()) == nparams
    assert used[-1] == ('d0_1/b' if variant == 'bg_nodm' else 'd0/w')
    assert variables['d0/w'].shape == ((5, 5, 16, 32) if variant == 'bg_nodm' else (5, 5, 3 if variant == 'nobg_nodm' else 4, 32))


@pytest.mark.parametrize("conf", [
    {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'fully_conv': ''},
    {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'predict_target_masks': 0.5},
    {'use_color': '', 'gen_sep_images': '', 'masked_image_loss': '', 'fully_conv': ''},
])
def test_multiobject_graph(conf):
    f = multiobj_feeds(np.random.default_rng(5), 2)
    _compare(models.multiobject_builder(conf), f)


def test_multiobject_main_model_graph():
    """multiobject_main_model.Base_Prediction_Model: colour outputs from direct 3-channel tanh decoders."""
    conf = {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'masked_image_loss': '', 'fully_conv': ''}
    variables, used = _compare(models.multiobject_builder(conf, direct_color=True), multiobj_feeds(np.random.default_rng(7), 2))
    assert variables['dec_image1/d0/w'].shape == (5, 5, 3, 32) and variables['dec_depth1_only1/d0/w'].shape == (5, 5, 1, 32)


def test_three_adam_steps_reduce_loss_and_are_deterministic():
    feeds = appflow_feeds(np.random.default_rng(6), 2)
    builder = models.appearance_flow_builder('base')
    variables, _ = _init_vars(builder, feeds)
    losses = []
    for rep in range(2):
        v = {k: a.copy() for k, a in variables.items()}
        adam = models.AdamState(1e-4)
        losses.append([models.step(builder, v, adam, feeds)[0] for _ in range(3)])
    assert losses[0] == losses[1]
    assert losses[0][2] < losses[0][0]
