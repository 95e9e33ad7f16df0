#!/usr/bin/env python
"""Generates tests/golden/model_golden.npz -- WHOLE-GRAPH vectors of the hot path's models.

A model of this path has up to 70 M parameters, so a fixture cannot carry weights or full gradients.  Everything is
regenerated from seeds -- inputs by tests/synth.py, weights by the oracle's reference initialisers
(oracle/graph.py: tf_utils.py:58-95) -- and the fixture stores, per tensor, a fingerprint: L2 norm, sum, and 16 elements at
seeded positions.  Stored per model: fingerprints of the regenerated weights (so a changed initialiser or variable order is
caught before anything else), of every output, of every variable gradient, the loss; for AppearanceFlowModel also the
loss sequence and the weight fingerprints after three TF-Adam steps.

Like tests/golden/make_golden.py this is still "parity unpinned" (the reference holds no expected outputs: its only
fixture is multi_view_model/tests/rectangle.png, an input): the vectors come from the float32 numpy oracle and are written
only if the independent float64 torch-autograd evaluation of the same graph (oracle/torch_tape.py) agrees (outputs and loss to 2e-5, gradients to 3e-4 of each
tensor's maximum: (float32 accumulation noise of the oracle itself; exact agreement in float64 is tests/test_oracle_models.py).

    python tests/golden/make_model_golden.py        # ~2 min, ~6 GB; rewrites model_golden.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import models as omodels          # noqa: E402
from oracle.graph import Tape                 # noqa: E402
from oracle.torch_tape import run_torch       # noqa: E402
from tests.synth import appflow_feeds, multiobj_feeds   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'model_golden.npz')
NSAMP = 16


def fingerprint(a, key):
    """[L2 norm, sum, 16 sampled elements] of a tensor; positions seeded by the tensor's name (stable across runs)"""
    a = np.asarray(a, np.float64).ravel()
    seed = int.from_bytes(key.encode()[:8].ljust(8, b'\0'), 'little') ^ a.size
    idx = np.random.default_rng(seed).integers(0, a.size, NSAMP)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[idx]])


def model_cases():
    """name -> (builder, feeds, batch)"""
    f2 = appflow_feeds(np.random.default_rng(3), 2)
    fb = appflow_feeds(np.random.default_rng(4), 2)
    fb['dimage0'] = fb['image0'][..., :1].copy()
    fb['dimage1'] = fb['image1'][..., :1].copy()
    fm = multiobj_feeds(np.random.default_rng(5), 2)
    conf_b = {'use_color': '', 'use_depth': '', 'depth_lr_factor': 0.1}
    conf_fc = {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'predict_target_masks': 0.5}
    conf_conv = {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'fully_conv': ''}
    return {
        'appflow': (omodels.appearance_flow_builder('base'), f2, {}),
        'basepred': (omodels.base_prediction_builder(conf_b), fb, conf_b),
        'multiobj_fc': (omodels.multiobject_builder(conf_fc), fm, conf_fc),
        'multiobj_conv': (omodels.multiobject_builder(conf_conv), fm, conf_conv),
    }


def init_variables(builder, feeds, seed=1234):
    """reference initialisers (seeded), biases perturbed so that their gradients are exercised -- as the parity tests do"""
    rng = np.random.default_rng(seed)
    t = Tape(None, rng=rng, dtype=np.float32)
    builder(t, {k: t.const(v) for k, v in feeds.items()})
    for k, v in t.vars.items():
        if k.endswith('/b'):
            v += rng.normal(0, 0.05, v.shape).astype(np.float32)
    return t.vars


def evaluate(name, builder, feeds, adam_steps=0):
    variables = init_variables(builder, feeds)
    rec = {}
    for k, v in variables.items():
        rec['%s/w0/%s' % (name, k)] = fingerprint(v, k)
    out, grads, _ = omodels.run(builder, {k: v.copy() for k, v in variables.items()}, feeds)
    rec[name + '/loss'] = np.float64(out['loss'])
    for k, v in out.items():
        if np.ndim(v) > 0:
            rec['%s/out/%s' % (name, k)] = fingerprint(v, k)
    for k, g in grads.items():
        rec['%s/grad/%s' % (name, k)] = fingerprint(g, k)
    if adam_steps:
        v2 = {k: v.copy() for k, v in variables.items()}
        adam = omodels.AdamState(1e-4)
        rec[name + '/losses'] = np.array([omodels.step(builder, v2, adam, feeds)[0] for _ in range(adam_steps)], np.float64)
        for k, v in v2.items():
            rec['%s/w%d/%s' % (name, adam_steps, k)] = fingerprint(v, k)
    return rec, variables, out, grads


def main():
    rec = {}
    for name, (builder, feeds, _) in model_cases().items():
        r, variables, out, grads = evaluate(name, builder, feeds, adam_steps=3 if name == 'appflow' else 0)
        # independent float64 evaluation (torch ops + autograd) of the same graph on the same float32 weights and inputs
        o64, g64 = run_torch(builder, {k: v.astype(np.float64) for k, v in variables.items()}, {k: v.astype(np.float64) for k, v in feeds.items()})
        assert abs(float(out['loss']) - float(o64['loss'])) <= 2e-5 * abs(float(o64['loss'])), name
        for k, g in grads.items():
            err = np.abs(g - g64[k]).max() / max(np.abs(g64[k]).max(), 1e-30)
            assert err < 3e-4, (name, k, err)        # float32 accumulation of the numpy oracle over up to 2 M-term sums
        for k, v in out.items():
            if np.ndim(v) > 0:
                err = np.abs(v - o64[k]).max() / max(np.abs(o64[k]).max(), 1e-30)
                assert err < 2e-5, (name, k, err)
        rec.update(r)
        print(name, 'loss', float(out['loss']), '%d gradients agree with float64 autograd' % len(grads), flush=True)
    np.savez_compressed(OUT, **rec)
    print('wrote', OUT, '%d entries, %d bytes' % (len(rec), os.path.getsize(OUT)))


if __name__ == '__main__':
    main()
