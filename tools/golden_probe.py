import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tests.test_model_golden import G, CASES, _build
from tests.golden.make_model_golden import fingerprint, init_variables
for name in CASES:
    builder, feeds, conf = CASES[name]
    model = _build(name, conf)
    g = model.graph
    g.set_variables(init_variables(builder, feeds))
    model.feed(**feeds)
    g.run_forward(); g.run_backward(); torch.cuda.synchronize()
    worst = dict(norm=(0, ''), sum=(0, ''), samp=(0, ''))
    for k, gr in g.get_gradients().items():
        got, want = fingerprint(gr, k), G['%s/grad/%s' % (name, k)]
        scale = max(abs(want[0]), 1e-30)
        e = dict(norm=abs(got[0] - want[0]) / scale, sum=abs(got[1] - want[1]) / scale,
                 samp=np.abs(got[2:] - want[2:]).max() / max(np.abs(want[2:]).max(), scale * 1e-3))
        for c in e:
            if e[c] > worst[c][0]: worst[c] = (e[c], k)
    print(name, {c: ('%.2e' % v[0], v[1]) for c, v in worst.items()})
