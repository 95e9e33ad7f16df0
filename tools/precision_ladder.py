#!/usr/bin/env python
"""Precision ladder of the matrix-core path, measured on the CPU oracle.

The HIP kernels evaluate an fp32 product a*b from bf16 pieces a = a_hi + a_lo, b = b_hi + b_lo (fp32 accumulation):
  3 products  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi     (what ships: csrc/bconv.hip, cconv.hip, wgrad_tile.hip, fc.hip)
  2 products  a_hi*b_hi + a_lo*b_hi = a * bf16(b)    second operand (filter; dy in the filter gradient) rounded to bf16
  2 products  a_hi*b_hi + a_hi*b_lo = bf16(a) * b    first operand (activation / incoming gradient; x in the filter gradient) rounded
  1 product   bf16(a) * bf16(b)                      plain bf16 operands (BASELINE config 2 names bf16)
bf16 x bf16 products are exact in fp32 and hi + lo carries 16 significant bits, so a rung is reproduced EXACTLY (up to fp32
summation order) by rounding the corresponding operand of every conv / deconv / fc GEMM to bf16 and multiplying in fp32 --
which is what LadderTape does on top of oracle.graph.Tape, for forward, data-gradient and filter-gradient GEMMs alike.
Reported per rung, against the unrounded fp32 oracle on identical inputs and weights (AppearanceFlowModel, config 2 shapes):
loss, gen and flow_field errors and the worst / median relative error (max-norm per tensor) over the 47 gradients.

    python tools/precision_ladder.py [batch]      (CPU only; ~1 min at batch 8)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ops, models as omodels
from oracle.graph import Tape, Node
from tests.synth import appflow_feeds


def bf16(a):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


class LadderTape(Tape):
    """q1 / q2: round the first (streamed activation or gradient) / second (filter; dy in the filter gradient) operand"""
    q1 = q2 = False

    def _r1(self, a):
        return bf16(a) if self.q1 else a

    def _r2(self, a):
        return bf16(a) if self.q2 else a

    def conv2d_msra(self, x, output_dim, k_h, k_w, d_h, d_w, name):
        cin = x.v.shape[-1]
        with self.variable_scope(name):
            wn, w = self._var('w', (k_h, k_w, cin, output_dim), None)
            bn, b = self._var('b', (output_dim,), None)
        y = Node(ops.conv2d_fwd(self._r1(x.v), self._r2(w), b, d_h, d_w))

        def back():
            _, dw, db = ops.conv2d_bwd(self._r1(x.v), w, self._r2(y.g), d_h, d_w, need_dx=False)
            self._acc_var(wn, dw)
            self._acc_var(bn, y.g.reshape(-1, output_dim).sum(0))
            if x.needs_grad:
                dx, _, _ = ops.conv2d_bwd(x.v, self._r2(w), self._r1(y.g), d_h, d_w, need_dx=True)
                x.acc(dx)
        self._rec([y], back)
        return y

    def deconv2d_msra(self, x, output_shape, k_h, k_w, d_h, d_w, name):
        cin = x.v.shape[-1]
        with self.variable_scope(name):
            wn, w = self._var('w', (k_h, k_w, output_shape[-1], cin), None)
        y = Node(ops.deconv2d_fwd(self._r1(x.v), self._r2(w), (output_shape[1], output_shape[2]), d_h, d_w))

        def back():
            dx, _ = ops.deconv2d_bwd(x.v, self._r2(w), self._r1(y.g), d_h, d_w)
            # filter gradient: the image-side operand (dy here) is the streamed one, the feature side (x) the second
            _, dw = ops.deconv2d_bwd(self._r2(x.v), w, self._r1(y.g), d_h, d_w)
            self._acc_var(wn, dw)
            x.acc(dx)
        self._rec([y], back)
        return y

    def linear_msra(self, x, output_size, name):
        fan_in = x.v.shape[-1]
        with self.variable_scope(name):
            mn, m = self._var('Matrix', (fan_in, output_size), None)
            bn, b = self._var('b', (output_size,), None)
        y = Node(ops.linear_fwd(self._r1(x.v), self._r2(m), b))

        def back():
            dx = self._r1(y.g) @ self._r2(m).T
            dm = self._r1(x.v).T @ self._r2(y.g)
            self._acc_var(mn, dm)
            self._acc_var(bn, y.g.sum(0))
            x.acc(dx)
        self._rec([y], back)
        return y


def run(builder, variables, feeds, q1, q2):
    t = LadderTape({k: v.copy() for k, v in variables.items()})
    t.q1, t.q2 = q1, q2
    out = builder(t, {k: t.const(v) for k, v in feeds.items()})
    grads = t.backward(out['loss'])
    return {k: n.v for k, n in out.items()}, grads


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    feeds = appflow_feeds(np.random.default_rng(3), B)
    builder = omodels.appearance_flow_builder('base')
    t = Tape(None, rng=np.random.default_rng(1234))
    builder(t, {k: t.const(v) for k, v in feeds.items()})
    variables = t.vars
    rng = np.random.default_rng(5)
    for k, v in variables.items():                       # non-zero biases, as the parity tests use
        if k.endswith('/b'):
            variables[k] = v + rng.normal(0, 0.05, v.shape).astype(np.float32)
    # At initialisation every flow is ~1e-3 pixel, so every sampling coordinate sits ON an integer and the sampler's cell choice
    # (floor) -- hence every upstream gradient -- flips under any perturbation.  A trained model predicts
    # flows of a few pixels: scale the flow head so that the coordinates are generic, as they are during all but the first steps.
    probe, _ = run(builder, variables, feeds, False, False)
    variables['flow_field/w'] = (variables['flow_field/w'] * (1.5 / max(float(probe['flow_field'].std()), 1e-12))).astype(np.float32)
    ref_out, ref_g = run(builder, variables, feeds, False, False)
    print("flow_field std after scaling the head: %.2f pixels" % float(ref_out['flow_field'].std()))
    print("AppearanceFlowModel, batch %d, reference initialisers; errors vs the unrounded fp32 oracle (max-norm relative per tensor)" % B)
    print("%-44s %10s %10s %10s %12s %12s" % ("rung", "loss", "gen", "flow_field", "grad worst", "grad median"))
    rows = [("3 products (ships; operands carry 16 bits)", None),
            ("2 products, filter / dy rounded to bf16", (False, True)),
            ("2 products, activation / x rounded to bf16", (True, False)),
            ("1 product, plain bf16 operands", (True, True))]
    for name, q in rows:
        if q is None:
            # hi + lo keeps 16 significant bits of each operand: emulate by rounding both operands to 16 bits
            print("%-44s %10s %10s %10s %12s %12s" % (name, "measured on the GPU: tests/test_gpu_model.py (2-7e-5 worst gradient)", "", "", "", ""))
            continue
        out, g = run(builder, variables, feeds, *q)
        errs = sorted(rel(g[k], ref_g[k]) for k in ref_g)
        print("%-44s %10.2e %10.2e %10.2e %12.2e %12.2e" % (name, abs(float(out['loss']) - float(ref_out['loss'])) / abs(float(ref_out['loss'])),
                                                            rel(out['gen'], ref_out['gen']), rel(out['flow_field'], ref_out['flow_field']),
                                                            errs[-1], errs[len(errs) // 2]), flush=True)


if __name__ == '__main__':
    main()
