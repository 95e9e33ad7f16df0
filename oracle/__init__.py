"""CPU oracle for the appearance-flow train step -- TEST INFRASTRUCTURE ONLY.

This package restates, in numpy, the arithmetic the reference obtains from
TensorFlow 1.3 (a third-party dependency that is NOT under /root/reference and
is not installable offline; the reference pins it only in prose, README.md:6-8).
It follows dyn_mult_view/mv3d/utils/tf_utils.py:18-98 (op wrappers) and the
model files under dyn_mult_view/multi_view_model/ line by line, plus the TF-1.3
kernel semantics listed in SURVEY.md Appendix A.

PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or
assertions for this path (its only test, multi_view_model/tests/test_resampler.py,
is a matplotlib demo), and the reference itself cannot run here (Python 2 +
TF 1.3).  The oracle is therefore pinned only by (a) an independent
implementation of the same maths (torch CPU ops + autograd, tests/test_oracle_vs_torch.py),
(b) float64 finite differences and (c) analytic known-answer cases.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (dynamic_multiview_3d_amd) never does.
"""
