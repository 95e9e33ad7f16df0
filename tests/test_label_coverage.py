"""Every kernel instance a benchmarked step launches is one a GPU parity test runs against the oracle.

The tile planner picks kernels by batch x spatial size: the batch-2 model tests and the small per-op shapes never reach
the 256- / 128-pixel, 4-phase and N64 instances that carry the batch-64 step ("What's weak" #1).  This
test records -- on the CPU, nothing is launched -- the launch plans of the benchmarked configurations and the plans of the
per-layer parity cases of tests/layer_cases.py (which tests/test_gpu_layers.py runs through the C ABI against oracle/ops.py
on the GPU), and asserts that no label of the former is missing from the latter.  A kernel label names ONE kernel
instance (mv3d::intern_label: template arguments included), so a planner change cannot outrun the tests again."""
import inspect

import pytest

from dynamic_multiview_3d_amd import _lib
from tests import layer_cases as LC

# launches that are not conv / deconv / fc layer calls: label -> the GPU test that holds that kernel to the oracle
ELEMENTWISE = {
    'fill': 'tests/test_gpu_ops.py::test_copy2d_group_sum_fill',
    'resample_loss': 'tests/test_gpu_ops.py::test_fused_resample_loss_matches_the_three_separate_ops',
    'resample_fwd': 'tests/test_gpu_ops.py::test_resampler_fwd_bwd_random_and_edges',
    'resample_bwd': 'tests/test_gpu_ops.py::test_resampler_fwd_bwd_random_and_edges',
    'pixel_loss': 'tests/test_gpu_ops.py::test_pixel_loss',
    'act_fwd': 'tests/test_gpu_ops.py::test_relu_signed_zero_and_grad',
    'act_bwd': 'tests/test_gpu_ops.py::test_relu_signed_zero_and_grad',
    'copy2d': 'tests/test_gpu_ops.py::test_copy2d_group_sum_fill',
    'group_sum': 'tests/test_gpu_ops.py::test_copy2d_group_sum_fill',
    'bconv_split_all': 'tests/test_gpu_layers.py::test_model_step_at_benchmark_batch',     # the bound-filter conversion
    'fc_wgrad_adam_b3': 'tests/test_gpu_layers.py::test_fused_fc_wgrad_adam_equals_wgrad_then_adam',
    'adam': 'tests/test_gpu_ops.py::test_adam_bit_exact_vs_oracle',
    'grad_finalize_adam': 'tests/test_gpu_layers.py::test_fused_step_equals_unfused_step',     # == reduce_slabs per layer + adam, bit for bit
    'adam_advance': 'tests/test_gpu_layers.py::test_fused_fc_wgrad_adam_equals_wgrad_then_adam',
    'small_fc_chain_fwd': 'tests/test_gpu_layers.py::test_small_fc_chain_and_fused_backward_equal_per_layer_calls',
    'small_fc_bwd': 'tests/test_gpu_layers.py::test_small_fc_chain_and_fused_backward_equal_per_layer_calls',
}


def _plan_labels(graph):
    out = set()
    for plan in (graph.plan_fwd, graph.plan_bwd, graph.plan_bwd_fused):
        if plan is None:
            continue
        out.update(o[0] for o in _lib.plan_ops(plan))
    return out


def _case_labels(conv_cases, fc_cases):
    out = set()
    for c in conv_cases:
        out.update(LC.record_conv_case(_lib, c))
    for c in fc_cases:
        out.update(LC.record_fc_case(_lib, c))
    return out


def _assert_covered(model_labels, case_labels, what):
    missing = sorted(l for l in model_labels if l not in case_labels and l not in ELEMENTWISE)
    assert not missing, "%s launches kernels no per-layer parity case dispatches: %s" % (what, missing)


def test_elementwise_coverage_table_points_at_real_tests():
    import importlib
    for label, where in ELEMENTWISE.items():
        path, name = where.split('::')
        mod = importlib.import_module(path[:-3].replace('/', '.'))
        assert inspect.isfunction(getattr(mod, name)), where


def test_appflow_batch64_labels_are_parity_tested():
    """BASELINE config 2 (the bench workload): AppearanceFlowModel at batch 64."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    m = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cpu')
    labels = _plan_labels(m.graph)
    assert any(l.startswith('s2conv<5x5,s1,C32,N32,128px') for l in labels)      # the kernel that carries the step (e0_0 / d1_0)
    _assert_covered(labels, _case_labels(LC.APPFLOW_B64, LC.FC_B64), 'AppearanceFlowModel B=64')


def test_highdim_batch64_labels_are_parity_tested():
    """BASELINE config 4: AppFlowHighDimAngle at 64 per GPU (a3 is 4352 -> 4096)."""
    from dynamic_multiview_3d_amd.highdim_angle import AppFlowHighDimAngle
    m = AppFlowHighDimAngle({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cpu')
    _assert_covered(_plan_labels(m.graph), _case_labels(LC.APPFLOW_B64, LC.FC_B64 + LC.FC_HIGHDIM), 'AppFlowHighDimAngle B=64')


def test_base_prediction_batch128_labels_are_parity_tested():
    """BASELINE config 3: main_model.Base_Prediction_Model colour + depth at batch 128."""
    from dynamic_multiview_3d_amd.main_model import Base_Prediction_Model
    conf = {'batch_size': 128, 'learning_rate': 1e-4, 'use_color': '', 'use_depth': '', 'depth_lr_factor': 0.1}
    m = Base_Prediction_Model(conf, load_tfrec=False, device='cpu')
    _assert_covered(_plan_labels(m.graph), _case_labels(LC.APPFLOW_B64 + LC.BASEPRED_B128, LC.FC_B64), 'Base_Prediction_Model B=128')


def test_multiobject_256_batch32_labels_are_parity_tested():
    """BASELINE config 5: MultiObjectAppFlow at 256 x 256, fully_conv, 32 per GPU."""
    from dynamic_multiview_3d_amd.multiobject_appflow import MultiObjectAppFlow
    m = MultiObjectAppFlow(dict(LC.MULTIOBJ_256_CONF, batch_size=32, learning_rate=1e-4), load_tfrec=False, device='cpu')
    _assert_covered(_plan_labels(m.graph), _case_labels(LC.MULTIOBJ_256_B32, LC.FC_MULTIOBJ), 'MultiObjectAppFlow 256x256 B=32')


def test_small_batch_cases_do_not_cover_the_benchmark():
    """The reason this file exists: the batch-2 shapes of tests/test_gpu_ops.py dispatch other kernels."""
    small = [(LC.CONV, 2, 64, 64, 32, 32, 5, 1, 32, 32, True), (LC.CONV, 2, 64, 64, 32, 32, 5, 2, 32, 32, True),
             (LC.DECONV, 2, 64, 64, 32, 64, 5, 2, 32, 64, True)]
    big = _case_labels(LC.APPFLOW_B64, [])
    assert not {l for l in big if l.startswith('s2conv<5x5,s1,C32,N32,128px')} <= _case_labels(small, [])


def test_extra_cases_reach_the_kernel_instances_they_are_there_for():
    """LC.EXTRA_KERNEL_CASES exists for instances no shipped configuration dispatches: check that the cases really reach them."""
    labels = _case_labels(LC.EXTRA_KERNEL_CASES, [])
    assert 'cconv<3x3,256px,N32,gmask>' in labels and 'cconv<3x3,256px,N32>' in labels, sorted(labels)
