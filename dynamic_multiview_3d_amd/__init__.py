"""dynamic_multiview_3d_amd -- MI355X-native appearance-flow train step (hot path of
aclike/dynamic_multiview_3d: dyn_mult_view/mv3d + dyn_mult_view/multi_view_model).

Python host classes with the reference's model-construction API on top of hand-written HIP
kernels (csrc/, C ABI in include/mv3d_hip.h).  PyTorch only provides device memory, streams and
torch.distributed.
"""
__version__ = "0.1.0"
