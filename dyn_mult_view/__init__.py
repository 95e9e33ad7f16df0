"""`dyn_mult_view` -- the reference's package name, served by dynamic_multiview_3d_amd for the appearance-flow train step
(see dynamic_multiview_3d_amd/compat.py for the module map).  With the repo root on sys.path,
`from dyn_mult_view.multi_view_model.appearance_flow_model import AppearanceFlowModel` and
`from dyn_mult_view.mv3d.utils.tf_utils import *` resolve to the MI355X implementation."""
from dynamic_multiview_3d_amd import compat as _compat

_compat.install()
__path__ = []          # submodules come from the finder, not from this directory
