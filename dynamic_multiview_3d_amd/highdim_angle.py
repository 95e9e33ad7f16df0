"""AppFlowHighDimAngle -- dyn_mult_view/multi_view_model/highdim_angle.py:5-10."""
from .appearance_flow_model import AppearanceFlowModel
from .tf_utils import *                     # noqa: F401,F403


class AppFlowHighDimAngle(AppearanceFlowModel):

    def decodeAngle(self):
        # a0 / a1 are created but unused in the reference (highdim_angle.py:8-9): they exist as
        # variables (and in checkpoints) and never receive a gradient or an update.
        a0 = lrelu(linear_msra(self.disp, 19, "a0"))
        a1 = lrelu(linear_msra(self.disp, 128, "a1"))
        return lrelu(linear_msra(self.disp, 256, "a2"))
