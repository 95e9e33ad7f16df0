"""AppFlowLowDimAngle -- dyn_mult_view/multi_view_model/lowdim_angle.py:5-8."""
from .appearance_flow_model import AppearanceFlowModel
from .tf_utils import *                     # noqa: F401,F403


class AppFlowLowDimAngle(AppearanceFlowModel):

    def decodeAngle(self):
        return lrelu(linear_msra(self.disp, 10, "a0"))
