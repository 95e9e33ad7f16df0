"""Base_Prediction_Model (multi-object) -- drop-in for dyn_mult_view/multi_view_model/multiobject_main_model.py:14-270
(SURVEY 8f rank 3).  Same towers, bottleneck (fc or fully_conv), joint decoder, representation split and loss terms as
MultiObjectAppFlow (multiobject_appflow.py) -- the reference's two files share that code verbatim -- but every output is
predicted directly: `decode(input, scope, num_outpus)` (multiobject_main_model.py:90-103) ends in a tanh deconvolution
with 3 channels for the colour outputs and 1 for depth / masks, where the appearance-flow model warps image0.
"""
from .tf_utils import *                     # noqa: F401,F403
from .multiobject_appflow import MultiObjectAppFlow


class Base_Prediction_Model(MultiObjectAppFlow):
    def decode(self, input, scope, num_outpus=3):
        H = self.image_shape[0]
        with variable_scope(scope):
            d1_0 = self._decode_trunk(input)
            self.pre_tanh = deconv2d_msra(d1_0, [self.batch_size, H, H, num_outpus], 5, 5, 2, 2, "d0")
            gen = tanh(self.pre_tanh)
        return gen

    # buildModel of the parent asks for these two; here both are the direct decoder
    def decode_flow(self, src_img, input, scope):
        return self.decode(input, scope)

    def decode_direct(self, input, scope, num_outputs=1):
        return self.decode(input, scope, num_outpus=num_outputs)
