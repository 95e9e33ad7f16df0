"""Base_Prediction_Model -- drop-in for dyn_mult_view/multi_view_model/main_model.py:12-162
(the model train.py falls back to when a conf has no 'model' key, train.py:57-60; e.g.
tensorflowdata/cars_colordepth/conf.py).  RGB and/or depth towers -> shared encoder / fc bottleneck /
decoder -> one tanh decoder per modality; loss = L2(color) + depth_lr_factor * L2(depth).

Deviation: the reference hard-codes self.batch_size = 64 (main_model.py:19) while reshaping with
conf['batch_size'] (main_model.py:47-50); here conf['batch_size'] is used throughout.
"""
from .tf_utils import *                     # noqa: F401,F403
from .model_base import ModelBase, AdamOptimizer


class Base_Prediction_Model(ModelBase):
    def __init__(self, conf, load_tfrec=True, build_loss=True, device=None, seed=1234):
        self.conf = conf
        self.batch_size = conf['batch_size']
        H = conf.get('image_size', 128)
        self.image_shape = [H, H, 3]
        self.max_iter = 1000000
        self.start_iter = 0
        self.train_cond = 1

        with self._make_graph(device, seed) as g:
            B = self.batch_size
            self.image0 = g.placeholder([B, H, H, 3], 'image0')
            self.image1 = g.placeholder([B, H, H, 3], 'image1')
            self.dimage0 = g.placeholder([B, H, H, 1], 'dimage0')
            self.dimage1 = g.placeholder([B, H, H, 1], 'dimage1')
            self.disp = g.placeholder([B, 2], 'disp')
            self.buildModel()
            if build_loss:
                self.build_loss()
        self._finish(build_loss)

    def image_preprocessing(self, input, scope):
        with variable_scope(scope):
            e0 = lrelu(conv2d_msra(input, 32, 5, 5, 2, 2, "e0"))  # 64x64
            e0_0 = lrelu(conv2d_msra(e0, 32, 5, 5, 1, 1, "e0_0"))
            e1 = lrelu(conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))  # 32x32
            e1_0 = lrelu(conv2d_msra(e1, 32, 5, 5, 1, 1, "e1_0"))
            e2 = lrelu(conv2d_msra(e1_0, 64, 5, 5, 2, 2, "e2"))  # 16x16
        return e2

    def decode(self, input, scope, num_channels):
        H = self.image_shape[0]
        with variable_scope(scope):
            d2 = lrelu(deconv2d_msra(input, [self.batch_size, H // 4, H // 4, 32], 5, 5, 2, 2, "d2"))
            d2_0 = lrelu(conv2d_msra(d2, 64, 5, 5, 1, 1, "d2_0"))
            d1 = lrelu(deconv2d_msra(d2_0, [self.batch_size, H // 2, H // 2, 32], 5, 5, 2, 2, "d1"))
            d1_0 = lrelu(conv2d_msra(d1, 32, 5, 5, 1, 1, "d1_0"))
            self.pre_tanh = deconv2d_msra(d1_0, [self.batch_size, H, H, num_channels], 5, 5, 2, 2, "d0")
            gen = tanh(self.pre_tanh)
        return gen

    def buildModel(self):
        # convolutional encoder
        concat_list = []
        if 'use_color' in self.conf:
            concat_list.append(self.image_preprocessing(self.image0, 'pre_image0'))
        if 'use_depth' in self.conf:
            concat_list.append(self.image_preprocessing(self.dimage0, 'pre_dimage0'))

        comb_enc = concat(axis=3, values=concat_list)

        e2_0 = lrelu(conv2d_msra(comb_enc, 64, 5, 5, 1, 1, "e2_0"))
        e3 = lrelu(conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))  # 8x8
        e3_0 = lrelu(conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
        e4 = lrelu(conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))  # 4x4
        e4_0 = lrelu(conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))
        e4r = reshape(e4_0, [self.batch_size, 4096])
        e5 = lrelu(linear_msra(e4r, 4096, "fc1"))

        # angle processing
        a0 = lrelu(linear_msra(self.disp, 64, "a0"))
        a1 = lrelu(linear_msra(a0, 64, "a1"))
        a2 = lrelu(linear_msra(a1, 64, "a2"))

        concated = concat(axis=1, values=[e5, a2])

        # joint processing
        a3 = lrelu(linear_msra(concated, 4096, "a3"))
        a4 = lrelu(linear_msra(a3, 4096, "a4"))
        a5 = lrelu(linear_msra(a4, 4096, "a5"))
        a5r = reshape(a5, [self.batch_size, 4, 4, 256])

        # joint convolutional decoder
        d4 = lrelu(deconv2d_msra(a5r, [self.batch_size, 8, 8, 128], 3, 3, 2, 2, "d4"))
        d4_0 = lrelu(conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
        d3 = lrelu(deconv2d_msra(d4_0, [self.batch_size, 16, 16, 64], 3, 3, 2, 2, "d3"))
        num_decode = 0
        if 'use_color' in self.conf:
            num_decode += 1
        if 'use_depth' in self.conf:
            num_decode += 1
        d3_0 = lrelu(conv2d_msra(d3, 64 * num_decode, 5, 5, 1, 1, "d3_0"))

        # splitting up the representation (decoders consume the splits from the LAST one: main_model.py:131-137)
        split_list = split(d3_0, num_decode, axis=3)

        if 'use_color' in self.conf:
            self.gen_image1 = self.decode(split_list.pop(), 'dec_image1', num_channels=3)
        if 'use_depth' in self.conf:
            self.gen_dimage1 = self.decode(split_list.pop(), 'dec_dimage1', num_channels=1)
        assert split_list == []

    def build_loss(self):
        self.loss = 0.
        if 'use_color' in self.conf:
            self.loss += euclidean_loss(self.gen_image1, self.image1)
        if 'use_depth' in self.conf:
            self.loss += euclidean_loss(self.gen_dimage1, self.dimage1) * self.conf['depth_lr_factor']
        self.train_op = AdamOptimizer(self.conf['learning_rate']).minimize(self.loss, self.graph)

    def visualize(self, sess=None, **feeds):
        """One forward pass, then the reference's qualitative outputs (visualize.py)."""
        from . import visualize as _v
        return _v.visualize_prediction(self, sess, **feeds)
