"""MultiObjectAppFlow -- drop-in for dyn_mult_view/multi_view_model/multiobject_appflow.py:14-286.

Four input towers (rgb, depth, object masks), fc or fully-convolutional bottleneck
(multiobject_appflow.py:147-164), one appearance-flow decoder per colour output (decode_flow) and
one tanh decoder per depth / mask output (decode_direct), multi-term loss incl. the optional
masked_image_loss.  The 13 reader tensors (multiobject_appflow.py:31-43) are fed by name.
`image_size` in the conf (default 128) scales every spatial constant; sizes other than 128 need
'fully_conv' (the fc path reshapes to 4096) -- BASELINE config 5 is the 256x256 extrapolation.
"""
from .tf_utils import *                     # noqa: F401,F403
from .model_base import ModelBase, AdamOptimizer

INPUTS = (('image0', 3), ('image0_mask0', 1), ('image0_mask1', 1), ('image1', 3), ('image1_only0', 3),
          ('image1_only1', 3), ('image1_mask0', 1), ('image1_mask1', 1), ('depth0', 1), ('depth1', 1),
          ('depth1_only0', 1), ('depth1_only1', 1))


class MultiObjectAppFlow(ModelBase):
    def __init__(self, conf, load_tfrec=True, build_loss=True, device=None, seed=1234):
        self.conf = conf
        self.batch_size = conf['batch_size']
        H = conf.get('image_size', 128)
        self.image_shape = [H, H, 3]
        self.scalar_imshape = [H, H, 1]
        self.max_iter = 1000000
        self.start_iter = 0
        self.train_cond = 1
        if H != 128 and 'fully_conv' not in conf:
            raise ValueError("image_size != 128 needs 'fully_conv' (the fc bottleneck reshapes to 4096)")

        with self._make_graph(device, seed) as g:
            for name, ch in INPUTS:
                setattr(self, name, g.placeholder([self.batch_size, H, H, ch], name))
            self.displacement = g.placeholder([self.batch_size, 2], 'displacement')
            self.buildModel()
            if build_loss:
                self.build_loss()
        self._finish(build_loss)

    def image_preprocessing(self, input, scope):
        with variable_scope(scope):
            e0 = lrelu(conv2d_msra(input, 32, 5, 5, 2, 2, "e0"))
            e0_0 = lrelu(conv2d_msra(e0, 32, 5, 5, 1, 1, "e0_0"))
            e1 = lrelu(conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
            e1_0 = lrelu(conv2d_msra(e1, 32, 5, 5, 1, 1, "e1_0"))
            e2 = lrelu(conv2d_msra(e1_0, 64, 5, 5, 2, 2, "e2"))
        return e2

    def _decode_trunk(self, input):
        H = self.image_shape[0]
        d2 = lrelu(deconv2d_msra(input, [self.batch_size, H // 4, H // 4, 32], 5, 5, 2, 2, "d2"))
        d2_0 = lrelu(conv2d_msra(d2, 64, 5, 5, 1, 1, "d2_0"))
        d1 = lrelu(deconv2d_msra(d2_0, [self.batch_size, H // 2, H // 2, 32], 5, 5, 2, 2, "d1"))
        return lrelu(conv2d_msra(d1, 32, 5, 5, 1, 1, "d1_0"))

    def decode_flow(self, src_img, input, scope):
        flow_field_channels = 2
        H = self.image_shape[0]
        with variable_scope(scope):
            d1_0 = self._decode_trunk(input)
            flow_field = deconv2d_msra(d1_0, [self.batch_size, H, H, flow_field_channels], 5, 5, 2, 2, "d0")
            warp_pts = warp_pts_layer(flow_field)
            gen = resample_layer(src_img, warp_pts)
        return gen

    def decode_direct(self, input, scope, num_outputs=1):
        channels = 1
        H = self.image_shape[0]
        with variable_scope(scope):
            d1_0 = self._decode_trunk(input)
            pre_tanh = deconv2d_msra(d1_0, [self.batch_size, H, H, channels], 5, 5, 2, 2, "d0")
            gen = tanh(pre_tanh)
        return gen

    def buildModel(self):
        # convolutional encoder
        concat_list = []
        if 'use_color' in self.conf:
            concat_list.append(self.image_preprocessing(self.image0, 'pre_image0_f'))
        if 'use_depth' in self.conf:
            concat_list.append(self.image_preprocessing(self.depth0, 'pre_dimage0_f'))
        concat_list.append(self.image_preprocessing(self.image0_mask0, 'pre_mask0_ob0'))
        concat_list.append(self.image_preprocessing(self.image0_mask1, 'pre_mask0_ob1'))

        comb_enc = concat(axis=3, values=concat_list)

        e2_0 = lrelu(conv2d_msra(comb_enc, 64, 5, 5, 1, 1, "e2_0"))
        e3 = lrelu(conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
        e3_0 = lrelu(conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
        e4 = lrelu(conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
        e4_0 = lrelu(conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))

        # angle processing
        a0 = lrelu(linear_msra(self.displacement, 64, "a0"))
        a1 = lrelu(linear_msra(a0, 64, "a1"))
        a2 = lrelu(linear_msra(a1, 64, "a2"))

        if 'fully_conv' in self.conf:
            # tf.reshape [B,1,1,64] + tf.tile over the bottleneck's spatial size (multiobject_appflow.py:148-149)
            smear = tile_spatial(a2, e4_0.shape[1], e4_0.shape[2])
            concated = concat(axis=3, values=[e4_0, smear])
            e4_1 = lrelu(conv2d_msra(concated, 256, 3, 3, 1, 1, "e4_1"))
            a5r = lrelu(conv2d_msra(e4_1, 256, 3, 3, 1, 1, "e4_2"))
        else:
            e4r = reshape(e4_0, [self.batch_size, 4096])
            e5 = lrelu(linear_msra(e4r, 4096, "fc1"))
            concated = concat(axis=1, values=[e5, a2])
            a3 = lrelu(linear_msra(concated, 4096, "a3"))
            a4 = lrelu(linear_msra(a3, 4096, "a4"))
            a5 = lrelu(linear_msra(a4, 4096, "a5"))
            a5r = reshape(a5, [self.batch_size, 4, 4, 256])

        # joint convolutional decoder
        hb = a5r.shape[1]
        d4 = lrelu(deconv2d_msra(a5r, [self.batch_size, 2 * hb, 2 * hb, 128], 3, 3, 2, 2, "d4"))
        d4_0 = lrelu(conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
        d3 = lrelu(deconv2d_msra(d4_0, [self.batch_size, 4 * hb, 4 * hb, 64], 3, 3, 2, 2, "d3"))
        num_decode = 0
        if 'use_color' in self.conf:
            if 'combination_image' in self.conf:
                num_decode += 1
            if 'gen_sep_images' in self.conf:
                num_decode += 2
        if 'use_depth' in self.conf:
            if 'combination_image' in self.conf:
                num_decode += 1
            if 'gen_sep_images' in self.conf:
                num_decode += 2
        if 'predict_target_masks' in self.conf:
            num_decode += 2

        d3_0 = lrelu(conv2d_msra(d3, 64 * num_decode, 5, 5, 1, 1, "d3_0"))

        # splitting up the representation
        split_list = split(d3_0, num_decode, axis=3)
        self.gen_image1 = self.gen_image1_only0 = self.gen_image1_only1 = None
        self.gen_depth1 = self.gen_depth1_only0 = self.gen_depth1_only1 = None
        self.gen_image1_mask0 = self.gen_image1_mask1 = None

        if 'use_color' in self.conf:
            if 'combination_image' in self.conf:
                self.gen_image1 = self.decode_flow(self.image0, split_list.pop(), 'dec_image1')
            if 'gen_sep_images' in self.conf:
                self.gen_image1_only0 = self.decode_flow(self.image0, split_list.pop(), 'dec_image1_only0')
                self.gen_image1_only1 = self.decode_flow(self.image0, split_list.pop(), 'dec_image1_only1')

        if 'use_depth' in self.conf:
            if 'combination_image' in self.conf:
                self.gen_depth1 = self.decode_direct(split_list.pop(), 'dec_dimage1_f', num_outputs=1)
            if 'gen_sep_images' in self.conf:
                self.gen_depth1_only0 = self.decode_direct(split_list.pop(), 'dec_depth1_only0', num_outputs=1)
                self.gen_depth1_only1 = self.decode_direct(split_list.pop(), 'dec_depth1_only1', num_outputs=1)

        if 'predict_target_masks' in self.conf:
            self.gen_image1_mask0 = self.decode_direct(split_list.pop(), 'dec_image1_mask0', num_outputs=1)
            self.gen_image1_mask1 = self.decode_direct(split_list.pop(), 'dec_image1_mask1', num_outputs=1)

        assert split_list == []

    def build_loss(self):
        self.loss = 0
        if 'use_color' in self.conf:
            colorloss = 0.
            if 'combination_image' in self.conf:
                colorloss += euclidean_loss(self.gen_image1, self.image1)
            if 'gen_sep_images' in self.conf:
                if 'masked_image_loss' in self.conf:
                    colorloss += masked_euclidean_loss(self.gen_image1_only0, self.image1_only0, self.image1_mask0)
                    colorloss += masked_euclidean_loss(self.gen_image1_only1, self.image1_only1, self.image1_mask1)
                else:
                    colorloss += euclidean_loss(self.gen_image1_only0, self.image1_only0)
                    colorloss += euclidean_loss(self.gen_image1_only1, self.image1_only1)
            self.loss += colorloss

        if 'use_depth' in self.conf:
            depthloss = 0.
            depth_factor = self.conf['use_depth']       # the flag's VALUE is the factor (multiobject_appflow.py:252)
            if 'combination_image' in self.conf:
                depthloss += euclidean_loss(self.gen_depth1, self.depth1)      # not scaled: multiobject_appflow.py:254
            if 'gen_sep_images' in self.conf:
                if 'masked_image_loss' in self.conf:
                    depthloss += masked_euclidean_loss(self.gen_depth1_only0, self.depth1_only0, self.image1_mask0) * depth_factor
                    depthloss += masked_euclidean_loss(self.gen_depth1_only1, self.depth1_only1, self.image1_mask1) * depth_factor
                else:
                    depthloss += euclidean_loss(self.gen_depth1_only0, self.depth1_only0) * depth_factor
                    depthloss += euclidean_loss(self.gen_depth1_only1, self.depth1_only1) * depth_factor
            self.loss += depthloss

        if 'predict_target_masks' in self.conf:
            mask_factor = self.conf['predict_target_masks']
            mask_loss = 0.
            mask_loss += euclidean_loss(self.gen_image1_mask0, self.image1_mask0) * mask_factor
            mask_loss += euclidean_loss(self.gen_image1_mask1, self.image1_mask1) * mask_factor
            self.loss += mask_loss

        self.train_op = AdamOptimizer(self.conf['learning_rate']).minimize(self.loss, self.graph)

    def visualize(self, sess=None, **feeds):
        """One forward pass, then the reference's qualitative outputs (visualize.py)."""
        from . import visualize as _v
        return _v.visualize_multiobject(self, sess, **feeds)
