"""AppearanceFlowModel -- drop-in for dyn_mult_view/multi_view_model/appearance_flow_model.py:17-130.

Same constructor signature, method names (buildModel, build_loss, decodeAngle) and attributes
(image0, image1, disp, flow_field, warp_pts, gen, loss, train_op, t_vars, saver); the TF session
step becomes train_step()/forward() (model_base.py).  Inputs are fed (feed-style API of
mv3d/nobg_nodm.py:146-150): the reference wires its TFRecord queue into the graph when load_tfrec is set
(appearance_flow_model.py:30-34); here the reader (read_tf_records.py, no TensorFlow) is a host-side feeder
that the train driver connects to train_step(**batch), so load_tfrec has no effect on the graph itself.
"""
from .tf_utils import *                     # noqa: F401,F403  (same star-import as the reference)
from .model_base import ModelBase, AdamOptimizer


class AppearanceFlowModel(ModelBase):

    def __init__(self, conf, load_tfrec=True, build_loss=True, device=None, seed=1234):
        self.conf = conf
        self.batch_size = conf['batch_size']
        self.image_shape = [128, 128, 3]
        self.max_iter = 1000000
        self.start_iter = 0
        self.train_cond = 1
        H = conf.get('image_size', 128)
        self.image_shape = [H, H, 3]

        with self._make_graph(device, seed) as g:
            # appearance_flow_model.py:53-56 (disp is [B,2]: read_tf_records.py:61,78)
            self.image0 = g.placeholder([self.batch_size, H, H, 3], 'image0')
            self.image1 = g.placeholder([self.batch_size, H, H, 3], 'image1')
            self.depth_image0 = g.placeholder([self.batch_size, H, H, 1], 'depth_image0')
            self.depth_image1 = g.placeholder([self.batch_size, H, H, 1], 'depth_image1')
            self.disp = g.placeholder([self.batch_size, 2], 'disp')
            self.buildModel()
            if build_loss:
                self.build_loss()
        self._finish(build_loss)

    def decodeAngle(self):
        a0 = lrelu(linear_msra(self.disp, 64, "a0"))
        a1 = lrelu(linear_msra(a0, 64, "a1"))
        return lrelu(linear_msra(a1, 64, "a2"))

    def build_loss(self):
        self.loss = euclidean_loss(self.gen, self.image1)
        self.train_op = AdamOptimizer(self.conf['learning_rate']).minimize(self.loss, self.graph)

    def buildModel(self):
        image0 = self.image0

        # convolutional encoder
        e0 = lrelu(conv2d_msra(image0, 32, 5, 5, 2, 2, "e0"))
        e0_0 = lrelu(conv2d_msra(e0, 32, 5, 5, 1, 1, "e0_0"))
        e1 = lrelu(conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
        e1_0 = lrelu(conv2d_msra(e1, 32, 5, 5, 1, 1, "e1_0"))
        e2 = lrelu(conv2d_msra(e1_0, 64, 5, 5, 2, 2, "e2"))
        e2_0 = lrelu(conv2d_msra(e2, 64, 5, 5, 1, 1, "e2_0"))
        e3 = lrelu(conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
        e3_0 = lrelu(conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
        e4 = lrelu(conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
        e4_0 = lrelu(conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))
        e4r = reshape(e4_0, [self.batch_size, 4096])
        e5 = lrelu(linear_msra(e4r, 4096, "fc1"))

        # angle processing
        concated = concat(axis=1, values=[e5, self.decodeAngle()])

        # joint processing
        a3 = lrelu(linear_msra(concated, 4096, "a3"))
        a4 = lrelu(linear_msra(a3, 4096, "a4"))
        a5 = lrelu(linear_msra(a4, 4096, "a5"))
        a5r = reshape(a5, [self.batch_size, 4, 4, 256])

        # convolutional decoder
        d4 = lrelu(deconv2d_msra(a5r, [self.batch_size, 8, 8, 128], 3, 3, 2, 2, "d4"))
        d4_0 = lrelu(conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
        d3 = lrelu(deconv2d_msra(d4_0, [self.batch_size, 16, 16, 64], 3, 3, 2, 2, "d3"))
        d3_0 = lrelu(conv2d_msra(d3, 64, 5, 5, 1, 1, "d3_0"))
        d2 = lrelu(deconv2d_msra(d3_0, [self.batch_size, 32, 32, 32], 5, 5, 2, 2, "d2"))
        d2_0 = lrelu(conv2d_msra(d2, 64, 5, 5, 1, 1, "d2_0"))
        d1 = lrelu(deconv2d_msra(d2_0, [self.batch_size, 64, 64, 32], 5, 5, 2, 2, "d1"))
        d1_0 = lrelu(conv2d_msra(d1, 32, 5, 5, 1, 1, "d1_0"))

        # appearance flow head
        self.flow_field = deconv2d_msra(d1_0, [self.batch_size, 128, 128, 2], 5, 5, 2, 2, "flow_field")
        self.warp_pts = warp_pts_layer(self.flow_field)
        self.gen = resample_layer(image0, self.warp_pts)

    def visualize(self, sess=None, **feeds):
        """One forward pass, then the reference's qualitative outputs (visualize.py)."""
        from . import visualize as _v
        return _v.visualize_appearance_flow(self, sess, **feeds)
