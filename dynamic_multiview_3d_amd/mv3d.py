"""The direct-prediction `mv3d` networks -- drop-ins for the `mv3d` classes of dyn_mult_view/mv3d/nobg_nodm.py:10-95,
nobg_dm.py:10-96 and bg_nodm.py:10-98: encoder -> fc bottleneck joined with the 5-d view label ->
decoder -> tanh image; `nobg_dm` adds a depth-map channel (L2 colour + 0.1 L1 depth), `bg_nodm` a silhouette channel
(L2 on the masked colour + 0.1 L2 against 0.75 x mask).

Attribute names follow the reference (images1, images2, labels, gen, loss, t_vars, saver, batch_size, image shapes);
the TF session / Panda3D renderer / snapshot bookkeeping of the reference classes is not part of the train step:
`buildModel()` runs in the constructor and `train_step(images1=.., images2=.., labels=..)` replaces
`sess.run([optim, self.loss], feed_dict)` (nobg_nodm.py:147-152).  Optimiser: Adam(1e-4, beta1 0.9) as in `train()`.
No new kernels: the tf.slice pairs become channel views, the mask product and the 0.75 target scale are folded into the
loss kernel (mv3d_pixel_loss_strided).
"""
from .tf_utils import *                     # noqa: F401,F403
from .model_base import ModelBase, AdamOptimizer


class _MV3DBase(ModelBase):
    variant = None
    input_shape = [128, 128, 3]
    output_shape = [128, 128, 3]

    def __init__(self, conf=None, load_tfrec=False, build_loss=True, device=None, seed=1234):
        conf = dict(conf or {})
        self.conf = conf
        self.batch_size = conf.get('batch_size', 64)                  # nobg_nodm.py:15
        self.learning_rate = conf.get('learning_rate', 0.0001)       # nobg_nodm.py:139-140
        self.image_shape = list(self.input_shape)
        self.max_iter = 1000000
        self.start_iter = 0
        with self._make_graph(device, seed) as g:
            B = self.batch_size
            self.images1 = g.placeholder([B] + list(self.input_shape), 'images1')
            self.images2 = g.placeholder([B] + list(self.output_shape), 'images2')
            self.labels = g.placeholder([B, 5], 'labels')
            self.buildModel()
            if build_loss:
                self.build_loss()
        self._finish(build_loss)

    # layer differences of bg_nodm.py:42-84: 3x3 instead of 5x5 stride-1 convs, names *_1, 16-channel e0 / d0
    bg = False

    def buildModel(self):
        B = self.batch_size
        k = 3 if self.bg else 5
        sfx = '_1' if self.bg else '_0'
        # convolutional encoder
        e0 = lrelu(conv2d_msra(self.images1, 16 if self.bg else 32, 5, 5, 2, 2, "e0"))
        e0_0 = lrelu(conv2d_msra(e0, 32, k, k, 1, 1, "e0" + sfx))
        e1 = lrelu(conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
        e1_0 = lrelu(conv2d_msra(e1, 32, k, k, 1, 1, "e1" + sfx))
        e2 = lrelu(conv2d_msra(e1_0, 64, k, k, 2, 2, "e2"))
        e2_0 = lrelu(conv2d_msra(e2, 64, k, k, 1, 1, "e2" + sfx))
        e3 = lrelu(conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
        e3_0 = lrelu(conv2d_msra(e3, 128, 3, 3, 1, 1, "e3" + sfx))
        e4 = lrelu(conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
        e4_0 = lrelu(conv2d_msra(e4, 256, 3, 3, 1, 1, "e4" + sfx))
        e4r = reshape(e4_0, [B, 4096])
        e5 = lrelu(linear_msra(e4r, 4096, "fc1"))
        # angle processing
        a0 = lrelu(linear_msra(self.labels, 64, "a0"))
        a1 = lrelu(linear_msra(a0, 64, "a1"))
        a2 = lrelu(linear_msra(a1, 64, "a2"))
        concated = concat(axis=1, values=[e5, a2])
        # joint processing
        a3 = lrelu(linear_msra(concated, 4096, "a3"))
        a4 = lrelu(linear_msra(a3, 4096, "a4"))
        a5 = lrelu(linear_msra(a4, 4096, "a5"))
        a5r = reshape(a5, [B, 4, 4, 256])
        # convolutional decoder
        d4 = lrelu(deconv2d_msra(a5r, [B, 8, 8, 128], 3, 3, 2, 2, "d4"))
        d4_0 = lrelu(conv2d_msra(d4, 128, 3, 3, 1, 1, "d4" + sfx))
        d3 = lrelu(deconv2d_msra(d4_0, [B, 16, 16, 64], 3, 3, 2, 2, "d3"))
        d3_0 = lrelu(conv2d_msra(d3, 64, k, k, 1, 1, "d3" + sfx))
        d2 = lrelu(deconv2d_msra(d3_0, [B, 32, 32, 32], 5, 5, 2, 2, "d2"))
        d2_0 = lrelu(conv2d_msra(d2, 32 if self.bg else 64, k, k, 1, 1, "d2" + sfx))
        d1 = lrelu(deconv2d_msra(d2_0, [B, 64, 64, 32], 5, 5, 2, 2, "d1"))
        d1_0 = lrelu(conv2d_msra(d1, 32, k, k, 1, 1, "d1" + sfx))
        if self.bg:
            d0 = lrelu(deconv2d_msra(d1_0, [B, 128, 128, 16], 5, 5, 2, 2, "d0"))
            self.gen = tanh(conv2d_msra(d0, 4, 3, 3, 1, 1, "d0_1"))
        else:
            self.gen = tanh(deconv2d_msra(d1_0, [B] + list(self.output_shape), 5, 5, 2, 2, "d0"))

    def build_loss(self):
        raise NotImplementedError

    def _minimize(self):
        self.train_op = AdamOptimizer(self.learning_rate).minimize(self.loss, self.graph)


class mv3d_nobg_nodm(_MV3DBase):
    """mv3d/nobg_nodm.py: RGB in, RGB out, loss = euclidean_loss(gen, images2) (:86)."""

    def build_loss(self):
        self.loss = euclidean_loss(self.gen, self.images2)
        self._minimize()


class mv3d_nobg_dm(_MV3DBase):
    """mv3d/nobg_dm.py: RGB in, RGB + depth map out; loss = L2(colour) + 0.1 * L1(depth) (:85-92)."""
    output_shape = [128, 128, 4]

    def build_loss(self):
        gt_cm, gt_dm = split(self.images2, [3, 1], axis=3)
        pr_cm, pr_dm = split(self.gen, [3, 1], axis=3)
        self.loss = euclidean_loss(gt_cm, pr_cm) + 0.1 * l1_loss(gt_dm, pr_dm)
        self._minimize()


class mv3d_bg_nodm(_MV3DBase):
    """mv3d/bg_nodm.py: RGB (with background) in, RGB + silhouette out; loss = L2 on the silhouette-masked colour +
    0.1 * L2(0.75 * silhouette, predicted silhouette) (:85-93)."""
    output_shape = [128, 128, 4]
    bg = True

    def build_loss(self):
        gt_cm, gt_sm = split(self.images2, [3, 1], axis=3)
        sm = gt_sm
        gt_sm = scale(gt_sm, 0.75)
        pr_cm, pr_sm = split(self.gen, [3, 1], axis=3)
        self.loss = euclidean_loss(multiply(gt_cm, sm), multiply(pr_cm, sm)) + 0.1 * euclidean_loss(gt_sm, pr_sm)
        self._minimize()
