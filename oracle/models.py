"""numpy restatement of the reference model graphs -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Each builder follows one buildModel()/
build_loss() of /root/reference/dyn_mult_view/multi_view_model/ (file:line in the
docstrings) on top of oracle.graph.Tape.  `step()` = what one
sess.run([model.loss, model.train_op]) does (train.py:122): forward, reverse pass,
TF-Adam on every variable that received a gradient.
"""
from collections import OrderedDict
import numpy as np
from . import ops
from .graph import Tape


# --------------------------------------------------------------------------- appearance flow family
def _decode_angle(t, disp, variant):
    if variant in ('base', 'tinghui'):
        # appearance_flow_model.py:63-66 / appearance_flow_tinghui.py:7-11
        a0 = t.lrelu(t.linear_msra(disp, 64, "a0"))
        a1 = t.lrelu(t.linear_msra(a0, 64, "a1"))
        return t.lrelu(t.linear_msra(a1, 64, "a2"))
    if variant == 'highdim':
        # highdim_angle.py:7-10 -- a0 and a1 are created but dead
        t.lrelu(t.linear_msra(disp, 19, "a0"))
        t.lrelu(t.linear_msra(disp, 128, "a1"))
        return t.lrelu(t.linear_msra(disp, 256, "a2"))
    if variant == 'lowdim':
        # lowdim_angle.py:7-8
        return t.lrelu(t.linear_msra(disp, 10, "a0"))
    raise ValueError(variant)


def appearance_flow(t, image0, image1, disp, variant='base', build_loss=True):
    """AppearanceFlowModel.buildModel/build_loss (appearance_flow_model.py:68-130);
    variant 'tinghui' = AppearanceFlowTinghui.buildModel (appearance_flow_tinghui.py:13-49)."""
    B = image0.v.shape[0]
    out = OrderedDict()
    if variant == 'tinghui':
        e0 = t.relu(t.conv2d_msra(image0, 16, 3, 3, 2, 2, "e0"))
        e1 = t.relu(t.conv2d_msra(e0, 32, 3, 3, 2, 2, "e1"))
        e2 = t.relu(t.conv2d_msra(e1, 64, 3, 3, 2, 2, "e2"))
        e3 = t.relu(t.conv2d_msra(e2, 128, 3, 3, 2, 2, "e3"))
        e4 = t.relu(t.conv2d_msra(e3, 256, 3, 3, 2, 2, "e4"))
        e4r = t.reshape(e4, [B, 4096])
        e_fc0 = t.relu(t.linear_msra(e4r, 2048, "e_fc0"))
        e_fc1 = t.relu(t.linear_msra(e_fc0, 2048, "e_fc1"))
        concated = t.concat(1, [e_fc1, _decode_angle(t, disp, variant)])
        d_fc0 = t.relu(t.linear_msra(concated, 2048, "a3"))
        d_fc1 = t.relu(t.linear_msra(d_fc0, 2048, "a4"))
        dr = t.reshape(d_fc1, [B, 8, 8, 32])
        d3 = t.relu(t.deconv2d_msra(dr, [B, 16, 16, 128], 3, 3, 2, 2, "d3"))
        d2 = t.relu(t.deconv2d_msra(d3, [B, 32, 32, 64], 3, 3, 2, 2, "d2"))
        d1 = t.relu(t.deconv2d_msra(d2, [B, 64, 64, 32], 3, 3, 2, 2, "d1"))
        d0 = t.relu(t.deconv2d_msra(d1, [B, 128, 128, 16], 3, 3, 2, 2, "d0"))
        flow = t.deconv2d_msra(d0, [B, 128, 128, 2], 3, 3, 1, 1, "flow_field")
    else:
        e0 = t.lrelu(t.conv2d_msra(image0, 32, 5, 5, 2, 2, "e0"))
        e0_0 = t.lrelu(t.conv2d_msra(e0, 32, 5, 5, 1, 1, "e0_0"))
        e1 = t.lrelu(t.conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
        e1_0 = t.lrelu(t.conv2d_msra(e1, 32, 5, 5, 1, 1, "e1_0"))
        e2 = t.lrelu(t.conv2d_msra(e1_0, 64, 5, 5, 2, 2, "e2"))
        e2_0 = t.lrelu(t.conv2d_msra(e2, 64, 5, 5, 1, 1, "e2_0"))
        e3 = t.lrelu(t.conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
        e3_0 = t.lrelu(t.conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
        e4 = t.lrelu(t.conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
        e4_0 = t.lrelu(t.conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))
        e4r = t.reshape(e4_0, [B, 4096])
        e5 = t.lrelu(t.linear_msra(e4r, 4096, "fc1"))
        concated = t.concat(1, [e5, _decode_angle(t, disp, variant)])
        a3 = t.lrelu(t.linear_msra(concated, 4096, "a3"))
        a4 = t.lrelu(t.linear_msra(a3, 4096, "a4"))
        a5 = t.lrelu(t.linear_msra(a4, 4096, "a5"))
        a5r = t.reshape(a5, [B, 4, 4, 256])
        d4 = t.lrelu(t.deconv2d_msra(a5r, [B, 8, 8, 128], 3, 3, 2, 2, "d4"))
        d4_0 = t.lrelu(t.conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
        d3 = t.lrelu(t.deconv2d_msra(d4_0, [B, 16, 16, 64], 3, 3, 2, 2, "d3"))
        d3_0 = t.lrelu(t.conv2d_msra(d3, 64, 5, 5, 1, 1, "d3_0"))
        d2 = t.lrelu(t.deconv2d_msra(d3_0, [B, 32, 32, 32], 5, 5, 2, 2, "d2"))
        d2_0 = t.lrelu(t.conv2d_msra(d2, 64, 5, 5, 1, 1, "d2_0"))
        d1 = t.lrelu(t.deconv2d_msra(d2_0, [B, 64, 64, 32], 5, 5, 2, 2, "d1"))
        d1_0 = t.lrelu(t.conv2d_msra(d1, 32, 5, 5, 1, 1, "d1_0"))
        flow = t.deconv2d_msra(d1_0, [B, 128, 128, 2], 5, 5, 2, 2, "flow_field")
    warp = t.warp_pts_layer(flow)
    gen = t.resample_layer(image0, warp)
    out.update(flow_field=flow, warp_pts=warp, gen=gen)
    if build_loss:
        out['loss'] = t.euclidean_loss(gen, image1)      # appearance_flow_model.py:73
    return out


# --------------------------------------------------------------------------- Base_Prediction_Model
def _image_preprocessing(t, inp, scope):
    """main_model.py:57-65 / multiobject_appflow.py:80-88."""
    with t.variable_scope(scope):
        e0 = t.lrelu(t.conv2d_msra(inp, 32, 5, 5, 2, 2, "e0"))
        e0_0 = t.lrelu(t.conv2d_msra(e0, 32, 5, 5, 1, 1, "e0_0"))
        e1 = t.lrelu(t.conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
        e1_0 = t.lrelu(t.conv2d_msra(e1, 32, 5, 5, 1, 1, "e1_0"))
        e2 = t.lrelu(t.conv2d_msra(e1_0, 64, 5, 5, 2, 2, "e2"))
    return e2


def _decode_trunk(t, inp, B, H):
    d2 = t.lrelu(t.deconv2d_msra(inp, [B, H // 4, H // 4, 32], 5, 5, 2, 2, "d2"))
    d2_0 = t.lrelu(t.conv2d_msra(d2, 64, 5, 5, 1, 1, "d2_0"))
    d1 = t.lrelu(t.deconv2d_msra(d2_0, [B, H // 2, H // 2, 32], 5, 5, 2, 2, "d1"))
    return t.lrelu(t.conv2d_msra(d1, 32, 5, 5, 1, 1, "d1_0"))


def _decode_direct(t, inp, scope, num_channels, B, H=128):
    """main_model.py:68-81 (num_channels) / multiobject_appflow.py:106-120 (always 1 channel)."""
    with t.variable_scope(scope):
        d1_0 = _decode_trunk(t, inp, B, H)
        pre = t.deconv2d_msra(d1_0, [B, H, H, num_channels], 5, 5, 2, 2, "d0")
        return t.tanh(pre)


def _decode_flow(t, src, inp, scope, B, H=128):
    """multiobject_appflow.py:90-104."""
    with t.variable_scope(scope):
        d1_0 = _decode_trunk(t, inp, B, H)
        flow = t.deconv2d_msra(d1_0, [B, H, H, 2], 5, 5, 2, 2, "d0")
        return t.resample_layer(src, t.warp_pts_layer(flow))


def _fc_bottleneck(t, e4_0, a2, B):
    e4r = t.reshape(e4_0, [B, 4096])
    e5 = t.lrelu(t.linear_msra(e4r, 4096, "fc1"))
    concated = t.concat(1, [e5, a2])
    a3 = t.lrelu(t.linear_msra(concated, 4096, "a3"))
    a4 = t.lrelu(t.linear_msra(a3, 4096, "a4"))
    a5 = t.lrelu(t.linear_msra(a4, 4096, "a5"))
    return t.reshape(a5, [B, 4, 4, 256])


def base_prediction(t, conf, image0, image1, dimage0, dimage1, disp, build_loss=True):
    """Base_Prediction_Model.buildModel/build_loss (main_model.py:83-162)."""
    B = disp.v.shape[0]
    out = OrderedDict()
    concat_list = []
    if 'use_color' in conf:
        concat_list.append(_image_preprocessing(t, image0, 'pre_image0'))
    if 'use_depth' in conf:
        concat_list.append(_image_preprocessing(t, dimage0, 'pre_dimage0'))
    comb_enc = t.concat(3, concat_list)
    e2_0 = t.lrelu(t.conv2d_msra(comb_enc, 64, 5, 5, 1, 1, "e2_0"))
    e3 = t.lrelu(t.conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
    e3_0 = t.lrelu(t.conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
    e4 = t.lrelu(t.conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
    e4_0 = t.lrelu(t.conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))
    e4r = t.reshape(e4_0, [B, 4096])
    e5 = t.lrelu(t.linear_msra(e4r, 4096, "fc1"))
    a0 = t.lrelu(t.linear_msra(disp, 64, "a0"))
    a1 = t.lrelu(t.linear_msra(a0, 64, "a1"))
    a2 = t.lrelu(t.linear_msra(a1, 64, "a2"))
    concated = t.concat(1, [e5, a2])
    a3 = t.lrelu(t.linear_msra(concated, 4096, "a3"))
    a4 = t.lrelu(t.linear_msra(a3, 4096, "a4"))
    a5 = t.lrelu(t.linear_msra(a4, 4096, "a5"))
    a5r = t.reshape(a5, [B, 4, 4, 256])
    d4 = t.lrelu(t.deconv2d_msra(a5r, [B, 8, 8, 128], 3, 3, 2, 2, "d4"))
    d4_0 = t.lrelu(t.conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
    d3 = t.lrelu(t.deconv2d_msra(d4_0, [B, 16, 16, 64], 3, 3, 2, 2, "d3"))
    num_decode = ('use_color' in conf) + ('use_depth' in conf)
    d3_0 = t.lrelu(t.conv2d_msra(d3, 64 * num_decode, 5, 5, 1, 1, "d3_0"))
    split_list = t.split(d3_0, num_decode, 3)
    if 'use_color' in conf:
        out['gen_image1'] = _decode_direct(t, split_list.pop(), 'dec_image1', 3, B)
    if 'use_depth' in conf:
        out['gen_dimage1'] = _decode_direct(t, split_list.pop(), 'dec_dimage1', 1, B)
    assert split_list == []
    if build_loss:
        loss = None
        if 'use_color' in conf:
            loss = t.euclidean_loss(out['gen_image1'], image1)
        if 'use_depth' in conf:
            dl = t.scale(t.euclidean_loss(out['gen_dimage1'], dimage1), conf['depth_lr_factor'])
            loss = dl if loss is None else t.add(loss, dl)
        out['loss'] = loss
    return out


# --------------------------------------------------------------------------- mv3d (direct prediction, the original models)
def mv3d(t, variant, images1, images2, labels, build_loss=True):
    """mv3d.buildModel of dyn_mult_view/mv3d/nobg_nodm.py:31-95, nobg_dm.py:30-96, bg_nodm.py:30-98: encoder ->
    fc bottleneck joined with the 5-d view label -> decoder -> tanh image (+ depth map / + silhouette mask)."""
    B = labels.v.shape[0]
    bg = variant == 'bg_nodm'
    k5 = (3, 3) if bg else (5, 5)            # bg_nodm swaps the 5x5 stride-1 convs for 3x3 ones and renames them *_1
    sfx = '_1' if bg else '_0'
    e0 = t.lrelu(t.conv2d_msra(images1, 16 if bg else 32, 5, 5, 2, 2, "e0"))
    e0_0 = t.lrelu(t.conv2d_msra(e0, 32, k5[0], k5[1], 1, 1, "e0" + sfx))
    e1 = t.lrelu(t.conv2d_msra(e0_0, 32, 5, 5, 2, 2, "e1"))
    e1_0 = t.lrelu(t.conv2d_msra(e1, 32, k5[0], k5[1], 1, 1, "e1" + sfx))
    e2 = t.lrelu(t.conv2d_msra(e1_0, 64, k5[0], k5[1], 2, 2, "e2"))
    e2_0 = t.lrelu(t.conv2d_msra(e2, 64, k5[0], k5[1], 1, 1, "e2" + sfx))
    e3 = t.lrelu(t.conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
    e3_0 = t.lrelu(t.conv2d_msra(e3, 128, 3, 3, 1, 1, "e3" + sfx))
    e4 = t.lrelu(t.conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
    e4_0 = t.lrelu(t.conv2d_msra(e4, 256, 3, 3, 1, 1, "e4" + sfx))
    e5 = t.lrelu(t.linear_msra(t.reshape(e4_0, [B, 4096]), 4096, "fc1"))
    a0 = t.lrelu(t.linear_msra(labels, 64, "a0"))
    a1 = t.lrelu(t.linear_msra(a0, 64, "a1"))
    a2 = t.lrelu(t.linear_msra(a1, 64, "a2"))
    a3 = t.lrelu(t.linear_msra(t.concat(1, [e5, a2]), 4096, "a3"))
    a4 = t.lrelu(t.linear_msra(a3, 4096, "a4"))
    a5 = t.lrelu(t.linear_msra(a4, 4096, "a5"))
    d4 = t.lrelu(t.deconv2d_msra(t.reshape(a5, [B, 4, 4, 256]), [B, 8, 8, 128], 3, 3, 2, 2, "d4"))
    d4_0 = t.lrelu(t.conv2d_msra(d4, 128, 3, 3, 1, 1, "d4" + sfx))
    d3 = t.lrelu(t.deconv2d_msra(d4_0, [B, 16, 16, 64], 3, 3, 2, 2, "d3"))
    d3_0 = t.lrelu(t.conv2d_msra(d3, 64, k5[0], k5[1], 1, 1, "d3" + sfx))
    d2 = t.lrelu(t.deconv2d_msra(d3_0, [B, 32, 32, 32], 5, 5, 2, 2, "d2"))
    d2_0 = t.lrelu(t.conv2d_msra(d2, 32 if bg else 64, k5[0], k5[1], 1, 1, "d2" + sfx))
    d1 = t.lrelu(t.deconv2d_msra(d2_0, [B, 64, 64, 32], 5, 5, 2, 2, "d1"))
    d1_0 = t.lrelu(t.conv2d_msra(d1, 32, k5[0], k5[1], 1, 1, "d1" + sfx))
    nout = 3 if variant == 'nobg_nodm' else 4
    if bg:
        d0 = t.lrelu(t.deconv2d_msra(d1_0, [B, 128, 128, 16], 5, 5, 2, 2, "d0"))
        gen = t.tanh(t.conv2d_msra(d0, 4, 3, 3, 1, 1, "d0_1"))
    else:
        gen = t.tanh(t.deconv2d_msra(d1_0, [B, 128, 128, nout], 5, 5, 2, 2, "d0"))
    out = OrderedDict(gen=gen)
    if build_loss:
        if variant == 'nobg_nodm':
            out['loss'] = t.euclidean_loss(gen, images2)
        else:
            gt_cm, gt_x = t.split(images2, [3, 1], 3)
            pr_cm, pr_x = t.split(gen, [3, 1], 3)
            if variant == 'nobg_dm':
                out['loss'] = t.add(t.euclidean_loss(gt_cm, pr_cm), t.scale(t.l1_loss(gt_x, pr_x), 0.1))
            else:
                sm = gt_x
                out['loss'] = t.add(t.euclidean_loss(t.multiply(gt_cm, sm), t.multiply(pr_cm, sm)),
                                    t.scale(t.euclidean_loss(t.scale(gt_x, 0.75), pr_x), 0.1))
    return out


def mv3d_builder(variant, build_loss=True):
    return lambda t, n: mv3d(t, variant, n['images1'], n['images2'], n['labels'], build_loss)


# --------------------------------------------------------------------------- MultiObjectAppFlow
def multiobject_appflow(t, conf, inp, build_loss=True, direct_color=False):
    """MultiObjectAppFlow.buildModel/build_loss (multiobject_appflow.py:123-286); direct_color=True: the sibling
    multiobject_main_model.Base_Prediction_Model (multiobject_main_model.py:106-270), whose colour outputs come from
    3-channel tanh decoders instead of appearance-flow decoders -- everything else is shared.
    `inp` maps the 13 reader attribute names (multiobject_appflow.py:31-43) to Nodes."""
    B = inp['displacement'].v.shape[0]
    H = inp['image0'].v.shape[1]
    out = OrderedDict()
    concat_list = []
    if 'use_color' in conf:
        concat_list.append(_image_preprocessing(t, inp['image0'], 'pre_image0_f'))
    if 'use_depth' in conf:
        concat_list.append(_image_preprocessing(t, inp['depth0'], 'pre_dimage0_f'))
    concat_list.append(_image_preprocessing(t, inp['image0_mask0'], 'pre_mask0_ob0'))
    concat_list.append(_image_preprocessing(t, inp['image0_mask1'], 'pre_mask0_ob1'))
    comb_enc = t.concat(3, concat_list)
    e2_0 = t.lrelu(t.conv2d_msra(comb_enc, 64, 5, 5, 1, 1, "e2_0"))
    e3 = t.lrelu(t.conv2d_msra(e2_0, 128, 3, 3, 2, 2, "e3"))
    e3_0 = t.lrelu(t.conv2d_msra(e3, 128, 3, 3, 1, 1, "e3_0"))
    e4 = t.lrelu(t.conv2d_msra(e3_0, 256, 3, 3, 2, 2, "e4"))
    e4_0 = t.lrelu(t.conv2d_msra(e4, 256, 3, 3, 1, 1, "e4_0"))
    a0 = t.lrelu(t.linear_msra(inp['displacement'], 64, "a0"))
    a1 = t.lrelu(t.linear_msra(a0, 64, "a1"))
    a2 = t.lrelu(t.linear_msra(a1, 64, "a2"))
    if 'fully_conv' in conf:
        hb, wb = e4_0.v.shape[1], e4_0.v.shape[2]
        smear = t.reshape(a2, [B, 1, 1, a2.v.shape[1]])
        smear = t.tile(smear, [1, hb, wb, 1])
        concated = t.concat(3, [e4_0, smear])
        e4_1 = t.lrelu(t.conv2d_msra(concated, 256, 3, 3, 1, 1, "e4_1"))
        a5r = t.lrelu(t.conv2d_msra(e4_1, 256, 3, 3, 1, 1, "e4_2"))
    else:
        a5r = _fc_bottleneck(t, e4_0, a2, B)
    hb = a5r.v.shape[1]
    d4 = t.lrelu(t.deconv2d_msra(a5r, [B, 2 * hb, 2 * hb, 128], 3, 3, 2, 2, "d4"))
    d4_0 = t.lrelu(t.conv2d_msra(d4, 128, 3, 3, 1, 1, "d4_0"))
    d3 = t.lrelu(t.deconv2d_msra(d4_0, [B, 4 * hb, 4 * hb, 64], 3, 3, 2, 2, "d3"))
    num_decode = 0
    for key in ('use_color', 'use_depth'):
        if key in conf:
            num_decode += ('combination_image' in conf) + 2 * ('gen_sep_images' in conf)
    if 'predict_target_masks' in conf:
        num_decode += 2
    d3_0 = t.lrelu(t.conv2d_msra(d3, 64 * num_decode, 5, 5, 1, 1, "d3_0"))
    split_list = t.split(d3_0, num_decode, 3)
    if 'use_color' in conf:
        if 'combination_image' in conf:
            out['gen_image1'] = (_decode_direct(t, split_list.pop(), 'dec_image1', 3, B, H) if direct_color else _decode_flow(t, inp['image0'], split_list.pop(), 'dec_image1', B, H))
        if 'gen_sep_images' in conf:
            out['gen_image1_only0'] = (_decode_direct(t, split_list.pop(), 'dec_image1_only0', 3, B, H) if direct_color else _decode_flow(t, inp['image0'], split_list.pop(), 'dec_image1_only0', B, H))
            out['gen_image1_only1'] = (_decode_direct(t, split_list.pop(), 'dec_image1_only1', 3, B, H) if direct_color else _decode_flow(t, inp['image0'], split_list.pop(), 'dec_image1_only1', B, H))
    if 'use_depth' in conf:
        if 'combination_image' in conf:
            out['gen_depth1'] = _decode_direct(t, split_list.pop(), 'dec_dimage1_f', 1, B, H)
        if 'gen_sep_images' in conf:
            out['gen_depth1_only0'] = _decode_direct(t, split_list.pop(), 'dec_depth1_only0', 1, B, H)
            out['gen_depth1_only1'] = _decode_direct(t, split_list.pop(), 'dec_depth1_only1', 1, B, H)
    if 'predict_target_masks' in conf:
        out['gen_image1_mask0'] = _decode_direct(t, split_list.pop(), 'dec_image1_mask0', 1, B, H)
        out['gen_image1_mask1'] = _decode_direct(t, split_list.pop(), 'dec_image1_mask1', 1, B, H)
    assert split_list == []
    if not build_loss:
        return out

    terms = []

    def sep_loss(gen0, gen1, tgt0, tgt1, factor):
        # multiobject_appflow.py:237-246, 257-266
        if 'masked_image_loss' in conf:
            l0 = t.masked_euclidean_loss(out[gen0], inp[tgt0], inp['image1_mask0'])
            l1 = t.masked_euclidean_loss(out[gen1], inp[tgt1], inp['image1_mask1'])
        else:
            l0 = t.euclidean_loss(out[gen0], inp[tgt0])
            l1 = t.euclidean_loss(out[gen1], inp[tgt1])
        if factor is not None:
            l0, l1 = t.scale(l0, factor), t.scale(l1, factor)
        terms.extend([l0, l1])

    if 'use_color' in conf:
        if 'combination_image' in conf:
            terms.append(t.euclidean_loss(out['gen_image1'], inp['image1']))
        if 'gen_sep_images' in conf:
            sep_loss('gen_image1_only0', 'gen_image1_only1', 'image1_only0', 'image1_only1', None)
    if 'use_depth' in conf:
        depth_factor = conf['use_depth']
        if 'combination_image' in conf:
            terms.append(t.euclidean_loss(out['gen_depth1'], inp['depth1']))   # not scaled: multiobject_appflow.py:254
        if 'gen_sep_images' in conf:
            sep_loss('gen_depth1_only0', 'gen_depth1_only1', 'depth1_only0', 'depth1_only1', depth_factor)
    if 'predict_target_masks' in conf:
        mf = conf['predict_target_masks']
        terms.append(t.scale(t.euclidean_loss(out['gen_image1_mask0'], inp['image1_mask0']), mf))
        terms.append(t.scale(t.euclidean_loss(out['gen_image1_mask1'], inp['image1_mask1']), mf))
    loss = terms[0]
    for l in terms[1:]:
        loss = t.add(loss, l)
    out['loss'] = loss
    return out


# --------------------------------------------------------------------------- train step
class AdamState:
    """Slots of tf.train.AdamOptimizer (Appendix A.7): m, v per variable WITH a gradient,
    beta1_power / beta2_power accumulators."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m, self.v = {}, {}
        self.beta1_power = np.float32(beta1)
        self.beta2_power = np.float32(beta2)

    def apply(self, variables, grads):
        for name, g in grads.items():
            p = variables[name]
            if name not in self.m:
                self.m[name] = np.zeros_like(p)
                self.v[name] = np.zeros_like(p)
            ops.adam_step(p, g.astype(p.dtype), self.m[name], self.v[name], self.beta1_power, self.beta2_power,
                          self.lr, self.beta1, self.beta2, self.eps)
        self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))


def run(builder, variables, feeds, dtype=np.float32, rng=None, backward=True, sign_override=None, warp_override=None, **kw):
    """Evaluate one graph: returns (outputs {name: array}, grads {var: array}, tape)."""
    t = Tape(variables, rng=rng, dtype=dtype, sign_override=sign_override, warp_override=warp_override)
    nodes = {k: t.const(v) for k, v in feeds.items()}
    out = builder(t, nodes, **kw)
    grads = t.backward(out['loss']) if (backward and 'loss' in out) else {}
    return OrderedDict((k, n.v) for k, n in out.items()), grads, t


def appearance_flow_builder(variant='base', build_loss=True):
    return lambda t, n: appearance_flow(t, n['image0'], n['image1'], n['disp'], variant, build_loss)


def base_prediction_builder(conf, build_loss=True):
    return lambda t, n: base_prediction(t, conf, n.get('image0'), n.get('image1'), n.get('dimage0'),
                                        n.get('dimage1'), n['disp'], build_loss)


def multiobject_builder(conf, build_loss=True, direct_color=False):
    return lambda t, n: multiobject_appflow(t, conf, n, build_loss, direct_color)


def step(builder, variables, adam, feeds):
    """One train.py:122 iteration on the oracle: returns (loss, outputs)."""
    out, grads, _ = run(builder, variables, feeds)
    adam.apply(variables, grads)
    return float(out['loss']), out
