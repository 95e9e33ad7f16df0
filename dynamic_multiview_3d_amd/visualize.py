"""`visualize()` outputs of the model classes: the files the reference writes for qualitative checks.

Mirrors multi_view_model/appearance_flow_model.py:132-179 (image grids, quiver of `warp_pts`, correspondence plot),
main_model.py:165-207 (colour / depth grids), multiobject_appflow.py:289-395 and multiobject_main_model.py:272-380
(clipped tensors pickled to `imgdata.pkl`), and the helpers `save_images / rescale_image / rescale_dm` of
mv3d/utils/tf_utils.py:101-147.  Host-side only: one forward pass of the model (the HIP path), then numpy / PIL /
matplotlib on the fetched arrays.  `scipy.misc.toimage` (removed from scipy) is restated for the two ways it is called.
"""
import math
import os
import pickle
import re

import numpy as np


def rescale_image(image):
    """tf_utils.py:140-142"""
    return (image / 1.5 + 0.5) * 255


def rescale_dm(image):
    """tf_utils.py:145-147"""
    return (image / 1.5 + 0.5) * 65535


def image_grid(images, size, color=True):
    """The tiling of tf_utils.save_images (tf_utils.py:101-117): image idx goes to row idx // size[1], column idx % size[1]."""
    images = np.asarray(images)
    h, w = images.shape[1], images.shape[2]
    img = np.zeros((h * size[0], w * size[1], 3) if color else (h * size[0], w * size[1]))
    for idx, image in enumerate(images[:size[0] * size[1]]):
        i = idx % size[1]
        j = int(math.floor(idx / size[1]))
        img[j * h:j * h + h, i * w:i * w + w] = image
    return img


def grid_to_uint8(img):
    """scipy.misc.toimage(rescale_image(img), cmin=0, cmax=255): bytescale with unit scale = clip to [0, 255], round half up."""
    return (np.clip(rescale_image(img), 0, 255) + 0.5).astype(np.uint8)


def grid_to_uint16(img):
    """scipy.misc.toimage(rescale_dm(img), cmin=0, cmax=65535, low=0, high=65535, mode='I'): unit scale, then an integer
    cast (toimage casts to uint32 and writes mode 'I', which PNG stores as 16 bits; values are clipped here instead of
    wrapping)."""
    return np.clip(rescale_dm(img), 0, 65535).astype(np.uint16)


def save_images(images, size, image_path, color=True):
    """tf_utils.py:101-124"""
    from PIL import Image
    img = image_grid(images, size, color)
    if color:
        Image.fromarray(grid_to_uint8(img), 'RGB').save(image_path)
    else:
        Image.fromarray(grid_to_uint16(img)).save(image_path)


def _iter_num(conf):
    return re.match('.*?([0-9]+)$', conf['visualize']).group(1)


def _fetch(model, names):
    out = {}
    for n in names:
        t = getattr(model, n, None)
        if t is not None and hasattr(t, 'numpy'):
            out[n] = t.numpy()
    return out


def _run(model, feeds):
    loss = model.forward(**feeds)
    if model.graph.loss_expr is not None:
        print('loss', float(loss))


def visualize_appearance_flow(model, sess=None, **feeds):
    """appearance_flow_model.py:132-179"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from matplotlib.patches import ConnectionPatch
    _run(model, feeds)
    f = _fetch(model, ['image0', 'image1', 'gen', 'warp_pts'])
    image0, image1, gen, warp_pts = f['image0'], f['image1'], f['gen'], f['warp_pts']
    print('max resample coord:', np.max(warp_pts))
    iter_num = _iter_num(model.conf)
    path = model.conf['output_dir']
    os.makedirs(path, exist_ok=True)
    save_images(gen, [8, 8], path + "/output_%s.png" % iter_num)
    save_images(image1, [8, 8], path + '/tr_gt_%s.png' % iter_num)
    save_images(image0, [8, 8], path + '/tr_input_%s.png' % iter_num)

    plt.figure()
    plt.axes([0, 0.025, 0.95, 0.95])
    plt.quiver(warp_pts[0, :, :, 0], warp_pts[0, :, :, 1])
    plt.savefig(path + '/quiver_%s.pdf' % iter_num)
    plt.close()

    plt.figure()
    ax1 = plt.subplot(121)
    ax2 = plt.subplot(122)
    ax1.imshow(np.clip(image0[0], 0, 1))
    ax2.imshow(np.clip(gen[0], 0, 1))
    H = image0.shape[1]
    rng = np.random.RandomState(0)
    for pt_output in rng.randint(int(H * 0.3125), int(H * 0.6875), size=(6, 2)):         # 40..88 at 128 pixels
        sampled_location = np.clip(warp_pts[0, pt_output[0], pt_output[1], :], 0, None).astype('uint32')
        ax2.add_artist(ConnectionPatch(xyA=np.flip(pt_output, 0), xyB=np.flip(sampled_location, 0), coordsA="data", coordsB="data",
                                       axesA=ax2, axesB=ax1, arrowstyle="<->", shrinkB=5))
    for ax in (ax1, ax2):
        ax.set_xlim(0, H)
        ax.set_ylim(0, H)
    plt.savefig(path + '/corr_plot_%s.pdf' % iter_num)
    plt.close()
    return f


def visualize_prediction(model, sess=None, **feeds):
    """main_model.py:165-207"""
    _run(model, feeds)
    conf = model.conf
    iter_num = _iter_num(conf)
    path = conf['output_dir']
    os.makedirs(path, exist_ok=True)
    f = _fetch(model, ['image0', 'image1', 'gen_image1', 'dimage0', 'dimage1', 'gen_dimage1'])
    if 'use_color' in conf:
        save_images(f['gen_image1'], [8, 8], path + "/output_%s.png" % iter_num)
        save_images(f['image1'], [8, 8], path + '/tr_gt_%s.png' % iter_num)
        save_images(f['image0'], [8, 8], path + '/tr_input_%s.png' % iter_num)
    if 'use_depth' in conf:
        save_images(np.squeeze(f['gen_dimage1'], -1), [8, 8], path + "/depth_output_%s.png" % iter_num, color=False)
        save_images(np.squeeze(f['dimage1'], -1), [8, 8], path + '/depth_tr_gt_%s.png' % iter_num, color=False)
        save_images(np.squeeze(f['dimage0'], -1), [8, 8], path + '/depth_tr_input_%s.png' % iter_num, color=False)
    return f


MULTIOBJECT_TENSORS = ('image0', 'image0_mask0', 'image0_mask1', 'image1', 'image1_only0', 'image1_only1', 'image1_mask0',
                       'image1_mask1', 'depth0', 'depth1', 'depth1_only0', 'depth1_only1', 'gen_image1', 'gen_image1_only0',
                       'gen_image1_only1', 'gen_image1_mask0', 'gen_image1_mask1', 'gen_depth1', 'gen_depth1_only0',
                       'gen_depth1_only1')


def visualize_multiobject(model, sess=None, **feeds):
    """multiobject_appflow.py:289-395: every input / output tensor, clipped to [0, 1], pickled to output_dir/imgdata.pkl
    (the reference fetches all twenty and therefore needs every decoder enabled; here absent ones are left out)."""
    _run(model, feeds)
    d = {k: np.clip(v, 0., 1.) for k, v in _fetch(model, MULTIOBJECT_TENSORS).items()}
    os.makedirs(model.conf['output_dir'], exist_ok=True)
    file = model.conf['output_dir'] + '/imgdata.pkl'
    with open(file, 'wb') as fh:
        pickle.dump(d, fh, protocol=2)
    print('written to file', file)
    return d
