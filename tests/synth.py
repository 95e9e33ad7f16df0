"""Seeded synthetic inputs shared by tests (small shapes)."""
import numpy as np


def car_like_batch(rng, b, h=128, c=3):
    """uint8-quantised grey background + one filled ellipse + noise, /255."""
    yy, xx = np.mgrid[0:h, 0:h]
    img = np.full((b, h, h, c), 127.0)
    for i in range(b):
        cy, cx = rng.uniform(0.3 * h, 0.7 * h, 2)
        ay, ax = rng.uniform(0.12 * h, 0.35 * h, 2)
        m = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1
        img[i][m] = rng.uniform(0, 255, c)
    img += rng.normal(0, 2, img.shape)
    return (np.clip(np.rint(img), 0, 255) / 255.0).astype(np.float32)


def appflow_feeds(rng, b, h=128):
    return dict(image0=car_like_batch(rng, b, h), image1=car_like_batch(rng, b, h),
                disp=np.stack([rng.uniform(-1, 1, b), rng.uniform(-6.28, 6.28, b)], 1).astype(np.float32))


def multiobj_feeds(rng, b, h=128):
    f = {}
    for k in ('image0', 'image1', 'image1_only0', 'image1_only1'):
        f[k] = car_like_batch(rng, b, h, 3)
    for k in ('depth0', 'depth1', 'depth1_only0', 'depth1_only1'):
        f[k] = car_like_batch(rng, b, h, 1)
    for k in ('image0_mask0', 'image0_mask1', 'image1_mask0', 'image1_mask1'):
        f[k] = (car_like_batch(rng, b, h, 1) > 0.55).astype(np.float32)
    f['displacement'] = rng.normal(10, 10, (b, 2)).astype(np.float32)
    return f
