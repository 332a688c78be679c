"""Golden vectors for GridMask (reference common/data_utils.py:276-361) made with the REAL Pillow: the rotated, cropped,
inverted mask of Grid.__call__ for a set of (h, w, d, st_h, st_w, r) draws, written by the reference's own statements with
PIL.Image.rotate doing the rotation.  Run here (Pillow is in this image):  python tests/golden/make_pil_gridmask.py
-> tests/golden/pil_gridmask.npz (masks bit-packed)."""
import math
import os

import numpy as np
from PIL import Image
import PIL


def pil_mask(h, w, d, st_h, st_w, r, ratio=0.5):
    hh = math.ceil(math.sqrt(h * h + w * w))
    l = math.ceil(d * ratio)
    mask = np.ones((hh, hh), np.float32)
    for i in range(-1, hh // d + 1):
        s = d * i + st_h
        t = s + l
        s = max(min(s, hh), 0)
        t = max(min(t, hh), 0)
        mask[s:t, :] *= 0
    for i in range(-1, hh // d + 1):
        s = d * i + st_w
        t = s + l
        s = max(min(s, hh), 0)
        t = max(min(t, hh), 0)
        mask[:, s:t] *= 0
    mask = Image.fromarray(np.uint8(mask))
    mask = mask.rotate(r)
    mask = np.asarray(mask)
    mask = mask[(hh - h) // 2:(hh - h) // 2 + h, (hh - w) // 2:(hh - w) // 2 + w]
    return 1 - mask


def main():
    rng = np.random.default_rng(20260403)
    cases, packed = [], []
    for (h, w) in ((513, 513), (320, 480), (65, 97), (33, 33)):
        for trial in range(8):
            d = int(rng.integers(max(2, w // 7), max(3, w // 3)))
            st_h, st_w = int(rng.integers(d)), int(rng.integers(d))
            r = [0, 90, 180, 270][trial] if trial < 4 else int(rng.integers(360))
            m = pil_mask(h, w, d, st_h, st_w, r)
            cases.append((h, w, d, st_h, st_w, r))
            packed.append(np.packbits(m.reshape(-1)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pil_gridmask.npz')
    np.savez_compressed(out, cases=np.asarray(cases, np.int32), pillow=np.asarray(PIL.__version__),
                        **{'m%d' % i: p for i, p in enumerate(packed)})
    print(out, len(cases), 'cases, Pillow', PIL.__version__)


if __name__ == '__main__':
    main()
