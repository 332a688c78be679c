"""Golden vectors for the PIL ImageEnhance adjustments the reference's generator applies (common/data_utils.py:83-239:
random_brightness / random_chroma / random_contrast / random_sharpness are `ImageEnhance.X(Image.fromarray(image))
.enhance(factor)` with factor drawn from rand(jitter, 1/jitter), jitter = 0.5).  The reference module itself cannot be
imported here (it imports cv2 on line 7), so this script makes the same PIL calls on seeded images; run in the build
container (Pillow 12.2.0):  python tests/golden/make_pil_enhance.py  ->  tests/golden/pil_enhance.npz"""
import os

import numpy as np
from PIL import Image, ImageEnhance

rng = np.random.default_rng(20260)
imgs = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8),
        # smooth gradients + saturated patches: blend results near 0 / 255 and the truncation boundaries
        np.clip(np.add.outer(np.arange(48) * 5, np.arange(40) * 6)[..., None] + np.array([0, 40, -60]), 0, 255).astype(np.uint8),
        np.full((5, 7, 3), 255, np.uint8), np.zeros((4, 4, 3), np.uint8)]
factors = [0.5, 0.7312, 1.0, 1.25, 1.9999, 2.0, 0.0]
ENH = [ImageEnhance.Brightness, ImageEnhance.Color, ImageEnhance.Contrast, ImageEnhance.Sharpness]
out = {'factors': np.array(factors, np.float64)}
for i, im in enumerate(imgs):
    out['img%d' % i] = im
    for op, E in enumerate(ENH):
        for j, f in enumerate(factors):
            out['out%d_op%d_f%d' % (i, op, j)] = np.asarray(E(Image.fromarray(im)).enhance(f))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pil_enhance.npz'), **out)
print('wrote', len(out), 'arrays')
