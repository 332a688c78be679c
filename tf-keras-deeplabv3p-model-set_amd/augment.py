"""Device-side counterparts of the byte-level augmentations in the reference's generator (deeplabv3p/data.py:72-104,
common/data_utils.py).  Same names and argument meaning as the reference functions, applied to a BATCH that already
sits on the GPU as bytes: images (N,H,W,3) uint8, labels (N,H,W) uint8 torch tensors.  The random draws are made on the
host exactly as the reference makes them (`rand() < prob` per image, `rand(jitter, 1/jitter)` per image); the kernels
(csrc/augment.hip) reproduce PIL / NumPy bit for bit for a given draw.

random_gridmask is here too: its rotation is PIL's (Image.rotate, NEAREST), restated from Pillow's 16.16 fixed-point affine
map and pinned against PIL itself.

random_grayscale and random_blur are here as UNPINNED restatements of OpenCV's published 8-bit arithmetic (cv2 is not in this image:
the kernels are held to oracle/np_augment.py's restatement of the same formulas, not to OpenCV itself).

Not here: random_zoom_rotate, random_histeq and the final cv2.resize -- OpenCV code (warpAffine's fixed-point interpolation tables,
CLAHE) that cannot be pinned without OpenCV; they stay on the host.
"""
import numpy as np
import torch

from ._lib import lib

BRIGHTNESS, COLOR, CONTRAST, SHARPNESS = 0, 1, 2, 3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def rand(a=0, b=1):
    return np.random.rand() * (b - a) + a


def enhance(images, op, factors, out=None):
    """PIL ImageEnhance.{Brightness, Color, Contrast, Sharpness}(img).enhance(factors[n]) for every image of the batch"""
    N, H, W, C = images.shape
    assert images.dtype == torch.uint8 and C == 3 and images.is_contiguous()
    f = torch.as_tensor(np.asarray(factors, np.float32).reshape(N)).to(images.device)
    out = torch.empty_like(images) if out is None else out
    sums = torch.zeros(N, dtype=torch.int64, device=images.device) if op == CONTRAST else None
    lib().aug_enhance_u8(images.data_ptr(), out.data_ptr(), f.data_ptr(), op, sums.data_ptr() if sums is not None else None,
                         N, H, W, _stream())
    return out


def flip_crop(images, labels, flags=None, yx=None, hw=None):
    """flags[n] bit 0 / 1 = horizontal / vertical flip; yx[n] = top-left of an hw = (h, w) window of the flipped image"""
    N, H, W, _ = images.shape
    assert images.dtype == torch.uint8 and labels.dtype == torch.uint8 and labels.shape == (N, H, W)
    h, w = (H, W) if hw is None else hw
    dev = images.device
    fl = None if flags is None else torch.as_tensor(np.asarray(flags, np.int32).reshape(N)).to(dev)
    off = None if yx is None else torch.as_tensor(np.asarray(yx, np.int32).reshape(N, 2)).to(dev)
    out = torch.empty((N, h, w, 3), dtype=torch.uint8, device=dev)
    lout = torch.empty((N, h, w), dtype=torch.uint8, device=dev)
    lib().aug_flip_crop_u8(images.data_ptr(), out.data_ptr(), labels.data_ptr(), lout.data_ptr(),
                           fl.data_ptr() if fl is not None else None, off.data_ptr() if off is not None else None, N, H, W, h, w,
                           _stream())
    return out, lout


# ---- the reference's function names (common/data_utils.py), batched
def random_horizontal_flip(images, labels, prob=.5):
    return flip_crop(images, labels, [1 if rand() < prob else 0 for _ in range(images.shape[0])])


def random_vertical_flip(images, labels, prob=.5):
    return flip_crop(images, labels, [2 if rand() < prob else 0 for _ in range(images.shape[0])])


def random_brightness(images, jitter=.5):
    return enhance(images, BRIGHTNESS, [rand(jitter, 1 / jitter) for _ in range(images.shape[0])])


def random_chroma(images, jitter=.5):
    return enhance(images, COLOR, [rand(jitter, 1 / jitter) for _ in range(images.shape[0])])


def random_contrast(images, jitter=.5):
    return enhance(images, CONTRAST, [rand(jitter, 1 / jitter) for _ in range(images.shape[0])])


def random_sharpness(images, jitter=.5):
    return enhance(images, SHARPNESS, [rand(jitter, 1 / jitter) for _ in range(images.shape[0])])


def gray_blur(images, flags, out=None):
    """flags[n] bit 0: cv2.cvtColor(BGR2GRAY) + GRAY2BGR; bit 1: cv2.GaussianBlur(image, (5, 5), 0) (grayscale first, as the generator
    applies them)"""
    N, H, W, C = images.shape
    assert images.dtype == torch.uint8 and C == 3 and images.is_contiguous()
    fl = torch.as_tensor(np.asarray(flags, np.int32).reshape(N)).to(images.device)
    out = torch.empty_like(images) if out is None else out
    lib().aug_gray_blur_u8(images.data_ptr(), out.data_ptr(), fl.data_ptr(), N, H, W, _stream())
    return out


def random_grayscale(images, prob=.2):
    return gray_blur(images, [1 if rand() < prob else 0 for _ in range(images.shape[0])])


def random_blur(images, prob=.5, size=5):
    if size != 5:
        raise ValueError('the device kernel restates the 5 x 5 table of cv2.GaussianBlur (the generator\'s default size)')
    return gray_blur(images, [2 if rand() < prob else 0 for _ in range(images.shape[0])])


def gridmask_params(h, w, d, st_h, st_w, r, ratio=0.5):
    """the 16 integers dl3p_aug_gridmask_u8 takes for one image, from the reference's draws (Grid.__call__,
    common/data_utils.py:288-339): hh = ceil(sqrt(h^2 + w^2)), band width l = ceil(d ratio), and Pillow's rotate as it
    computes it -- transpose fast paths at 0 / 90 / 180 / 270 degrees, otherwise the affine matrix of -r (entries rounded to 15
    decimals, centre (hh/2, hh/2) fixed) in 16.16 fixed point, FIX(v) = floor(65536 v + 0.5), half-pixel offset folded in"""
    import math
    hh = math.ceil(math.sqrt(h * h + w * w))
    l = math.ceil(d * ratio)
    r = r % 360
    kind = {0: 0, 90: 1, 180: 2, 270: 3}.get(r, 4)
    a0 = a1 = a2 = a3 = a4 = a5 = 0
    if kind == 4:
        ang = -math.radians(r)
        a = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
        c = hh / 2.0
        a[2] = a[0] * -c + a[1] * -c + a[2] + c
        a[5] = a[3] * -c + a[4] * -c + a[5] + c
        fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
        a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
        a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
        a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    return [1, hh, d, l, st_h, st_w, kind, a0, a1, a2, a3, a4, a5, (hh - h) // 2, (hh - w) // 2, 0]


def gridmask(images, labels, draws):
    """apply GridMask IN PLACE; draws[n] = None (image left alone) or (d, st_h, st_w, r)"""
    N, H, W, _ = images.shape
    assert images.dtype == torch.uint8 and labels.dtype == torch.uint8 and labels.shape == (N, H, W)
    assert images.is_contiguous() and labels.is_contiguous()
    rows = [[0] * 16 if dr is None else gridmask_params(H, W, *dr) for dr in draws]
    prm = torch.as_tensor(np.asarray(rows, np.int32)).to(images.device)
    lib().aug_gridmask_u8(images.data_ptr(), labels.data_ptr(), prm.data_ptr(), N, H, W, _stream())
    return images, labels


def random_gridmask(images, labels, prob=0.2):
    """random_gridmask (common/data_utils.py:342-361) for a batch, in place.  Per image, in the reference's order:
    np.random.rand() > prob -> untouched; d = randint(W // 7, W // 3); st_h = randint(d); st_w = randint(d); r = randint(360)"""
    N, H, W, _ = images.shape
    draws = []
    for _ in range(N):
        if np.random.rand() > prob:
            draws.append(None)
            continue
        d = np.random.randint(W // 7, W // 3)
        st_h = np.random.randint(d)
        st_w = np.random.randint(d)
        draws.append((int(d), int(st_h), int(st_w), int(np.random.randint(360))))
    return gridmask(images, labels, draws)


def random_crop(images, labels, crop_shape, prob=.1):
    """the crop branch of the reference's random_crop (common/data_utils.py:364-400) for a batch.  DEVIATION, by
    construction: the reference decides `rand() < prob` per image; a batch tensor needs one output shape, so the decision is
    drawn ONCE for the batch.  Within a cropping batch the window of every image is drawn as the reference draws it -- x
    (`randrange(W - crop_w)`) first, then y (data_utils.py:390-391) -- so a seeded `random` yields the reference's windows.
    Returns the inputs unchanged when the draw says no or the window is not smaller than the image (the reference then
    resizes with cv2, which stays on the host)."""
    import random
    N, H, W, _ = images.shape
    if not (rand() < prob) or not (crop_shape[0] < H and crop_shape[1] < W):
        return images, labels
    yx = []
    for _ in range(N):
        x = random.randrange(W - crop_shape[1])
        y = random.randrange(H - crop_shape[0])
        yx.append((y, x))
    return flip_crop(images, labels, None, yx, tuple(crop_shape))
