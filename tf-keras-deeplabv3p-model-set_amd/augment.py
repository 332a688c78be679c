"""Device-side counterparts of the byte-level augmentations in the reference's generator (deeplabv3p/data.py:72-104,
common/data_utils.py).  Same names and argument meaning as the reference functions, applied to a BATCH that already
sits on the GPU as bytes: images (N,H,W,3) uint8, labels (N,H,W) uint8 torch tensors.  The random draws are made on the
host exactly as the reference makes them (`rand() < prob` per image, `rand(jitter, 1/jitter)` per image); the kernels
(csrc/augment.hip) reproduce PIL / NumPy bit for bit for a given draw.

Not here: random_zoom_rotate, random_gridmask's rotation, random_grayscale, random_blur, random_histeq and the final
cv2.resize -- OpenCV code whose fixed-point arithmetic cannot be pinned in an image without OpenCV; they stay on the host.
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
