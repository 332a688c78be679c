"""TEST INFRASTRUCTURE (checker only).  NumPy restatement of the byte-level augmentations that
SegmentationGenerator.__getitem__ applies (reference deeplabv3p/data.py:72-104, common/data_utils.py): the four PIL
ImageEnhance adjustments the reference calls (data_utils.py:83-239), the flips (:14-60, cv2.flip) and the crop branch of
random_crop (:364-400).

The arithmetic of ImageEnhance lives in Pillow, a third-party dependency of the reference (requirements.txt `pillow`, the
version in this image is 12.2.0); what is restated here is its published algorithm:
  ImageEnhance._Enhance.enhance(f) = Image.blend(degenerate, image, f)            (PIL/ImageEnhance.py)
  ImagingBlend (libImaging/Blend.c): 0 <= f <= 1: (UINT8)(in1 + f * (in2 - in1)) in float32, truncation;
                                     otherwise the same value clipped to [0, 255] first
  degenerate images: Brightness black; Color image.convert('L') (libImaging/Convert.c rgb2l:
                     (R*19595 + G*38470 + B*7471 + 0x8000) >> 16); Contrast the constant int(mean(L) + 0.5);
                     Sharpness image.filter(ImageFilter.SMOOTH): 3x3 (1,1,1,1,5,1,1,1,1)/13, +0.5 and truncation,
                     border pixels copied (libImaging/Filter.c)
PINNED: tests/golden/pil_enhance.npz was produced by PIL itself (tests/golden/make_pil_enhance.py); tests/test_augment.py
checks this module against it bit for bit (and against the live PIL when it is importable)."""
import numpy as np

BRIGHTNESS, COLOR, CONTRAST, SHARPNESS = 0, 1, 2, 3


def luma(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, img, f):
    d = degenerate.astype(np.int32)
    t = d.astype(np.float32) + np.float32(f) * (img.astype(np.int32) - d).astype(np.float32)
    if 0.0 <= f <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def smooth(img):
    """ImageFilter.SMOOTH on an (H,W,3) uint8 image"""
    H, W, _ = img.shape
    out = img.copy()
    if H < 3 or W < 3:
        return out
    f = img.astype(np.int64)
    s = np.zeros((H - 2, W - 2, 3), np.int64)
    for dy in range(3):
        for dx in range(3):
            s += (5 if (dy, dx) == (1, 1) else 1) * f[dy:dy + H - 2, dx:dx + W - 2]
    out[1:-1, 1:-1] = ((2 * s + 13) // 26).astype(np.uint8)      # floor(s / 13 + 0.5)
    return out


def enhance(img, op, f):
    """ImageEnhance.{Brightness, Color, Contrast, Sharpness}(Image.fromarray(img)).enhance(f) as an array"""
    if op == BRIGHTNESS:
        deg = np.zeros_like(img)
    elif op == COLOR:
        deg = np.repeat(luma(img)[..., None], 3, -1)
    elif op == CONTRAST:
        L = luma(img)
        deg = np.full_like(img, int(int(L.astype(np.int64).sum()) / L.size + 0.5))
    elif op == SHARPNESS:
        deg = smooth(img)
    else:
        raise ValueError(op)
    return blend(deg, img, f)


def flip_crop(img, label, flags=0, yx=None, hw=None):
    """random_horizontal_flip (bit 0) / random_vertical_flip (bit 1), then the crop window of random_crop"""
    if flags & 1:
        img, label = img[:, ::-1], label[:, ::-1]
    if flags & 2:
        img, label = img[::-1], label[::-1]
    if yx is not None:
        y, x = yx
        h, w = hw
        img, label = img[y:y + h, x:x + w], label[y:y + h, x:x + w]
    return np.ascontiguousarray(img), np.ascontiguousarray(label)
