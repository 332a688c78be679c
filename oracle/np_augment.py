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


# ---------------------------------------------------------------------------------------------------- GridMask
def gridmask_keep(h, w, d, st_h, st_w, r, ratio=0.5):
    """The uint8 (h, w) factor the reference's Grid.__call__ multiplies image and label with (common/data_utils.py:288-339,
    mode = 1), for one set of draws (d = randint(d1, d2), st_h = randint(d), st_w = randint(d), r = randint(360)).

    Restated, not called: a square of ones with edge hh = ceil(sqrt(h^2 + w^2)), zero on the row bands
    [d i + st_h, d i + st_h + l) and the column bands [d i + st_w, ...), l = ceil(d ratio); rotated by r degrees with
    PIL's Image.rotate (NEAREST, no expand, zero fill); centre crop to (h, w); 1 - mask.

    Pillow's rotate (Image.py, 12.x): angle 0 -> copy, 180 -> ROTATE_180, 90 / 270 on a square image -> ROTATE_90 / ROTATE_270;
    otherwise matrix = [cos, sin, 0, -sin, cos, 0] of -angle, each rounded to 15 decimals, the centre (hh/2, hh/2) mapped onto
    itself, and ImagingTransformAffine with the NEAREST filter in 16.16 FIXED POINT (Geometry.c affine_fixed):
        a0 = FIX(a[0]) ... a2 = FIX(a[2] + a[0]/2 + a[1]/2), FIX(v) = floor(v 65536 + 0.5)
        source x of output (x, y) = (a2 + a1 y + a0 x) >> 16 (the C code accumulates these sums step by step: integers, same value),
        out of range -> 0.
    Pinned against PIL itself: tests/golden/make_pil_gridmask.py, tests/test_augment.py."""
    import math
    hh = math.ceil(math.sqrt(h * h + w * w))
    l = math.ceil(d * ratio)
    yy = np.arange(hh)
    band_r = ((yy - st_h) % d) < l
    band_c = ((yy - st_w) % d) < l
    m = (~(band_r[:, None] | band_c[None, :])).astype(np.uint8)      # 1 = kept by the grid
    r = r % 360
    if r == 0:
        rot = m
    elif r == 180:
        rot = m[::-1, ::-1]
    elif r == 90:
        rot = np.rot90(m, 1)
    elif r == 270:
        rot = np.rot90(m, 3)
    else:
        ang = -math.radians(r)
        a = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
        cx = cy = hh / 2.0
        a[2] = a[0] * -cx + a[1] * -cy + a[2]
        a[5] = a[3] * -cx + a[4] * -cy + a[5]
        a[2] += cx
        a[5] += cy
        fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
        a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
        a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
        a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
        X, Y = np.meshgrid(np.arange(hh, dtype=np.int64), np.arange(hh, dtype=np.int64))
        xin = (a2 + a1 * Y + a0 * X) >> 16
        yin = (a5 + a4 * Y + a3 * X) >> 16
        ok = (xin >= 0) & (xin < hh) & (yin >= 0) & (yin < hh)
        rot = np.where(ok, m[np.clip(yin, 0, hh - 1), np.clip(xin, 0, hh - 1)], 0).astype(np.uint8)
    t, lft = (hh - h) // 2, (hh - w) // 2
    return (1 - rot[t:t + h, lft:lft + w]).astype(np.uint8)


def gridmask_params(h, w, d, st_h, st_w, r, ratio=0.5):
    """the integers the device kernel takes for one image: [apply, hh, d, l, st_h, st_w, kind, a0, a1, a2, a3, a4, a5, top, left]
    (kind 0 identity, 1 ROTATE_90, 2 ROTATE_180, 3 ROTATE_270, 4 affine in 16.16 fixed point)"""
    import math
    hh = math.ceil(math.sqrt(h * h + w * w))
    l = math.ceil(d * ratio)
    r = r % 360
    kind = {0: 0, 90: 1, 180: 2, 270: 3}.get(r, 4)
    a0 = a1 = a2 = a3 = a4 = a5 = 0
    if kind == 4:
        ang = -math.radians(r)
        a = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
        cx = cy = hh / 2.0
        a[2] = a[0] * -cx + a[1] * -cy + a[2] + cx
        a[5] = a[3] * -cx + a[4] * -cy + a[5] + cy
        fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
        a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
        a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
        a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    return [1, hh, d, l, st_h, st_w, kind, a0, a1, a2, a3, a4, a5, (hh - h) // 2, (hh - w) // 2]


# ---- random_grayscale / random_blur (common/data_utils.py:152-172, 105-124).  PARITY UNPINNED: the arithmetic lives in OpenCV
# (requirements.txt `opencv-python`), which is not in this image and whose wheels cannot be fetched; the functions below restate
# its published 8-bit algorithms and nothing checks them against OpenCV itself:
#   cvtColor BGR2GRAY, uint8 (imgproc/src/color_yuv / color_rgb: RGB2Gray<uchar>): fixed point with yuv_shift = 14,
#     B2Y = 1868, G2Y = 9617, R2Y = 4899: gray = CV_DESCALE(c0 * B2Y + c1 * G2Y + c2 * R2Y, 14) = (... + (1 << 13)) >> 14
#     -- c0 is the array's FIRST channel: the reference hands an RGB array (PIL) to a BGR conversion, kept as it is;
#   GaussianBlur((5, 5), sigmaX = 0) on uint8 (imgproc/src/smooth: getGaussianKernel's small_gaussian_tab for ksize <= 7 and
#     sigma <= 0 -> [1, 4, 6, 4, 1] / 16; the bit-exact 8-bit path (fixedSmoothInvoker, ufixedpoint16 with 8 fractional bits) keeps
#     the horizontal pass exact and rounds once after the vertical pass: (sum_ij w_i w_j p + 128) >> 8; BORDER_DEFAULT =
#     BORDER_REFLECT_101.
def cv_gray(img):
    c = img.astype(np.int64)
    g = ((c[..., 0] * 1868 + c[..., 1] * 9617 + c[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    return np.stack([g, g, g], axis=-1)


def cv_gaussian5(img):
    w = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), ((2, 2), (2, 2), (0, 0)), mode='reflect')       # numpy 'reflect' = BORDER_REFLECT_101
    H, W, _ = img.shape
    h = sum(w[j] * p[:, j:j + W] for j in range(5))
    v = sum(w[i] * h[i:i + H] for i in range(5))
    return ((v + 128) >> 8).astype(np.uint8)


def gray_blur(img, flags):
    out = img
    if flags & 1:
        out = cv_gray(out)
    if flags & 2:
        out = cv_gaussian5(out)
    return out.copy()
