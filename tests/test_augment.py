"""Byte-level augmentations (reference deeplabv3p/data.py:72-104, common/data_utils.py:14-60,83-239,364-400).
CPU: the NumPy restatement (oracle/np_augment.py) against vectors made by PIL itself (tests/golden/pil_enhance.npz) and,
when PIL is importable, against live PIL on larger images.  GPU: the device kernels against the restatement, bit for bit."""
import os

import numpy as np
import pytest

from oracle import np_augment as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pil_enhance.npz')


def test_oracle_matches_pil_golden_vectors():
    g = np.load(GOLD)
    factors = g['factors']
    n = 0
    for i in range(4):
        img = g['img%d' % i]
        for op in range(4):
            for j, f in enumerate(factors):
                want = g['out%d_op%d_f%d' % (i, op, j)]
                got = A.enhance(img, op, float(f))
                assert np.array_equal(got, want), (i, op, f, int(np.abs(got.astype(int) - want.astype(int)).max()))
                n += 1
    assert n == 4 * 4 * len(factors)


def test_oracle_matches_live_pil():
    PIL = pytest.importorskip('PIL')
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(3)
    ENH = [ImageEnhance.Brightness, ImageEnhance.Color, ImageEnhance.Contrast, ImageEnhance.Sharpness]
    for shape in ((129, 161, 3), (3, 3, 3), (2, 9, 3), (300, 210, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for op, E in enumerate(ENH):
            for f in (0.5, rng.uniform(0.5, 2.0), 1.0, 2.0):
                want = np.asarray(E(Image.fromarray(img)).enhance(f))
                assert np.array_equal(A.enhance(img, op, f), want), (shape, op, f)


def test_flip_crop_restatement():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    lab = rng.integers(0, 21, (9, 11), dtype=np.uint8)
    a, b = A.flip_crop(img, lab, 3, (2, 1), (5, 7))
    assert np.array_equal(a, img[::-1, ::-1][2:7, 1:8]) and np.array_equal(b, lab[::-1, ::-1][2:7, 1:8])
    a, b = A.flip_crop(img, lab, 1)
    assert np.array_equal(a, np.flip(img, 1)) and np.array_equal(b, np.flip(lab, 1))       # cv2.flip(x, 1)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 37, 53), (2, 513, 513), (1, 1, 1), (5, 2, 3), (2, 64, 1024)])
def test_device_enhance_matches_restatement(shape):
    import torch
    from conftest import load_pkg
    aug = load_pkg('augment')
    N, H, W = shape
    rng = np.random.default_rng(H * W + N)
    imgs = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    if N > 1:
        imgs[1] = np.clip(imgs[1].astype(int) // 3 + 170, 0, 255)      # an image that saturates under f > 1
    t = torch.from_numpy(imgs).cuda()
    for op in range(4):
        for fs in ([0.5] * N, [2.0] * N, list(rng.uniform(0.5, 2.0, N)), [1.0] * N, [0.0] * N):
            got = aug.enhance(t, op, fs).cpu().numpy()
            for n in range(N):
                want = A.enhance(imgs[n], op, float(np.float32(fs[n])))
                assert np.array_equal(got[n], want), (shape, op, fs[n], int(np.abs(got[n].astype(int) - want.astype(int)).max()))
    # in place (allowed for the pointwise ones)
    t2 = t.clone()
    aug.enhance(t2, A.COLOR, [1.5] * N, out=t2)
    assert np.array_equal(t2.cpu().numpy(), np.stack([A.enhance(imgs[n], A.COLOR, 1.5) for n in range(N)]))


@pytest.mark.gpu
def test_device_flip_crop_matches_numpy():
    import torch
    from conftest import load_pkg
    aug = load_pkg('augment')
    rng = np.random.default_rng(8)
    N, H, W = 6, 45, 67
    imgs = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    labs = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    ti, tl = torch.from_numpy(imgs).cuda(), torch.from_numpy(labs).cuda()
    flags = [0, 1, 2, 3, 1, 2]
    yx = [(0, 0), (12, 30), (5, 0), (13, 31), (1, 2), (7, 7)]
    for args in ((flags, None, None), (None, yx, (32, 36)), (flags, yx, (32, 36))):
        a, b = aug.flip_crop(ti, tl, *args)
        for n in range(N):
            wa, wb = A.flip_crop(imgs[n], labs[n], 0 if args[0] is None else args[0][n], None if args[1] is None else args[1][n],
                                 args[2])
            assert np.array_equal(a[n].cpu().numpy(), wa) and np.array_equal(b[n].cpu().numpy(), wb), (args, n)
    # the reference-named wrappers run and keep shapes / dtypes
    np.random.seed(0)
    a, b = aug.random_horizontal_flip(ti, tl)
    a, b = aug.random_vertical_flip(a, b)
    a = aug.random_sharpness(aug.random_contrast(aug.random_chroma(aug.random_brightness(a))))
    a, b = aug.random_crop(a, b, (33, 33), prob=1.0)
    assert a.shape == (N, 33, 33, 3) and b.shape == (N, 33, 33) and a.dtype == torch.uint8


# ---------------------------------------------------------------------------------------------------- GridMask
def _gridmask_cases():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pil_gridmask.npz'))
    for i, (h, w, d, st_h, st_w, r) in enumerate(g['cases']):
        yield (int(h), int(w), int(d), int(st_h), int(st_w), int(r)), np.unpackbits(g['m%d' % i])[:h * w].reshape(h, w)


def test_gridmask_restatement_matches_pil_golden_and_live_pil():
    """oracle/np_augment.gridmask_keep (Pillow's rotate restated: transpose fast paths + 16.16 fixed-point affine map) against
    the vectors PIL wrote (tests/golden/make_pil_gridmask.py) and against PIL itself on fresh draws"""
    n = 0
    for case, want in _gridmask_cases():
        assert np.array_equal(A.gridmask_keep(*case), want), case
        n += 1
    assert n == 32
    from golden.make_pil_gridmask import pil_mask
    rng = np.random.default_rng(5)
    for (h, w) in ((129, 129), (96, 160), (17, 40)):
        for _ in range(25):
            d = int(rng.integers(max(2, w // 7), max(3, w // 3)))
            case = (h, w, d, int(rng.integers(d)), int(rng.integers(d)), int(rng.integers(360)))
            assert np.array_equal(A.gridmask_keep(*case), pil_mask(*case)), case


def test_gridmask_draws_follow_the_reference_order():
    """augment.random_gridmask draws rand(), randint(W//7, W//3), randint(d), randint(d), randint(360) per image, in the
    reference's order (Grid.__call__, data_utils.py:292-322): a seeded np.random gives the reference's grids"""
    from conftest import load_pkg
    aug = load_pkg('augment')
    np.random.seed(11)
    want = []
    for _ in range(5):
        if np.random.rand() > 0.6:
            want.append(None)
            continue
        d = np.random.randint(97 // 7, 97 // 3)
        want.append((d, np.random.randint(d), np.random.randint(d), np.random.randint(360)))
    got = []
    np.random.seed(11)
    orig = aug.gridmask
    aug.gridmask = lambda images, labels, draws: got.extend(draws) or (images, labels)
    try:
        class _T:          # shape-only stand-in: the draws do not touch the data
            shape = (5, 65, 97, 3)
        aug.random_gridmask(_T(), None, prob=0.6)
    finally:
        aug.gridmask = orig
    assert got == want and any(g is not None for g in got)
    # the parameter block the kernel takes agrees with the oracle's
    assert aug.gridmask_params(65, 97, 20, 3, 7, 123)[:15] == A.gridmask_params(65, 97, 20, 3, 7, 123)


@pytest.mark.gpu
def test_device_gridmask_matches_pil():
    """dl3p_aug_gridmask_u8 against the masks PIL produced: image and label times the mask, bit for bit, untouched images
    stay untouched"""
    import torch
    from conftest import load_pkg
    aug = load_pkg('augment')
    rng = np.random.default_rng(3)
    by_shape = {}
    for case, mask in _gridmask_cases():
        by_shape.setdefault(case[:2], []).append((case, mask))
    for (h, w), items in by_shape.items():
        N = len(items) + 1
        imgs = rng.integers(1, 256, (N, h, w, 3), dtype=np.uint8)
        labs = rng.integers(1, 22, (N, h, w), dtype=np.uint8)
        ti, tl = torch.from_numpy(imgs).cuda(), torch.from_numpy(labs).cuda()
        draws = [c[2:] for c, _ in items] + [None]
        aug.gridmask(ti, tl, draws)
        gi, gl = ti.cpu().numpy(), tl.cpu().numpy()
        for n, (case, mask) in enumerate(items):
            assert np.array_equal(gi[n], imgs[n] * mask[..., None]), case
            assert np.array_equal(gl[n], labs[n] * mask), case
        assert np.array_equal(gi[-1], imgs[-1]) and np.array_equal(gl[-1], labs[-1])


# ---- random_grayscale / random_blur: UNPINNED restatements of OpenCV's 8-bit arithmetic (oracle/np_augment.py says why)
def test_gray_and_blur_restatement_known_answers():
    """what the published formulas give on inputs whose answer can be worked out by hand"""
    flat = np.full((9, 11, 3), 77, np.uint8)
    assert np.array_equal(A.cv_gaussian5(flat), flat)                     # the taps sum to 256: a constant image is a fixed point
    assert np.array_equal(A.cv_gray(flat), flat)                          # 1868 + 9617 + 4899 = 16384 = 1 << 14
    px = np.zeros((1, 1, 3), np.uint8); px[0, 0] = (255, 0, 0)
    assert A.cv_gray(px)[0, 0, 0] == (255 * 1868 + 8192) >> 14 == 29      # the FIRST channel carries the blue weight (RGB array, BGR conversion)
    px[0, 0] = (0, 0, 255)
    assert A.cv_gray(px)[0, 0, 0] == 76
    imp = np.zeros((9, 9, 3), np.uint8); imp[4, 4] = 255
    b = A.cv_gaussian5(imp)[..., 0]
    w = np.array([1, 4, 6, 4, 1])
    assert np.array_equal(b[2:7, 2:7], (np.outer(w, w) * 255 + 128) >> 8) and b.sum() == b[2:7, 2:7].sum()
    # BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba): the mirror does NOT repeat the edge pixel, so an impulse one pixel inside the
    # border is seen twice from the border pixel (taps -1 and +1) and the corner pixel itself only once
    imp = np.zeros((9, 9, 3), np.uint8); imp[0, 1] = 255
    b = A.cv_gaussian5(imp)[..., 0]
    assert b[0, 0] == ((4 + 4) * 6 * 255 + 128) >> 8 and b[0, 1] == ((6 + 1) * 6 * 255 + 128) >> 8
    imp = np.zeros((9, 9, 3), np.uint8); imp[0, 0] = 255
    b = A.cv_gaussian5(imp)[..., 0]
    assert b[0, 0] == (36 * 255 + 128) >> 8 and b[0, 1] == (4 * 6 * 255 + 128) >> 8 and b[1, 1] == (4 * 4 * 255 + 128) >> 8


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(4, 37, 53), (2, 513, 513), (4, 3, 3), (4, 5, 64)])
def test_device_gray_blur_matches_restatement(shape):
    import torch
    from conftest import load_pkg
    aug = load_pkg('augment')
    N, H, W = shape
    rng = np.random.default_rng(H * W + N)
    imgs = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    t = torch.from_numpy(imgs).cuda()
    flags = [0, 1, 2, 3][:N] + [3] * max(0, N - 4)
    got = aug.gray_blur(t, flags).cpu().numpy()
    for n in range(N):
        assert np.array_equal(got[n], A.gray_blur(imgs[n], flags[n])), (shape, flags[n])
    # the reference's names: the draws made on the host as the reference makes them
    np.random.seed(3)
    g = aug.random_grayscale(t, prob=0.5).cpu().numpy()
    np.random.seed(3)
    want = [A.gray_blur(imgs[n], 1 if np.random.rand() < 0.5 else 0) for n in range(N)]
    assert all(np.array_equal(g[n], want[n]) for n in range(N))
    np.random.seed(4)
    b = aug.random_blur(t, prob=0.5).cpu().numpy()
    np.random.seed(4)
    want = [A.gray_blur(imgs[n], 2 if np.random.rand() < 0.5 else 0) for n in range(N)]
    assert all(np.array_equal(b[n], want[n]) for n in range(N))
