"""Evaluation row: the oracle's restatement of eval.py's confusion matrix / mIOU
arithmetic against a brute-force count, and the product's host-side summary against the oracle."""
import numpy as np

from conftest import load_pkg
from oracle import np_ops as O


def test_confusion_matrix_is_a_pixel_count():
    rng = np.random.default_rng(0)
    C = 7
    gt = rng.integers(0, C, (3, 11, 13))
    gt[rng.uniform(size=gt.shape) < 0.1] = 255            # ignored label
    gt[0, 0, 0] = -1
    pr = rng.integers(0, C, gt.shape)
    cm = O.confusion_matrix(gt, pr, C)
    want = np.zeros((C, C), np.int64)
    for g, p in zip(gt.ravel(), pr.ravel()):
        if 0 <= g < C:
            want[g, p] += 1                                 # eval.py:446-453 (the commented-out loop)
    assert np.array_equal(cm, want) and cm.sum() == ((gt >= 0) & (gt < C)).sum()


def test_miou_summary_known_answer_and_host_summary():
    cm = np.array([[8, 2, 0], [1, 5, 0], [0, 0, 0]])        # class 2 never occurs and is never predicted
    s = O.miou_summary(cm)
    iou = np.array([8 / 11, 5 / 8, 0.0])                    # I / (row + col - I); 0/0 -> 0 (eval.py:473)
    assert np.allclose(s['IoU'], iou) and np.isclose(s['mIoU'], iou.mean())
    assert np.isclose(s['PixelAcc'], 13 / 16)
    assert np.allclose(s['ClassAcc'], [0.8, 5 / 6, 0.0]) and np.isclose(s['mClassAcc'], (0.8 + 5 / 6) / 3)
    assert np.isclose(s['FWIoU'], 10 / 16 * 8 / 11 + 6 / 16 * 5 / 8)
    assert np.allclose(s['Dice'], [16 / 19, 10 / 13, 0.0])
    mine = load_pkg().miou_from_confusion(cm, class_names=['a', 'b', 'c'])
    for k in ('mIoU', 'PixelAcc', 'mClassAcc', 'FWIoU'):
        assert np.isclose(mine[k], s[k]), k
    for k in ('IoU', 'ClassAcc', 'Dice', 'Freq'):
        assert np.allclose(mine[k], s[k]), k
    assert list(mine['IoU_by_class']) == ['a', 'b', 'c']    # sorted by IoU, descending (eval.py:493)


def test_focal_and_weighted_loss_gradients_are_derivatives():
    """the oracle's analytic d loss / d logits of the two optional losses (loss.py:63-118, 159-191) against central
    differences, and their known limits: gamma = 0, alpha = 1 is the plain loss; unit weights are the plain loss"""
    rng = np.random.default_rng(3)
    z = rng.standard_normal((2, 5, 4, 6)) * 2
    lab = rng.integers(0, 6, (2, 5, 4)).astype(np.float64)
    lab[0, 0, 0] = 255
    w = rng.uniform(0.3, 2.5, 6)
    for spec in (('weighted', w), ('focal', 2.0, 0.25), ('focal', 1.0, 0.5), ('focal', 3.5, 1.0)):
        loss, _, g = O.loss_fwd_bwd(z, lab, spec)
        num = np.zeros_like(z)
        eps = 1e-6
        for i in np.ndindex(z.shape):
            zp, zm = z.copy(), z.copy()
            zp[i] += eps; zm[i] -= eps
            num[i] = (O.loss_fwd_bwd(zp, lab, spec)[0] - O.loss_fwd_bwd(zm, lab, spec)[0]) / (2 * eps)
        assert np.abs(num - g).max() < 1e-8, spec[0]
        assert np.abs(g[0, 0, 0]).max() == 0                # ignored pixel
    ce, _, gce = O.sparse_ce_fwd_bwd(z, lab)
    l0, _, g0 = O.loss_fwd_bwd(z, lab, ('focal', 0.0, 1.0))
    l1, _, g1 = O.loss_fwd_bwd(z, lab, ('weighted', np.ones(6)))
    assert np.isclose(l0, ce) and np.allclose(g0, gce) and np.isclose(l1, ce) and np.allclose(g1, gce)


def test_jaccard_metric_restatement_and_count_form():
    """oracle Jaccard (deeplabv3p/metrics.py:29-46) on a hand example, and the product's count-based evaluation of it"""
    lab = np.array([[0, 0, 1, 1, 255, 2], [0, 0, 0, 0, 0, 0]])
    pred = np.array([[0, 1, 1, 1, 1, 0], [0, 0, 0, 2, 2, 0]])
    # class 0: image 0 inter 1 union 3; image 1 inter 4 union 6 -> mean(1/3, 4/6) = 0.5
    # class 1: only image 0: inter 2, union 4 (the ignored pixel predicted as 1 counts in the union) -> 0.5
    # class 2: only image 0: inter 0, union 1 (label) -> 0      => mean = 1/3
    assert np.isclose(O.jaccard_metric(lab, pred, 3), (0.5 + 0.5 + 0.0) / 3)
    C = 3
    counts = np.zeros((2, 3, C))
    for c in range(C):
        counts[:, 0, c] = ((lab == c) & (pred == c)).sum(1)
        counts[:, 1, c] = (lab == c).sum(1)
        counts[:, 2, c] = (pred == c).sum(1)
    assert np.isclose(load_pkg().jaccard_from_counts(counts), O.jaccard_metric(lab, pred, 3))


def test_prepare_labels_matches_sklearn_golden():
    """the oracle's restatement of the generator's label tail (deeplabv3p/data.py:116-145) against vectors made with
    the real sklearn compute_class_weight (tests/golden/make_label_weights.py)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'label_weights.npz'))
    C, ign = int(g['num_classes']), int(g['ignore_index'])
    for i in range(int(g['n'])):
        lab, w = O.prepare_labels(g['u8_%d' % i], C, ign, adaptive=True)
        assert lab.dtype == np.float32 and np.array_equal(lab, g['labels_%d' % i])
        assert w.dtype == np.float32 and np.array_equal(w, g['weights_%d' % i])          # bit-exact
        lab2, none = O.prepare_labels(g['u8_%d' % i], C, ign)
        assert none is None and np.array_equal(lab2, lab)
