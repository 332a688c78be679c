"""Rows of the measured dispatch tables (csrc/gemm_tuned.h, csrc/dw_tuned.h) for the tests that walk them."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'tf-keras-deeplabv3p-model-set_amd', 'csrc')


def _rows(path, n):
    out = []
    pat = re.compile(r'^\s*\{' + ', '.join([r'(-?\d+)'] * n) + r'\}')
    for line in open(path):
        m = pat.match(line)
        if m:
            r = tuple(int(v) for v in m.groups())
            if r[0] >= 0:
                out.append(r)
    return out


def gemm_rows():
    """(role, M, K, N, nt, mi, pc): role 0 forward, 1 forward + statistics, 2 data gradient, 3 data gradient + BN sums,
    4 weight gradient (nt = tile index, mi = workgroups per CU)"""
    return _rows(os.path.join(CSRC, 'gemm_tuned.h'), 7)


def sb_rows():
    """(role + 5, M, K, N, nt, mi, pc) of the split-bf16 GEMM (csrc/sb_tuned.h, g_sb_tuned; pc > 100: the wide family, wm = pc - 100)"""
    return [r for r in _rows(os.path.join(CSRC, 'sb_tuned.h'), 7)]


def sb_pays_rows():
    """(role, M, K, N, pays): the measured split-against-fp32 verdicts that differ from the executor's threshold rule (g_sb_pays)"""
    return _rows(os.path.join(CSRC, 'sb_tuned.h'), 5)


def dw_rows():
    """(role, N, H, W, C, k, stride, rate, per_cu, want, maxth, tw): role 0 forward, 1 data gradient, 2 data gradient + BN
    sums, 3 weight gradient; geometry as the planner sees it"""
    return _rows(os.path.join(CSRC, 'dw_tuned.h'), 12)


def same_geometry(H, W, k, stride, rate):
    """TF 'SAME'  -> Ho, Wo, pad_t, pad_l"""
    keff = k + (k - 1) * (rate - 1)
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + keff - H, 0)
    pw = max((Wo - 1) * stride + keff - W, 0)
    return Ho, Wo, ph // 2, pw // 2
