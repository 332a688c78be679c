"""Parity for EVERY row of the measured dispatch tables and for every value of the dispatch knobs.

csrc/gemm_tuned.h and csrc/dw_tuned.h pick a kernel instantiation / tile / grid per exact production shape; the tuners
only time.  A wrong-but-fast instantiation for a shape that only production hits would pass every small-shape test, so:

  * CPU (no launch): every row is REACHED by the dispatcher at its key (dl3p_gemm_plan_query / dl3p_dw_plan_query report
    `from table` and the row's own choice) -- a row nobody reaches is a dead row and fails here;
  * GPU: every row's exact (role, M, K, N) / (role, N, H, W, C, k, stride, rate) runs through the library with the table ON
    and is compared with a float64 reference at the tolerances of tests/test_production_shapes_gpu.py (PW_PROD / DW_PROD):
    the whole output against float64 torch (rocBLAS dgemm / elementwise tap sums -- no kernel of this repo), and 2048 sampled
    rows (GEMM) / one image (depthwise) against float64 NumPy and the oracle's own depthwise loops;
  * every value of gemm_nt x gemm_mi, gemm_per_cu, wgrad_tile x wgrad_per_cu, dw_tw / dw_per_cu / dw_want / dw_maxth pinned
    on ragged shapes.
"""
import ctypes

import numpy as np
import pytest

from conftest import load_pkg
from _tuned import gemm_rows, dw_rows, sb_rows, sb_pays_rows, same_geometry

GEMM = gemm_rows()
DW = dw_rows()
SB = sb_rows()
SB_PAYS = sb_pays_rows()


def _lib():
    return load_pkg('_lib').lib()


def _gemm_plan(L, role, M, K, N):
    out = (ctypes.c_int * 6)()
    L.gemm_plan_query(role, M, K, N, out)
    return list(out)


def _dw_plan(L, role, N, H, W, C, k, s, r):
    Ho, Wo, pt, pl = same_geometry(H, W, k, s, r)
    out = (ctypes.c_int * 6)()
    L.dw_plan_query(role, N, H, W, C, k, s, r, pt, pl, Ho, Wo, out)
    return list(out)


# ------------------------------------------------------------------------------------------------ CPU: reachability
def test_every_gemm_table_row_is_reached_by_the_dispatcher():
    L = _lib()
    L.set_option(b'pw_small_min_rows', -1)
    L.set_option(b'gemm_tuned', 1)
    try:
        assert len(GEMM) > 300
        dead = []
        for (role, M, K, N, nt, mi, pc) in GEMM:
            fam, q_nt, q_mi, gx, gy, from_table = _gemm_plan(L, role, M, K, N)
            if fam != 0 or not from_table or q_nt != nt or (role < 4 and q_mi != mi) or (role == 4 and q_mi != mi):
                dead.append(((role, M, K, N, nt, mi, pc), (fam, q_nt, q_mi, from_table)))
        assert not dead, dead[:10]
        # and the switch: with the table off the same keys fall back to the heuristics
        L.set_option(b'gemm_tuned', 0)
        assert all(_gemm_plan(L, r[0], r[1], r[2], r[3])[5] == 0 for r in GEMM[:20])
    finally:
        L.set_option(b'gemm_tuned', 1)
        L.set_option(b'pw_small_min_rows', 64)


def test_every_split_gemm_table_row_is_reached_by_the_dispatcher():
    """csrc/sb_tuned.h: the tile rows through dl3p_gemm_plan_query(role + 5 ...), the verdict rows through dl3p_pwconv_sb_pays"""
    L = _lib()
    L.set_option(b'pw_small_min_rows', -1)
    L.set_option(b'gemm_tuned', 1)
    try:
        assert len(SB) > 100 and len(SB_PAYS) > 20
        dead = []
        for (role, M, K, N, nt, mi, pc) in SB:
            assert 5 <= role <= 9
            if role == 9:       # the weight gradient: {tile index, workgroups per CU}
                fam, q_tile, q_kf, slabs, tiles, from_table = _gemm_plan(L, 9, M, K, N)
                if fam != 4 or not from_table or q_tile != nt or slabs < 1:
                    dead.append(((role, M, K, N, nt, mi, pc), (fam, q_tile, q_kf, slabs, from_table)))
                continue
            fam, q_nt, q_mi, q_wm, wgs, from_table = _gemm_plan(L, role, M, K, N)
            want_wm = pc - 100 if pc > 100 else 1
            if fam != 3 or not from_table or q_nt != nt or q_mi != mi or q_wm != want_wm:
                dead.append(((role, M, K, N, nt, mi, pc), (fam, q_nt, q_mi, q_wm, from_table)))
        assert not dead, dead[:10]
        assert all(L.pwconv_sb_pays(r[0], r[1], r[2], r[3]) == r[4] for r in SB_PAYS)
        assert L.pwconv_sb_pays(0, 12345, 128, 128) == -1
        L.set_option(b'gemm_tuned', 0)
        assert all(_gemm_plan(L, r[0], r[1], r[2], r[3])[5] == 0 for r in SB[:20])
        assert all(L.pwconv_sb_pays(r[0], r[1], r[2], r[3]) == -1 for r in SB_PAYS[:20])
    finally:
        L.set_option(b'gemm_tuned', 1)
        L.set_option(b'pw_small_min_rows', 64)


def test_every_depthwise_table_row_is_reached_by_the_dispatcher():
    L = _lib()
    L.set_option(b'dw_tuned', 1)
    assert len(DW) > 80
    dead = []
    for row in DW:
        role, N, H, W, C, k, s, r, per_cu, want, maxth, tw = row
        kind, q_tw, th, nbands, nbx, from_table = _dw_plan(L, role, N, H, W, C, k, s, r)
        # (gather / residue-lattice launches, kind 0 / 3, take only the row's workgroups-per-CU; the strip width binds for
        # the 3x3 stride-1 window kernel)
        if not from_table or (kind == 1 and k == 3 and tw and q_tw != tw):
            dead.append((row, (kind, q_tw, th, nbands, nbx, from_table)))
    assert not dead, dead[:10]


# ------------------------------------------------------------------------------------------------ GPU: numerics per row
DEV = 'cuda'


def _t():
    import torch
    return torch


def _relmax(got, want):
    """max |got - want| / max |want| on the device, float64"""
    torch = _t()
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def _sample_rows(M, n, seed):
    rng = np.random.default_rng(seed)
    idx = np.unique(np.concatenate([rng.integers(0, M, n), [0, M - 1], np.arange(max(0, M - 70), M)]))   # + the ragged tail
    return idx


def _act6(u):
    return u.clamp(0.0, 6.0)


def run_gemm_row(ops, role, M, K, N, seed=0, split=False):
    """one (role, M, K, N) of the tiled pointwise GEMM as the executor launches it, against float64 (split: the same launch
    on the split-bf16 kernel, held to the SAME bounds)"""
    torch = _t()
    g = torch.Generator(device=DEV)
    g.manual_seed(1000 * role + M % 9973 + 7 * K + 13 * N + seed)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    if role in (0, 1):
        x = rnd(M, K)
        wt = rnd(N, K) / K ** 0.5
        sc = torch.rand(K, device=DEV, generator=g) + 0.5
        sh = rnd(K) * 0.3
        part = ops.new_partials(N, DEV) if role == 1 else None
        if split:
            out = ops.pwconv_fwd_sb(x, ops.split_bf16x3(wt), K, None, sc, sh, ops.ACT_RELU6, partials=part)
        else:
            out = ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, partials=part)
        y, rows = out if role == 1 else (out, 0)
        a64 = _act6(x.double() * sc.double() + sh.double())
        y64 = a64 @ wt.double().t()
        assert _relmax(y, y64) < 2e-5, 'forward vs float64 torch'
        idx = _sample_rows(M, 2048, K + N)
        xs = x[idx].cpu().numpy().astype(np.float64)
        a_np = np.clip(xs * sc.cpu().numpy().astype(np.float64) + sh.cpu().numpy().astype(np.float64), 0.0, 6.0)
        y_np = a_np @ wt.cpu().numpy().astype(np.float64).T
        got = y[idx].cpu().numpy().astype(np.float64)
        assert np.abs(got - y_np).max() < 2e-5 * np.abs(y_np).max(), 'forward vs float64 NumPy (sampled rows)'
        if role == 1:
            p = part[:rows * 2 * N].reshape(rows, 2, N).double().sum(0)
            s1, s2 = y64.sum(0), (y64 * y64).sum(0)
            assert float((p[0] - s1).abs().max()) < 1e-4 * max(float(s1.abs().max()), float(M) ** 0.5), 'stat sum'
            assert float((p[1] - s2).abs().max()) < 1e-4 * float(s2.abs().max()), 'stat sum of squares'
    elif role in (2, 3):
        # data gradient: dy (M, K) x w (conv cin = N, conv cout = K) -> gx (M, N)
        dy = rnd(M, K)
        w = rnd(N, K) / K ** 0.5
        gx64 = dy.double() @ w.double().t()
        if role == 2:
            gx = ops.pwconv_bwd_data_sb(dy, ops.split_bf16x3(w), K) if split else ops.pwconv_bwd_data(dy, w)
        else:
            z = rnd(M, N)
            sc = torch.rand(N, device=DEV, generator=g) + 0.5
            sh = rnd(N) * 0.3
            mean = z.mean(0)
            invstd = 1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-3)
            part = ops.new_partials(N, DEV)
            if split:
                gx, rows = ops.pwconv_bwd_data_sb(dy, ops.split_bf16x3(w), K, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean,
                                                  invstd=invstd, partials=part)
            else:
                gx, rows = ops.pwconv_bwd_data_bn(dy, w, z, sc, sh, ops.ACT_RELU6, mean, invstd, part)
            u = z.double() * sc.double() + sh.double()
            d = gx64 * ((u > 0) & (u < 6))
            xh = (z.double() - mean.double()) * invstd.double()
            p = part[:rows * 2 * N].reshape(rows, 2, N).double().sum(0)
            assert float((p[0] - d.sum(0)).abs().max()) < 2e-4 * float(d.abs().sum(0).max()), 'BN backward sum'
            assert float((p[1] - (d * xh).sum(0)).abs().max()) < 2e-4 * float((d * xh).abs().sum(0).max()), 'BN backward sum * xhat'
        assert _relmax(gx, gx64) < 2e-5, 'data gradient vs float64 torch'
        idx = _sample_rows(M, 2048, K + N)
        g_np = dy[idx].cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64).T
        assert np.abs(gx[idx].cpu().numpy() - g_np).max() < 2e-5 * np.abs(g_np).max(), 'data gradient vs float64 NumPy (sampled rows)'
    else:
        x = rnd(M, K)
        dy = rnd(M, N)
        sc = torch.rand(K, device=DEV, generator=g) + 0.5
        sh = rnd(K) * 0.3
        gw = ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU6)
        a64 = _act6(x.double() * sc.double() + sh.double())
        gw64 = a64.t() @ dy.double()
        err = float((gw.double() - gw64).abs().max())
        assert err < 3e-5 * M ** 0.5 * max(1.0, float(a64.max())), 'weight gradient (absolute, fp32 slabs in fixed order)'
        assert err < 5e-3 * float(gw64.abs().max()), 'weight gradient (relative to the largest entry)'
        # a float64 NumPy cross-check of a corner block of the gradient (all M rows, 8 x 8 entries)
        a_np = a64[:, :8].cpu().numpy()
        blk = a_np.T @ dy[:, :8].cpu().numpy().astype(np.float64)
        assert np.abs(gw[:8, :8].cpu().numpy() - blk).max() < 3e-5 * M ** 0.5 * max(1.0, float(a_np.max()))


@pytest.mark.gpu
@pytest.mark.parametrize('row', GEMM, ids=lambda r: 'r%d_%dx%dx%d' % r[:4])
def test_gemm_table_row_matches_float64(ops, row):
    L = _lib()
    L.set_option(b'pw_small_min_rows', -1)           # production dispatch
    L.set_option(b'gemm_tuned', 1)
    try:
        role, M, K, N, nt, mi, pc = row
        plan = _gemm_plan(L, role, M, K, N)
        assert plan[0] == 0 and plan[5] == 1 and plan[1] == nt, plan
        L.set_option(b'split_wgrad', 0)          # this row is the fp32-input MFMA kernel's (the split kernel has its own: SB, role 9)
        run_gemm_row(ops, role, M, K, N)
    finally:
        L.set_option(b'split_wgrad', 1)
        L.set_option(b'pw_small_min_rows', 64)


@pytest.mark.gpu
@pytest.mark.parametrize('row', SB, ids=lambda r: 's%d_%dx%dx%d' % (r[0] - 5, r[1], r[2], r[3]))      # (s4 = the weight gradient, role 9)
def test_split_gemm_table_row_matches_float64(ops, row):
    """every tile row of csrc/sb_tuned.h at its exact launch shape, at the bounds of the fp32-input MFMA kernel's rows"""
    L = _lib()
    L.set_option(b'pw_small_min_rows', -1)
    L.set_option(b'gemm_tuned', 1)
    try:
        role, M, K, N, nt, mi, pc = row
        plan = _gemm_plan(L, role, M, K, N)
        if role == 9:       # weight gradient: dl3p_pwconv_bwd_weight takes the split kernel by itself (wgrad_sb_route)
            L.set_option(b'split_wgrad', 1)
            assert plan[0] == 4 and plan[5] == 1 and plan[1] == nt, plan
            run_gemm_row(ops, 4, M, K, N)
        else:
            assert plan[0] == 3 and plan[5] == 1 and plan[1] == nt and plan[2] == mi, plan
            run_gemm_row(ops, role - 5, M, K, N, split=True)
    finally:
        L.set_option(b'pw_small_min_rows', 64)


# ragged shapes for the knob sweeps: M not a multiple of 64, K not of 32, N not of 16 * nt
SWEEP_SHAPES = [(4357, 100, 200), (1089, 36, 24), (9001, 304, 256), (2600, 728, 132)]


@pytest.mark.gpu
@pytest.mark.parametrize('mi', [1, 2])
@pytest.mark.parametrize('nt', [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_gemm_tile_choice_is_correct(ops, nt, mi):
    L = _lib()
    L.set_option(b'pw_small_min_rows', 1 << 30)      # keep the streaming kernels out: this is about the tiled kernel
    L.set_option(b'gemm_nt', nt)
    L.set_option(b'gemm_mi', mi)
    try:
        for (M, K, N) in SWEEP_SHAPES:
            for role in (0, 1, 2, 3):
                plan = _gemm_plan(L, role, M, K, N)
                assert plan[0] == 0 and plan[1] == nt and plan[2] == mi, plan
                run_gemm_row(ops, role, M, K, N, seed=nt * 2 + mi)
    finally:
        L.set_option(b'gemm_nt', 0)
        L.set_option(b'gemm_mi', 0)
        L.set_option(b'pw_small_min_rows', 64)


@pytest.mark.gpu
@pytest.mark.parametrize('pc', [1, 2, 3, 4, 5, 6, 8])
def test_every_gemm_per_cu_choice_is_correct(ops, pc):
    L = _lib()
    L.set_option(b'pw_small_min_rows', 1 << 30)
    L.set_option(b'gemm_per_cu', pc)
    try:
        for (M, K, N) in [(70001, 100, 200), (266256 // 4 + 3, 304, 256)]:     # enough row tiles for the persistent loop
            for role in (0, 1, 2, 3):
                run_gemm_row(ops, role, M, K, N, seed=pc)
    finally:
        L.set_option(b'gemm_per_cu', 0)
        L.set_option(b'pw_small_min_rows', 64)


@pytest.mark.gpu
@pytest.mark.parametrize('per_cu', [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize('tile', [0, 1, 2, 3])
def test_every_wgrad_tile_choice_is_correct(ops, tile, per_cu):
    L = _lib()
    L.set_option(b'wgrad_tile', tile)
    L.set_option(b'wgrad_per_cu', per_cu)
    L.set_option(b'split_wgrad', 0)
    try:
        for (M, K, N) in [(70001, 100, 200), (9001, 304, 256), (2600, 728, 132)]:
            plan = _gemm_plan(L, 4, M, K, N)
            assert plan[0] == 0 and plan[1] == tile, plan
            run_gemm_row(ops, 4, M, K, N, seed=tile * 8 + per_cu)
    finally:
        L.set_option(b'wgrad_tile', -1)
        L.set_option(b'wgrad_per_cu', 0)
        L.set_option(b'split_wgrad', 1)


@pytest.mark.gpu
@pytest.mark.parametrize('per_cu', [1, 2, 3, 4, 6])
@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4])
def test_every_split_wgrad_tile_choice_is_correct(ops, tile, per_cu):
    """pw_wgrad_sb_kernel: 128 x 128 (waves 2 x 2), 64 x 128, 128 x 64, 64 x 64, 128 x 256 (round 6, dynamic LDS) tiles x workgroups
    per CU, ragged M / K / N"""
    L = _lib()
    L.set_option(b'split_wgrad', 1)
    L.set_option(b'split_wgrad_tile', tile)
    L.set_option(b'split_wgrad_per_cu', per_cu)
    try:
        for (M, K, N) in [(70001, 100, 200), (9001, 304, 256), (2600, 728, 132), (33333, 132, 68)]:
            plan = _gemm_plan(L, 9, M, K, N)
            assert plan[0] == 4 and plan[1] == tile, plan
            run_gemm_row(ops, 4, M, K, N, seed=tile * 8 + per_cu)
    finally:
        L.set_option(b'split_wgrad_tile', -1)
        L.set_option(b'split_wgrad_per_cu', 0)


# ------------------------------------------------------------------------------------------------ depthwise rows
def _dw_ref(a64, w64, k, s, r, pads, Ho, Wo):
    """float64 depthwise conv as a sum of shifted slices (plain torch elementwise ops); a64 (N,H,W,C), w64 (k,k,C)"""
    torch = _t()
    pt, pb, pl, pr = pads
    ap = torch.nn.functional.pad(a64, (0, 0, pl, pr, pt, pb))
    y = torch.zeros((a64.shape[0], Ho, Wo, a64.shape[3]), dtype=torch.float64, device=a64.device)
    for ky in range(k):
        for kx in range(k):
            y = y + ap[:, ky * r: ky * r + (Ho - 1) * s + 1: s, kx * r: kx * r + (Wo - 1) * s + 1: s, :] * w64[ky, kx]
    return y


def run_dw_row(ops, role, N, H, W, C, k, s, r, seed=0):
    torch = _t()
    from oracle import np_ops as O
    g = torch.Generator(device=DEV)
    g.manual_seed(role * 1000 + H * 7 + C + r + seed)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    Ho, Wo, pt, pl = same_geometry(H, W, k, s, r)
    keff = k + (k - 1) * (r - 1)
    pb = max((Ho - 1) * s + keff - H, 0) - pt
    pr = max((Wo - 1) * s + keff - W, 0) - pl
    x = rnd(N, H, W, C)
    w = rnd(k, k, C) * 0.3
    sc = torch.rand(C, device=DEV, generator=g) + 0.5
    sh = rnd(C) * 0.3
    a64 = (x.double() * sc.double() + sh.double()).clamp(0.0, 6.0).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = _dw_ref(a64, w64, k, s, r, (pt, pb, pl, pr), Ho, Wo)
    if role == 0:
        part = ops.new_partials(C, DEV)
        y, rows = ops.dwconv2d_fwd(x, w, s, r, 'same', sc, sh, ops.ACT_RELU6, partials=part)
        assert _relmax(y, y64.detach()) < 1e-5, 'forward vs float64 torch'
        # the oracle's own loops on the first image (independent of torch)
        y_np = O.dwconv2d_fwd(a64[:1].detach().cpu().numpy(), w.cpu().numpy().astype(np.float64), s, r, 'same')
        assert np.abs(y[:1].cpu().numpy() - y_np).max() < 1e-5 * np.abs(y_np).max(), 'forward vs the oracle'
        p = part[:rows * 2 * C].reshape(rows, 2, C).double().sum(0)
        yf = y64.detach().reshape(-1, C)
        assert float((p[0] - yf.sum(0)).abs().max()) < 1e-4 * max(float(yf.sum(0).abs().max()), float(yf.shape[0]) ** 0.5), 'stat sum'
        assert float((p[1] - (yf * yf).sum(0)).abs().max()) < 1e-4 * float((yf * yf).sum(0).max()), 'stat sum of squares'
        return
    gy = rnd(N, Ho, Wo, C)
    y64.backward(gy.double())
    gx64, gw64 = a64.grad, w64.grad
    if role == 1:
        gx = ops.dwconv2d_bwd_data(gy, w, (N, H, W, C), s, r, 'same')
        assert _relmax(gx, gx64) < 1e-5, 'data gradient'
    elif role == 2:
        z = rnd(N, H, W, C)
        zf = z.reshape(-1, C)
        mean = zf.mean(0)
        invstd = 1.0 / torch.sqrt(zf.var(0, unbiased=False) + 1e-3)
        part = ops.new_partials(C, DEV)
        gx, rows = ops.dwconv2d_bwd_data_bn(gy, w, (N, H, W, C), z, sc, sh, ops.ACT_RELU6, mean, invstd, part, s, r, 'same')
        assert _relmax(gx, gx64) < 1e-5, 'data gradient (+BN sums)'
        u = zf.double() * sc.double() + sh.double()
        d = gx64.reshape(-1, C) * ((u > 0) & (u < 6))
        xh = (zf.double() - mean.double()) * invstd.double()
        p = part[:rows * 2 * C].reshape(rows, 2, C).double().sum(0)
        assert float((p[0] - d.sum(0)).abs().max()) < 2e-4 * float(d.abs().sum(0).max()), 'BN backward sum'
        assert float((p[1] - (d * xh).sum(0)).abs().max()) < 2e-4 * float((d * xh).abs().sum(0).max()), 'BN backward sum * xhat'
    else:
        gw = ops.dwconv2d_bwd_weight(x, gy, k, s, r, 'same', sc, sh, ops.ACT_RELU6)
        M = N * Ho * Wo
        err = float((gw.double() - gw64).abs().max())
        assert err < 3e-5 * M ** 0.5 * max(1.0, float(a64.max())), 'weight gradient (absolute)'
        assert err < 5e-3 * float(gw64.abs().max()), 'weight gradient (relative to the largest entry)'


@pytest.mark.gpu
@pytest.mark.parametrize('row', DW, ids=lambda r: 'r%d_%dx%dx%dx%d_k%ds%dr%d' % r[:8])
def test_depthwise_table_row_matches_float64(ops, row):
    L = _lib()
    L.set_option(b'dw_tuned', 1)
    role, N, H, W, C, k, s, r = row[:8]
    plan = _dw_plan(L, role, N, H, W, C, k, s, r)
    assert plan[5] == 1, plan
    run_dw_row(ops, role, N, H, W, C, k, s, r)


@pytest.mark.gpu
@pytest.mark.parametrize('knob,value', [('dw_tw', 2), ('dw_tw', 4), ('dw_per_cu', 1), ('dw_per_cu', 3), ('dw_per_cu', 8),
                                        ('dw_want', 128), ('dw_want', 768), ('dw_want', 2048), ('dw_maxth', 4), ('dw_maxth', 8),
                                        ('dw_maxth', 33), ('dw_maxth', 65)])
def test_every_depthwise_plan_knob_is_correct(ops, knob, value):
    L = _lib()
    L.set_option(knob.encode(), value)
    try:
        # ragged maps: widths that are not multiples of the strip, heights that do not split evenly, a rate whose sub-lattices
        # have different sizes, stride 2
        for (N, H, W, C, k, s, r) in [(3, 33, 35, 40, 3, 1, 1), (2, 67, 33, 24, 3, 1, 2), (2, 41, 45, 16, 3, 1, 6),
                                       (2, 65, 67, 32, 3, 2, 1)]:
            for role in (0, 1, 2, 3):
                run_dw_row(ops, role, N, H, W, C, k, s, r, seed=value)
    finally:
        L.set_option(knob.encode(), 0)
