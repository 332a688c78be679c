"""GPU parity of every C-ABI op against the CPU oracle (oracle/np_ops.py, fp64) on identical seeded
inputs.  Tolerances are fp32-accumulation level (well inside north_star's 1e-3)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import np_ops as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def close(got, want, rtol=2e-4, atol=2e-5, what=''):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want).max()
    assert err <= atol * scale + rtol * scale, '%s: max err %g (scale %g)' % (what, err, scale)


def stats_from(partials, rows, C):
    p = partials[:rows * 2 * C].reshape(rows, 2, C).double().sum(0).cpu().numpy()
    return p[0], p[1]


DW_CASES = [
    # N, H, W, C, k, stride, rate, padding
    (2, 33, 33, 320, 3, 1, 18, 'same'),
    (2, 33, 33, 320, 3, 1, 12, 'same'),      # 3 x 3 pixels per residue class: dw_fwd_lattice3
    (3, 33, 29, 24, 3, 1, 12, 'same'),       # ... ragged: classes of 3 x 3, 3 x 2, 2 x 3 (rows / columns 9 .. 11 hold two pixels)
    (1, 20, 31, 8, 3, 1, 11, 'same'),        # ... 2 x 3 and 1 x 3 classes
    (2, 97, 97, 16, 3, 1, 36, 'same'),       # BASELINE configs[3]: ASPP rate 36 on the 97 x 97 map
    (2, 33, 33, 320, 3, 1, 6, 'same'),
    (2, 33, 33, 64, 3, 1, 2, 'same'),
    (2, 17, 19, 144, 3, 1, 1, 'same'),
    (1, 65, 65, 96, 3, 2, 1, 'same'),
    (2, 16, 20, 24, 3, 2, 1, 'same'),       # even sizes: extra pad bottom/right
    (2, 33, 33, 128, 3, 2, 1, (1, 1, 1, 1)),  # Xception explicit pad + VALID
    (1, 16, 24, 40, 5, 1, 2, 'same'),
    (1, 16, 24, 72, 5, 2, 1, 'same'),
    (2, 33, 33, 160, 5, 1, 2, 'same'),      # MobileNetV3 last blocks at OS16: 5x5 on the 17x17 rate-2 sub-lattices
    (1, 33, 33, 672, 5, 1, 1, 'same'),      # 4-column strips, 3 channel slabs
    (1, 65, 47, 120, 5, 1, 1, 'same'),      # 2-column strips (wide map)
    (1, 7, 6, 24, 5, 1, 1, 'same'),         # map smaller than the window
    (3, 9, 9, 16, 3, 1, 1, 'same'),
    (1, 5, 5, 2048, 3, 1, 4, 'same'),
]


@pytest.mark.parametrize('case', DW_CASES)
def test_dwconv_fwd_bwd(ops, case):
    N, H, W, C, k, s, r, pad = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((N, H, W, C))
    w = rng.standard_normal((k, k, C)) * 0.3
    sc = rng.uniform(0.5, 1.5, C)
    sh = rng.standard_normal(C) * 0.3
    a = O.act_fwd(x * sc + sh, O.ACT_RELU6)
    y_ref = O.dwconv2d_fwd(a, w, s, r, pad)
    part = ops.new_partials(C, DEV)
    y, rows = ops.dwconv2d_fwd(T(x), T(w), s, r, pad, T(sc), T(sh), ops.ACT_RELU6, partials=part)
    close(y, y_ref, what='dw fwd')
    s1, s2 = stats_from(part, rows, C)
    close(s1, y_ref.reshape(-1, C).sum(0), rtol=1e-4, atol=1e-4 * y_ref.shape[0] * y_ref.shape[1], what='dw stat sum')
    close(s2, (y_ref ** 2).reshape(-1, C).sum(0), rtol=1e-4, what='dw stat sumsq')
    gy = rng.standard_normal(y_ref.shape)
    gx_ref, gw_ref = O.dwconv2d_bwd(a, w, gy, s, r, pad)
    gx = ops.dwconv2d_bwd_data(T(gy), T(w), (N, H, W, C), s, r, pad)
    close(gx, gx_ref, what='dw bwd data')
    base = rng.standard_normal((N, H, W, C))
    gx2 = ops.dwconv2d_bwd_data(T(gy), T(w), (N, H, W, C), s, r, pad, out=T(base), accumulate=True)
    close(gx2, gx_ref + base, what='dw bwd data accumulate')
    gw = ops.dwconv2d_bwd_weight(T(x), T(gy), k, s, r, pad, T(sc), T(sh), ops.ACT_RELU6)
    close(gw, gw_ref, rtol=3e-4, what='dw bwd weight')


@pytest.fixture
def strips_of_two():
    """the 3x3 stride-1 window kernels with 2-output strips (three waves per SIMD) instead of 4 -- what the tuned plan table
    (csrc/dw_tuned.h) selects for some shapes"""
    L = load_pkg('_lib').lib()
    L.set_option(b'dw_tw', 2)
    yield
    L.set_option(b'dw_tw', 0)


@pytest.mark.parametrize('case', [c for c in DW_CASES if c[4] == 3 and c[5] == 1] + [(3, 9, 7, 8, 3, 1, 1, 'same'),
                                                                                  (1, 129, 129, 24, 3, 1, 1, 'same')])
def test_dwconv_strips_of_two(ops, strips_of_two, case):
    test_dwconv_fwd_bwd(ops, case)
    N, H, W, C, k, s, r, pad = case
    test_dwconv_bwd_data_fused_bn_stats(ops, (N, H, W, C, k, s, r, pad, O.ACT_RELU6))


def test_reduce_rows_batched_is_bitwise_the_per_layer_reduction(ops):
    """dl3p_reduce_rows_batched (every weight gradient of a step in two launches) against dl3p_reduce_rows job by job: both
    kernel variants, ragged element counts, 1 .. 300 rows"""
    L = ops.lib()
    rng = np.random.default_rng(21)
    shapes = [(1, 36), (7, 16 * 96), (16, 70000), (17, 864), (64, 28 * 32), (300, 640), (2, 65536), (33, 9 * 960 + 4)]
    srcs, dsts, refs, rec = [], [], [], []
    st = torch.cuda.current_stream().cuda_stream
    for rows, n in shapes:
        a = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32) * 100).to(DEV)
        ref = torch.empty(n, dtype=torch.float32, device=DEV)
        L.reduce_rows(a.data_ptr(), rows, n, ref.data_ptr(), 0, st)
        out = torch.full((n,), float('nan'), dtype=torch.float32, device=DEV)
        srcs.append(a); dsts.append(out); refs.append(ref)
        rec.append((a.data_ptr(), out.data_ptr(), rows, n))
    jobs = np.array(rec, dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('rows', '<i4'), ('n', '<i4')]))
    maps = ([], [])
    for j, (_, _, rows, n) in enumerate(rec):
        v = L.reduce_rows_variant(rows, n)
        be = L.reduce_rows_block_elements(v)
        maps[v].extend((j, b) for b in range((n + be - 1) // be))
    assert maps[0] and maps[1], 'both variants exercised'
    jt = torch.from_numpy(jobs.view(np.uint8).copy()).to(DEV)
    m0, m1 = (torch.tensor(m, dtype=torch.int32, device=DEV) for m in maps)
    L.reduce_rows_batched(jt.data_ptr(), m0.data_ptr(), len(maps[0]), m1.data_ptr(), len(maps[1]), st)
    for out, ref in zip(dsts, refs):
        assert torch.equal(out, ref)


PW_CASES = [
    # M, K, N
    (2 * 33 * 33, 320, 256), (1000, 24, 144), (777, 144, 24), (4096 + 5, 32, 16), (513, 16, 96),
    (300, 304, 256), (129 * 7, 256, 24), (64, 1280, 256), (16, 320, 256), (2 * 17 * 17, 960, 160),
    (1, 8, 4), (130, 576, 96),
    # small K x N: the wave-independent streaming kernels (ragged last row tile, padded K / N tiles)
    (70001, 32, 32), (5003, 96, 24), (3001, 32, 192), (2005, 192, 32), (4007, 24, 48), (2 * 129 * 129, 144, 32),
]


@pytest.mark.parametrize('case', [(4356, 2048, 256, True), (4356, 1280, 256, True), (1089 * 2, 2048, 256, False), (4356 + 3, 1028, 128, True),
                                  (2 * 33 * 33, 2048, 512, True), (16 * 128, 1536, 256, False)])
def test_pwconv_fwd_splitk_matches_float64_and_the_one_launch_kernel(ops, case):
    """dl3p_pwconv_fwd_wt_splitk (few rows, long reduction: Xception's ASPP 1x1 convs) against float64, with the lazy prologue, the
    bias and the BatchNorm statistic rows; and against dl3p_pwconv_fwd_wt on the same operands (equal to rounding)"""
    M, K, Nn, stats = case
    L = ops.lib()
    S = L.pwconv_fwd_splitk_plan(M, K, Nn)
    assert S > 1, 'the rule serves this shape'
    assert L.pwconv_fwd_splitk_workspace(M, K, Nn) == 4 * S * M * Nn
    rng = np.random.default_rng(M + K + Nn)
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((K, Nn)) / np.sqrt(K)
    b = rng.standard_normal(Nn)
    sc, sh = rng.uniform(0.5, 1.5, K), rng.standard_normal(K) * 0.3
    xa = np.clip(x * sc + sh, 0.0, 6.0)
    ref = xa @ w + b
    wt = T(np.ascontiguousarray(w.T))
    part = ops.new_partials(Nn, DEV) if stats else None
    got = ops.pwconv_fwd_wt_splitk(T(x), wt, T(b), T(sc), T(sh), O.ACT_RELU6, partials=part)
    one = ops.pwconv_fwd_wt(T(x), wt, T(b), T(sc), T(sh), O.ACT_RELU6)
    y = got[0] if stats else got
    close(y, ref, rtol=2e-5, atol=2e-5, what='split-K forward vs float64')
    assert float((y - one).abs().max()) <= 2e-5 * float(one.abs().max())
    if stats:
        rows = got[1]
        assert 1 <= rows <= 2048
        pr = part[:rows * 2 * Nn].reshape(rows, 2, Nn).double().sum(0).cpu().numpy()
        close(pr[0], ref.sum(0), rtol=1e-4, atol=1e-2, what='statistic rows: sum')
        close(pr[1], (ref ** 2).sum(0), rtol=1e-4, atol=1e-2, what='statistic rows: sum of squares')
    # shapes the rule leaves to the one-launch kernel are refused, not silently served
    assert L.pwconv_fwd_splitk_plan(66564, 2048, 256) == 0 and L.pwconv_fwd_splitk_plan(4356, 320, 256) == 0
    with pytest.raises(ops.Dl3pError):
        ops.pwconv_fwd_wt_splitk(T(x[:, :320]), T(np.ascontiguousarray(w[:320].T)))


@pytest.mark.parametrize('case', PW_CASES)
def test_pwconv_fwd_bwd(ops, case):
    M, K, Nn = case
    rng = np.random.default_rng(M * 7 + K * 3 + Nn)
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((K, Nn)) / np.sqrt(K)
    b = rng.standard_normal(Nn)
    sc = rng.uniform(0.5, 1.5, K)
    sh = rng.standard_normal(K) * 0.3
    a = O.act_fwd(x * sc + sh, O.ACT_RELU)
    y_ref = a @ w
    part = ops.new_partials(Nn, DEV)
    y, rows = ops.pwconv_fwd(T(x), T(w), None, T(sc), T(sh), ops.ACT_RELU, partials=part)
    close(y, y_ref, what='pw fwd')
    s1, s2 = stats_from(part, rows, Nn)
    close(s1, y_ref.sum(0), rtol=1e-4, atol=1e-4 * M, what='pw stat sum')
    close(s2, (y_ref ** 2).sum(0), rtol=1e-4, what='pw stat sumsq')
    yb = ops.pwconv_fwd(T(x), T(w), T(b))
    close(yb, x @ w + b, what='pw fwd bias, no prologue')
    gy = rng.standard_normal((M, Nn))
    gx = ops.pwconv_bwd_data(T(gy), T(w))
    close(gx, gy @ w.T, what='pw bwd data')
    base = rng.standard_normal((M, K))
    gx2 = ops.pwconv_bwd_data(T(gy), T(w), out=T(base), accumulate=True)
    close(gx2, gy @ w.T + base, what='pw bwd data accumulate')
    gw, gb = ops.pwconv_bwd_weight(T(x), T(gy), T(sc), T(sh), ops.ACT_RELU, with_bias=True)
    close(gw, a.T @ gy, rtol=3e-4, what='pw bwd weight')
    close(gb, gy.sum(0), rtol=3e-4, what='pw bwd bias')


def test_pwconv_fwd_transposed_kernel(ops):
    """dl3p_transpose_batch + dl3p_pwconv_fwd_wt == dl3p_pwconv_fwd (tiled and small-K.N kernels)"""
    rng = np.random.default_rng(3)
    mats = [(320, 256), (24, 144), (28, 32), (960, 160)]
    offs, off = [], 0
    for K, Nn in mats:
        offs.append(off)
        off += K * Nn
    flat = torch.tensor(rng.standard_normal(off), dtype=torch.float32, device=DEV)
    flat_t = torch.zeros_like(flat)
    table = torch.tensor([[o, K, Nn, 0] for o, (K, Nn) in zip(offs, mats)], dtype=torch.int32, device=DEV)
    ops.transpose_batch(flat, flat_t, table)
    for o, (K, Nn) in zip(offs, mats):
        w = flat[o:o + K * Nn].view(K, Nn)
        wt = flat_t[o:o + K * Nn].view(Nn, K)
        assert torch.equal(wt, w.t().contiguous())
        for M in (777, 3000):
            x = torch.tensor(rng.standard_normal((M, K)), dtype=torch.float32, device=DEV)
            sc = torch.tensor(rng.uniform(0.5, 1.5, K), dtype=torch.float32, device=DEV)
            sh = torch.tensor(rng.standard_normal(K) * 0.3, dtype=torch.float32, device=DEV)
            p1, p2 = ops.new_partials(Nn, DEV), ops.new_partials(Nn, DEV)
            y1, r1 = ops.pwconv_fwd(x, w, None, sc, sh, ops.ACT_RELU6, partials=p1)
            y2, r2 = ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, partials=p2)
            assert torch.equal(y1, y2) and r1 == r2       # same products in the same order
            assert torch.equal(p1[:r1 * 2 * Nn], p2[:r2 * 2 * Nn])


@pytest.mark.parametrize('case', [(16, 960, 240), (16, 240, 960), (17, 72, 24), (1, 24, 72), (40, 160, 256), (64, 672, 168),
                                  (3, 8, 4)])
def test_pwconv_few_rows(ops, case):
    """M <= 64 (convs behind a global pooling: ASPP image pooling, MobileNetV3 squeeze-excite): the row-parallel
    kernels of pw_tiny.hip in all three roles, with prologue, bias, statistics and accumulation"""
    M, K, Nn = case
    rng = np.random.default_rng(M + K + Nn)
    x = rng.standard_normal((M, K)); w = rng.standard_normal((K, Nn)) / np.sqrt(K); b = rng.standard_normal(Nn)
    sc = rng.uniform(0.5, 1.5, K); sh = rng.standard_normal(K) * 0.3
    a = O.act_fwd(x * sc + sh, O.ACT_HSWISH)
    wt = T(np.ascontiguousarray(w.T))
    part = ops.new_partials(Nn, DEV)
    y, rows = ops.pwconv_fwd_wt(T(x), wt, T(b), T(sc), T(sh), ops.ACT_HSWISH, partials=part)
    y_ref = a @ w + b
    close(y, y_ref, what='few-row fwd')
    assert rows == 1
    s1, s2 = stats_from(part, rows, Nn)
    close(s1, y_ref.sum(0), rtol=1e-4, atol=1e-4 * M, what='few-row stat sum')
    close(s2, (y_ref ** 2).sum(0), rtol=1e-4, what='few-row stat sumsq')
    close(ops.pwconv_fwd_wt(T(x), wt), x @ w, what='few-row fwd, bare')
    gy = rng.standard_normal((M, Nn)); base = rng.standard_normal((M, K))
    close(ops.pwconv_bwd_data(T(gy), T(w)), gy @ w.T, what='few-row bwd data')
    close(ops.pwconv_bwd_data(T(gy), T(w), out=T(base), accumulate=True), gy @ w.T + base, what='few-row bwd data accumulate')
    gw, gb = ops.pwconv_bwd_weight(T(x), T(gy), T(sc), T(sh), ops.ACT_HSWISH, with_bias=True)
    close(gw, a.T @ gy, rtol=3e-4, what='few-row bwd weight')
    close(gb, gy.sum(0), rtol=3e-4, what='few-row bwd bias')
    # strided rows (a channel slice of a wider buffer) on both sides
    wide_in = torch.zeros((M, K + 8), device=DEV); wide_in[:, 4:4 + K] = T(x)
    wide_out = torch.zeros((M, Nn + 12), device=DEV)
    ops.pwconv_fwd_wt(wide_in[:, 4:4 + K], wt, out=wide_out[:, 8:8 + Nn])
    close(wide_out[:, 8:8 + Nn], x @ w, what='few-row fwd, strided views')
    assert float(wide_out[:, :8].abs().max()) == 0 and float(wide_out[:, 8 + Nn:].abs().max()) == 0


@pytest.mark.parametrize('case', [(2 * 33 * 33, 96, 576, 2), (1000, 24, 144, 1), (3001, 256, 256, 2), (777, 160, 960, 0)])
def test_pwconv_bwd_data_fused_bn_stats(ops, case):
    """dl3p_pwconv_bwd_data_bn == dl3p_pwconv_bwd_data followed by dl3p_bn_bwd_reduce on its result"""
    M, K, Nn, act = case
    rng = np.random.default_rng(K + Nn)
    dy = T(rng.standard_normal((M, Nn)))
    w = T(rng.standard_normal((K, Nn)) / np.sqrt(Nn))
    z = T(rng.standard_normal((M, K)) * 2)
    sc = T(rng.uniform(0.5, 1.5, K)); sh = T(rng.standard_normal(K) * 0.5)
    mu = T(rng.standard_normal(K) * 0.2); inv = T(rng.uniform(0.5, 2.0, K))
    base = T(rng.standard_normal((M, K)))
    for accumulate in (False, True):
        ref = ops.pwconv_bwd_data(dy, w, out=base.clone() if accumulate else None, accumulate=accumulate)
        p_ref = ops.new_partials(K, DEV)
        rows_ref = ctypes.c_int(0)
        ops.lib().bn_bwd_reduce(ref.data_ptr(), K, z.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), act, mu.data_ptr(),
                                inv.data_ptr(), p_ref.data_ptr(), ctypes.byref(rows_ref), M, K,
                                torch.cuda.current_stream().cuda_stream)
        part = ops.new_partials(K, DEV)
        gx, rows = ops.pwconv_bwd_data_bn(dy, w, z, sc, sh, act, mu, inv, part, out=base.clone() if accumulate else None,
                                          accumulate=accumulate)
        assert torch.equal(gx, ref)
        s_ref = p_ref[:rows_ref.value * 2 * K].view(rows_ref.value, 2, K).double().sum(0).cpu().numpy()
        s_fus = part[:rows * 2 * K].view(rows, 2, K).double().sum(0).cpu().numpy()
        scale = np.abs(s_ref).max(1, keepdims=True) + 1e-6
        assert (np.abs(s_fus - s_ref) / scale).max() < 2e-5, (np.abs(s_fus - s_ref) / scale).max()


@pytest.mark.parametrize('case', [(2, 33, 33, 96, 3, 1, 1, 'same', 2), (1, 65, 65, 144, 3, 2, 1, 'same', 2),
                                  (2, 16, 20, 24, 3, 2, 1, 'same', 1), (2, 33, 33, 64, 3, 1, 2, 'same', 2),
                                  (2, 33, 33, 320, 3, 1, 18, 'same', 2), (1, 16, 24, 40, 5, 1, 1, 'same', 2),
                                  (2, 33, 33, 160, 5, 1, 2, 'same', 2), (1, 65, 47, 120, 5, 1, 1, 'same', 1),
                                  (1, 33, 33, 672, 5, 1, 1, 'same', 2)])
def test_dwconv_bwd_data_fused_bn_stats(ops, case):
    """dl3p_dwconv2d_bwd_data_bn == dl3p_dwconv2d_bwd_data followed by dl3p_bn_bwd_reduce (fused window / quad kernels and
    the two-launch fallback of the other decompositions)"""
    N, H, W, C, k, stride, rate, pad, act = case
    rng = np.random.default_rng(C + H)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, k, stride, rate, pad)
    dy = T(rng.standard_normal((N, Ho, Wo, C)))
    w = T(rng.standard_normal((k, k, C)) * 0.3)
    z = T(rng.standard_normal((N, H, W, C)) * 2)
    sc = T(rng.uniform(0.5, 1.5, C)); sh = T(rng.standard_normal(C) * 0.5)
    mu = T(rng.standard_normal(C) * 0.2); inv = T(rng.uniform(0.5, 2.0, C))
    base = T(rng.standard_normal((N, H, W, C)))
    st = torch.cuda.current_stream().cuda_stream
    for accumulate in (False, True):
        ref = ops.dwconv2d_bwd_data(dy, w, (N, H, W, C), stride, rate, pad, out=base.clone() if accumulate else None,
                                    accumulate=accumulate)
        p_ref = ops.new_partials(C, DEV)
        rows_ref = ctypes.c_int(0)
        ops.lib().bn_bwd_reduce(ref.data_ptr(), C, z.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), act, mu.data_ptr(),
                                inv.data_ptr(), p_ref.data_ptr(), ctypes.byref(rows_ref), N * H * W, C, st)
        part = ops.new_partials(C, DEV)
        gx, rows = ops.dwconv2d_bwd_data_bn(dy, w, (N, H, W, C), z, sc, sh, act, mu, inv, part, stride, rate, pad,
                                            out=base.clone() if accumulate else None, accumulate=accumulate)
        assert torch.equal(gx, ref)
        s_ref = p_ref[:rows_ref.value * 2 * C].view(rows_ref.value, 2, C).double().sum(0).cpu().numpy()
        s_fus = part[:rows * 2 * C].view(rows, 2, C).double().sum(0).cpu().numpy()
        scale = np.abs(s_ref).max(1, keepdims=True) + 1e-6
        assert (np.abs(s_fus - s_ref) / scale).max() < 2e-5, (np.abs(s_fus - s_ref) / scale).max()


def test_pwconv_concat_slices(ops):
    """producers write channel slices of a concat buffer, the consumer reads it with the
    concatenated per-channel prologue (layers.py:155 Concatenate costs no pass)"""
    rng = np.random.default_rng(5)
    M = 2 * 9 * 9
    x = rng.standard_normal((M, 32))
    w1 = rng.standard_normal((32, 48))
    w2 = rng.standard_normal((32, 16))
    cat = torch.zeros((M, 64), device=DEV)
    ops.pwconv_fwd(T(x), T(w1), out=cat[:, :48])
    ops.pwconv_fwd(T(x), T(w2), out=cat[:, 48:])
    ref = np.concatenate([x @ w1, x @ w2], -1)
    close(cat, ref, what='slice writes')
    sc = rng.uniform(0.5, 1.5, 64)
    sh = rng.standard_normal(64)
    w3 = rng.standard_normal((64, 8))
    y = ops.pwconv_fwd(cat, T(w3), None, T(sc), T(sh), ops.ACT_RELU)
    close(y, O.act_fwd(ref * sc + sh, O.ACT_RELU) @ w3, rtol=3e-4, what='concat consumer')
    g = rng.standard_normal((M, 8))
    gcat = torch.zeros((M, 64), device=DEV)
    ops.pwconv_bwd_data(T(g), T(w3), out=gcat)
    gw1 = ops.pwconv_bwd_weight(T(x), gcat[:, :48])
    close(gw1, x.T @ (g @ w3.T)[:, :48], rtol=3e-4, what='wgrad from slice')


@pytest.mark.parametrize('case', [(2, 33, 33, 3, 32, 3, 2, 'same'), (1, 32, 48, 3, 16, 3, 2, (0, 1, 0, 1)),
                                  (1, 17, 17, 8, 16, 3, 1, 'same')])
def test_conv2d_stem(ops, case):
    N, H, W, Cin, Cout, k, s, pad = case
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) * 0.2
    y_ref = O.conv2d_fwd(x, w, s, 1, pad)
    part = ops.new_partials(Cout, DEV)
    y, rows = ops.conv2d_fwd(T(x), T(w), s, 1, pad, partials=part)
    close(y, y_ref, what='conv fwd')
    s1, s2 = stats_from(part, rows, Cout)
    close(s1, y_ref.reshape(-1, Cout).sum(0), rtol=1e-4, atol=1e-2, what='conv stat')
    close(s2, (y_ref ** 2).reshape(-1, Cout).sum(0), rtol=1e-4, what='conv stat sq')
    gy = rng.standard_normal(y_ref.shape)
    gx_ref, gw_ref, _ = O.conv2d_bwd(x, w, gy, s, 1, pad)
    gw = ops.conv2d_bwd_weight(T(x), T(gy), k, s, 1, pad)
    close(gw, gw_ref, rtol=3e-4, what='conv bwd weight')
    gx = ops.conv2d_bwd_data(T(gy), T(w), (N, H, W, Cin), s, 1, pad)
    close(gx, gx_ref, rtol=3e-4, what='conv bwd data')


# (N, H, W, Cin, Cout, k, stride, rate, padding, act)
CONV_GEMM = [(2, 33, 33, 32, 64, 3, 1, 1, 'same', O.ACT_RELU),          # Xception entry_flow_conv1_2
             (2, 33, 33, 64, 128, 1, 2, 1, 'same', O.ACT_NONE),         # strided 1x1 shortcut, odd size
             (1, 32, 40, 128, 256, 1, 2, 1, 'same', O.ACT_NONE),        # ... even size
             (2, 17, 23, 64, 64, 3, 2, 1, 'same', O.ACT_RELU),          # ResNet 3x3 stride 2, odd
             (1, 16, 24, 8, 24, 3, 2, 1, (0, 1, 0, 1), O.ACT_RELU6),    # explicit (0,1) padding + 'valid', even
             (1, 19, 19, 16, 32, 3, 1, 2, 'same', O.ACT_RELU6),         # atrous
             (3, 9, 11, 12, 20, 5, 1, 1, 'same', O.ACT_HSWISH),         # 5x5, channel counts not multiples of 16
             (1, 21, 21, 4, 8, 7, 2, 1, 'same', O.ACT_NONE),            # 7x7 stride 2
             (2, 65, 65, 728, 1024, 1, 2, 1, 'same', O.ACT_NONE)]       # exit-flow shortcut width


@pytest.mark.parametrize('case', CONV_GEMM)
def test_conv2d_implicit_gemm(ops, case):
    """dense convolutions on the GEMM kernels with the patch operand gathered during LDS staging (no im2col / col2im):
    forward + statistics, data gradient (stride-2 parity gaps written as zeros, accumulate), weight gradient"""
    N, H, W, Cin, Cout, k, s, r, pad, act = case
    rng = np.random.default_rng(Cin + Cout + k)
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    sc = rng.uniform(0.5, 1.5, Cin)
    sh = rng.standard_normal(Cin) * 0.3
    a = O.act_fwd(x * sc + sh, act) if act != O.ACT_NONE else x
    pro = (T(sc), T(sh), act) if act != O.ACT_NONE else (None, None, O.ACT_NONE)
    y_ref = O.conv2d_fwd(a, w, s, r, pad)
    part = ops.new_partials(Cout, DEV)
    y, rows = ops.conv2d_gemm_fwd(T(x), T(w), s, r, pad, *pro, partials=part)
    close(y, y_ref, rtol=3e-4, what='implicit conv fwd')
    s1, s2 = stats_from(part, rows, Cout)
    close(s1, y_ref.reshape(-1, Cout).sum(0), rtol=2e-4, atol=1e-2, what='implicit conv stat')
    close(s2, (y_ref ** 2).reshape(-1, Cout).sum(0), rtol=2e-4, what='implicit conv stat sq')
    gy = rng.standard_normal(y_ref.shape)
    gx_ref, gw_ref, _ = O.conv2d_bwd(a, w, gy, s, r, pad)
    gx = ops.conv2d_gemm_bwd_data(T(gy), T(w), (N, H, W, Cin), s, r, pad)
    close(gx, gx_ref, rtol=3e-4, what='implicit conv bwd data')
    base = rng.standard_normal((N, H, W, Cin))
    acc = T(base)
    ops.conv2d_gemm_bwd_data(T(gy), T(w), (N, H, W, Cin), s, r, pad, out=acc, accumulate=True)
    close(acc, base + gx_ref, rtol=3e-4, what='implicit conv bwd data accumulate')
    gw, gb = ops.conv2d_gemm_bwd_weight(T(x), T(gy), k, s, r, pad, *pro, with_bias=True)
    close(gw, gw_ref, rtol=5e-4, what='implicit conv bwd weight')
    close(gb, gy.reshape(-1, Cout).sum(0), rtol=2e-4, atol=1e-3, what='implicit conv bias gradient')
    b = rng.standard_normal(Cout)
    close(ops.conv2d_gemm_fwd(T(x), T(w), s, r, pad, *pro, bias=T(b)), y_ref + b, rtol=3e-4, what='implicit conv fwd + bias')


CONV_GEMM_SB = [(2, 33, 33, 32, 64, 3, 1, 1, 'same', O.ACT_RELU),         # Xception entry_flow_conv1_2
                (1, 65, 65, 64, 64, 3, 1, 1, 'same', O.ACT_RELU),         # ResNet50 stage 2
                (4, 33, 33, 64, 64, 3, 2, 1, 'same', O.ACT_RELU),         # 3x3 stride 2, odd size
                (4, 32, 40, 128, 128, 3, 2, 1, (0, 1, 0, 1), O.ACT_NONE),  # ... even size, explicit padding
                (1, 33, 33, 128, 128, 3, 1, 2, 'same', O.ACT_RELU),       # atrous (ResNet50 at output stride 16 / 8)
                (1, 33, 33, 256, 256, 3, 1, 4, 'same', O.ACT_RELU),
                (2, 35, 37, 36, 72, 3, 1, 1, 'same', O.ACT_RELU6),        # channel counts not multiples of 32, K tail, N tail
                (1, 40, 40, 16, 32, 5, 1, 1, 'same', O.ACT_HSWISH),       # 5x5
                (2, 65, 65, 728, 1024, 1, 2, 1, 'same', O.ACT_NONE)]      # strided 1x1 shortcut


@pytest.mark.parametrize('case', CONV_GEMM_SB)
def test_conv2d_implicit_gemm_split(ops, case):
    """the same implicit GEMMs on the split-bf16 kernels (dl3p_conv2d_gemm_fwd_sb / _bwd_data_sb, and the weight gradient's split
    route inside dl3p_conv2d_gemm_bwd_weight): fp32-accurate, so the fp32 kernels' tolerances"""
    N, H, W, Cin, Cout, k, s, r, pad, act = case
    L = ops.lib()
    L.set_option(b'conv_sb', 2)          # wherever supported (the production rule asks for long GEMMs)
    try:
        rng = np.random.default_rng(Cin + Cout + k)
        x = rng.standard_normal((N, H, W, Cin))
        w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
        sc = rng.uniform(0.5, 1.5, Cin)
        sh = rng.standard_normal(Cin) * 0.3
        a = O.act_fwd(x * sc + sh, act) if act != O.ACT_NONE else x
        pro = (T(sc), T(sh), act) if act != O.ACT_NONE else (None, None, O.ACT_NONE)
        y_ref = O.conv2d_fwd(a, w, s, r, pad)
        Ho, Wo = y_ref.shape[1:3]
        assert L.conv2d_gemm_sb_supported(1, N * Ho * Wo, k * k * Cin, Cout) and L.conv2d_gemm_sb_supported(2, N * H * W, k * k * Cout, Cin)
        part = ops.new_partials(Cout, DEV)
        y, rows = ops.conv2d_gemm_fwd_sb(T(x), T(w), s, r, pad, *pro, partials=part)
        close(y, y_ref, rtol=3e-4, what='split implicit conv fwd')
        s1, s2 = stats_from(part, rows, Cout)
        close(s1, y_ref.reshape(-1, Cout).sum(0), rtol=2e-4, atol=1e-2, what='split implicit conv stat')
        close(s2, (y_ref ** 2).reshape(-1, Cout).sum(0), rtol=2e-4, what='split implicit conv stat sq')
        b = rng.standard_normal(Cout)
        close(ops.conv2d_gemm_fwd_sb(T(x), T(w), s, r, pad, *pro, bias=T(b)), y_ref + b, rtol=3e-4, what='split implicit conv fwd + bias')
        gy = rng.standard_normal(y_ref.shape)
        gx_ref, gw_ref, _ = O.conv2d_bwd(a, w, gy, s, r, pad)
        gx = ops.conv2d_gemm_bwd_data_sb(T(gy), T(w), (N, H, W, Cin), s, r, pad)
        close(gx, gx_ref, rtol=3e-4, what='split implicit conv bwd data')
        base = rng.standard_normal((N, H, W, Cin))
        acc = T(base)
        ops.conv2d_gemm_bwd_data_sb(T(gy), T(w), (N, H, W, Cin), s, r, pad, out=acc, accumulate=True)
        close(acc, base + gx_ref, rtol=3e-4, what='split implicit conv bwd data accumulate')
        assert L.conv2d_gemm_sb_pays(4, N * Ho * Wo, k * k * Cin, Cout)
        gw, gb = ops.conv2d_gemm_bwd_weight(T(x), T(gy), k, s, r, pad, *pro, with_bias=True)
        close(gw, gw_ref, rtol=5e-4, what='split implicit conv bwd weight')
        close(gb, gy.reshape(-1, Cout).sum(0), rtol=2e-4, atol=1e-3, what='implicit conv bias gradient')
        for tile in range(4):           # every tile of the weight-gradient kernel
            L.set_option(b'split_wgrad_tile', tile)
            close(ops.conv2d_gemm_bwd_weight(T(x), T(gy), k, s, r, pad, *pro), gw_ref, rtol=5e-4, what='split implicit conv bwd weight, tile %d' % tile)
    finally:
        L.set_option(b'conv_sb', -1)
        L.set_option(b'split_wgrad_tile', -1)


@pytest.mark.parametrize('case', [(2, 33, 33, 32, 'same'), (1, 32, 48, 16, (0, 1, 0, 1)), (3, 65, 129, 32, 'same'),
                                  (1, 64, 258, 16, 'same'), (2, 17, 263, 32, (0, 1, 0, 1)), (1, 3, 3, 32, 'same'),
                                  (1, 130, 513, 32, 'same')])
def test_stem_conv_direct(ops, case):
    """the LDS-staged implicit GEMM of the RGB stem (csrc/stem.hip) against the oracle's dense convolution: row segments
    of 64 output pixels, so widths around the 64 / 128 boundaries, one-pixel tails, even sizes with Keras' explicit
    (0,1) padding (deeplabv3p_mobilenetv3.py ZeroPadding2D(correct_pad) + 'valid')"""
    N, H, W, Cout, pad = case
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, (N, H, W, 3))
    w = rng.standard_normal((3, 3, 3, Cout)) * 0.2
    y_ref = O.conv2d_fwd(x, w, 2, 1, pad)
    part = ops.new_partials(Cout, DEV)
    y, rows = ops.stem_conv_fwd(T(x), T(w), pad, partials=part)
    close(y, y_ref, what='stem fwd')
    s1, s2 = stats_from(part, rows, Cout)
    close(s1, y_ref.reshape(-1, Cout).sum(0), rtol=1e-4, atol=1e-2, what='stem stat')
    close(s2, (y_ref ** 2).reshape(-1, Cout).sum(0), rtol=1e-4, what='stem stat sq')
    assert torch.equal(ops.stem_conv_fwd(T(x), T(w), pad), y), 'statistics epilogue changes the product'
    gy = rng.standard_normal(y_ref.shape)
    _, gw_ref, _ = O.conv2d_bwd(x, w, gy, 2, 1, pad)
    gw = ops.stem_conv_bwd_weight(T(x), T(gy), pad)
    close(gw, gw_ref, rtol=3e-4, what='stem bwd weight')
    # and against the im2col route it replaces
    close(gw, ops.conv2d_bwd_weight(T(x), T(gy), 3, 2, 1, pad).cpu().numpy().astype(np.float64), rtol=3e-4,
          what='stem bwd weight vs im2col route')


@pytest.mark.parametrize('case', [(2, 65, 65, 32, O.ACT_RELU6), (1, 130, 129, 16, O.ACT_HSWISH), (3, 33, 47, 32, O.ACT_NONE)])
def test_stem_weight_gradient_with_the_batchnorm_apply_folded_in(ops, case):
    """dl3p_stem_conv_bwd_weight_slabs_bn == dl3p_bn_bwd_apply followed by dl3p_stem_conv_bwd_weight (the stem has no data gradient:
    dz is formed while the gradient rows are staged and never written), and == float64"""
    N, H, W, Cout, act = case
    rng = np.random.default_rng(H + Cout)
    x = rng.uniform(-1, 1, (N, H, W, 3))
    w = rng.standard_normal((3, 3, 3, Cout)) * 0.3
    gamma, beta = rng.uniform(0.5, 1.5, Cout), rng.standard_normal(Cout) * 0.2
    part = ops.new_partials(Cout, DEV)
    z, rows = ops.stem_conv_fwd(T(x), T(w), 'same', partials=part)
    bn = ops.BNState(Cout, DEV, 1e-3, 0.99)
    bn.gamma.copy_(T(gamma)); bn.beta.copy_(T(beta))
    M = z.numel() // Cout
    ops.bn_finalize(bn, part, rows, M)
    g = T(rng.standard_normal(tuple(z.shape)))
    dz = ops.bn_backward(bn, g, z, act, part, out=torch.empty_like(g))        # leaves bn.coef; g untouched
    two = ops.stem_conv_bwd_weight(T(x), dz, 'same')
    one = ops.stem_conv_bwd_weight_bn(T(x), g, z, bn, act, 'same')
    scale = float(two.abs().max())
    assert float((one - two).abs().max()) <= 2e-5 * scale, (float((one - two).abs().max()), scale)
    z64 = O.conv2d_fwd(x, w, 2, 1, 'same')
    y_ref, cache, _ = O.bn_train_fwd(z64, gamma, beta, 1e-3)
    gz_ref, _, _ = O.bn_train_bwd(O.act_bwd(y_ref, g.cpu().numpy().astype(np.float64), act), cache)
    _, gw_ref, _ = O.conv2d_bwd(x, w, gz_ref, 2, 1, 'same')
    close(one, gw_ref, rtol=1e-3, atol=2e-4 * float(np.abs(gw_ref).max()), what='folded stem weight gradient vs float64')


@pytest.mark.parametrize('shape,act', [((2, 9, 9, 32), O.ACT_RELU6), ((3, 1, 1, 256), O.ACT_RELU),
                                       ((1, 33, 33, 24), O.ACT_NONE), ((2, 8, 8, 64), O.ACT_HSWISH)])
def test_batchnorm_fwd_bwd(ops, shape, act):
    rng = np.random.default_rng(3)
    C = shape[-1]
    z = rng.standard_normal(shape) * 2 + 0.5
    gamma = rng.uniform(0.5, 1.5, C)
    beta = rng.standard_normal(C) * 0.2
    eps, mom = 1e-3, 0.99
    y_ref, cache, (bm, bv) = O.bn_train_fwd(z, gamma, beta, eps)
    bn = ops.BNState(C, DEV, eps, mom)
    bn.gamma.copy_(T(gamma)); bn.beta.copy_(T(beta))
    # statistics produced by a producer kernel: identity depthwise conv is the simplest producer
    part = ops.new_partials(C, DEV)
    w = np.zeros((3, 3, C)); w[1, 1] = 1
    zz, rows = ops.dwconv2d_fwd(T(z), T(w), partials=part)
    M = z.size // C
    ops.bn_finalize(bn, part, rows, M)
    close(bn.scale, gamma * cache[1], what='scale')
    close(bn.shift, beta - z.reshape(-1, C).mean(0) * gamma * cache[1], what='shift', atol=1e-4)
    close(bn.moving_mean, 0 * mom + bm * (1 - mom), what='moving mean', atol=1e-5)
    close(bn.moving_var, 1 * mom + bv * (1 - mom), what='moving var', atol=1e-5)
    # update_moving = 2: the Bessel-corrected variance of the fused BatchNormalization (np_ops.bn_moving_variance_of)
    bn.moving_mean.zero_(); bn.moving_var.fill_(1.0)
    ops.bn_finalize(bn, part, rows, M, update_moving=2)
    close(bn.moving_var, 1 * mom + O.bn_moving_variance_of(bv, M, 'unbiased') * (1 - mom), what='moving var (unbiased)', atol=1e-5)
    close(bn.moving_mean, 0 * mom + bm * (1 - mom), what='moving mean', atol=1e-5)
    close(bn.scale, gamma * cache[1], what='scale (unchanged by the moving rule)')
    a = ops.affine_act(zz, bn.scale, bn.shift, act)
    close(a, O.act_fwd(y_ref, act), rtol=3e-4, atol=1e-4, what='bn apply + act')
    g = rng.standard_normal(shape)
    gy = O.act_bwd(y_ref, g, act)
    gz_ref, gg_ref, gb_ref = O.bn_train_bwd(gy, cache)
    dz = ops.bn_backward(bn, T(g), T(z), act, part)
    close(dz, gz_ref, rtol=5e-4, atol=5e-4, what='bn bwd dz')
    close(bn.dgamma, gg_ref, rtol=5e-4, atol=5e-4, what='dgamma')
    close(bn.dbeta, gb_ref, rtol=5e-4, atol=5e-4, what='dbeta')
    # frozen / inference mode
    bn.moving_mean.copy_(T(rng.standard_normal(C))); bn.moving_var.copy_(T(rng.uniform(0.5, 2, C)))
    ops.bn_infer_coeffs(bn)
    mm, mv = bn.moving_mean.cpu().numpy().astype(np.float64), bn.moving_var.cpu().numpy().astype(np.float64)
    a2 = ops.affine_act(T(z), bn.scale, bn.shift, act)
    y2 = O.bn_infer_fwd(z, gamma, beta, mm, mv, eps)
    close(a2, O.act_fwd(y2, act), rtol=3e-4, atol=1e-4, what='bn infer')
    dz2 = ops.bn_backward(bn, T(g), T(z), act, part, frozen=True)
    close(dz2, O.act_bwd(y2, g, act) * (gamma / np.sqrt(mv + eps)), rtol=3e-4, atol=1e-4, what='bn frozen bwd')


@pytest.mark.parametrize('M,K,N,act', [(1000, 64, 384, O.ACT_RELU6), (2 * 33 * 33, 64, 384, O.ACT_RELU6), (333, 32, 256, O.ACT_RELU),
                                       (777, 144, 24, O.ACT_NONE), (70001, 16, 96, O.ACT_RELU6), (5003, 96, 24, O.ACT_NONE),
                                       (3001, 32, 192, O.ACT_RELU6), (2005, 192, 32, O.ACT_NONE), (4007, 24, 48, O.ACT_HSWISH),
                                       (2 * 129 * 129, 24, 144, O.ACT_RELU6), (65, 16, 16, O.ACT_RELU)])
def test_pwconv_bwd_weight_with_folded_bn_apply(ops, M, K, N, act):
    """dl3p_pwconv_bwd_weight_slabs_bn = dl3p_bn_bwd_apply + dl3p_pwconv_bwd_weight: same dz (written once), same gw"""
    L = ops.lib()
    if (K, N) == (32, 192):
        assert L.pwconv_bwd_weight_bn_supported(M, K, N) == 0, 'the (2, 12)-tile streaming kernel has no registers for the fold'
        pytest.skip('not served')
    assert L.pwconv_bwd_weight_bn_supported(M, K, N) == 1
    assert L.pwconv_bwd_weight_bn_supported(2 * 33 * 33, 320, 256) == 0, 'several k tiles: the apply pass stays'
    rng = np.random.default_rng(31)
    x = rng.standard_normal((M, K))
    z = rng.standard_normal((M, N)) * 1.5 + 0.3
    g = rng.standard_normal((M, N))
    gamma, beta = rng.uniform(0.5, 1.5, N), rng.standard_normal(N) * 0.3
    eps = 1e-3
    y_ref, cache, _ = O.bn_train_fwd(z, gamma, beta, eps)
    gz_ref, _, _ = O.bn_train_bwd(O.act_bwd(y_ref, g, act), cache)
    bn = ops.BNState(N, DEV, eps, 0.99)
    bn.gamma.copy_(T(gamma)); bn.beta.copy_(T(beta))
    part = ops.new_partials(N, DEV)
    # statistics of z through the producer's epilogue, then the backward sums and coefficients; apply stays out
    w1 = np.zeros((3, 3, N)); w1[1, 1] = 1
    _, rows = ops.dwconv2d_fwd(T(z.reshape(1, 1, M, N)), T(w1), partials=part)
    ops.bn_finalize(bn, part, rows, M)
    gt, zt, xt = T(g), T(z), T(x)
    dz_sep = ops.bn_backward(bn, gt.clone(), zt, act, part)            # reduce + finalize (bn.coef) + apply
    gw_sep = ops.pwconv_bwd_weight(xt, dz_sep)
    gw, dz = ops.pwconv_bwd_weight_bn(xt, gt, zt, bn, act)
    close(dz, gz_ref, rtol=5e-4, atol=5e-4, what='folded dz vs oracle')
    close(dz, dz_sep.cpu().numpy().astype(np.float64), rtol=1e-5, atol=2e-6, what='folded dz vs apply kernel')
    close(gw, x.T @ gz_ref, rtol=3e-4, atol=3e-3, what='folded gw vs oracle')
    close(gw, gw_sep.cpu().numpy().astype(np.float64), rtol=1e-4, atol=1e-4, what='folded gw vs separate')
    gw2, none = ops.pwconv_bwd_weight_bn(xt, gt, zt, bn, act, want_dz=False)
    assert none is None and torch.equal(gw2, gw)


@pytest.mark.parametrize('shape,stride,rate,act', [((2, 33, 33, 64), 1, 1, O.ACT_RELU6), ((1, 65, 65, 32), 1, 1, O.ACT_RELU),
                                                   ((2, 34, 30, 24), 2, 1, O.ACT_RELU6), ((2, 33, 33, 32), 1, 2, O.ACT_NONE),
                                                   ((1, 129, 129, 16), 1, 1, O.ACT_HSWISH), ((3, 17, 19, 96), 2, 1, O.ACT_NONE),
                                                   ((2, 40, 40, 304), 1, 1, O.ACT_RELU)])
def test_dwconv_bwd_weight_with_folded_bn_apply(ops, shape, stride, rate, act):
    """dl3p_dwconv2d_bwd_weight_slabs_bn = dl3p_bn_bwd_apply + dl3p_dwconv2d_bwd_weight: same dz (every output pixel written
    once), same gw"""
    L = ops.lib()
    N, H, W, C = shape
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, 3, stride, rate, 'same')
    if not L.dwconv2d_bwd_weight_bn_supported(N, H, W, C, 3, stride, rate, pt, pl, Ho, Wo):
        pytest.skip('geometry served by the gather kernel')
    rng = np.random.default_rng(33)
    x = rng.standard_normal(shape)
    z = rng.standard_normal((N, Ho, Wo, C)) * 1.5 + 0.3
    g = rng.standard_normal((N, Ho, Wo, C))
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3
    isc, ish = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.2
    eps = 1e-3
    y_ref, cache, _ = O.bn_train_fwd(z, gamma, beta, eps)
    gz_ref, _, _ = O.bn_train_bwd(O.act_bwd(y_ref, g, act), cache)
    bn = ops.BNState(C, DEV, eps, 0.99)
    bn.gamma.copy_(T(gamma)); bn.beta.copy_(T(beta))
    part = ops.new_partials(C, DEV)
    w1 = np.zeros((3, 3, C)); w1[1, 1] = 1
    _, rows = ops.dwconv2d_fwd(T(z), T(w1), partials=part)
    ops.bn_finalize(bn, part, rows, z.size // C)
    gt, zt, xt = T(g), T(z), T(x)
    dz_sep = ops.bn_backward(bn, gt.clone(), zt, act, part)
    kw = dict(in_scale=T(isc), in_shift=T(ish), in_act=O.ACT_RELU6)
    gw_sep = ops.dwconv2d_bwd_weight(xt, dz_sep, 3, stride, rate, **kw)
    gw, dz = ops.dwconv2d_bwd_weight_bn(xt, gt, zt, bn, act, 3, stride, rate, **kw)
    close(dz, gz_ref, rtol=5e-4, atol=5e-4, what='folded dz vs oracle')
    close(dz, dz_sep.cpu().numpy().astype(np.float64), rtol=1e-5, atol=2e-6, what='folded dz vs apply kernel')
    close(gw, gw_sep.cpu().numpy().astype(np.float64), rtol=1e-4, atol=1e-4, what='folded gw vs separate')
    gw2, none = ops.dwconv2d_bwd_weight_bn(xt, gt, zt, bn, act, 3, stride, rate, want_dz=False, **kw)
    assert none is None and torch.equal(gw2, gw)


@pytest.mark.parametrize('ratio', [30.0, 300.0])
def test_batchnorm_variance_when_mean_dwarfs_sigma(ops, ratio):
    """E[x^2] - E[x]^2 loses digits when |mean| >> sigma.  The reference computes exactly that in float32
    (Keras SyncBatchNormalization: reduce_sum(y), reduce_sum(square(y)), variance = E[y^2] - mean^2; layers.py:19-26 make it
    the CustomBatchNormalization); here the per-workgroup partial sums are float32 and everything above them is double.
    The kernel has to be at least as close to the float64 statistics as that float32 formula is."""
    rng = np.random.default_rng(17)
    N, H, W, C = 4, 33, 33, 32
    sigma = 0.02
    mean = ratio * sigma * rng.choice([-1.0, 1.0], C)
    z = (rng.standard_normal((N, H, W, C)) * sigma + mean).astype(np.float32).astype(np.float64)
    M = z.size // C
    eps = 1e-5
    var64 = z.reshape(-1, C).var(0)
    inv64 = 1.0 / np.sqrt(var64 + eps)
    # the reference's formula carried out in float32 (pairwise sums, as NumPy / Eigen do them)
    z32 = z.reshape(-1, C).astype(np.float32)
    m32 = z32.sum(0, dtype=np.float32) / np.float32(M)
    v32 = np.maximum((z32 * z32).sum(0, dtype=np.float32) / np.float32(M) - m32 * m32, 0)
    inv32 = 1.0 / np.sqrt(v32.astype(np.float64) + eps)
    bn = ops.BNState(C, DEV, eps, 0.99)
    part = ops.new_partials(C, DEV)
    w = np.zeros((3, 3, C)); w[1, 1] = 1
    _, rows = ops.dwconv2d_fwd(T(z), T(w), partials=part)
    ops.bn_finalize(bn, part, rows, M)
    inv = bn.invstd.cpu().numpy().astype(np.float64)
    err = np.abs(inv / inv64 - 1).max()
    err_ref = np.abs(inv32 / inv64 - 1).max()
    assert err <= max(2 * err_ref, 2e-6), (err, err_ref)
    assert err < 1e-2 * (ratio / 300.0) ** 2 + 1e-5, err          # measured: 1/sigma off by 0.4 % at mean = 300 sigma, 4e-5 at 30
    close(bn.mean, z.reshape(-1, C).mean(0), rtol=1e-6, what='mean')


def test_residual_dropout(ops):
    rng = np.random.default_rng(8)
    M, C = 500, 48
    x = rng.standard_normal((M, C)); r = rng.standard_normal((M, C))
    sc = rng.uniform(0.5, 1.5, C); sh = rng.standard_normal(C)
    y = ops.affine_act(T(x), T(sc), T(sh), ops.ACT_NONE, residual=T(r))
    close(y, x * sc + sh + r, what='residual add')
    step = torch.tensor([3], dtype=torch.int64, device=DEV)
    mask = ops.dropout_mask((M, C), 0.5, 1234, step, DEV).cpu().numpy()
    assert 0.4 < mask.mean() < 0.6
    y = ops.affine_act(T(x), T(sc), T(sh), ops.ACT_RELU, dropout_rate=0.5, seed=1234, step_counter=step)
    close(y, O.dropout_fwd(O.act_fwd(x * sc + sh, O.ACT_RELU), mask, 0.5), what='dropout fwd')
    g = rng.standard_normal((M, C))
    gx = ops.scale_mask_bwd(T(g), 0.5, 1234, step)
    close(gx, O.dropout_bwd(g, mask, 0.5), what='dropout bwd')
    step2 = torch.tensor([4], dtype=torch.int64, device=DEV)
    mask2 = ops.dropout_mask((M, C), 0.5, 1234, step2, DEV).cpu().numpy()
    assert (mask != mask2).mean() > 0.3


def test_global_avgpool(ops):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 33, 33, 320))
    sc = rng.uniform(0.5, 1.5, 320); sh = rng.standard_normal(320)
    y = ops.global_avgpool_fwd(T(x), T(sc), T(sh), ops.ACT_NONE)
    close(y, O.global_avgpool_fwd(x * sc + sh), what='gap fwd')
    ys = ops.global_avgpool_fwd(T(x), out_scale=33 * 33)
    close(ys, x.sum((1, 2), keepdims=True), rtol=3e-4, what='gap sum')
    g = rng.standard_normal((3, 1, 1, 320))
    gx = ops.global_avgpool_bwd(T(g), 33, 33)
    close(gx, O.global_avgpool_bwd(g, 33, 33), what='gap bwd')


@pytest.mark.parametrize('shape', [(4, 65, 65, 120), (16, 33, 33, 960), (3, 33, 33, 672), (2, 65, 65, 72), (1, 5, 3, 8)])
def test_chunked_per_image_reductions(ops, shape):
    """pooling and the SE backward split images into pixel chunks (tickets + partial rows in a workspace): same
    numbers as the one-workgroup-per-image path, bit-identical from call to call, tickets left at zero"""
    L = ops.lib()
    N, H, W, C = shape
    rng = np.random.default_rng(C)
    x = rng.standard_normal(shape); s = rng.standard_normal((N, 1, 1, C)) * 2
    sc = rng.uniform(0.5, 1.5, C); sh = rng.standard_normal(C) * 0.3
    gy = rng.standard_normal(shape)
    xt, st, sct, sht, gyt = T(x), T(s), T(sc), T(sh), T(gy)
    nbytes = L.pool_workspace(N, H * W, C)
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=DEV)
    ys, gss = [], []
    for _ in range(3):                                   # the same workspace, call after call
        y = torch.empty((N, 1, 1, C), dtype=torch.float32, device=DEV)
        L.global_avgpool_fwd(xt.data_ptr(), C, sct.data_ptr(), sht.data_ptr(), ops.ACT_HSWISH, y.data_ptr(), C, 1.0, N,
                             H * W, C, ws.data_ptr(), nbytes, ops._stream())
        gx = torch.empty(shape, dtype=torch.float32, device=DEV)
        gs = torch.empty((N, 1, 1, C), dtype=torch.float32, device=DEV)
        L.scale_bcast_bwd(gyt.data_ptr(), C, xt.data_ptr(), C, sct.data_ptr(), sht.data_ptr(), ops.ACT_HSWISH,
                          st.data_ptr(), C, ops.ACT_HSIGMOID, gx.data_ptr(), C, 0, gs.data_ptr(), C, N, H * W, C,
                          ws.data_ptr(), nbytes, ops._stream())
        ys.append(y.cpu().numpy()); gss.append(gs.cpu().numpy())
    assert all(np.array_equal(ys[0], v) for v in ys[1:]) and all(np.array_equal(gss[0], v) for v in gss[1:])
    tickets = ws[:N * 64].view(torch.int32)[:N].cpu().numpy()
    assert (tickets == 0).all()
    a = O.act_fwd(x * sc + sh, O.ACT_HSWISH)
    close(ys[0], a.mean((1, 2), keepdims=True), rtol=3e-4, atol=1e-5, what='chunked gap')
    close(gss[0], (gy * a).sum((1, 2), keepdims=True), rtol=3e-4, atol=1e-4, what='chunked se bwd s')
    close(gx, gy * O.act_fwd(s, O.ACT_HSIGMOID), what='chunked se bwd x')
    y1 = ops.global_avgpool_fwd(xt, sct, sht, ops.ACT_HSWISH, chunked=False)
    _, gs1 = ops.scale_bcast_bwd(gyt, xt, st, sct, sht, ops.ACT_HSWISH, ops.ACT_HSIGMOID, chunked=False)
    close(ys[0], y1.cpu().numpy(), rtol=1e-5, atol=1e-6, what='chunked vs whole-image gap')
    close(gss[0], gs1.cpu().numpy(), rtol=1e-4, atol=1e-4, what='chunked vs whole-image se bwd')
    yf = ops.scale_bcast_fwd(xt, st, sct, sht, ops.ACT_HSWISH, ops.ACT_HSIGMOID)
    close(yf, a * O.act_fwd(s, O.ACT_HSIGMOID), what='se multiply')


@pytest.mark.parametrize('case', [(2, 33, 33, 256, 129, 129), (1, 1, 1, 256, 33, 33), (2, 9, 13, 24, 33, 50),
                                  (1, 129, 129, 24, 513, 513), (1, 16, 32, 8, 64, 128), (1, 10, 10, 4, 7, 5)])
def test_resize_bilinear(ops, case):
    N, h, w, C, H, W = case
    rng = np.random.default_rng(12)
    x = rng.standard_normal((N, h, w, C))
    y = ops.resize_bilinear_fwd(T(x), H, W)
    close(y, O.resize_bilinear_fwd(x, H, W), what='resize fwd')
    g = rng.standard_normal((N, H, W, C))
    gx = ops.resize_bilinear_bwd(T(g), h, w)
    close(gx, O.resize_bilinear_bwd(g, h, w), rtol=3e-4, what='resize bwd')


@pytest.mark.parametrize('C,ignore', [(21, 255), (19, 255), (21, 0), (2, 255), (5, 255), (13, 255), (27, 255), (32, 255),
                                      (33, 255), (150, 255), (253, 255)])      # > 32 classes: the class-walking kernels (train.py:34)
def test_head_softmax_ce(ops, C, ignore):
    rng = np.random.default_rng(C)
    N, h, w, H, W = 2, 9, 9, 33, 33
    cp = ((C + 3) // 4) * 4
    z = np.zeros((N, h, w, cp)); z[..., :C] = rng.standard_normal((N, h, w, C)) * 3
    lab = rng.integers(0, C, (N, H, W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.1] = 255
    big = O.resize_bilinear_fwd(z[..., :C], H, W)
    loss_ref, p_ref, g_ref = O.sparse_ce_fwd_bwd(big, lab, ignore)
    out = ops.upsample_softmax_ce(T(z), C, H, W, T(lab.reshape(N, H * W, 1)), ignore, want_probs=True,
                                  want_logits=True, want_grad=True)
    close(out['logits'][..., :C], big, what='pred_resize logits')
    close(out['probs'], p_ref, rtol=1e-4, atol=1e-6, what='probs')
    close(out['loss'], [loss_ref], rtol=1e-4, what='loss')
    close(out['dlogits'][..., :C], g_ref, rtol=1e-4, atol=1e-9, what='dlogits')
    assert C == cp or float(out['dlogits'][..., C:].abs().max()) == 0.0


@pytest.mark.parametrize('spec', [('weighted',), ('focal', 2.0, 0.25), ('focal', 1.0, 0.5), ('focal', 3.5, 1.0)])
def test_head_optional_losses(ops, spec):
    """the reference's other two training losses in the same head kernel (train.py:108-137): class-weighted CE
    (loss.py:159-191) and softmax focal loss (loss.py:63-118), value and gradient against the oracle"""
    rng = np.random.default_rng(11)
    N, h, w, H, W, C = 2, 9, 9, 33, 33, 21
    z = np.zeros((N, h, w, 24)); z[..., :C] = rng.standard_normal((N, h, w, C)) * 3
    z[0, 0, 0, 3] = 60.0                                     # a saturated pixel: p_y -> 1 (label 3) or -> 0 (others)
    lab = rng.integers(0, C, (N, H, W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.1] = 255
    lab[0, 0, 0], lab[0, 0, 1] = 3, 5
    big = O.resize_bilinear_fwd(z[..., :C], H, W)
    if spec[0] == 'weighted':
        wts = rng.uniform(0.2, 3.0, C)
        ospec, dspec = ('weighted', wts), ('weighted', T(wts))
    else:
        ospec = dspec = spec
    loss_ref, _, g_ref = O.loss_fwd_bwd(big, lab, ospec, 255)
    out = ops.upsample_softmax_ce(T(z), C, H, W, T(lab.reshape(N, H * W, 1)), 255, want_grad=True, loss=dspec)
    close(out['loss'], [loss_ref], rtol=2e-4, what='%s loss' % spec[0])
    close(out['dlogits'][..., :C], g_ref, rtol=2e-4, atol=1e-9, what='%s dlogits' % spec[0])
    assert float(out['dlogits'][..., C:].abs().max()) == 0.0
    assert np.isfinite(out['dlogits'].cpu().numpy()).all()


@pytest.mark.parametrize('spec', [('ce',), ('focal', 2.0, 0.25)])
def test_head_sample_weights(ops, spec):
    """Keras sample weights in 'temporal' mode (train.py:116-120, deeplabv3p/data.py:140-152): per-pixel weights on the
    loss and its gradient"""
    rng = np.random.default_rng(13)
    N, h, w, H, W, C = 2, 5, 5, 17, 17, 21
    z = np.zeros((N, h, w, 24)); z[..., :C] = rng.standard_normal((N, h, w, C)) * 2
    lab = rng.integers(0, C, (N, H, W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.1] = 255
    sw = rng.uniform(0.3, 3.0, (N, H, W))
    big = O.resize_bilinear_fwd(z[..., :C], H, W)
    loss_ref, _, g_ref = O.loss_fwd_bwd(big, lab, spec, 255, sample_weight=sw)
    out = ops.upsample_softmax_ce(T(z), C, H, W, T(lab.reshape(N, H * W, 1)), 255, want_grad=True, loss=spec,
                                  pixel_weights=T(sw.reshape(N, H * W)))
    close(out['loss'], [loss_ref], rtol=2e-4, what='weighted loss')
    close(out['dlogits'][..., :C], g_ref, rtol=2e-4, atol=1e-9, what='weighted dlogits')
    ones = ops.upsample_softmax_ce(T(z), C, H, W, T(lab.reshape(N, H * W, 1)), 255, want_grad=True, loss=spec,
                                   pixel_weights=torch.ones(N * H * W, device=DEV))
    plain = ops.upsample_softmax_ce(T(z), C, H, W, T(lab.reshape(N, H * W, 1)), 255, want_grad=True, loss=spec)
    assert torch.equal(ones['dlogits'], plain['dlogits']) and torch.equal(ones['loss'], plain['loss'])


@pytest.mark.parametrize('case', [(2, 9, 9, 33, 33, 21, 255), (1, 33, 33, 129, 129, 19, 255), (2, 17, 23, 65, 89, 21, 0),
                                  (1, 129, 129, 513, 513, 21, 255)])
def test_head_train_fused(ops, case):
    """fused training head == softmax-CE head followed by the transposed pred_resize, and == the oracle"""
    N, h, w, H, W, C, ignore = case
    assert ops.head_train_supported(h, w, C, H, W)
    rng = np.random.default_rng(C + h)
    cp = ((C + 3) // 4) * 4
    z = np.zeros((N, h, w, cp)); z[..., :C] = rng.standard_normal((N, h, w, C)) * 3
    lab = rng.integers(0, C, (N, H, W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.1] = 255
    labels = T(lab.reshape(N, H * W, 1))
    loss, gz = ops.head_train(T(z), C, H, W, labels, ignore)
    two = ops.upsample_softmax_ce(T(z), C, H, W, labels, ignore, want_grad=True)
    gz2 = ops.resize_bilinear_bwd(two['dlogits'], h, w)
    assert torch.equal(gz, gz2), float((gz - gz2).abs().max())          # same summation order -> same bits
    close(loss, two['loss'].cpu().numpy(), rtol=1e-5, what='fused loss vs two-kernel loss')
    if H * W <= 129 * 129:
        big = O.resize_bilinear_fwd(z[..., :C], H, W)
        loss_ref, _, g_ref = O.sparse_ce_fwd_bwd(big, lab, ignore)
        close(loss, [loss_ref], rtol=1e-4, what='fused loss')
        close(gz[..., :C], O.resize_bilinear_bwd(g_ref, h, w), rtol=1e-4, atol=1e-9, what='fused d loss / d z')
    assert not ops.head_train_supported(33, 33, 21, 513, 513)


@pytest.mark.parametrize('case', [(2, 9, 9, 33, 33, 21, 255), (1, 33, 33, 129, 129, 19, 255), (2, 17, 23, 65, 89, 21, 0),
                                  (3, 5, 7, 5, 7, 18, 255), (2, 11, 6, 41, 21, 32, 255), (1, 129, 129, 513, 513, 21, 255),
                                  (1, 97, 97, 385, 385, 21, 255), (1, 17, 33, 65, 130, 21, 255), (2, 2, 2, 3, 2, 21, 255),
                                  # rows wider than LDS holds: column segments (769 x 769 at OS 8 / 1024 x 2048 map widths; 2, 4 and 3 segments)
                                  (1, 33, 193, 129, 769, 19, 255), (1, 17, 512, 65, 2048, 19, 255), (2, 9, 300, 33, 1199, 30, 0)])
def test_head_train_rows_form(ops, case):
    """the row-walking fused training head == the two-kernel head to rounding (x-then-y summation), and == the oracle"""
    N, h, w, H, W, C, ignore = case
    assert ops.head_train_rows_supported(h, w, C, H, W)
    rng = np.random.default_rng(C + h)
    cp = ((C + 3) // 4) * 4
    z = np.zeros((N, h, w, cp)); z[..., :C] = rng.standard_normal((N, h, w, C)) * 3
    lab = rng.integers(0, C, (N, H, W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.1] = 255
    labels = T(lab.reshape(N, H * W, 1))
    loss, gz = ops.head_train(T(z), C, H, W, labels, ignore, rows_form=True)
    two = ops.upsample_softmax_ce(T(z), C, H, W, labels, ignore, want_grad=True)
    gz2 = ops.resize_bilinear_bwd(two['dlogits'], h, w)
    scale = float(gz2.abs().max())
    assert float((gz - gz2).abs().max()) <= 4e-6 * scale, (float((gz - gz2).abs().max()), scale)
    assert float(gz[..., C:].abs().max()) == 0.0 if cp > C else True
    close(loss, two['loss'].cpu().numpy(), rtol=1e-5, what='rows-form loss vs two-kernel loss')
    if H * W <= 129 * 129:
        big = O.resize_bilinear_fwd(z[..., :C], H, W)
        loss_ref, _, g_ref = O.sparse_ce_fwd_bwd(big, lab, ignore)
        close(loss, [loss_ref], rtol=1e-4, what='rows-form loss')
        close(gz[..., :C], O.resize_bilinear_bwd(g_ref, h, w), rtol=1e-4, atol=1e-9, what='rows-form d loss / d z')
    # accumulate: a second launch on top of the first doubles the gradient
    L = ops.lib()
    import ctypes as ct
    part = torch.zeros(4096, device=DEV); rows = ct.c_int(0)
    zt = T(z)
    wsb = L.head_train_rows_workspace(N, h, w, C, H, W)
    assert wsb == 4 * N * H * w * cp
    ws = torch.empty(wsb // 4, device=DEV)
    L.head_train_rows(zt.data_ptr(), cp, labels.data_ptr(), int(ignore or 0), 1.0 / (N * H * W), gz.data_ptr(), cp, 1,
                      part.data_ptr(), ct.byref(rows), ws.data_ptr(), wsb, N, h, w, C, H, W, None)
    torch.cuda.synchronize()
    assert float((gz - 2 * gz2).abs().max()) <= 8e-6 * scale
    assert 1 <= rows.value <= 2048                        # (row chunks x column segments; never more than the loss partial rows)
    with pytest.raises(ops.Dl3pError):
        L.head_train_rows(zt.data_ptr(), cp, labels.data_ptr(), int(ignore or 0), 1.0 / (N * H * W), gz.data_ptr(), cp, 1,
                          part.data_ptr(), ct.byref(rows), ws.data_ptr(), wsb - 16, N, h, w, C, H, W, None)
    assert not ops.head_train_rows_supported(33, 33, 21, 513, 513) and not ops.head_train_rows_supported(9, 9, 40, 33, 33)
    assert ops.head_train_rows_supported(256, 512, 19, 1024, 2048) and ops.head_train_rows_supported(193, 193, 19, 769, 769)


def test_se_multiply_and_bare_activation(ops):
    rng = np.random.default_rng(21)
    N, H, W, C = 3, 7, 9, 24
    x = rng.standard_normal((N, H, W, C)); s = rng.standard_normal((N, 1, 1, C)) * 2
    sc = rng.uniform(0.5, 1.5, C); sh = rng.standard_normal(C) * 0.3
    a = O.act_fwd(x * sc + sh, O.ACT_HSWISH)
    sv = O.act_fwd(s, O.ACT_HSIGMOID)
    y = ops.scale_bcast_fwd(T(x), T(s), T(sc), T(sh), ops.ACT_HSWISH, ops.ACT_HSIGMOID)
    close(y, a * sv, what='se multiply')
    gy = rng.standard_normal((N, H, W, C))
    gx, gs = ops.scale_bcast_bwd(T(gy), T(x), T(s), T(sc), T(sh), ops.ACT_HSWISH, ops.ACT_HSIGMOID)
    close(gx, gy * sv, what='se bwd x')
    close(gs, (gy * a).sum((1, 2), keepdims=True), rtol=3e-4, what='se bwd s')
    base = rng.standard_normal(s.shape)
    out = ops.act_bwd(T(gs.cpu().numpy()), T(s), ops.ACT_HSIGMOID, T(base), accumulate=True)
    close(out, base + O.act_bwd(s, gs.cpu().numpy().astype(np.float64), O.ACT_HSIGMOID), rtol=3e-4, what='bare act bwd')


@pytest.mark.parametrize('case', [(2, 17, 17, 8, 3, 1, 'same'), (1, 16, 20, 16, 1, 2, (0, 0, 0, 0)), (2, 9, 9, 4, 3, 2, (1, 1, 1, 1))])
def test_im2col_col2im(ops, case):
    N, H, W, Cin, k, s, pad = case
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, 12)) * 0.2
    col = ops.im2col(T(x), k, s, 1, pad)
    y = col.reshape(-1, col.shape[-1]).double().cpu().numpy()[:, :k * k * Cin] @ w.reshape(-1, 12)
    y_ref = O.conv2d_fwd(x, w, s, 1, pad)
    close(y.reshape(y_ref.shape), y_ref, what='im2col @ w == conv')
    gy = rng.standard_normal(y_ref.shape)
    gx_ref, _, _ = O.conv2d_bwd(x, w, gy, s, 1, pad)
    kp = col.shape[-1]
    gcol = np.zeros((gy.reshape(-1, 12).shape[0], kp))
    gcol[:, :k * k * Cin] = gy.reshape(-1, 12) @ w.reshape(-1, 12).T
    gx = ops.col2im(T(gcol.reshape(col.shape)), (N, H, W, Cin), k, s, 1, pad)
    close(gx, gx_ref, rtol=3e-4, what='col2im')


def test_sgd(ops):
    rng = np.random.default_rng(2)
    n = 1003
    w, v, g = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    wt, vt = T(np.pad(w, (0, 1))), T(np.pad(v, (0, 1)))
    lr = torch.tensor([0.01], device=DEV)
    ops.sgd_momentum(wt[:n], vt[:n], T(np.pad(g, (0, 1)))[:n], lr, 0.9, 2e-5)
    w2, v2 = O.sgd_momentum_step(w, v, g, 0.01, 0.9, 2e-5)
    close(wt[:n], w2, what='sgd w'); close(vt[:n], v2, what='sgd v')


def test_bad_arguments_fail_loudly(ops):
    x = torch.zeros((1, 4, 4, 6), device=DEV)   # C % 4 != 0
    with pytest.raises(ops.Dl3pError):
        ops.dwconv2d_fwd(x, torch.zeros((3, 3, 6), device=DEV))
    with pytest.raises(ops.Dl3pError):
        ops.pwconv_fwd(torch.zeros((4, 8)), torch.zeros((8, 8)))   # CPU tensors are refused


@pytest.mark.parametrize('case', [(2, 9, 9, 21, 33, 33), (1, 17, 33, 19, 65, 129), (3, 5, 7, 4, 20, 28), (1, 33, 33, 32, 129, 129),
                                  (2, 9, 9, 2, 33, 33), (2, 9, 9, 27, 33, 33), (1, 9, 9, 10, 40, 33)])
def test_argmax_confusion(ops, case):
    """evaluation head: argmax of the upsampled logits + confusion matrix == oracle resize -> np.argmax ->
    generate_matrix (eval.py:33-36, 368-373); counters accumulate across calls; the mask equals the argmax of predict's
    probabilities"""
    N, h, w, C, H, W = case
    rng = np.random.default_rng(C + H)
    cp = 20 if C == 4 else ((C + 3) // 4) * 4               # C = 4 also checks a row stride wider than the classes
    z = np.zeros((N, h, w, cp))
    z[..., :C] = rng.standard_normal((N, h, w, C)) * 3
    z[0, 0, 0, :C] = 1.5                                       # an exact tie: the lowest class index wins
    lab = rng.integers(0, C, (N, H * W)).astype(np.float64)
    lab[rng.uniform(size=lab.shape) < 0.07] = 255
    zt, lt = T(z), T(lab)
    pred, cm = ops.argmax_confusion(zt, C, H, W, labels=lt, want_mask=True)
    big = O.resize_bilinear_fwd(z[..., :C], H, W)
    ref = big.argmax(-1)
    got = pred.cpu().numpy()
    # fp32 interpolation vs fp64: allow disagreement only where the two best logits are within rounding distance
    srt = np.sort(big, -1)
    close_call = (srt[..., -1] - srt[..., -2]) < 1e-4
    assert ((got == ref) | close_call).all() and (got != ref).mean() < 1e-3
    assert got[0, 0, 0] == 0
    probs = ops.upsample_softmax_ce(zt, C, H, W, want_probs=True)['probs'].cpu().numpy()
    assert np.array_equal(got, probs.argmax(-1)) or ((got != probs.argmax(-1)) <= close_call).all()
    want = O.confusion_matrix(lab.reshape(N, H, W), got, C)     # the matrix itself is exact integer counting
    assert np.array_equal(cm.cpu().numpy(), want)
    _, cm2 = ops.argmax_confusion(zt, C, H, W, labels=lt, confusion=cm)
    assert np.array_equal(cm2.cpu().numpy(), 2 * want)
    only_mask, none = ops.argmax_confusion(zt, C, H, W, want_mask=True)
    assert none is None and np.array_equal(only_mask.cpu().numpy(), got)


@pytest.mark.parametrize('case', [(2, 33, 33, 64, 3, 2, (1, 1, 1, 1)), (1, 16, 22, 8, 3, 2, (1, 1, 1, 1)), (2, 9, 9, 12, 2, 2, (0, 0, 0, 0)),
                                  (1, 12, 10, 4, 3, 1, (1, 1, 1, 1))])
def test_maxpool2d(ops, case):
    """ZeroPadding2D + MaxPooling2D (deeplabv3p_resnet50.py:266-267) with the BN + ReLU prologue, forward and backward"""
    N, H, W, C, k, s, pad = case
    rng = np.random.default_rng(H + C)
    x = rng.standard_normal((N, H, W, C))
    sc = rng.uniform(0.5, 1.5, C); sh = rng.standard_normal(C) * 0.3
    a = O.act_fwd(x * sc + sh, O.ACT_RELU)                 # many exact zeros: ties with the zero padding
    y_ref, arg = O.maxpool2d_fwd(a, k, s, pad)
    y = ops.maxpool2d_fwd(T(x), k, s, pad, T(sc), T(sh), ops.ACT_RELU)
    close(y, y_ref, what='maxpool fwd')
    gy = rng.standard_normal(y_ref.shape)
    g_ref = O.maxpool2d_bwd(gy, arg, x.shape, k, s, pad)
    gx = ops.maxpool2d_bwd(T(x), T(gy), k, s, pad, T(sc), T(sh), ops.ACT_RELU)
    # gradient routed to a zero (ReLU-dead or padding) position is arbitrary among the tied zeros in Keras too and is
    # killed by the ReLU backward: compare where the activation is alive
    alive = a > 0
    assert np.abs(np.where(alive, gx.cpu().numpy() - g_ref, 0)).max() < 1e-6
    base = rng.standard_normal(x.shape)
    gx2 = ops.maxpool2d_bwd(T(x), T(gy), k, s, pad, T(sc), T(sh), ops.ACT_RELU, out=T(base), accumulate=True)
    assert np.abs((gx2 - gx).cpu().numpy() - base).max() < 1e-5
    # recorded winners: the forward stores each window's first maximum, the backward reads it back -> same gradient
    Ho, Wo = y_ref.shape[1:3]
    rec = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=DEV)
    y2 = ops.maxpool2d_fwd(T(x), k, s, pad, T(sc), T(sh), ops.ACT_RELU, argmax=rec)
    assert torch.equal(y2, y)
    gx3 = ops.maxpool2d_bwd(T(x), T(gy), k, s, pad, T(sc), T(sh), ops.ACT_RELU, argmax=rec)
    assert torch.equal(gx3, gx)
    raw, arg_raw = O.maxpool2d_fwd(x, k, s, pad)           # no prologue: negative inputs lose against the padded zeros
    close(ops.maxpool2d_fwd(T(x), k, s, pad), raw, what='maxpool fwd, bare')
    g_raw = O.maxpool2d_bwd(gy, arg_raw, x.shape, k, s, pad)
    close(ops.maxpool2d_bwd(T(x), T(gy), k, s, pad), g_raw, what='maxpool bwd, bare (no ties)')


def test_adam_rmsprop_kernels(ops):
    """dl3p_adam_step / dl3p_rmsprop_step over 6 steps == the oracle's Keras 2.11 rules (bias correction from the device
    step counter, folded l2 term, frozen elements untouched)"""
    rng = np.random.default_rng(2)
    n = 10007
    L = ops.lib()
    w0 = rng.standard_normal(n); l2 = np.where(rng.uniform(size=n) < 0.5, 2e-5, 0.0); frozen = rng.uniform(size=n) < 0.1
    lre = np.where(frozen, 0.0, 1.0)
    st = ops._stream()
    for kind in ('adam', 'rmsprop'):
        w = T(w0); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
        wr, mr, vr = w0.copy(), np.zeros(n), np.zeros(n)
        lr = torch.tensor([1e-3], dtype=torch.float32, device=DEV)
        step = torch.zeros(1, dtype=torch.int64, device=DEV)
        l2t, lret = T(l2), T(lre)
        for t in range(1, 7):
            g = rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 0, n)
            gt = T(g)
            step += 1
            if kind == 'adam':
                L.adam_step(w.data_ptr(), m.data_ptr(), v.data_ptr(), gt.data_ptr(), n, lr.data_ptr(), step.data_ptr(), 0.9, 0.999,
                            1e-7, 0.5, l2t.data_ptr(), lret.data_ptr(), st)
                wn, mn, vn = O.adam_step(wr, mr, vr, g * 0.5, t, 1e-3, l2=l2)
                mr = np.where(frozen, mr, mn)
            else:
                L.rmsprop_step(w.data_ptr(), v.data_ptr(), gt.data_ptr(), n, lr.data_ptr(), 0.9, 1e-7, 0.5, l2t.data_ptr(),
                               lret.data_ptr(), st)
                wn, vn = O.rmsprop_step(wr, vr, g * 0.5, 1e-3, l2=l2)
            wr, vr = np.where(frozen, wr, wn), np.where(frozen, vr, vn)
        assert np.array_equal(w.cpu().numpy()[frozen], w0.astype(np.float32)[frozen])
        assert np.abs(w.cpu().numpy() - wr).max() < 2e-6, kind
        close(v, vr, rtol=1e-5, atol=1e-12, what=kind + ' second moment')



def test_label_prepare_matches_sklearn_golden(ops):
    """dl3p_label_prepare == the reference generator's label tail (deeplabv3p/data.py:116-145), bit-exact against
    vectors made with the real sklearn (tests/golden/make_label_weights.py) and against the oracle on a full-size,
    odd-sized batch (513*513 pixels per image: image bases are only byte aligned)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'label_weights.npz'))
    C, ign = int(g['num_classes']), int(g['ignore_index'])
    for i in range(int(g['n'])):
        u8 = torch.as_tensor(g['u8_%d' % i]).cuda().view(1, -1)
        lab, w = ops.label_prepare(u8, C, ign, adaptive=True)
        assert np.array_equal(lab.cpu().numpy().ravel(), g['labels_%d' % i])
        assert np.array_equal(w.cpu().numpy().ravel(), g['weights_%d' % i])
        lab2, none = ops.label_prepare(u8, C, ign)
        assert none is None and torch.equal(lab2, lab)
    rng = np.random.default_rng(3)
    N, P = 5, 513 * 513
    u8 = rng.integers(0, 30, (N, P)).astype(np.uint8)
    u8[rng.uniform(size=u8.shape) < 0.03] = 255
    u8[1] = 4                                                # a single-class image: weight 1 everywhere
    u8[2, :1000] = 9; u8[2, 1000:] = 255                     # mostly ignored
    lab, w = ops.label_prepare(torch.as_tensor(u8).cuda(), C, ign, adaptive=True)
    lab, w = lab.cpu().numpy(), w.cpu().numpy()
    for n in range(N):
        lr, wr = O.prepare_labels(u8[n], C, ign, adaptive=True)
        assert np.array_equal(lab[n], lr) and np.array_equal(w[n], wr), n
    assert np.all(w[1] == 1.0)
    src, dst = torch.zeros(4, dtype=torch.uint8, device='cuda'), torch.empty(4, device='cuda')
    with pytest.raises(RuntimeError):
        ops.lib().label_prepare(src.data_ptr(), dst.data_ptr(), dst.data_ptr(), None, 1, 4, C, ign, None)   # no workspace
