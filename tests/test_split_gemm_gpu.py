"""The fp32-accurate GEMMs on the bf16 matrix pipe (csrc/pw_split.hip) at the UNCHANGED tolerances of the
fp32 kernels: forward 2e-5 of the output's scale (tests/test_production_shapes_gpu.py PW_PROD), statistics 1e-4, data
gradient 2e-5, BatchNorm-backward sums 2e-4 -- against float64.  Plus the split itself: exact (a1 + a2 + a3 == a bit for bit),
and the comparison the opt-in is judged by: its error against float64 next to the fp32 MFMA kernel's on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bf16_planes_to_f32(sp):
    """(3, rows, pitch) int16 -> three float32 arrays"""
    u = sp.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    return u.view(np.float32)


def test_split_is_exact(ops):
    g = torch.Generator(device=DEV); g.manual_seed(3)
    a = torch.randn(37, 70, device=DEV, generator=g) * torch.exp(torch.randn(37, 70, device=DEV, generator=g) * 4)
    a[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-39, 6.0, 1e30, -7.5e-20], device=DEV)      # zero, a denormal, large / tiny
    sp = ops.split_bf16x3(a)
    assert sp.shape == (3, 37, 96)
    h, m, l = _bf16_planes_to_f32(sp)
    an = a.cpu().numpy()
    s = (h[:, :70].astype(np.float64) + m[:, :70].astype(np.float64)) + l[:, :70].astype(np.float64)
    # exact: the three pieces carry the 24 significant bits of a float32 (denormals lose what bf16 cannot hold: flushed)
    normal = np.abs(an) > 1e-30
    assert np.array_equal(s[normal], an.astype(np.float64)[normal])
    assert np.all(h[:, 70:] == 0) and np.all(m[:, 70:] == 0) and np.all(l[:, 70:] == 0)
    # magnitudes fall by 2^-8 per level (round to nearest)
    assert np.all(np.abs(m[:, :70][normal]) <= np.abs(h[:, :70][normal]) * 2.0 ** -8 * 1.0001)
    assert np.all(np.abs(l[:, :70][normal]) <= np.abs(h[:, :70][normal]) * 2.0 ** -16 * 1.0001)


SHAPES = [(16 * 129 * 129, 304, 256), (16 * 129 * 129, 256, 256), (16 * 33 * 33, 1280, 256), (16 * 33 * 33, 960, 160),
          (16 * 33 * 33, 160, 960), (4 * 33 * 33, 728, 728), (4 * 33 * 33, 2048, 256), (4 * 33 * 33, 1536, 2048),
          (16 * 65 * 65, 192, 64), (16 * 129 * 129, 256, 24),
          # ragged: M not a multiple of the tile, K not of 32, N not of 16
          (4357, 100, 200), (1089, 36, 24), (9001, 304, 252), (2600, 728, 132), (130, 20, 12)]


@pytest.mark.parametrize('pipe', [1, 0], ids=['producer_consumer', 'symmetric'])
@pytest.mark.parametrize('case', SHAPES)
def test_split_gemm_matches_float64_at_the_fp32_tolerances(ops, case, pipe):
    M, K, N = case
    L = ops.lib()
    L.set_option(b'pw_small_min_rows', -1)
    L.set_option(b'sb_pipe', pipe)
    try:
        g = torch.Generator(device=DEV); g.manual_seed(M % 9973 + 7 * K + 13 * N)
        rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
        x = rnd(M, K)
        wt = rnd(N, K) / K ** 0.5
        sc = torch.rand(K, device=DEV, generator=g) + 0.5
        sh = rnd(K) * 0.3
        bias = rnd(N) * 0.1
        a64 = (x.double() * sc.double() + sh.double()).clamp(0.0, 6.0)
        y64 = a64 @ wt.double().t()
        wsp = ops.split_bf16x3(wt)
        part = ops.new_partials(N, DEV)
        y, rows = ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU6, partials=part)
        scale = float(y64.abs().max())
        e_sb = float((y.double() - y64).abs().max()) / scale
        assert e_sb < 2e-5, ('forward', e_sb)
        p = part[:rows * 2 * N].reshape(rows, 2, N).double().sum(0)
        s1, s2 = y64.sum(0), (y64 * y64).sum(0)
        assert float((p[0] - s1).abs().max()) < 1e-4 * max(float(s1.abs().max()), float(M) ** 0.5), 'stat sum'
        assert float((p[1] - s2).abs().max()) < 1e-4 * float(s2.abs().max()), 'stat sum of squares'
        # the fp32 MFMA kernel on the same inputs: the split kernel's error is of the same size (recorded, bounded at 3x)
        y32 = ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6)
        e_32 = float((y32.double() - y64).abs().max()) / scale
        assert e_sb < 3 * e_32 + 1e-7, (e_sb, e_32)
        # bias, no statistics, no prologue
        y2 = ops.pwconv_fwd_sb(x, wsp, K, bias)
        assert float((y2.double() - (x.double() @ wt.double().t() + bias.double())).abs().max()) < 2e-5 * float((x.double() @ wt.double().t()).abs().max())
        del y, y2, y32, a64, y64
        # data gradient: dy (M, N) . w (K, N)^T
        dy = rnd(M, N)
        w = wt.t().contiguous()          # (K, N)
        w_sp = ops.split_bf16x3(w)
        gx64 = dy.double() @ w.double().t()
        gx = ops.pwconv_bwd_data_sb(dy, w_sp, N)
        assert float((gx.double() - gx64).abs().max()) < 2e-5 * float(gx64.abs().max()), 'data gradient'
        base = rnd(M, K)
        gx_acc = ops.pwconv_bwd_data_sb(dy, w_sp, N, out=base.clone(), accumulate=True)
        assert float((gx_acc.double() - (gx64 + base.double())).abs().max()) < 2e-5 * float(gx64.abs().max()), 'accumulate'
        z = rnd(M, K)
        mean = z.mean(0)
        invstd = 1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-3)
        part = ops.new_partials(K, DEV)
        gx2, rows = ops.pwconv_bwd_data_sb(dy, w_sp, N, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean, invstd=invstd,
                                           partials=part)
        assert float((gx2.double() - gx64).abs().max()) < 2e-5 * float(gx64.abs().max()), 'data gradient (+BN sums)'
        u = z.double() * sc.double() + sh.double()
        d = gx64 * ((u > 0) & (u < 6))
        xh = (z.double() - mean.double()) * invstd.double()
        p = part[:rows * 2 * K].reshape(rows, 2, K).double().sum(0)
        assert float((p[0] - d.sum(0)).abs().max()) < 2e-4 * float(d.abs().sum(0).max()), 'BN backward sum'
        assert float((p[1] - (d * xh).sum(0)).abs().max()) < 2e-4 * float((d * xh).abs().sum(0).max()), 'BN backward sum * xhat'
    finally:
        L.set_option(b'pw_small_min_rows', 64)
        L.set_option(b'sb_pipe', 0)


@pytest.mark.parametrize('pipe', [1, 0], ids=['producer_consumer', 'symmetric'])
@pytest.mark.parametrize('mi', [1, 2])
@pytest.mark.parametrize('nt', [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_split_gemm_tile_choice(ops, nt, mi, pipe):
    L = ops.lib()
    L.set_option(b'pw_small_min_rows', 1 << 30)
    L.set_option(b'gemm_nt', nt)
    L.set_option(b'gemm_mi', mi)
    L.set_option(b'sb_pipe', pipe)
    try:
        for case in [(4357, 100, 200), (2600, 728, 132)]:
            _run(ops, case)
    finally:
        L.set_option(b'gemm_nt', 0)
        L.set_option(b'gemm_mi', 0)
        L.set_option(b'sb_pipe', 0)
        L.set_option(b'pw_small_min_rows', 64)


def _run(ops, case):
    L = ops.lib()
    M, K, N = case
    g = torch.Generator(device=DEV); g.manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV, generator=g)
    wt = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    part = ops.new_partials(N, DEV)
    y, rows = ops.pwconv_fwd_sb(x, ops.split_bf16x3(wt), K, partials=part)
    y64 = x.double() @ wt.double().t()
    assert float((y.double() - y64).abs().max()) < 2e-5 * float(y64.abs().max())
    p = part[:rows * 2 * N].reshape(rows, 2, N).double().sum(0)
    assert float((p[1] - (y64 * y64).sum(0)).abs().max()) < 1e-4 * float((y64 * y64).sum(0).max())
    dy = torch.randn(M, N, device=DEV, generator=g)
    w = wt.t().contiguous()
    z = torch.randn(M, K, device=DEV, generator=g)
    one, zero = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    part = ops.new_partials(K, DEV)
    gx, rows = ops.pwconv_bwd_data_sb(dy, ops.split_bf16x3(w), N, z=z, scale=one, shift=zero, act=ops.ACT_RELU, mean=zero, invstd=one,
                                      partials=part)
    gx64 = dy.double() @ w.double().t()
    assert float((gx.double() - gx64).abs().max()) < 2e-5 * float(gx64.abs().max())
    d = gx64 * (z.double() > 0)
    p = part[:rows * 2 * K].reshape(rows, 2, K).double().sum(0)
    assert float((p[0] - d.sum(0)).abs().max()) < 2e-4 * float(d.abs().sum(0).max())


@pytest.mark.parametrize('nt,mi,wm', [(16, 2, 1), (16, 1, 1), (16, 1, 2), (12, 2, 1), (12, 2, 2), (8, 2, 2)])
def test_every_wide_split_gemm_tile(ops, nt, mi, wm):
    """the one-workgroup-per-CU family (128 / 256 rows x up to 256 columns, 256 or 512 threads)"""
    L = ops.lib()
    L.set_option(b'pw_small_min_rows', 1 << 30)
    L.set_option(b'sb_wm', wm)
    L.set_option(b'sb_nt', nt)
    L.set_option(b'gemm_mi', mi)
    try:
        for case in [(4357, 100, 200), (2600, 728, 252), (70001, 304, 256), (1301, 260, 132)]:
            _run(ops, case)
    finally:
        L.set_option(b'sb_wm', 0)
        L.set_option(b'sb_nt', 0)
        L.set_option(b'gemm_mi', 0)
        L.set_option(b'pw_small_min_rows', 64)
