"""The costliest launches of BASELINE configs[2] (Xception 513x513, batch 4 per GPU) and configs[4] (MobileNetV3-Large 1024x2048,
bf16, batch 1) as stand-alone op calls at their exact launch shapes -- what scripts/pmc_ops.sh runs under rocprofv3 --pmc to get
FETCH_SIZE / WRITE_SIZE / MFMA-busy counters for kernels outside the headline graph (whole-step PMC runs of those graphs abort
inside the profiler: "AQL packet is malformed").   python3 scripts/pmc_ops.py xception | bf16"""
import importlib
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
dev = 'cuda'
REPS = 5


def xception():
    """BASELINE configs[2]: the 4356-row GEMMs through the entry points the STEP takes them on: the split-bf16
    kernels wherever the library's verdict table / rule says so (middle flow 728 -> 728, exit flow 1536 -> 2048, the split weight
    gradients), the fp32-input kernels elsewhere (aspp0 2048 -> 256 and concat_projection 1280 -> 256 forward)"""
    for (M, K, N) in [(4356, 1536, 2048), (4356, 728, 728), (4356, 2048, 256), (4356, 1280, 256), (66564, 304, 256)]:
        x = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) / K ** 0.5
        w = wt.t().contiguous()
        wsp, w_sp = ops.split_bf16x3(wt), ops.split_bf16x3(w)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        part, partk = ops.new_partials(N, dev), ops.new_partials(K, dev)
        dy, z = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        mean, invstd = torch.zeros(K, device=dev), torch.ones(K, device=dev)
        # the executor's own question (executor._use_sb): measured verdict first, then the rule K >= 128, N >= 128, 16384 rows
        def sb(role, kred, nout):
            pays = L.pwconv_sb_pays(role, M, kred, nout)
            if pays < 0:
                pays = int(kred >= 128 and nout >= 128 and M >= (60000 if role == 3 else 16384))
            return bool(pays) and bool(L.pwconv_sb_supported(role, M, kred, nout))
        for _ in range(REPS):
            if sb(1, K, N):
                ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU, partials=part)
            else:
                ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU, partials=part)
            if sb(3, N, K):
                ops.pwconv_bwd_data_sb(dy, w_sp, N, z=z, scale=sc, shift=sh, act=ops.ACT_RELU, mean=mean, invstd=invstd, partials=partk)
            else:
                ops.pwconv_bwd_data_bn(dy, w, z, sc, sh, ops.ACT_RELU, mean, invstd, partk)
            ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU)        # (routes itself onto pw_wgrad_sb_kernel: wgrad_sb_route)
    for r in (6, 12, 18):
        x = torch.randn(4, 33, 33, 2048, device=dev)
        w = torch.randn(3, 3, 2048, device=dev) * 0.3
        sc, sh = torch.rand(2048, device=dev) + 0.5, torch.randn(2048, device=dev) * 0.3
        part = ops.new_partials(2048, dev)
        gy = torch.randn(4, 33, 33, 2048, device=dev)
        for _ in range(REPS):
            ops.dwconv2d_fwd(x, w, 1, r, 'same', sc, sh, ops.ACT_RELU, partials=part)
            ops.dwconv2d_bwd_data(gy, w, (4, 33, 33, 2048), 1, r, 'same')
            ops.dwconv2d_bwd_weight(x, gy, 3, 1, r, 'same', sc, sh, ops.ACT_RELU)
    torch.cuda.synchronize()


def xception769():
    """BASELINE configs[3]: Xception, 769 x 769, output stride 8, batch 2 per GPU -- the split-bf16 GEMMs at 2 x 97^2 = 18818 and
    2 x 193^2 = 74498 rows through the production entry points (sb where the executor takes it), the ASPP rates 12 / 24 / 36"""
    for (M, K, N) in [(18818, 728, 728), (18818, 1536, 2048), (18818, 2048, 256), (74498, 304, 256), (74498, 256, 256)]:
        x = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) / K ** 0.5
        w = wt.t().contiguous()
        wsp, w_sp = ops.split_bf16x3(wt), ops.split_bf16x3(w)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        part, partk = ops.new_partials(N, dev), ops.new_partials(K, dev)
        dy, z = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        mean, invstd = torch.zeros(K, device=dev), torch.ones(K, device=dev)
        for _ in range(REPS):
            ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU, partials=part)
            ops.pwconv_bwd_data_sb(dy, w_sp, N, z=z, scale=sc, shift=sh, act=ops.ACT_RELU, mean=mean, invstd=invstd, partials=partk)
            ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU)
    for r in (12, 24, 36):
        x = torch.randn(2, 97, 97, 2048, device=dev)
        w = torch.randn(3, 3, 2048, device=dev) * 0.3
        sc, sh = torch.rand(2048, device=dev) + 0.5, torch.randn(2048, device=dev) * 0.3
        part = ops.new_partials(2048, dev)
        gy = torch.randn(2, 97, 97, 2048, device=dev)
        for _ in range(REPS):
            ops.dwconv2d_fwd(x, w, 1, r, 'same', sc, sh, ops.ACT_RELU, partials=part)
            ops.dwconv2d_bwd_data(gy, w, (2, 97, 97, 2048), 1, r, 'same')
            ops.dwconv2d_bwd_weight(x, gy, 3, 1, r, 'same', sc, sh, ops.ACT_RELU)
    torch.cuda.synchronize()


def headline_gemms():
    """the decoder GEMMs of BASELINE configs[1] (266256 rows) on the kernels the step launches them on: wide tiled forward,
    row-stationary data gradient with the folded BatchNorm-backward apply and the fused sums, split weight gradient"""
    M = 266256
    for (K, N) in [(304, 256), (256, 256)]:
        x = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) / K ** 0.5
        w = wt.t().contiguous()
        wsp, w_sp = ops.split_bf16x3(wt), ops.split_bf16x3(w)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        bsc, bsh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.3
        mu, istd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
        coef = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1]).contiguous()
        part, partk = ops.new_partials(N, dev), ops.new_partials(K, dev)
        g, zo, z = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        dz = torch.empty(M, N, device=dev)
        mean, invstd = torch.zeros(K, device=dev), torch.ones(K, device=dev)
        for _ in range(REPS):
            ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU, partials=part)
            ops.pwconv_bwd_data_sb_apply(g, zo, bsc, bsh, ops.ACT_RELU, mu, istd, coef, w_sp, N, dz=dz, z=z, scale=sc, shift=sh, act=ops.ACT_RELU,
                                         mean=mean, invstd=invstd, partials=partk)
            ops.pwconv_bwd_data_sb(dz, w_sp, N, z=z, scale=sc, shift=sh, act=ops.ACT_RELU, mean=mean, invstd=invstd, partials=partk)
            ops.pwconv_bwd_weight(x, dz, sc, sh, ops.ACT_RELU)
    torch.cuda.synchronize()


def bf16():
    bf = torch.bfloat16
    for (M, K, N) in [(131072, 304, 256), (131072, 256, 256), (8192, 1280, 256), (524288, 16, 64)]:
        x = torch.randn(M, K, device=dev).to(bf)
        w = torch.randn(K, N, device=dev) / K ** 0.5
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        part = ops.new_partials(N, dev)
        dy = torch.randn(M, N, device=dev).to(bf)
        for _ in range(REPS):
            ops.pwconv_fwd_bf16(x, w, None, sc, sh, ops.ACT_HSWISH, partials=part)
            ops.pwconv_bwd_data_bf16(dy, w)
            ops.pwconv_bwd_weight_bf16(x, dy, sc, sh, ops.ACT_HSWISH)
    for (N, H, W, C, k, s, r) in [(1, 256, 512, 304, 3, 1, 1), (1, 64, 128, 960, 5, 1, 2), (1, 64, 128, 160, 3, 1, 18),
                                  (1, 64, 128, 160, 3, 1, 6)]:
        x = torch.randn(N, H, W, C, device=dev).to(bf)
        w = torch.randn(k, k, C, device=dev) * 0.3
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
        part = ops.new_partials(C, dev)
        gy = torch.randn(N, H, W, C, device=dev).to(bf)
        for _ in range(REPS):
            ops.dwconv2d_fwd_bf16(x, w, s, r, 'same', sc, sh, ops.ACT_RELU6, partials=part)
            ops.dwconv2d_bwd_data_bf16(gy, w, (N, H, W, C), s, r, 'same')
            ops.dwconv2d_bwd_weight_bf16(x, gy, k, s, r, 'same', sc, sh, ops.ACT_RELU6)
    torch.cuda.synchronize()


if __name__ == '__main__':
    {'xception': xception, 'xception769': xception769, 'headline_gemms': headline_gemms, 'bf16': bf16}[sys.argv[1]]()
