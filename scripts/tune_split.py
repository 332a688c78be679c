"""Measure the split-bf16 GEMM (csrc/pw_split.hip) against the fp32-input MFMA kernel's production pick for every compute-bound
pointwise-conv launch of the BASELINE graphs, per role (forward, forward + BN statistics, data gradient, data gradient + fused BN
sums), and write tf-keras-deeplabv3p-model-set_amd/csrc/sb_tuned.h:

  g_sb_tuned  the split kernel's tile per shape where the best candidate beats gemm_plan_sb's heuristic by more than 3 %
              (2-workgroups-per-CU tiles nt x mi [x persistent workgroups per CU], or the wide one-workgroup-per-CU family);
  g_sb_pays   per shape, whether the best split launch beats the fp32 kernel (tables on) by more than 3 % -- listed only where
              that differs from the executor's threshold rule (K >= 128, N >= 128, >= 16384 rows; >= 60000 with the fused sums).

A tile only enters the table if its output equals the heuristic pick's to rounding (same six products, another grouping);
tests/test_tuned_tables_gpu.py then holds every row against float64.  GPU box, repo root:  python3 scripts/tune_split.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tune_gemm as T          # noqa: E402  (shapes of the BASELINE graphs, the library's event-pair timer)

ops, L, ROOT = T.ops, T.L, T.ROOT
dev = 'cuda'
WIDE = [(16, 1, 2), (16, 2, 1), (12, 2, 1), (8, 2, 2), (16, 1, 1)]


def rule(role, M, K, N):
    """the executor's threshold rule (Executor._use_sb; role 4: wgrad_sb_route in pwconv.hip)"""
    return K >= 128 and N >= 128 and M >= 16384 and (role != 3 or M >= 60000)


def pin(c):
    for k in (b'gemm_nt', b'gemm_mi', b'gemm_per_cu', b'sb_wm', b'sb_nt'):
        L.set_option(k, 0)
    L.set_option(b'sb_rs', 0)          # (the heuristic pick and the tile candidates are the tiled kernels')
    if c is None:
        return
    nt, mi, pc = c
    if pc == 103:                      # the row-stationary form (csrc/pw_split_rs.hip): {2, 1, 103} = wm 3
        L.set_option(b'sb_rs', 1)
    elif pc > 100:
        L.set_option(b'sb_wm', pc - 100); L.set_option(b'sb_nt', nt); L.set_option(b'gemm_mi', mi)
    else:
        L.set_option(b'sb_wm', -1); L.set_option(b'gemm_nt', nt); L.set_option(b'gemm_mi', mi); L.set_option(b'gemm_per_cu', pc)


def make(role, M, K, N, seed=None):
    """-> (run_fp32, run_split, outputs) for one launch; (M, K, N) as launched: K the reduction, N the output columns"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed if seed is not None else role + M + K + N)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    NB = 3
    if role in (0, 1):
        xs = [rnd(M, K) for _ in range(NB)]
        wt = rnd(N, K) / K ** 0.5
        wsp = ops.split_bf16x3(wt)
        sc, sh = torch.rand(K, device=dev, generator=g) + 0.5, rnd(K) * 0.3
        ys = [torch.empty(M, N, device=dev) for _ in range(NB)]
        part = ops.new_partials(N, dev) if role == 1 else None
        i = [0]

        def f32():
            i[0] = (i[0] + 1) % NB
            return ops.pwconv_fwd_wt(xs[i[0]], wt, None, sc, sh, ops.ACT_RELU6, out=ys[i[0]], partials=part)

        def sb():
            i[0] = (i[0] + 1) % NB
            return ops.pwconv_fwd_sb(xs[i[0]], wsp, K, None, sc, sh, ops.ACT_RELU6, out=ys[i[0]], partials=part)

        def outs(fn):
            i[0] = 0
            o = fn()
            if role == 1:
                return [o[0].double(), part[:o[1] * 2 * N].reshape(o[1], 2, N).double().sum(0)]
            return [o.double()]
        return f32, sb, outs
    gs = [rnd(M, K) for _ in range(NB)]
    w = rnd(N, K) / K ** 0.5
    w_sp = ops.split_bf16x3(w)
    gxs = [torch.empty(M, N, device=dev) for _ in range(NB)]
    zs = [rnd(M, N) for _ in range(2)]
    sc, sh = torch.rand(N, device=dev, generator=g) + 0.5, rnd(N) * 0.3
    mean, invstd = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    part = ops.new_partials(N, dev)
    i = [0]

    def f32():
        i[0] = (i[0] + 1) % NB
        if role == 2:
            return ops.pwconv_bwd_data(gs[i[0]], w, out=gxs[i[0]])
        return ops.pwconv_bwd_data_bn(gs[i[0]], w, zs[i[0] % 2], sc, sh, ops.ACT_RELU6, mean, invstd, part, out=gxs[i[0]])

    def sb():
        i[0] = (i[0] + 1) % NB
        if role == 2:
            return ops.pwconv_bwd_data_sb(gs[i[0]], w_sp, K, out=gxs[i[0]])
        return ops.pwconv_bwd_data_sb(gs[i[0]], w_sp, K, out=gxs[i[0]], z=zs[i[0] % 2], scale=sc, shift=sh, act=ops.ACT_RELU6,
                                      mean=mean, invstd=invstd, partials=part)

    def outs(fn):
        i[0] = 0
        o = fn()
        if role == 3:
            return [o[0].double(), part[:o[1] * 2 * N].reshape(o[1], 2, N).double().sum(0)]
        return [o.double()]
    return f32, sb, outs


def measure(role, M, K, N):
    f32, sb, outs = make(role, M, K, N)
    res = {}
    L.set_option(b'gemm_tuned', 1)
    pin(None)
    t32 = T.timeit(f32)
    L.set_option(b'gemm_tuned', 0)
    # the baseline is what production launches without a table row: the planner with its row-stationary RULE live (sb_rs = -1: role >= 2,
    # M >= 131072, K > 224 -- pwconv.hip sb_rs_route).  A tiled tile that beats it by 3 % becomes an explicit table row, and a table row
    # with pc != 103 is how the table says "not row-stationary" for such a shape.
    L.set_option(b'sb_rs', -1)
    res[None] = T.timeit(sb)
    L.set_option(b'sb_rs', 0)
    ntiles = (N + 15) // 16
    for mi in (1, 2):
        for nt in range(3, 9):
            if nt > ntiles:
                continue
            if role == 3 and mi == 2 and nt > 4:       # spills
                continue
            pin((nt, mi, 0))
            res[(nt, mi, 0)] = T.timeit(sb)
    best = min((k for k in res if k is not None), key=lambda k: res[k])
    for pc in (2, 3, 4, 6):
        pin((best[0], best[1], pc))
        res[(best[0], best[1], pc)] = T.timeit(sb)
    if M >= 8192:
        for (nt, mi, wm) in WIDE:
            if N <= 16 * (nt - 4):
                continue
            pin((nt, mi, 100 + wm))
            res[(nt, mi, 100 + wm)] = T.timeit(sb)
    out6 = (ctypes.c_int * 6)()
    pin(None)                          # (a pinned tile keeps the planner off the row-stationary form)
    L.set_option(b'sb_rs', 1)
    L.gemm_plan_query(role + 5, M, K, N, out6)
    if out6[0] == 3 and out6[3] == 3 and M >= 65536:       # served by the row-stationary form
        pin((2, 1, 103))
        res[(2, 1, 103)] = T.timeit(sb)
    pin(None)
    best = min((k for k in res if k is not None), key=lambda k: res[k])
    # a faster tile only counts if it computes the same thing as the production pick
    bad = ''
    if res[best] < 0.97 * res[None]:
        L.set_option(b'sb_rs', -1)
        ref = outs(sb)
        pin(best)
        got = outs(sb)
        pin(None)
        for a, b in zip(got, ref):
            d = float((a - b).abs().max())
            if not torch.isfinite(a).all() or d > 1e-4 * max(1e-30, float(b.abs().max())):
                bad = 'max diff %.3g of %.3g' % (d, float(b.abs().max()))
    L.set_option(b'gemm_tuned', 1)
    L.set_option(b'sb_rs', -1)         # (never leave the process on the tiled-only planner)
    return t32, res, best, bad


def measure_wgrad(M, K, N):
    """weight gradient: the fp32-input MFMA kernel (tables on) against pw_wgrad_sb_kernel over tile x workgroups-per-CU; cost = kernel
    time + the slabs' share of the batched reduction (tune_gemm.py's rate) -> (fp32 us, {None | (tile, per_cu, 0): us}, best, refusal)"""
    xs = [torch.randn(M, K, device=dev) for _ in range(3)]
    gs = [torch.randn(M, N, device=dev) for _ in range(3)]
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    ws = torch.empty(96 << 20, device=dev)
    rows = ctypes.c_int(0)
    st = torch.cuda.current_stream().cuda_stream
    i = [0]

    def run():
        i[0] = (i[0] + 1) % 3
        L.pwconv_bwd_weight_slabs(xs[i[0]].data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, gs[i[0]].data_ptr(), N, ws.data_ptr(),
                                  ws.numel() * 4, ctypes.byref(rows), M, K, N, st)

    def cost():
        t = T.timeit(run)
        return t + rows.value * K * N * 4 / 3.3e6

    def out():
        i[0] = 0
        run()
        return ws[:rows.value * K * N].reshape(rows.value, K, N).double().sum(0)
    L.set_option(b'gemm_tuned', 1)
    L.set_option(b'split_wgrad', 0)
    t32 = cost()
    L.set_option(b'split_wgrad', 1)
    L.set_option(b'gemm_tuned', 0)
    res = {}
    L.set_option(b'split_wgrad_tile', -1); L.set_option(b'split_wgrad_per_cu', 0)
    # (a pinned tile / workgroups-per-CU bypasses the verdict tables: wgrad_sb_route in pwconv.hip)
    for tile in range(5):            # (4: the 128 x 256 tile, round 6 -- only wide layers: N >= 256)
        if tile == 4 and N < 256:
            continue
        for pc in (1, 2, 3, 4, 6):      # (1: half the slabs of the default -- the few-row layers with large K x N, Xception's middle flow)
            L.set_option(b'split_wgrad_tile', tile); L.set_option(b'split_wgrad_per_cu', pc)
            res[(tile, pc, 0)] = cost()
    # the heuristic's own pick: its tile at its workgroups per CU (pw_split.hip: dl3p_wgrad_sb_plan)
    tk = [128, 64, 128, 64]; tn = [128, 128, 64, 64]
    area = [((K + tk[i] - 1) // tk[i] * tk[i]) * ((N + tn[i] - 1) // tn[i] * tn[i]) * (1 + 0.5 * (64 / tk[i] + 64 / tn[i])) for i in range(4)]
    ht = min(range(4), key=lambda i: area[i])
    res[None] = res[(ht, 2 if ht == 0 else 4, 0)]
    best = min((k for k in res if k is not None), key=lambda k: res[k])
    bad = ''
    if res[best] < 0.97 * res[None]:
        L.set_option(b'split_wgrad_tile', ht); L.set_option(b'split_wgrad_per_cu', 2 if ht == 0 else 4)
        ref = out()
        L.set_option(b'split_wgrad_tile', best[0]); L.set_option(b'split_wgrad_per_cu', best[1])
        got = out()
        d = float((got - ref).abs().max())
        if not torch.isfinite(got).all() or d > 3e-5 * M ** 0.5:
            bad = 'max diff %.3g of %.3g' % (d, float(ref.abs().max()))
    L.set_option(b'split_wgrad_tile', -1); L.set_option(b'split_wgrad_per_cu', 0)
    L.set_option(b'gemm_tuned', 1)
    return t32, res, best, bad


def main():
    L.set_option(b'pw_small_min_rows', -1)
    roles = {0, 1, 2, 3, 4}
    min_rows = 4096
    for a in sys.argv[1:]:
        if a.startswith('--roles='):
            roles = {int(v) for v in a.split('=')[1].split(',')}
        if a.startswith('--min-rows='):       # (a partial re-tune, e.g. --roles=2,3 --min-rows=60000: merge gpurun_out/sb_tuned.h by hand)
            min_rows = int(a.split('=')[1])
    sh = T.shapes()
    tuned, pays, log = [], [], []
    for (role, M, K, N), mt in sorted(sh.items()):
        if role not in roles or role > 4 or K < 64 or N < 64 or M < min_rows or M * max(K, N) * 4 >= (1 << 32):
            continue
        plan = (ctypes.c_int * 6)()
        L.gemm_plan_query(role, M, K, N, plan)
        if plan[0] != 0 or (role < 4 and not L.pwconv_sb_supported(role, M, K, N)):
            continue
        try:
            t32, res, best, bad = measure(role, M, K, N) if role < 4 else measure_wgrad(M, K, N)
        except Exception as e:      # noqa: BLE001
            log.append('skip %s: %s' % ((role, M, K, N), str(e)[:80]))
            continue
        tb = res[best] if not bad else res[None]
        line = 'role %d M=%7d K=%4d N=%4d (%s): fp32 %7.1f us | split heuristic %7.1f, best (%d,%d,%d) %7.1f us%s | split/fp32 %.2f rule=%d' % (
            role, M, K, N, mt, t32, res[None], best[0], best[1], best[2], res[best], (' REFUSED ' + bad) if bad else '', tb / t32,
            rule(role, M, K, N))
        log.append(line)
        print(line, flush=True)
        use_best = not bad and res[best] < 0.97 * res[None]
        t_split = res[best] if use_best else res[None]
        verdict = 1 if t_split < 0.97 * t32 else 0
        if use_best and verdict:          # (a tile row for a launch that stays on the fp32 kernel would never be read)
            tuned.append((role + 5, M, K, N, best[0], best[1], best[2], res[None], res[best]))      # (weight gradient: role 9 {tile, per CU})
        if verdict != int(rule(role, M, K, N)):
            pays.append((role, M, K, N, verdict, t32, t_split))
    out = ['// GENERATED by scripts/tune_split.py on an MI355X -- the split-bf16 GEMM (pw_split.hip) per GEMM shape of the BASELINE graphs.',
           '// g_sb_tuned: {role + 5, M, K, N, nt, mi, pc} where the best measured tile beats the planner\'s own pick (row-stationary rule live) by more than 3 %',
           '//             (pc > 100: the wide family, wm = pc - 100 -- {2, 1, 103}: the row-stationary form, pw_split_rs.hip; else persistent workgroups per CU, 0 = by tile width)   // heuristic us -> tuned us',
           '// g_sb_pays:  {role, M, K, N, pays}: 1 where the best split launch beat the fp32-input MFMA kernel\'s production pick by more than',
           '//             3 %, 0 where it did not -- only rows where that differs from the executor\'s threshold rule (K, N >= 128, rows >= 16384)',
           '//             // fp32 us, split us',
           'static const GemmTuned g_sb_tuned[] = {', '    {-1, 0, 0, 0, 0, 0, 0},']
    for r in tuned:
        out.append('    {%d, %d, %d, %d, %d, %d, %d},   // %.1f -> %.1f' % r)
    out += ['};', 'static const SbPays g_sb_pays[] = {', '    {-1, 0, 0, 0, 0},']
    for r in pays:
        out.append('    {%d, %d, %d, %d, %d},   // %.1f, %.1f' % r)
    out.append('};')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    open(os.path.join(ROOT, 'gpurun_out', 'sb_tuned.h'), 'w').write('\n'.join(out) + '\n')
    open(os.path.join(ROOT, 'gpurun_out', 'sb_tune_log.txt'), 'w').write('\n'.join(log) + '\n')
    L.set_option(b'sb_rs', -1)
    print('%d tile rows, %d verdict rows -> gpurun_out/sb_tuned.h' % (len(tuned), len(pays)))


if __name__ == '__main__':
    main()
