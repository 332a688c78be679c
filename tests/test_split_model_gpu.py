"""Whole-model parity with the split-bf16 GEMMs: the same train-step / predict comparisons against the
float64 oracle as tests/test_model_gpu.py and tests/test_production_shapes_gpu.py, at their UNCHANGED tolerances (logits and loss
1e-3, parameter gradients 8e-3 / 1e-2 at 65 x 65 and 5e-3 at 513 x 513, weights after the step 1e-3).  That they hold is the
condition under which the split kernels became the default for the compute-bound 1 x 1 convs (executor.split_gemm_enabled).
At the default thresholds only layers with >= 16384 rows take them, so the small-model tests here lower the thresholds; the last
tests run the 513 x 513 steps with the switch OFF, so the fp32-input MFMA kernels keep their production-dispatch coverage."""
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def split_on(monkeypatch):
    monkeypatch.setenv('DL3P_SPLIT_GEMM', '1')
    monkeypatch.setenv('DL3P_SPLIT_MIN_K', '32')      # small test models: let every GEMM the tiled kernel serves take the split path
    monkeypatch.setenv('DL3P_SPLIT_MIN_N', '16')
    monkeypatch.setenv('DL3P_SPLIT_MIN_ROWS', '64')
    monkeypatch.setenv('DL3P_SPLIT_MIN_ROWS_BN', '64')
    # ... and every weight gradient the tiled kernel serves (>= 1024 rows) onto pw_wgrad_sb_kernel: a pinned plan bypasses its
    # verdict table / row threshold (pwconv.hip, wgrad_sb_route)
    L = load_pkg('_lib').lib()
    L.set_option(b'split_wgrad_per_cu', 2)
    # ... and the dense k x k convs wherever the split implicit-GEMM kernels serve them (>= 1024 rows: Xception's entry_flow_conv1_2,
    # ResNet50's stage-2 3x3 convs at these sizes), not only where the measured rule says they pay
    L.set_option(b'conv_sb', 2)
    yield
    L.set_option(b'split_wgrad_per_cu', 0)
    L.set_option(b'conv_sb', -1)


def _uses_split(m):
    ex = next(iter(m._exec.values()))
    return m._store.Sb is not None and any('pwconv_fwd_sb' in lab[0] for lab in ex.fwd.labels)


@pytest.mark.parametrize('model_type,H,W,freeze,OS', [('mobilenetv2', 65, 65, 0, 16), ('xception', 65, 65, 0, 16),
                                                      ('mobilenetv3large', 64, 96, 0, 16), ('resnet50', 65, 65, 0, 16),
                                                      ('xception', 97, 97, 0, 8)])
def test_train_step_matches_oracle_with_split_gemms(model_type, H, W, freeze, OS):
    import test_model_gpu as T
    T.test_train_step_matches_oracle(model_type, H, W, freeze, OS)


@pytest.mark.parametrize('model_type,H,W', [('mobilenetv2', 65, 65), ('xception', 65, 65)])
def test_predict_matches_oracle_with_split_gemms(model_type, H, W):
    import test_model_gpu as T
    T.test_predict_matches_oracle(model_type, H, W)


# (Xception at 513 x 513 on the split GEMMs is what test_production_shapes_gpu.py::test_train_step_513_production_dispatch[xception] runs:
# the default dispatch; here it is a once-per-release repeat)
@pytest.mark.parametrize('model_type', ['mobilenetv2', pytest.param('xception', marks=pytest.mark.release)])
def test_train_step_513_with_split_gemms(model_type, monkeypatch):
    """production shapes, production thresholds of the split dispatch (K >= 128, N >= 128, 16384 rows)"""
    for k in ('DL3P_SPLIT_MIN_K', 'DL3P_SPLIT_MIN_N', 'DL3P_SPLIT_MIN_ROWS', 'DL3P_SPLIT_MIN_ROWS_BN'):
        monkeypatch.delenv(k)
    import test_production_shapes_gpu as T
    L = load_pkg('_lib').lib()
    L.set_option(b'split_wgrad_per_cu', 0)        # the weight gradients by their own table / rule, as in production
    L.set_option(b'conv_sb', -1)                  # ... and the dense convs by theirs
    L.set_option(b'pw_small_min_rows', -1)
    try:
        T.test_train_step_513_production_dispatch(model_type)
    finally:
        L.set_option(b'pw_small_min_rows', 64)


def test_the_split_path_is_actually_taken():
    import numpy as np
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    m.use_graphs = False
    x = np.random.default_rng(0).uniform(-1, 1, (2, 65, 65, 3)).astype(np.float32)
    y = np.zeros((2, 65 * 65, 1), np.float32)
    m.train_on_batch(x, y)
    assert _uses_split(m)
    ex = m._executor(2, True)
    names = [lab[0] for lab in ex.fwd.labels + ex.bwd.labels + ex.opt.labels]
    assert any('pwconv_bwd_data_sb' in n for n in names) and any('split_bf16x3_batch' in n for n in names)
    # the weight gradients too (decided inside the library): the 2 x 33 x 33-row layers of this model report the split plan
    import ctypes
    out = (ctypes.c_int * 6)()
    L = load_pkg('_lib').lib()
    L.gemm_plan_query(9, 2 * 33 * 33, 96, 144, out)      # (rows, cin, cout) of a tiled-kernel layer at this size
    assert out[0] == 4, list(out)


@pytest.mark.parametrize('model_type,N', [('xception', 2), ('resnet50', 4)])
def test_the_dense_convs_take_the_split_kernels(model_type, N):
    """the models above really ran their dense convs on the split implicit-GEMM kernels: forward, data gradient (behind the split of
    the re-laid kernel) and -- decided inside the library -- the weight gradient"""
    import ctypes
    import numpy as np
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model(model_type, 21, (65, 65), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    m.use_graphs = False
    x = np.random.default_rng(0).uniform(-1, 1, (N, 65, 65, 3)).astype(np.float32)
    m.train_on_batch(x, np.zeros((N, 65 * 65, 1), np.float32))
    ex = m._executor(N, True)
    fwd, bwd, opt = ([lab[0] for lab in plan.labels] for plan in (ex.fwd, ex.bwd, ex.opt))
    assert any('conv2d_gemm_fwd_sb' in n for n in fwd), sorted(set(fwd))
    i = [j for j, n in enumerate(bwd) if 'conv2d_gemm_bwd_data_sb' in n]
    assert i and all('split_bf16x3_batch' in bwd[j - 1] and 'conv2d_gemm_dgrad_weights' in bwd[j - 2] for j in i), sorted(set(bwd))
    dense = [op for op in m.graph.ops if op.kind == 'conv_dense' and op in ex._sb_used_f]
    assert dense and any('split_bf16x3_batch' in n for n in opt)        # the optimiser step refreshes their planes
    L = load_pkg('_lib').lib()
    op = dense[0]
    assert L.conv2d_gemm_sb_pays(4, N * op.Ho * op.Wo, op.k * op.k * op.cin, op.cout)
    L.set_option(b'conv_sb', -1)        # the production rule: none of these few-row layers
    assert not L.conv2d_gemm_sb_pays(4, N * op.Ho * op.Wo, op.k * op.k * op.cin, op.cout)
    assert L.conv2d_gemm_sb_pays(4, 264196, 288, 64) and L.conv2d_gemm_sb_pays(1, 33800, 1152, 128) and not L.conv2d_gemm_sb_pays(1, 264196, 288, 64)


# (the fp32-input kernels keep their own op-level and 65 x 65 / 97 x 97 whole-step cases for Xception; its 513 x 513 repeat is once-per-release)
@pytest.mark.parametrize('model_type', ['mobilenetv2', pytest.param('xception', marks=pytest.mark.release)])
def test_train_step_513_on_the_fp32_input_mfma_kernels_only(model_type, monkeypatch):
    """DL3P_SPLIT_GEMM=0: the path the headline ran on until round 3, still selectable, still held to the same bounds"""
    monkeypatch.setenv('DL3P_SPLIT_GEMM', '0')
    import test_production_shapes_gpu as T
    T.test_train_step_513_production_dispatch(model_type)
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    m._ensure_store()
    assert m._store.Sb is None


def test_an_executor_that_did_not_take_the_optimiser_step_reads_fresh_planes(monkeypatch):
    """the optimiser step refreshes only the split planes its own executor's GEMMs read; an executor of another batch size on the
    same weights (predict after fit) must see every plane re-split first (Executor._sb_sync)"""
    import numpy as np
    pkg = load_pkg()
    rng = np.random.default_rng(3)
    S = 129

    def build():
        m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (S, S), 16, training=True)
        m.compile(optimizer=pkg.SGD(0.1), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
        m.use_graphs = False
        return m
    x1 = rng.uniform(-1, 1, (1, S, S, 3)).astype(np.float32)
    y1 = rng.integers(0, 21, (1, S * S, 1)).astype(np.float32)
    xb = rng.uniform(-1, 1, (2, S, S, 3)).astype(np.float32)
    for min_rows in (400, 150, 1500, 6000):          # a row threshold some map size sits under at batch 1 and over at batch 2
        monkeypatch.setenv('DL3P_SPLIT_MIN_ROWS', str(min_rows))
        monkeypatch.setenv('DL3P_SPLIT_MIN_ROWS_BN', str(min_rows))
        m = build()
        p0 = m.predict(xb)              # the batch-2 executor exists before the steps (its planes are fresh NOW)
        m.train_on_batch(x1, y1)
        exA, exB = m._executor(1, True), m._executor(2, False)
        if exB._sb_used_f - exA._sb_used_f:
            break
    else:
        raise AssertionError('no threshold made the two executors read different planes')
    for _ in range(2):
        m.train_on_batch(x1, y1)
    p = m.predict(xb)
    assert float(np.abs(p - p0).max()) > 1e-4          # the steps moved the weights
    m2 = build()
    m2.set_weights_by_name(m.get_weights_by_name())
    assert np.array_equal(m2.predict(xb), p)
