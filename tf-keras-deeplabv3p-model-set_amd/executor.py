"""Executor: binds a static layer graph (graph.py) to device buffers and traces it into launch plans
(forward, loss+backward, SGD) made of C-ABI calls (include/dl3p.h).

MI355X-first execution model: one process per GPU; every buffer (activations, gradients, the flat
parameter / gradient / momentum buffers, workspaces) is allocated once -- 288 GB of HBM make
activation recomputation and buffer juggling unnecessary -- and the traced plan is captured into a
hipGraph, so a training step is a single graph launch instead of ~900 eager kernel launches (no
tracing compiler, no per-op Python on the hot path).  Plans are segmented at collectives: under
data parallelism the gradient all-reduce (and SyncBatchNorm's statistics all-reduce) are RCCL calls
between graph segments.

PyTorch only provides memory, streams, graphs and torch.distributed here.
"""
import ctypes
import os
import zlib
import numpy as np
import torch

from ._lib import lib
from .graph import ACT_NONE

MAX_ROWS = 2048


def split_gemm_enabled():
    """The compute-bound pointwise convs (K >= 128, N >= 128, >= 16384 rows) run as fp32-accurate split-bf16 GEMMs on the bf16
    matrix pipe (csrc/pw_split.hip): every fp32 operand split exactly into three bf16 pieces, six cross products
    accumulated in fp32 -- parity held at the tolerances of the fp32-input MFMA kernels (tests/test_split_gemm_gpu.py, and the whole
    GPU suite runs with it).  On by default since round 3; DL3P_SPLIT_GEMM=0 (bench.py --split-gemm 0), read when the parameter
    store is built, keeps every GEMM on the fp32-input MFMA kernels."""
    return os.environ.get('DL3P_SPLIT_GEMM', '1') not in ('', '0')


class ParamStore:
    """all weights of a model in three flat float32 device buffers (value, gradient, momentum), laid out
    in Keras weight order with 16-byte aligned offsets.  One SGD launch and one all-reduce cover it."""

    def __init__(self, graph, device, bf16=False):
        self.graph = graph
        self.device = device
        self.bf16 = bool(bf16)
        self.params = graph.all_params()
        self.offset, off = param_offsets(self.params)
        self.total = off
        f = dict(dtype=torch.float32, device=device)
        self.P = torch.zeros(off, **f)
        self.G = torch.zeros(off, **f)
        self.V = torch.zeros(off, **f)
        self.V2 = None                        # second-moment buffer, allocated when Adam asks for it
        # optimiser iteration counter (Adam's bias correction, the dropout stream): one per model, shared by the
        # executors of every batch size, so a last partial batch continues the count instead of restarting it
        self.step = torch.zeros(1, dtype=torch.int64, device=device)
        # Adam's iteration count (bias correction) is a separate counter: compile() with a NEW optimizer object restarts it
        # (train.py:190-224 builds a fresh optimizer for the second stage) while the dropout stream above keeps counting --
        # in Keras the dropout RNG does not depend on optimizer.iterations, so stage 2 must not replay stage 1's masks
        self.opt_step = torch.zeros(1, dtype=torch.int64, device=device)
        self.l2 = torch.zeros(off, **f)
        self.lr_scale = torch.zeros(off, **f)
        # transposed copies [N][K] of the pointwise / im2col'd kernels (same offsets as in P): the forward GEMM reads
        # its B fragments from them as 16-B vectors; refreshed after every optimiser step (dl3p_transpose_batch)
        self.Pt = torch.zeros(off, **f)
        rows = [[self.offset[op.w], op.cin if op.kind == 'conv_pw' else op.kp, op.cout, 0]
                for op in graph.ops if op.kind in ('conv_pw', 'conv_dense')]
        self.tr_table = torch.tensor(rows, dtype=torch.int32, device=device) if rows else None
        # mixed precision: bf16 mirrors of the fp32 master copy, same offsets.  Pb = the parameters as they are (read by the
        # data-gradient GEMM as W[K][N] and by the depthwise kernels), Pbt = the [N][K] transposes the forward GEMM reads
        # (they take the place of Pt); both refreshed after every optimiser step
        self.Pb = torch.zeros(off, dtype=torch.bfloat16, device=device) if self.bf16 else None
        self.Pbt = torch.zeros(off, dtype=torch.bfloat16, device=device) if self.bf16 else None
        # fp32-accurate GEMMs on the bf16 matrix pipe (csrc/pw_split.hip; split_gemm_enabled()): a pointwise kernel pre-split into
        # three bf16 planes, [3][N][Kpad] (from the transposed copy: the forward's B operand) and [3][K][Npad] (from the kernel as
        # stored: the data gradient's), refreshed after every optimiser step like Pt.  Planes are allocated when the first GEMM
        # that reads them is traced (sb_ptr) -- 12 B per parameter for a layer that takes the split kernel, nothing for the many
        # that never do (few rows, short reductions, narrow outputs: for Xception that was 0.5 GB and a split of all of it at every
        # upload).  Sb = an anchor the job tables' destination offsets are relative to (None: split GEMMs off).
        self.Sb = None
        self.sb_fwd, self.sb_bwd = {}, {}           # op -> (its planes, pitch), once allocated
        self.sb_row_f, self.sb_row_b = {}, {}       # op -> its row of the dl3p_split_bf16x3_batch job table, once allocated
        self.sb_shape_f, self.sb_shape_b = {}, {}   # every op that may take the split kernels -> (rows, cols, pitch) of its operand
        self._sb_tables = None                      # (forward table, data-gradient table) over the allocated planes; None: stale
        self._sb_job_tables = []                    # one-row tables of the splits issued at allocation (read by launches in flight)
        # An optimiser step refreshes only the planes ITS executor's GEMMs read.  sb_partial = the executor whose subset is fresh,
        # None = every plane is; another executor on the same store (a different batch size, predict after fit) re-splits
        # everything allocated before it runs (Executor._sb_sync).
        self.sb_partial = None
        if split_gemm_enabled() and not self.bf16:
            for op in graph.ops:
                if op.kind == 'conv_pw':
                    K, Nn = op.cin, op.cout
                    self.sb_shape_f[op] = (Nn, K, (K + 31) // 32 * 32)
                    self.sb_shape_b[op] = (K, Nn, (Nn + 31) // 32 * 32)
                elif (op.kind == 'conv_dense' and op.kp == op.k * op.k * op.cin
                      and lib().conv2d_gemm_supported(op.cin, op.cout, op.k, op.stride)):
                    # dense k x k convs on the implicit-GEMM path (dl3p_conv2d_gemm_fwd_sb): the forward's [3][Cout][pitch >= k k Cin]
                    # from the transposed copy; the data gradient's operand is re-laid per step (conv2d_gemm_dgrad_weights) and
                    # split by the executor that owns that buffer
                    self.sb_shape_f[op] = (op.cout, op.kp, (op.kp + 31) // 32 * 32)
            if self.sb_shape_f:
                self.Sb = torch.zeros(64, dtype=torch.int16, device=device)
        self.upload()
        self.refresh_masks()

    def transpose(self):
        st = torch.cuda.current_stream().cuda_stream
        if self.bf16:
            lib().f32_to_bf16(self.P.data_ptr(), self.Pb.data_ptr(), self.total, st)
            if self.tr_table is not None:
                lib().transpose_batch_bf16(self.P.data_ptr(), self.Pbt.data_ptr(), self.tr_table.data_ptr(),
                                           int(self.tr_table.shape[0]), st)
            return
        if self.tr_table is not None:
            lib().transpose_batch(self.P.data_ptr(), self.Pt.data_ptr(), self.tr_table.data_ptr(),
                                  int(self.tr_table.shape[0]), st)
        if self.Sb is not None:
            if self._sb_tables is None:
                i64 = dict(dtype=torch.int64, device=self.device)
                self._sb_tables = (torch.tensor(list(self.sb_row_f.values()), **i64) if self.sb_row_f else None,
                                   torch.tensor(list(self.sb_row_b.values()), **i64) if self.sb_row_b else None)
            tf, tb = self._sb_tables
            if tf is not None:
                lib().split_bf16x3_batch(self.Pt.data_ptr(), self.Sb.data_ptr(), tf.data_ptr(), int(tf.shape[0]), st)
            if tb is not None:
                lib().split_bf16x3_batch(self.P.data_ptr(), self.Sb.data_ptr(), tb.data_ptr(), int(tb.shape[0]), st)
            self.sb_partial = None

    def sb_ptr(self, op, fwd):
        """(device pointer, pitch) of the op's pre-split planes; the first call allocates them and splits the current weights"""
        planes, rows_of, shapes = (self.sb_fwd, self.sb_row_f, self.sb_shape_f) if fwd else (self.sb_bwd, self.sb_row_b, self.sb_shape_b)
        if op not in planes:
            rows, cols, pitch = shapes[op]
            t = torch.zeros(3 * rows * pitch, dtype=torch.int16, device=self.device)
            delta = t.data_ptr() - self.Sb.data_ptr()       # (either sign; allocations are 256-byte aligned)
            assert delta % 16 == 0
            planes[op] = (t, pitch)
            rows_of[op] = [self.offset[op.w], rows, cols, cols, delta // 2, pitch]
            self._sb_tables = None
            tab = torch.tensor([rows_of[op]], dtype=torch.int64, device=self.device)
            self._sb_job_tables.append(tab)
            lib().split_bf16x3_batch((self.Pt if fwd else self.P).data_ptr(), self.Sb.data_ptr(), tab.data_ptr(), 1,
                                     torch.cuda.current_stream().cuda_stream)
        t, pitch = planes[op]
        return t.data_ptr(), pitch

    def view(self, p, buf=None):
        buf = self.P if buf is None else buf
        o = self.offset[p]
        return buf[o:o + p.dev_size].view(p.dev_shape)

    def ptr(self, p, buf=None):
        buf = self.P if buf is None else buf
        return buf.data_ptr() + buf.element_size() * self.offset[p]

    @staticmethod
    def _pad(p, value):
        a = np.asarray(value, dtype=np.float32)
        if len(p.dev_shape) != len(p.shape):      # dense conv kernel (k,k,cin,cout) -> [k*k*cin (padded)][cout]
            a = a.reshape(-1, p.shape[-1])
            return np.pad(a, [(0, p.dev_shape[0] - a.shape[0]), (0, p.dev_shape[1] - a.shape[1])])
        if p.dev_shape != p.shape:
            pad = [(0, d - s) for s, d in zip(p.shape, p.dev_shape)]
            a = np.pad(a, pad)
        return a

    def upload(self, only=None):
        host = np.zeros(self.total, np.float32)
        for p in self.params:
            o = self.offset[p]
            host[o:o + p.dev_size] = self._pad(p, p.value).reshape(-1)
        self.P.copy_(torch.from_numpy(host))
        self.transpose()

    def download(self):
        host = self.P.detach().cpu().numpy()
        for p in self.params:
            o = self.offset[p]
            p.value = self._unpad(p, host[o:o + p.dev_size].reshape(p.dev_shape))

    @staticmethod
    def _unpad(p, a):
        if len(p.dev_shape) != len(p.shape):
            rows = int(np.prod(p.shape[:-1]))
            return a[:rows, :p.shape[-1]].reshape(p.shape).copy()
        return a[tuple(slice(0, s) for s in p.shape)].copy()

    def get(self, p, buf=None):
        return self._unpad(p, self.view(p, buf).detach().cpu().numpy())

    def refresh_masks(self):
        """per-element l2 factor and learning-rate multiplier (0 = frozen / not a trainable weight)"""
        l2 = np.zeros(self.total, np.float32)
        lr = np.zeros(self.total, np.float32)
        for p in self.params:
            o = self.offset[p]
            l2[o:o + p.dev_size] = p.l2
            lr[o:o + p.dev_size] = 1.0 if p.trainable else 0.0
        self.l2.copy_(torch.from_numpy(l2))
        self.lr_scale.copy_(torch.from_numpy(lr))


def param_offsets(params):
    """flat-buffer offsets (in floats, 16-byte aligned) of the parameters in Keras weight order -> ({param: offset}, total)"""
    offset, off = {}, 0
    for p in params:
        offset[p] = off
        off += (p.dev_size + 3) // 4 * 4
    return offset, off


def bucket_edges(graph, offset, total, n_buckets):
    """Gradient buckets of the flat buffer for the data-parallel all-reduce.  -> ({op: (lo, hi)}, first_bucket_hi):
    when backward (reversed(graph.ops)) is ABOUT TO process `op`, every gradient in [lo, hi) is final and the slice can
    go on the wire; [0, first_bucket_hi) follows after the last op.

    The flat buffer is in layer-creation order but backward visits the ops in reversed EXECUTION order, and the two
    differ (layers._atrous_first runs the ASPP depthwise convs first, so their gradients are the LAST of the ASPP
    block to be written although they sit above image_pooling / aspp0 in the buffer).  A suffix [lo, total) is final
    only once no op that is still to be processed owns a parameter at or above lo: lo = the highest parameter end
    among the unprocessed layers (a BatchNormalization fused into its consumer's data-gradient kernel finishes
    earlier than its own position -- counting it at its position is conservative)."""
    rng = {}
    for p, o in offset.items():
        e = o + (p.dev_size + 3) // 4 * 4
        lo, hi = rng.get(p.layer, (o, e))
        rng[p.layer] = (min(lo, o), max(hi, e))
    ops_r = [o for o in reversed(graph.ops) if getattr(o, 'layer', None) in rng]
    pending_end = [0] * (len(ops_r) + 1)          # highest parameter end among ops_r[i:]
    for i in range(len(ops_r) - 1, -1, -1):
        pending_end[i] = max(pending_end[i + 1], rng[ops_r[i].layer][1])
    n = max(1, int(n_buckets))
    target = total / n
    edges, hi = {}, total
    for i, op in enumerate(ops_r):
        done_lo = pending_end[i]
        if done_lo < hi and hi - done_lo >= target and len(edges) < n - 1:
            edges[op] = (done_lo, hi)
            hi = done_lo
    return edges, hi


def _op_label(op):
    n = getattr(op, 'name', None)
    if n is None and getattr(op, 'bn', None) is not None and op.kind == 'bn':
        n = op.bn.layer.name
    if n is None and getattr(op, 'out', None) is not None:
        n = op.out.name
    return '%s:%s' % (op.kind, n)


class Plan:
    """a traced list of C-ABI launches (+ python callbacks at collectives), replayable eagerly or as
    hipGraph segments"""

    def __init__(self):
        self.items = []
        self.labels = []          # (entry point, graph op) per item, for scripts/step_table.py
        self.ctx = ''
        self.segments = None
        self.tags = {}

    def k(self, fn, *args, tag=None):
        fn(*args, torch.cuda.current_stream().cuda_stream)
        if tag is not None:
            self.tags[tag] = len(self.items)
        self.items.append((fn, args))
        self.labels.append((getattr(fn, '__name__', str(fn)), self.ctx))

    def probe(self, tag, probe):
        """keep one launch outside the graph segments and issue it with the library's HIP event pair
        (dl3p_probe_arm: kernel start/stop events on the launch stream)"""
        i = self.tags[tag]
        fn, args = self.items[i]
        L = lib()

        def timed():
            L.probe_arm(probe.base + probe.count % 2048)
            probe.count += 1
            fn(*args, torch.cuda.current_stream().cuda_stream)
        timed.no_capture = True
        self.items[i] = (None, timed)
        self.segments = None

    def py(self, fn):
        fn()
        self.items.append((None, fn))
        self.labels.append(('py', self.ctx))

    def coll(self, fn):
        """a collective (RCCL all-reduce); counted for the bench line"""
        self.n_collectives = getattr(self, 'n_collectives', 0) + 1
        self.py(fn)

    def run(self):
        if self.segments is not None:
            for seg in self.segments:
                if isinstance(seg, torch.cuda.CUDAGraph):
                    seg.replay()
                else:
                    seg()
            return
        st = torch.cuda.current_stream().cuda_stream
        for fn, args in self.items:
            if fn is None:
                args()
            else:
                fn(*args, st)

    def capture(self, collectives_in_graph=True):
        """turn the plan into hipGraphs.  With collectives_in_graph the RCCL calls (SyncBatchNorm
        statistics, gradient buckets on the side stream) are captured INTO the graph, so a whole
        forward / backward is one graph launch and the ~130 tiny all-reduces of a step cost GPU time
        only (no host round trip each).  If the runtime refuses to capture them, fall back to one graph
        per run of kernels with the collectives issued eagerly in between."""
        has_py = any(it[0] is None for it in self.items)
        if has_py and collectives_in_graph and not any(getattr(it[1], 'no_capture', False) for it in self.items if it[0] is None):
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    st = torch.cuda.current_stream().cuda_stream
                    for fn, args in self.items:
                        if fn is None:
                            args()
                        else:
                            fn(*args, st)
                self.segments = [g]
                self.single_graph = True
                return self
            except Exception as e:   # noqa: BLE001 - any capture failure -> segmented replay
                import warnings
                warnings.warn('collectives could not be captured into the hipGraph (%s); using segments' % (e,))
                torch.cuda.synchronize()
        segs, cur = [], []
        for it in self.items:
            if it[0] is None:
                if cur:
                    segs.append(cur)
                    cur = []
                segs.append(it[1])
            else:
                cur.append(it)
        if cur:
            segs.append(cur)
        out = []
        for seg in segs:
            if not isinstance(seg, list):
                out.append(seg)
                continue
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                st = torch.cuda.current_stream().cuda_stream
                for fn, args in seg:
                    fn(*args, st)
            out.append(g)
        self.segments = out
        return self

    @property
    def n_launches(self):
        return sum(1 for it in self.items if it[0] is not None)


class Probe:
    """`base`: first of the 2048 event slots this probe cycles through (two probes can run in one step)"""

    def __init__(self, op, kernel_name, base=0):
        self.op, self.kernel_name, self.count, self.first, self.base = op, kernel_name, 0, 0, base

    def reset(self):
        torch.cuda.synchronize()
        self.first = self.count

    def mean_ms(self):
        torch.cuda.synchronize()
        L = lib()
        ts = []
        for i in range(max(self.first, self.count - 2048), self.count):
            ms = ctypes.c_float(0)
            L.probe_read(self.base + i % 2048, ctypes.addressof(ms))
            ts.append(ms.value)
        return sum(ts) / max(1, len(ts))


class Executor:
    def __init__(self, graph, head, store, batch, training, num_classes, ignore_index=255, dist=None,
                 seed=1234, momentum=0.9, loss=('ce',), optimizer=None, sample_weighted=False, class_counts=False):
        self.g, self.head, self.store = graph, head, store
        self.N, self.training, self.C = batch, training, num_classes
        self.ignore_index = ignore_index
        # loss: ('ce',) | ('weighted', weights[C]) | ('focal', gamma, alpha)  (model.loss_spec)
        self._u8 = {}
        self._pool_arg = {}
        self.sample_weighted = bool(sample_weighted)
        self.want_class_counts = bool(class_counts)
        self.loss_kind = {'ce': 0, 'weighted': 1, 'focal': 2}[loss[0]]
        self.loss_gamma, self.loss_alpha = (float(loss[1]), float(loss[2])) if loss[0] == 'focal' else (0.0, 0.0)
        self._loss_weights_host = np.asarray(loss[1], np.float32) if loss[0] == 'weighted' else None
        # DL3P_FORCE_DIST=1 exercises the collective path on a single rank (plumbing test)
        force = bool(os.environ.get('DL3P_FORCE_DIST'))
        self.dist = dist if (dist is not None and (dist.world_size > 1 or force)) else None
        self.sync_bn = self.dist is not None and self.dist.sync_bn
        # optimizer: ('sgd', momentum) | ('adam', beta_1, beta_2, epsilon) | ('rmsprop', rho, epsilon)
        self.optimizer = optimizer or ('sgd', momentum)
        self.seed, self.momentum = seed, (self.optimizer[1] if self.optimizer[0] == 'sgd' else 0.0)
        if self.optimizer[0] == 'adam' and store.V2 is None:
            store.V2 = torch.zeros_like(store.V)
        self.dev = store.device
        self.L = lib()
        # which batch variance feeds BatchNormalization's moving average (dl3p_bn_finalize update_moving): 1 biased (Keras
        # SyncBatchNormalization), 2 Bessel-corrected (fused BatchNormalization); graph.bn_moving_variance, model.py
        self.moving_mode = 2 if getattr(graph, 'bn_moving_variance', 'biased') == 'unbiased' else 1
        self.f32 = dict(dtype=torch.float32, device=self.dev)
        # mixed precision (train.py:37-46, BASELINE configs[4]): activations / activation gradients in bf16, fp32 everywhere
        # else; the logits tensor (conv_upsample output) and its gradient stay fp32 so the softmax / loss head is unchanged
        self.bf16 = bool(getattr(store, 'bf16', False))
        self.adt = torch.bfloat16 if self.bf16 else torch.float32
        self._sb_used_f, self._sb_used_b = set(), set()      # pointwise convs whose forward / data gradient took the split GEMM
        # the weight gradients pick the split kernel inside the library (dl3p_pwconv_bwd_weight[_slabs], wgrad_sb_route): the switch
        # is process-wide there, so every executor states its own before it sizes workspaces, traces, or runs eagerly
        self._split_wgrad = int(split_gemm_enabled() and not self.bf16 and os.environ.get('DL3P_SPLIT_WGRAD', '1') not in ('', '0'))
        self.L.set_option(b'split_wgrad', self._split_wgrad)
        # ... and the other process-wide knobs that decide how many slabs / partial rows a traced launch writes: recorded now, pinned
        # again before every eager replay (_pin_options; a captured graph carries its launches' grids with it)
        self._pinned = {k: self.L.get_option(k) for k in (b'conv_sb', b'sb_rs', b'sb_pipe', b'splitk', b'sb3')}
        self._pinned_irb = tuple(self.L.irb_get_plan(i) for i in range(4))
        self._find_irb()
        self._find_up()
        self._alloc()
        # tracing runs every kernel once on zero inputs: keep the weights / optimiser state intact
        snap_p, snap_v, snap_step, snap_ostep = store.P.clone(), store.V.clone(), store.step.clone(), store.opt_step.clone()
        snap_v2 = store.V2.clone() if store.V2 is not None else None
        if training:
            self.fwd = self._trace_forward()
            self.bwd = self._trace_backward()
            self.opt = self._trace_sgd()
        else:
            self.fwd = self._trace_forward()
        torch.cuda.synchronize()
        store.P.copy_(snap_p)
        store.V.copy_(snap_v)
        if snap_v2 is not None:
            store.V2.copy_(snap_v2)
        store.transpose()
        store.step.copy_(snap_step)
        store.opt_step.copy_(snap_ostep)
        self.graphed = False

    # ---------------------------------------------------------------- buffers
    def _alloc(self):
        g, N = self.g, self.N
        self.buf, self.grad = {}, {}
        self._mark_requires_grad()
        direct_cols = {op.col.id for op in g.ops if op.kind == 'conv_dense' and (self._stem_direct(op) or self._dense_gemm(op))}
        # the data-gradient GEMM of an implicit-GEMM conv reads the kernel as [Cin][k*k*Cout] (rebuilt every step)
        self.dense_wd_sb = {}       # id(op) -> (its three bf16 planes, the one-row job table of the split)
        self.dense_wd = {id(op): torch.zeros(op.k * op.k * op.cin * op.cout, **self.f32) for op in g.ops
                         if op.kind == 'conv_dense' and self.training and self._dense_gemm(op) and op.k > 1}
        # (a strided 1x1 conv keeps the gradient of its compact patch matrix = dz @ W^T on the output pixels: scattering
        # that onto the strided input pixels is cheaper than a data-gradient GEMM over all input pixels, 3/4 of them zero)
        grad_cols = {op.col.id for op in g.ops if op.kind == 'conv_dense' and self._dense_gemm(op) and op.k == 1}
        for t in g.tensors:
            dt = torch.float32 if t is self.head.tensor else self.adt
            if t.id in direct_cols and not (t.id in grad_cols and self.training and t.requires_grad):
                continue
            if t.id in self._irb_tensors:          # the expanded tensor of a fused inverted-residual block is never in HBM
                if self._irb_keep_z:               # (test hook: a copy for the parity tests to look at; no kernel of the step reads it)
                    self.buf[t.id] = torch.zeros(N * t.H * t.W * t.C, dtype=dt, device=self.dev)
                continue
            if not getattr(t, 'grad_only', False) and t.id not in direct_cols:
                self.buf[t.id] = torch.zeros(N * t.H * t.W * t.C, dtype=dt, device=self.dev)
            if self.training and t.requires_grad:
                self.grad[t.id] = torch.zeros(N * t.H * t.W * t.C, dtype=dt, device=self.dev)
        self.gscale, self.gshift = {}, {}
        for grp in g.groups:
            self.gscale[grp.id] = torch.ones(grp.C, **self.f32)
            self.gshift[grp.id] = torch.zeros(grp.C, **self.f32)
        self.bn_aux = {}
        cmax = 4
        for bn in g.bns:
            self.bn_aux[bn] = dict(mean=torch.zeros(bn.C, **self.f32), invstd=torch.ones(bn.C, **self.f32),
                                   coef=torch.zeros(3 * bn.C, **self.f32),
                                   sums=torch.zeros(2 * bn.C, dtype=torch.float64, device=self.dev))
            cmax = max(cmax, bn.C)
        # fused inverted-residual blocks: covariance rows / sums of the block input (dl3p_irb_cov_stats), float64
        self.irb_cov_rows = self.irb_cov_sums = None
        if self._irb_expand and self.training:
            kmax = max(op.cin for op in self._irb_expand)
            self.irb_cov_cap = self.L.irb_cov_rows_max()
            self.irb_cov_rows = torch.zeros(self.irb_cov_cap * (kmax + kmax * kmax), dtype=torch.float64, device=self.dev)
            self.irb_cov_sums = torch.zeros(kmax + kmax * kmax, dtype=torch.float64, device=self.dev)
        self.partials = torch.zeros(MAX_ROWS * 2 * cmax, **self.f32)
        self.partials2 = torch.zeros(MAX_ROWS * 2 * cmax, **self.f32) if self.training else None   # sums that wait (_presums)
        # SyncBatchNorm: the (sum, sum^2) / (sum dy, sum dy xhat) vectors of BatchNorms whose statistics are needed at the
        # same point of the graph travel in ONE all-reduce.  Slices of this buffer are handed out in trace order, so the
        # BatchNorms waiting for a flush are contiguous (forward and backward use separate halves).
        irb_k = {b.bn: e.cin for e, b, d in self._irb_expand.values()}
        # (a fused block's expand BatchNorm stages the covariance sums of its K-channel input in the forward half)
        self._stage_len = {bn: max(2 * bn.C, irb_k[bn] + irb_k[bn] ** 2) if bn in irb_k else 2 * bn.C for bn in g.bns}
        self._sync_total = sum(self._stage_len.values())
        self.sync_stage = (torch.zeros(2 * self._sync_total, dtype=torch.float64, device=self.dev)
                           if (self.sync_bn and self.training) else None)
        ws = 1 << 20
        L = self.L
        pw_ws = L.pwconv_bwd_weight_workspace_bf16 if self.bf16 else L.pwconv_bwd_weight_workspace
        dw_ws = L.dwconv2d_bwd_weight_workspace_bf16 if self.bf16 else L.dwconv2d_bwd_weight_workspace
        for op in g.ops:
            if op.kind == 'conv_pw':
                ws = max(ws, pw_ws(N * op.Ho * op.Wo, op.cin, op.cout))
            elif op.kind == 'conv_dw':
                ws = max(ws, dw_ws(N, op.Ho, op.Wo, op.c, op.k))
            elif op.kind == 'conv_dense':
                ws = max(ws, pw_ws(N * op.Ho * op.Wo, op.kp, op.cout))
                if self._stem_direct(op):
                    ws = max(ws, L.stem_conv_bwd_weight_workspace(N, op.Ho, op.Wo, op.cout))
                if self._dense_gemm(op):
                    ws = max(ws, L.conv2d_gemm_bwd_weight_workspace(N, op.Ho, op.Wo, op.cin, op.cout, op.k))
        self.workspace = torch.zeros(ws // 4 + 4, **self.f32) if self.training else None
        # slabs of the split-K forward GEMMs (dl3p_pwconv_fwd_wt_splitk), training and inference
        sk = 0 if self.bf16 else max([L.pwconv_fwd_splitk_workspace(N * op.Ho * op.Wo, op.cin, op.cout)
                                      for op in g.ops if op.kind == 'conv_pw'] + [0])
        self.splitk_ws = torch.zeros(sk // 4 + 4, **self.f32) if sk else None
        # tickets + partial rows of the chunked per-image reductions (pooling, SE backward): zero once, every call
        # leaves its tickets at zero again; one stream runs all of them
        pws = 0
        for op in g.ops:
            if op.kind in ('gap', 'se_mul', 'broadcast'):
                t = op.x.tensor if op.kind != 'broadcast' else op.out
                pws = max(pws, (L.pool_workspace_bf16 if self.bf16 else L.pool_workspace)(N, t.H * t.W, t.C))
        self.pool_ws = torch.zeros(pws // 4 + 4, **self.f32)
        self.pool_wsb = pws
        H, W, _ = g.input_shape
        self.H, self.W = H, W
        self.cpad = self.head.tensor.C       # padded class count of the logits rows (a multiple of 4; of 8 on the bf16 path)
        self.labels = torch.zeros(N * H * W, **self.f32)
        self.loss_partials = torch.zeros(MAX_ROWS, **self.f32)
        self.loss = torch.zeros(1, **self.f32)
        # The fused training heads (plain cross-entropy only).  dl3p_head_train_rows (x / y separable: a quarter of the full-resolution
        # gradient's traffic) is the default where it is served; DL3P_FUSED_HEAD=0: the two-kernel path, =tile: dl3p_head_train (one
        # launch, no workspace, bit-identical to the two-kernel path but measured slower: 506 us vs 141 + 165 us at batch 16).
        zt = self.head.tensor
        self.class_weights = None
        if self._loss_weights_host is not None:
            assert self._loss_weights_host.shape == (self.C,), 'one class weight per class'
            self.class_weights = torch.from_numpy(self._loss_weights_host).to(self.dev)
        # Keras sample weights, sample_weight_mode='temporal' (train.py:116-120): ones until set_inputs receives some
        self.pixel_weights = torch.ones(N * H * W, **self.f32) if (self.training and self.sample_weighted) else None
        # per-image class counts for the Jaccard training metric (deeplabv3p/metrics.py:29-46), refreshed every step
        self.metric_counts = (torch.zeros(N * 3 * self.C, dtype=torch.int32, device=self.dev).view(N, 3, self.C)
                              if (self.training and self.want_class_counts) else None)
        want = os.environ.get('DL3P_FUSED_HEAD', 'rows')
        plain = bool(self.training and self.loss_kind == 0 and not self.sample_weighted and
                     self.class_weights is None and zt is not None and zt.requires_grad)
        # (the logits and their gradient stay fp32 under the bf16 policy -- _is_f32 -- so the separable head serves it too; the padded
        # classes of the gradient buffer are never written by anyone and keep the zeros they were allocated with)
        self.fused_head_rows = bool(plain and want == 'rows' and L.head_train_rows_supported(zt.H, zt.W, self.C, H, W))
        self.fused_head = self.fused_head_rows or bool(plain and not self.bf16 and want not in ('0', 'rows') and
                                                       L.head_train_supported(zt.H, zt.W, self.C, H, W))
        self.head_wsb = L.head_train_rows_workspace(N, zt.H, zt.W, self.C, H, W) if self.fused_head_rows else 0
        self.head_ws = torch.zeros(self.head_wsb // 4, **self.f32) if self.fused_head_rows else None
        self.dlogits_big = (torch.zeros(N * H * W * self.cpad, **self.f32)
                            if (self.training and not self.fused_head) else None)
        self.probs = None if self.training else torch.zeros(N * H * W * self.C, **self.f32)
        self.logits_big = None
        self.step = self.store.step
        self.lr = torch.full((1,), 0.01, **self.f32)

    # ---------------------------------------------------------------- fused inverted-residual blocks
    def _find_irb(self):
        """expand 1x1 conv -> BatchNorm -> activation -> 3x3 depthwise conv (deeplabv3p_mobilenetv2.py:43-60) as ONE unit where the
        expanded tensor is large: csrc/irb_fwd.hip / irb_bwd.hip recompute the expand conv instead of storing its output.
        self._irb_expand / _irb_dw / _irb_bn: the three ops of a fused block -> (expand, bn op, depthwise); fp32, every layer of the
        block trainable (training) -- anything else keeps the unfused kernels.  Under SyncBatchNorm the expand BatchNorm's
        covariance sums (K + K*K doubles) and its backward sums travel in the same all-reduces as every other BatchNorm's"""
        self._irb_expand, self._irb_dw, self._irb_bn, self._irb_tensors = {}, {}, {}, set()
        # DL3P_IRB_DEBUG_Z=1: the forward ALSO writes the expand output with the unfused kernel, for tests that inspect every conv
        # output (activation branch patterns); the fused kernels never read it
        self._irb_keep_z = os.environ.get('DL3P_IRB_DEBUG_Z', '0') == '1'
        if self.bf16 or os.environ.get('DL3P_IRB', '1') == '0':
            return
        # the fused backward leaves its weight gradients as slabs for the batched slab reduction: with that switched off
        # (DL3P_BATCHED_WGRAD=0, an A/B switch) a training executor keeps the unfused kernels
        if self.training and os.environ.get('DL3P_BATCHED_WGRAD', '1') == '0':
            return
        g, N, L = self.g, self.N, self.L
        # measured on the headline step: the 257 x 257 and 129 x 129 blocks pay (12.75 -> 12.22 ms), the 65 x 65 ones do not (12.47 with them)
        min_rows = int(os.environ.get('DL3P_IRB_MIN_ROWS', '131072'))
        readers = {}
        for op in g.ops:
            for slot in ('x', 'r', 's'):
                v = getattr(op, slot, None)
                if v is not None:
                    readers.setdefault(v.tensor.root.id, []).append((op, slot, v))
            if op.kind == 'bn':
                readers.setdefault(op.z.root.id, []).append((op, 'z', None))
        views = {t.id for (t, _, _) in g.act_views.values()}
        for e in g.ops:
            if e.kind != 'conv_pw' or e.b is not None or e.bn is None or e.out.base is not None or e.out is self.head.tensor:
                continue
            rd = readers.get(e.out.id, [])
            if len(rd) != 2 or e.out.id in views:
                continue
            bn_ops = [o for o, slot, _ in rd if o.kind == 'bn' and slot == 'z']
            dws = [(o, v) for o, slot, v in rd if o.kind == 'conv_dw' and slot == 'x']
            if len(bn_ops) != 1 or len(dws) != 1:
                continue
            b, (d, v) = bn_ops[0], dws[0]
            if b.bn is not e.bn or v.bn is not e.bn or v.tensor is not e.out or getattr(v, 'view_grad', None) is not None:
                continue
            xt = e.x.tensor
            if N * xt.H * xt.W < min_rows or e.x.tensor.ld % 4 or xt.H != e.Ho or xt.W != e.Wo:
                continue
            # a stride-1 depthwise conv reaches every expanded pixel with all nine taps (a stride-2 one with 2.25 on average): the
            # recomputing backward then costs what the unfused kernels cost (129 x 129 x 24 -> 144: 437 against 421 us per step)
            if d.stride == 1 and N * xt.H * xt.W < int(os.environ.get('DL3P_IRB_S1_MIN_ROWS', '400000')) and 'DL3P_IRB_MIN_ROWS' not in os.environ:
                continue
            geo = (N, xt.H, xt.W, e.cin, e.cout, d.k, d.stride, d.rate, d.pad_t, d.pad_l, d.Ho, d.Wo)
            if d.c != e.cout or not L.irb_supported(*geo):
                continue
            if self.training:
                lay = (e.layer.trainable, b.bn.layer.trainable, d.layer.trainable)
                if not all(lay) or not L.irb_bwd_supported(*geo):
                    continue
            rec = (e, b, d)
            self._irb_expand[e], self._irb_bn[b], self._irb_dw[d] = rec, rec, rec
            self._irb_tensors.add(e.out.id)

    def _find_up(self):
        """Decoder_block (layers.py:207-215): img_resize -> Concatenate([x, skip]) -> 3x3 depthwise conv.  Where the resized tensor's
        only reader is that conv, the resize launch goes and the conv's forward and weight gradient form those channels from the
        low-resolution map while they load (dl3p_dw_upsampled_input: the resize kernel's arithmetic, bit for bit) -- 272 MB less
        written and 2 x 272 MB less read per step at BASELINE configs[1].  self._up_conv: conv_dw op -> the resize op it absorbs"""
        self._up_conv, self._up_resize = {}, set()
        # OPT-IN (DL3P_FOLD_RESIZE=1): built, bit-identical, and measured SLOWER on MI355X -- 12.62 against 12.02 ms per headline step
        # (profiles/r06_resize_fold.txt): the fused forward takes 468 us against 140 + 80 for the pair it replaces.  Four dependent
        # 16-byte loads and three lerps per input vector turn an HBM-bound window kernel (2 waves per SIMD at 217-256 registers) into
        # a latency-bound one: the compiler serialises the conditional gathers (57 s_waitcnt vmcnt(0) in the loop against 9)
        if self.bf16 or os.environ.get('DL3P_FOLD_RESIZE', '0') != '1':
            return
        g, N, L = self.g, self.N, self.L
        for r in g.ops:
            if r.kind != 'resize' or r.out.base is None or r.out.c0 != 0:
                continue
            root = r.out.base
            readers = [o for o in g.ops for slot in ('x', 'r', 's') if getattr(o, slot, None) is not None
                       and getattr(o, slot).tensor.root is root and o is not r]
            # (the other slices of the buffer are WRITTEN by their producers -- `out`, not a read -- and read through the root)
            if len(readers) != 1 or readers[0].kind != 'conv_dw':
                continue
            d = readers[0]
            if d.x.tensor is not root or d.x.group is None or d.x.act == ACT_NONE or d.k != 3 or d.stride != 1 or d.rate != 1:
                continue
            xl = r.x.tensor
            if xl.base is not None or r.x.group is not None or xl.C != r.out.C or xl.ld % 4 or root.ld % 4:
                continue
            geo = (N, root.H, root.W, d.c, r.out.C, d.k, d.stride, d.rate, d.pad_t, d.pad_l, d.Ho, d.Wo)
            if not L.dw_upsampled_input_supported(0, *geo):
                continue
            if self.training and d.layer.trainable and not L.dw_upsampled_input_supported(1, *geo):
                continue
            self._up_conv[d] = r
            self._up_resize.add(r)

    def _up_args(self, d):
        """the arguments of dl3p_dw_upsampled_input for a conv that absorbed its resize (None otherwise)"""
        r = self._up_conv.get(d)
        if r is None:
            return None
        xl = r.x.tensor
        return (self.tptr(xl), xl.ld, xl.H, xl.W, xl.C)

    def _irb_geo(self, rec):
        e, b, d = rec
        xt = e.x.tensor
        return (self.N, xt.H, xt.W, e.cin, e.cout, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo)

    def _irb_args(self, rec):
        """(x, ldx, in_scale, in_shift, in_act, w1, bn_scale, bn_shift, bn_act): the leading arguments of every dl3p_irb_* call"""
        e, b, d = rec
        bn = b.bn
        xp, ldx, sp, hp, act = self.vargs(e.x)
        return (xp, ldx, sp, hp, act, self.store.ptr(e.w), self.gscale[bn.group.id].data_ptr() + 4 * bn.offset,
                self.gshift[bn.group.id].data_ptr() + 4 * bn.offset, bn.act)

    def _stem_direct(self, op):
        """the RGB stem runs as the LDS-staged implicit GEMM (csrc/stem.hip) instead of im2col + GEMM: fp32 path, raw image
        input (no lazy BatchNorm / activation on it, no gradient wanted for it), no bias"""
        v = op.x
        xt = v.tensor
        return (not self.bf16 and op.b is None and v.is_plain
                and not (xt.requires_grad or xt.root.requires_grad)
                and self.L.stem_conv_supported(op.cin, op.cout, op.k, op.stride, op.rate)
                and os.environ.get('DL3P_STEM_DIRECT', '1') != '0')

    def _slab_bytes(self, op):
        """workspace bytes of a conv's weight gradient when it can leave its slabs for the batched reduction (0: it has a bias
        gradient, runs on <= 64 rows, or is a dense conv on the im2col route)"""
        L, N = self.L, self.N
        if op in self._irb_expand or op in self._irb_dw:
            e, b, d = self._irb_expand.get(op) or self._irb_dw[op]
            xt = e.x.tensor
            return L.irb_bwd_workspace(1 if op is e else 0, N, xt.H, xt.W, e.cin, e.cout, d.stride, d.pad_t, d.pad_l)
        if getattr(op, 'b', None) is not None or N * op.Ho * op.Wo <= 64:
            return 0
        if self.bf16:
            if op.kind == 'conv_dw':
                return L.dwconv2d_bwd_weight_workspace_bf16(N, op.Ho, op.Wo, op.c, op.k)
            return L.pwconv_bwd_weight_workspace_bf16(N * op.Ho * op.Wo, op.cin if op.kind == 'conv_pw' else op.kp, op.cout)
        if op.kind == 'conv_pw':
            return L.pwconv_bwd_weight_workspace(N * op.Ho * op.Wo, op.cin, op.cout)
        if op.kind == 'conv_dw':
            return L.dwconv2d_bwd_weight_workspace(N, op.Ho, op.Wo, op.c, op.k)
        if self._stem_direct(op):
            return L.stem_conv_bwd_weight_workspace(N, op.Ho, op.Wo, op.cout)
        if self._dense_gemm(op):
            return L.conv2d_gemm_bwd_weight_workspace(N, op.Ho, op.Wo, op.cin, op.cout, op.k)
        return 0

    def _reduce_all(self, P, jobs):
        """one dl3p_reduce_rows_batched call for every (slabs, destination, rows, n) job of the step"""
        import numpy as np
        L = self.L
        rec = np.zeros(len(jobs), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('rows', '<i4'), ('n', '<i4')]))
        maps = ([], [])
        for j, (src, dst, rows, n) in enumerate(jobs):
            rec[j] = (src, dst, rows, n)
            v = L.reduce_rows_variant(rows, n)
            be = L.reduce_rows_block_elements(v)
            maps[v].extend((j, b) for b in range((n + be - 1) // be))
        jt = torch.from_numpy(rec.view(np.uint8).copy()).to(self.dev)
        mt = [torch.tensor(m if m else [(0, 0)], dtype=torch.int32, device=self.dev) for m in maps]
        self._wgrad_tables.append((jt, mt))         # the launches read them at every replay
        self._wgrad_jobs = jt
        P.k(L.reduce_rows_batched, jt.data_ptr(), mt[0].data_ptr(), len(maps[0]), mt[1].data_ptr(), len(maps[1]))

    def _dense_gemm(self, op):
        """dense conv with Cin % 4 == 0 on the fp32 path: implicit GEMM, the patch operand gathered while the GEMM stages
        its A tile (csrc/pwconv.hip, dl3p_conv2d_gemm_*) -- no im2col matrix, no col2im pass"""
        xt = op.x.tensor
        return (not self.bf16 and not self._stem_direct(op)
                and bool(self.L.conv2d_gemm_supported(op.cin, op.cout, op.k, op.stride))
                # the gather decodes row indices in 24-bit arithmetic; larger tensors keep the im2col route
                and self.N * xt.H * xt.W < (1 << 24) and self.N * op.Ho * op.Wo < (1 << 24))

    def _mark_requires_grad(self):
        for t in self.g.tensors:
            t.requires_grad = False
        slices = {}

        def rg(v):
            t = v.tensor
            return t.requires_grad or t.root.requires_grad

        def setrg(t, val):
            if val:
                t.requires_grad = True
                t.root.requires_grad = True
        for op in self.g.ops:
            k = op.kind
            if k in ('conv_pw', 'conv_dense', 'conv_dw'):
                setrg(op.out, op.layer.trainable or rg(op.x))
                if k == 'conv_dense':
                    setrg(op.col, rg(op.x))
            elif k == 'bn':
                setrg(op.z, op.layer.trainable)
            elif k == 'materialize':
                setrg(op.out, rg(op.x) or (op.r is not None and rg(op.r)))
            elif k in ('gap', 'resize', 'broadcast', 'maxpool'):
                setrg(op.out, rg(op.x))
            elif k == 'se_mul':
                setrg(op.out, rg(op.x) or rg(op.s))
        for (t, act, vt) in self.g.act_views.values():
            setrg(vt, t.requires_grad or t.root.requires_grad)

    # ---------------------------------------------------------------- addressing helpers
    def tptr(self, t, grad=False):
        store = self.grad if grad else self.buf
        b = store[t.root.id]
        return b.data_ptr() + b.element_size() * t.c0

    def _is_f32(self, t):
        return 1 if (t.root is self.head.tensor or t is self.head.tensor) else 0

    def vargs(self, v):
        """(ptr, ld, scale_ptr, shift_ptr, act) of a Value as a kernel prologue"""
        sp = hp = None
        if v.group is not None:
            sp = self.gscale[v.group.id].data_ptr() + 4 * v.goff
            hp = self.gshift[v.group.id].data_ptr() + 4 * v.goff
        return self.tptr(v.tensor), v.tensor.ld, sp, hp, v.act

    def view(self, t, grad=False, weights=None):
        """torch view (N,H,W,C) of a graph tensor (test / debug hook).  weights: {parameter name: array} standing in for the
        parameter store where a tensor has to be re-formed (after a training step the store holds the UPDATED kernels; a test that
        wants the tensor as the forward saw it passes the kernels the step started from)"""
        store = self.grad if grad else self.buf
        root = t.root
        if root.id not in store and root.id in self._irb_tensors and not grad:
            # the expand output of a fused inverted-residual block exists in no buffer: formed here, for the caller only, by the
            # unfused pointwise kernel from the block input as the device holds it
            from . import ops
            e = [o for o in self._irb_expand if o.out is root][0]
            v = e.x
            xin = self.view(v.tensor)
            sc = sh = None
            if v.group is not None:
                sc = self.gscale[v.group.id][v.goff:v.goff + v.tensor.C]
                sh = self.gshift[v.group.id][v.goff:v.goff + v.tensor.C]
            if weights is not None and e.w.name in weights:
                w = torch.from_numpy(np.ascontiguousarray(weights[e.w.name], dtype=np.float32).reshape(e.cin, e.cout)).to(self.dev)
            else:
                w = self.store.view(e.w).reshape(e.cin, e.cout)
            return ops.pwconv_fwd(xin.reshape(-1, e.cin), w, in_scale=sc, in_shift=sh, in_act=v.act).view(self.N, root.H, root.W, root.C)
        full = store[root.id].view(self.N, root.H, root.W, root.C)
        return full[..., t.c0:t.c0 + t.C]

    # ---------------------------------------------------------------- forward
    def _trace_forward(self):
        P, L, N, st = Plan(), self.L, self.N, self.store
        train = self.training
        if train:
            P.k(L.increment_counter, self.step.data_ptr())
            if self.optimizer[0] == 'adam':
                P.k(L.increment_counter, self.store.opt_step.data_ptr())
        self._fwd_pending, self._fwd_stage_off = [], 0
        self._irb_stage = {}
        for op in self.g.ops:
            k = op.kind
            P.ctx = _op_label(op)
            if self._fwd_pending and k not in ('bn', 'broadcast') and self._reads_pending(op):
                self._flush_bn_forward(P)
            if k in ('conv_pw', 'conv_dense', 'conv_dw'):
                # (the depthwise conv of a fused inverted-residual block reads no buffer of its own: its input is recomputed)
                xp, ldx, sp, hp, act = self.vargs(op.x) if op not in self._irb_dw else (None, 0, None, None, ACT_NONE)
                bn = op.bn
                want_stats = train and bn is not None and bn.layer.trainable
                part = self.partials.data_ptr() if want_stats else None
                rows = ctypes.c_int(0)
                xt = op.x.tensor
                if op in self._irb_expand:
                    # fused block: the expand output is never formed.  Its BatchNorm's statistics come from the covariance of the
                    # block INPUT (z = x W is linear): one pass over the K-channel tensor, finalised at the 'bn' op
                    if want_stats:
                        crow = ctypes.c_int(0)
                        P.k(L.irb_cov_stats, xp, ldx, sp, hp, act, self.irb_cov_rows.data_ptr(), self.irb_cov_cap, ctypes.byref(crow),
                            N * xt.H * xt.W, op.cin)
                        if self.sync_bn:
                            off = self._fwd_stage_off
                            self._fwd_stage_off += self._stage_len[bn]
                            self._irb_stage[bn] = off
                            P.k(L.irb_cov_reduce, self.irb_cov_rows.data_ptr(), crow.value, op.cin, self.sync_stage[off:].data_ptr())
                        else:
                            P.k(L.irb_cov_reduce, self.irb_cov_rows.data_ptr(), crow.value, op.cin, self.irb_cov_sums.data_ptr())
                    if self._irb_keep_z:
                        P.k(L.pwconv_fwd_wt, xp, ldx, sp, hp, act, st.ptr(op.w, st.Pt), None, self.tptr(op.out), op.out.ld, None,
                            ctypes.byref(ctypes.c_int(0)), N * op.Ho * op.Wo, op.cin, op.cout)
                elif op in self._irb_dw:
                    rec = self._irb_dw[op]
                    P.k(L.irb_fwd, *self._irb_args(rec), st.ptr(op.w), self.tptr(op.out), op.out.ld, part, ctypes.byref(rows),
                        *self._irb_geo(rec), tag=op.name)
                elif self.bf16:
                    self._conv_forward_bf16(P, op, xp, ldx, sp, hp, act, part, rows)
                elif k == 'conv_pw' and self._use_sb(op, True, part is not None):
                    wsp, pitch = st.sb_ptr(op, True)
                    P.k(L.pwconv_fwd_sb, xp, ldx, sp, hp, act, wsp, pitch, st.ptr(op.b) if op.b else None,
                        self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), N * op.Ho * op.Wo, op.cin, op.cout,
                        tag='pw:' + op.name)
                elif k == 'conv_pw' and self.splitk_ws is not None and L.pwconv_fwd_splitk_plan(N * op.Ho * op.Wo, op.cin, op.cout):
                    # few rows, long reduction (Xception's / ResNet50's ASPP 1x1 convs on the 33 x 33 map): split-K, two launches
                    P.k(L.pwconv_fwd_wt_splitk, xp, ldx, sp, hp, act, st.ptr(op.w, st.Pt), st.ptr(op.b) if op.b else None,
                        self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), self.splitk_ws.data_ptr(), self.splitk_ws.numel() * 4,
                        N * op.Ho * op.Wo, op.cin, op.cout, tag='pw:' + op.name)
                elif k == 'conv_pw':
                    P.k(L.pwconv_fwd_wt, xp, ldx, sp, hp, act, st.ptr(op.w, st.Pt), st.ptr(op.b) if op.b else None,
                        self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), N * op.Ho * op.Wo, op.cin, op.cout,
                        tag='pw:' + op.name)
                elif k == 'conv_dw':
                    if self._up_args(op) is not None:
                        P.k(L.dw_upsampled_input, *self._up_args(op))
                    P.k(L.dwconv2d_fwd, xp, ldx, sp, hp, act, st.ptr(op.w), self.tptr(op.out), op.out.ld, part,
                        ctypes.byref(rows), N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l,
                        op.Ho, op.Wo, tag=op.name)
                elif self._stem_direct(op):
                    P.k(L.stem_conv_fwd, xp, ldx, st.ptr(op.w), self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), N,
                        xt.H, xt.W, op.cout, op.pad_t, op.pad_l, op.Ho, op.Wo, tag=op.name)
                elif self._dense_gemm(op) and self._use_sb_dense(op, 1 if part is not None else 0):
                    wsp, pitch = st.sb_ptr(op, True)
                    P.k(L.conv2d_gemm_fwd_sb, xp, ldx, sp, hp, act, wsp, pitch, st.ptr(op.b) if op.b else None,
                        self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), N, xt.H, xt.W, op.cin, op.cout, op.k,
                        op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo, tag=op.name)
                elif self._dense_gemm(op):
                    P.k(L.conv2d_gemm_fwd, xp, ldx, sp, hp, act, st.ptr(op.w, st.Pt), st.ptr(op.b) if op.b else None,
                        self.tptr(op.out), op.out.ld, part, ctypes.byref(rows), N, xt.H, xt.W, op.cin, op.cout, op.k,
                        op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo, tag=op.name)
                else:
                    # what is left (Cin not a multiple of 4, e.g. a 7x7 RGB stem): im2col once, then the MFMA GEMM
                    P.k(L.im2col, xp, ldx, sp, hp, act, self.tptr(op.col), op.col.ld, N, xt.H, xt.W, op.cin, op.k,
                        op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
                    P.k(L.pwconv_fwd_wt, self.tptr(op.col), op.col.ld, None, None, ACT_NONE, st.ptr(op.w, st.Pt),
                        st.ptr(op.b) if op.b else None, self.tptr(op.out), op.out.ld, part, ctypes.byref(rows),
                        N * op.Ho * op.Wo, op.kp, op.cout)
                op.rows = rows.value
            elif k == 'bn':
                self._bn_forward(P, op)
            elif k == 'materialize':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                rp = ldr = rsp = rhp = None
                ract = ACT_NONE
                if op.r is not None:
                    rp, ldr, rsp, rhp, ract = self.vargs(op.r)
                rate = op.rate if train else 0.0
                t = op.out
                P.k(L.affine_act_bf16 if self.bf16 else L.affine_act, xp, ldx, sp, hp, act, rp, ldr or 0, rsp, rhp, ract, float(rate),
                    self._dropout_seed(op), self.step.data_ptr(), self.tptr(t), t.ld, N * t.H * t.W, t.C)
            elif k == 'gap':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                xt = op.x.tensor
                P.k(L.global_avgpool_fwd_bf16 if self.bf16 else L.global_avgpool_fwd, xp, ldx, sp, hp, act, self.tptr(op.out),
                    op.out.ld, 1.0, N, xt.H * xt.W, xt.C, self.pool_ws.data_ptr(), self.pool_wsb)
            elif k == 'maxpool':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                xt, t = op.x.tensor, op.out
                arg = None
                if train:          # the backward pass reads the winning taps instead of re-evaluating the windows
                    arg = self._pool_arg[op] = torch.empty(N * op.Ho * op.Wo * xt.C, dtype=torch.uint8, device=self.dev)
                P.k(L.maxpool2d_fwd_bf16 if self.bf16 else L.maxpool2d_fwd, xp, ldx, sp, hp, act, self.tptr(t), t.ld,
                    None if arg is None else arg.data_ptr(), N,
                    xt.H, xt.W, xt.C, op.k, op.stride, op.pad_t, op.pad_l, op.Ho, op.Wo)
            elif k == 'se_mul':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                s_ptr, lds, _, _, sact = self.vargs(op.s)
                t = op.out
                P.k(L.scale_bcast_fwd_bf16 if self.bf16 else L.scale_bcast_fwd, xp, ldx, sp, hp, act, s_ptr, lds, sact,
                    self.tptr(t), t.ld, N, t.H * t.W, t.C)
            elif k in ('resize', 'broadcast'):
                if op in self._up_resize:
                    continue                # (formed inside the depthwise conv that reads it: _find_up)
                xt, t = op.x.tensor, op.out
                P.k(L.resize_bilinear_fwd_bf16 if self.bf16 else L.resize_bilinear_fwd, self.tptr(xt), xt.ld, self.tptr(t),
                    t.ld, N, xt.H, xt.W, xt.C, t.H, t.W)
            else:
                raise NotImplementedError(k)
        if self._fwd_pending:
            self._flush_bn_forward(P)
        # head: pred_resize + softmax (+ loss and its gradient when training)
        P.ctx = 'head'
        zt = self.head.tensor
        rows = ctypes.c_int(0)
        if train and self.fused_head:
            # loss + d loss / d (conv_upsample output) in one launch; the full-resolution gradient is never written
            if self.fused_head_rows:
                P.k(L.head_train_rows, self.tptr(zt), zt.ld, self.labels.data_ptr(), int(self.ignore_index or 0),
                    1.0 / float(N * self.H * self.W), self.tptr(zt, True), zt.ld, 0, self.loss_partials.data_ptr(),
                    ctypes.byref(rows), self.head_ws.data_ptr(), self.head_wsb, N, zt.H, zt.W, self.C, self.H, self.W)
            else:
                P.k(L.head_train, self.tptr(zt), zt.ld, self.labels.data_ptr(), int(self.ignore_index or 0),
                    1.0 / float(N * self.H * self.W), self.tptr(zt, True), zt.ld, 0, self.loss_partials.data_ptr(),
                    ctypes.byref(rows), N, zt.H, zt.W, self.C, self.H, self.W)
            P.k(L.reduce_rows, self.loss_partials.data_ptr(), rows.value, 1, self.loss.data_ptr(), 0)
        elif train:
            P.k(L.upsample_softmax_loss, self.tptr(zt), zt.ld, self.labels.data_ptr(), int(self.ignore_index or 0),
                1.0 / float(N * self.H * self.W), self.loss_kind,
                None if self.class_weights is None else self.class_weights.data_ptr(), self.loss_gamma, self.loss_alpha,
                None if self.pixel_weights is None else self.pixel_weights.data_ptr(),
                None, None, self.dlogits_big.data_ptr(), self.cpad,
                self.loss_partials.data_ptr(), ctypes.byref(rows), N, zt.H, zt.W, self.C, self.H, self.W)
            P.k(L.reduce_rows, self.loss_partials.data_ptr(), rows.value, 1, self.loss.data_ptr(), 0)
        else:
            P.k(L.upsample_softmax_ce, self.tptr(zt), zt.ld, None, 0, 1.0, None, self.probs.data_ptr(), None, self.cpad,
                None, ctypes.byref(rows), N, zt.H, zt.W, self.C, self.H, self.W)
        if train and self.metric_counts is not None:
            P.k(L.fill, self.metric_counts.data_ptr(), 0.0, N * 3 * self.C)       # int32 zeros share the bit pattern
            P.k(L.class_counts, self.tptr(zt), zt.ld, self.labels.data_ptr(), self.metric_counts.data_ptr(), N, zt.H, zt.W,
                self.C, self.H, self.W)
        return P

    def _conv_forward_bf16(self, P, op, xp, ldx, sp, hp, act, part, rows):
        L, N, st, k = self.L, self.N, self.store, op.kind
        xt = op.x.tensor
        if k == 'conv_pw':
            P.k(L.pwconv_fwd_bf16, xp, ldx, self._is_f32(xt), sp, hp, act, st.ptr(op.w, st.Pbt), st.ptr(op.b) if op.b else None,
                self.tptr(op.out), op.out.ld, self._is_f32(op.out), part, ctypes.byref(rows), N * op.Ho * op.Wo, op.cin, op.cout)
        elif k == 'conv_dw':
            P.k(L.dwconv2d_fwd_bf16, xp, ldx, sp, hp, act, st.ptr(op.w, st.Pb), self.tptr(op.out), op.out.ld, part,
                ctypes.byref(rows), N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo,
                tag=op.name)
        else:
            P.k(L.im2col_bf16, xp, ldx, sp, hp, act, self.tptr(op.col), op.col.ld, N, xt.H, xt.W, op.cin, op.k, op.stride,
                op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
            P.k(L.pwconv_fwd_bf16, self.tptr(op.col), op.col.ld, 0, None, None, ACT_NONE, st.ptr(op.w, st.Pbt),
                st.ptr(op.b) if op.b else None, self.tptr(op.out), op.out.ld, self._is_f32(op.out), part, ctypes.byref(rows),
                N * op.Ho * op.Wo, op.kp, op.cout)

    def _dropout_seed(self, op):
        return (self.seed * 1000003 + zlib.crc32((op.dropout_name or '').encode()) % 65521) & 0x7FFFFFFFFFFFFFFF

    def _bn_forward(self, P, op):
        bn, L, st, N = op.bn, self.L, self.store, self.N
        aux = self.bn_aux[bn]
        lp = {p.key: p for p in bn.layer.params}
        sp = self.gscale[bn.group.id].data_ptr() + 4 * bn.offset
        hp = self.gshift[bn.group.id].data_ptr() + 4 * bn.offset
        z = op.z
        count = float(N * z.H * z.W)
        if self.training and bn.layer.trainable and self.sync_bn and op in self._irb_bn:
            self._fwd_pending.append((op, self._irb_stage[bn], P.ctx))      # (its covariance sums are in the slice already)
        elif self.training and bn.layer.trainable and self.sync_bn:
            # local sums into this BatchNorm's slice of the staging buffer; the all-reduce and the finalize wait until a
            # consumer needs the coefficients (_flush_bn_forward), together with every other BatchNorm pending by then
            off = self._fwd_stage_off
            self._fwd_stage_off += self._stage_len[bn]
            sl = self.sync_stage[off:off + 2 * bn.C]
            P.k(L.bn_reduce_partials, self.partials.data_ptr(), op.producer.rows, 2 * bn.C, sl.data_ptr())
            self._fwd_pending.append((op, off, P.ctx))
        elif self.training and bn.layer.trainable and op in self._irb_bn:
            e = self._irb_bn[op][0]
            P.k(L.irb_bn_finalize_cov, self.irb_cov_sums.data_ptr(), st.ptr(e.w), e.cin, e.cout, count, st.ptr(lp['gamma']),
                st.ptr(lp['beta']), bn.eps, bn.momentum, st.ptr(lp['moving_mean']), st.ptr(lp['moving_variance']),
                self.moving_mode, sp, hp, aux['mean'].data_ptr(), aux['invstd'].data_ptr())
        elif self.training and bn.layer.trainable:
            rows = op.producer.rows
            sums = None
            P.k(L.bn_finalize, self.partials.data_ptr(), rows, sums, bn.C, count, st.ptr(lp['gamma']),
                st.ptr(lp['beta']), bn.eps, bn.momentum, st.ptr(lp['moving_mean']), st.ptr(lp['moving_variance']),
                self.moving_mode, sp, hp, aux['mean'].data_ptr(), aux['invstd'].data_ptr())
        else:
            P.k(L.bn_infer_coeffs, st.ptr(lp['gamma']), st.ptr(lp['beta']), st.ptr(lp['moving_mean']),
                st.ptr(lp['moving_variance']), bn.eps, sp, hp, aux['mean'].data_ptr(), aux['invstd'].data_ptr(), bn.C)

    def _reads_pending(self, op):
        """does `op` apply the coefficients of a BatchNorm whose statistics are still waiting for their all-reduce?"""
        groups = {id(b.bn.group) for b, _, _ in self._fwd_pending}
        # (a fused block reads the coefficients in FRONT of it twice: in the covariance pass at its expand op, in the fused launch at
        # its depthwise op)
        if op in self._irb_dw and self._reads_pending_value(self._irb_dw[op][0].x, groups):
            return True
        for slot in ('x', 'r', 's'):
            v = getattr(op, slot, None)
            if v is not None and v.group is not None and id(v.group) in groups:
                return True
        return False

    @staticmethod
    def _reads_pending_value(v, groups):
        return v is not None and v.group is not None and id(v.group) in groups

    def _flush_bn_forward(self, P):
        """ONE all-reduce for every pending BatchNorm (their slices are contiguous), then their finalize kernels"""
        pend, self._fwd_pending = self._fwd_pending, []
        lo = pend[0][1]
        hi = pend[-1][1] + self._stage_len[pend[-1][0].bn]
        ctx = P.ctx
        P.ctx = 'syncbn:' + '+'.join(b.bn.name for b, _, _ in pend)
        P.coll(lambda t=self.sync_stage[lo:hi]: self.dist.all_reduce(t))
        st, L = self.store, self.L
        for op, off, c in pend:
            bn = op.bn
            P.ctx = c
            aux = self.bn_aux[bn]
            lp = {p.key: p for p in bn.layer.params}
            sp = self.gscale[bn.group.id].data_ptr() + 4 * bn.offset
            hp = self.gshift[bn.group.id].data_ptr() + 4 * bn.offset
            count = float(self.N * op.z.H * op.z.W) * self.dist.world_size
            if op in self._irb_bn:
                e = self._irb_bn[op][0]
                P.k(L.irb_bn_finalize_cov, self.sync_stage[off:].data_ptr(), st.ptr(e.w), e.cin, e.cout, count, st.ptr(lp['gamma']),
                    st.ptr(lp['beta']), bn.eps, bn.momentum, st.ptr(lp['moving_mean']), st.ptr(lp['moving_variance']),
                    self.moving_mode, sp, hp, aux['mean'].data_ptr(), aux['invstd'].data_ptr())
                continue
            P.k(L.bn_finalize, None, 0, self.sync_stage[off:].data_ptr(), bn.C, count, st.ptr(lp['gamma']),
                st.ptr(lp['beta']), bn.eps, bn.momentum, st.ptr(lp['moving_mean']), st.ptr(lp['moving_variance']),
                self.moving_mode, sp, hp, aux['mean'].data_ptr(), aux['invstd'].data_ptr())
        P.ctx = ctx

    # ---------------------------------------------------------------- backward
    def _acc(self, t):
        """accumulate flag for a write into grad(t): 0 the first time, 1 afterwards"""
        key = t.id
        if key in self._written or (t.base is not None and t.base.id in self._written):
            return 1
        self._written.add(key)
        return 0

    def _gbuf(self, v):
        """(ptr, ld, key tensor) of the buffer that collects d/d(value v)"""
        vt = getattr(v, 'view_grad', None)
        if vt is not None:
            # bare activation of a materialised tensor: consumers write d/d(act(T)) here; folded into T's
            # gradient (g * act'(T)) when backward reaches T's producer
            self._pending_views.setdefault(v.tensor.id, []).append((v.tensor, v.act, vt))
            return self.tptr(vt, grad=True), vt.ld, vt
        # a Concatenate value carries the activation of its branches: each branch's BatchNormalization
        # backward applies act' to its own channel slice, so the buffer itself takes the plain gradient
        if v.bn is None and v.group is None and v.act != ACT_NONE:
            raise NotImplementedError('activation-only lazy value without a view buffer: ' + v.tensor.name)
        return self.tptr(v.tensor, grad=True), v.tensor.ld, v.tensor

    def _flush_views(self, P, t):
        """T.grad (+)= view.grad * act'(T) for every bare-activation view of T"""
        done = set()
        for (tt, act, vt) in self._pending_views.pop(t.id, []):
            if vt.id in done:
                continue
            done.add(vt.id)
            P.k(self.L.bn_bwd_apply_bf16 if self.bf16 else self.L.bn_bwd_apply, self.tptr(vt, True), vt.ld, self.tptr(tt),
                tt.ld, None, None, act, None, None, None, self.tptr(tt, True), tt.ld, self._acc(tt), self.N * tt.H * tt.W, tt.C)

    def _trace_backward(self):
        P, L, N, st = Plan(), self.L, self.N, self.store
        self._written = set()
        self._pending_views = {}
        G = st.G
        zt = self.head.tensor
        # d(loss)/d(pred_resize output) -> d/d(conv_upsample output): transpose of the bilinear upsample
        if zt.requires_grad and self.fused_head:
            self._acc(zt)        # the forward plan's head_train launch already wrote grad(zt)
        elif zt.requires_grad:
            P.k(L.resize_bilinear_bwd, self.dlogits_big.data_ptr(), self.cpad, self.tptr(zt, True), zt.ld,
                self._acc(zt), N, zt.H, zt.W, zt.C, self.H, self.W)
        ws, wsb = self.workspace.data_ptr(), self.workspace.numel() * 4
        # data parallel: the flat gradient buffer is produced back to front; each finished bucket is
        # all-reduced on the side stream while the remaining backward kernels run
        bucket_edges = self._bucket_edges() if self.dist is not None else {}
        # SyncBatchNorm: a layer's weight gradient feeds nothing in the backward chain, so it is held back and
        # issued while the NEXT BatchNorm's statistics all-reduce is on the wire (hides the collective's latency)
        self._deferred = []
        defer = self.sync_bn
        self._deferred_mode = bool(defer)
        fuse = self._bn_fusion_map()
        if self.bf16:
            # bf16: the pointwise GEMMs can carry the sums (dl3p_pwconv_bwd_data_bn_bf16, more than 64 rows), the depthwise
            # kernels cannot.  Opt-in: measured slower than the separate reduce pass (the epilogue meets z in 8-byte pieces:
            # MobileNetV2 513x513 batch 16 12.37 against 11.83 ms, MobileNetV3-Large 1024x2048 batch 1 6.92 against 6.86)
            fuse = {c: b for c, b in fuse.items() if c.kind == 'conv_pw' and self.N * c.Ho * c.Wo > 64
                    and os.environ.get('DL3P_BF16_FUSE_BN_BWD', '0') == '1'}
        bn_done = set()
        self._bwd_pending, self._bwd_stage_off = [], self._sync_total
        processed = set()
        readers = self._bn_readers()
        bn_of = {id(o.z): o for o in self.g.ops if o.kind == 'bn'}
        self._galias = {}                 # z tensor id -> (ptr, ld) of the buffer that already holds d/d(BN(z) output)
        alias_ok = os.environ.get('DL3P_GRAD_ALIAS', '1') != '0'
        self._sync_g = {}                 # SyncBatchNorm: 'bn' op -> (ptr, ld) its apply reads the gradient from (after the all-reduce)
        # BatchNorm -> residual Add -> pointwise conv: the conv's data gradient is the last writer of d/d(Add output), which
        # IS the gradient of the BatchNorm output, so it carries that BatchNorm's backward sums too (no bn_bwd_reduce pass)
        fuse_add = self._bn_fusion_through_adds(readers) if not self.bf16 else {}
        self._presums = {}                # 'bn' op -> partial rows left in self.partials2 by such a data gradient
        rops = list(reversed(self.g.ops))
        self._bwd_ctx = (fuse, fuse_add, bn_done)       # (what a fused block's second pass needs when it is issued from a flush)

        def wgrad(fn, *args):
            if defer:
                self._deferred.append((fn, args, P.ctx))
            else:
                P.k(fn, *args)

        # the weight-gradient kernels leave their (fp32) slabs in per-layer regions of one buffer and ONE pair of launches
        # reduces them all (dl3p_reduce_rows_batched; 65 reduce launches of 5-13 us each otherwise) -- at the end of
        # backward on one GPU, per gradient bucket under data parallelism (in front of the bucket's all-reduce).  Same
        # per-element arithmetic as the per-layer reduction, whichever way the jobs are grouped.
        batch = os.environ.get('DL3P_BATCHED_WGRAD', '1') != '0'
        self._batch_wgrad = batch and not defer
        if self._batch_wgrad and not self.bf16 and getattr(self, 'dz_scratch', None) is None:
            need = max([N * op.Ho * op.Wo * (op.cout if op.kind == 'conv_pw' else op.c) for op in self.g.ops
                        if op.kind in ('conv_pw', 'conv_dw') and op not in self._irb_expand] or [0])
            self.dz_scratch = torch.zeros(need + 64, **self.f32)      # dz of the layer whose weight gradient just ran
        self._folded = {}                 # z tensor id -> BatchNorm-backward apply arguments taken over by the conv's wgrad
        self._dz_not_kept = set()         # (debug record) z tensors whose gradient buffer still holds d/d(BN output) after backward
        self._folded_dg = {}              # ... taken over by the conv's DATA gradient (row-stationary split GEMM, pw_split_rs.hip)
        jobs = self._jobs = []            # (slab pointer, destination pointer, rows, n) issued and not yet reduced
        self._wgrad_tables = []
        slab_off = self._slab_off = [0]
        slab_need = 0
        if batch:
            for op in self.g.ops:
                if op.kind in ('conv_pw', 'conv_dense', 'conv_dw') and op.layer.trainable and self._slab_bytes(op):
                    slab_need += (self._slab_bytes(op) + 255) // 256 * 256
            self.slab_ws = torch.zeros(slab_need // 4 + 64, **self.f32) if slab_need else None

        def wgrad_slabs(fn, n, dst, nbytes, *args, up=None):
            """fn(*args[:split], region, bytes, &rows, *args[split:]) with args given as (before, after); up: the arguments of the
            dl3p_dw_upsampled_input call that describes this launch's input (a conv that absorbed its resize)"""
            before, after = args
            region = self.slab_ws.data_ptr() + 4 * slab_off[0]
            slab_off[0] += ((nbytes + 255) // 256 * 256) // 4

            def issue(P2):
                rows = ctypes.c_int(0)
                if up is not None:
                    P2.k(L.dw_upsampled_input, *up)
                P2.k(fn, *before, region, nbytes, ctypes.byref(rows), *after)
                self._jobs.append((region, dst, rows.value, n))
            if defer:
                self._deferred.append((issue, None, P.ctx))      # runs at the next _flush_deferred, like the plain ones
            else:
                issue(P)
        for ri, op in enumerate(rops):
            k = op.kind
            P.ctx = _op_label(op)
            # SyncBatchNorm: the producer of a pending BatchNorm's input needs that BatchNorm's dz now
            if self._bwd_pending and getattr(op, 'out', None) is not None and any(b.z is op.out for b, _, _ in self._bwd_pending):
                self._flush_bn_backward(P)
            if op in bucket_edges:
                self._flush_deferred(P)        # every gradient of the finished bucket must have been produced
                self._reduce_pending(P)        # ... and reduced from its slabs
                lo, hi = bucket_edges[op]
                P.coll(lambda lo=lo, hi=hi: self.dist.all_reduce_async(G[lo:hi]))
            out = getattr(op, 'out', None)
            if out is not None and out.id in self._pending_views:
                self._flush_views(P, out)
            if k == 'bn':
                if op.z.requires_grad and op not in bn_done:
                    self._bn_backward(P, op)
                    bn_done.add(op)
                    if self.sync_bn and op.bn.layer.trainable:
                        # other BatchNorms whose gradient is complete already (every reader of their output has been
                        # processed: the ASPP branches behind concat_projection, a shortcut beside its residual branch):
                        # take their local sums now, so that one all-reduce serves them all
                        for op2 in rops[ri + 1:]:
                            if (op2.kind == 'bn' and op2 not in bn_done and op2.z.requires_grad and op2.bn.layer.trainable
                                    and readers.get(op2) and readers[op2] <= processed and op2 not in fuse.values()):
                                ctx = P.ctx
                                P.ctx = _op_label(op2)
                                self._bn_backward(P, op2)
                                bn_done.add(op2)
                                P.ctx = ctx
                processed.add(op)
                continue
            processed.add(op)
            if out is None or not out.requires_grad:
                continue
            if op in self._irb_expand:
                continue                      # its gradients were produced together with the depthwise conv's (below)
            if op in self._irb_dw:
                self._irb_backward(P, self._irb_dw[op], fuse, fuse_add, bn_done, batch)
                continue
            if self.bf16 and k in ('conv_pw', 'conv_dense', 'conv_dw'):
                fused_bn = fuse.get(op) if (op in fuse and fuse[op].z.requires_grad) else None
                self._conv_backward_bf16(P, op, wgrad, ws, wsb, wgrad_slabs if batch else None, fused_bn)
                if fused_bn is not None:
                    bn_done.add(fused_bn)
            elif k in ('conv_pw', 'conv_dense', 'conv_dw'):
                xp, ldx, sp, hp, act = self.vargs(op.x)
                xt = op.x.tensor
                dz, lddz = self.tptr(out, True), out.ld
                need_gx = xt.requires_grad or xt.root.requires_grad
                dgrad_done = False
                if k == 'conv_pw' and out.id in self._folded_dg:
                    # data gradient FIRST: it forms dz = BatchNorm-backward apply of (g, z) in its staging pass and leaves it at
                    # (dz, lddz) for the weight gradient below
                    fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef, dz, lddz = self._folded_dg.pop(out.id)
                    gp, ldg, keyt = self._gbuf(op.x)
                    acc = self._acc(keyt)
                    wsp, pitch = st.sb_ptr(op, False)
                    M_ = N * op.Ho * op.Wo
                    fold_args = (fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef, dz, lddz, wsp, pitch, gp, ldg, acc, M_, op.cin, op.cout)
                    front = fuse.get(op) if (op in fuse and fuse[op].z.requires_grad) else None
                    front_add = fuse_add.get(op) if (front is None and op in fuse_add and fuse_add[op].z.requires_grad) else None
                    bn_front = front or front_add
                    rows_f = ctypes.c_int(0)
                    if bn_front is not None:
                        bnf = bn_front.bn
                        auxf = self.bn_aux[bnf]
                        part_f = self.partials if front is not None else self.partials2
                        P.k(L.pwconv_bwd_data_sb_apply, *fold_args, self.tptr(bn_front.z), bn_front.z.ld,
                            self.gscale[bnf.group.id].data_ptr() + 4 * bnf.offset, self.gshift[bnf.group.id].data_ptr() + 4 * bnf.offset,
                            bnf.act, auxf['mean'].data_ptr(), auxf['invstd'].data_ptr(), part_f.data_ptr(), ctypes.byref(rows_f))
                    else:
                        P.k(L.pwconv_bwd_data_sb_apply, *fold_args, None, 0, None, None, ACT_NONE, None, None, None, None)
                    dgrad_done = True
                if op.layer.trainable and batch and self._slab_bytes(op):
                    gw = st.ptr(op.w, G)
                    nb = self._slab_bytes(op)
                    if k == 'conv_pw' and out.id in self._folded:
                        fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef = self._folded.pop(out.id)
                        dz, lddz = (self.dz_scratch.data_ptr(), op.cout) if need_gx else (None, 0)
                        wgrad_slabs(L.pwconv_bwd_weight_slabs_bn, op.cin * op.cout, gw, nb,
                                    (xp, ldx, sp, hp, act, fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef, dz, lddz),
                                    (N * op.Ho * op.Wo, op.cin, op.cout))
                    elif k == 'conv_pw':
                        wgrad_slabs(L.pwconv_bwd_weight_slabs, op.cin * op.cout, gw, nb, (xp, ldx, sp, hp, act, dz, lddz),
                                    (N * op.Ho * op.Wo, op.cin, op.cout))
                    elif k == 'conv_dw' and out.id in self._folded:
                        fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef = self._folded.pop(out.id)
                        dz, lddz = (self.dz_scratch.data_ptr(), op.c) if need_gx else (None, 0)
                        wgrad_slabs(L.dwconv2d_bwd_weight_slabs_bn, op.k * op.k * op.c, gw, nb,
                                    (xp, ldx, sp, hp, act, fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef, dz, lddz),
                                    (N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo),
                                    up=self._up_args(op))
                    elif k == 'conv_dw':
                        wgrad_slabs(L.dwconv2d_bwd_weight_slabs, op.k * op.k * op.c, gw, nb, (xp, ldx, sp, hp, act, dz, lddz),
                                    (N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo),
                                    up=self._up_args(op))
                    elif self._stem_direct(op) and out.id in self._folded:
                        # the stem has no data gradient: dz = BatchNorm-backward apply of (g, z) is formed in the weight gradient's
                        # staging pass and never written
                        fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef = self._folded.pop(out.id)
                        wgrad_slabs(L.stem_conv_bwd_weight_slabs_bn, 28 * op.cout, gw, nb,
                                    (xp, ldx, fg, fldg, fz, fldz, fsp, fhp, fact, fmean, finv, fcoef),
                                    (N, xt.H, xt.W, op.cout, op.pad_t, op.pad_l, op.Ho, op.Wo))
                    elif self._stem_direct(op):
                        wgrad_slabs(L.stem_conv_bwd_weight_slabs, 28 * op.cout, gw, nb, (xp, ldx, dz, lddz),
                                    (N, xt.H, xt.W, op.cout, op.pad_t, op.pad_l, op.Ho, op.Wo))
                    else:
                        wgrad_slabs(L.conv2d_gemm_bwd_weight_slabs, op.k * op.k * op.cin * op.cout, gw, nb,
                                    (xp, ldx, sp, hp, act, dz, lddz),
                                    (N, xt.H, xt.W, op.cin, op.cout, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo))
                elif op.layer.trainable:
                    gw = st.ptr(op.w, G)
                    if k == 'conv_pw':
                        wgrad(L.pwconv_bwd_weight, xp, ldx, sp, hp, act, dz, lddz, gw, st.ptr(op.b, G) if op.b else None,
                              ws, wsb, N * op.Ho * op.Wo, op.cin, op.cout)
                    elif k == 'conv_dw':
                        if self._up_args(op) is not None:
                            wgrad(L.dw_upsampled_input, *self._up_args(op))
                        wgrad(L.dwconv2d_bwd_weight, xp, ldx, sp, hp, act, dz, lddz, gw, ws, wsb, N, xt.H, xt.W, op.c,
                              op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
                    elif self._stem_direct(op):
                        wgrad(L.stem_conv_bwd_weight, xp, ldx, dz, lddz, gw, ws, wsb, N, xt.H, xt.W, op.cout, op.pad_t,
                              op.pad_l, op.Ho, op.Wo)
                    elif self._dense_gemm(op):
                        wgrad(L.conv2d_gemm_bwd_weight, xp, ldx, sp, hp, act, dz, lddz, gw, st.ptr(op.b, G) if op.b else None,
                              ws, wsb, N, xt.H, xt.W, op.cin, op.cout, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho,
                              op.Wo)
                    else:
                        wgrad(L.pwconv_bwd_weight, self.tptr(op.col), op.col.ld, None, None, ACT_NONE, dz, lddz, gw,
                              st.ptr(op.b, G) if op.b else None, ws, wsb, N * op.Ho * op.Wo, op.kp, op.cout)
                if dgrad_done:
                    # (the data gradient ran in front of the weight gradient; what is left is the BatchNorm in front of the conv)
                    if front is not None:
                        ctx = P.ctx
                        P.ctx = _op_label(front)
                        self._bn_backward(P, front, fused_rows=rows_f.value)
                        P.ctx = ctx
                        bn_done.add(front)
                    elif front_add is not None:
                        self._presums[front_add] = rows_f.value
                elif need_gx:
                    gp, ldg, keyt = self._gbuf(op.x)
                    acc = self._acc(keyt)
                    if k == 'conv_pw' and op in fuse and fuse[op].z.requires_grad:
                        bn_op = fuse[op]
                        bn = bn_op.bn
                        aux = self.bn_aux[bn]
                        rows = ctypes.c_int(0)
                        self._pw_dgrad_bn(P, op, dz, lddz, gp, ldg, acc, bn_op, self.partials, rows)
                        ctx = P.ctx
                        P.ctx = _op_label(bn_op)
                        self._bn_backward(P, bn_op, fused_rows=rows.value)
                        P.ctx = ctx
                        bn_done.add(bn_op)
                    elif k == 'conv_pw' and op in fuse_add and fuse_add[op].z.requires_grad:
                        bn_op = fuse_add[op]
                        bn = bn_op.bn
                        aux = self.bn_aux[bn]
                        rows = ctypes.c_int(0)
                        self._pw_dgrad_bn(P, op, dz, lddz, gp, ldg, acc, bn_op, self.partials2, rows)
                        self._presums[bn_op] = rows.value
                    elif k == 'conv_pw' and self._use_sb(op, False, False):
                        wsp, pitch = st.sb_ptr(op, False)
                        P.k(L.pwconv_bwd_data_sb, dz, lddz, wsp, pitch, gp, ldg, acc, N * op.Ho * op.Wo, op.cin, op.cout,
                            None, 0, None, None, ACT_NONE, None, None, None, None)
                    elif k == 'conv_pw':
                        P.k(L.pwconv_bwd_data, dz, lddz, st.ptr(op.w), gp, ldg, acc, N * op.Ho * op.Wo, op.cin,
                            op.cout)
                    elif k == 'conv_dw' and op in fuse and fuse[op].z.requires_grad:
                        bn_op = fuse[op]
                        bn = bn_op.bn
                        aux = self.bn_aux[bn]
                        rows = ctypes.c_int(0)
                        P.k(L.dwconv2d_bwd_data_bn, dz, lddz, st.ptr(op.w), gp, ldg, acc, N, xt.H, xt.W, op.c, op.k,
                            op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo, self.tptr(bn_op.z), bn_op.z.ld,
                            self.gscale[bn.group.id].data_ptr() + 4 * bn.offset,
                            self.gshift[bn.group.id].data_ptr() + 4 * bn.offset, bn.act, aux['mean'].data_ptr(),
                            aux['invstd'].data_ptr(), self.partials.data_ptr(), ctypes.byref(rows))
                        ctx = P.ctx
                        P.ctx = _op_label(bn_op)
                        self._bn_backward(P, bn_op, fused_rows=rows.value)
                        P.ctx = ctx
                        bn_done.add(bn_op)
                    elif k == 'conv_dw':
                        P.k(L.dwconv2d_bwd_data, dz, lddz, st.ptr(op.w), gp, ldg, acc, N, xt.H, xt.W, op.c, op.k,
                            op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
                    elif k == 'conv_dense' and self._dense_gemm(op) and op.k > 1:
                        wd = self.dense_wd[id(op)]
                        P.k(L.conv2d_gemm_dgrad_weights, st.ptr(op.w), wd.data_ptr(), op.k, op.cin, op.cout)
                        if self._use_sb_dense(op, 2):
                            # the re-laid kernel [Cin][k k Cout] split into its three bf16 planes (a few hundred KB), then the
                            # data gradient on the bf16 matrix pipe
                            kd = op.k * op.k * op.cout
                            pitch = (kd + 31) // 32 * 32
                            sp3 = self.dense_wd_sb.get(id(op))
                            if sp3 is None:
                                sp3 = self.dense_wd_sb[id(op)] = (torch.zeros(3 * op.cin * pitch, dtype=torch.int16, device=self.dev),
                                                                  torch.tensor([[0, op.cin, kd, kd, 0, pitch]], dtype=torch.int64, device=self.dev))
                            P.k(L.split_bf16x3_batch, wd.data_ptr(), sp3[0].data_ptr(), sp3[1].data_ptr(), 1)
                            P.k(L.conv2d_gemm_bwd_data_sb, dz, lddz, sp3[0].data_ptr(), pitch, gp, ldg, acc, N, xt.H, xt.W, op.cin, op.cout,
                                op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
                        else:
                            P.k(L.conv2d_gemm_bwd_data, dz, lddz, wd.data_ptr(), gp, ldg, acc, N, xt.H, xt.W, op.cin, op.cout,
                                op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
                    else:
                        # d/d(im2col matrix) by the GEMM, then the transposed gather back onto the input pixels
                        P.k(L.pwconv_bwd_data, dz, lddz, st.ptr(op.w), self.tptr(op.col, True), op.col.ld, 0,
                            N * op.Ho * op.Wo, op.kp, op.cout)
                        P.k(L.col2im, self.tptr(op.col, True), op.col.ld, gp, ldg, acc, N, xt.H, xt.W, op.cin, op.k,
                            op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
            elif k == 'materialize':
                gt, ldt = self.tptr(out, True), out.ld
                M = N * out.H * out.W
                if op.x.tensor.requires_grad or op.x.tensor.root.requires_grad:
                    xb = bn_of.get(id(op.x.tensor)) if op.x.bn is not None else None
                    if (alias_ok and xb is not None and float(op.rate) == 0.0 and getattr(op.x, 'view_grad', None) is None
                            and op.x.tensor.C == out.C and readers.get(xb) == {op} and op.x.tensor.id not in self._written
                            and (op.x.tensor.base is None or op.x.tensor.base.id not in self._written)):
                        # a residual Add hands its gradient to the BatchNorm branch unchanged: that BatchNorm's backward
                        # reads it where it is (no copy into the branch's own buffer, one launch less per Add)
                        self._galias[op.x.tensor.id] = (gt, ldt)
                        self._written.add(op.x.tensor.id)
                    else:
                        gp, ldg, keyt = self._gbuf(op.x)
                        P.k(L.scale_mask_bwd_bf16 if self.bf16 else L.scale_mask_bwd, gt, ldt, float(op.rate),
                            self._dropout_seed(op), self.step.data_ptr(), gp, ldg, self._acc(keyt), M, out.C)
                if op.r is not None and (op.r.tensor.requires_grad or op.r.tensor.root.requires_grad):
                    rt = op.r.tensor
                    rb = bn_of.get(id(rt)) if op.r.bn is not None else None
                    fresh = (alias_ok and getattr(op.r, 'view_grad', None) is None and rt.C == out.C
                             and rt.id not in self._written and (rt.base is None or rt.base.id not in self._written))
                    if fresh and rb is not None and readers.get(rb) == {op}:
                        self._galias[rt.id] = (gt, ldt)        # the other branch is a BatchNorm too (Xception shortcut)
                        self._written.add(rt.id)
                    elif (fresh and (op.r.is_plain or rb is not None) and rt.base is None and out.base is None and rt.ld == out.ld
                          and rt.id in self.grad and out.id in self.grad
                          and self.grad[rt.id].numel() == self.grad[out.id].numel()):
                        # the identity branch: d/d(r) starts as d/d(out) and only ever gains contributions that are issued
                        # after every reader of d/d(out) (the producer of r precedes this Add's other branch): the two
                        # gradients share one buffer from here on, no copy
                        self.grad[rt.id] = self.grad[out.id]
                        self._written.add(rt.id)
                    else:
                        gp, ldg, keyt = self._gbuf(op.r)
                        P.k(L.scale_mask_bwd_bf16 if self.bf16 else L.scale_mask_bwd, gt, ldt, 0.0, 0, None, gp, ldg,
                            self._acc(keyt), M, out.C)
            elif k == 'se_mul':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                s_ptr, lds, _, _, sact = self.vargs(op.s)
                gp, ldg, keyx = self._gbuf(op.x)
                gsp, ldgs, keys = self._gbuf(op.s)
                assert self._acc(keys) == 0, 'SE scale gradient has a single producer'
                P.k(L.scale_bcast_bwd_bf16 if self.bf16 else L.scale_bcast_bwd, self.tptr(out, True), out.ld, xp, ldx, sp, hp,
                    act, s_ptr, lds, sact, gp, ldg,
                    self._acc(keyx), gsp, ldgs, N, out.H * out.W, out.C, self.pool_ws.data_ptr(), self.pool_wsb)
            elif k == 'maxpool':
                xp, ldx, sp, hp, act = self.vargs(op.x)
                xt = op.x.tensor
                gp, ldg, keyt = self._gbuf(op.x)
                if self.bf16:
                    P.k(L.maxpool2d_bwd_bf16, self.tptr(out, True), out.ld, self._pool_arg[op].data_ptr(), gp, ldg, self._acc(keyt),
                        N, xt.H, xt.W, xt.C, op.k, op.stride, op.pad_t, op.pad_l, op.Ho, op.Wo)
                else:
                    P.k(L.maxpool2d_bwd, xp, ldx, sp, hp, act, self.tptr(out, True), out.ld, self._pool_arg[op].data_ptr(), gp,
                        ldg, self._acc(keyt), N, xt.H, xt.W, xt.C, op.k, op.stride, op.pad_t, op.pad_l, op.Ho, op.Wo)
            elif k == 'gap':
                xt = op.x.tensor
                gp, ldg, keyt = self._gbuf(op.x)
                P.k(L.global_avgpool_bwd_bf16 if self.bf16 else L.global_avgpool_bwd, self.tptr(out, True), out.ld, gp, ldg,
                    self._acc(keyt), N, xt.H * xt.W, xt.C)
            elif k == 'broadcast':
                # gradient w.r.t. the (lazy) 1x1 value = sum over the pixels it was broadcast to
                xt = op.x.tensor
                assert self._acc(xt) == 0
                P.k(L.global_avgpool_fwd_bf16 if self.bf16 else L.global_avgpool_fwd, self.tptr(out, True), out.ld, None, None,
                    ACT_NONE, self.tptr(xt, True), xt.ld, float(out.H * out.W), N, out.H * out.W, out.C,
                    self.pool_ws.data_ptr(), self.pool_wsb)
            elif k == 'resize':
                xt = op.x.tensor
                P.k(L.resize_bilinear_bwd_bf16 if self.bf16 else L.resize_bilinear_bwd, self.tptr(out, True), out.ld,
                    self.tptr(xt, True), xt.ld, self._acc(xt), N, xt.H, xt.W, xt.C, out.H, out.W)
            else:
                raise NotImplementedError(k)
        if self._bwd_pending:
            self._flush_bn_backward(P)
        self._flush_deferred(P)
        self._reduce_pending(P)
        if self.dist is not None:
            P.coll(lambda hi=self._first_bucket_hi: self.dist.all_reduce_async(G[0:hi]))
            # join the side stream inside this plan: a captured graph may not end with forked work in flight
            P.py(self.dist.wait_all)
        return P

    def _irb_region(self, op):
        nb = self._slab_bytes(op)
        ptr = self.slab_ws.data_ptr() + 4 * self._slab_off[0]
        self._slab_off[0] += ((nb + 255) // 256 * 256) // 4
        return ptr, nb

    def _irb_backward(self, P, rec, fuse, fuse_add, bn_done, batch):
        """backward of a fused inverted-residual block, issued where the loop reaches its depthwise conv: pass A (depthwise kernel
        gradient + the expand BatchNorm's backward sums), that BatchNorm's finalize, pass B (expand kernel gradient + gradient of
        the block input, with the backward sums of a BatchNorm in front of the block where the unfused path would carry them).
        Under SyncBatchNorm pass B waits for the all-reduce of the sums: it is issued from _flush_bn_backward, which the loop
        triggers when it reaches the expand conv"""
        e, b, d = rec
        L, st = self.L, self.store
        if not batch:
            raise RuntimeError('fused inverted-residual blocks leave their weight gradients as slabs (DL3P_BATCHED_WGRAD=0 and a '
                               'fused block in the same executor: _find_irb should not have fused)')
        aux = self.bn_aux[b.bn]
        dz, lddz = self.tptr(d.out, True), d.out.ld
        ctx = P.ctx
        rg, nb = self._irb_region(d)
        rows_a = ctypes.c_int(0)
        P.k(L.irb_bwd_sums, *self._irb_args(rec), aux['mean'].data_ptr(), aux['invstd'].data_ptr(), st.ptr(d.w), dz, lddz, rg, nb,
            ctypes.byref(rows_a), self.partials.data_ptr(), *self._irb_geo(rec), tag=d.name)
        self._jobs.append((rg, st.ptr(d.w, st.G), rows_a.value, d.k * d.k * d.c))
        P.ctx = _op_label(b)
        self._bn_backward(P, b, fused_rows=rows_a.value)
        bn_done.add(b)
        if not self.sync_bn:
            self._irb_pass_b(P, rec)
        P.ctx = ctx

    def _irb_pass_b(self, P, rec):
        e, b, d = rec
        L, st = self.L, self.store
        fuse, fuse_add, bn_done = self._bwd_ctx
        aux = self.bn_aux[b.bn]
        mean, invstd, coef = aux['mean'].data_ptr(), aux['invstd'].data_ptr(), aux['coef'].data_ptr()
        dz, lddz = self.tptr(d.out, True), d.out.ld
        ctx = P.ctx
        P.ctx = _op_label(e)
        xt = e.x.tensor
        need_gx = xt.requires_grad or xt.root.requires_grad
        gp = ldg = None
        acc = 0
        front = front_add = None
        if need_gx:
            gp, ldg, keyt = self._gbuf(e.x)
            acc = self._acc(keyt)
            front = fuse.get(e) if (e in fuse and fuse[e].z.requires_grad) else None
            front_add = fuse_add.get(e) if (front is None and e in fuse_add and fuse_add[e].z.requires_grad) else None
        bn_front = front or front_add
        fargs = (None, 0, None, None, ACT_NONE, None, None, None)
        if bn_front is not None:
            bnf = bn_front.bn
            auxf = self.bn_aux[bnf]
            part_f = self.partials if front is not None else self.partials2
            fargs = (self.tptr(bn_front.z), bn_front.z.ld, self.gscale[bnf.group.id].data_ptr() + 4 * bnf.offset,
                     self.gshift[bnf.group.id].data_ptr() + 4 * bnf.offset, bnf.act, auxf['mean'].data_ptr(),
                     auxf['invstd'].data_ptr(), part_f.data_ptr())
        rg, nb = self._irb_region(e)
        rows_b = ctypes.c_int(0)
        P.k(L.irb_bwd_data, *self._irb_args(rec), mean, invstd, coef, st.ptr(d.w), dz, lddz, rg, nb, ctypes.byref(rows_b), gp,
            ldg or 0, acc, *fargs, *self._irb_geo(rec), tag='pw:' + e.name)
        self._jobs.append((rg, st.ptr(e.w, st.G), rows_b.value, e.cin * e.cout))
        if front is not None:
            P.ctx = _op_label(front)
            self._bn_backward(P, front, fused_rows=rows_b.value)
            bn_done.add(front)
        elif front_add is not None:
            self._presums[front_add] = rows_b.value
        P.ctx = ctx

    def _conv_backward_bf16(self, P, op, wgrad, ws, wsb, wgrad_slabs=None, fused_bn=None):
        """weight and data gradient of one conv on the bf16 path; `fused_bn`: the 'bn' op whose backward sums the pointwise
        data gradient carries (its finalize + apply are issued right behind it)"""
        L, N, st, k = self.L, self.N, self.store, op.kind
        G, out = st.G, op.out
        xp, ldx, sp, hp, act = self.vargs(op.x)
        xt = op.x.tensor
        dz, lddz, dzf = self.tptr(out, True), out.ld, self._is_f32(out)
        need_gx = xt.requires_grad or xt.root.requires_grad
        M = N * op.Ho * op.Wo
        if op.layer.trainable:
            gw = st.ptr(op.w, G)
            gb = st.ptr(op.b, G) if getattr(op, 'b', None) else None
            nb = self._slab_bytes(op) if wgrad_slabs else 0
            if nb and k == 'conv_dw':
                wgrad_slabs(L.dwconv2d_bwd_weight_slabs_bf16, op.k * op.k * op.c, gw, nb, (xp, ldx, sp, hp, act, dz, lddz),
                            (N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo))
            elif nb and k == 'conv_pw':
                wgrad_slabs(L.pwconv_bwd_weight_slabs_bf16, op.cin * op.cout, gw, nb, (xp, ldx, sp, hp, act, dz, lddz, dzf),
                            (M, op.cin, op.cout))
            elif nb:
                wgrad_slabs(L.pwconv_bwd_weight_slabs_bf16, op.kp * op.cout, gw, nb,
                            (self.tptr(op.col), op.col.ld, None, None, ACT_NONE, dz, lddz, dzf), (M, op.kp, op.cout))
            elif k == 'conv_pw':
                wgrad(L.pwconv_bwd_weight_bf16, xp, ldx, sp, hp, act, dz, lddz, dzf, gw, gb, ws, wsb, M, op.cin, op.cout)
            elif k == 'conv_dw':
                wgrad(L.dwconv2d_bwd_weight_bf16, xp, ldx, sp, hp, act, dz, lddz, gw, ws, wsb, N, xt.H, xt.W, op.c, op.k,
                      op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
            else:
                wgrad(L.pwconv_bwd_weight_bf16, self.tptr(op.col), op.col.ld, None, None, ACT_NONE, dz, lddz, dzf, gw, gb,
                      ws, wsb, M, op.kp, op.cout)
        if not need_gx:
            return
        gp, ldg, keyt = self._gbuf(op.x)
        acc = self._acc(keyt)
        self._conv_dgrad_bf16(P, op, fused_bn, dz, lddz, dzf, gp, ldg, acc, M)

    def _conv_dgrad_bf16(self, P, op, fused_bn, dz, lddz, dzf, gp, ldg, acc, M):
        L, N, st, k = self.L, self.N, self.store, op.kind
        xt = op.x.tensor
        if k == 'conv_pw' and fused_bn is not None and not dzf:
            # the BatchNorm-backward sums of the BatchNorm behind this gradient ride on the data gradient (dl3p_pwconv_bwd_data_bn)
            bn = fused_bn.bn
            aux = self.bn_aux[bn]
            rows = ctypes.c_int(0)
            P.k(L.pwconv_bwd_data_bn_bf16, dz, lddz, st.ptr(op.w, st.Pb), gp, ldg, acc, M, op.cin, op.cout,
                self.tptr(fused_bn.z), fused_bn.z.ld, self.gscale[bn.group.id].data_ptr() + 4 * bn.offset,
                self.gshift[bn.group.id].data_ptr() + 4 * bn.offset, bn.act, aux['mean'].data_ptr(),
                aux['invstd'].data_ptr(), self.partials.data_ptr(), ctypes.byref(rows))
            ctx = P.ctx
            P.ctx = _op_label(fused_bn)
            self._bn_backward(P, fused_bn, fused_rows=rows.value)
            P.ctx = ctx
        elif k == 'conv_pw':
            if fused_bn is not None:          # (an fp32 gradient operand: the kernel with the sums takes bf16 only)
                ctx = P.ctx
                P.k(L.pwconv_bwd_data_bf16, dz, lddz, dzf, st.ptr(op.w, st.Pb), gp, ldg, acc, M, op.cin, op.cout)
                P.ctx = _op_label(fused_bn)
                self._bn_backward(P, fused_bn)
                P.ctx = ctx
                return
            P.k(L.pwconv_bwd_data_bf16, dz, lddz, dzf, st.ptr(op.w, st.Pb), gp, ldg, acc, M, op.cin, op.cout)
        elif k == 'conv_dw':
            P.k(L.dwconv2d_bwd_data_bf16, dz, lddz, st.ptr(op.w, st.Pb), gp, ldg, acc, N, xt.H, xt.W, op.c, op.k, op.stride,
                op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo)
        else:
            # dense k x k conv (Xception's entry_flow_conv1_2, ResNet50's 3x3 convs): d/d(im2col matrix) by the GEMM, then the
            # transposed gather back onto the input pixels -- the route of the forward (im2col_bf16 + GEMM), mirrored
            P.k(L.pwconv_bwd_data_bf16, dz, lddz, dzf, st.ptr(op.w, st.Pb), self.tptr(op.col, True), op.col.ld, 0, M, op.kp,
                op.cout)
            P.k(L.col2im_bf16, self.tptr(op.col, True), op.col.ld, gp, ldg, acc, N, xt.H, xt.W, op.cin, op.k, op.stride, op.rate,
                op.pad_t, op.pad_l, op.Ho, op.Wo)

    def _use_sb(self, op, fwd, stats):
        """does this pointwise conv run on the split-bf16 GEMM (forward / data-gradient role)?  Only where the tiled kernel
        serves the shape and the product is compute-bound enough for the bf16 pipe to pay: reduction length and output width from
        DL3P_SPLIT_MIN_K / DL3P_SPLIT_MIN_N (measured: scripts/micro/sb_gemm.py)"""
        st = self.store
        if st.Sb is None or op not in st.sb_shape_f:
            return False
        M = self.N * op.Ho * op.Wo
        kred, nout = (op.cin, op.cout) if fwd else (op.cout, op.cin)
        # where it pays (profiles/r03_split_gemm.txt): many output tiles and a moderate reduction length.  Few-row layers with a
        # long reduction (Xception's 4356 x 2048 -> 256) and narrow outputs (576 -> 96) are faster on the fp32-input MFMA kernel;
        # the data gradient with the fused BatchNorm sums only wins on the long decoder layers.
        env = os.environ.get
        role = (1 if stats else 0) if fwd else (3 if stats else 2)
        pinned = any(k in os.environ for k in ('DL3P_SPLIT_MIN_K', 'DL3P_SPLIT_MIN_N', 'DL3P_SPLIT_MIN_ROWS', 'DL3P_SPLIT_MIN_ROWS_BN'))
        # the measured verdict for this exact launch where there is one (csrc/sb_tuned.h, scripts/tune_split.py), else the rule
        pays = -1 if pinned else self.L.pwconv_sb_pays(role, M, kred, nout)
        if pays == 0:
            return False
        if pays < 0:
            if kred < int(env('DL3P_SPLIT_MIN_K', '128')) or nout < int(env('DL3P_SPLIT_MIN_N', '128')):
                return False
            if M < int(env('DL3P_SPLIT_MIN_ROWS', '16384')) or (not fwd and stats and M < int(env('DL3P_SPLIT_MIN_ROWS_BN', '60000'))):
                return False
        if not self.L.pwconv_sb_supported(role, M, kred, nout):
            return False
        (self._sb_used_f if fwd else self._sb_used_b).add(op)       # the optimiser step refreshes these planes (_trace_sgd)
        return True

    def _use_sb_dense(self, op, role):
        """does this dense conv's forward (role 0 / 1: without / with BatchNorm statistics) or data gradient (2) run on the split
        kernel?  The measured rule of the library (dl3p_conv2d_gemm_sb_pays); the weight gradient routes itself inside
        dl3p_conv2d_gemm_bwd_weight*"""
        st = self.store
        if st.Sb is None or op not in st.sb_shape_f:
            return False
        xt = op.x.tensor
        if role == 2:
            M, K, Nn = self.N * xt.H * xt.W, op.k * op.k * op.cout, op.cin
        else:
            M, K, Nn = self.N * op.Ho * op.Wo, op.k * op.k * op.cin, op.cout
        if not self.L.conv2d_gemm_sb_pays(role, M, K, Nn):
            return False
        if role != 2:
            self._sb_used_f.add(op)          # the optimiser step refreshes this plane (_trace_sgd)
        return True

    def _pw_dgrad_bn(self, P, op, dz, lddz, gp, ldg, acc, bn_op, partials, rows):
        """data gradient of a pointwise conv + the BatchNorm-backward partial sums of the BatchNorm in front of it"""
        st, L, bn = self.store, self.L, bn_op.bn
        aux = self.bn_aux[bn]
        M = self.N * op.Ho * op.Wo
        bnargs = (self.tptr(bn_op.z), bn_op.z.ld, self.gscale[bn.group.id].data_ptr() + 4 * bn.offset,
                  self.gshift[bn.group.id].data_ptr() + 4 * bn.offset, bn.act, aux['mean'].data_ptr(), aux['invstd'].data_ptr(),
                  partials.data_ptr(), ctypes.byref(rows))
        if self._use_sb(op, False, True):
            wsp, pitch = st.sb_ptr(op, False)
            P.k(L.pwconv_bwd_data_sb, dz, lddz, wsp, pitch, gp, ldg, acc, M, op.cin, op.cout, *bnargs)
        else:
            P.k(L.pwconv_bwd_data_bn, dz, lddz, st.ptr(op.w), gp, ldg, acc, M, op.cin, op.cout, *bnargs)

    def _bn_fusion_map(self):
        """{consumer conv op: 'bn' op} for every trainable BatchNorm whose output value is read FIRST (in graph order) by a
        pointwise or depthwise conv: its data-gradient kernel is then the last writer of the gradient
        of the BN output and emits the BN-backward partial sums itself (dl3p_pwconv_bwd_data_bn / dl3p_dwconv2d_bwd_data_bn); the BN's
        finalize + apply are issued right behind it and the separate reduce pass over (g, z) disappears."""
        if os.environ.get('DL3P_FUSE_BN_BWD', '1') == '0':
            return {}
        cons = {}
        for op in self.g.ops:
            for slot in ('x', 'r', 's'):
                v = getattr(op, slot, None)
                if v is not None and getattr(v, 'bn', None) is not None:
                    cons.setdefault(v.bn, []).append((op, slot, v))
        bn_ops = {op.bn: op for op in self.g.ops if op.kind == 'bn'}
        fuse = {}
        for bn, lst in cons.items():
            if bn not in bn_ops or not bn.layer.trainable:
                continue
            # the FIRST consumer in graph order is the last to run in backward: every other reader of the BatchNorm output
            # (a residual Add's identity input, ...) has added its share to the gradient buffer by then
            op, slot, v = lst[0]
            if any(getattr(vv, 'view_grad', None) is not None for _, _, vv in lst):
                continue
            if len(lst) > 1 and os.environ.get('DL3P_FUSE_BN_BWD', '1') == '1single':
                continue
            if slot == 'x' and op.kind in ('conv_pw', 'conv_dw') and v.tensor is bn.z and op not in fuse:
                fuse[op] = bn_ops[bn]
        return fuse

    def _bn_fusion_through_adds(self, readers):
        """{pointwise conv op: 'bn' op} for BatchNorm -> Add (BatchNorm branch + another branch, no dropout) -> conv, where the
        conv is the FIRST consumer of the Add's output in graph order (so its data gradient is issued last in backward and
        completes d/d(Add output) = d/d(BatchNorm output)), reads it plain, and the Add is the BatchNorm's only reader"""
        if os.environ.get('DL3P_FUSE_BN_BWD', '1') == '0':
            return {}
        bn_of = {id(o.z): o for o in self.g.ops if o.kind == 'bn'}
        first = {}                        # tensor id -> first op (graph order) that reads it
        for op in self.g.ops:
            for slot in ('x', 'r', 's'):
                v = getattr(op, slot, None)
                if v is not None:
                    first.setdefault(v.tensor.id, op)
                    if v.tensor.base is not None:
                        first.setdefault(v.tensor.base.id, op)
        out = {}
        for add in self.g.ops:
            if add.kind != 'materialize' or add.r is None or float(add.rate) != 0.0 or add.x.bn is None:
                continue
            bn_op = bn_of.get(id(add.x.tensor))
            if (bn_op is None or not bn_op.bn.layer.trainable or readers.get(bn_op) != {add} or add.out.base is not None
                    or getattr(add.x, 'view_grad', None) is not None or add.x.tensor.C != add.out.C):
                continue
            conv = first.get(add.out.id)
            if (conv is not None and conv.kind == 'conv_pw' and conv.x.tensor is add.out and conv.x.is_plain
                    and getattr(conv.x, 'view_grad', None) is None and conv not in out):
                out[conv] = bn_op
        return out

    def _flush_deferred(self, P):
        ctx = P.ctx
        for fn, args, c in self._deferred:
            P.ctx = c
            if args is None:
                fn(P)                 # a slab-leaving weight gradient: launches and files its reduction job
            else:
                P.k(fn, *args)
        self._deferred = []
        P.ctx = ctx

    def _reduce_pending(self, P):
        if getattr(self, '_jobs', None):
            ctx = P.ctx
            P.ctx = 'wgrad:reduce_all'
            self._reduce_all(P, self._jobs)
            del self._jobs[:]
            P.ctx = ctx

    def _bucket_edges(self):
        edges, self._first_bucket_hi = bucket_edges(self.g, self.store.offset, self.store.total, self.dist.n_buckets)
        return edges

    def _bn_backward(self, P, op, fused_rows=None):
        bn, L, st, N = op.bn, self.L, self.store, self.N
        aux = self.bn_aux[bn]
        lp = {p.key: p for p in bn.layer.params}
        z = op.z
        M = N * z.H * z.W
        sp = self.gscale[bn.group.id].data_ptr() + 4 * bn.offset
        hp = self.gshift[bn.group.id].data_ptr() + 4 * bn.offset
        if op in self._irb_bn:              # (a fused block's expand output and its gradient have no buffers: sums in, triple out)
            g = ldg = dzo = lddzo = zp = ldz = None
        else:
            g, ldg = self.tptr(z, True), z.ld
            dzo, lddzo = g, ldg             # where dz goes (g itself unless the gradient is read from another buffer)
            if z.id in getattr(self, '_galias', {}):
                g, ldg = self._galias.pop(z.id)
            zp, ldz = self.tptr(z), z.ld
        mean, invstd, coef = aux['mean'].data_ptr(), aux['invstd'].data_ptr(), aux['coef'].data_ptr()
        frozen = not bn.layer.trainable
        G = st.G
        if frozen:
            P.k(L.bn_bwd_finalize, None, 0, None, bn.C, float(M), st.ptr(lp['gamma']), invstd, sp, 1, None, None, coef)
        else:
            part = self.partials.data_ptr()
            if fused_rows is None and op in getattr(self, '_presums', {}):
                fused_rows, part = self._presums.pop(op), self.partials2.data_ptr()
            rows = ctypes.c_int(fused_rows or 0)
            if fused_rows is None:      # (otherwise the producer of g already left the partial sums)
                P.k(L.bn_bwd_reduce_bf16 if self.bf16 else L.bn_bwd_reduce, g, ldg, zp, ldz, sp, hp, bn.act, mean, invstd,
                    part, ctypes.byref(rows), M, bn.C)
            P.k(L.bn_bwd_finalize, part, rows.value, None, bn.C, float(M), st.ptr(lp['gamma']),
                invstd, sp, 0, st.ptr(lp['gamma'], G), st.ptr(lp['beta'], G), coef)
            if self.sync_bn:
                # parameter gradients stay local (they are averaged with every other gradient); the normalisation terms
                # use the global sums: local sums into this BatchNorm's slice of the staging buffer, all-reduce +
                # finalize + apply when the producer of z is reached (_flush_bn_backward), together with every other
                # BatchNorm pending by then
                off = self._bwd_stage_off
                self._bwd_stage_off += 2 * bn.C
                P.k(L.bn_reduce_partials, part, rows.value, 2 * bn.C, self.sync_stage[off:].data_ptr())
                self._bwd_pending.append((op, off, P.ctx))
                self._sync_g[op] = (g, ldg)
                return
        if not frozen and op in self._irb_bn:
            return                          # dl3p_irb_bwd_data forms dz from the coefficient triple while it recomputes the expand conv
        if not frozen and self._folds_apply_dgrad(op):
            # the conv that produced z forms dz while its DATA gradient stages its operand, writes it where the apply pass would
            # have, and its weight gradient (issued behind the data gradient for that) reads it there
            self._folded_dg[z.id] = (g, ldg, zp, ldz, sp, hp, bn.act, mean, invstd, coef, dzo, lddzo)
            return
        if not frozen and self._folds_apply(op):
            # the conv that produced z forms dz inside its weight-gradient kernel and hands it to its data gradient
            self._folded[z.id] = (g, ldg, zp, ldz, sp, hp, bn.act, mean, invstd, coef)
            self._dz_not_kept.add(z.root.id)
            return
        P.k(L.bn_bwd_apply_bf16 if self.bf16 else L.bn_bwd_apply, g, ldg, zp, ldz, sp, hp, bn.act, mean, invstd, coef, dzo,
            lddzo, 0, M, bn.C)

    def _folds_apply_dgrad(self, bn_op):
        """BatchNorm-backward apply folded into the staged operand of the DATA gradient of the pointwise conv that produced z
        (dl3p_pwconv_bwd_data_sb_apply: the long decoder layers on the row-stationary split GEMM; fp32, local statistics)"""
        conv = getattr(bn_op, 'producer', None)
        if (self.bf16 or self.sync_bn or self.dist is not None or conv is None or conv.kind != 'conv_pw' or conv.out is not bn_op.z
                or os.environ.get('DL3P_FOLD_APPLY', '1') == '0' or os.environ.get('DL3P_FOLD_APPLY_DGRAD', '1') == '0'
                or conv in self._irb_expand):
            return False
        xt = conv.x.tensor
        if not (xt.requires_grad or xt.root.requires_grad):
            return False
        M = self.N * conv.Ho * conv.Wo
        # the launch this decides about (_trace_backward, `out.id in self._folded_dg`): with the sums of a BatchNorm in front of the
        # conv (role 3) or without (role 2), the gradient written at the pitch of the buffer that collects d/d(conv input)
        fuse, fuse_add, _ = self._bwd_ctx
        front = fuse.get(conv) if (conv in fuse and fuse[conv].z.requires_grad) else None
        if front is None and conv in fuse_add and fuse_add[conv].z.requires_grad:
            front = fuse_add[conv]
        vt = getattr(conv.x, 'view_grad', None)
        ldg = vt.ld if vt is not None else xt.ld
        if M * max(xt.ld, ldg, bn_op.z.ld, conv.cout, front.z.ld if front is not None else 0) * 4 >= 2 ** 32:
            return False
        if not self.L.pwconv_bwd_data_sb_apply_supported(M, conv.cin, conv.cout, bn_op.bn.act, 1 if front is not None else 0):
            return False
        return self._use_sb(conv, False, front is not None)

    def _folds_apply(self, bn_op):
        """BatchNorm-backward apply folded into the weight gradient of the pointwise conv that produced z (fp32, local
        statistics, weight gradient issued in line in front of the data gradient)"""
        conv = getattr(bn_op, 'producer', None)
        if (self.bf16 or self.sync_bn or self.dist is not None or conv is None or conv.kind not in ('conv_pw', 'conv_dw', 'conv_dense')
                or conv.out is not bn_op.z or not conv.layer.trainable or not getattr(self, '_batch_wgrad', False)
                or os.environ.get('DL3P_FOLD_APPLY', '1') == '0' or not self._slab_bytes(conv)):
            return False
        if conv in self._irb_dw or conv in self._irb_expand:
            return False
        if conv.kind == 'conv_dense':
            # the direct stem kernel (3 input channels, stride 2) whose input needs no gradient: dl3p_stem_conv_bwd_weight_slabs_bn
            xt = conv.x.tensor
            return bool(self._stem_direct(conv) and not (xt.requires_grad or xt.root.requires_grad)
                        and os.environ.get('DL3P_FOLD_APPLY_STEM', '1') != '0')
        M = self.N * conv.Ho * conv.Wo
        if conv.kind == 'conv_dw':
            # measured on MI355X: the depthwise window kernel with the fold takes as much longer as the apply pass it
            # replaces took (MobileNetV2 step 14.12 ms against 14.05 with the pointwise folds alone, 14.25 without any)
            # ... on every depthwise conv.  On the two 129 x 129 decoder layers alone (324 / 272 MB tensors: the apply pass is
            # 174 / 136 us of pure traffic) the folded weight gradient is 319 / 236 us against 329 / 264 for the pair it replaces
            # (round 5, same box): those two take it; DL3P_FOLD_APPLY=2 folds every depthwise conv
            big = M * conv.c * 4 >= int(os.environ.get('DL3P_FOLD_APPLY_DW_MIN_BYTES', str(250 << 20)))
            if os.environ.get('DL3P_FOLD_APPLY', '1') != '2' and not big:
                return False
            xt = conv.x.tensor
            return bool(self.L.dwconv2d_bwd_weight_bn_supported(self.N, xt.H, xt.W, conv.c, conv.k, conv.stride, conv.rate,
                                                                conv.pad_t, conv.pad_l, conv.Ho, conv.Wo))
        if M * max(conv.x.tensor.ld, bn_op.z.ld, conv.cout) * 4 >= 2 ** 32:
            return False
        return bool(self.L.pwconv_bwd_weight_bn_supported(M, conv.cin, conv.cout))

    def _bn_readers(self):
        """{'bn' op: set of ops that read the BatchNorm's output} (a Concatenate's consumer reads every branch)"""
        out = {}
        by_t = {}
        for op in self.g.ops:
            for slot in ('x', 'r', 's'):
                v = getattr(op, slot, None)
                if v is not None:
                    by_t.setdefault(v.tensor.id, set()).add(op)
        for op in self.g.ops:
            if op.kind == 'bn':
                z = op.z
                rd = set(by_t.get(z.id, ()))
                if z.base is not None:
                    rd |= by_t.get(z.base.id, set())
                out[op] = rd
        return out

    def _flush_bn_backward(self, P):
        """ONE all-reduce (on the side stream, beside the deferred weight gradients) for every pending BatchNorm, then
        their finalize (global sums -> coefficients) and apply kernels"""
        pend, self._bwd_pending = self._bwd_pending, []
        lo = pend[0][1]
        hi = pend[-1][1] + 2 * pend[-1][0].bn.C
        assert all(b[1] - a[1] == 2 * a[0].bn.C for a, b in zip(pend, pend[1:])), 'backward SyncBatchNorm slices are contiguous'
        L, st = self.L, self.store
        ctx = P.ctx
        P.ctx = 'syncbn:' + '+'.join(b.bn.name for b, _, _ in pend)
        P.coll(lambda t=self.sync_stage[lo:hi]: self.dist.bn_all_reduce_begin(t))
        self._flush_deferred(P)
        P.py(self.dist.bn_all_reduce_end)
        for op, off, c in pend:
            bn = op.bn
            P.ctx = c
            aux = self.bn_aux[bn]
            lp = {p.key: p for p in bn.layer.params}
            z = op.z
            M = self.N * z.H * z.W
            sp = self.gscale[bn.group.id].data_ptr() + 4 * bn.offset
            hp = self.gshift[bn.group.id].data_ptr() + 4 * bn.offset
            mean, invstd, coef = aux['mean'].data_ptr(), aux['invstd'].data_ptr(), aux['coef'].data_ptr()
            if op in self._irb_bn:
                # a fused block: the global coefficient triple, then its second pass (which forms dz while it recomputes the expand)
                self._sync_g.pop(op, None)
                P.k(L.bn_bwd_finalize, None, 0, self.sync_stage[off:].data_ptr(), bn.C, float(M * self.dist.world_size),
                    st.ptr(lp['gamma']), invstd, sp, 0, None, None, coef)
                self._irb_pass_b(P, self._irb_bn[op])
                continue
            dzo, lddzo, zp = self.tptr(z, True), z.ld, self.tptr(z)
            g, ldg = self._sync_g.pop(op, (dzo, lddzo))          # (a residual Add's buffer when the gradient was handed on in place)
            P.k(L.bn_bwd_finalize, None, 0, self.sync_stage[off:].data_ptr(), bn.C, float(M * self.dist.world_size),
                st.ptr(lp['gamma']), invstd, sp, 0, None, None, coef)
            P.k(L.bn_bwd_apply_bf16 if self.bf16 else L.bn_bwd_apply, g, ldg, zp, z.ld, sp, hp, bn.act, mean, invstd, coef,
                dzo, lddzo, 0, M, bn.C)
        P.ctx = ctx

    # ---------------------------------------------------------------- optimiser
    def _trace_sgd(self):
        P, L, st = Plan(), self.L, self.store
        scale = 1.0
        if self.dist is not None:
            scale = 1.0 / self.dist.world_size
        kind = self.optimizer[0]
        if kind == 'adam':
            _, b1, b2, eps = self.optimizer
            P.k(L.adam_step, st.P.data_ptr(), st.V.data_ptr(), st.V2.data_ptr(), st.G.data_ptr(), st.total,
                self.lr.data_ptr(), st.opt_step.data_ptr(), b1, b2, eps, scale, st.l2.data_ptr(), st.lr_scale.data_ptr())
        elif kind == 'rmsprop':
            _, rho, eps = self.optimizer
            P.k(L.rmsprop_step, st.P.data_ptr(), st.V.data_ptr(), st.G.data_ptr(), st.total, self.lr.data_ptr(), rho, eps,
                scale, st.l2.data_ptr(), st.lr_scale.data_ptr())
        else:
            P.k(L.sgd_momentum, st.P.data_ptr(), st.V.data_ptr(), st.G.data_ptr(), st.total, self.lr.data_ptr(),
                float(self.momentum), 0.0, scale, st.l2.data_ptr(), st.lr_scale.data_ptr())
        if self.bf16:                       # refresh the bf16 mirrors the next forward / backward read
            P.k(L.f32_to_bf16, st.P.data_ptr(), st.Pb.data_ptr(), st.total)
            if st.tr_table is not None:
                P.k(L.transpose_batch_bf16, st.P.data_ptr(), st.Pbt.data_ptr(), st.tr_table.data_ptr(), int(st.tr_table.shape[0]))
        elif st.tr_table is not None:      # the forward GEMMs read the transposed kernel copies
            P.k(L.transpose_batch, st.P.data_ptr(), st.Pt.data_ptr(), st.tr_table.data_ptr(), int(st.tr_table.shape[0]))
        if st.Sb is not None:               # ... and the split-bf16 GEMMs of THIS executor the pre-split planes they read
            order = [op for op in self.g.ops if op.kind in ('conv_pw', 'conv_dense')]
            rf = [st.sb_row_f[op] for op in order if op in self._sb_used_f and op in st.sb_row_f]
            rb = [st.sb_row_b[op] for op in order if op in self._sb_used_b and op in st.sb_row_b]
            self._sb_tab_f = torch.tensor(rf, dtype=torch.int64, device=self.dev) if rf else None
            self._sb_tab_b = torch.tensor(rb, dtype=torch.int64, device=self.dev) if rb else None
            if rf:
                P.k(L.split_bf16x3_batch, st.Pt.data_ptr(), st.Sb.data_ptr(), self._sb_tab_f.data_ptr(), len(rf))
            if rb:
                P.k(L.split_bf16x3_batch, st.P.data_ptr(), st.Sb.data_ptr(), self._sb_tab_b.data_ptr(), len(rb))
        return P

    # ---------------------------------------------------------------- running
    def _stage_u8(self, src, slot):
        src = torch.as_tensor(src).reshape(-1)
        if src.is_cuda:                 # (a batch staged ahead by model.BatchFeeder: already on the device)
            return src
        stage = self._u8.get(slot)
        if stage is None or stage.numel() != src.numel():
            stage = self._u8[slot] = torch.empty(src.numel(), dtype=torch.uint8, device=self.dev)
        stage.copy_(src, non_blocking=True)
        return stage

    def _upload_u8(self, src, dst, div, sub, slot):
        """bytes over PCIe, float32 (or bf16 activations) on the device (dl3p_u8_to_float / dl3p_u8_to_bf16)"""
        stage = self._stage_u8(src, slot)
        fn = self.L.u8_to_bf16 if dst.dtype == torch.bfloat16 else self.L.u8_to_float
        fn(stage.data_ptr(), dst.data_ptr(), stage.numel(), div, sub, torch.cuda.current_stream().cuda_stream)

    def set_inputs(self, x, y=None, sample_weight=None):
        """x float32 in [-1, 1] -- or uint8 pixels, normalised on the device like normalize_image does on the host
        (common/data_utils.py:403-417); y float / integer class ids -- or uint8, which then also get the generator's
        `label > num_classes-1 -> ignore_index` (deeplabv3p/data.py:116-121); sample_weight (N, H*W) or 'adaptive':
        the generator's balanced per-image class weights (data.py:134-145) computed on the device from uint8 labels"""
        inp = self.buf[self.g.input.tensor.id]
        if getattr(x, 'dtype', None) in (np.uint8, torch.uint8):
            self._upload_u8(x, inp, 127.5, 1.0, 'x')
        else:
            x = torch.as_tensor(x, dtype=torch.float32)
            if self.bf16:
                # float images over PCIe as they are, rounded to bf16 on the device (dl3p_f32_to_bf16)
                stage = self._u8.get('xf')
                if stage is None or stage.numel() != inp.numel():
                    stage = self._u8['xf'] = torch.empty(inp.numel(), **self.f32)
                stage.copy_(x.reshape(-1), non_blocking=True)
                self.L.f32_to_bf16(stage.data_ptr(), inp.data_ptr(), stage.numel(), torch.cuda.current_stream().cuda_stream)
            else:
                inp.copy_(x.reshape(-1), non_blocking=True)
        adaptive = isinstance(sample_weight, str)
        if adaptive:
            if sample_weight != 'adaptive':
                raise ValueError("sample_weight must be an array or 'adaptive'")
            if self.pixel_weights is None:
                raise ValueError("sample weights need compile(sample_weight_mode='temporal')")
            if y is None or getattr(y, 'dtype', None) not in (np.uint8, torch.uint8) or self.ignore_index is None:
                raise ValueError("'adaptive' weights are computed on the device from uint8 labels (and an ignore_index)")
        if y is not None:
            if getattr(y, 'dtype', None) in (np.uint8, torch.uint8):
                if self.ignore_index is not None and 0 <= int(self.ignore_index) <= 255 and self.C <= 256:
                    stage = self._stage_u8(y, 'y')
                    if adaptive and 'hist' not in self._u8:
                        self._u8['hist'] = torch.empty(self.N * 256, dtype=torch.int32, device=self.dev)
                    self.L.label_prepare(stage.data_ptr(), self.labels.data_ptr(),
                                         self.pixel_weights.data_ptr() if adaptive else None,
                                         self._u8['hist'].data_ptr() if adaptive else None,
                                         self.N, stage.numel() // self.N, self.C, int(self.ignore_index),
                                         torch.cuda.current_stream().cuda_stream)
                else:
                    self._upload_u8(y, self.labels, 1.0, 0.0, 'y')
            else:
                y = torch.as_tensor(y, dtype=torch.float32)
                self.labels.copy_(y.reshape(-1), non_blocking=True)
        if sample_weight is not None and not adaptive:
            if self.pixel_weights is None:
                raise ValueError("sample weights need compile(sample_weight_mode='temporal')")
            self.pixel_weights.copy_(torch.as_tensor(sample_weight, dtype=torch.float32).reshape(-1), non_blocking=True)

    def capture(self):
        """capture the traced plans into hipGraphs (done once, after a warm-up eager step)"""
        torch.cuda.synchronize()
        self.L.set_option(b'split_wgrad', self._split_wgrad)      # (the capture re-issues every C call)
        # the trace already ran every kernel once; undo its side effects on the step counter only
        in_graph = os.environ.get('DL3P_COLLECTIVES_IN_GRAPH', '1') != '0'     # 0: force the segmented fallback
        for plan in ([self.fwd, self.bwd, self.opt] if self.training else [self.fwd]):
            plan.capture(collectives_in_graph=in_graph)
        self.graphed = True

    def _sb_sync(self):
        """the split planes this executor reads are current: after another executor's optimiser step only ITS subset is
        (ParamStore.sb_partial), so everything is re-split once before this one runs"""
        st = self.store
        if st.Sb is not None and st.sb_partial is not None and st.sb_partial is not self:
            st.transpose()

    def _pin_options(self):
        if not self.graphed:
            self.L.set_option(b'split_wgrad', self._split_wgrad)
            for k, v in self._pinned.items():
                if self.L.get_option(k) != v:
                    self.L.set_option(k, v)
            if self._irb_expand and tuple(self.L.irb_get_plan(i) for i in range(4)) != self._pinned_irb:
                self.L.irb_set_plan(*self._pinned_irb[:2])
                self.L.irb_set_bwd_plan(*self._pinned_irb[2:])

    def train_step(self):
        self._sb_sync()
        self._pin_options()
        self.fwd.run()
        self.bwd.run()
        self.opt.run()
        if self.store.Sb is not None:
            self.store.sb_partial = self

    def forward(self):
        self._sb_sync()
        self._pin_options()
        self.fwd.run()

    def eval_step(self, confusion, pred=None):
        """inference forward WITHOUT the probability tensor, then argmax + confusion-matrix accumulation against
        self.labels on the device (eval.py:33-36, 368-373, 440-443): per batch nothing but the C x C counters changes
        (predict() writes and downloads N*H*W*C probabilities: 354 MB at batch 16).  confusion: int64 [C*C] tensor."""
        assert not self.training
        if getattr(self, '_eval_plan', None) is None:
            P = Plan()
            P.items = list(self.fwd.items[:-1])            # the body; the last item is the softmax head
            P.labels = list(self.fwd.labels[:-1])
            self._eval_plan = P
        self._sb_sync()
        self._eval_plan.run()
        zt = self.head.tensor
        self.L.argmax_confusion(self.tptr(zt), zt.ld, None if confusion is None else self.labels.data_ptr(),
                                None if pred is None else pred.data_ptr(),
                                None if confusion is None else confusion.data_ptr(), self.N, zt.H, zt.W, self.C, self.H,
                                self.W, torch.cuda.current_stream().cuda_stream)

    def eval_loss_step(self, counts=None):
        """validation step on the device: inference forward, the compiled loss against self.labels (no gradient, no
        probability tensor) and, if asked for, the per-image class counts of the Jaccard metric; -> loss tensor [1]"""
        assert not self.training
        if getattr(self, '_eval_plan', None) is None:
            P = Plan()
            P.items = list(self.fwd.items[:-1])
            P.labels = list(self.fwd.labels[:-1])
            self._eval_plan = P
        self._sb_sync()
        self._eval_plan.run()
        zt, L = self.head.tensor, self.L
        st = torch.cuda.current_stream().cuda_stream
        rows = ctypes.c_int(0)
        L.upsample_softmax_loss(self.tptr(zt), zt.ld, self.labels.data_ptr(), int(self.ignore_index or 0),
                                1.0 / float(self.N * self.H * self.W), self.loss_kind,
                                None if self.class_weights is None else self.class_weights.data_ptr(), self.loss_gamma,
                                self.loss_alpha, None, None, None, None, self.cpad, self.loss_partials.data_ptr(),
                                ctypes.byref(rows), self.N, zt.H, zt.W, self.C, self.H, self.W, st)
        L.reduce_rows(self.loss_partials.data_ptr(), rows.value, 1, self.loss.data_ptr(), 0, st)
        if counts is not None:
            L.fill(counts.data_ptr(), 0.0, counts.numel(), st)
            L.class_counts(self.tptr(zt), zt.ld, self.labels.data_ptr(), counts.data_ptr(), self.N, zt.H, zt.W, self.C,
                           self.H, self.W, st)
        return self.loss

    def install_probe(self, name):
        """time one forward depthwise launch with HIP events inside the steps (bench.py roofline)"""
        op = [o for o in self.g.ops if getattr(o, 'name', None) == name and o.kind == 'conv_dw'][0]
        seg = op.rate == 1 and op.stride in (1, 2) and op.Wo >= 4
        lat2 = (op.k == 3 and op.stride == 1 and 2 * op.rate >= max(op.x.tensor.H, op.x.tensor.W) and op.rate < min(op.x.tensor.H, op.x.tensor.W))
        lat3 = (not lat2 and op.k == 3 and op.stride == 1 and 3 * op.rate >= max(op.x.tensor.H, op.x.tensor.W) and op.rate < min(op.x.tensor.H, op.x.tensor.W))
        kname = 'dw_fwd_lattice2' if lat2 else 'dw_fwd_lattice3' if lat3 else (('dw_fwd_seg<%d,%d,%d>' % (op.k, 4 if op.stride == 1 else 2, op.stride)) if seg else 'dw_fwd_gather<%d>' % op.k)
        if self.bf16:
            kname = ('dwb_fwd_strip<%d, %d' if op.stride == 1 else 'dwb_fwd<%d, %d') % (8 if (op.k == 3 and op.c % 8 == 0) else 4, op.k)
            # 3x3 convs the window plan serves run on the bf16 instantiation of the sliding-window kernels (csrc/dw_bf16_window.h)
            plan = (ctypes.c_int * 6)()
            xt = op.x.tensor
            self.L.dw_plan_query(0, self.N, xt.H, xt.W, op.c, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo, plan)
            if op.k == 3 and plan[0] in (1, 2) and os.environ.get('DL3P_BF16_DW_WINDOW', '1') != '0':
                kname = 'dwb_fwd_seg<3, %d, %d' % (plan[1] if plan[0] == 1 else 2, op.stride)
        probe = Probe(op, kname)
        self.fwd.probe(name, probe)
        return probe

    def time_tagged_launch(self, name, reps=20, context_from=None):
        """mean duration (ms) of the forward launch tagged `name`, re-issued `reps` times on the buffers of the last step with
        the library's HIP event pair -- for runs whose forward has to stay one graph (collectives captured at N > 1).
        `context_from`: tag of an earlier launch; the conv / elementwise launches from there up to `name` are re-issued in
        front of every timed launch so that it meets the cache state it has inside the step (its input freshly written by
        its producer, the other readers of that input just gone) instead of a cache warmed by its own previous run."""
        i1 = self.fwd.tags[name]
        i0 = self.fwd.tags[context_from] if context_from in self.fwd.tags else i1
        stateful = ('bn_finalize', 'bn_reduce_partials', 'increment_counter')
        seq = [(fn, args) for (fn, args), (lab, _) in zip(self.fwd.items[i0:i1], self.fwd.labels[i0:i1])
               if fn is not None and not any(s in lab for s in stateful)]
        fn, args = self.fwd.items[i1]
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        ts = []
        for i in range(reps + 2):
            for f2, a2 in seq:
                f2(*a2, st)
            L.probe_arm(3000 + i)
            fn(*args, st)
        torch.cuda.synchronize()
        for i in range(2, reps + 2):
            ms = ctypes.c_float(0)
            L.probe_read(3000 + i, ctypes.addressof(ms))
            ts.append(ms.value)
        return sum(ts) / len(ts)

    def install_pw_probe(self, name):
        """the same for one forward pointwise GEMM (fp32 path): the MFMA roofline entry of the bench line"""
        op = [o for o in self.g.ops if getattr(o, 'name', None) == name and o.kind == 'conv_pw'][0]
        probe = Probe(op, 'pw_gemm_kernel', base=2048)
        self.fwd.probe('pw:' + name, probe)
        return probe

    def dropout_mask(self, op):
        t = op.out
        m = torch.empty(self.N * t.H * t.W * t.C, **self.f32)
        self.L.dropout_mask(float(op.rate), self._dropout_seed(op), self.step.data_ptr(), m.data_ptr(),
                            self.N * t.H * t.W, t.C, torch.cuda.current_stream().cuda_stream)
        return m.view(self.N, t.H, t.W, t.C)
