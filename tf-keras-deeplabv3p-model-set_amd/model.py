"""create deeplabv3p models -- the drop-in counterpart of the reference's deeplabv3p/model.py:
`deeplab_model_map` (:23-48) and `get_deeplabv3p_model(model_type, num_classes, model_input_shape,
output_stride, freeze_level=0, weights_path=None, training=True, use_subpixel=False)` (:51-117), same
names, argument meaning and error behaviour.  What it returns is not a tf.keras.Model wrapping
TensorFlow ops but a `DeeplabModel` façade over the MI355X HIP executor, exposing what the reference's
callers use: compile / fit / fit_generator / train_on_batch / predict / summary / save / load_weights /
layers[i].trainable (train.py:155-247, eval.py:33-36, deeplab.py:71-113).
"""
import os
import sys
import time
from functools import partial

import numpy as np

from .graph import ACT_NONE
from .mobilenetv2 import Deeplabv3pMobileNetV2, Deeplabv3pLiteMobileNetV2

#
# A map of model type to construction function for DeepLabv3+ (reference model.py:23-48).  Entries that
# are out of this build's hot-path scope raise the same ValueError as an unknown key.
#
deeplab_model_map = {
    'mobilenetv2': partial(Deeplabv3pMobileNetV2, alpha=1.0),
    'mobilenetv2_lite': partial(Deeplabv3pLiteMobileNetV2, alpha=1.0),
}


def _register_optional():
    try:
        from .xception import Deeplabv3pXception
        deeplab_model_map['xception'] = Deeplabv3pXception
    except ImportError:
        pass
    try:
        from .mobilenetv3 import (Deeplabv3pMobileNetV3Large, Deeplabv3pLiteMobileNetV3Large,
                                  Deeplabv3pMobileNetV3Small, Deeplabv3pLiteMobileNetV3Small)
        deeplab_model_map['mobilenetv3large'] = partial(Deeplabv3pMobileNetV3Large, alpha=1.0)
        deeplab_model_map['mobilenetv3large_lite'] = partial(Deeplabv3pLiteMobileNetV3Large, alpha=1.0)
        deeplab_model_map['mobilenetv3small'] = partial(Deeplabv3pMobileNetV3Small, alpha=1.0)
        deeplab_model_map['mobilenetv3small_lite'] = partial(Deeplabv3pLiteMobileNetV3Small, alpha=1.0)
    except ImportError:
        pass
    try:
        from .resnet50 import Deeplabv3pResNet50
        deeplab_model_map['resnet50'] = Deeplabv3pResNet50
    except ImportError:
        pass


_register_optional()


class SGD:
    """Keras SGD(learning_rate, momentum=0.9, nesterov=False) (common/model_utils.py:124)"""

    def __init__(self, learning_rate=0.01, momentum=0.9, nesterov=False):
        if nesterov:
            raise ValueError('nesterov momentum is not on the hot path')
        self.learning_rate, self.momentum = learning_rate, momentum
        self.iterations = 0          # Keras `optimizer.iterations`: a schedule starts at 0 with a NEW optimizer

    def lr_at(self, step):
        lr = self.learning_rate
        return float(lr(step)) if callable(lr) else float(lr)

    def spec(self):
        return ('sgd', float(self.momentum))


class Adam(SGD):
    """Keras Adam(learning_rate, epsilon=1e-7, amsgrad=False) (common/model_utils.py:119)"""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False):
        if amsgrad:
            raise ValueError('amsgrad is not built (the reference passes amsgrad=False)')
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.momentum = 0.0
        self.iterations = 0

    def spec(self):
        return ('adam', float(self.beta_1), float(self.beta_2), float(self.epsilon))


class RMSprop(SGD):
    """Keras RMSprop(learning_rate, rho=0.9, momentum=0.0, centered=False) (common/model_utils.py:121)"""

    def __init__(self, learning_rate=0.001, rho=0.9, momentum=0.0, epsilon=1e-7, centered=False):
        if momentum or centered:
            raise ValueError('RMSprop momentum / centered are not built (the reference uses momentum=0.0, centered=False)')
        self.learning_rate, self.rho, self.epsilon = learning_rate, rho, epsilon
        self.momentum = 0.0
        self.iterations = 0

    def spec(self):
        return ('rmsprop', float(self.rho), float(self.epsilon))


def get_optimizer(optim_type, learning_rate, average_type=None, decay_type=None, decay_steps=100000):
    """common/model_utils.py:112-130: 'sgd' (momentum 0.9), 'adam', 'rmsprop' with the four learning-rate schedules"""
    optim_type = optim_type.lower()
    if optim_type not in ('sgd', 'adam', 'rmsprop'):
        raise ValueError('Unsupported optimizer type')
    if average_type:
        raise ValueError('averaged optimizers are out of scope')
    lr = learning_rate
    if decay_type:
        d = decay_type.lower()
        if d == 'cosine':          # CosineDecay(alpha=0.2)
            lr = lambda s: learning_rate * (0.8 * 0.5 * (1 + np.cos(np.pi * min(s, decay_steps) / decay_steps)) + 0.2)
        elif d == 'exponential':   # ExponentialDecay(decay_rate=0.9)
            lr = lambda s: learning_rate * 0.9 ** (s / decay_steps)
        elif d == 'polynomial':    # PolynomialDecay(end=lr/100, power=1)
            lr = lambda s: (learning_rate - learning_rate / 100) * (1 - min(s, decay_steps) / decay_steps) + learning_rate / 100
        elif d == 'piecewise_constant':
            b = [500, int(decay_steps * 0.9), decay_steps]
            v = [0.001, learning_rate, learning_rate / 10., learning_rate / 100.]
            lr = lambda s: v[sum(1 for x in b if s > x)]
        else:
            raise ValueError('Unsupported lr decay type')
    if optim_type == 'adam':
        return Adam(lr, epsilon=1e-7)
    if optim_type == 'rmsprop':
        return RMSprop(lr, rho=0.9)
    return SGD(lr, momentum=0.9)


class SparseCategoricalCrossEntropy(object):
    """deeplabv3p/loss.py:121-156: configuration holder -- the arithmetic is fused into the HIP head kernel"""

    def __init__(self, ignore_index=None, from_logits=False):
        if from_logits:
            raise ValueError('the model emits probabilities (Softmax pred_mask); from_logits is not supported')
        self.ignore_index = ignore_index
        self.from_logits = from_logits
        self.__name__ = 'sparse_categorical_crossentropy'


class WeightedSparseCategoricalCrossEntropy(object):
    """deeplabv3p/loss.py:159-191 (train.py:115, --weighted_type balanced): -w[y] * log(p_y)"""

    def __init__(self, weights, ignore_index=None, from_logits=False):
        if from_logits:
            raise ValueError('the model emits probabilities (Softmax pred_mask); from_logits is not supported')
        self.weights = np.array(weights).astype('float32')
        self.ignore_index = ignore_index
        self.from_logits = from_logits
        self.__name__ = 'weighted_sparse_categorical_crossentropy'


class SparseSoftmaxFocalLoss(object):
    """deeplabv3p/loss.py:63-118 (train.py:131, --loss focal): -alpha * (1 - p_y)^gamma * log(p_y)"""

    def __init__(self, gamma=2.0, alpha=0.25, ignore_index=None, from_logits=False):
        if from_logits:
            raise ValueError('the model emits probabilities (Softmax pred_mask); from_logits is not supported')
        self.gamma, self.alpha = gamma, alpha
        self.ignore_index = ignore_index
        self.from_logits = from_logits
        self.__name__ = 'sparse_softmax_focal_loss'


def Jaccard(y_true=None, y_pred=None):
    """deeplabv3p/metrics.py:29-46, train.py:140 `metrics = {'pred_mask': Jaccard}`: a marker for compile(metrics=...);
    the counts come from dl3p_class_counts on the device and jaccard_from_counts evaluates them"""
    raise RuntimeError('Jaccard is evaluated on the device: pass it to compile(metrics=...)')


def jaccard_from_counts(counts):
    """counts (N,3,C) = per image [intersection, label pixels, predicted pixels] per class -> metrics.py:29-46"""
    c = np.asarray(counts, dtype=np.float64)
    inter, true, pred = c[:, 0], c[:, 1], c[:, 2]
    union = true + pred - inter
    legal = true > 0
    ious = []
    for i in range(c.shape[2]):
        if legal[:, i].any():
            ious.append(float(np.mean(inter[legal[:, i], i] / union[legal[:, i], i])))
    return float(np.mean(ious)) if ious else float('nan')


def _wants_jaccard(metrics):
    if not metrics:
        return False
    items = metrics.values() if isinstance(metrics, dict) else metrics
    flat = []
    for it in items:
        flat += list(it) if isinstance(it, (list, tuple)) else [it]
    return any(it is Jaccard or getattr(it, '__name__', it) == 'Jaccard' for it in flat)


def loss_spec(loss):
    """-> ('ce',) | ('weighted', weights) | ('focal', gamma, alpha): what the head kernel (and the oracle) need"""
    if isinstance(loss, WeightedSparseCategoricalCrossEntropy):
        return ('weighted', loss.weights)
    if isinstance(loss, SparseSoftmaxFocalLoss):
        return ('focal', float(loss.gamma), float(loss.alpha))
    return ('ce',)


class DistContext:
    """one process per GPU; collectives are RCCL (torch.distributed backend 'nccl') over xGMI.
    Gradient buckets are all-reduced on a side HIP stream so that they overlap the rest of backward."""

    def __init__(self, sync_bn=True, n_buckets=4):
        import torch.distributed as dist
        self.dist = dist
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.sync_bn = sync_bn
        self.n_buckets = n_buckets
        self._streams = {}
        self._bn_pg = None

    def all_reduce(self, t):
        """blocking (stream-ordered) sum all-reduce: SyncBatchNorm statistics of the forward pass"""
        self.dist.all_reduce(t, group=self._bn_group())

    def _bn_group(self):
        # SyncBatchNorm gets its own communicator (own RCCL stream): a statistics all-reduce never queues behind
        # a gradient bucket that is still on the wire.  DL3P_ONE_COMM=1 keeps everything on the default communicator
        # (one of the switches watchdog.FirstStepsGuard names when the first steps of a multi-rank job hang)
        if os.environ.get('DL3P_ONE_COMM', '0') not in ('', '0'):
            return None
        if self._bn_pg is None and self.dist.is_initialized() and self.dist.get_backend() == 'nccl':
            self._bn_pg = self.dist.new_group(backend='nccl')
        return self._bn_pg

    def broadcast(self, t, src=0):
        self.dist.broadcast(t, src)

    def _side(self, which):
        import torch
        if self._streams.get(which) is None:
            self._streams[which] = torch.cuda.Stream()
        return self._streams[which]

    def all_reduce_async(self, t):
        """sum all-reduce of a gradient bucket, ordered after everything already queued on the
        compute stream but running beside what is queued later"""
        import torch
        if not t.is_cuda:
            self.dist.all_reduce(t)
            return
        side = self._side('grad')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.dist.all_reduce(t)

    def wait_all(self):
        import torch
        if self._streams.get('grad') is not None:
            torch.cuda.current_stream().wait_stream(self._streams['grad'])

    def bn_all_reduce_begin(self, t):
        """SyncBatchNorm backward sums: start the all-reduce beside the compute stream (the deferred weight
        gradient of the previous layer runs meanwhile) ..."""
        import torch
        if not t.is_cuda:
            self.dist.all_reduce(t)
            return
        side = self._side('bn')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.dist.all_reduce(t, group=self._bn_group())

    def bn_all_reduce_end(self):
        """... and make the compute stream wait for it"""
        import torch
        if self._streams.get('bn') is not None:
            torch.cuda.current_stream().wait_stream(self._streams['bn'])


class StagedBatch:
    """a batch on its way to the device (BatchFeeder.put): device tensors with their host shapes, the staging slot they live in"""
    def __init__(self, slot, x, y, sample_weight):
        self.slot, self.x, self.y, self.sample_weight = slot, x, y, sample_weight


class BatchFeeder:
    """fit_generator's host -> device boundary (train.py:177-187 feeds numpy batches): TWO sets of pinned host buffers and device
    staging buffers and a copy stream, so that batch k + 1 crosses PCIe while the device runs step k.  put() copies the arrays into
    pinned memory (CPU) and starts the DMA on the copy stream; the consumer makes the compute stream wait for that DMA
    (consume), enqueues the step (whose first kernels read the staging buffers: uint8 -> float normalisation, label
    preparation) and releases the slot (release) -- the copy stream does not overwrite a slot before its last consumer ran."""
    def __init__(self, device='cuda'):
        import torch
        self.dev = device
        self.stream = torch.cuda.Stream()
        self.bufs = [{}, {}]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [None, None]
        self.k = 0

    def _move(self, slot, name, a):
        import torch
        t = torch.as_tensor(a)
        shape = tuple(t.shape)
        if t.dtype not in (torch.uint8, torch.float32):
            t = t.to(torch.float32)
        t = t.reshape(-1)
        pair = self.bufs[slot].get(name)
        if pair is None or pair[0].numel() != t.numel() or pair[0].dtype != t.dtype:
            pair = self.bufs[slot][name] = (torch.empty(t.numel(), dtype=t.dtype, pin_memory=True),
                                            torch.empty(t.numel(), dtype=t.dtype, device=self.dev))
        host, dev = pair
        # one thread's memcpy (0.5 ms for 12.6 MB).  Tensor.copy_ spreads it over every core of the host and the OpenMP workers then
        # spin for their block time: on the 256-thread GPU box that starved the HIP runtime's own thread and a staged step took
        # 35 ms instead of 12 (scripts/micro/feed_dbg.py)
        np.copyto(host.numpy(), t.numpy())
        with torch.cuda.stream(self.stream):
            dev.copy_(host, non_blocking=True)
        return dev.view(shape)

    def put(self, x, y=None, sample_weight=None):
        slot = self.k & 1
        self.k += 1
        self.ready[slot].synchronize()          # (the DMA that last read this slot's pinned buffers; long done)
        if self.free[slot] is not None:
            self.stream.wait_event(self.free[slot])
        sw = sample_weight
        out = StagedBatch(slot, self._move(slot, 'x', x), None if y is None else self._move(slot, 'y', y),
                          sw if (sw is None or isinstance(sw, str)) else self._move(slot, 'sw', sw))
        self.ready[slot].record(self.stream)
        return out

    def consume(self, staged):
        import torch
        torch.cuda.current_stream().wait_event(self.ready[staged.slot])

    def release(self, staged):
        import torch
        ev = self.free[staged.slot] = self.free[staged.slot] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())


class LateScalar:
    """a device scalar read back one step late: an asynchronous copy into pinned memory behind the step that produced it, the value
    taken when the NEXT step has been enqueued (fit's loss: the device never waits for the host to look at a number)"""
    def __init__(self):
        import torch
        self.host = [torch.zeros(1, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.evt = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0

    def post(self, dev_scalar):
        i = self.k & 1
        self.k += 1
        self.host[i].copy_(dev_scalar.reshape(-1)[:1], non_blocking=True)
        self.evt[i].record()
        return i

    def value(self, i):
        self.evt[i].synchronize()
        return float(self.host[i][0])


class DeeplabModel:
    def __init__(self, graph, head, model_type, num_classes, input_shape, training, backbone_len, seed=0, bf16=False):
        self.graph, self.head = graph, head
        self.bf16 = bool(bf16)          # mixed_precision policy at build time (train.py:37-46)
        self.name = 'deeplabv3p_' + model_type
        self.model_type, self.num_classes = model_type, num_classes
        self.input_shape_hw = tuple(input_shape)
        self.flatten_output = training      # Reshape((H*W, C)) when built for training (model.py:79-80)
        self.backbone_len = backbone_len
        # `model.layers` as Keras orders a functional model (by depth from the output, graph.keras_layer_order): what
        # layers[i].trainable (train.py:222-229, model.py:106-115), get_weights() and topological weight loading index.
        # The flat device buffers keep creation order (graph.all_params()).
        self.layers = graph.keras_layer_order(getattr(graph, 'output_layer', None))
        graph.init_weights()
        self.optimizer = None
        self.loss = None
        self.dist = None
        self.seed = seed
        self._store = None
        self._exec = {}
        self._steps = 0
        self.use_graphs = True
        self.stop_training = False

    # ---- Keras surface ------------------------------------------------------------------
    @property
    def input_shape(self):
        return (None,) + self.input_shape_hw + (3,)

    @property
    def output_shape(self):
        H, W = self.input_shape_hw
        return (None, H * W, self.num_classes) if self.flatten_output else (None, H, W, self.num_classes)

    def get_layer(self, name):
        return self.graph.layer_by_name[name]

    def count_params(self):
        return sum(p.size for p in self.graph.all_params())

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        print_fn('%-48s %-24s %12s' % ('Layer (type)', 'Output Shape', 'Param #'))
        print_fn('=' * 86)
        for l in self.layers:
            shape = (None,) + tuple(l.output_shape) if l.output_shape else ''
            print_fn('%-48s %-24s %12d' % ('%s (%s)' % (l.name, l.kind), str(shape), l.count_params()))
        tr = sum(p.size for p in self.graph.all_params() if p.trainable)
        tot = self.count_params()
        print_fn('=' * 86)
        print_fn('Total params: {:,}'.format(tot))
        print_fn('Trainable params: {:,}'.format(tr))
        print_fn('Non-trainable params: {:,}'.format(tot - tr))
        # which batch variance feeds the moving averages is a build-time choice that changes inference after training:
        # say so where a user looks, so weights trained here and reference-trained weights are not compared unawares
        mv = getattr(self, 'bn_moving_variance', 'biased')
        print_fn("BatchNormalization moving variance: %s batch variance (%s)" % (
            mv, 'Keras SyncBatchNormalization, what layers.py:63-70 selects for TF 2.2 .. 2.9' if mv == 'biased'
            else "fused BatchNormalization with Bessel's correction, what layers.py:63-70 selects under the pinned tensorflow==2.11.0"))

    def compile(self, optimizer=None, loss=None, metrics=None, sample_weight_mode=None, distributed=None,
                sync_bn=True, **kw):
        """model.compile(optimizer, loss, ...) (train.py:157,224): (re)binds optimiser and loss and drops the
        traced plans, so `layers[i].trainable` changes take effect like a Keras recompile"""
        # Keras keeps the slot variables and `iterations` on the optimizer OBJECT: compiling again with the same object
        # (train.py:224 after unfreezing) continues its schedule and momentum, a new object (train.py:190-224 builds one
        # for the second stage) starts its schedule at step 0 with fresh slots
        same_optimizer = optimizer is not None and optimizer is self.optimizer
        self.optimizer = optimizer or SGD(0.01)
        self.loss = loss or SparseCategoricalCrossEntropy(ignore_index=255)
        if sample_weight_mode not in (None, 'temporal'):
            raise ValueError("sample_weight_mode must be None or 'temporal' (train.py:116-120)")
        self.sample_weight_mode = sample_weight_mode
        self.metrics = metrics
        self._jaccard = _wants_jaccard(metrics)
        if distributed is None:
            import torch.distributed as dist
            distributed = dist.is_available() and dist.is_initialized() and (
                dist.get_world_size() > 1 or bool(os.environ.get('DL3P_FORCE_DIST')))
        if os.environ.get('DL3P_SYNC_BN', '1') in ('0',):
            sync_bn = False            # per-replica BatchNorm statistics (no forward collectives)
        # `distributed` may also be a DistContext (or a test double of one: tests/test_dist_gpu.py drives the executor's
        # world-size arithmetic on one GPU with a context whose sum over two identical ranks is a multiplication by 2)
        self.dist = distributed if isinstance(distributed, DistContext) else (DistContext(sync_bn) if distributed else None)
        self._exec = {}
        if self._store is not None:
            self._store.refresh_masks()
            if not same_optimizer:
                self._store.V.zero_()
                if self._store.V2 is not None:
                    self._store.V2.zero_()
                self._store.opt_step.zero_()       # (the dropout stream, store.step, keeps counting)
        return self

    def _ensure_store(self):
        import torch
        if self._store is None:
            if not torch.cuda.is_available():
                raise RuntimeError('the DeepLabV3+ HIP path needs an MI355X device; there is no CPU fallback')
            from .executor import ParamStore
            self._store = ParamStore(self.graph, torch.device('cuda', torch.cuda.current_device()), bf16=self.bf16)
            if self.dist is not None:
                self.dist.broadcast(self._store.P, 0)   # identical replicas (MirroredStrategy semantics)
                self._store.transpose()
        return self._store

    def _executor(self, batch, training):
        key = (batch, training)
        if key not in self._exec:
            from .executor import Executor
            store = self._ensure_store()
            ignore = self.loss.ignore_index if self.loss is not None else 255
            opt = self.optimizer.spec() if self.optimizer is not None else ('sgd', 0.9)
            rank = self.dist.rank if self.dist is not None else 0
            self._exec[key] = Executor(self.graph, self.head, store, batch, training, self.num_classes,
                                       ignore_index=ignore, dist=self.dist if training else None,
                                       seed=self.seed + 7919 * rank, loss=loss_spec(self.loss), optimizer=opt,
                                       sample_weighted=getattr(self, 'sample_weight_mode', None) == 'temporal',
                                       class_counts=getattr(self, '_jaccard', False))
        return self._exec[key]

    def train_on_batch(self, x, y, sample_weight=None, return_tensor=False):
        """one optimiser step on (x (B,H,W,3) float32 in [-1,1] or uint8, y (B,H*W,1) class ids[, sample_weight (B,H*W)
        with compile(sample_weight_mode='temporal'); 'adaptive' = the generator's balanced class weights, computed on
        the device from uint8 labels]); returns the data loss"""
        if self.optimizer is None:
            raise RuntimeError('You must compile your model before training')
        staged = x if isinstance(x, StagedBatch) else None
        if staged is not None:              # (prefetch_batch: the arrays are already on their way; the compute stream waits for the DMA)
            x, y, sample_weight = staged.x, staged.y, staged.sample_weight
            self._feeder().consume(staged)
        # more than one rank: the executor trace (every collective once), the graph capture and the first replays are where
        # a multi-process job can hang -- bounded by watchdog.FirstStepsGuard (prints the switches to try, exits non-zero)
        guarded = self.dist is not None and self.dist.world_size > 1 and self._steps < 3
        from .watchdog import FirstStepsGuard
        # (an unguarded step arms nothing: world 1 / timeout 0; `with` cancels the timer on every exit path -- an exception
        # the caller catches must not leave it armed to os._exit(3) the process minutes later)
        with FirstStepsGuard(self.dist.rank if guarded else 0, self.dist.world_size if guarded else 1,
                             'train step %d' % self._steps):
            ex = self._executor(int(x.shape[0]), True)
            ex.set_inputs(x, y, sample_weight)
            if staged is not None:
                self._feeder().release(staged)
            ex.lr.fill_(self.optimizer.lr_at(self.optimizer.iterations))
            if self.use_graphs and not ex.graphed and self._steps_on(ex) >= 1:
                ex.capture()
            ex.train_step()
            if guarded:
                import torch
                torch.cuda.synchronize()
        ex._steps = self._steps_on(ex) + 1
        self._steps += 1
        self.optimizer.iterations += 1
        if return_tensor:
            return ex.loss
        loss = float(ex.loss.item())
        self.last_metrics = {'Jaccard': jaccard_from_counts(ex.metric_counts.cpu().numpy())} if ex.metric_counts is not None else {}
        return loss

    @staticmethod
    def _steps_on(ex):
        return getattr(ex, '_steps', 0)

    def _feeder(self):
        if getattr(self, '_feed', None) is None:
            self._feed = BatchFeeder()
        return self._feed

    def prefetch_batch(self, x, y, sample_weight=None):
        """start moving a host batch to the device (pinned staging, copy stream; double-buffered: at most one batch ahead of the one
        in flight) -> a StagedBatch that train_on_batch takes in place of the arrays.  fit / fit_generator does this for every
        batch: the copy of batch k + 1 runs beside step k"""
        return self._feeder().put(x, y, sample_weight)

    def predict(self, x, verbose=0, batch_size=None):
        """probabilities (B,H*W,C) for a training-shaped model, (B,H,W,C) otherwise (eval.py:33-36)"""
        if isinstance(x, (list, tuple)):
            x = x[0]
        x = np.asarray(x)
        if x.dtype != np.uint8:                 # uint8 pixels are normalised on the device (Executor.set_inputs)
            x = x.astype(np.float32, copy=False)
        B = x.shape[0]
        ex = self._executor(B, False)
        ex.set_inputs(x)
        ex.forward()
        H, W = self.input_shape_hw
        p = ex.probs.view(B, H, W, self.num_classes).cpu().numpy()
        return p.reshape(B, H * W, self.num_classes) if self.flatten_output else p

    def predict_mask(self, x):
        """class ids (B,H,W) int32 = np.argmax(predict(x), -1) (eval.py:33-36, deeplab.py:97-99) taken on the device: the
        mask is 1/84 of the bytes of the 21-class probability tensor"""
        import torch
        x = np.asarray(x[0] if isinstance(x, (list, tuple)) else x)
        if x.dtype != np.uint8:
            x = x.astype(np.float32, copy=False)
        B = x.shape[0]
        ex = self._executor(B, False)
        ex.set_inputs(x)
        H, W = self.input_shape_hw
        pred = torch.empty(B * H * W, dtype=torch.int32, device='cuda')
        ex.eval_step(None, pred)
        return pred.view(B, H, W).cpu().numpy()

    def fit(self, x=None, steps_per_epoch=None, epochs=1, initial_epoch=0, verbose=1, callbacks=None,
            validation_data=None, validation_steps=None, weighted_type=None, **kw):
        """Keras-like loop over a Sequence/generator yielding (images, labels) (train.py:177-187).
        weighted_type='adaptive' (train.py --weighted_type, deeplabv3p/data.py:134-145): a generator that yields uint8
        labels and no weights gets the balanced per-image class weights computed on the device"""
        gen = x
        n = steps_per_epoch or len(gen)
        history = {'loss': []}
        for cb in callbacks or []:
            getattr(cb, 'set_model', lambda m: None)(self)
        for epoch in range(initial_epoch, epochs):
            t0, losses, metric_sums = time.time(), [], {}
            it = iter(gen) if not hasattr(gen, '__getitem__') else None
            fetch = (lambda i: gen[i]) if it is None else (lambda i: next(it))

            def stage(batch):
                # generators yield (x, y) or, in adaptive weighting mode, (x, y, sample_weight) (deeplabv3p/data.py:149-154)
                sw = batch[2] if len(batch) > 2 else None
                if isinstance(sw, dict):
                    sw = next(iter(sw.values()))          # {'pred_mask': weights}
                if sw is None and weighted_type == 'adaptive' and getattr(batch[1], 'dtype', None) == np.uint8:
                    sw = 'adaptive'
                return self.prefetch_batch(batch[0], batch[1], sw), int(np.shape(batch[0])[0])
            late = getattr(self, '_late_loss', None) or LateScalar()
            self._late_loss = late
            pending = None                                 # the loss of the step before, still on its way to the host
            staged, bsz = stage(fetch(0)) if n > 0 else (None, 0)
            for i in range(n):
                # the step is only enqueued here; while the device runs it the generator prepares the next batch (decode, augment,
                # resize on the host) and that batch crosses PCIe on the copy stream; the loss is looked at one step late
                loss_t = self.train_on_batch(staged, None, return_tensor=True)
                ex = self._executor(bsz, True)
                slot = late.post(loss_t)
                counts = ex.metric_counts.cpu().numpy() if ex.metric_counts is not None else None
                if i + 1 < n:
                    staged, bsz = stage(fetch(i + 1))
                if pending is not None:
                    losses.append(late.value(pending))
                pending = slot
                if counts is not None:
                    metric_sums.setdefault('Jaccard', []).append(jaccard_from_counts(counts))
                if losses and not np.isfinite(losses[-1]):      # TerminateOnNaN (train.py:64), one step late
                    self.stop_training = True
                    break
            if pending is not None:
                losses.append(late.value(pending))
                if not np.isfinite(losses[-1]):
                    self.stop_training = True
            logs = {'loss': float(np.mean(losses))}
            for mk, mv in metric_sums.items():             # Keras averages a metric over the epoch's batches
                logs[mk] = float(np.nanmean(mv))
            if validation_data is not None:
                logs['val_loss'] = self.evaluate(validation_data, validation_steps)
                for mk, mv in getattr(self, 'last_val_metrics', {}).items():
                    logs['val_' + mk] = mv
            history['loss'].append(logs['loss'])
            if verbose:
                print('Epoch %d/%d - %.1fs - loss: %.4f%s' % (
                    epoch + 1, epochs, time.time() - t0, logs['loss'],
                    (' - val_loss: %.4f' % logs['val_loss']) if 'val_loss' in logs else ''))
            if hasattr(gen, 'on_epoch_end'):
                gen.on_epoch_end()
            for cb in callbacks or []:
                getattr(cb, 'on_epoch_end', lambda e, l=None: None)(epoch, logs)
            if self.stop_training:
                break
        return history

    fit_generator = fit

    # ---- weights ------------------------------------------------------------------------
    def _sync_to_host(self):
        if self._store is not None:
            self._store.download()

    def _keras_params(self):
        return [p for l in self.layers for p in l.params]

    def get_weights(self):
        self._sync_to_host()
        return [p.value for p in self._keras_params()]

    def set_weights(self, weights):
        ps = self._keras_params()
        assert len(weights) == len(ps), (len(weights), len(ps))
        for p, w in zip(ps, weights):
            w = np.asarray(w, dtype=np.float32)
            assert w.shape == p.shape, (p.name, w.shape, p.shape)
            p.value = w.copy()
        if self._store is not None:
            self._store.upload()

    def get_weights_by_name(self):
        self._sync_to_host()
        return {p.name: p.value for p in self.graph.all_params()}

    def set_weights_by_name(self, d, strict=True):
        for p in self.graph.all_params():
            if p.name in d:
                w = np.asarray(d[p.name], dtype=np.float32)
                assert w.shape == p.shape, (p.name, w.shape, p.shape)
                p.value = w.copy()
            elif strict:
                raise KeyError(p.name)
        if self._store is not None:
            self._store.upload()

    def _keras_layers(self):
        """[(layer name, [(Keras weight name, array), ...]), ...] in topological order, weightless layers included"""
        self._sync_to_host()
        return [(l.name, [(p.name + ':0', p.value) for p in l.params]) for l in self.layers]

    def save(self, path):
        """whole-model checkpoint (train.py:247, deeplab.py:113).  `*.h5`: the Keras HDF5 layout (weights under
        `model_weights`, see h5io.py; `model_config` records the factory arguments -- it is not a Keras-deserialisable
        layer config, `keras.Model.load_weights` reads the file, `load_model` does not).  Anything else: `.npz` keyed
        by Keras weight name."""
        if path.endswith('.h5') or path.endswith('.hdf5'):
            import json
            from . import h5io
            cfg = json.dumps({'class_name': 'DeeplabV3p', 'config': {
                'name': self.name, 'factory': {'model_type': self.model_type, 'num_classes': self.num_classes,
                                               'model_input_shape': list(self.input_shape_hw),
                                               'output_stride': getattr(self.graph, 'output_stride', None),
                                               'training': bool(self.flatten_output),
                                               # (the moving statistics in this file were accumulated under this rule)
                                               'bn_moving_variance': getattr(self, 'bn_moving_variance', 'biased')}}})
            h5io.write_keras_h5(path, self._keras_layers(), whole_model=True, model_config=cfg)
            return
        np.savez(path if path.endswith('.npz') else path + '.npz', **self.get_weights_by_name())

    def save_weights(self, path):
        """`model.save_weights`: the same tree at the file root"""
        if path.endswith('.h5') or path.endswith('.hdf5'):
            from . import h5io
            h5io.write_keras_h5(path, self._keras_layers(), whole_model=False)
            return
        self.save(path)

    def _load_keras_h5(self, path, by_name, skip_mismatch):
        """Keras `load_weights_from_hdf5_group` (topological: the i-th layer WITH weights of the file feeds the i-th
        layer with weights of the model, names ignored) / `..._by_name` (layers matched by name, weights by position)"""
        from . import h5io
        file_layers, _ = h5io.read_keras_h5(path)
        mine = [l for l in self.layers if l.params]

        def assign(layer, ws, lname):
            if len(ws) != len(layer.params):
                if skip_mismatch:
                    return
                raise ValueError('Layer %r expects %d weight(s), but the saved layer %r holds %d'
                                 % (layer.name, len(layer.params), lname, len(ws)))
            for p, (wn, w) in zip(layer.params, ws):
                if tuple(w.shape) != p.shape:
                    if skip_mismatch:
                        continue
                    raise ValueError('Shape mismatch for %s: model %s, file %s %s' % (p.name, p.shape, wn, w.shape))
                p.value = np.ascontiguousarray(w, dtype=np.float32)

        if by_name:
            index = {n: ws for n, ws in file_layers if ws}
            for l in mine:
                if l.name in index:
                    assign(l, index[l.name], l.name)
        else:
            theirs = [(n, ws) for n, ws in file_layers if ws]
            if len(theirs) != len(mine):
                raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                                 % (len(theirs), len(mine)))
            for l, (n, ws) in zip(mine, theirs):
                assign(l, ws, n)
        if self._store is not None:
            self._store.upload()

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        """model.py:102-104 `model.load_weights(weights_path, by_name=False)`; Keras `.h5` files or `.npz`"""
        if not os.path.exists(path) and os.path.exists(path + '.npz'):
            path = path + '.npz'
        with open(path, 'rb') as fh:
            magic = fh.read(8)
        if magic == b'\x89HDF\r\n\x1a\n':
            return self._load_keras_h5(path, by_name, skip_mismatch)
        data = np.load(path)
        if by_name:
            self.set_weights_by_name({k: data[k] for k in data.files}, strict=False)
        else:
            self.set_weights([data[p.name] for p in self._keras_params()])


def _evaluate(self, gen, steps=None):
    """Keras `evaluate` / the validation pass of `fit`: mean of the compiled loss with inference-mode BN and no dropout,
    evaluated on the device (the probability tensor is neither written nor downloaded); fills last_val_metrics"""
    import torch
    n = steps or len(gen)
    tot, cnt, jac = 0.0, 0, []
    for i in range(n):
        x, y = np.asarray(gen[i][0]), np.asarray(gen[i][1])
        B = x.shape[0]
        ex = self._executor(B, False)
        ex.set_inputs(x if x.dtype == np.uint8 else x.astype(np.float32, copy=False),
                      y if y.dtype == np.uint8 else y.astype(np.float32, copy=False))
        counts = None
        if getattr(self, '_jaccard', False):
            counts = torch.zeros((B, 3, self.num_classes), dtype=torch.int32, device='cuda')
        loss = float(ex.eval_loss_step(counts).item())
        tot += loss * B
        cnt += B
        if counts is not None:
            jac.append(jaccard_from_counts(counts.cpu().numpy()))
    self.last_val_metrics = {'Jaccard': float(np.nanmean(jac))} if jac else {}
    return tot / max(cnt, 1)


DeeplabModel.evaluate = _evaluate


def miou_from_confusion(cm, class_names=None):
    """eval.py:462-497 on a (C,C) confusion matrix (rows = ground truth): PixelAcc, per-class ClassAcc / IoU / Dice /
    Freq, mClassAcc, mIoU (NaN -> 0 before the mean, like the reference), FWIoU"""
    cm = np.asarray(cm, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        inter = np.diag(cm)
        rows, cols, tot = cm.sum(axis=1), cm.sum(axis=0), cm.sum()
        pixel_acc = inter.sum() / tot
        class_acc = np.nan_to_num(inter / rows, nan=0.0)
        union = rows + cols - inter
        iou = np.nan_to_num(inter / union, nan=0.0)
        freq = np.nan_to_num(rows / tot, nan=0.0)
        dice = np.nan_to_num(2 * inter / (union + inter), nan=0.0)
    out = {'PixelAcc': float(pixel_acc), 'mClassAcc': float(class_acc.mean()), 'mIoU': float(iou.mean()),
           'FWIoU': float((freq[freq > 0] * iou[freq > 0]).sum()), 'IoU': iou, 'ClassAcc': class_acc, 'Dice': dice,
           'Freq': freq, 'confusion_matrix': cm}
    if class_names is not None:
        out['IoU_by_class'] = dict(sorted(zip(class_names, iou.tolist()), key=lambda kv: -kv[1]))
    return out


def _evaluate_miou(self, gen, steps=None, class_names=None, verbose=0):
    """eval.py:376-512 `eval_mIOU` for a generator of (images, labels) batches: inference forward, argmax and the
    confusion matrix stay on the device (Executor.eval_step); only the C x C counters come back.  -> miou_from_confusion"""
    import torch
    n = steps or len(gen)
    C = self.num_classes
    cm = torch.zeros(C * C, dtype=torch.int64, device='cuda')
    for i in range(n):
        batch = gen[i]
        x, y = np.asarray(batch[0]), np.asarray(batch[1])
        ex = self._executor(x.shape[0], False)
        ex.set_inputs(x if x.dtype == np.uint8 else x.astype(np.float32, copy=False),
                      y if y.dtype == np.uint8 else y.astype(np.float32, copy=False))
        ex.eval_step(cm)
    res = miou_from_confusion(cm.view(C, C).cpu().numpy(), class_names)
    if verbose:
        print('mIoU=%.3f' % (res['mIoU'] * 100))
        print('FWIoU=%.3f' % (res['FWIoU'] * 100))
        print('PixelAcc=%.3f' % (res['PixelAcc'] * 100))
        print('mClassAcc=%.3f' % (res['mClassAcc'] * 100))
    return res


DeeplabModel.evaluate_miou = _evaluate_miou


# model types whose graphs contain ops without a bf16 kernel on the TRAINING path are rejected when the model is built, not at
# the first train step.  Empty since round 3: the data gradient of a dense k x k conv (dl3p_col2im_bf16 behind the
# GEMM) and max pooling (dl3p_maxpool2d_*_bf16) closed the gap for Xception and ResNet50 -- train.py:37-46 applies the policy to
# every model type.
_NO_BF16_TRAINING = {}


def get_deeplabv3p_model(model_type, num_classes, model_input_shape, output_stride, freeze_level=0,
                         weights_path=None, training=True, use_subpixel=False, seed=0, bn_moving_variance='biased'):
    """deeplabv3p/model.py:51-117, same positional / keyword arguments.  Two extras, keyword-only in spirit:
    `seed` (weight initialisation) and `bn_moving_variance`, which of the two Keras classes CustomBatchNormalization
    (layers.py:63-70) can resolve to is restated for the MOVING variance: 'biased' = SyncBatchNormalization, Keras' non-fused
    path (TF 2.2 .. 2.9, and what north_star asks for: SyncBN over the global batch); 'unbiased' = plain fused
    BatchNormalization with Bessel's correction, which is what the string compare at layers.py:64 selects under the pinned
    tensorflow==2.11.0.  Training-mode outputs, losses and gradients do not depend on it; inference after
    training does (factor count / (count - 1) in the variance each step contributes)."""
    if bn_moving_variance not in ('biased', 'unbiased'):
        raise ValueError("bn_moving_variance must be 'biased' or 'unbiased' (got %r)" % (bn_moving_variance,))
    from . import graph as graph_mod, mixed_precision
    bf16 = mixed_precision.is_bf16()        # the global policy at build time, like Keras layers pick theirs up
    graph_mod.CHANNEL_ALIGN = 8 if bf16 else 4
    try:
        model = _build_model(model_type, num_classes, model_input_shape, output_stride, freeze_level, weights_path, training,
                             use_subpixel, seed, bf16)
        model.graph.bn_moving_variance = model.bn_moving_variance = bn_moving_variance
        return model
    finally:
        graph_mod.CHANNEL_ALIGN = 4


def _build_model(model_type, num_classes, model_input_shape, output_stride, freeze_level, weights_path, training,
                 use_subpixel, seed, bf16):
    # check if model type is valid
    if model_type not in deeplab_model_map.keys():
        raise ValueError('This model type is not supported now')
    if use_subpixel:
        raise ValueError('Subpixel head is experimental in the reference (README TODO) and not on the hot path')
    if not 1 <= int(num_classes) <= 253:
        # PNG label maps hold fewer than 254 classes (train.py:34); up to 32 classes a pixel's class vector stays in
        # registers in the head kernels, beyond that they walk the classes (dl3p_upsample_softmax_loss)
        raise ValueError('num_classes must be in [1, 253] (got %r)' % (num_classes,))

    model_function = deeplab_model_map[model_type]
    H, W = model_input_shape
    if bf16 and training and model_type in _NO_BF16_TRAINING:
        # the policy of train.py:37-46 applies to every model type there; here the mixed-precision kernels cover the
        # MobileNet families (BASELINE configs[4]).  Say so when the model is built, not at the first train step.
        raise ValueError("mixed_bfloat16 training is not built for model type '%s' (%s); build it under the float32 policy"
                         % (model_type, _NO_BF16_TRAINING[model_type]))
    # the reference builds a 21-class stub head and cuts it off again at layers[-5] (model.py:59-65);
    # the builders here stop at that tensor.  weights=None: offline, random init.
    g, x, backbone_len = model_function(input_shape=(H, W, 3), weights=None, num_classes=21, OS=output_stride,
                                        seed=seed)
    print('backbone layers number: {}'.format(backbone_len))
    base_len = len(g.layers)           # every layer but the new head (= len(base_model.layers), model.py:108)

    # new head (model.py:75-86): conv_upsample 1x1 (+bias) -> pred_resize -> [Reshape] -> Softmax('pred_mask').
    # The class dimension is padded to a multiple of 4 on the device (pad weights/bias stay exactly 0).
    cpad = (num_classes + g.align - 1) // g.align * g.align
    x = g.conv2d(x, num_classes, 1, 'conv_upsample', use_bias=True, pad_to=cpad)
    out = g.passthrough(x, 'Lambda', (H, W, num_classes), name='pred_resize')
    if training:
        out = g.passthrough(out, 'Reshape', (H * W, num_classes))
    out = g.passthrough(out, 'Softmax', (H * W, num_classes) if training else (H, W, num_classes), name='pred_mask')
    g.output_layer = out.klayer
    if x.tensor.C != cpad or not x.is_plain:
        raise AssertionError('head tensor layout')

    model = DeeplabModel(g, x, model_type, num_classes, (H, W), training, backbone_len, seed=seed, bf16=bf16)

    if weights_path:
        model.load_weights(weights_path, by_name=False)
        print('Load weights {}.'.format(weights_path))

    if freeze_level in [1, 2]:
        # Freeze the backbone part or freeze all but final feature map & input layers.
        num = (backbone_len, base_len)[freeze_level - 1]
        for i in range(num):
            model.layers[i].trainable = False
        print('Freeze the first {} layers of total {} layers.'.format(num, len(model.layers)))
    elif freeze_level == 0:
        # Unfreeze all layers.
        for i in range(len(model.layers)):
            model.layers[i].trainable = True
        print('Unfreeze all of the layers.')
    return model


class EvalCallBack:
    """common/callbacks.py:33-53: every `eval_epoch_interval` epochs run the mIOU evaluation and, with
    save_eval_checkpoint, keep the best model under the reference's file name pattern.  The reference re-reads the
    dataset from disk inside eval_mIOU; here the evaluation batches come from `eval_generator` (images, labels) and the
    argmax + confusion matrix stay on the device (DeeplabModel.evaluate_miou)."""

    def __init__(self, eval_generator, class_names=None, log_dir='.', eval_epoch_interval=10, save_eval_checkpoint=False,
                 steps=None):
        self.eval_generator, self.class_names, self.log_dir = eval_generator, class_names, log_dir
        self.eval_epoch_interval, self.save_eval_checkpoint, self.steps = eval_epoch_interval, save_eval_checkpoint, steps
        self.best_mIOU = 0.0
        self.history = []
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        if (epoch + 1) % self.eval_epoch_interval:
            return
        logs = logs or {}
        mIOU = float(self.model.evaluate_miou(self.eval_generator, class_names=self.class_names, steps=self.steps)['mIoU'])
        self.history.append((epoch + 1, mIOU))
        if self.save_eval_checkpoint and mIOU > self.best_mIOU:
            self.best_mIOU = mIOU
            nan = float('nan')
            name = 'ep{epoch:03d}-loss{loss:.3f}-Jaccard{Jaccard:.3f}-val_loss{val_loss:.3f}-val_Jaccard{val_Jaccard:.3f}-mIOU{mIOU:.3f}.h5'
            self.model.save(os.path.join(self.log_dir, name.format(
                epoch=epoch + 1, loss=logs.get('loss', nan), Jaccard=logs.get('Jaccard', nan),
                val_loss=logs.get('val_loss', nan), val_Jaccard=logs.get('val_Jaccard', nan), mIOU=mIOU)))
