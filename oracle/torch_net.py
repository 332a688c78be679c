"""torch-CPU (oneDNN) restatement of the same DeepLabV3+ graphs: the block / backbone builders of oracle/np_net.py
(which follow the reference's deeplabv3p/models/*.py line by line) run unchanged, only the primitive layers are
re-implemented with torch.nn.functional ops and torch autograd instead of NumPy and the hand-written tape.

TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  Two uses:
  * whole-model triangulation of the NumPy oracle (tests/test_oracle_ops.py::test_whole_model_matches_torch_autograd):
    an independent implementation of every op AND of reverse-mode differentiation agrees with it to 1e-9 in fp64;
  * bench.py's `cpu_baseline`: BASELINE.md section 3 asks for "the build's own CPU restatement of the same graph using
    torch-CPU ops (oneDNN)" as the stand-in for the tf.keras reference, which cannot be installed here.
Tensors keep the NHWC *shape* the builders index (x.v.shape[-1] channels); a permuted view hands torch its NCHW
(channels_last memory format) operand.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops as O
from .np_net import Net, Var, OracleModel


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


class TorchNet(Net):
    def __init__(self, dtype=np.float32, seed=0):
        super().__init__(np.float64, seed)
        self.tdtype = torch.float64 if dtype in (np.float64, torch.float64) else torch.float32
        self.t = {}              # name -> torch parameter / buffer
        self.reg = []            # (l2, tensor) of the trainable, regularised kernels met in this forward

    def q(self, a):              # (the oracle passes the input batch through q: here it becomes a torch tensor)
        return torch.as_tensor(np.asarray(a), dtype=self.tdtype) if not torch.is_tensor(a) else a

    def tparam(self, name, shape, init, trainable=True, l2=0.0):
        self.param(name, shape, init, trainable, l2)
        if name not in self.t:
            self.t[name] = torch.tensor(self.params[name], dtype=self.tdtype,
                                        requires_grad=trainable and self.layer_is_trainable(name.rsplit('/', 1)[0]))
        if l2 and self.t[name].requires_grad:
            self.reg.append((l2, self.t[name]))
        return self.t[name]

    def begin(self, training=True):
        super().begin(training)
        self.reg = []

    @staticmethod
    def _pad(x, H, W, k, stride, rate, padding):
        _, _, (pt, pb, pl, pr) = O.resolve_padding(H, W, k, stride, rate, padding)
        return F.pad(x, (pl, pr, pt, pb)) if (pt or pb or pl or pr) else x

    def conv2d(self, x, filters, k, name, stride=1, rate=1, padding='same', use_bias=False, he_normal=False, keep_f32=False):
        cin = x.v.shape[-1]
        init = ((lambda s: O.he_normal(self.rng, s, k * k * cin)) if he_normal else
                (lambda s: O.glorot_uniform(self.rng, s, k * k * cin, k * k * filters)))
        w = self.tparam(name + '/kernel', (k, k, cin, filters), init, l2=O.L2_FACTOR)
        b = self.tparam(name + '/bias', (filters,), np.zeros, l2=O.L2_FACTOR) if use_bias else None
        xi = self._pad(_nchw(x.v), x.v.shape[1], x.v.shape[2], k, stride, rate, padding)
        y = F.conv2d(xi, w.permute(3, 2, 0, 1), b, stride=stride, dilation=rate)
        return Var(_nhwc(y))

    def dwconv2d(self, x, k, name, stride=1, rate=1, padding='same'):
        c = x.v.shape[-1]
        w = self.tparam(name + '/depthwise_kernel', (k, k, c, 1), lambda s: O.glorot_uniform(self.rng, s, k * k * c, k * k * 1))
        xi = self._pad(_nchw(x.v), x.v.shape[1], x.v.shape[2], k, stride, rate, padding)
        y = F.conv2d(xi, w.permute(2, 3, 0, 1), None, stride=stride, dilation=rate, groups=c)
        return Var(_nhwc(y))

    def bn(self, x, name, eps=1e-3, momentum=0.99):
        c = x.v.shape[-1]
        gamma = self.tparam(name + '/gamma', (c,), np.ones)
        beta = self.tparam(name + '/beta', (c,), np.zeros)
        mm = self.tparam(name + '/moving_mean', (c,), np.zeros, trainable=False)
        mv = self.tparam(name + '/moving_variance', (c,), np.ones, trainable=False)
        train = self.training and self.layer_is_trainable(name)
        # (torch feeds the unbiased variance into the running average, Keras' non-fused path the biased one: the moving
        # statistics of this net are for timing only, values and gradients do not depend on them in training mode)
        y = F.batch_norm(_nchw(x.v), mm, mv, gamma, beta, training=train, momentum=1.0 - momentum, eps=eps)
        out = Var(_nhwc(y))
        out.tag = name
        return out

    def act(self, x, kind):
        v = x.v
        if kind == O.ACT_RELU:
            y = F.relu(v)
        elif kind == O.ACT_RELU6:
            y = torch.clamp(v, 0.0, 6.0)
        elif kind == O.ACT_HSIGMOID:
            y = torch.clamp(v + 3.0, 0.0, 6.0) / 6.0
        elif kind == O.ACT_HSWISH:
            y = v * (torch.clamp(v + 3.0, 0.0, 6.0) / 6.0)
        else:
            y = v
        return Var(y)

    def add(self, a, b):
        return Var(a.v + b.v)

    def mul_bcast(self, x, s):
        return Var(x.v * s.v)

    def concat(self, xs):
        return Var(torch.cat([x.v for x in xs], dim=-1))

    def maxpool2d(self, x, k, stride, pad):
        pt, pb, pl, pr = pad
        return Var(_nhwc(F.max_pool2d(F.pad(_nchw(x.v), (pl, pr, pt, pb)), k, stride)))

    def global_avgpool(self, x):
        return Var(x.v.mean(dim=(1, 2), keepdim=True))

    def resize(self, x, out_h, out_w, keep_f32=False):
        return Var(_nhwc(F.interpolate(_nchw(x.v), size=(out_h, out_w), mode='bilinear', align_corners=False)))

    def dropout(self, x, name, rate=0.5):
        mask = self.dropout_masks.get(name)
        if not self.training or mask is None:
            return x
        return Var(x.v * torch.as_tensor(mask, dtype=self.tdtype) / (1.0 - rate))


class TorchModel(OracleModel):
    """OracleModel with torch tensors and autograd underneath; train_step / loss_and_grads keep the oracle's signatures"""
    net_class = TorchNet

    def predict(self, x):
        with torch.no_grad():
            logits = self._forward_graph(np.asarray(x), x.shape[1], x.shape[2], training=False).v
        return logits.numpy(), torch.softmax(logits, -1).numpy()

    def loss_and_grads(self, x, labels, dropout_masks=None, ignore_index=255, loss=None, sample_weight=None):
        net = self.net
        for t in net.t.values():
            t.grad = None
        net.dropout_masks = dropout_masks or {}
        logits = self._forward_graph(np.asarray(x), x.shape[1], x.shape[2], training=True).v
        N, H, W, C = logits.shape
        lab = torch.as_tensor(np.asarray(labels).reshape(N, H, W)).long()
        # mean over ALL entries, ignored ones included (Keras reduction, loss.py:121-156); the 1e-7 clip of the
        # probabilities is not restated here (it binds only for p_y < 1e-7)
        ce = F.cross_entropy(_nchw(logits), lab, ignore_index=ignore_index if ignore_index else -100, reduction='sum') / (N * H * W)
        reg = sum(l2 * (t.double() ** 2).sum() for l2, t in net.reg) if net.reg else 0.0
        ce.backward()
        net.grads = {n: (t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))) for n, t in net.t.items()
                     if t.requires_grad}
        return float(ce.detach()) + float(reg), float(ce.detach()), logits.detach().numpy()

    def sgd_step(self, lr=1e-2, momentum=0.9, optimizer='sgd'):
        net = self.net
        with torch.no_grad():
            for n, t in net.t.items():
                if not t.requires_grad:
                    continue
                g = t.grad if t.grad is not None else torch.zeros_like(t)
                g = g + 2.0 * net.l2.get(n, 0.0) * t
                v = self.velocity.get(n)
                v = momentum * v - lr * g if v is not None else -lr * g
                self.velocity[n] = v
                t += v
