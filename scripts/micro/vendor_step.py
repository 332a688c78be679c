"""the same network trained by PyTorch-ROCm itself (MIOpen / rocBLAS kernels, eager autograd): HuggingFace transformers'
MobileNetV2ForSemanticSegmentation -- MobileNetV2 at output stride 16 + the DeepLabV3 ASPP-Lite head, i.e. the BASELINE configs[0]
model, layer for layer the graph of get_deeplabv3p_model('mobilenetv2_lite') (tests/test_oracle_vs_transformers.py shows the two
agree to 1e-10 with shared weights) -- forward + CE(ignore 255) on the upsampled logits + backward + SGD(momentum 0.9, weight decay),
513 x 513, batch 16, fp32, BatchNorm in training mode.  A yardstick for the step time, not a dependency.
GPU box: python3 scripts/micro/vendor_step.py [batch]"""
import sys
import time

import torch
import transformers

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''      # 'nchw' / 'nhwc': one memory format only (MIOpen's first-call search takes minutes)
H = W = 513
dev = 'cuda'
torch.backends.cudnn.benchmark = True
cfg = transformers.MobileNetV2Config(output_stride=16, tf_padding=True, finegrained_output=True, hidden_act='relu6', layer_norm_eps=1e-3,
                                     num_labels=21, classifier_dropout_prob=0.5, semantic_loss_ignore_index=255)
for fmt_name, fmt in (('channels_last (the NHWC memory the product uses)', torch.channels_last), ('NCHW', torch.contiguous_format)):
    if (ONLY == 'nchw' and fmt is torch.channels_last) or (ONLY == 'nhwc' and fmt is not torch.channels_last):
        continue
    model = transformers.MobileNetV2ForSemanticSegmentation(cfg).to(dev).to(memory_format=fmt).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=4e-5)
    x = torch.rand(B, 3, H, W, device=dev).mul_(2).sub_(1).contiguous(memory_format=fmt)
    y = torch.randint(0, 21, (B, H, W), device=dev)
    y[torch.rand(B, H, W, device=dev) < 0.05] = 255

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(pixel_values=x, labels=y)
        out.loss.backward()
        opt.step()
        return out.loss
    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print('PyTorch-ROCm eager, %s: %.2f ms/step, %.1f images/s (batch %d, loss %.4f, torch %s)' % (fmt_name, dt * 1e3, B / dt, B, float(loss.detach()), torch.__version__), flush=True)
    del model, opt
