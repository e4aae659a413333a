"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Restatement (torch, CPU) of `lpips.LPIPS(net='vgg', eval_mode=True)` as the reference's patch loss calls it
(lib/models/losses/lpips_loss.py:8-42: inputs in [0, 1] mapped to [-1, 1], the module run in bf16 / fp16, one value per pair),
from the published lpips==0.1.4 algorithm (requirements.txt:9; package and weights are absent here):
  ScalingLayer ((x - shift) / scale, shift = (-.030, -.088, -.188), scale = (.458, .448, .450)) -> torchvision VGG16 `features`
  cut after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 -> per layer: normalize_tensor (x / (sqrt(sum_c x^2) + 1e-10)), squared
  difference, 1x1 `lin` conv without bias (NetLinLayer; dropout is inactive in eval mode), spatial mean -> sum over the layers.
Gradients come from torch autograd over this restatement.

PARITY UNPINNED: neither `lpips` nor torchvision (nor their weights) exist in this image and the reference holds no vectors; the
state-dict names are lpips' own so that the real checkpoint loads unchanged."""
import torch
import torch.nn.functional as F

VGG_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG_CH = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512),
          (512, 512), (512, 512), (512, 512))
VGG_SLICE = (1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5)
SHIFT, SCALE = (-.030, -.088, -.188), (.458, .448, .450)


def param_shapes():
    s = {'scaling_layer.shift': (1, 3, 1, 1), 'scaling_layer.scale': (1, 3, 1, 1)}
    for i, (ci, co) in enumerate(VGG_CH):
        s[f'net.slice{VGG_SLICE[i]}.{VGG_IDX[i]}.weight'] = (co, ci, 3, 3)
        s[f'net.slice{VGG_SLICE[i]}.{VGG_IDX[i]}.bias'] = (co,)
    for k, c in enumerate((64, 128, 256, 512, 512)):
        s[f'lin{k}.model.1.weight'] = (1, c, 1, 1)
    return s


def random_params(seed=0):
    """Seeded stand-ins: He-initialised convs (activations stay O(1) through 13 ReLU layers), non-negative lin weights as in the
    trained model."""
    g = torch.Generator().manual_seed(seed)
    out = {'scaling_layer.shift': torch.tensor(SHIFT).view(1, 3, 1, 1), 'scaling_layer.scale': torch.tensor(SCALE).view(1, 3, 1, 1)}
    for name, shape in param_shapes().items():
        if name in out:
            continue
        if name.endswith('.bias'):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        elif name.startswith('lin'):
            out[name] = torch.rand(shape, generator=g) * 2.0 / shape[1]
        else:
            out[name] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * 9)) ** 0.5
    return out


def features(sd, x, q=None):
    q = q or (lambda t: t)
    taps = []
    h = q((x - sd['scaling_layer.shift']) / sd['scaling_layer.scale'])
    for i in range(13):
        if i in (2, 4, 7, 10):
            h = F.max_pool2d(h, 2, 2)
        p = f'net.slice{VGG_SLICE[i]}.{VGG_IDX[i]}'
        h = q(F.relu(q(F.conv2d(h, sd[f'{p}.weight'], sd[f'{p}.bias'], padding=1))))
        if i in (1, 3, 6, 9, 12):
            taps.append(h)
    return taps


def lpips(sd, pred, target, normalize=True, q=None):
    """-> [B] (the reference flattens the module's [B, 1, 1, 1])."""
    sd = {k: v.float() for k, v in sd.items()}
    pred, target = pred.float(), target.float()
    if normalize:
        pred, target = pred * 2 - 1, target * 2 - 1
    f0, f1 = features(sd, pred, q), features(sd, target, q)
    total = 0
    for k in range(5):
        n0 = f0[k] / (torch.sqrt(torch.sum(f0[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1[k] / (torch.sqrt(torch.sum(f1[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        total = total + F.conv2d((n0 - n1) ** 2, sd[f'lin{k}.model.1.weight']).mean(dim=(2, 3))
    return total.flatten()


def lpips_half(sd, pred, target, dtype, normalize=True):
    """The module exactly as the reference runs it: parameters, inputs, activations AND autograd in `dtype` (lpips_loss.py:31-41:
    `.to(lpips_dtype)` on both); returns the [B] distances in fp32, differentiable w.r.t. `pred` (which must be a `dtype` leaf)."""
    sdh = {k: v.to(dtype) for k, v in sd.items()}
    target = target.to(dtype)
    if normalize:
        pred, target = pred * 2 - 1, target * 2 - 1
    f0, f1 = features(sdh, pred), features(sdh, target)
    total = 0
    for k in range(5):
        n0 = f0[k] / (torch.sqrt(torch.sum(f0[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1[k] / (torch.sqrt(torch.sum(f1[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        total = total + F.conv2d((n0 - n1) ** 2, sdh[f'lin{k}.model.1.weight']).mean(dim=(2, 3))
    return total.flatten().float()
