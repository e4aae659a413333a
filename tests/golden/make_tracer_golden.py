"""Writes tests/golden/tracer_ref.npz by EXECUTING the reference's TRACER-B7 modules (lib/models/architecture/tracerb7/*.py and the forward
of lib/models/segmentors/tracer_b7.py) in this container on seeded weights and inputs.  The vendored architecture files import only torch,
so they are loaded as they lie; the wrapper class is taken from its file with `ast` and given stand-ins for the mmcv / mmgen / torchvision
imports its forward does not need (load_checkpoint, the logger; torchvision's Resize / Normalize / resize are restated with the torch
calls they reduce to for tensors with antialias=False).  Nothing is copied into the repo.
Run from the repo root (needs /root/reference):  python tests/golden/make_tracer_golden.py"""
import ast
import importlib.util
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import tracer_oracle as T  # noqa: E402

REF = '/root/reference/lib/models'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tracer_ref.npz')


def load_reference():
    pkg = types.ModuleType('tb7')
    pkg.__path__ = [REF + '/architecture/tracerb7']
    sys.modules['tb7'] = pkg
    for n in ('effi_utils', 'conv_modules', 'efficientnet', 'att_modules', 'tracer'):
        spec = importlib.util.spec_from_file_location('tb7.' + n, f'{REF}/architecture/tracerb7/{n}.py')
        m = importlib.util.module_from_spec(spec)
        sys.modules['tb7.' + n] = m
        spec.loader.exec_module(m)
    from tb7.tracer import TracerDecoder
    from tb7.efficientnet import EfficientEncoderB7

    # torchvision stand-ins: for float tensors, Resize(size, antialias=False) == F.interpolate(bilinear, align_corners=False) and
    # Normalize == (x - mean) / std (torchvision/transforms/functional_tensor.py)
    class Resize:
        def __init__(self, size, antialias=False):
            self.size = tuple(size)

        def __call__(self, x):
            return F.interpolate(x, size=self.size, mode='bilinear', align_corners=False, antialias=False)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean.to(x)) / self.std.to(x)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    transforms = types.SimpleNamespace(Compose=Compose, Resize=Resize, Normalize=Normalize)
    F_t = types.SimpleNamespace(resize=lambda x, size, antialias=False: F.interpolate(x, size=tuple(size), mode='bilinear', align_corners=False, antialias=False))
    tree = ast.parse(open(REF + '/segmentors/tracer_b7.py').read())
    ns = dict(nn=nn, F=F, torch=torch, transforms=transforms, F_t=F_t, TracerDecoder=TracerDecoder, EfficientEncoderB7=EfficientEncoderB7,
              List=__import__('typing').List, Union=__import__('typing').Union, logging=types.SimpleNamespace(ERROR=40), get_root_logger=lambda **k: None,
              load_checkpoint=lambda *a, **k: None)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == 'TracerUniversalB7':
            exec(compile(ast.Module([node], []), REF + '/segmentors/tracer_b7.py', 'exec'), ns)
    return ns['TracerUniversalB7']


def main():
    cls = load_reference()
    sd = T.random_params(seed=11)
    out = {}
    for tag, size, hw, n in (('s192', 192, (150, 170), 2), ('s256', 256, (96, 96), 3)):
        net = cls(input_image_size=size, batch_size=2, torch_dtype='float32', pretrained='none', erosion=1)
        missing, unexpected = net.model.load_state_dict(sd, strict=False)
        assert not unexpected and all(m.endswith('num_batches_tracked') for m in missing), (missing[:4], unexpected[:4])
        g = torch.Generator().manual_seed(5)
        x = torch.rand(n, 3, *hw, generator=g)
        # blob-like content so that the mask is not constant: a bright disc on noise
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, hw[0]), torch.linspace(-1, 1, hw[1]), indexing='ij')
        x = 0.3 * x + 0.7 * ((yy ** 2 + xx ** 2) < 0.4).float()[None, None]
        with torch.no_grad():
            y = net(x)
            img = net.transform(x[:2])
            feats = net.model.encoder(img)
            raw = net.model(img)
        out[f'{tag}_x'] = x.numpy()
        out[f'{tag}_mask'] = y.numpy()
        out[f'{tag}_raw'] = raw.numpy()
        for i, f in enumerate(feats):
            out[f'{tag}_feat{i}'] = f[:, :8].numpy().astype(np.float32)          # first 8 channels of every feature map (size)
            out[f'{tag}_feat{i}_stat'] = np.array([float(f.mean()), float(f.std()), float(f.abs().max())], dtype=np.float64)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
