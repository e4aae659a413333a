"""`mvedit_amd.dropin.install()`: the zero-edit drop-in under the reference.

The reference cannot be imported here (diffusers / mmcv / tinycudann / nvdiffrast are absent), so the hook is exercised on a SKELETON of the
reference's package -- same module paths, class and function names, import statements and constructor keyword names, bodies that record what they
were given -- written to a temporary directory, imported as `lib`, and driven the way lib/apis/adapter3d.py:971-975 drives the real one.  Where the
reference tree is present (the build container) a second test checks with `ast` that every module path, name and constructor keyword the hook
relies on exists there, so the skeleton cannot drift from the real thing; and the rebound `get_noise_pred` is run on the stand-in networks against
the golden output of the REFERENCE's own method (tests/golden/mixin_ref.npz, tests/golden/make_mixin_golden.py).  Host logic only: no GPU."""
import ast
import os
import sys
import textwrap

import numpy as np
import pytest
import torch

import stubs

SKELETON = {
    'lib/__init__.py': '',
    'lib/ops/__init__.py': 'from .raymarching import *\nfrom .shencoder import *\n',
    # the real packages build / load CUDA extensions on import: the seeded modules must keep these files from ever being executed
    'lib/ops/raymarching/__init__.py': 'raise ImportError("CUDA extension: must not be imported under mvedit_amd.dropin")\n',
    'lib/ops/shencoder/__init__.py': 'raise ImportError("CUDA extension: must not be imported under mvedit_amd.dropin")\n',
    'lib/models/__init__.py': '', 'lib/models/architecture/__init__.py': '',
    'lib/models/architecture/diffusers.py': 'def unet_enc(*a, **k):\n    raise RuntimeError("reference unet_enc")\n\ndef unet_dec(*a, **k):\n    raise RuntimeError("reference unet_dec")\n',
    'lib/pipelines/__init__.py': ('from .mvedit_3d_pipeline import MVEdit3DPipeline\nfrom .mvedit_texture_pipeline import MVEditTexturePipeline\n'
                                  'from .mvedit_texture_superres_pipeline import MVEditTextureSuperResPipeline\nfrom .zero123plus import Zero123PlusPipeline\n'),
    'lib/pipelines/adapter3d_mixin.py': textwrap.dedent('''
        from lib.models.architecture.diffusers import unet_enc, unet_dec
        class Adapter3DMixin:
            def get_noise_pred(self, *a, **k):
                raise RuntimeError("reference get_noise_pred")
            def get_noise_pred_p1(self, *a, **k):
                raise RuntimeError("reference get_noise_pred_p1")
            def get_noise_pred_p2(self, *a, **k):
                raise RuntimeError("reference get_noise_pred_p2")
            def load_cam_weights(self):
                return "untouched"
        '''),
}
# constructor keywords and bases as in the reference (checked against it by the last test)
PIPELINES = [('mvedit_3d_pipeline', 'MVEdit3DPipeline', 'Adapter3DMixin',
              ['vae', 'text_encoder', 'tokenizer', 'unet', 'controlnet', 'scheduler', 'nerf', 'mesh_renderer', 'image_enhancer', 'segmentation', 'normal_model', 'tonemapping']),
             ('mvedit_texture_pipeline', 'MVEditTexturePipeline', 'MVEdit3DPipeline',
              ['vae', 'text_encoder', 'tokenizer', 'unet', 'controlnet', 'scheduler', 'nerf', 'mesh_renderer']),
             ('mvedit_texture_superres_pipeline', 'MVEditTextureSuperResPipeline', 'MVEditTexturePipeline',
              ['vae', 'text_encoder', 'tokenizer', 'unet', 'controlnet', 'scheduler', 'nerf', 'mesh_renderer']),
             ('zero123plus', 'Zero123PlusPipeline', 'object',
              ['vae', 'text_encoder', 'tokenizer', 'unet', 'scheduler', 'vision_encoder', 'feature_extractor_clip', 'feature_extractor_vae', 'ramping_coefficients', 'safety_checker'])]
_IMPORTS = {'Adapter3DMixin': 'from .adapter3d_mixin import Adapter3DMixin\n', 'MVEdit3DPipeline': 'from .mvedit_3d_pipeline import MVEdit3DPipeline\n',
            'MVEditTexturePipeline': 'from .mvedit_texture_pipeline import MVEditTexturePipeline\n', 'object': ''}
for _mod, _cls, _base, _names in PIPELINES:
    SKELETON[f'lib/pipelines/{_mod}.py'] = _IMPORTS[_base] + f'class {_cls}({_base}):\n    def __init__(self, {", ".join(n + "=None" for n in _names)}):\n' + \
        ''.join(f'        self.{n} = {n}\n' for n in _names) + ('        self.bg_color = getattr(nerf, "bg_color", None)\n' if 'nerf' in _names else '')


class FakeModule(torch.nn.Module):
    """A loaded torch module of the reference: parameters + the attributes the makers read."""

    def __init__(self, **attrs):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(4, dtype=torch.float16))
        for k, v in attrs.items():
            setattr(self, k, v)


@pytest.fixture()
def skeleton(tmp_path, monkeypatch):
    for rel, src in SKELETON.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(src)
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.')]:
        monkeypatch.delitem(sys.modules, k)
    from mvedit_amd import dropin
    yield dropin
    dropin.uninstall()
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.')]:
        sys.modules.pop(k, None)


def test_install_rebinds_the_operator_seams(lib, skeleton):
    dropin = skeleton
    from mvedit_amd import raymarching, shencoder, unet
    from mvedit_amd.pipelines import Adapter3DMixin as Ours
    dropin.install()
    dropin.install()                                            # idempotent
    import lib as ref                                           # noqa: F401  (the skeleton)
    import lib.ops as ops
    for n in dropin.RAYMARCHING_NAMES:                          # lib/ops/__init__.py:2 `from .raymarching import *`
        assert getattr(ops, n) is getattr(raymarching, n), n
    assert ops.SHEncoder is shencoder.SHEncoder
    from lib.ops.raymarching import march_rays_train            # the spelling of lib/models/decoders/base_volume_renderer.py
    assert march_rays_train is raymarching.march_rays_train
    from lib.pipelines import MVEdit3DPipeline, Zero123PlusPipeline
    from lib.pipelines.adapter3d_mixin import Adapter3DMixin
    import lib.models.architecture.diffusers as arch
    assert arch.unet_enc is unet.unet_enc and arch.unet_dec is unet.unet_dec
    for n in dropin.MIXIN_METHODS:
        assert vars(Adapter3DMixin)[n] is vars(Ours)[n], n
    assert Adapter3DMixin().load_cam_weights() == 'untouched'  # nothing else of the class moves
    assert MVEdit3DPipeline.__init__._mve_wrapped and Zero123PlusPipeline.__init__._mve_wrapped
    dropin.uninstall()
    with pytest.raises(RuntimeError, match='reference get_noise_pred'):
        Adapter3DMixin().get_noise_pred()
    assert not getattr(MVEdit3DPipeline.__init__, '_mve_wrapped', False)


def test_install_after_the_reference_was_imported(lib, skeleton, monkeypatch):
    """Modules that are already in sys.modules are patched in place (install() called late)."""
    dropin = skeleton
    from mvedit_amd import raymarching, shencoder
    for name, names, src in (('lib.ops.raymarching', dropin.RAYMARCHING_NAMES, raymarching), ('lib.ops.shencoder', dropin.SHENCODER_NAMES, shencoder)):
        m = type(sys)(name)                                     # stands for the reference's CUDA-backed module, already imported
        for n in names:
            setattr(m, n, lambda *a, **k: 'cuda')
        m.__all__, m.__path__ = list(names), []
        monkeypatch.setitem(sys.modules, name, m)
    import lib.ops as ops
    import lib.pipelines.adapter3d_mixin as M
    assert ops.packbits() == 'cuda'
    dropin.install()
    assert ops.packbits is raymarching.packbits and sys.modules['lib.ops.raymarching'].packbits is raymarching.packbits
    from mvedit_amd.pipelines import Adapter3DMixin as Ours
    assert vars(M.Adapter3DMixin)['get_noise_pred'] is vars(Ours)['get_noise_pred']


def test_pipeline_construction_swaps_the_loaded_modules_for_engines(lib, skeleton, monkeypatch):
    """lib/apis/adapter3d.py:971-975: a pipeline object per request from the SAME loaded modules -- engines are built once per module."""
    dropin = skeleton
    built = []

    def maker(kind):
        def make(m):
            built.append(kind)
            return ('engine', kind, id(m))
        return make
    monkeypatch.setattr(dropin, 'MAKERS', {k: maker(k) for k in dropin.SWAPPED_ATTRS})
    monkeypatch.setattr(dropin, '_native_nerf_render', lambda nerf: (lambda *a, **k: ('native render', a[3:5])))
    dropin.install()
    from lib.pipelines import MVEdit3DPipeline
    mods = dict(vae=FakeModule(), unet=FakeModule(), controlnet=FakeModule(nets=[FakeModule(), FakeModule()]), image_enhancer=FakeModule(),
                segmentation=FakeModule(), mesh_renderer=FakeModule(near=0.1, far=10, ssaa=1))
    nerf = FakeModule(bg_color=1.0, grid_size=128)
    pipe = MVEdit3DPipeline(text_encoder='te', tokenizer='tok', scheduler='sch', nerf=nerf, normal_model=None, tonemapping='tm', **mods)
    for k, m in mods.items():
        assert getattr(pipe, k) == ('engine', k, id(m)), k
    assert pipe.text_encoder == 'te' and pipe.scheduler == 'sch' and pipe.nerf is nerf          # not seams: untouched
    assert pipe.nerf.render('dec', None, 'bits', 64, 48, 'K', 'P') == ('native render', (64, 48))
    assert sorted(built) == sorted(dropin.SWAPPED_ATTRS)
    pipe2 = MVEdit3DPipeline(nerf=nerf, **mods)                                                # the next request: cached engines
    assert sorted(built) == sorted(dropin.SWAPPED_ATTRS) and pipe2.unet == pipe.unet
    with torch.no_grad():
        mods['unet'].w.add_(1)                                                                 # weights edited in place (LoRA merge, reload): rebuilt
    MVEdit3DPipeline(nerf=nerf, **mods)
    assert built.count('unet') == 2 and built.count('vae') == 1


def test_rebound_get_noise_pred_reproduces_the_reference_method(lib, skeleton):
    """The rebound method, called the way the pipelines call it, on the stand-in networks: equal to what the reference's OWN method returned
    (golden written by executing lib/pipelines/adapter3d_mixin.py:68-135 over the same stand-ins)."""
    dropin = skeleton
    dropin.install()
    from lib.pipelines import MVEdit3DPipeline
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mixin_ref.npz'))
    pipe = MVEdit3DPipeline(unet=stubs.StubUNet(), controlnet=stubs.StubControlNet())
    for name, kw in stubs.cases().items():
        with torch.no_grad():
            out = pipe.get_noise_pred(**kw)
        np.testing.assert_allclose(out.numpy(), G[name], rtol=1e-5, atol=1e-6)


REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present (GPU box)')
def test_the_names_the_hook_relies_on_exist_in_the_reference():
    from mvedit_amd import dropin

    def tree(rel):
        return ast.parse(open(os.path.join(REF, rel)).read())

    def assigned_all(t):
        for n in t.body:
            if isinstance(n, ast.Assign) and any(isinstance(x, ast.Name) and x.id == '__all__' for x in n.targets):
                return [e.value for e in n.value.elts]
    assert assigned_all(tree('lib/ops/raymarching/__init__.py')) == list(dropin.RAYMARCHING_NAMES)
    star = [n.module for n in tree('lib/ops/__init__.py').body if isinstance(n, ast.ImportFrom) and n.names[0].name == '*']
    assert 'raymarching' in star and 'shencoder' in star
    assert 'SHEncoder' in [a.name for n in tree('lib/ops/shencoder/__init__.py').body if isinstance(n, ast.ImportFrom) for a in n.names]
    mix = [n for n in tree('lib/pipelines/adapter3d_mixin.py').body if isinstance(n, ast.ClassDef) and n.name == 'Adapter3DMixin'][0]
    assert set(dropin.MIXIN_METHODS) <= {f.name for f in mix.body if isinstance(f, ast.FunctionDef)}
    assert {'unet_enc', 'unet_dec'} <= {f.name for f in tree('lib/models/architecture/diffusers.py').body if isinstance(f, ast.FunctionDef)}
    exported = assigned_all(tree('lib/pipelines/__init__.py'))
    assert sorted(exported) == sorted(dropin.PIPELINE_CLASSES.values())
    for mod, cls, base, names in PIPELINES:
        t = tree(f'lib/pipelines/{mod}.py')
        c = [n for n in t.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
        init = [f for f in c.body if isinstance(f, ast.FunctionDef) and f.name == '__init__'][0]
        assert [a.arg for a in init.args.args][1:] == names, (cls, [a.arg for a in init.args.args])
        assert dropin.PIPELINE_CLASSES[f'lib.pipelines.{mod}'] == cls
        assert base == 'object' or base in [ast.unparse(b) for b in c.bases], (cls, [ast.unparse(b) for b in c.bases])
    # the attributes the makers read off the loaded modules
    src = open(os.path.join(REF, 'lib/models/decoders/image_space_ss.py')).read()
    for a in ('num_in_ch', 'num_out_ch', 'num_feat', 'num_conv', 'upscale', 'act_type'):
        assert f'self.{a} = {a}' in src, a
    src = open(os.path.join(REF, 'lib/models/decoders/mesh_renderer/base_mesh_renderer.py')).read()
    for a in ('near', 'far', 'ssaa', 'texture_filter'):
        assert f'self.{a} = {a}' in src, a
    src = open(os.path.join(REF, 'lib/models/decoders/ingp_decoder.py')).read()
    for a in ('self.encoder = tcnn.Encoding', 'self.mlp = MLP', 'self.n_levels = n_levels', 'self.max_resolution = max_resolution', 'self.blob_density', 'self.blob_radius',
              'self.sigmoid_saturation'):
        assert a in src, a
