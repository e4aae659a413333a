"""Zero-edit drop-in under the reference (Lakonik/MVEdit): ONE import line ahead of the reference's own imports.

    import mvedit_amd.dropin; mvedit_amd.dropin.install()        # first line of app.py / of the serving entry point
    from lib.apis.adapter3d import Adapter3DRunner               # ... the reference, unchanged

What `install()` rebinds -- the operator seams of SURVEY.md section 8(b), nothing else of the reference:

  1. `lib.ops.raymarching` and `lib.ops.shencoder` (pybind11 / CUDA extensions, lib/ops/raymarching/src/bindings.cpp:5-19): the module names
     are pre-seeded in `sys.modules` with modules that export the reference's names (`__all__` of lib/ops/raymarching/__init__.py:1-8) bound to
     `mvedit_amd.raymarching` / `mvedit_amd.shencoder`, so `from .raymarching import *` in lib/ops/__init__.py:2 picks them up and the CUDA
     extension is never built or loaded.
  2. `lib.pipelines.adapter3d_mixin.Adapter3DMixin.get_noise_pred / get_noise_pred_p1 / get_noise_pred_p2` (adapter3d_mixin.py:68-317) and
     `lib.models.architecture.diffusers.unet_enc / unet_dec` (diffusers.py:57-164): rebound to this package's mirrors right after those
     modules are executed (a post-import hook on `sys.meta_path`; modules that are already imported are patched at once).
  3. The four classes `lib.pipelines` exports (lib/pipelines/__init__.py:1-7) get their `__init__` wrapped: after the reference's own
     constructor has registered the loaded torch modules (lib/apis/adapter3d.py:971-975), `pipe.unet`, `pipe.controlnet`, `pipe.vae`,
     `pipe.image_enhancer`, `pipe.segmentation` and `pipe.mesh_renderer` are replaced by engines built from those modules' own configs and
     state dicts, and `pipe.nerf.render` by the native renderer.  Engines are cached on the source module (`Adapter3DRunner` builds a pipeline
     object per request from the same loaded modules), and rebuilt when the module's parameters have been replaced.

Nothing here computes: every replacement is one of the engines / mirrors INTEGRATION.md documents seam by seam, and a conversion that fails
raises -- there is no fallback to the torch module.  `uninstall()` restores everything (tests).
"""
import importlib.abc
import importlib.util
import sys
import types

RAYMARCHING_NAMES = ('near_far_from_aabb', 'sph_from_ray', 'morton3D', 'morton3D_invert', 'packbits', 'march_rays_train',
                     'composite_rays_train', 'march_rays', 'composite_rays', 'batch_near_far_from_aabb', 'batch_composite_rays_train')
SHENCODER_NAMES = ('SHEncoder', 'sh_encode')
MIXIN_METHODS = ('get_noise_pred', 'get_noise_pred_p1', 'get_noise_pred_p2')
PIPELINE_CLASSES = {'lib.pipelines.mvedit_3d_pipeline': 'MVEdit3DPipeline',
                    'lib.pipelines.mvedit_texture_pipeline': 'MVEditTexturePipeline',
                    'lib.pipelines.mvedit_texture_superres_pipeline': 'MVEditTextureSuperResPipeline',
                    'lib.pipelines.zero123plus': 'Zero123PlusPipeline'}
SWAPPED_ATTRS = ('unet', 'controlnet', 'vae', 'image_enhancer', 'segmentation', 'mesh_renderer')

_state = dict(installed=False, finder=None, undo=[], seeded=[])


# ---------------------------------------------------------------------------------------------------------------------------------------
# engine construction from the reference's loaded torch modules
# ---------------------------------------------------------------------------------------------------------------------------------------
def _module_dtype_device(m):
    import torch
    p = next(iter(m.parameters()))
    dtype = p.dtype if p.dtype in (torch.float16, torch.bfloat16) else torch.float16
    return dtype, p.device


def _fingerprint(m):
    """Identity of a module's parameter storage: a reloaded / re-typed module gets a new engine."""
    return tuple((k, v.data_ptr(), v._version) for k, v in list(m.state_dict().items())[:4])


def make_unet(m):
    from .unet import UNet2DConditionEngine, config_from_diffusers
    dtype, device = _module_dtype_device(m)
    return UNet2DConditionEngine.from_state_dict(m.state_dict(), config_from_diffusers(m.config), dtype=dtype, device=device)


def make_controlnet(m):
    """`pipe.controlnet` is a diffusers MultiControlNetModel (`.nets`) or one ControlNetModel."""
    from .controlnet import ControlNetEngine, MultiControlNetEngine
    from .unet import config_from_diffusers
    nets = list(m.nets) if hasattr(m, 'nets') else [m]
    engines = []
    for n in nets:
        dtype, device = _module_dtype_device(n)
        engines.append(ControlNetEngine.from_state_dict(n.state_dict(), config_from_diffusers(n.config), dtype=dtype, device=device))
    return MultiControlNetEngine(engines) if hasattr(m, 'nets') else engines[0]


def make_vae(m):
    from .vae import AutoencoderKLEngine
    dtype, device = _module_dtype_device(m)
    return AutoencoderKLEngine.from_state_dict(m.state_dict(), dict(m.config), dtype=dtype, device=device)


def make_image_enhancer(m):
    """SRVGGNetCompact (lib/models/decoders/image_space_ss.py:8-70): body = conv, [act, conv] x num_conv, act?, conv -> pixel shuffle."""
    from .image_enhancer import SRVGGNetCompactEngine
    dtype, device = _module_dtype_device(m)
    eng = SRVGGNetCompactEngine(num_in_ch=m.num_in_ch, num_out_ch=m.num_out_ch, num_feat=m.num_feat, num_conv=m.num_conv, upscale=m.upscale,
                                act_type=getattr(m, 'act_type', 'prelu'), dtype=dtype, device=device)
    return eng.load_state_dict(m.state_dict())


def make_segmentation(m):
    """TracerUniversalB7 (lib/models/segmentors/tracer_b7.py:17-25)."""
    from .segmentor import TracerUniversalB7Engine
    dtype, device = _module_dtype_device(m)
    eng = TracerUniversalB7Engine(input_image_size=getattr(m, 'input_image_size', 640), batch_size=getattr(m, 'batch_size', 8),
                                  torch_dtype=dtype, erosion=getattr(m, 'erosion', 1), device=device)
    eng.load_state_dict(m.state_dict())
    return eng


def make_mesh_renderer(m):
    from .mesh_ops import MeshRenderer
    return MeshRenderer(near=m.near, far=m.far, ssaa=m.ssaa, texture_filter=getattr(m, 'texture_filter', 'linear-mipmap-linear'))


def decoder_params(decoder):
    """iNGPDecoder (lib/models/decoders/ingp_decoder.py:44-120) -> mvedit_amd.nerf.INGPDecoderParams: tcnn's flat hash table as [rows, 2], the two
    MLP layers."""
    from .nerf import INGPDecoderParams
    table = decoder.encoder.params.detach().float().reshape(-1, 2)
    l1, l2 = decoder.mlp.net[0], decoder.mlp.net[1]
    return INGPDecoderParams(table, l1.weight.detach(), l1.bias.detach(), l2.weight.detach(), l2.bias.detach(), n_levels=decoder.n_levels,
                             max_resolution=decoder.max_resolution, bound=getattr(decoder, 'bound', 1.0), blob_density=decoder.blob_density,
                             blob_radius=decoder.blob_radius, sigmoid_saturation=decoder.sigmoid_saturation, device=table.device)


MAKERS = dict(unet=make_unet, controlnet=make_controlnet, vae=make_vae, image_enhancer=make_image_enhancer, segmentation=make_segmentation,
              mesh_renderer=make_mesh_renderer)


def engine_for(kind, module):
    """The engine standing in for `module` (cached on it)."""
    if module is None or getattr(module, '_mve_is_engine', False) or not hasattr(module, 'state_dict') and kind != 'mesh_renderer':
        return module
    fp = _fingerprint(module) if hasattr(module, 'state_dict') else None
    cached = getattr(module, '_mve_engine', None)
    if cached is not None and cached[0] == fp:
        return cached[1]
    eng = MAKERS[kind](module)
    try:
        object.__setattr__(eng, '_mve_is_engine', True)
    except (AttributeError, TypeError):
        pass
    object.__setattr__(module, '_mve_engine', (fp, eng))
    return eng


def _native_nerf_render(nerf):
    """Bound replacement for `BaseNeRF.render` (lib/models/autoencoders/base_nerf.py:489-560) on ONE nerf object."""
    from .nerf import NeRFRenderer

    def render(decoder, code, density_bitfield, h, w, intrinsics, poses, cfg=dict(), bg_color=None, perturb=False, normal_bg=(0.5, 0.5, 1.0)):
        cached = getattr(decoder, '_mve_engine', None)
        fp = _fingerprint(decoder)
        if cached is None or cached[0] != fp:
            cached = (fp, decoder_params(decoder))
            object.__setattr__(decoder, '_mve_engine', cached)
        r = NeRFRenderer(grid_size=getattr(nerf, 'grid_size', 128), bg_color=nerf.bg_color)
        return r.render(cached[1], code, density_bitfield, h, w, intrinsics, poses, cfg=cfg, bg_color=bg_color, perturb=perturb, normal_bg=normal_bg)
    return render


def swap_engines(pipe):
    """After the reference constructor: the seams of one pipeline object."""
    for name in SWAPPED_ATTRS:
        m = getattr(pipe, name, None)
        if m is not None:
            object.__setattr__(pipe, name, engine_for(name, m))       # (object.__setattr__: past DiffusionPipeline's config bookkeeping)
    nerf = getattr(pipe, 'nerf', None)
    if nerf is not None and not getattr(nerf, '_mve_render_bound', False):
        object.__setattr__(nerf, 'render', _native_nerf_render(nerf))
        object.__setattr__(nerf, '_mve_render_bound', True)
    return pipe


# ---------------------------------------------------------------------------------------------------------------------------------------
# patches
# ---------------------------------------------------------------------------------------------------------------------------------------
def _set(obj, name, value):
    had = name in vars(obj) if isinstance(obj, type) else hasattr(obj, name)
    old = vars(obj).get(name) if isinstance(obj, type) else getattr(obj, name, None)
    setattr(obj, name, value)
    _state['undo'].append((obj, name, had, old))


def _patch_mixin(mod):
    from .pipelines import Adapter3DMixin as Ours
    cls = getattr(mod, 'Adapter3DMixin')
    for n in MIXIN_METHODS:
        _set(cls, n, vars(Ours)[n])
    # helpers the mirrors call on `self` / class attributes they read
    for n, v in vars(Ours).items():
        if n.startswith('_') and not n.startswith('__') and n not in vars(cls):
            _set(cls, n, v)
    for n in ('fuse_chunks', 'detect_repeated_cond'):
        if hasattr(Ours, n) and n not in vars(cls):
            _set(cls, n, getattr(Ours, n))


def _patch_arch_diffusers(mod):
    from . import unet as U
    _set(mod, 'unet_enc', U.unet_enc)
    _set(mod, 'unet_dec', U.unet_dec)


def _patch_pipeline(mod):
    cls = getattr(mod, PIPELINE_CLASSES[mod.__name__])
    orig = cls.__init__
    if getattr(orig, '_mve_wrapped', False):
        return

    def __init__(self, *args, **kwargs):
        orig(self, *args, **kwargs)
        swap_engines(self)
    __init__._mve_wrapped = True
    __init__.__wrapped__ = orig
    _set(cls, '__init__', __init__)


PATCHERS = {'lib.pipelines.adapter3d_mixin': _patch_mixin, 'lib.models.architecture.diffusers': _patch_arch_diffusers}
PATCHERS.update({k: _patch_pipeline for k in PIPELINE_CLASSES})


class _PostImport(importlib.abc.MetaPathFinder):
    """Runs the patcher of a target module right after the module has been executed."""

    def find_spec(self, name, path=None, target=None):
        if name not in PATCHERS:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, 'find_spec'):
                continue
            spec = f.find_spec(name, path, target)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, 'exec_module'):
                spec.loader = _Loader(spec.loader, PATCHERS[name])
                return spec
        return None


class _Loader(importlib.abc.Loader):
    def __init__(self, inner, patch):
        self.inner, self.patch = inner, patch

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        self.patch(module)

    def __getattr__(self, n):
        return getattr(self.inner, n)


def _seed_module(name, names, source):
    m = types.ModuleType(name)
    m.__doc__ = f'mvedit_amd.dropin: {source.__name__} under the reference name {name}'
    for n in names:
        setattr(m, n, getattr(source, n))
    m.__all__ = list(names)
    m.__path__ = []           # a package in the reference (lib/ops/raymarching/): submodule imports resolve to nothing rather than to the CUDA build
    m._mve_seeded = True
    _state['seeded'].append((name, sys.modules.get(name)))
    sys.modules[name] = m
    parent = sys.modules.get(name.rpartition('.')[0])
    if parent is not None:
        setattr(parent, name.rpartition('.')[2], m)
        for n in names:       # `from .raymarching import *` already ran in lib/ops/__init__.py: rebind the star-imported names too
            if hasattr(parent, n):
                _set(parent, n, getattr(source, n))


def install():
    """Idempotent.  Call before the reference's modules are imported (already-imported ones are patched in place)."""
    if _state['installed']:
        return
    from . import raymarching, shencoder
    _seed_module('lib.ops.raymarching', RAYMARCHING_NAMES, raymarching)
    _seed_module('lib.ops.shencoder', SHENCODER_NAMES, shencoder)
    for name, patch in PATCHERS.items():
        if name in sys.modules:
            patch(sys.modules[name])
    _state['finder'] = _PostImport()
    sys.meta_path.insert(0, _state['finder'])
    _state['installed'] = True


def uninstall():
    if not _state['installed']:
        return
    if _state['finder'] in sys.meta_path:
        sys.meta_path.remove(_state['finder'])
    for obj, name, had, old in reversed(_state['undo']):
        if had:
            setattr(obj, name, old)
        else:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
    for name, old in reversed(_state['seeded']):
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
    _state.update(installed=False, finder=None, undo=[], seeded=[])
