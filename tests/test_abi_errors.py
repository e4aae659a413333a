"""Error behaviour of the C ABI (host logic only, no GPU): every entry point validates its arguments BEFORE touching the device and
answers a negative status + a message through mve_last_error(), which the Python mirrors turn into MveError -- the counterpart of the
TORCH_CHECK / assert behaviour of the reference's natives and diffusers modules.  Pointers passed here are host buffers that are never
dereferenced, because each call is built to fail its argument checks."""
import ctypes

import pytest
import torch


@pytest.fixture(scope='module')
def L(lib):
    buf = torch.zeros(4096)                 # any non-null address
    return lib, ctypes.c_void_p(buf.data_ptr()), buf


def _bad(lib, name, *args, match=None):
    with pytest.raises(lib.MveError) as e:
        lib.call(name, *args)
    assert match is None or match in str(e.value), str(e.value)


def test_gemm_conv_attention_argument_checks(L):
    lib, p, _ = L
    F16 = 1
    _bad(lib, 'mve_gemm', F16, p, 64, p, 64, p, 64, 16, 12, 64, None, None, 0, 0, None, 0, 0, 1.0, None, 0, 0, None, match='multiples of 8')
    _bad(lib, 'mve_gemm', 0, p, 64, p, 64, p, 64, 16, 16, 64, None, None, 0, 0, None, 0, 0, 1.0, None, 0, 0, None, match='dtype')
    _bad(lib, 'mve_conv3x3', F16, p, 64, None, 0, 1, 8, 8, 3, 0, p, 64, p, 64, None, None, 0, None, 0, 0, 1.0, None, 0, None, match='stride')
    _bad(lib, 'mve_conv3x3', F16, p, 64, None, 0, 1, 8, 8, 1, 0, p, 64, p, 64, None, None, 0, None, 0, 32, 1.0, None, 0, None, match='MVE_CONV_PAD_BR')
    _bad(lib, 'mve_conv3x3', F16, p, 60, None, 0, 1, 8, 8, 1, 0, p, 64, p, 64, None, None, 0, None, 0, 0, 1.0, None, 0, None, match='multiples of 8')
    _bad(lib, 'mve_attention', F16, p, 64, p, 64, p, 64, None, 0, None, 0, p, 64, 1, 16, 16, 0, 2, 48, 0.1, None, match='head dim')
    _bad(lib, 'mve_attention', F16, p, 60, p, 64, p, 64, None, 0, None, 0, p, 64, 1, 16, 16, 0, 2, 40, 0.1, None, match='multiples of 8')
    _bad(lib, 'mve_groupnorm_silu', F16, p, 64, None, 0, 1, 16, 24, 1e-5, p, p, 1, p, p, None, match='divisible')
    _bad(lib, 'mve_softmax_rows', F16, p, 64, 4, 62, p, 64, None, match='multiples of 4')
    _bad(lib, 'mve_layernorm', F16, p, 4096, p, 4096, 4, 4096, p, p, 1e-5, None)
    _bad(lib, 'mve_maxpool2x2', F16, p, 1, 7, 8, 64, p, None, match='even')
    _bad(lib, 'mve_prelu', F16, p, p, 12, p, 120, None, match='multiple of 8')


def test_engine_constructor_checks(L):
    lib, p, _ = L
    h = ctypes.c_void_p()
    arr = lambda *v: (ctypes.c_int * len(v))(*v)
    _bad(lib, 'mve_unet_create', ctypes.byref(h), 1, 4, 4, 2, arr(320, 650), 1, arr(1, 0), arr(8, 8), arr(1, 1), 768, 32, 1e-5, 0, match='incompatible')
    _bad(lib, 'mve_unet_create', ctypes.byref(h), 0, 4, 4, 2, arr(320, 640), 1, arr(1, 0), arr(8, 8), arr(1, 1), 768, 32, 1e-5, 0, match='dtype')
    _bad(lib, 'mve_vae_create', ctypes.byref(h), 1, 3, 4, 3, 2, arr(64, 128), 1, 32, 1e-6, match='half')
    _bad(lib, 'mve_vae_create', ctypes.byref(h), 1, 1, 4, 3, 2, arr(64, 100), 1, 32, 1e-6, match='incompatible')
    _bad(lib, 'mve_srvgg_create', ctypes.byref(h), 1, 3, 1, 64, 2, 4, match='num_out_ch == num_in_ch')
    _bad(lib, 'mve_lpips_create', ctypes.byref(h), 0, 1, match='dtype')
    assert not h.value


def test_engine_call_checks(L):
    """Calls on a handle of the wrong kind, with parameters missing, or with a latent size the topology cannot take."""
    lib, p, _ = L
    arr = lambda *v: (ctypes.c_int * len(v))(*v)
    vae, sr = ctypes.c_void_p(), ctypes.c_void_p()
    lib.call('mve_vae_create', ctypes.byref(vae), 1, 1, 4, 3, 2, arr(64, 128), 1, 32, 1e-6)
    lib.call('mve_srvgg_create', ctypes.byref(sr), 1, 3, 3, 64, 2, 4)
    try:
        ws, n = ctypes.c_size_t(), ctypes.c_int()
        _bad(lib, 'mve_srvgg_plan', vae, 1, 8, 8, 1, ctypes.byref(ws), ctypes.byref(n), None, match='SRVGG')
        _bad(lib, 'mve_vae_plan', sr, 1, 8, 8, 1, ctypes.byref(ws), ctypes.byref(n), None, match='VAE')
        _bad(lib, 'mve_vae_forward', vae, p, 1, 1, 8, 8, p, p, 1 << 20, None, None, match='not loaded')
        _bad(lib, 'mve_vae_plan', vae, 1, 3, 5, 1, ctypes.byref(ws), ctypes.byref(n), None, match='multiple of 8')
        _bad(lib, 'mve_unet_forward', vae, 0, p, 1, p, p, 1, 8, 8, 77, 1, None, None, 0, p, p, 1 << 20, None, None, match='VAE')
        _bad(lib, 'mve_controlnet_forward', vae, p, 1, p, p, p, 1, 8, 8, 77, 1.0, 0, None, p, 1 << 20, None, None, match='ControlNet')
    finally:
        lib.raw('mve_unet_destroy')(vae)
        lib.raw('mve_unet_destroy')(sr)


def test_render_and_mesh_argument_checks(L):
    lib, p, _ = L
    _bad(lib, 'mve_tonemap_lut', p, 16, p, p, 100, 0, 0, p, None, match='steps')
    _bad(lib, 'mve_shade_views', p, p, p, 1, 16, 0.1, 1.0, p, None, 16, p, None, match='both tables')
    _bad(lib, 'mve_x0_prediction', p, p, 0.0, 1.0, 16, p, None, match='bad arguments')
    _bad(lib, 'mve_interpolate_backward_rast', p, 2, 8, 3, p, 3, 4, 4, p, 4, p, p, None, match='bad arguments')
    _bad(lib, 'mve_rasterize_backward', p, 1, 8, p, 4, 4, 4, p, None, p, None, match='null')
    _bad(lib, 'mve_antialias_backward_pos', p, p, 1, 4, 4, 3, p, p, 8, p, 4, None, p, None, match='null')
    _bad(lib, 'mve_lpips_layer', 1, p, p, 2, 16, 60, 0, p, p, None, match='bad arguments')
    assert lib.call('mve_maxpool2x2', 1, p, 0, 8, 8, 64, p, None) == 0          # empty batch: a no-op, not an error


def test_recon_loss_descriptor_checks(L):
    lib, p, _ = L
    from mvedit_amd.recon_loss import _Desc
    raw = lib.raw('mve_recon_loss_workspace_bytes')
    assert raw(0, 128, 0) == 0 and raw(8, 128, 1000) == 4 * (20 * 8 * 128 * 128 + 5 * (2 * 512 + 4))

    def desc(**kw):
        d = _Desc(P=2, ps=8, shaded=1, is_init=0, lut_steps=0, ambient_light=0.2, bg_color=1.0, pixel_loss_weight=1.2, bg_width=0.015, M=0)
        for f in ('d_image', 'd_weights_sum', 'd_depth', 'd_target_dir', 'd_target_rgbs', 'd_target_m', 'd_patch_w', 'd_patch_lights'):
            setattr(d, f, p.value)
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    fwd = lambda d, ws=1 << 20: ('mve_recon_loss_forward', ctypes.byref(d), p, ws, p, p, p, None)
    _bad(lib, *fwd(desc(ps=1)), match='patch geometry')
    _bad(lib, *fwd(desc(d_depth=None)), match='null pointer')
    _bad(lib, *fwd(desc(M=5)), match='no weights')
    _bad(lib, *fwd(desc(lut_steps=16)), match='tone-mapping table')
    _bad(lib, *fwd(desc(bg_width=0.0)), match='bg_width')
    _bad(lib, *fwd(desc(), ws=64), match='workspace')
    _bad(lib, 'mve_recon_loss_backward', ctypes.byref(desc()), p, 64, None, None, None, p, p, p, p, None, match='workspace')
    _bad(lib, 'mve_recon_loss_backward', ctypes.byref(desc()), p, 1 << 20, None, None, None, None, p, p, p, None, match='null output')


def test_mesh_reg_argument_checks(L):
    lib, p, _ = L
    raw = lib.raw('mve_mesh_reg_workspace_bytes')
    assert raw(0, 10) == 0 and raw(1000, 2000) >= 6 * 2000 * 8 + 3 * 1000 * 4 * 2
    _bad(lib, 'mve_mesh_reg_forward', p, 100, p, 200, p, p, 64, p, None, match='workspace')
    _bad(lib, 'mve_mesh_reg_forward', p, 0, p, 200, p, p, 1 << 20, p, None, match='mesh size')
    _bad(lib, 'mve_mesh_reg_forward', p, 100, None, 200, p, p, 1 << 20, p, None, match='null pointer')
    _bad(lib, 'mve_mesh_reg_backward', p, 100, p, 200, p, p, 1 << 20, None, None, p, None, match='null output')
    _bad(lib, 'mve_mesh_normals_forward', p, 100, p, 0, p, p, p, None, match='mesh size')
    _bad(lib, 'mve_mesh_normals_backward', p, 100, p, 200, p, None, None, None, p, None, match='null pointer')


def test_mesh_loss_descriptor_checks(L):
    lib, p, _ = L
    from mvedit_amd.recon_loss import _MeshDesc
    raw = lib.raw('mve_mesh_loss_workspace_bytes')
    assert raw(0, 512) == 0 and raw(6, 512) == 4 * (9 * 6 * 512 * 512 + 6 * 6144)

    def desc(**kw):
        d = _MeshDesc(n=2, size=8, mesh_is_simplified=0, pixel_loss_weight=1.2, normal_reg_weight=1.0)
        for f in ('d_rgba', 'd_normal', 'd_depth', 'd_target_dir', 'd_target_rgbs', 'd_target_m_erode', 'd_target_m_blur', 'd_view_w'):
            setattr(d, f, p.value)
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    _bad(lib, 'mve_mesh_loss_forward', ctypes.byref(desc(size=1)), p, 1 << 20, p, p, p, None, match='view geometry')
    _bad(lib, 'mve_mesh_loss_forward', ctypes.byref(desc(d_depth=None)), p, 1 << 20, p, p, p, None, match='null pointer')
    _bad(lib, 'mve_mesh_loss_forward', ctypes.byref(desc()), p, 64, p, p, p, None, match='workspace')
    _bad(lib, 'mve_mesh_loss_backward', ctypes.byref(desc()), p, 1 << 20, None, None, None, None, p, None, match='null output')


def test_gaussian_blur_argument_checks(L):
    lib, p, buf = L
    q = ctypes.c_void_p(buf.data_ptr() + 4096)
    o = ctypes.c_void_p(buf.data_ptr() + 8192)
    _bad(lib, 'mve_gaussian_blur', p, 1, 64, 64, 30, 5.0, 0, None, 0.0, q, o, None, match='odd')
    _bad(lib, 'mve_gaussian_blur', p, 1, 64, 64, 31, 0.0, 0, None, 0.0, q, o, None, match='sigma')
    _bad(lib, 'mve_gaussian_blur', p, 1, 12, 64, 31, 5.0, 0, None, 0.0, q, o, None, match='reflect padding')
    _bad(lib, 'mve_gaussian_blur', p, 1, 64, 64, 31, 5.0, 0, None, 0.0, p, o, None, match='alias')


def test_sh_encode_argument_checks(L):
    lib, p, _ = L
    _bad(lib, 'mve_sh_encode', p, 16, 9, p, None, None, match='degree')
    _bad(lib, 'mve_sh_encode', p, 16, 0, p, None, None, match='degree')
    _bad(lib, 'mve_sh_encode', None, 16, 4, p, None, None, match='null pointer')
    _bad(lib, 'mve_sh_encode_backward', p, None, 16, 4, p, None, match='null pointer')


def test_shade_points_argument_checks(L):
    lib, p, _ = L
    _bad(lib, 'mve_shade_points', p, p, p, 8, 0.2, None, None, 0, None, None, None, None, None, match='forward needs out')
    _bad(lib, 'mve_shade_points', p, p, p, 8, 0.2, None, None, 0, None, p, None, p, None, match='backward needs')
    _bad(lib, 'mve_shade_points', p, p, p, 8, 0.2, p, None, 16, p, None, None, None, None, match='both tables')
    _bad(lib, 'mve_shade_points', p, None, p, 8, 0.2, None, None, 0, p, None, None, None, None, match='null pointer')
