"""ORACLE -- test infrastructure only: numpy front-end of libdevcore.so, the HOST build of the product's host/device core headers
(mvedit_amd/csrc/raster_grad_core.h, shading_core.h; harness: oracle/devcore_host.cpp).  Same arithmetic as the HIP kernels
mve_rasterize_backward / mve_interpolate_backward_rast / mve_antialias_backward_pos / mve_tonemap_lut / mve_shade_views, sequential."""
import ctypes

import numpy as np

from . import build_devcore

_lib = ctypes.CDLL(build_devcore())
_p, _i, _f, _u, _z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint, ctypes.c_size_t
_lib.dc_interpolate_backward_rast.argtypes = [_p, _i, _i, _i, _p, _i, _i, _i, _p, _i, _p, _p]
_lib.dc_rasterize_backward.argtypes = [_p, _i, _i, _p, _i, _i, _i, _p, _p, _p]
_lib.dc_antialias_backward_pos.argtypes = [_p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p]
_lib.dc_tonemap_lut.argtypes = [_p, _z, _p, _p, _i, _i, _i, _p]
_lib.dc_tonemap_lut_grad.argtypes = [_p, _z, _p, _p, _i, _i, _i, _p]
_lib.dc_tonemap_lut_grad.restype = None
_lib.dc_shade_points.argtypes = [_p, _p, _p, _z, _f, _p, _p, _i, _p, _p, _p, _p]
_lib.dc_shade_points.restype = None
_lib.dc_shade_views.argtypes = [_p, _p, _p, _u, _u, _f, _f, _p, _p, _i, _p]
for f in (_lib.dc_interpolate_backward_rast, _lib.dc_rasterize_backward, _lib.dc_antialias_backward_pos, _lib.dc_tonemap_lut, _lib.dc_shade_views):
    f.restype = None

_d = ctypes.c_double
_lib.dc_recon_loss.argtypes = [_i] * 5 + [_f, _f, _p] + [_f] * 5 + [_p] * 7 + [_i] + [_p] * 9 + [_f] + [_p] * 8
_lib.dc_recon_loss.restype = None

_lib.dc_mesh_loss.argtypes = [_i, _i, _i, _p, _f, _f] + [_p] * 11 + [_f] + [_p] * 6
_lib.dc_mesh_loss.restype = None
_lib.dc_mesh_normals.argtypes = [_p, _i, _p, _i] + [_p] * 7
_lib.dc_mesh_normals.restype = None
_lib.dc_gaussian_blur.argtypes = [_p, _i, _i, _i, _i, _f, _i, _p, _f, _p]
_lib.dc_gaussian_blur.restype = None
_lib.dc_sh_encode.argtypes = [_p, _i, _i, _p, _p]
_lib.dc_sh_encode.restype = None
_lib.dc_mesh_reg.argtypes = [_p, _i, _p, _i, _p, _f, _f, _p, _p, _p, _p]
_lib.dc_mesh_reg.restype = None

_c = lambda a, dt=np.float32: np.ascontiguousarray(a, dt)
_ptr = lambda a: a.ctypes.data_as(_p) if a is not None else None


def interpolate_backward_rast(attr, rast, tri, grad_out):
    attr, rast, tri, grad_out = _c(attr), _c(rast), _c(tri, np.int32), _c(grad_out)
    B, H, W, _ = rast.shape
    out = np.zeros_like(rast)
    _lib.dc_interpolate_backward_rast(_ptr(attr), attr.shape[0], attr.shape[1], attr.shape[2], _ptr(rast), B, H, W, _ptr(tri), tri.shape[0],
                                      _ptr(grad_out), _ptr(out))
    return out


def rasterize_backward(pos, tri, rast, grad_rast):
    pos, rast, tri, grad_rast = _c(pos), _c(rast), _c(tri, np.int32), _c(grad_rast)
    B, H, W, _ = rast.shape
    out = np.zeros_like(pos)
    _lib.dc_rasterize_backward(_ptr(pos), B, pos.shape[1], _ptr(tri), tri.shape[0], H, W, _ptr(rast), _ptr(grad_rast), _ptr(out))
    return out


def antialias_backward_pos(color, rast, pos, tri, opp, grad_out):
    color, rast, pos, tri, opp, grad_out = _c(color), _c(rast), _c(pos), _c(tri, np.int32), _c(opp, np.int32), _c(grad_out)
    B, H, W, C = color.shape
    out = np.zeros_like(pos)
    _lib.dc_antialias_backward_pos(_ptr(color), _ptr(grad_out), B, H, W, C, _ptr(rast), _ptr(pos), pos.shape[1], _ptr(tri), tri.shape[0], _ptr(opp),
                                   _ptr(out))
    return out


def tonemap_lut(x, lut_x, lut_y, inverse=False, linear=False):
    x, lut_x, lut_y = _c(x), _c(lut_x), _c(lut_y)
    out = np.empty_like(x)
    _lib.dc_tonemap_lut(_ptr(x), x.size, _ptr(lut_x), _ptr(lut_y), lut_x.size, int(inverse), int(linear), _ptr(out))
    return out


def tonemap_lut_grad(x, lut_x, lut_y, inverse=False, linear=False):
    x, lut_x, lut_y = _c(x), _c(lut_x), _c(lut_y)
    out = np.empty_like(x)
    _lib.dc_tonemap_lut_grad(_ptr(x), x.size, _ptr(lut_x), _ptr(lut_y), lut_x.size, int(inverse), int(linear), _ptr(out))
    return out


def shade_views(rgba, normal_fg, cam_lights, ambient, bg, lut_x=None, lut_y=None):
    rgba, normal_fg, cam_lights = _c(rgba), _c(normal_fg), _c(cam_lights)
    lx, ly = (_c(lut_x), _c(lut_y)) if lut_x is not None else (None, None)
    b = cam_lights.shape[0]
    n = rgba.size // 4
    out = np.empty(rgba.shape[:-1] + (3,), np.float32)
    _lib.dc_shade_views(_ptr(rgba), _ptr(normal_fg), _ptr(cam_lights), b, n // b, ambient, bg, _ptr(lx), _ptr(ly), lx.size if lx is not None else 0, _ptr(out))
    return out


def recon_loss(image, weights_sum, depth, weights, ts, target_rgbs, target_m_blur, target_dir, patch_w, patch_lights, *, target_n=None,
               target_depth=None, lut_x=None, lut_y=None, shaded=True, is_init=False, ambient_light=0.2, bg_color=1.0, normal_bg=(0.5, 0.5, 1.0),
               pixel_loss_weight=1.2, normal_reg_weight=0.0, depth_weight=0.0, entropy_weight=0.0, bg_width=0.015, g_rgb_ext=None,
               g_nrm_ext=None, gl=1.0):
    """Host run of recon_loss_core.h in the order recon_loss.hip launches it.  Shapes as oracle/recon_loss_oracle.nerf_optim_loss;
    ts [M, 2].  Returns dict(losses[6] = total, rgb, alpha, tv, depth, entropy; out_rgbs, out_normals, g_image, g_weights_sum, g_depth, g_weights)."""
    P, ps = target_rgbs.shape[:2]
    N = P * ps * ps
    f = lambda a: None if a is None else _c(a)
    image, alpha, depth, weights, ts = _c(image), _c(weights_sum), _c(depth), _c(weights), _c(ts)
    arrs = [f(target_dir), f(target_rgbs), f(target_m_blur), f(target_n), f(target_depth), f(patch_w), f(patch_lights), f(g_rgb_ext), f(g_nrm_ext)]
    lx, ly = f(lut_x), f(lut_y)
    nbg = _c(normal_bg)
    ws = np.zeros(20 * N, np.float32)
    losses = np.zeros(6, np.float64)
    out = dict(out_rgbs=np.zeros((P, ps, ps, 3), np.float32), out_normals=np.zeros((P, ps, ps, 3), np.float32), g_image=np.zeros((N, 3), np.float32),
               g_weights_sum=np.zeros(N, np.float32), g_depth=np.zeros(N, np.float32), g_weights=np.zeros(weights.size, np.float32))
    _lib.dc_recon_loss(P, ps, int(shaded), int(is_init), 0 if lx is None else lx.size, ambient_light, bg_color, _ptr(nbg), pixel_loss_weight,
                       normal_reg_weight, depth_weight, entropy_weight, bg_width, _ptr(lx), _ptr(ly), _ptr(image), _ptr(alpha), _ptr(depth),
                       _ptr(weights), _ptr(ts), weights.size, *[_ptr(a) for a in arrs], gl, _ptr(ws), losses.ctypes.data_as(_p),
                       *[_ptr(out[k]) for k in ('out_rgbs', 'out_normals', 'g_image', 'g_weights_sum', 'g_depth', 'g_weights')])
    out['losses'] = losses
    return out


def mesh_reg(verts, faces, face_normals, gl_lap=1.0, gl_nc=1.0):
    """Host run of mesh_reg_core.h in the order mesh_reg.hip launches it -> dict(losses[2] = laplacian_smooth_loss, normal_consistency;
    n_edges; g_verts [V, 3]; g_face_normals [F, 3])."""
    verts, faces, face_normals = _c(verts), _c(faces, np.int32), _c(face_normals)
    losses, ne = np.zeros(2, np.float64), ctypes.c_int(0)
    g_v, g_fn = np.zeros_like(verts), np.zeros_like(face_normals)
    _lib.dc_mesh_reg(_ptr(verts), verts.shape[0], _ptr(faces), faces.shape[0], _ptr(face_normals), gl_lap, gl_nc, losses.ctypes.data_as(_p),
                     ctypes.cast(ctypes.byref(ne), _p), _ptr(g_v), _ptr(g_fn))
    return dict(losses=losses, n_edges=ne.value, g_verts=g_v, g_face_normals=g_fn)


def mesh_loss(rgba, normal, depth, target_rgbs, target_m_erode, target_m_blur, target_dir, view_w, *, target_n=None, simplified=False,
              normal_bg=(0.5, 0.5, 1.0), pixel_loss_weight=1.2, normal_reg_weight=0.0, g_rgb_ext=None, g_nrm_ext=None, gl=1.0):
    """Host run of mesh_loss_core.h in recon_loss.hip's launch order; shapes as oracle/recon_loss_oracle.mesh_optim_loss.
    Returns dict(losses[4] = total, rgb, alpha, tv; out_rgbs, out_normals, g_rgba, g_normal)."""
    n, S = rgba.shape[:2]
    N = n * S * S
    f = lambda a: None if a is None else _c(a)
    arrs = [f(rgba), f(normal), f(depth), f(target_dir), f(target_rgbs), f(target_m_erode), f(target_m_blur), f(target_n), f(view_w), f(g_rgb_ext),
            f(g_nrm_ext)]
    nbg = _c(normal_bg)
    ws = np.zeros(9 * N, np.float32)
    losses = np.zeros(4, np.float64)
    out = dict(out_rgbs=np.zeros((n, S, S, 3), np.float32), out_normals=np.zeros((n, S, S, 3), np.float32), g_rgba=np.zeros((n, S, S, 4), np.float32),
               g_normal=np.zeros((n, S, S, 3), np.float32))
    _lib.dc_mesh_loss(n, S, int(simplified), _ptr(nbg), pixel_loss_weight, normal_reg_weight, *[_ptr(a) for a in arrs], gl, _ptr(ws),
                      losses.ctypes.data_as(_p), *[_ptr(out[k]) for k in ('out_rgbs', 'out_normals', 'g_rgba', 'g_normal')])
    out['losses'] = losses
    return out


def mesh_normals(verts, faces, g_vn=None, g_face_normals=None):
    """Host run of the auto_normal functions of mesh_reg_core.h -> dict(face_normals [F, 3], vn [V, 3], g_verts [V, 3])."""
    verts, faces = _c(verts), _c(faces, np.int32)
    f = lambda a: None if a is None else _c(a)
    g_vn, g_fn = f(g_vn), f(g_face_normals)
    fn, vs, vn = np.zeros((faces.shape[0], 3), np.float32), np.zeros_like(verts), np.zeros_like(verts)
    gs, gv = np.zeros_like(verts), np.zeros_like(verts)
    _lib.dc_mesh_normals(_ptr(verts), verts.shape[0], _ptr(faces), faces.shape[0], _ptr(g_vn), _ptr(g_fn), _ptr(fn), _ptr(vs), _ptr(vn), _ptr(gs), _ptr(gv))
    return dict(face_normals=fn, vn=vn, g_verts=gv)


def gaussian_blur(x, ksize, sigma, adjoint=False, base=None, offset=0.0):
    """Host run of blur_core.h: x [..., H, W] -> blur (or its adjoint); base given: offset + base - blur(x) (highpass and its backward)."""
    x = _c(x)
    H, W = x.shape[-2:]
    out = np.empty_like(x)
    b = None if base is None else _c(base)
    _lib.dc_gaussian_blur(_ptr(x), x.size // (H * W), H, W, int(ksize), float(sigma), int(adjoint), _ptr(b), float(offset), _ptr(out))
    return out


def sh_encode(xyz, degree, jacobian=False):
    """Host run of sh_core.h -> out [B, degree^2] (and jac [B, 3, degree^2])"""
    xyz = _c(xyz)
    B = xyz.shape[0]
    out = np.zeros((B, degree * degree), np.float32)
    jac = np.zeros((B, 3, degree * degree), np.float32) if jacobian else None
    _lib.dc_sh_encode(_ptr(xyz), B, int(degree), _ptr(out), _ptr(jac))
    return (out, jac) if jacobian else out


def shade_points(albedo, normal, lights, ambient, lut_x=None, lut_y=None, grad_out=None):
    """Host run of sh_shade_point -> out [N, 3], or (g_albedo, g_normal) when grad_out is given"""
    albedo, normal, lights = _c(albedo), _c(normal), _c(lights)
    lx, ly = (_c(lut_x), _c(lut_y)) if lut_x is not None else (None, None)
    N = albedo.shape[0]
    if grad_out is None:
        out = np.empty_like(albedo)
        _lib.dc_shade_points(_ptr(albedo), _ptr(normal), _ptr(lights), N, ambient, _ptr(lx), _ptr(ly), 0 if lx is None else lx.size, _ptr(out), None, None, None)
        return out
    g = _c(grad_out)
    ga, gn = np.empty_like(albedo), np.empty_like(albedo)
    _lib.dc_shade_points(_ptr(albedo), _ptr(normal), _ptr(lights), N, ambient, _ptr(lx), _ptr(ly), 0 if lx is None else lx.size, None, _ptr(g), _ptr(ga), _ptr(gn))
    return ga, gn
