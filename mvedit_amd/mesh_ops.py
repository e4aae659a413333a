"""Mirrors of the reference's texture-space helpers used by the mesh path (lib/ops/edge_dilation.py)."""
import torch

from . import _lib



def _inference_only(name, *tensors):
    """Ops whose reference counterparts are differentiable torch expressions but which the optimisation loops never differentiate
    (ssaa = 1 and dilate_edges = 0 there): refuse loudly instead of silently cutting the graph."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(f'{name}: native forward only -- no backward is built for it (call it under torch.no_grad())')


def edge_dilation(img, mask, radius=3, iters=7):
    """Same signature and result as lib.ops.edge_dilation.edge_dilation: img (n,c,h,w), mask (n,1,h,w) -> dilated img."""
    if radius == 0 or iters == 0:
        return img
    _inference_only('edge_dilation', img, mask)
    assert img.is_cuda and mask.is_cuda and img.dim() == 4 and mask.shape[1] == 1
    dtype = img.dtype
    x = img.float().contiguous()
    m = mask.float().contiguous()
    n, c, h, w = x.shape
    out, mout = torch.empty_like(x), torch.empty_like(m)
    tmp, mtmp = (torch.empty_like(x), torch.empty_like(m)) if iters > 1 else (None, None)
    with torch.cuda.device(x.device):
        _lib.call('mve_edge_dilation', _lib.ptr(x), _lib.ptr(m), n, c, h, w, float(radius), int(iters), _lib.ptr(out),
                  _lib.ptr(mout), _lib.ptr(tmp), _lib.ptr(mtmp), _lib.stream_ptr(x.device))
    return out.to(dtype)


class _RasterizeFn(torch.autograd.Function):
    """Geometry gradient of the rasteriser: d (u, v, z/w) / d clip-space vertices with every pixel's triangle held fixed."""

    @staticmethod
    def forward(ctx, pos, tri, h, w):
        rast = _rasterize_raw(pos, tri, (h, w))
        ctx.save_for_backward(pos.float().contiguous(), tri.to(torch.int32).contiguous(), rast)
        return rast

    @staticmethod
    def backward(ctx, g):
        pos, tri, rast = ctx.saved_tensors
        B, V, _ = pos.shape
        _, h, w, _ = rast.shape
        g_pos = torch.zeros_like(pos)
        g = g.float().contiguous()
        with torch.cuda.device(pos.device):
            _lib.call('mve_rasterize_backward', _lib.ptr(pos), B, V, _lib.ptr(tri), tri.shape[0], h, w, _lib.ptr(rast), _lib.ptr(g), _lib.ptr(g_pos),
                      _lib.stream_ptr(pos.device))
        return g_pos, None, None, None


def rasterize(pos, tri, resolution):
    """dr.rasterize(glctx, pos, tri, (h, w)) -> rast [B,h,w,4] = (u, v, z/w, triangle_id+1).  pos [B,V,4] clip space, tri [F,3] int32.
    Differentiable w.r.t. pos (through u, v, z/w) when pos requires grad."""
    if torch.is_grad_enabled() and pos.requires_grad:
        return _RasterizeFn.apply(pos, tri, int(resolution[0]), int(resolution[1]))
    return _rasterize_raw(pos, tri, resolution)


def _rasterize_raw(pos, tri, resolution):
    h, w = resolution
    pos = pos.float().contiguous()
    tri = tri.to(torch.int32).contiguous()
    B, V, _ = pos.shape
    F = tri.shape[0]
    rast = torch.empty(B, h, w, 4, dtype=torch.float32, device=pos.device)
    nbytes = _lib.raw('mve_rasterize_workspace_bytes')(B, h, w, F)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pos.device)
    with torch.cuda.device(pos.device):
        _lib.call('mve_rasterize', _lib.ptr(pos), B, V, _lib.ptr(tri), F, h, w, _lib.ptr(rast), _lib.ptr(ws), nbytes,
                  _lib.stream_ptr(pos.device))
    return rast


class _InterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        ctx.save_for_backward(rast, tri, attr)
        ctx.attr_shape = tuple(attr.shape)
        return _interpolate_raw(attr, rast, tri)

    @staticmethod
    def backward(ctx, g):
        rast, tri, attr = ctx.saved_tensors
        Ba, V, A = ctx.attr_shape
        B, h, w, _ = rast.shape
        g_attr = torch.zeros(Ba, V, A, dtype=torch.float32, device=rast.device)
        g = g.float().contiguous()
        with torch.cuda.device(rast.device):
            _lib.call('mve_interpolate_backward', _lib.ptr(g), Ba, V, A, _lib.ptr(rast), B, h, w, _lib.ptr(tri), tri.shape[0], _lib.ptr(g_attr),
                      _lib.stream_ptr(rast.device))
        g_rast = None
        if ctx.needs_input_grad[1]:            # d out / d (u, v): the geometry path (rast came from a differentiable rasterize)
            g_rast = torch.empty_like(rast)
            a = attr.float().contiguous()
            with torch.cuda.device(rast.device):
                _lib.call('mve_interpolate_backward_rast', _lib.ptr(a), Ba, V, A, _lib.ptr(rast), B, h, w, _lib.ptr(tri), tri.shape[0], _lib.ptr(g),
                          _lib.ptr(g_rast), _lib.stream_ptr(rast.device))
        return g_attr, g_rast, None


def interpolate(attr, rast, tri):
    """dr.interpolate(attr, rast, tri)[0]: attr [1 or B, V, A] -> [B,h,w,A].  Differentiable w.r.t. attr and, when rast comes from a
    differentiable `rasterize`, w.r.t. its (u, v)."""
    if torch.is_grad_enabled() and (attr.requires_grad or rast.requires_grad):
        return _InterpolateFn.apply(attr.float().contiguous(), rast.contiguous(), tri.to(torch.int32).contiguous())
    return _interpolate_raw(attr, rast, tri)


def _interpolate_raw(attr, rast, tri):
    attr = attr.detach().float().contiguous()
    tri = tri.to(torch.int32).contiguous()
    B, h, w, _ = rast.shape
    out = torch.empty(B, h, w, attr.shape[-1], dtype=torch.float32, device=rast.device)
    with torch.cuda.device(rast.device):
        _lib.call('mve_interpolate', _lib.ptr(attr), attr.shape[0], attr.shape[1], attr.shape[2], _lib.ptr(rast.contiguous()), B, h, w,
                  _lib.ptr(tri), tri.shape[0], _lib.ptr(out), _lib.stream_ptr(rast.device))
    return out


def edge_opposites(tri):
    """opp [F,3] int32: vertex opposite to edge e of every triangle in the adjacent triangle (-1: boundary / non-manifold)."""
    tri = tri.to(torch.int32).contiguous()
    F = tri.shape[0]
    opp = torch.empty_like(tri)
    nbytes = _lib.raw('mve_edge_opposites_workspace_bytes')(F)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=tri.device)
    with torch.cuda.device(tri.device):
        _lib.call('mve_edge_opposites', _lib.ptr(tri), F, _lib.ptr(opp), _lib.ptr(ws), nbytes, _lib.stream_ptr(tri.device))
    return opp


class _AntialiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp):
        ctx.save_for_backward(rast, pos, tri, opp)
        ctx.color = color.detach() if ctx.needs_input_grad[2] else None        # the silhouette gradient needs the blended colours
        return _antialias_raw(color, rast, pos, tri, opp)

    @staticmethod
    def backward(ctx, g):
        rast, pos, tri, opp = ctx.saved_tensors
        g = g.float().contiguous()
        B, h, w, C = g.shape
        g_in = torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.call('mve_antialias_backward', _lib.ptr(g), B, h, w, C, _lib.ptr(rast), _lib.ptr(pos), pos.shape[1], _lib.ptr(tri), tri.shape[0],
                      _lib.ptr(opp), _lib.ptr(g_in), _lib.stream_ptr(g.device))
        g_pos = None
        if ctx.needs_input_grad[2]:            # d out / d clip-space vertices of the crossed silhouette edges
            g_pos = torch.zeros_like(pos)
            with torch.cuda.device(g.device):
                _lib.call('mve_antialias_backward_pos', _lib.ptr(ctx.color), _lib.ptr(g), B, h, w, C, _lib.ptr(rast), _lib.ptr(pos), pos.shape[1],
                          _lib.ptr(tri), tri.shape[0], _lib.ptr(opp), _lib.ptr(g_pos), _lib.stream_ptr(g.device))
        return g_in, None, g_pos, None, None


def antialias(color, rast, pos, tri, opp=None):
    """dr.antialias(color, rast, pos, tri): color [B,h,w,C] -> same shape (rules: oracle/raster_oracle.c).  Differentiable w.r.t. color
    and, when pos requires grad, w.r.t. the clip-space vertices of the silhouette edges."""
    pos_grad = torch.is_grad_enabled() and pos.requires_grad
    rast = rast.detach().contiguous()
    pos = pos.float().contiguous() if pos_grad else pos.detach().float().contiguous()
    tri = tri.to(torch.int32).contiguous()
    opp = edge_opposites(tri) if opp is None else opp
    if torch.is_grad_enabled() and (color.requires_grad or pos_grad):
        return _AntialiasFn.apply(color.float().contiguous(), rast, pos, tri, opp)
    return _antialias_raw(color, rast, pos, tri, opp)


def _antialias_raw(color, rast, pos, tri, opp):
    color = color.detach().float().contiguous()
    B, h, w, C = color.shape
    out = torch.empty_like(color)
    with torch.cuda.device(color.device):
        _lib.call('mve_antialias', _lib.ptr(color), B, h, w, C, _lib.ptr(rast), _lib.ptr(pos), pos.shape[1], _lib.ptr(tri), tri.shape[0],
                  _lib.ptr(opp), _lib.ptr(out), _lib.stream_ptr(color.device))
    return out


def rasterize_db(pos, tri, rast):
    """The second output of dr.rasterize: barycentric pixel differentials (du/dX, du/dY, dv/dX, dv/dY) [B,h,w,4] of `rast` (no gradient:
    every reference call site that uses it for textures passes grad_db=False or never differentiates it)."""
    pos, tri, rast = pos.detach().float().contiguous(), tri.to(torch.int32).contiguous(), rast.detach().contiguous()
    B, V, _ = pos.shape
    _, h, w, _ = rast.shape
    out = torch.empty_like(rast)
    with torch.cuda.device(pos.device):
        _lib.call('mve_rasterize_db', _lib.ptr(pos), B, V, _lib.ptr(tri), tri.shape[0], _lib.ptr(rast), h, w, _lib.ptr(out), _lib.stream_ptr(pos.device))
    return out


def interpolate_da(attr, rast, rast_db, tri):
    """The second output of dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs='all'): [B,h,w,2C] = (dA0/dX, dA0/dY, dA1/dX, ...)."""
    attr = attr.detach().float().contiguous()
    if attr.dim() == 2:
        attr = attr[None]
    rast, rast_db, tri = rast.detach().contiguous(), rast_db.contiguous(), tri.to(torch.int32).contiguous()
    B, h, w, _ = rast.shape
    Ba, V, C = attr.shape
    out = torch.empty(B, h, w, 2 * C, dtype=torch.float32, device=rast.device)
    with torch.cuda.device(rast.device):
        _lib.call('mve_interpolate_da', _lib.ptr(attr), Ba, V, C, _lib.ptr(rast), _lib.ptr(rast_db), B, h * w, _lib.ptr(tri), tri.shape[0],
                  _lib.ptr(out), _lib.stream_ptr(rast.device))
    return out


def _mip_levels(H, W, max_mip_level=None):
    full = _lib.raw('mve_mip_levels')(int(H), int(W))
    return full if max_mip_level is None else min(full, int(max_mip_level))


def build_mips(tex, max_mip_level=None):
    """tex [Bt,H,W,C] fp32 -> (mips [Bt, mip_texels*C], levels): the box-filtered level stack dr.texture builds for the mip-mapped filters."""
    tex = tex.detach().float().contiguous()
    Bt, H, W, C = tex.shape
    lv = _mip_levels(H, W, max_mip_level)
    mips = torch.empty(Bt, _lib.raw('mve_mip_texels')(H, W, lv) * C, dtype=torch.float32, device=tex.device)
    with torch.cuda.device(tex.device):
        _lib.call('mve_mip_build', _lib.ptr(tex), Bt, H, W, C, lv, _lib.ptr(mips), _lib.stream_ptr(tex.device))
    return mips, lv


def _texture_mip_raw(tex, mips, lv, uv, uv_da, rast):
    n, h, w, _ = uv.shape
    Bt, H, W, C = tex.shape
    out = torch.empty(n, h, w, C, dtype=torch.float32, device=uv.device)
    with torch.cuda.device(uv.device):
        _lib.call('mve_texture_mip', _lib.ptr(tex), _lib.ptr(mips), Bt, H, W, C, lv, _lib.ptr(uv), _lib.ptr(uv_da),
                  _lib.ptr(rast) if rast is not None else None, n, h, w, _lib.ptr(out), _lib.stream_ptr(uv.device))
    return out


class _TextureMipFn(torch.autograd.Function):
    """dr.texture(..., filter_mode='linear-mipmap-linear') with its gradient w.r.t. the texture (through the level stack)."""

    @staticmethod
    def forward(ctx, tex, uv, uv_da, rast, max_mip_level):
        mips, lv = build_mips(tex, max_mip_level)
        ctx.save_for_backward(uv, uv_da, rast)
        ctx.tex_shape, ctx.lv = tuple(tex.shape), lv
        return _texture_mip_raw(tex, mips, lv, uv, uv_da, rast)

    @staticmethod
    def backward(ctx, g):
        uv, uv_da, rast = ctx.saved_tensors
        Bt, H, W, C = ctx.tex_shape
        n, h, w, _ = uv.shape
        g = g.float().contiguous()
        g_tex = torch.empty(Bt, H, W, C, dtype=torch.float32, device=uv.device)
        g_mips = torch.empty(Bt, max(1, _lib.raw('mve_mip_texels')(H, W, ctx.lv) * C), dtype=torch.float32, device=uv.device)
        with torch.cuda.device(uv.device):
            _lib.call('mve_texture_mip_backward', _lib.ptr(g), Bt, H, W, C, ctx.lv, _lib.ptr(uv), _lib.ptr(uv_da),
                      _lib.ptr(rast) if rast is not None else None, n, h, w, _lib.ptr(g_tex), _lib.ptr(g_mips), _lib.stream_ptr(uv.device))
        return g_tex, None, None, None, None


class _TextureFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, rast):
        ctx.save_for_backward(uv, rast)
        ctx.tex_shape = tuple(tex.shape)
        return _texture_raw(tex, uv, rast)

    @staticmethod
    def backward(ctx, g):
        uv, rast = ctx.saved_tensors
        Bt, th, tw, C = ctx.tex_shape
        n, h, w, _ = uv.shape
        g_tex = torch.zeros(Bt, th, tw, C, dtype=torch.float32, device=uv.device)
        g = g.float().contiguous()
        with torch.cuda.device(uv.device):
            _lib.call('mve_texture_bilinear_backward', _lib.ptr(g), Bt, th, tw, C, _lib.ptr(uv), _lib.ptr(rast), n, h, w, _lib.ptr(g_tex),
                      _lib.stream_ptr(uv.device))
        return g_tex, None, None


def texture(tex, uv, rast=None, uv_da=None, filter_mode='linear', max_mip_level=None):
    """dr.texture(tex [1|n,th,tw,C], uv [n,h,w,2], uv_da=..., filter_mode=...), wrap addressing; background -> 0 when rast is given.
    filter_mode 'linear' (bilinear) or 'linear-mipmap-linear' (needs uv_da [n,h,w,4], the attribute differentials of uv).
    Differentiable w.r.t. tex; uv / uv_da carry no gradient (refused loudly: the reference propagates it through nvdiffrast)."""
    if torch.is_grad_enabled() and (uv.requires_grad or (uv_da is not None and uv_da.requires_grad)):
        raise NotImplementedError('texture: no gradient w.r.t. uv / uv_da is built (geometry of a textured mesh is not optimised by the '
                                  'pipelines); detach them or call under torch.no_grad()')
    assert filter_mode in ('linear', 'linear-mipmap-linear'), filter_mode
    if filter_mode == 'linear-mipmap-linear':
        assert uv_da is not None, "filter_mode='linear-mipmap-linear' needs uv_da (interpolate_da of the texture coordinates)"
        uvc, dac = uv.detach().float().contiguous(), uv_da.detach().float().contiguous()
        rc = rast.detach().contiguous() if rast is not None else None
        if torch.is_grad_enabled() and tex.requires_grad:
            return _TextureMipFn.apply(tex.float().contiguous(), uvc, dac, rc, max_mip_level)
        t = tex.detach().float().contiguous()
        mips, lv = build_mips(t, max_mip_level)
        return _texture_mip_raw(t, mips, lv, uvc, dac, rc)
    if torch.is_grad_enabled() and tex.requires_grad:
        return _TextureFn.apply(tex.float().contiguous(), uv.detach().float().contiguous(), rast.contiguous() if rast is not None else None)
    return _texture_raw(tex, uv, rast)


def _texture_raw(tex, uv, rast=None):
    tex, uv = tex.detach().float().contiguous(), uv.detach().float().contiguous()
    n, h, w, _ = uv.shape
    out = torch.empty(n, h, w, tex.shape[-1], dtype=torch.float32, device=uv.device)
    with torch.cuda.device(uv.device):
        _lib.call('mve_texture_bilinear', _lib.ptr(tex), tex.shape[0], tex.shape[1], tex.shape[2], tex.shape[3], _lib.ptr(uv),
                  _lib.ptr(rast.contiguous()) if rast is not None else None, n, h, w, _lib.ptr(out), _lib.stream_ptr(uv.device))
    return out


def box_downsample(x, factor):
    """interpolate_hwc(x, 1/factor) (mode='area') for x [..., H, W, C]."""
    _inference_only('box_downsample', x)
    lead = x.shape[:-3]
    H, W, C = x.shape[-3:]
    x = x.float().contiguous()
    y = torch.empty(*lead, H // factor, W // factor, C, dtype=torch.float32, device=x.device)
    B = 1
    for d in lead:
        B *= d
    with torch.cuda.device(x.device):
        _lib.call('mve_box_downsample', _lib.ptr(x), B, H, W, C, int(factor), _lib.ptr(y), _lib.stream_ptr(x.device))
    return y


class MeshRenderer:
    """The reference's MeshRenderer for one mesh (num_scenes = 1, as in every MVEdit pipeline):
    forward (base_mesh_renderer.py:207-395), get_cam_weights_uv (:425-505) and bake_multiview (:507-603), with the reference's
    `texture_filter` ('linear-mipmap-linear' by default, :196: screen-space UV differentials out of rasterize / interpolate, a box-filtered
    level stack and a trilinear fetch; 'linear' = plain bilinear).  forward is differentiable w.r.t. the texture / vertex colours and,
    through rasterize / interpolate / antialias, w.r.t. the geometry."""

    def __init__(self, near=0.1, far=10, ssaa=1, texture_filter='linear-mipmap-linear', allow_detached_uv=False, min_render_bs=32):
        assert texture_filter in ('linear', 'linear-mipmap-linear')
        # `render_bs` of get_cam_weights_uv / bake_multiview is the reference's memory knob for a 24 GB card; views are independent and the atlas
        # accumulates them in view order whatever the chunking (bitwise the same result), so by default chunks of at least 32 views are walked
        # (up to 4x the workspace the caller's render_bs asks for: min_render_bs=1 honours the caller's value exactly)
        self.min_render_bs = int(min_render_bs)
        # a textured mesh with trainable vertices: False (default) = fail loudly (d albedo / d uv is not built), True = treat the fetch position
        # as a constant and warn once per renderer
        self.allow_detached_uv = bool(allow_detached_uv)
        self._warned_uv = False
        self.near, self.far, self.ssaa, self.texture_filter = near, far, ssaa, texture_filter

    def project(self, v, poses, intrinsics, h, w):
        """v [V,3], poses [b,3,4] c2w (OpenCV), intrinsics [b,4] -> (v_cam [b,V,3], v_clip [b,V,4]); :222-237."""
        r_c2w = torch.cat([poses[:, :3, :1], -poses[:, :3, 1:3]], dim=-1)         # opencv -> opengl
        proj = poses.new_zeros(poses.shape[0], 4, 4)
        proj[:, 0, 0] = 2 * intrinsics[:, 0] / w
        proj[:, 0, 2] = -2 * intrinsics[:, 2] / w + 1
        proj[:, 1, 1] = -2 * intrinsics[:, 1] / h
        proj[:, 1, 2] = -2 * intrinsics[:, 3] / h + 1
        proj[:, 2, 2] = -(self.far + self.near) / (self.far - self.near)
        proj[:, 2, 3] = -(2 * self.far * self.near) / (self.far - self.near)
        proj[:, 3, 2] = -1
        v_cam = (v[None] - poses[:, None, :3, 3]) @ r_c2w
        v_clip = torch.nn.functional.pad(v_cam, (0, 1), value=1.0) @ proj.transpose(-1, -2)
        return v_cam, v_clip, r_c2w

    def render_geometry(self, v, f, vn, fn, poses, intrinsics, h, w, normal_bg=(0.5, 0.5, 1.0)):
        """rasterise + depth + camera-space normals + alpha (the texture-free part of forward)."""
        if self.ssaa > 1:
            h, w, intrinsics = h * self.ssaa, w * self.ssaa, intrinsics * self.ssaa
        v_cam, v_clip, r_c2w = self.project(v.float(), poses.float(), intrinsics.float(), h, w)
        rast = rasterize(v_clip, f, (h, w))
        fg = rast[..., 3] > 0
        depth = 1 / interpolate(-v_cam[..., 2:3].contiguous(), rast, f)[..., 0]
        depth = depth.masked_fill(~fg, 0)
        normal = torch.nn.functional.normalize(interpolate(vn[None], rast, fn), dim=-1)
        rot_normal = (normal @ r_c2w[:, None]) / 2 + 0.5
        rot_normal[~fg] = rot_normal.new_tensor(normal_bg)
        return dict(rast=rast, alpha=fg.float()[..., None], depth=depth, normal=rot_normal, v_clip=v_clip, world_normal=normal)

    def forward(self, meshes, poses, intrinsics, h, w, shading_fun=None, dilate_edges=0, normal_bg=(0.5, 0.5, 1.0), aa=True,
                render_vc=False):
        """Same signature / result as the reference: poses [1,b,3|4,4], intrinsics [1,b,4] ->
        dict(rgba [1,b,h,w,4], depth [1,b,h,w], normal [1,b,h,w,3])."""
        assert len(meshes) == 1 and poses.shape[0] == 1, 'one mesh per call (num_scenes = 1)'
        mesh = meshes[0]
        ssaa = self.ssaa
        self.ssaa = 1                                               # render_geometry would scale again
        try:
            hh, ww, intr = h * ssaa, w * ssaa, intrinsics[0].float() * ssaa
            f = mesh.f.to(torch.int32).contiguous()
            fn = mesh.fn if getattr(mesh, 'fn', None) is not None else f
            g = self.render_geometry(mesh.v.float(), f, mesh.vn.float(), fn, poses[0].float(), intr, hh, ww, normal_bg)
        finally:
            self.ssaa = ssaa
        rast, alpha, depth, rot_normal = g['rast'], g['alpha'], g['depth'], g['normal']
        fg = rast[..., 3] > 0
        if getattr(mesh, 'vt', None) is not None and getattr(mesh, 'albedo', None) is not None:
            texc = interpolate(mesh.vt[None], rast, mesh.ft)
            texc_da = None
            if self.texture_filter == 'linear-mipmap-linear':                         # :241, :260-261
                texc_da = interpolate_da(mesh.vt[None], rast, rasterize_db(g['v_clip'], f, rast), mesh.ft)
            if torch.is_grad_enabled() and (texc.requires_grad or (texc_da is not None and texc_da.requires_grad)):
                # trainable vertices under a textured mesh: the reference propagates d albedo / d uv through dr.texture; no such kernel is
                # built here (the shipped pipelines render in_mesh.detach()).  The fetch position is treated as a constant -- the gradient
                # still reaches the vertices through rasterize / interpolate / antialias and the texture through the fetch -- and says so
                # once per renderer; the public texture() op keeps refusing.  OPT-IN (allow_detached_uv=True): by default the call fails loudly, as
                # an incomplete gradient must not pass for the reference's.
                if not self.allow_detached_uv:
                    raise NotImplementedError(
                        'MeshRenderer.forward: a textured mesh with trainable vertices needs d albedo / d uv through the texture fetch, which is not '
                        'built; render mesh.detach() (as the shipped pipelines do) or construct MeshRenderer(allow_detached_uv=True) to treat the '
                        'fetch position as a constant (vertex gradients through the albedo term are then missing)')
                if not self._warned_uv:
                    import warnings
                    warnings.warn('MeshRenderer.forward: texture coordinates are detached (no gradient w.r.t. uv through the texture fetch); '
                                  'vertex gradients through the albedo term are missing', stacklevel=2)
                    self._warned_uv = True
                texc = texc.detach()
                texc_da = texc_da.detach() if texc_da is not None else None
            albedo = texture(mesh.albedo[None, ..., :3], texc, rast, uv_da=texc_da, filter_mode=self.texture_filter)   # background 0 (:264)
        elif getattr(mesh, 'vc', None) is not None:
            rgba = interpolate(mesh.vc.float()[None] if mesh.vc.dim() == 2 else mesh.vc.float(), rast, f)
            alpha = alpha * rgba[..., 3:4]
            albedo = rgba[..., :3] * alpha
        else:
            albedo = torch.zeros_like(rot_normal)
        if shading_fun is not None:                                                   # :271-281
            xyz = interpolate(mesh.v.float()[None], rast, f)
            rgb = shading_fun(world_pos=xyz[fg], albedo=albedo[fg], world_normal=g['world_normal'][fg], fg_mask=fg[None])
            albedo = torch.zeros_like(albedo)
            albedo[fg] = rgb.float()
        rgba = torch.cat([albedo, alpha], dim=-1)
        if dilate_edges > 0:
            x = rgba.permute(0, 3, 1, 2)
            rgba = edge_dilation(x, x[:, 3:], dilate_edges).permute(0, 2, 3, 1)
        if aa:
            packed = antialias(torch.cat([rgba, depth[..., None], rot_normal], dim=-1), rast, g['v_clip'], f)
            rgba, depth, rot_normal = packed[..., :4], packed[..., 4], packed[..., 5:8]
        if ssaa > 1:
            rgba = box_downsample(rgba, ssaa)
            depth = box_downsample(depth[..., None], ssaa)[..., 0]
            rot_normal = box_downsample(rot_normal, ssaa)
        return dict(rgba=rgba[None], depth=depth[None], normal=rot_normal[None])

    __call__ = forward

    # ---------------------------------------------------------------------------------------------------------------
    def bake_xyz_shading_fun(self, meshes, shading_fun, map_size=1024, force_auto_uv=False, dilation_iters=7):
        """base_mesh_renderer.py:397-423: evaluate `shading_fun(world_pos=...)` at the surface point behind every texel of the
        UV atlas and store the result as the albedo map.  (mesh.auto_uv -- xatlas -- is mesh I/O and out of scope: the mesh must
        carry vt / ft.)"""
        assert len(meshes) == 1, 'only support one mesh'
        mesh = meshes[0]
        assert mesh.vt is not None and not force_auto_uv, 'UV unwrapping (mesh.auto_uv) is not part of this engine'
        assert len(mesh.ft) == len(mesh.f)
        vt = mesh.vt.float().contiguous()
        vt_clip = torch.cat([vt * 2 - 1, vt.new_tensor([[0., 1.]]).expand(vt.size(0), -1)], dim=-1)
        rast = rasterize(vt_clip[None], mesh.ft, (map_size, map_size))
        valid = rast[0, ..., 3] > 0
        xyz = interpolate(mesh.v.detach().float()[None], rast, mesh.f)[0]
        rgb = shading_fun(world_pos=xyz[valid])
        albedo = xyz.new_zeros((map_size, map_size, 3))
        albedo[valid] = rgb.float()
        albedo = edge_dilation(albedo.permute(2, 0, 1)[None], valid[None, None].float(), iters=dilation_iters)[0].permute(1, 2, 0)
        mesh.albedo = torch.cat([albedo.clamp(min=0, max=1), torch.ones_like(albedo[..., :1])], dim=-1)
        mesh.textureless = False
        return [mesh]

    def _view_batch(self, v, f, vt, ft, poses, intrinsics, alphas, h, w, map_size, cos_weight_pow):
        """Per-view geometry shared by bake_multiview and get_cam_weights_uv (base_mesh_renderer.py:441-481 == :527-566):
        -> (visibility fp32 [bs,map,map] = `visibility_grad`, eroded cos^pow * alpha weight image [bs,h,w], v_img [bs,V,2], and for the
        bilinear filter the visibility in its 2^-32 fixed-point form, which mve_bake_accumulate consumes)."""
        dev = v.device
        bs = poses.shape[0]
        sp = _lib.stream_ptr(dev)
        mip = self.texture_filter == 'linear-mipmap-linear'
        v_cam, v_clip, _ = self.project(v, poses, intrinsics, h, w)
        rast = rasterize(v_clip, f, (h, w))
        texc = interpolate(vt[None], rast, ft)
        depth = 1 / interpolate(-v_cam[..., 2:3].contiguous(), rast, f)[..., 0]
        depth = depth.masked_fill(~(rast[..., 3] > 0), 0).contiguous()
        v_img = (v_clip[..., :2] / v_clip[..., 3:] * 0.5 + 0.5).contiguous()
        tmp = torch.empty(bs, h, w, dtype=torch.float32, device=dev)
        w_img = torch.empty_like(tmp)
        alpha_b, intr_b = alphas.float().contiguous(), intrinsics.float().contiguous()
        with torch.cuda.device(dev):
            if mip:                 # gradient of the trilinear fetch w.r.t. a texture of ones, through the level stack (:466-475)
                texc_da = interpolate_da(vt[None], rast, rasterize_db(v_clip, f, rast), ft)
                lv = _mip_levels(map_size, map_size)
                nbytes = _lib.raw('mve_visibility_mip_workspace_bytes')(bs, map_size, lv)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                vis, vis64 = torch.empty(bs, map_size, map_size, dtype=torch.float32, device=dev), None
                _lib.call('mve_visibility_mip', _lib.ptr(texc), _lib.ptr(texc_da), _lib.ptr(rast), bs, h, w, map_size, lv, _lib.ptr(ws), nbytes,
                          _lib.ptr(vis), sp)
            else:
                vis64 = torch.empty(bs, map_size, map_size, dtype=torch.int64, device=dev)
                _lib.call('mve_splat_visibility', _lib.ptr(texc), _lib.ptr(rast), bs, h, w, map_size, _lib.ptr(vis64), sp)
                vis = (vis64.double() / 4294967296.0).float()
            _lib.call('mve_view_weight', _lib.ptr(depth), _lib.ptr(alpha_b), _lib.ptr(intr_b), bs, h, w, float(cos_weight_pow),
                      _lib.ptr(tmp), _lib.ptr(w_img), sp)
        return vis, w_img, v_img, vis64

    def get_cam_weights_uv(self, meshes, poses, intrinsics, alphas=None, render_size=512, map_size=1024, render_bs=8, cos_weight_pow=1.0):
        """base_mesh_renderer.py:425-505: per-view, per-texel blending weights (cos^pow of the viewing angle, eroded, fetched at the
        texel's projection, times the texel's visibility footprint) -> (weights [1,n,map,map,1], valid [1,map,map])."""
        assert len(meshes) == 1, 'only support one mesh'
        mesh = meshes[0]
        n = max(poses.size(-3), intrinsics.size(-2))
        poses = poses[0].expand(n, -1, -1).float()
        intrinsics = intrinsics[0].expand(n, -1).float().contiguous()
        if alphas is not None:
            _, h, w, _ = alphas.size()
            assert render_size == h == w
        else:
            h = w = render_size
            alphas = torch.ones((n, h, w, 1), device=poses.device, dtype=torch.float32)
        v, f = mesh.v.detach().float(), mesh.f.to(torch.int32).contiguous()
        vt, ft = mesh.vt.float().contiguous(), mesh.ft.to(torch.int32).contiguous()
        vt_clip = torch.cat([vt * 2 - 1, vt.new_tensor([[0., 1.]]).expand(vt.size(0), -1)], dim=-1)
        tex_rast = rasterize(vt_clip[None], ft, (map_size, map_size))
        valid = tex_rast[0, ..., 3] > 0
        mip = self.texture_filter == 'linear-mipmap-linear'
        tex_rast_db = rasterize_db(vt_clip[None], ft, tex_rast) if mip else None
        out = []
        render_bs = max(int(render_bs), int(self.min_render_bs))
        for i0 in range(0, n, render_bs):
            sl = slice(i0, min(i0 + render_bs, n))
            bs = sl.stop - sl.start
            vis, w_img, v_img, _ = self._view_batch(v, f, vt, ft, poses[sl], intrinsics[sl], alphas[sl], h, w, map_size, cos_weight_pow)
            tr = tex_rast.expand(bs, -1, -1, -1).contiguous()
            imgc = interpolate(v_img, tr, f)
            imgc_da = interpolate_da(v_img, tr, tex_rast_db.expand(bs, -1, -1, -1).contiguous(), f) if mip else None      # :496-497
            tex = texture(w_img[..., None], imgc, uv_da=imgc_da, filter_mode=self.texture_filter)
            out.append(tex * vis[..., None])
        return torch.cat(out, dim=0)[None], valid[None]

    def bake_multiview(self, meshes, images, alphas, poses, intrinsics, map_size=1024, cos_weight_pow=8.0, base_weight=0.0,
                       render_bs=8, return_debug=False):
        """Texture back-projection with the reference's signature (base_mesh_renderer.py:507-603).

        meshes: a list with ONE object exposing v [V,3], f [F,3], vt [Vt,2], ft [F,3] (and optionally albedo [h,w,3|4]);
        images [1,n,h,w,3], alphas [1,n,h,w,1], poses [1,n,3|4,4], intrinsics [1,n,4].  Sets mesh.albedo [map,map,4] and
        mesh.textureless = False, returns [mesh].  Filtering follows self.texture_filter (csrc/texture_mip.hip / csrc/raster.hip)."""
        assert len(meshes) == 1, 'only support one mesh'
        mesh = meshes[0]
        images, alphas = images[0].float().contiguous(), alphas[0].float().contiguous()
        n, h, w, _ = images.shape
        dev = images.device
        poses = poses[0].expand(n, -1, -1).float()
        intrinsics = intrinsics[0].expand(n, -1).float().contiguous()
        v, f = mesh.v.detach().float(), mesh.f.to(torch.int32).contiguous()
        vt, ft = mesh.vt.float().contiguous(), mesh.ft.to(torch.int32).contiguous()
        sp = _lib.stream_ptr(dev)

        vt_clip = torch.cat([vt * 2 - 1, vt.new_tensor([[0., 1.]]).expand(vt.size(0), -1)], dim=-1)
        tex_rast = rasterize(vt_clip[None], ft, (map_size, map_size))[0].contiguous()
        valid = tex_rast[..., 3] > 0
        mip = self.texture_filter == 'linear-mipmap-linear'
        tex_rast_db = rasterize_db(vt_clip[None], ft, tex_rast[None])[0].contiguous() if mip else None          # :521
        accum = torch.zeros(map_size, map_size, 4, dtype=torch.float32, device=dev)
        debug = dict(vis=[], wimg=[], tex_rast=tex_rast)

        render_bs = max(int(render_bs), int(self.min_render_bs))
        for i0 in range(0, n, render_bs):
            sl = slice(i0, min(i0 + render_bs, n))
            bs = sl.stop - sl.start
            vis, w_img, v_img, vis64 = self._view_batch(v, f, vt, ft, poses[sl], intrinsics[sl], alphas[sl], h, w, map_size, cos_weight_pow)
            img_b = images[sl].contiguous()
            with torch.cuda.device(dev):
                if mip:             # dr.texture(cat([images, img_space_weight]), imgc, uv_da=imgc_db) per texel and view (:573-577)
                    img4 = torch.cat([img_b, w_img[..., None]], dim=-1).contiguous()
                    mips, lv = build_mips(img4)
                    _lib.call('mve_bake_accumulate_mip', _lib.ptr(tex_rast), _lib.ptr(tex_rast_db), _lib.ptr(f), f.shape[0], _lib.ptr(v_img),
                              v_img.shape[1], _lib.ptr(img4), _lib.ptr(mips), h, w, lv, _lib.ptr(vis), bs, map_size, _lib.ptr(accum), sp)
                else:
                    _lib.call('mve_bake_accumulate', _lib.ptr(tex_rast), _lib.ptr(f), f.shape[0], _lib.ptr(v_img), v_img.shape[1],
                              _lib.ptr(img_b), _lib.ptr(w_img), _lib.ptr(vis64), bs, h, w, map_size, _lib.ptr(accum), sp)
            if return_debug:
                debug['vis'].append(vis)
                debug['wimg'].append(w_img)

        if base_weight > 0 and getattr(mesh, 'albedo', None) is not None:                     # :584-591
            tex = mesh.albedo.float()
            if tex.shape[0] != map_size or tex.shape[1] != map_size:
                tex = torch.nn.functional.interpolate(tex.permute(2, 0, 1)[None], size=map_size, mode='bilinear')[0].permute(1, 2, 0)
            wgt = (tex[..., 3:4] * valid[..., None] if tex.size(-1) == 4 else valid[..., None].float()) * (base_weight ** cos_weight_pow)
            accum[..., :3] += tex[..., :3] * wgt
            accum[..., 3:] += wgt

        albedo = torch.empty(1, 3, map_size, map_size, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_bake_finalize', _lib.ptr(accum), map_size, _lib.ptr(albedo), sp)
        albedo = edge_dilation(albedo, valid[None, None].float())[0].permute(1, 2, 0)
        mesh.albedo = torch.cat([albedo.clamp(min=0, max=1), torch.ones_like(albedo[..., :1])], dim=-1)
        mesh.textureless = False
        if return_debug:
            debug.update(accum=accum, valid=valid)
            return [mesh], debug
        return [mesh]


class _AutoNormalFn(torch.autograd.Function):
    """(vn, face_normals) = Mesh.auto_normal(v, f) with the native backward w.r.t. v."""

    @staticmethod
    def forward(ctx, verts, faces):
        assert verts.is_cuda and faces.is_cuda and verts.dim() == 2 and verts.shape[1] == 3 and faces.dim() == 2 and faces.shape[1] == 3
        v = verts.detach().to(torch.float32).contiguous()
        f = faces.to(torch.int32).contiguous()
        V, F = v.shape[0], f.shape[0]
        fn = torch.empty(F, 3, dtype=torch.float32, device=v.device)
        vs, vn = torch.empty_like(v), torch.empty_like(v)
        with torch.cuda.device(v.device):
            _lib.call('mve_mesh_normals_forward', _lib.ptr(v), V, _lib.ptr(f), F, _lib.ptr(fn), _lib.ptr(vs), _lib.ptr(vn), _lib.stream_ptr(v.device))
        ctx.save_for_backward(v, f, vs)
        ctx.dtype = verts.dtype
        return vn.to(verts.dtype), fn.to(verts.dtype)

    @staticmethod
    def backward(ctx, g_vn, g_fn):
        v, f, vs = ctx.saved_tensors
        g_vn = None if g_vn is None else g_vn.detach().to(torch.float32).contiguous()
        g_fn = None if g_fn is None else g_fn.detach().to(torch.float32).contiguous()
        scratch, g_v = torch.empty_like(v), torch.empty_like(v)
        with torch.cuda.device(v.device):
            _lib.call('mve_mesh_normals_backward', _lib.ptr(v), v.shape[0], _lib.ptr(f), f.shape[0], _lib.ptr(vs), _lib.ptr(g_vn), _lib.ptr(g_fn),
                      _lib.ptr(scratch), _lib.ptr(g_v), _lib.stream_ptr(v.device))
        return g_v.to(ctx.dtype), None


class Mesh:
    """Stand-in for lib.models.decoders.mesh_renderer.mesh_utils.Mesh: the attributes the render / bake / optimisation paths touch, and
    `auto_normal` (mesh_utils.py:359-382) on native kernels."""

    def __init__(self, v, f, vt=None, ft=None, vn=None, fn=None, albedo=None, vc=None, device=None):
        self.v, self.f, self.vt, self.ft, self.vn, self.fn, self.albedo, self.vc = v, f, vt, ft, vn, fn, albedo, vc
        self.face_normals = None
        self.textureless = albedo is None

    def auto_normal(self, seamless=False):
        """vn, fn, face_normals as the reference sets them; differentiable w.r.t. v.  seamless=True (vertex welding through
        torch.unique) is not on the optimisation path and not built."""
        if seamless:
            raise NotImplementedError('Mesh.auto_normal(seamless=True) is outside the native path')
        self.vn, self.face_normals = _AutoNormalFn.apply(self.v, self.f)
        self.fn = self.f.to(torch.int32)


class _DMTetFn(torch.autograd.Function):
    """verts = DMTet(pos, sdf) with the native backward (d verts -> d pos, d sdf); faces carry no gradient."""

    @staticmethod
    def forward(ctx, dm, pos, sdf, tets32):
        verts, faces, edges = dm._extract(pos, sdf, tets32, want_edges=True)
        ctx.save_for_backward(pos, sdf, edges)
        ctx.mark_non_differentiable(faces)
        return verts, faces

    @staticmethod
    def backward(ctx, g_verts, _g_faces):
        pos, sdf, edges = ctx.saved_tensors
        g_pos, g_sdf = torch.zeros_like(pos), torch.zeros_like(sdf)
        g = g_verts.float().contiguous()
        with torch.cuda.device(pos.device):
            _lib.call('mve_dmtet_backward', _lib.ptr(pos), _lib.ptr(sdf), _lib.ptr(edges), edges.shape[0], _lib.ptr(g), _lib.ptr(g_pos),
                      _lib.ptr(g_sdf), _lib.stream_ptr(pos.device))
        return None, g_pos, g_sdf, None


class DMTet:
    """Mirror of the reference's DMTet (base_mesh_renderer.py:104-188): `DMTet(device)(pos_nx3, sdf_n, tet_fx4) -> verts, faces`.
    When pos or sdf require grad the vertices carry autograd history (native backward), as in the reference's mesh optimisation."""

    def __init__(self, device='cuda'):
        self.device = torch.device(device)
        self._tets_key, self._tets32 = None, None

    def __call__(self, pos_nx3, sdf_n, tet_fx4):
        key = (tet_fx4.data_ptr(), tuple(tet_fx4.shape), tet_fx4.dtype)
        if key != self._tets_key:                                   # the tet grid is constant across iterations: convert once
            self._tets32 = tet_fx4.to(self.device, torch.int32).contiguous()
            self._tets_key = key
        if torch.is_grad_enabled() and (pos_nx3.requires_grad or sdf_n.requires_grad):
            pos = pos_nx3.to(self.device, torch.float32).contiguous()
            sdf = sdf_n.to(self.device, torch.float32).contiguous()
            verts, faces = _DMTetFn.apply(self, pos, sdf, self._tets32)
            return verts, faces.long()
        verts, faces, _ = self._extract(pos_nx3.detach().to(self.device, torch.float32).contiguous(),
                                        sdf_n.detach().to(self.device, torch.float32).contiguous(), self._tets32)
        return verts, faces.long()

    def _extract(self, pos, sdf, tets, want_edges=False):
        pos, sdf = pos.detach(), sdf.detach()
        nv, nt = pos.shape[0], tets.shape[0]
        nbytes = _lib.raw('mve_dmtet_workspace_bytes')(nv, nt)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        counts = torch.zeros(2, dtype=torch.int32, device=self.device)
        sp = _lib.stream_ptr(self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_dmtet_count', _lib.ptr(sdf), _lib.ptr(tets), nv, nt, _lib.ptr(counts), _lib.ptr(ws), nbytes, sp)
            n_verts, n_faces = (int(x) for x in counts.tolist())     # one host read, as march_rays_train
            verts = torch.empty(n_verts, 3, dtype=torch.float32, device=self.device)
            faces = torch.empty(n_faces, 3, dtype=torch.int32, device=self.device)
            edges = torch.empty(n_verts, 2, dtype=torch.int32, device=self.device) if want_edges else None
            _lib.call('mve_dmtet_write', _lib.ptr(pos), _lib.ptr(sdf), _lib.ptr(tets), nv, nt, _lib.ptr(verts), _lib.ptr(faces),
                      _lib.ptr(edges), _lib.ptr(ws), nbytes, sp)
        return verts, faces, edges


class _MeshRegFn(torch.autograd.Function):
    """(laplacian_smooth_loss, normal_consistency) in one pass over the mesh: mve_mesh_reg_forward / _backward."""

    @staticmethod
    def forward(ctx, verts, face_normals, faces):
        assert verts.is_cuda and faces.is_cuda and verts.dim() == 2 and verts.shape[1] == 3 and faces.dim() == 2 and faces.shape[1] == 3
        dev = verts.device
        v = verts.detach().to(torch.float32).contiguous()
        fn = face_normals.detach().to(device=dev, dtype=torch.float32).contiguous()
        f = faces.to(torch.int32).contiguous()
        V, F = v.shape[0], f.shape[0]
        assert fn.shape == (F, 3)
        ws = torch.empty(_lib.raw('mve_mesh_reg_workspace_bytes')(V, F), dtype=torch.uint8, device=dev)     # carried to backward
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_mesh_reg_forward', _lib.ptr(v), V, _lib.ptr(f), F, _lib.ptr(fn), _lib.ptr(ws), ws.numel(), _lib.ptr(losses),
                      _lib.stream_ptr(dev))
        ctx.save_for_backward(v, f, fn, ws)
        ctx.dtypes = (verts.dtype, face_normals.dtype)
        return losses[0].clone(), losses[1].clone()

    @staticmethod
    def backward(ctx, g_lap, g_nc):
        v, f, fn, ws = ctx.saved_tensors
        dev = v.device
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        gl = torch.stack([zero if g_lap is None else g_lap.detach().float().reshape(()), zero if g_nc is None else g_nc.detach().float().reshape(())])
        g_v, g_fn = torch.empty_like(v), torch.empty_like(fn)
        with torch.cuda.device(dev):
            _lib.call('mve_mesh_reg_backward', _lib.ptr(v), v.shape[0], _lib.ptr(f), f.shape[0], _lib.ptr(fn), _lib.ptr(ws), ws.numel(), _lib.ptr(gl),
                      _lib.ptr(g_v), _lib.ptr(g_fn), _lib.stream_ptr(dev))
        return g_v.to(ctx.dtypes[0]), g_fn.to(ctx.dtypes[1]), None


def mesh_regularizers(verts, faces, face_normals):
    """-> (laplacian_smooth_loss(verts, faces), normal_consistency(face_normals, faces)) of
    lib/models/decoders/mesh_renderer/base_mesh_renderer.py:55-101 in one native pass (the pair the mesh-optimisation loop adds at
    mvedit_3d_pipeline.py:775-776), differentiable w.r.t. verts and face_normals."""
    return _MeshRegFn.apply(verts, face_normals, faces)


def laplacian_smooth_loss(verts, faces):
    """base_mesh_renderer.py:94-101"""
    return _MeshRegFn.apply(verts, torch.zeros(faces.shape[0], 3, dtype=torch.float32, device=verts.device), faces)[0]


def normal_consistency(face_normals, t_pos_idx, num_verts=None):
    """base_mesh_renderer.py:55-68; num_verts saves the device read of t_pos_idx.max() when the caller knows it"""
    V = int(t_pos_idx.max()) + 1 if num_verts is None else int(num_verts)
    return _MeshRegFn.apply(torch.zeros(V, 3, dtype=torch.float32, device=face_normals.device), face_normals, t_pos_idx)[1]
