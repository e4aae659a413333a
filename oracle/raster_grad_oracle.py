"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Geometry gradients of the mesh rasteriser ("raster / interpolate backward"): what `dr.rasterize` /
`dr.interpolate` of nvdiffrast@c5caf7b (requirements.txt:3, absent) give the reference when it optimises DMTet vertices through
`MeshRenderer.forward` (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:240-252).  With the triangle of every pixel held
fixed, rast = (u, v, z/w) is a smooth function of the clip-space vertices:

    s_k = ((x_k / w_k) / 2 + 1/2) * (W, H)                           screen position of vertex k
    E_k = edge function of the two other vertices at the pixel centre; b_k = E_k / sum_j E_j
    q_k = b_k / w_k;  u = q_0 / sum q;  v = q_1 / sum q;  z = sum_k b_k z_k / w_k

`rast_continuous` states that with torch ops (autograd differentiates it); `rasterize_backward` is the hand-derived chain rule the HIP
kernel implements, kept here so that the derivation is checked against autograd on the CPU.

PARITY UNPINNED (nvdiffrast absent); the forward it differentiates is the repo's own specification, oracle/raster_oracle.c."""
import torch


def rast_continuous(pos, tri, b_idx, ids, px, py, H, W):
    """pos [B,V,4]; for N pixels (view b_idx[n], triangle ids[n], pixel (px[n], py[n])) -> u, v, z [N] (unsnapped)."""
    P = pos[b_idx[:, None], tri[ids].long()]                     # [N,3,4]
    w = P[..., 3]
    sx = (P[..., 0] / w * 0.5 + 0.5) * W
    sy = (P[..., 1] / w * 0.5 + 0.5) * H
    cx, cy = px.to(pos.dtype) + 0.5, py.to(pos.dtype) + 0.5
    E = []
    for k in range(3):
        a, c = (k + 1) % 3, (k + 2) % 3
        E.append((sx[:, c] - sx[:, a]) * (cy - sy[:, a]) - (sy[:, c] - sy[:, a]) * (cx - sx[:, a]))
    E = torch.stack(E, dim=1)
    b = E / E.sum(dim=1, keepdim=True)
    q = b / w
    S = q.sum(dim=1)
    return q[:, 0] / S, q[:, 1] / S, (b * P[..., 2] / w).sum(dim=1)


def rasterize_backward(pos, tri, rast, grad_rast):
    """Closed form, operation for operation what k_rasterize_bwd does.  rast / grad_rast [B,H,W,4] (grad of u, v, z/w; the id channel
    carries none) -> grad_pos [B,V,4]."""
    B, H, W, _ = rast.shape
    g_pos = torch.zeros_like(pos)
    ids = rast[..., 3].long() - 1
    b_idx, yy, xx = torch.nonzero(ids >= 0, as_tuple=True)
    f = ids[b_idx, yy, xx]
    vid = tri[f].long()                                          # [N,3]
    P = pos[b_idx[:, None], vid]
    x, y, z, w = P.unbind(-1)
    sx, sy = (x / w * 0.5 + 0.5) * W, (y / w * 0.5 + 0.5) * H
    cx, cy = xx.to(pos.dtype) + 0.5, yy.to(pos.dtype) + 0.5
    gu, gv, gz = grad_rast[b_idx, yy, xx, 0], grad_rast[b_idx, yy, xx, 1], grad_rast[b_idx, yy, xx, 2]
    E = torch.stack([(sx[:, (k + 2) % 3] - sx[:, (k + 1) % 3]) * (cy - sy[:, (k + 1) % 3])
                     - (sy[:, (k + 2) % 3] - sy[:, (k + 1) % 3]) * (cx - sx[:, (k + 1) % 3]) for k in range(3)], dim=1)
    tot = E.sum(1, keepdim=True)
    b = E / tot
    iw, zw = 1 / w, z / w
    q = b * iw
    S = q.sum(1)
    u, v = q[:, 0] / S, q[:, 1] / S
    common = gu * u + gv * v
    gq = torch.stack([(gu - common) / S, (gv - common) / S, -common / S], dim=1)
    gb = gq * iw + gz[:, None] * zw
    g_iw, g_zw = gq * b, gz[:, None] * b
    gE = (gb - (gb * b).sum(1, keepdim=True)) / tot
    gsx, gsy = torch.zeros_like(sx), torch.zeros_like(sy)
    for k in range(3):
        a, c = (k + 1) % 3, (k + 2) % 3
        gsx[:, a] += gE[:, k] * (sy[:, c] - cy)
        gsy[:, a] += gE[:, k] * (cx - sx[:, c])
        gsx[:, c] += gE[:, k] * (cy - sy[:, a])
        gsy[:, c] += -gE[:, k] * (cx - sx[:, a])
    gx = gsx * (0.5 * W) * iw
    gy = gsy * (0.5 * H) * iw
    gw = -(gsx * (0.5 * W) * x + gsy * (0.5 * H) * y + g_zw * z + g_iw) * iw * iw
    gzc = g_zw * iw
    g = torch.stack([gx, gy, gzc, gw], dim=-1)                   # [N,3,4]
    g_pos.index_put_((b_idx[:, None].expand(-1, 3), vid), g, accumulate=True)
    return g_pos


def interpolate_backward_rast(attr, tri, rast, grad_out):
    """d out / d (u, v) of dr.interpolate: out = u a0 + v a1 + (1 - u - v) a2  ->  grad_rast [B,H,W,4] (channels 2, 3 zero)."""
    B, H, W, _ = rast.shape
    ids = rast[..., 3].long() - 1
    fg = ids >= 0
    vid = tri[ids.clamp(min=0)].long()                            # [B,H,W,3]
    at = attr if attr.shape[0] == B else attr.expand(B, -1, -1)
    a = torch.gather(at[:, None, None].expand(-1, H, W, -1, -1), 3, vid[..., None].expand(-1, -1, -1, -1, attr.shape[-1]))   # [B,H,W,3,A]
    g = torch.zeros_like(rast)
    g[..., 0] = (grad_out * (a[..., 0, :] - a[..., 2, :])).sum(-1) * fg
    g[..., 1] = (grad_out * (a[..., 1, :] - a[..., 2, :])).sum(-1) * fg
    return g


# ---------------------------------------------------------------------------------------------------------------------------
# dr.antialias: gradient w.r.t. the clip-space vertices (silhouette term).  The forward (oracle/raster_oracle.c: aa_pair) blends a
# pixel pair across a silhouette edge with weight |tt - 1/2|, tt = where the edge crosses the segment between the two pixel
# centres; with the discrete choices (front triangle, edge, direction) held fixed, tt is a smooth function of the edge's two
# vertices.  `aa_pair` is a Python restatement of the C rule that also reports those choices; `antialias_backward_pos` is the
# chain rule the HIP kernel implements.  tests/test_mesh_grad.py checks it against finite differences of the C oracle's forward.
# ---------------------------------------------------------------------------------------------------------------------------
def aa_pair(rast, pos, tri, opp, H, W, px, py, qx, qy):
    """-> None or dict(dst_is_p, wgt, sign, tri, a, b (vertex ids of the crossed edge), horizontal, s, d)   [one view, numpy float32]."""
    import numpy as np
    f32 = np.float32
    rp, rq = rast[py, px], rast[qy, qx]
    ip, iq = int(rp[3]) - 1, int(rq[3]) - 1
    if ip == iq:
        return None
    if ip < 0:
        use_p = False
    elif iq < 0:
        use_p = True
    elif rp[2] != rq[2]:
        use_p = bool(rp[2] < rq[2])
    else:
        use_p = (py * W + px) < (qy * W + qx)
    t = ip if use_p else iq
    if t < 0 or t >= tri.shape[0]:
        return None
    ox, oy, nx, ny = (px, py, qx, qy) if use_p else (qx, qy, px, py)
    vi = tri[t]
    v = pos[vi]
    if (v[:, 3] <= 1e-6).any():
        return None
    sx = (v[:, 0] / v[:, 3] * f32(0.5) + f32(0.5)) * f32(W)
    sy = (v[:, 1] / v[:, 3] * f32(0.5) + f32(0.5)) * f32(H)
    cx, cy = f32(ox + 0.5), f32(oy + 0.5)
    dx, dy = f32(nx - ox), f32(ny - oy)
    best, info = f32(2.0), None
    for e in range(3):
        a, b, c = e, (e + 1) % 3, (e + 2) % 3
        ex, ey = sx[b] - sx[a], sy[b] - sy[a]
        o = int(opp[t, e])
        if o >= 0:
            if o >= pos.shape[0]:
                continue
            vo = pos[o]
            if vo[3] <= 1e-6:
                continue
            oxs = (vo[0] / vo[3] * f32(0.5) + f32(0.5)) * f32(W)
            oys = (vo[1] / vo[3] * f32(0.5) + f32(0.5)) * f32(H)
            sc = ex * (sy[c] - sy[a]) - ey * (sx[c] - sx[a])
            so = ex * (oys - sy[a]) - ey * (oxs - sx[a])
            if not (sc * so >= 0):
                continue
        if dy == 0:
            if ey == 0:
                continue
            s = (cy - sy[a]) / ey
            tt = ((sx[a] + s * ex) - cx) * dx
        else:
            if ex == 0:
                continue
            s = (cx - sx[a]) / ex
            tt = ((sy[a] + s * ey) - cy) * dy
        if 0 <= s <= 1 and 0 <= tt <= 1 and tt < best:
            best, info = tt, dict(a=int(vi[a]), b=int(vi[b]), horizontal=bool(dy == 0), s=float(s), d=float(dx if dy == 0 else dy),
                                  ex=float(ex), ey=float(ey))
    if best > 1 or best == f32(0.5):
        return None
    info.update(tri=t, dst_is_p=(not use_p) if best > 0.5 else use_p, wgt=float(abs(best - f32(0.5))), sign=1.0 if best > 0.5 else -1.0)
    return info


def antialias_forward(color, rast, pos, tri, opp):
    """The C oracle's orc_antialias through the Python pair rule (used only to confirm that the restated rule makes the same choices)."""
    import numpy as np
    B, H, W, C = color.shape
    out = color.copy()
    for b in range(B):
        for y in range(H):
            for x in range(W):
                for ddx, ddy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    qx, qy = x + ddx, y + ddy
                    if qx < 0 or qx >= W or qy < 0 or qy >= H:
                        continue
                    r = aa_pair(rast[b], pos[b], tri, opp, H, W, x, y, qx, qy)
                    if r is None or not r['dst_is_p']:
                        continue
                    out[b, y, x] = out[b, y, x] + np.float32(r['wgt']) * (color[b, qy, qx] - color[b, y, x])
    return out


def antialias_backward_pos(color, rast, pos, tri, opp, grad_out):
    """d sum(out * grad_out) / d pos [B,V,4] with the pair decisions held fixed (float64 accumulation)."""
    import numpy as np
    B, H, W, C = color.shape
    g_pos = np.zeros(pos.shape, np.float64)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                for ddx, ddy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    qx, qy = x + ddx, y + ddy
                    if qx < 0 or qx >= W or qy < 0 or qy >= H:
                        continue
                    r = aa_pair(rast[b], pos[b], tri, opp, H, W, x, y, qx, qy)
                    if r is None or not r['dst_is_p']:
                        continue
                    # out[p] = col[p] + |tt - 1/2| (col[q] - col[p])  ->  d / d tt = sign * <g[p], col[q] - col[p]>
                    g_tt = r['sign'] * float(np.dot(grad_out[b, y, x].astype(np.float64), (color[b, qy, qx] - color[b, y, x]).astype(np.float64)))
                    s, d, ex, ey = r['s'], r['d'], r['ex'], r['ey']
                    if r['horizontal']:
                        gsx = {'a': (1 - s) * d, 'b': s * d}
                        gsy = {'a': ex * (s - 1) / ey * d, 'b': -ex * s / ey * d}
                    else:
                        gsy = {'a': (1 - s) * d, 'b': s * d}
                        gsx = {'a': ey * (s - 1) / ex * d, 'b': -ey * s / ex * d}
                    for k in ('a', 'b'):
                        vi = r[k]
                        xk, yk, _, wk = (float(t) for t in pos[b, vi])
                        hx, hy = g_tt * gsx[k] * 0.5 * W, g_tt * gsy[k] * 0.5 * H
                        g_pos[b, vi, 0] += hx / wk
                        g_pos[b, vi, 1] += hy / wk
                        g_pos[b, vi, 3] += -(hx * xk + hy * yk) / (wk * wk)
    return g_pos
