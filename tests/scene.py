"""Deterministic synthetic NeRF scene shared by the ray-marching tests (numpy only).

Occupancy = solid sphere of radius `radius` inside the [-bound,bound]^3 cube, stored the way the
reference stores it: a Morton-ordered density grid [C, H^3] packed 8 cells per byte
(lib/models/autoencoders/base_nerf.py:208-216, base_volume_renderer.py:105-177).
Cameras: pinhole, looking at the origin from `dist` (lib/apis/adapter3d.py:991-996: distance 3.7,
fov 30 deg).
"""
import numpy as np


def morton_np(x, y, z):
    def spread(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    return spread(x) | (spread(y) << np.uint32(1)) | (spread(z) << np.uint32(2))


def sphere_density_grid(H=128, C=1, bound=1.0, radius=0.5, seed=0):
    """-> density grid f32 [C, H^3] in Morton order; value 1+noise inside the sphere, noise*0.005 outside."""
    rng = np.random.default_rng(seed)
    g = np.arange(H)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    idx = morton_np(X.ravel(), Y.ravel(), Z.ravel())
    grid = np.zeros((C, H ** 3), np.float32)
    for c in range(C):
        b = min(2.0 ** c, bound)
        px = ((X.ravel() + 0.5) / H * 2 - 1) * b
        py = ((Y.ravel() + 0.5) / H * 2 - 1) * b
        pz = ((Z.ravel() + 0.5) / H * 2 - 1) * b
        inside = (px * px + py * py + pz * pz) < radius * radius
        vals = np.where(inside, 1.0 + rng.random(H ** 3), 0.005 * rng.random(H ** 3)).astype(np.float32)
        grid[c, idx] = vals
    return grid


def camera_rays(n_views=2, S=32, dist=3.7, fov_deg=30.0, seed=0, jitter=True):
    """-> rays_o, rays_d  f32 [n_views*S*S, 3] (unit directions)."""
    rng = np.random.default_rng(seed)
    f = S / (2 * np.tan(np.deg2rad(fov_deg) / 2))
    os_, ds_ = [], []
    for v in range(n_views):
        az = 2 * np.pi * v / n_views + 0.1
        el = -0.3 + 0.9 * (v / max(n_views - 1, 1))
        eye = dist * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, np.array([0, 0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        j, i = np.meshgrid(np.arange(S), np.arange(S), indexing='ij')
        u = (i + 0.5 - S / 2) / f
        w = (j + 0.5 - S / 2) / f
        if jitter:
            u = u + rng.normal(0, 1e-3, u.shape)
        d = fwd[None, None] + u[..., None] * right[None, None] - w[..., None] * up[None, None]
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        os_.append(np.broadcast_to(eye, d.shape).reshape(-1, 3))
        ds_.append(d.reshape(-1, 3))
    return (np.ascontiguousarray(np.concatenate(os_), dtype=np.float32),
            np.ascontiguousarray(np.concatenate(ds_), dtype=np.float32))


def icosphere(subdiv=3, radius=0.6):
    """-> vertices [V,3] f32, faces [F,3] i32 (CCW seen from outside)."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, np.int32)


def face_atlas(f, pad=0.08):
    """One chart per triangle, two triangles per grid cell (what a UV unwrapper yields in the limit): vt [3F,2], ft [F,3]."""
    F = f.shape[0]
    cells = int(np.ceil(np.sqrt((F + 1) // 2)))
    vt = np.zeros((3 * F, 2), np.float32)
    lo, hi = pad, 1.0 - pad
    for i in range(F):
        c = i // 2
        ox, oy = c % cells, c // cells
        if i % 2 == 0:
            tri = np.array([[lo, lo], [hi - pad, lo], [lo, hi - pad]], np.float32)
        else:
            tri = np.array([[hi, hi], [lo + pad, hi], [hi, lo + pad]], np.float32)
        vt[3 * i:3 * i + 3] = (tri + np.array([ox, oy], np.float32)) / cells
    ft = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
    return vt, ft


def tet_grid(n):
    """Regular tetrahedral grid on [-1,1]^3: (n+1)^3 vertices, every cube split into the 6 Kuhn tetrahedra around its main diagonal
    (stands in for demo/tets/{128,256}_tets.npz, which the reference tree does not ship -- SURVEY.md F12)."""
    import itertools
    g = np.linspace(-1, 1, n + 1, dtype=np.float32)
    pos = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
    vid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k
    ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij')
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    tets = []
    for perm in itertools.permutations(range(3)):
        cur = [ii.copy(), jj.copy(), kk.copy()]
        verts = [vid(*cur)]
        for ax in perm:
            cur[ax] = cur[ax] + 1
            verts.append(vid(*cur))
        tets.append(np.stack(verts, -1))
    tets = np.stack(tets, 1).reshape(-1, 4)                   # cube-major, 6 tets per cube
    return pos, tets.astype(np.int64)


def blob_sdf(pos, seed=0, radius=0.6, noise=0.08):
    """occupancy sign convention of the reference's DMTet: inside where sdf > 0."""
    rng = np.random.default_rng(seed)
    return (radius - np.linalg.norm(pos, axis=-1) + noise * rng.standard_normal(pos.shape[0])).astype(np.float32)
