"""Host-side camera bookkeeping that decides how many views a step has -- and therefore how `mvedit_amd.parallel` re-partitions
them across GPUs: mirrors of `get_camera_dists` / `prune_cameras`
(lib/pipelines/utils.py:350-379 of the reference, called at lib/pipelines/mvedit_3d_pipeline.py:1180-1215).  Pure torch, tiny
tensors ([V, V] with V <= 64): no kernel here, just the same decisions as the reference so that every rank prunes identically."""
import torch


def rotation_to_unit_quaternion(R):
    """R [..., 3, 3] -> unit quaternions [..., 4] (w, x, y, z), best-conditioned branch of Shepperd's method.  The sign of the
    result is arbitrary; callers only use |q1 . q2|."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    four = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
    cand = torch.stack([
        torch.stack([four[..., 0], m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, four[..., 1], m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, four[..., 2], m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, four[..., 3]], dim=-1)], dim=-2)
    best = four.argmax(dim=-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return q / q.norm(dim=-1, keepdim=True)


def get_camera_dists(camera_poses, cam_weights, device):
    """Pairwise camera distance = position distance + 4 * half the relative rotation angle, scaled by the row camera's weight,
    with a huge diagonal so that a camera is never its own nearest neighbour (lib/pipelines/utils.py:350-363)."""
    n = camera_poses.size(0)
    pos = camera_poses[:, :3, 3]
    q = rotation_to_unit_quaternion(camera_poses[:, :3, :3])
    half_theta = torch.acos((q @ q.t()).abs().clamp(max=1))
    dists = (pos[:, None] - pos[None]).norm(dim=-1) + 4 * half_theta
    if cam_weights is not None:
        dists = dists * cam_weights[:, None]
    return dists + 999999 * torch.eye(n, dtype=dists.dtype, device=device)


def prune_cameras(dists, num_keep_views, max_num_cameras, device, pixel_dist=None):
    """Greedily drop the view closest to another view (never one of the first num_keep_views) until max_num_cameras remain
    (lib/pipelines/utils.py:366-379).  Returns (keep_ids into the original numbering, reduced dists)."""
    keep_ids = torch.arange(dists.size(0), device=device)
    if pixel_dist is not None:
        pixel_dist = pixel_dist.clone()
    for _ in range(dists.size(0) - max_num_cameras):
        importance = dists[num_keep_views:].amin(dim=1)
        if pixel_dist is not None:
            importance = importance - pixel_dist[num_keep_views:] * 0.05
        remove = int(importance.argmin()) + num_keep_views
        mask = torch.arange(len(keep_ids), device=device) != remove
        keep_ids, dists = keep_ids[mask], dists[mask][:, mask]
        if pixel_dist is not None:
            pixel_dist = pixel_dist[mask]
    return keep_ids, dists


# ---------------------------------------------------------------------------------------------------------------------------------
# torchvision's gaussian_blur and the reference's `highpass` (lib/pipelines/utils.py:187-188) on the native separable kernel
# (csrc/blur.hip), differentiable w.r.t. the image like the torch expressions they replace.  CUDA tensors only: no fallback.
class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ksize, sigma, highpass_offset):
        from .. import _lib
        assert x.is_cuda and x.dim() >= 2, 'native path: CUDA tensors [..., H, W]'
        xf = x.detach().to(torch.float32).contiguous()
        H, W = xf.shape[-2:]
        out, tmp = torch.empty_like(xf), torch.empty_like(xf)
        hp = highpass_offset is not None
        with torch.cuda.device(xf.device):
            _lib.call('mve_gaussian_blur', _lib.ptr(xf), xf.numel() // (H * W), H, W, int(ksize), float(sigma), 0, _lib.ptr(xf) if hp else None,
                      float(highpass_offset or 0.0), _lib.ptr(tmp), _lib.ptr(out), _lib.stream_ptr(xf.device))
        ctx.args, ctx.dtype = (int(ksize), float(sigma), hp), x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        ksize, sigma, hp = ctx.args
        gf = g.detach().to(torch.float32).contiguous()
        H, W = gf.shape[-2:]
        out, tmp = torch.empty_like(gf), torch.empty_like(gf)
        with torch.cuda.device(gf.device):
            _lib.call('mve_gaussian_blur', _lib.ptr(gf), gf.numel() // (H * W), H, W, ksize, sigma, 1, _lib.ptr(gf) if hp else None, 0.0,
                      _lib.ptr(tmp), _lib.ptr(out), _lib.stream_ptr(gf.device))
        return out.to(ctx.dtype), None, None, None


def gaussian_blur(img, kernel_size, sigma):
    """torchvision.transforms.functional.gaussian_blur(img [..., H, W], kernel_size (odd int), sigma (float)); reflect padding."""
    return _BlurFn.apply(img, kernel_size, sigma, None)


def highpass(x, std=5, offset=0.5):
    """lib/pipelines/utils.py:187-188: offset + x - gaussian_blur(x, int(round(std)) * 6 + 1, std), one fused pass pair"""
    return _BlurFn.apply(x, int(round(std)) * 6 + 1, std, float(offset))


def init_tet(decoder, tet_vertices, tet_indices, density_thresh=5.0):
    """NeRF -> DMTet hand-over (lib/pipelines/utils.py:156-184 `init_tet`; once per request, when the loop switches to the mesh stage):
    scale the tetrahedral grid to the bounding box of the NeRF's occupied region and initialise the SDF from its density.

    decoder: anything with `point_decode(xyzs [N,3], density_only=True) -> (sigma [N], None)` (mvedit_amd.nerf.INGPDecoderParams);
    tet_vertices [Nv,3] / tet_indices [Nt,4]: the contents of the reference's `demo/tets/{resolution}_tets.npz` ('vertices', 'indices'),
    which it downloads (no network here: the caller supplies them).  -> (verts [Nv,3] f32, indices [Nt,4] i64, sdf [Nv] f32)"""
    verts = -torch.as_tensor(tet_vertices, dtype=torch.float32) * 2                     # covers [-1, 1]
    dev = getattr(decoder, 'device', verts.device)
    verts = verts.to(dev)
    indices = torch.as_tensor(tet_indices).to(device=dev, dtype=torch.long)
    with torch.no_grad():
        occupied = verts[decoder.point_decode(verts, density_only=True)[0] > density_thresh]
        hi, lo = occupied.amax(dim=0) + 0.1, occupied.amin(dim=0) - 0.1
        verts = verts * ((hi - lo).max() / 2) + (hi + lo) / 2
        sdf = (decoder.point_decode(verts, density_only=True)[0] - density_thresh).clamp(-1, 1)
        sdf[(verts < -1).any(dim=-1) | (verts > 1).any(dim=-1)] = -1
    return verts, indices, sdf


def do_segmentation(in_imgs, seg_model, padding=0, bg_color=None, color_threshold=0.25):
    """Tensor path of the reference's do_segmentation (lib/pipelines/utils.py:73-107; the SAM refinement and the numpy / PIL conveniences
    are outside the loop): in_imgs [N, 3, H, W] float in [0, 1] -> [N, 4, H, W] = images + foreground mask.  `seg_model` is a
    `mvedit_amd.segmentor.TracerUniversalB7Engine` (or anything with its call signature).  Replicate padding helps the model find objects
    that touch the border; pixels that differ from the background colour by more than the threshold in some channel are foreground
    regardless of the model."""
    assert in_imgs.size(1) == 3
    dev = getattr(seg_model, 'device', in_imgs.device)
    x = in_imgs.to(dev)
    if padding > 0:
        masks = seg_model(torch.nn.functional.pad(x, (padding, padding, padding, padding), mode='replicate'))[:, :, padding:-padding, padding:-padding]
    else:
        masks = seg_model(x)
    masks = masks.to(x.dtype).clone()
    if bg_color is not None:
        bg = x.new_tensor(bg_color)[..., None, None]
        non_fg = torch.all(bg - color_threshold <= x, dim=1) & torch.all(x <= bg + color_threshold, dim=1)
        masks[~non_fg.unsqueeze(1)] = 1
    return torch.cat([x, masks], dim=1)
