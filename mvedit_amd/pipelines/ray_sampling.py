"""Host-side mirrors of the ray / patch selection that feeds every NeRF optimisation iteration: `BaseNeRF.ray_sample` and
`BaseNeRF.get_raybatch_inds` (lib/models/autoencoders/base_nerf.py:245-322; called at lib/pipelines/mvedit_3d_pipeline.py:496-523).
Pure indexing over tensors that already live in HBM -- no kernel of its own -- but the ORDER matters: the patches come out
(image, patch row, patch column)-major with pixels row-major inside a patch, which is the layout `recon_loss.nerf_optim_loss` and the LPIPS
patch loss consume, and the random permutations are drawn with the same calls in the same order as the reference, so a seeded run selects
the same rays on every rank ("Randomness")."""
import torch


def _patchify(t, ps):
    """[S, I, h, w, C] -> [S, I * (h / ps) * (w / ps), ps, ps, C]"""
    S, I, h, w, C = t.shape
    t = t.reshape(S, I, h // ps, ps, w // ps, ps, C)
    return t.permute(0, 1, 2, 4, 3, 5, 6).reshape(S, I * (h // ps) * (w // ps), ps, ps, C)


def get_raybatch_inds(cond_imgs, n_inverse_rays, patch_size=None):
    """cond_imgs [S, I, h, w, 3].  -> (tuple of [S, units per batch] index tensors, number of batches), or (None, None) when one batch
    holds every ray.  Units are rays (patch_size None: `patch_loss is None`) or patches."""
    S, I, h, w, _ = cond_imgs.shape
    pixels = I * h * w
    if pixels <= n_inverse_rays:
        return None, None
    unit = 1 if patch_size is None else patch_size ** 2
    perms = torch.stack([torch.randperm(pixels // unit, device=cond_imgs.device) for _ in range(S)], dim=0)
    batches = perms.split(n_inverse_rays // unit, dim=1)
    return batches, len(batches)


def ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_samples, sample_inds=None, cond_extras=None, patch_size=None):
    """cond_* [S, I, h, w, C]; sample_inds [S, units] or None (drawn here).  -> rays_o [S, n, 3], rays_d [S, n, 3], target_rgbs
    ([S, n, 3] without patches, [S * patches, ps, ps, 3] with), then one tensor per entry of cond_extras in the target layout."""
    S, I, h, w, _ = cond_rays_o.shape
    pixels = I * h * w
    extras = list(cond_extras or [])
    if patch_size is None:
        group = lambda t: t.reshape(S, pixels, t.shape[-1])
        units = n_samples
    else:
        assert n_samples % (patch_size ** 2) == 0
        group = lambda t: _patchify(t, patch_size)
        units = n_samples // patch_size ** 2
    rays_o, rays_d, target, extras = group(cond_rays_o), group(cond_rays_d), group(cond_imgs), [group(e) for e in extras]
    if pixels > n_samples:
        if sample_inds is None:
            sample_inds = torch.stack([torch.randperm(target.shape[1], device=cond_rays_o.device)[:units] for _ in range(S)], dim=0)
        pick = lambda t: torch.gather(t, 1, sample_inds.reshape(S, -1, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:]))
        rays_o, rays_d, target, extras = pick(rays_o), pick(rays_d), pick(target), [pick(e) for e in extras]
    if patch_size is not None:
        rays_o, rays_d = rays_o.reshape(S, -1, 3), rays_d.reshape(S, -1, 3)
        target = target.reshape(-1, patch_size, patch_size, 3)
        extras = [e.reshape(-1, patch_size, patch_size, e.shape[-1]) for e in extras]
    return (rays_o, rays_d, target, *extras)
