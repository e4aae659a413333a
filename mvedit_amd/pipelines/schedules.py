"""Progress schedules of the 3D pipeline's outer loop (lib/pipelines/mvedit_3d_pipeline.py:41-78, the defaults of `__call__`'s schedule
arguments): how many views survive a step (which drives the per-rank re-partition of `mvedit_amd.parallel`), the render size (below 512 the
image enhancer runs), learning rates and the loss weights handed to `recon_loss.nerf_optim_loss` / `mesh_optim_loss`.  Scalar host
arithmetic; `progress` runs from 0 to 1 over the denoising steps."""


def _ramp(progress, start, end):
    return start + (end - start) * progress


def default_lr_multiplier(progress, progress_to_dmtet):
    """1 during the NeRF stage, then linear to 0 at the end of the DMTet stage"""
    return min((1 - progress) / (1 - progress_to_dmtet), 1)


def default_max_num_views(progress, progress_to_dmtet, start_num=32, mid_num=16, end_num=9, power=3):
    """views kept at this progress: a power-law decay from start_num to mid_num, times a linear factor that takes mid_num down to end_num
    over the DMTet stage (the caller rounds and prunes cameras, :1180-1215)"""
    ratio = end_num / mid_num
    decay = (start_num - mid_num) * (1 - progress) ** power + mid_num
    return decay * (default_lr_multiplier(progress, progress_to_dmtet) * (1 - ratio) + ratio)


def default_render_size_p(progress):
    return 128 if progress <= 0.3 else (256 if progress <= 0.6 else 512)


def default_lr_schedule(progress, start_lr=0.01, end_lr=0.005):
    return start_lr - (start_lr - end_lr) * progress


def default_patch_rgb_weight(progress, start_weight=0.3, end_weight=1.5):
    return _ramp(progress, start_weight, end_weight)


def default_patch_normal_weight(progress, start_weight=0.0, end_weight=3.0):
    return _ramp(progress, start_weight, end_weight)


def default_entropy_weight(progress, start_weight=0.0, end_weight=4.0):
    return start_weight - (start_weight - end_weight) * progress


def default_normal_reg_weight(progress, start_weight=4.0, end_weight=0.0):
    return start_weight - (start_weight - end_weight) * progress
