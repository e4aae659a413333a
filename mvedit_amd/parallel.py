"""View-parallel execution across the GPUs of one node (new design; the reference pipelines are
single-GPU: lib/pipelines/mvedit_3d_pipeline.py:11 imports torch.distributed and never calls it).

Views are independent in get_noise_pred (lib/pipelines/adapter3d_mixin.py:77-129) and in rendering
(lib/pipelines/mvedit_3d_pipeline.py:1341-1389); the 3D update consumes all views.  So: contiguous blocks of
views per rank, ONE all-gather per outer step (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests),
then every rank runs the identical seeded 3D update.  Camera pruning (mvedit_3d_pipeline.py:1180-1215)
shrinks V during the loop: partitions are recomputed from the current V every step.
"""
import torch
import torch.distributed as dist


def partition_views(num_views, world_size, rank):
    """Contiguous [lo, hi) block of views for `rank`; the first (num_views % world_size) ranks get one extra.
    Rank 0 always owns view 0 (the reference's keep_views are reordered to the front,
    mvedit_3d_pipeline.py:1149-1175)."""
    assert 0 <= rank < world_size and num_views >= 0
    q, r = divmod(num_views, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def set_partition_invariant(flag=True):
    """Bitwise equality of a view's result across ANY partition -- one GPU with 64 images against 8 ranks with 8 -- needs the strict K-slice mode
    of the GEMM dispatcher (mve_gemm_tune bit 30 / MVE_GEMM_STRICT_SPLITK=1: every launch rounds as the slice rule's slices, whatever the batch;
    about 3.8 ms per 64-image step on one GPU, nothing on launches that split anyway).  Without it the slice count of the deep levels depends on
    how many rows a launch has (csrc/gemm.hip: launch_gemm -- one chain where the launch fills the chip, `ceil(256 / tiles)` slices at the 8 x 8
    level: 8 slices at 8 images, 4 at 64, 2 at 128, 1 at 256), so results agree bitwise only among ranks / batches that take the same decisions:
    every rank of an N-GPU job whose shards are equal does, a 1-GPU run of the whole batch does not.  The values differ by fp32 summation order
    only (tests/test_unet_ops.py::test_unsplit_chain_vs_sliced_sum).  Returns the previous setting."""
    from . import _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    tune((old | (1 << 30)) if flag else (old & ~(1 << 30)))
    return bool(old & (1 << 30))


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def partition_sizes(num_views, world_size):
    return [partition_views(num_views, world_size, r)[1] - partition_views(num_views, world_size, r)[0]
            for r in range(world_size)]


def all_gather_views(local, num_views, group=None, force=False):
    """local: [v_local, ...] tensor of this rank's views -> [num_views, ...] on every rank.
    force: issue the collective even in a one-rank group (bench.py's RCCL smoke on a 1-GPU box).

    One collective.  Equal shards use all_gather_into_tensor (a single RCCL call writing straight into the
    output); ragged shards (V not divisible by the world size, e.g. after camera pruning 32 -> 9) pad to
    the largest shard and trim.
    """
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        assert local.shape[0] == num_views
        return local
    world = dist.get_world_size(group)
    sizes = partition_sizes(num_views, world)
    assert local.shape[0] == sizes[dist.get_rank(group)], (local.shape, sizes)
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((num_views,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    big = max(sizes)
    pad = torch.zeros((big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * big:r * big + s] for r, s in enumerate(sizes)], 0)


def repartition(per_view, old_num, keep_ids, group=None):
    """After camera pruning: per_view holds this rank's slice of `old_num` views; keep_ids (global indices, sorted)
    are the surviving views.  Returns this rank's slice of the new, smaller view set.  Implemented as
    all-gather + local slice: per-view state is small (latents 32 KiB/view) next to one UNet step."""
    full = all_gather_views(per_view, old_num, group)
    kept = full[torch.as_tensor(keep_ids, device=full.device, dtype=torch.long)]
    world = dist.get_world_size(group) if _dist_on() else 1
    rank = dist.get_rank(group) if _dist_on() else 0
    lo, hi = partition_views(len(keep_ids), world, rank)
    return kept[lo:hi].contiguous()


def sync_scene(tensors, src=0, group=None, force=False):
    """Re-align the replicated scene (hash table, MLP weights, density grid / bitfield, or SDF / deformation / texture) with rank `src`.

    The 3D update is replicated, not sharded, and several of its backward kernels scatter with float atomics (hash-table gradient, texture
    and vertex-attribute gradients, vertex-normal splat), as tiny-cuda-nn / nvdiffrast do in the reference: the ranks' scenes agree to
    rounding after one iteration, not bitwise, and hundreds of optimiser iterations per denoise step would let them drift apart.  One
    broadcast per outer step of the whole scene (about 52 MiB for the hash grid + 4 MiB of density grid: well under a millisecond over
    xGMI) removes the drift at its source.  ONE collective per dtype: the tensors are coalesced into a flat buffer and copied back in place.
    """
    if not _dist_on() or (dist.get_world_size(group) == 1 and not force):      # force: bench.py's one-rank RCCL smoke
        return tensors
    # `src` is a rank of `group`; dist.broadcast wants the global rank
    src_global = dist.get_global_rank(group, src) if group is not None else src
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src_global, group=group)
        off = 0
        with torch.no_grad():
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    return tensors
