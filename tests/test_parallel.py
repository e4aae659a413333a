"""View partitioning + the single all-gather of the multi-GPU design, exercised with world_size 2 and 3 on
CPU (gloo).  On the GPU box the same code runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvedit_amd.parallel import partition_views, partition_sizes, all_gather_views, repartition


def test_partition_covers_views_exactly():
    for V in (0, 1, 6, 9, 16, 32, 33, 64):
        for W in (1, 2, 3, 4, 8):
            blocks = [partition_views(V, W, r) for r in range(W)]
            assert blocks[0][0] == 0 and blocks[-1][1] == V
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = partition_sizes(V, W)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == V


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, keep):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = torch.arange(V * 4 * 2 * 2, dtype=torch.float32).reshape(V, 4, 2, 2)
        lo, hi = partition_views(V, world, rank)
        got = all_gather_views(full[lo:hi].clone(), V)
        assert torch.equal(got, full), (rank, got.shape)
        # camera pruning: keep a subset, re-partition; the union over ranks must be full[keep]
        mine = repartition(full[lo:hi].clone(), V, keep)
        lo2, hi2 = partition_views(len(keep), world, rank)
        assert torch.equal(mine, full[keep][lo2:hi2])
        # view-parallel "denoise": each rank transforms its own views, one gather, every rank has the full step result
        step = all_gather_views(full[lo:hi] * 2 + 1, V)
        assert torch.equal(step, full * 2 + 1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,V', [(2, 32), (2, 9), (3, 16)])
def test_all_gather_and_repartition_gloo(world, V):
    keep = sorted(set(range(0, V, 2)) | {1})
    mp.spawn(_worker, args=(world, _free_port(), V, keep), nprocs=world, join=True)


GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'reference_py.npz')


def test_camera_pruning_matches_reference_output():
    """get_camera_dists / prune_cameras (lib/pipelines/utils.py:350-379) against outputs of the reference's own functions
    (tests/golden/make_reference_py_golden.py): same distances, same pruning decisions."""
    import numpy as np
    from mvedit_amd.pipelines.utils import get_camera_dists, prune_cameras
    g = np.load(GOLD)
    poses, cw, pd = (torch.from_numpy(g[k]) for k in ('prune_poses', 'prune_cam_weights', 'prune_pixel_dist'))
    d = get_camera_dists(poses, cw, 'cpu')
    np.testing.assert_allclose(d.numpy(), g['prune_dists'], rtol=0, atol=2e-5)
    k16, d16 = prune_cameras(torch.from_numpy(g['prune_dists']).clone(), 1, 16, 'cpu')
    k9, d9 = prune_cameras(torch.from_numpy(g['prune_dists']).clone(), 4, 9, 'cpu', pixel_dist=pd)
    assert (k16.numpy() == g['prune_keep_16']).all() and (k9.numpy() == g['prune_keep_9']).all()
    assert (d16.numpy() == g['prune_dists_16']).all() and (d9.numpy() == g['prune_dists_9']).all()
    assert set(range(4)) <= set(k9.tolist()), 'keep_views are never pruned'


def _prune_worker(rank, world, port):
    """Every rank prunes from the same replicated camera set (identical decisions), then the per-view state is re-partitioned."""
    import numpy as np
    from mvedit_amd.pipelines.utils import get_camera_dists, prune_cameras
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = np.load(GOLD)
        poses, cw = torch.from_numpy(g['prune_poses']), torch.from_numpy(g['prune_cam_weights'])
        V = poses.shape[0]
        keep, _ = prune_cameras(get_camera_dists(poses, cw, 'cpu'), 1, 16, 'cpu')
        gathered = [torch.empty_like(keep) for _ in range(world)]
        dist.all_gather(gathered, keep)
        assert all(torch.equal(k, keep) for k in gathered), 'ranks must agree on the surviving views'
        latents = torch.arange(V * 4, dtype=torch.float32).reshape(V, 4)
        lo, hi = partition_views(V, world, rank)
        mine = repartition(latents[lo:hi].clone(), V, keep.tolist())
        lo2, hi2 = partition_views(len(keep), world, rank)
        assert torch.equal(mine, latents[keep][lo2:hi2])
    finally:
        dist.destroy_process_group()


def test_prune_then_repartition_gloo():
    mp.spawn(_prune_worker, args=(2, _free_port()), nprocs=2, join=True)


def _sync_worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mvedit_amd.parallel import sync_scene
        g = torch.Generator().manual_seed(7)
        base = [torch.rand(1000, 2, generator=g), torch.rand(16, 64, generator=g), torch.randint(0, 255, (4096,), generator=g, dtype=torch.uint8),
                torch.rand(33, generator=g).half()]
        # every rank's replica has drifted by its own rounding-sized perturbation, as float-atomic gradients make it
        mine = [t.clone() if t.dtype == torch.uint8 else t + (rank * 1e-4) for t in base]
        mine[2] = (mine[2].int() + rank).clamp(max=255).to(torch.uint8)
        ptrs = [t.data_ptr() for t in mine]
        out = sync_scene(mine, src=0)
        assert all(o.data_ptr() == p for o, p in zip(out, ptrs))                      # in place: optimiser state keeps pointing at them
        for o, b in zip(out, base):
            assert torch.equal(o, b), (rank, o.dtype)                                  # every rank now holds rank 0's scene bit for bit
    finally:
        dist.destroy_process_group()


def test_sync_scene_gloo():
    mp.spawn(_sync_worker, args=(3, _free_port()), nprocs=3, join=True)
    from mvedit_amd.parallel import sync_scene
    t = [torch.ones(3)]
    assert sync_scene(t) is t                                                          # no process group: a no-op


# ---------------------------------------------------------------------------------------------------------------------------------------------
# One whole sharded OUTER step on a stand-in engine: partition -> per-rank noise prediction -> the one all-gather ->
# replicated 3D update -> scene re-sync -> camera pruning 32 -> 16 -> 9 -> repartition, for three outer steps, against the 1-rank run.
# ---------------------------------------------------------------------------------------------------------------------------------------------
def _stub_noise_pred(latents, t):
    """Per-view function of the view's own latent and the timestep (like get_noise_pred: views are independent, adapter3d_mixin.py:77-129)."""
    return torch.tanh(latents * 0.7 + 0.01 * t) - 0.1 * latents.mean(dim=(1, 2, 3), keepdim=True)


def _stub_render(scene, view_ids):
    """Per-view 'rendered maps' from the replicated scene (8 channels, as the design's RGBD + normal payload)."""
    base = scene['table'][:64].reshape(8, 4, 4)
    return torch.stack([base * (1.0 + 0.01 * int(v)) + scene['mlp'].sum() for v in view_ids])


def _outer_loop(world, rank, use_dist):
    from mvedit_amd.parallel import sync_scene
    g = torch.Generator().manual_seed(0)
    V = 32
    latents_all = torch.randn(V, 4, 8, 8, generator=g)
    scene = dict(table=torch.randn(256, 2, generator=g), mlp=torch.randn(8, 8, generator=g))
    view_ids = list(range(V))
    lo, hi = partition_views(V, world, rank)
    mine = latents_all[lo:hi].clone()
    log = []
    for step, (t, keep_n) in enumerate([(981, 16), (741, 9), (499, 9)]):
        Vc = len(view_ids)
        lo, hi = partition_views(Vc, world, rank)
        assert mine.shape[0] == hi - lo
        noise = _stub_noise_pred(mine, t)                                   # sharded: this rank's views only
        mine = mine - 0.1 * noise                                           # scheduler state stays local
        maps = _stub_render(scene, view_ids[lo:hi])                         # sharded render of the replicated scene
        maps_all = all_gather_views(maps, Vc) if use_dist else maps         # THE collective of the step
        assert maps_all.shape[0] == Vc
        # replicated 3D update: every rank the same seeded arithmetic over all views' maps; a rank-dependent perturbation stands for the float-atomic
        # scatter order of the real backward kernels (ranks agree to rounding, not bitwise) ...
        upd = maps_all.mean(dim=(0, 2, 3))
        scene['table'] = scene['table'] + 0.01 * upd.sum() + (1e-7 * rank if use_dist else 0.0)
        scene['mlp'] = scene['mlp'] * 0.99 + 0.001 * upd[:8].reshape(8, 1)
        # ... which the one broadcast per outer step removes (the second, stated collective)
        if use_dist:
            sync_scene([scene['table'], scene['mlp']], src=0)
        # camera pruning (mvedit_3d_pipeline.py:1180-1215): keep the keep_n views with the largest map energy, view 0 always
        if keep_n < Vc:
            score = maps_all.flatten(1).norm(dim=1)
            score[0] = float('inf')
            keep = sorted(score.topk(keep_n).indices.tolist())
            mine = repartition(mine, Vc, keep) if use_dist else mine[keep]
            view_ids = [view_ids[k] for k in keep]
        log.append((list(view_ids), scene['table'].clone(), scene['mlp'].clone()))
    lo, hi = partition_views(len(view_ids), world, rank)
    return log, mine, (lo, hi)


def _outer_worker(rank, world, port, ref_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ref_log, ref_latents = torch.load(ref_path)
        log, mine, (lo, hi) = _outer_loop(world, rank, True)
        for (ids, table, mlp), (rids, rtable, rmlp) in zip(log, ref_log):
            assert ids == rids, (rank, ids, rids)                            # the same cameras survive on every rank
            assert torch.equal(table, rtable) and torch.equal(mlp, rmlp), rank  # rank 0's scene everywhere == the 1-rank run's scene
        assert torch.equal(mine, ref_latents[lo:hi]), rank                   # this rank's latents == its block of the 1-rank run's
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_outer_steps_equal_the_one_rank_run(world, tmp_path):
    ref_log, ref_latents, _ = _outer_loop(1, 0, False)
    path = str(tmp_path / 'ref.pt')
    torch.save((ref_log, ref_latents), path)
    mp.spawn(_outer_worker, args=(world, _free_port(), path), nprocs=world, join=True)
