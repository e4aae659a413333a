"""CPU restatement of DMTet.__call__ (/root/reference/lib/models/decoders/mesh_renderer/base_mesh_renderer.py:104-188).
TEST INFRASTRUCTURE ONLY.  Pinned: tests/golden/dmtet_ref.npz holds the output of the reference class itself, executed on CPU
torch by tests/golden/make_dmtet_golden.py; tests/test_dmtet.py checks this restatement against it bit for bit."""
import numpy as np

TRIANGLE_TABLE = np.array([
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4], [3, 1, 5, -1, -1, -1],
    [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1], [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2],
    [1, 3, 5, -1, -1, -1], [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]], np.int64)
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], np.int64)
BASE_TET_EDGES = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], np.int64)


def dmtet(pos, sdf, tets):
    """pos [N,3] f32, sdf [N] f32, tets [F,4] int -> verts [Nv,3] f32, faces [Nf,3] int64  (:140-188)."""
    pos, sdf, tets = np.asarray(pos, np.float32), np.asarray(sdf, np.float32), np.asarray(tets, np.int64)
    occ = sdf > 0
    occ4 = occ[tets]
    occ_sum = occ4.sum(-1)
    valid = (occ_sum > 0) & (occ_sum < 4)
    all_edges = tets[valid][:, BASE_TET_EDGES].reshape(-1, 2)
    all_edges = np.sort(all_edges, axis=1)                                              # sort_edges
    unique_edges, idx_map = np.unique(all_edges, axis=0, return_inverse=True)
    idx_map = idx_map.reshape(-1)
    mask_edges = occ[unique_edges].sum(-1) == 1
    mapping = np.full(unique_edges.shape[0], -1, np.int64)
    mapping[mask_edges] = np.arange(mask_edges.sum())
    idx_map = mapping[idx_map].reshape(-1, 6)
    interp_v = unique_edges[mask_edges]
    p = pos[interp_v]                                                                   # [E,2,3]
    s = sdf[interp_v][..., None].copy()                                                 # [E,2,1]
    s[:, -1] *= np.float32(-1)
    den = s.sum(1, keepdims=True)
    w = (s[:, ::-1] / den).astype(np.float32)
    verts = (p * w).sum(1).astype(np.float32)
    tetindex = (occ4[valid] * (2 ** np.arange(4))).sum(-1)
    ntri = NUM_TRIANGLES[tetindex]
    one, two = ntri == 1, ntri == 2
    faces = np.concatenate([
        np.take_along_axis(idx_map[one], TRIANGLE_TABLE[tetindex[one]][:, :3], axis=1).reshape(-1, 3),
        np.take_along_axis(idx_map[two], TRIANGLE_TABLE[tetindex[two]][:, :6], axis=1).reshape(-1, 3)], axis=0)
    return verts, faces
