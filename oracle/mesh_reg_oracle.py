"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain torch, of the two mesh regularisers of the mesh-optimisation loop:
lib/models/decoders/mesh_renderer/base_mesh_renderer.py -- compute_edge_to_face_mapping (:20-52), normal_consistency (:55-68),
laplacian_uniform (:71-91), laplacian_smooth_loss (:94-101); call sites lib/pipelines/mvedit_3d_pipeline.py:775-776.

PINNED: tests/golden/mesh_reg_ref.npz holds the values, gradients and the edge-to-face table of the reference's OWN functions executed on
the CPU (tests/golden/make_mesh_reg_golden.py); tests/test_mesh_reg.py checks this restatement against them.  Differentiable, any dtype."""
import torch


def edge_to_face(faces):
    """-> (edges [E, 2] sorted lexicographically like torch.unique(dim=0), tris_per_edge [E, 2]): column 0 = the face that lists the edge as
    (min, max), column 1 = the face that lists it as (max, min), 0 where that side is missing (:41-50)."""
    f = faces.long()
    he = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], dim=1).view(-1, 2)          # packed by triangle (:24-28)
    side = (he[:, 0] > he[:, 1]).long()
    key = torch.minimum(he[:, 0], he[:, 1]) * (int(f.max()) + 1) + torch.maximum(he[:, 0], he[:, 1])
    uniq, inv = torch.unique(key, return_inverse=True)
    tris = torch.arange(f.shape[0], device=f.device).repeat_interleave(3)
    tpe = torch.zeros(uniq.shape[0], 2, dtype=torch.int64, device=f.device)
    for s in (0, 1):
        m = side == s
        tpe[inv[m], s] = tris[m]
    n = int(f.max()) + 1
    return torch.stack([uniq // n, uniq % n], dim=-1), tpe


def normal_consistency(face_normals, faces):
    _, tpe = edge_to_face(faces)
    term = (face_normals[tpe[:, 0]] * face_normals[tpe[:, 1]]).sum(-1, keepdim=True).clamp(-1.0, 1.0)
    return (1.0 - term).abs().mean()


def laplacian_smooth_loss(verts, faces):
    """mean_i |(D - A) v|_i with A the 0 / 1 adjacency of distinct neighbours (:71-101)"""
    f = faces.long()
    V = verts.shape[0]
    ii, jj = f[:, [1, 2, 0]].flatten(), f[:, [2, 0, 1]].flatten()
    adj = torch.stack([torch.cat([ii, jj]), torch.cat([jj, ii])], dim=0).unique(dim=1)
    A = torch.zeros(V, V, dtype=verts.dtype, device=verts.device)
    A[adj[0], adj[1]] = 1.0
    L = torch.diag(A.sum(1)) - A
    return (L @ verts).norm(dim=1).mean()
