"""ORACLE -- test infrastructure only.  Restatement of `TriPlaneDecoder.point_decode` (lib/models/decoders/triplane_decoder.py:107-199) and
`TriPlaneiNGPDecoder.point_decode` (lib/models/decoders/triplane_ingp_decoder.py:142-212) for one scene in torch (float64 capable).
PINNED by tests/golden/triplane_ref.npz (tests/golden/make_triplane_golden.py executes the reference's own methods); the SH direction
encoding and the hash-grid encoding inside it come from oracle/sh_oracle.py (pinned) and oracle/nerf_oracle.py (tiny-cuda-nn: UNPINNED)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import nerf_oracle as NO
from . import sh_oracle as SH

ACT = dict(relu=torch.relu, silu=F.silu, softplus=F.softplus)


def point_decode(xyz, dirs, code, w, plane_cfg=('xy', 'xz', 'yz'), flip_z=False, activation='silu', sigmoid_saturation=0.001,
                 hash=None, density_only=False):
    """xyz [N,3], dirs [N,3] or None, code [3,C,h,w]; w: dict of torch tensors with the reference's Linear weights
    (base_w [H,3C], base_b, dens_w [1,H], dens_b, col1_w [H2,H+16], col1_b, col2_w [3,H2], col2_b, optionally ingp_w [H,2L], ingp_b);
    hash = dict(table [rows,2] numpy, n_levels, max_resolution, log2_hashmap_size, bound) for TriPlaneiNGPDecoder."""
    dt = code.dtype
    axis = dict(x=0, y=1, z=2)
    p = xyz.clone()
    if flip_z:
        p[:, 2] = -p[:, 2]
    grids = torch.stack([torch.stack([p[:, axis[pl[0]]], p[:, axis[pl[1]]]], -1) for pl in plane_cfg], 0)[:, None]        # [3,1,N,2]
    pc = F.grid_sample(code, grids.to(dt), mode='bilinear', padding_mode='border', align_corners=False)                    # [3,C,1,N]
    feat = pc[:, :, 0].permute(2, 1, 0).reshape(xyz.shape[0], -1)                                                          # [N, C*3], k = c*3 + p
    base = feat @ w['base_w'].T + w['base_b']
    if hash is not None:
        enc = NO.hashgrid_encode(((xyz + hash['bound']) / (2 * hash['bound'])).float().numpy(), hash['table'], hash['n_levels'],
                                 hash['max_resolution'], hash['bound'], hash['log2_hashmap_size'])
        base = base + torch.from_numpy(enc).to(dt) @ w['ingp_w'].T + w['ingp_b']
    a = ACT[activation](base)
    sigma = torch.exp(a @ w['dens_w'].T + w['dens_b'])[:, 0]                                                               # trunc_exp forward
    if density_only or dirs is None:
        return sigma, None
    sh = torch.from_numpy(SH.sh_encode(dirs.numpy(), 4)).to(dt)
    h2 = ACT[activation](torch.cat([a, sh], -1) @ w['col1_w'].T + w['col1_b'])
    rgb = torch.sigmoid(h2 @ w['col2_w'].T + w['col2_b'])
    return sigma, rgb * (1 + 2 * sigmoid_saturation) - sigmoid_saturation
