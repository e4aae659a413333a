"""oracle/mesh_forward_oracle.py (the composition tests/test_mesh_ops.py checks the HIP MeshRenderer.forward against) vs the reference's own
`MeshRenderer.forward` EXECUTED over a stand-in `dr` module (tests/golden/mesh_forward_ref.npz, written by
tests/golden/make_mesh_forward_golden.py from base_mesh_renderer.py:207-395): textured mesh with antialias, 2x SSAA, and the
vertex-colour + shading_fun + edge-dilation path.  Run on the projected vertices the reference handed to dr.rasterize (recorded in the
golden file), so both sides rasterise bit-identical input."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import bake_oracle as BO
from oracle import mesh_forward_oracle as MF

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, 'golden', 'mesh_forward_ref.npz'))


def _mod():
    spec = importlib.util.spec_from_file_location('make_mesh_forward_golden', os.path.join(HERE, 'golden', 'make_mesh_forward_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_edge_dilation_restatement_is_bit_exact():
    g = np.load(os.path.join(HERE, 'golden', 'reference_py.npz'))
    assert np.array_equal(MF.edge_dilation(g['dil_img'], g['dil_mask'], 3, 7), g['dil_out_r3_i7'])
    assert np.array_equal(MF.edge_dilation(g['dil_img'], g['dil_mask'], 1, 2), g['dil_out_r1_i2'])


@pytest.mark.parametrize('tag,ssaa', [('tex_aa', 1), ('tex_aa_ssaa2', 2), ('vc_shade_dilate', 1), ('texmip_aa', 1), ('texmip_aa_ssaa2', 2)])
def test_mesh_forward_restatement_equals_reference_output(tag, ssaa):
    """texmip_*: the reference's default texture_filter 'linear-mipmap-linear' (a 256^2 atlas seen at 64^2 / 128^2)."""
    mod = _mod()
    v, f, vn, vt, ft, tex, vcol, poses, intr, S = mod.scene()
    v_cam, _ = BO.project(v, poses, intr * ssaa, S * ssaa, S * ssaa, 0.01, 100.0)
    r_c2w = np.concatenate([poses[:, :3, :1], -poses[:, :3, 1:3]], -1)
    kw = dict(vt=vt, ft=ft, albedo=G['tex_big'], texture_filter='linear-mipmap-linear') if tag.startswith('texmip') else dict(
        vt=vt, ft=ft, albedo=tex) if tag.startswith('tex') else dict(
        vc=vcol, shading_fun=mod.shade, aa=False,
        dilate=lambda rgba: MF.edge_dilation(rgba.transpose(0, 3, 1, 2), rgba.transpose(0, 3, 1, 2)[:, 3:], 1).transpose(0, 2, 3, 1))
    rgba, depth, normal = MF.mesh_forward(v, f, vn, f, (v_cam, G[f'{tag}_v_clip']), r_c2w, S, S, ssaa=ssaa, **kw)
    assert ((rgba[..., 3] > 0) & (rgba[..., 3] < 1)).mean() > 0.001 or tag == 'vc_shade_dilate'       # antialiased silhouettes exist
    if tag.startswith('texmip'):       # the filter matters: the bilinear fetch of the same atlas is far from the mip-mapped reference output
        lin = MF.mesh_forward(v, f, vn, f, (v_cam, G[f'{tag}_v_clip']), r_c2w, S, S, ssaa=ssaa, vt=vt, ft=ft, albedo=G['tex_big'])[0]
        assert np.abs(lin - G[f'{tag}_rgba']).max() > 0.1
    np.testing.assert_allclose(rgba, G[f'{tag}_rgba'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(depth, G[f'{tag}_depth'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(normal, G[f'{tag}_normal'], rtol=0, atol=3e-6)
