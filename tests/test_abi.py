"""The C-ABI library builds, loads, and exports every symbol include/mvedit_amd.h declares (no compute)."""
import ctypes
import os
import subprocess

import pytest


def test_library_exports_every_declared_symbol(lib):
    protos = lib.parse_header()
    assert len(protos) >= 25
    out = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [n for n in protos if n not in exported]
    assert not missing, f'declared in mvedit_amd.h but not exported: {missing}'
    stray = [n for n in exported if n.startswith('mve_') and n not in protos]
    assert not stray, f'exported but not declared in mvedit_amd.h: {stray}'


def test_version_and_error_plumbing(lib):
    assert lib.raw('mve_version')() == 1
    # argument validation fails loudly without touching a GPU
    rc = lib.raw('mve_gemm')(1, None, 8, None, 8, None, 8, 16, 12, 8, None, None, 0, 0, None, 0, 0, ctypes.c_float(1.0), None, 0, 0, None)
    assert rc == -1 and 'multiples of 8' in lib.last_error()
    rc = lib.raw('mve_attention')(1, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8, None, 0, None, 0,
                                  ctypes.c_void_p(16), 8, 1, 4, 4, 0, 8, 48, ctypes.c_float(1.0), None)
    assert rc == -1 and 'head dim' in lib.last_error()
    assert lib.raw('mve_march_scratch_bytes')(1000) >= 4 * 4


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.build()/smoke() and bench.py's cpu_baseline leg may touch it.
    The package and the tools never do; bench.py does so in exactly one function."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in fn.names] if isinstance(fn, ast.Import) else [fn.module or '']
                if any(n == 'oracle' or n.startswith('oracle.') for n in names):
                    hits.append(fn.lineno)
        return hits

    for d, _, files in os.walk(os.path.join(root, 'mvedit_amd')):
        for f in files:
            if f.endswith('.py'):
                assert not oracle_imports(os.path.join(d, f)), f'{f} imports the oracle'
    for f in os.listdir(os.path.join(root, 'tools')):
        if f.endswith('.py'):
            assert not oracle_imports(os.path.join(root, 'tools', f)), f'tools/{f} imports the oracle'
    tree = ast.parse(open(os.path.join(root, 'bench.py')).read())
    owners = {fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
              for node in ast.walk(fn) if isinstance(node, ast.ImportFrom) and (node.module or '').startswith('oracle')}
    assert owners == {'cpu_baseline'}, owners
    # no C/HIP source of the product #includes anything from oracle/ (comments may cite oracle files)
    import re
    for f in os.listdir(os.path.join(root, 'mvedit_amd', 'csrc')):
        for inc in re.findall(r'#include\s+[<"]([^>"]+)[>"]', open(os.path.join(root, 'mvedit_amd', 'csrc', f)).read()):
            assert 'oracle' not in inc, (f, inc)


def test_every_python_call_site_passes_the_declared_number_of_arguments():
    """ctypes would accept a call with a missing trailing argument and read garbage for it: check every `_lib.call('mve_x', ...)` /
    `_lib.raw('mve_x')(...)` in the package, the tools, bench.py and the tests against the prototypes parsed from include/mvedit_amd.h."""
    import ast
    import glob
    from mvedit_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, 'mvedit_amd', '**', '*.py'), recursive=True) + glob.glob(os.path.join(root, 'tools', '*.py')) \
        + glob.glob(os.path.join(root, 'tests', '*.py')) + [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')]
    checked, problems = 0, []
    for path in files:
        for node in ast.walk(ast.parse(open(path).read())):
            if not isinstance(node, ast.Call) or any(isinstance(a, ast.Starred) for a in node.args):
                continue
            f, name = node.func, None
            if isinstance(f, ast.Attribute) and f.attr == 'call' and node.args and isinstance(node.args[0], ast.Constant) \
                    and isinstance(node.args[0].value, str):
                name, nargs = node.args[0].value, len(node.args) - 1
            elif isinstance(f, ast.Call) and isinstance(f.func, ast.Attribute) and f.func.attr == 'raw' and f.args \
                    and isinstance(f.args[0], ast.Constant):
                name, nargs = f.args[0].value, len(node.args)
            if name is None or not str(name).startswith('mve_'):
                continue
            checked += 1
            if name not in _lib.PROTOS:
                problems.append(f'{path}:{node.lineno} calls undeclared {name}')
            elif len(_lib.PROTOS[name][1]) != nargs:
                problems.append(f'{path}:{node.lineno} {name}: passes {nargs} arguments, the header declares {len(_lib.PROTOS[name][1])}')
    assert checked > 100 and not problems, problems


def test_host_side_dispatch_knobs_round_trip_and_do_not_depend_on_the_batch():
    """Pure host logic behind the C ABI, no GPU: the tuning words come back whole (old = tune(x); ...; tune(old) restores every switch), and
    the cuts that decide a reduction's summation order -- the depthwise kernel's pixel slabs, GroupNorm's row split -- are functions of
    the layer only, never of the batch (batch / chunk invariance by construction)."""
    from mvedit_amd import _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        for word in (256, 1, 0, 256 | (1 << 26), 256 | (1 << 28), 64 | (1 << 27) | (1 << 29), 256 | (1 << 25), 256 | (1 << 30), 1 | (1 << 30) | (1 << 27)):
            tune(word)
            assert tune(-1) == word, hex(word)
    finally:
        tune(old)
    assert tune(-1) == old
    gn = _lib.raw('mve_groupnorm_tune')
    prev = gn(-1)
    try:
        assert gn(0) == prev and gn(-1) == 0 and gn(4096) == 0 and gn(-1) == 4096
    finally:
        gn(prev)
    slabs = _lib.raw('mve_seg_dwconv_slabs')
    for (ho, c, k, st) in ((320, 64, 3, 1), (160, 192, 3, 2), (160, 288, 3, 1), (80, 480, 5, 1), (40, 960, 3, 1), (40, 1344, 5, 1), (20, 2304, 5, 1), (20, 3840, 3, 1)):
        n = [slabs(b, ho, ho, c, k, st) for b in (1, 2, 8, 32)]
        assert len(set(n)) == 1 and 1 <= n[0] <= 256, (ho, c, k, st, n)
    ws = _lib.raw('mve_groupnorm_workspace_bytes')
    for (hw, c) in ((4096, 320), (4096, 960), (1024, 640), (256, 1280), (64, 2560), (262144, 128)):
        per_image = [(ws(b, hw, c, 32) - 64) // b for b in (1, 2, 8, 64)]      # partials + statistics scale with the batch and with nothing else
        assert max(per_image) - min(per_image) <= 64, (hw, c, per_image)


def test_slice_decisions_of_the_gemm_dispatcher():
    """Host logic of round 4's dispatcher (csrc/gemm.hip): the slice RULE depends on the layer only; what a launch actually runs with depends on
    whether it fills the chip -- one chain where the un-split launch does (64 images at the 32 x 32 / 16 x 16 levels), just enough slices where
    the rule would over-fill it (64 images at 8 x 8: 64 tiles x 8 slices -> 4), the rule's count for small batches and in the strict mode."""
    from mvedit_amd import _lib
    eff = _lib.raw('mve_gemm_effective_splitk')
    ws = _lib.raw('mve_gemm_workspace_bytes')
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        tune(256)
        rule = lambda M, N, K, rpi: ws(M, N, K, rpi) // (M * N * 4) if ws(M, N, K, rpi) else 1
        for rpi, N, K in ((1024, 640, 5760), (256, 1280, 11520), (64, 1280, 11520), (256, 1280, 5120)):
            assert len({rule(B * rpi, N, K, rpi) for B in (1, 2, 8, 64)}) == 1, 'the rule never looks at the batch'
        assert rule(64 * 1024, 640, 5760, 1024) == 2 and eff(64 * 1024, 640, 5760, 1024) == 1         # 32 x 32 level, 64 images: one chain
        assert rule(64 * 256, 1280, 11520, 256) == 4 and eff(64 * 256, 1280, 11520, 256) == 1         # 16 x 16 level
        assert rule(64 * 64, 1280, 11520, 64) == 8 and eff(64 * 64, 1280, 11520, 64) == 4             # 8 x 8 level: 64 tiles -> 4 slices fill 256 CUs
        for B in (1, 2, 8):                                                                          # small batches (every rank of an 8-GPU job): the rule
            for rpi, N, K in ((1024, 640, 5760), (256, 1280, 11520), (64, 1280, 11520)):
                assert eff(B * rpi, N, K, rpi) == rule(B * rpi, N, K, rpi), (B, rpi)
        tune(256 | (1 << 30))                                                                        # strict: the rule at any batch
        for rpi, N, K in ((1024, 640, 5760), (256, 1280, 11520), (64, 1280, 11520)):
            assert eff(64 * rpi, N, K, rpi) == rule(64 * rpi, N, K, rpi)
        assert eff(64 * 4096, 320, 2880, 4096) == 1                                                  # 64 x 64 level: the rule itself never splits
    finally:
        tune(old)


def test_upsample_phase_conv_shape_rule():
    """mve_upsample_conv_phases_supported: a function of the shape -- 64-channel slabs, an output width the 256-row tiles come in (multiples of
    128), a power-of-two source width (grouped output rows), at least 64 source pixels in the launch."""
    from mvedit_amd import _lib
    ok = _lib.raw('mve_upsample_conv_phases_supported')
    for C, H in ((1280, 8), (1280, 16), (640, 32)):            # the UNet's three upsamplers at a 64 x 64 latent, any batch
        assert all(ok(C, C, B, H, H) == 1 for B in (1, 2, 64))
    for C, H in ((512, 64), (512, 128), (256, 256)):           # the VAE decoder's at 512 x 512
        assert ok(C, C, 1, H, H) == 1
    assert ok(1280, 1280, 1, 4, 4) == 0 and ok(1280, 1280, 4, 4, 4) == 1        # 16 source pixels per image: from batch 4 on
    assert ok(1280, 1280, 2, 12, 12) == 0                      # 96 x 96 latent: source width 12
    assert ok(96, 128, 2, 8, 8) == 0 and ok(128, 200, 2, 8, 8) == 0
    assert _lib.raw('mve_upsample_conv_phases_workspace_bytes')(1280, 1280, 2, 8, 8) == _lib.raw("mve_gemm_workspace_bytes")(4 * 128, 1280, 5120, 4 * 64)


def test_effective_splitk_by_batch(lib):
    """the K-slice count a launch RUNS with depends on its rows, not only on the rule (rows per image, N, K).  Pin the decisions at
    the batches that matter -- 8 images (a rank of an 8-GPU job), 64 (one GPU, the benchmark), 128, 256 -- in the default and in the strict mode
    (MVE_GEMM_STRICT_SPLITK / mve_gemm_tune bit 30 / parallel.set_partition_invariant), which reports the rule's count at every batch."""
    from mvedit_amd import parallel
    esk = lib.raw('mve_gemm_effective_splitk')
    shapes = dict(conv_16x16=(256, 1280, 9 * 1280), conv_8x8=(64, 1280, 9 * 1280), conv_32x32=(1024, 640, 9 * 640), ff_out_8x8=(64, 1280, 5120))
    got = {k: [esk(B * r, N, K, r) for B in (2, 8, 32, 64, 128, 256)] for k, (r, N, K) in shapes.items()}
    assert got['conv_16x16'] == [4, 4, 2, 1, 1, 1]
    assert got['conv_8x8'] == [8, 8, 8, 4, 2, 1]
    assert got['conv_32x32'] == [2, 2, 1, 1, 1, 1]
    assert got['ff_out_8x8'] == [8, 8, 8, 4, 2, 1]
    assert parallel.set_partition_invariant(True) is False
    try:
        strict = {k: [esk(B * r, N, K, r) for B in (2, 8, 32, 64, 128, 256)] for k, (r, N, K) in shapes.items()}
        assert all(len(set(v)) == 1 for v in strict.values()), strict      # the rule's count at every batch
        assert [v[0] for v in strict.values()] == [4, 8, 2, 8]
    finally:
        assert parallel.set_partition_invariant(False) is True
    assert esk(64 * 256, 1280, 9 * 1280, 256) == 1


def test_slice_rule_rounds_up_and_keeps_power_of_two_tile_counts(lib):
    """Round 6: the slice rule is ceil(64 / tiles of one image) (capped by 8 K tiles per slice and 16).  Tile counts that divide 64 -- every level of
    a 64 x 64 latent, the shapes every pinned result of rounds 1-5 was produced with -- keep their counts; the 30 x 20 level of Zero123++'s 120 x 80
    latent (600 rows = 5 row tiles x 8 column tiles = 40 tiles) gets 2 slices instead of 1 (a CFG pair ran 80 blocks through K = 11 520 alone)."""
    ws = lib.raw('mve_gemm_workspace_bytes')
    slices = lambda rows, N, K: max(1, ws(rows, N, K, rows) // (rows * N * 4))          # workspace = slices x M x N fp32 (0 = unsliced)
    # 64 x 64 latent: 4096 / 1024 / 256 / 64 rows per image
    assert [slices(r, n, 9 * n) for r, n in ((4096, 320), (1024, 640), (256, 1280), (64, 1280))] == [1, 2, 4, 8]
    assert [slices(r, 1280, 1280) for r in (256, 64)] == [2, 2]                         # (K = 1280: at least 8 K tiles per slice)
    # 120 x 80 latent (Zero123++): 9600 / 2400 / 600 / 150 rows per image
    assert [slices(r, n, 9 * n) for r, n in ((9600, 320), (2400, 640), (600, 1280), (150, 1280))] == [1, 1, 2, 4]
    assert ws(2 * 600, 1280, 9 * 1280, 600) == 2 * ws(600, 1280, 9 * 1280, 600)          # a function of the rows PER IMAGE: batch invariant


def test_residual_pair_is_a_plan_option(lib):
    """mve_unet_set_residual_mode (plan-time only, no GPU): the pair mode -- the UNet's DEFAULT since round 5 -- doubles the residual-stream tensors
    of the workspace, keeps the op list, round-trips, and is refused by the non-UNet executors; ControlNet handles start with the 16-bit stream."""
    import torch
    from mvedit_amd.unet import UNet2DConditionEngine, SD15_CONFIG
    eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16, device='cpu')
    assert eng.residual_pair is True                                # the default: end-to-end error inside north_star's 1e-3
    b = eng.plan(2, 64, 64, 77)
    ups_b = [(lab, fl) for _, _, fl, lab in eng.op_table() if lab.startswith('upsample+conv')]
    assert eng.set_residual_pair(False) is True and eng.residual_pair is False
    a = eng.plan(2, 64, 64, 77)
    ups_a = [(lab, fl) for _, _, fl, lab in eng.op_table() if lab.startswith('upsample+conv')]
    # either mode runs Upsample2D as four 2 x 2 phase convs (4 / 9 of the multiply-adds; the summed-weight rounding fits the pair mode's budget:
    # tests/rounding_budget_experiment.py --phase)
    assert [l for l, _ in ups_a] == ['upsample+conv (4 phases)'] * 3 and [l for l, _ in ups_b] == ['upsample+conv (4 phases)'] * 3
    assert [f for _, f in ups_a] == [f for _, f in ups_b]          # the plan prices an op at the reference's form of it either way
    assert b['n_ops'] == a['n_ops'] and b['flops'] == a['flops']
    assert a['workspace_bytes'] < b['workspace_bytes'] < 2 * a['workspace_bytes']
    assert eng.set_residual_pair(True) is False and eng.plan(2, 64, 64, 77)['workspace_bytes'] == b['workspace_bytes']
    from mvedit_amd.controlnet import ControlNetEngine
    assert ControlNetEngine(SD15_CONFIG, torch.float16, device='cpu').residual_pair is False
    from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG
    vae = AutoencoderKLEngine(dict(SD_VAE_CONFIG), torch.float16, 'cpu')
    with pytest.raises(lib.MveError):
        lib.call('mve_unet_set_residual_mode', vae.decoder._h, 1)


def test_residual_pair_environment_switch():
    """MVE_RESIDUAL_PAIR=0 in the environment creates UNet handles on the 16-bit stream (read by mve_unet_create, i.e. in a fresh process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import torch; from mvedit_amd.unet import UNet2DConditionEngine, SD15_CONFIG; '
            'print(UNet2DConditionEngine(SD15_CONFIG, torch.float16, device="cpu").residual_pair)')
    for val, want in (('0', 'False'), ('1', 'True'), (None, 'True')):
        env = dict(os.environ)
        env.pop('MVE_RESIDUAL_PAIR', None)
        if val is not None:
            env['MVE_RESIDUAL_PAIR'] = val
        out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == want, (val, out.stdout, out.stderr[-500:])


def test_controlnet_shared_conditioning_is_a_plan_option(lib):
    """mve_controlnet_set_cond_repeat (plan-time only, no GPU): with R = 2 the conditioning embedding is planned for half the batch (its conv flops
    halve, conv_in runs once per half), everything else is unchanged; UNet handles refuse the option."""
    import torch
    from mvedit_amd.controlnet import ControlNetEngine
    from mvedit_amd.unet import UNet2DConditionEngine, SD15_CONFIG
    cn = ControlNetEngine(SD15_CONFIG, torch.float16, device='cpu')
    rep = lib.raw('mve_controlnet_set_cond_repeat')
    a = cn.plan(4, 64, 64, 77)
    emb_a = sum(f for _, _, f, lab in cn.op_table() if lab == 'cond_embedding.conv')
    n_in_a = sum(lab == 'conv_in + cond_embedding' for _, _, _, lab in cn.op_table())
    assert rep(cn._h, 2) == 1
    b = cn.plan(4, 64, 64, 77)
    emb_b = sum(f for _, _, f, lab in cn.op_table() if lab == 'cond_embedding.conv')
    n_in_b = sum(lab == 'conv_in + cond_embedding' for _, _, _, lab in cn.op_table())
    assert n_in_a == 1 and n_in_b == 2 and abs(emb_a - 2 * emb_b) <= 1e-9 * emb_a
    assert abs((a['flops']['conv3x3'] - b['flops']['conv3x3']) - emb_b) <= 1e-9 * emb_a and a['flops']['linear'] == b['flops']['linear']
    assert b['workspace_bytes'] <= a['workspace_bytes']
    assert rep(cn._h, 3) == 2
    with pytest.raises(lib.MveError):
        cn.plan(4, 64, 64, 77)                                   # 4 items, 3 shares
    assert rep(cn._h, 1) == 3 and cn.plan(4, 64, 64, 77)['flops'] == a['flops']
    unet = UNet2DConditionEngine(SD15_CONFIG, torch.float16, device='cpu')
    with pytest.raises(lib.MveError):
        lib.call('mve_controlnet_set_cond_repeat', unet._h, 2)
