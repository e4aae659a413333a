"""ISA invariants of hand-scheduled kernels, checked on the compiler's output (CPU-only: hipcc cross-compiles gfx950)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')), reason='hipcc not available')
def test_pingpong_gemm_loop_has_no_compiler_generated_waits_or_m0_users():
    """csrc/gemm_pp.hip counts vmcnt by hand and owns M0 across its K loop: a spill reload, a compiler-inserted vmcnt wait, a vector
    memory instruction or an M0 user that hipcc adds to the loop (a new compiler version, a register-pressure change) would silently
    turn the schedule into drain-every-step or corrupt the LDS-DMA destinations.  tools/check_pp_isa.py reads the device assembly."""
    spec = importlib.util.spec_from_file_location('check_pp_isa', os.path.join(ROOT, 'tools', 'check_pp_isa.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = mod.device_asm()
    assert len(mod.functions(lines)) >= 12          # f16 / bf16 x GEMM / conv x {320, 320 sequential split-K, 256, 128}
    errs = mod.check(lines)
    assert not errs, '\n'.join(errs[:20])
    # the checker itself must notice an intruder
    fns = mod.functions(lines)
    body = next(iter(fns.values()))
    a, _ = mod.k_loops(body)[0]
    marker = next(ln for ln, in_asm in body[a:] if 'v_mfma' in ln)
    bad = list(lines)
    at = bad.index(marker)
    while ';;#ASMSTART' not in bad[at]:             # the MFMAs are inline asm: step out of the block
        at -= 1
    bad.insert(at, '\ts_waitcnt vmcnt(0)')
    bad.insert(at, '\ts_mov_b32 m0, s5')
    found = mod.check(bad)
    assert any('vmcnt' in e for e in found) and any('m0' in e for e in found)
    # sequential split-K kernels: compiler memory traffic is legal ONLY inside the marked fold; a load hoisted out of it must be flagged
    seq_name, seq_body = next((n, b) for n, b in fns.items() if 'Lb1ELi320' in n)
    n_seq, n_fold = mod.fold_regions(lines)
    assert n_seq >= 4 and n_fold >= 2 * n_seq
    a, _ = mod.k_loops(seq_body)[0]
    marker = next(ln for ln, in_asm in seq_body[a:] if 'v_mfma' in ln)
    start = next(i for i, ln in enumerate(lines) if ln.startswith(seq_name + ':'))
    bad = list(lines)
    at = bad.index(marker, start)
    while ';;#ASMSTART' not in bad[at]:
        at -= 1
    bad.insert(at, '\tglobal_load_dwordx4 v[0:3], v[4:5], off')
    assert any('vector memory instruction' in e for e in mod.check(bad))
