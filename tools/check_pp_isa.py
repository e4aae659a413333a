"""ISA invariants of the ping-pong GEMM / conv kernel (csrc/gemm_pp.hip) that its hand-counted waits depend on.

The kernel keeps LDS-DMA in flight across barriers with `s_waitcnt vmcnt(N)` counted by hand and owns M0 for the whole K loop.
Both are only correct if hipcc adds nothing of its own to the loop:
  1. M0 is touched only inside the kernel's inline asm (;;#ASMSTART ... ;;#ASMEND): no compiler-generated user between
     pp_m0_take() and pp_m0_give();
  2. inside the K loops (the loops that contain MFMAs) there is no scratch access (a spill reload is a vector memory operation: it
     joins the in-order queue behind the LDS-DMA pieces and the compiler waits for it with vmcnt(0)), no compiler-generated
     `s_waitcnt vmcnt` and no vector memory instruction outside the asm -- except, in the sequential split-K variants (SEQ), between
     the pp_fold_begin / pp_fold_end markers: the slice fold is ordinary code fenced by vmcnt(0) on both sides, and ONLY there may the
     compiler move data (a load hoisted out of the fold would shift the counted waits: it is flagged like anywhere else);
  3. every MFMA of the loop is in place (destination == C operand): the accumulators stay put.
  4. no kernel spills inside a K loop at all (ScratchSize may be non-zero only through the epilogue).

usage: python tools/check_pp_isa.py            (compiles the device code, ~35 s; exit status 1 on a violation)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'mvedit_amd', 'csrc', 'gemm_pp.hip')


def device_asm():
    # the flags of the shipped object (mvedit_amd/build.py): the checked assembly must be the assembly that is linked
    sys.path.insert(0, ROOT)
    from mvedit_amd import build as mve_build
    hipcc = os.environ.get('HIPCC') or '/opt/rocm/bin/hipcc'
    flags = mve_build.COMMON_FLAGS + mve_build.EXTRA_FLAGS.get(os.path.basename(SRC), [])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'gemm_pp.s')
        cmd = [hipcc] + flags + ['-S', '--offload-device-only', SRC, '-o', out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stderr[-2000:])
        return open(out).read().splitlines()


def functions(lines):
    """name -> (list of (line text, inside inline asm?))"""
    out, cur, name, in_asm = {}, None, None, False
    for ln in lines:
        m = re.match(r'^(_ZN\S*k_gemm_pp\S*):', ln)
        if m:
            name, cur, in_asm = m.group(1), [], False
            out[name] = cur
            continue
        if cur is None:
            continue
        if ';;#ASMSTART' in ln:
            in_asm = True
            continue
        if ';;#ASMEND' in ln:
            in_asm = False
            continue
        cur.append((ln, in_asm))
        if ln.strip() == 's_endpgm':
            cur = None
    return out


def k_loops(body):
    """[(start, end)] index ranges of the K loops: from the pp_kloop_begin marker (pp_m0_take, in front of the prologue's pieces) to
    the pp_kloop_end marker (pp_m0_give) of each role's copy.  hipcc may lay a rarely taken block of a loop out behind the end marker
    (the conv's full address decode): blocks with MFMAs are always inside, and rule 1 (M0) is checked over the whole function."""
    loops, start = [], None
    for i, (ln, in_asm) in enumerate(body):
        if in_asm and 'pp_kloop_begin' in ln:
            start = i
        elif in_asm and 'pp_kloop_end' in ln and start is not None:
            loops.append((start, i))
            start = None
    return loops


def check(lines):
    errs = []
    fns = functions(lines)
    if not fns:
        return ['no k_gemm_pp kernels found in the device assembly']
    for name, body in fns.items():
        seq = 'Lb1ELi320' in name                       # template <Tag, MODE, SEQ = true, 320, ...>
        short = name[name.index('k_gemm_pp'):][:60]
        for ln, in_asm in body:
            code = ln.split(';')[0]
            if not in_asm and re.search(r'\bm0\b', code):
                errs.append(f'{short}: compiler-generated use of m0: {ln.strip()}')
        loops = k_loops(body)
        if len(loops) != 2 or any(not any('v_mfma' in body[j][0] for j in range(a, b + 1)) for a, b in loops):
            errs.append(f'{short}: expected two marked K loops (one per role) with MFMAs, found {len(loops)}')
        inside = set(j for a, b in loops for j in range(a, b + 1))
        for j, (ln, _) in enumerate(body):
            if 'v_mfma' in ln and j not in inside:
                errs.append(f'{short}: MFMA outside the marked K loops: {ln.strip()}')
        for a, b in loops:
            in_fold, folds = False, 0
            for ln, in_asm in body[a:b + 1]:
                if in_asm and 'pp_fold_begin' in ln:
                    in_fold, folds = True, folds + 1
                elif in_asm and 'pp_fold_end' in ln:
                    in_fold = False
                code = ln.split(';')[0].strip()
                if not code:
                    continue
                if 'scratch_' in code:
                    errs.append(f'{short}: scratch access inside a K loop: {code}')
                if 'v_mfma' in code:
                    m = re.match(r'v_mfma\S+\s+(\S+),\s*\S+,\s*\S+,\s*(\S+)', code)
                    if not m or m.group(1) != m.group(2):
                        errs.append(f'{short}: MFMA not in place: {code}')
                if in_asm or (seq and in_fold):
                    continue
                if re.match(r's_waitcnt\b.*vmcnt', code):
                    errs.append(f'{short}: compiler-generated vmcnt wait inside a K loop: {code}')
                if re.match(r'(global_|buffer_|flat_)', code):
                    errs.append(f'{short}: compiler-generated vector memory instruction inside a K loop: {code}')
    return errs


def fold_regions(lines):
    """(SEQ kernels, fold regions found in their K loops): the exemption must actually be anchored somewhere"""
    n_seq = n_fold = 0
    for name, body in functions(lines).items():
        if 'Lb1ELi320' not in name:
            continue
        n_seq += 1
        n_fold += sum(1 for ln, in_asm in body if in_asm and 'pp_fold_begin' in ln)
    return n_seq, n_fold


def main():
    lines = device_asm()
    errs = check(lines)
    n = len(functions(lines))
    n_seq, n_fold = fold_regions(lines)
    if n_seq and n_fold < 2 * n_seq:
        errs.append(f'{n_seq} SEQ kernels but only {n_fold} marked fold regions (one per role expected)')
    if errs:
        print('\n'.join(errs[:40]))
        print(f'{len(errs)} violation(s) in {n} kernels')
        return 1
    print(f'ok: {n} k_gemm_pp kernels -- M0 only in inline asm, K loops free of scratch / compiler vmcnt waits / stray vector memory instructions, MFMAs in place')
    return 0


if __name__ == '__main__':
    sys.exit(main())
