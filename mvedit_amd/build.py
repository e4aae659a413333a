"""In-tree build of libmvedit_amd.so (hipcc, gfx950 only).

`python -m mvedit_amd.build` compiles every translation unit under
mvedit_amd/csrc into mvedit_amd/_build/*.o and links mvedit_amd/libmvedit_amd.so.
hipcc cross-compiles without a GPU.  Objects are rebuilt only when the source,
any header in csrc/ or include/, or the flags changed.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, 'csrc')
OBJDIR = os.path.join(ROOT, '_build')
LIB = os.path.join(ROOT, 'libmvedit_amd.so')
ARCH = 'gfx950'

COMMON_FLAGS = [
    f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
    '-Wall', '-Wno-unused-function', '-Wno-unknown-pragmas',
    '-I' + os.path.join(REPO, 'include'),
]
# Per-file extra flags.  The ray marcher must not contract a*b+c into fma: its
# index buffers are compared bit-for-bit with the C oracle.
EXTRA_FLAGS = {
    # softmax maxima of MFMA outputs: without this every fmaxf operand gets a canonicalising v_max_f32 x, x (NaN inputs give NaN outputs either way)
    'attention.hip': ['-fno-honor-nans'],
    'raymarching.hip': ['-ffp-contract=off'],
    'nerf.hip': ['-ffp-contract=off'],
    'raster.hip': ['-ffp-contract=off'],
    'dmtet.hip': ['-ffp-contract=off'],
    'shading.hip': ['-ffp-contract=off'],
    'recon_loss.hip': ['-ffp-contract=off'],
    'mesh_reg.hip': ['-ffp-contract=off'],
    'blur.hip': ['-ffp-contract=off'],
    'sh.hip': ['-ffp-contract=off'],
    'triplane.hip': ['-ffp-contract=off'],
}


# A/B builds: MVE_BUILD_TAG=<tag> writes libmvedit_amd_<tag>.so (objects under _build_<tag>/) with the extra -D switches of MVE_BUILD_DEFS
# (comma separated), e.g. MVE_BUILD_TAG=staged64 MVE_BUILD_DEFS=MVE_EPI_STAGED64 for the rounds 1-3 GEMM epilogue; mvedit_amd._lib loads the
# tagged library when MVE_LIB_TAG=<tag> is set.
_TAG = os.environ.get('MVE_BUILD_TAG', '')
if _TAG:
    OBJDIR = os.path.join(ROOT, '_build_' + _TAG)
    LIB = os.path.join(ROOT, f'libmvedit_amd_{_TAG}.so')
    COMMON_FLAGS = COMMON_FLAGS + ['-D' + d for d in os.environ.get('MVE_BUILD_DEFS', '').split(',') if d]

if os.environ.get('MVE_ATTN_LAB') == '1':        # development build: the timing-only ablation instantiations of k_attention3 (tools/ab_attention_ablate.py)
    EXTRA_FLAGS['attention.hip'] = EXTRA_FLAGS['attention.hip'] + ['-DMVE_ATTN_LAB']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC)')


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers_digest():
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(REPO, 'include')):
        for f in sorted(os.listdir(d)):
            if f.endswith(('.h', '.hpp', '.inc')):
                with open(os.path.join(d, f), 'rb') as fh:
                    h.update(f.encode())
                    h.update(fh.read())
    return h.hexdigest()


_INC_RE = None


def _deps_digest(path, seen=None):
    """sha256 over the quoted #include closure of `path` (csrc/ and include/): a header edit rebuilds only the translation units that see it."""
    global _INC_RE
    import re
    if _INC_RE is None:
        _INC_RE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
    seen = seen if seen is not None else {}
    if path in seen:
        return ''
    seen[path] = True
    with open(path, 'rb') as fh:
        data = fh.read()
    h = hashlib.sha256(data)
    for inc in _INC_RE.findall(data.decode('utf-8', 'replace')):
        for d in (os.path.dirname(path), CSRC, os.path.join(REPO, 'include')):
            cand = os.path.join(d, inc)
            if os.path.exists(cand):
                h.update(_deps_digest(cand, seen).encode())
                break
    return h.hexdigest()


def _compile_one(hipcc, src, hdr_digest, verbose):
    path = os.path.join(CSRC, src)
    hdr_digest = _deps_digest(path)
    obj = os.path.join(OBJDIR, src + '.o')
    stamp = obj + '.stamp'
    flags = COMMON_FLAGS + EXTRA_FLAGS.get(src, [])
    with open(path, 'rb') as fh:
        key = hashlib.sha256(fh.read() + hdr_digest.encode() + json.dumps(flags).encode()).hexdigest()
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj, False
    cmd = [hipcc] + flags + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', path, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    with open(stamp, 'w') as fh:
        fh.write(key)
    return obj, True


def build(verbose=False, force=False):
    """Compile + link; returns the path of libmvedit_amd.so."""
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    hdr = _headers_digest()
    srcs = _sources()
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for obj, changed in ex.map(lambda s: _compile_one(hipcc, s, hdr, verbose), srcs):
            objs.append(obj)
            rebuilt |= changed
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
