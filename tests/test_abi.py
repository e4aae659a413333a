"""The C-ABI library builds, loads, and exports every symbol include/mvedit_amd.h declares (no compute)."""
import ctypes
import subprocess


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
