"""ctypes binding of libmvedit_amd.so.

The prototypes are parsed from include/mvedit_amd.h at import time so the
header is the single source of truth for the C ABI.  There is NO fallback: if
the shared library is missing (or a symbol the header declares is not
exported) importing this module raises.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must be imported first: pins ONE libamdhip64 in the process)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'mvedit_amd.h')
LIB_PATH = os.path.join(_HERE, 'libmvedit_amd.so' if not os.environ.get('MVE_LIB_TAG') else f"libmvedit_amd_{os.environ['MVE_LIB_TAG']}.so")      # (tagged: an A/B build of mvedit_amd.build)

_CTYPE = {
    'void': None,
    'int': ctypes.c_int,
    'float': ctypes.c_float,
    'double': ctypes.c_double,
    'size_t': ctypes.c_size_t,
    'uint32_t': ctypes.c_uint32,
    'int32_t': ctypes.c_int32,
    'uint64_t': ctypes.c_uint64,
    'int64_t': ctypes.c_int64,
    'uint8_t': ctypes.c_uint8,
    'char': ctypes.c_char,
}


class MveError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every MVE_API prototype."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'MVE_API\s+([\w\s\*]+?)\s*\b(mve_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'(.*?)(\w+)$', a)
                arglist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


def _to_ctype(tstr):
    t = tstr.replace('const', ' ').strip()
    stars = t.count('*')
    base = t.replace('*', ' ').split()
    base = base[-1] if base else 'void'
    if stars:
        if base == 'char' and stars == 1:
            return ctypes.c_char_p
        return ctypes.c_void_p  # all data pointers are raw addresses (tensor.data_ptr())
    if base in _CTYPE:
        return _CTYPE[base]
    return ctypes.c_int  # enums


PROTOS = parse_header()

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: build it with `python -m mvedit_amd.build` '
        '(mvedit_amd has no CPU or PyTorch fallback for its HIP kernels)')

_dll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)

for _name, (_ret, _args) in PROTOS.items():
    try:
        _fn = getattr(_dll, _name)
    except AttributeError as e:  # header/ABI drift is a hard error
        raise ImportError(f'libmvedit_amd.so does not export {_name} declared in mvedit_amd.h') from e
    _fn.restype = _to_ctype(_ret)
    _fn.argtypes = [_to_ctype(t) for t, _ in _args]

_dll.mve_last_error.restype = ctypes.c_char_p


def last_error():
    return _dll.mve_last_error().decode()


def call(name, *args):
    """Call an int-returning entry point; raise MveError on a negative status."""
    rc = getattr(_dll, name)(*args)
    if rc < 0:
        raise MveError(f'{name} failed ({rc}): {last_error()}')
    return rc


def raw(name):
    return getattr(_dll, name)


def ptr(t):
    """Device (or host) address of a tensor, or None."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` as a void*."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
