"""ORACLE -- test infrastructure only: numpy front-end of raster_oracle.c (rasterise / interpolate with the output
convention of nvdiffrast as used by lib/models/decoders/mesh_renderer/base_mesh_renderer.py:240-252)."""
import ctypes

import numpy as np

from . import build

_lib = ctypes.CDLL(build())
_p, _i = ctypes.c_void_p, ctypes.c_int
_lib.orc_rasterize.argtypes = [_p, _i, _i, _p, _i, _i, _i, _p]
_lib.orc_interpolate.argtypes = [_p, _i, _i, _i, _p, _i, _i, _p, _i, _p]
_lib.orc_edge_opposites.argtypes = [_p, _i, _p]
_lib.orc_antialias.argtypes = [_p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p]


def _ptr(a):
    return a.ctypes.data_as(_p)


def rasterize(pos, tri, resolution):
    h, w = resolution
    pos = np.ascontiguousarray(pos, np.float32)
    tri = np.ascontiguousarray(tri, np.int32)
    B, V, _ = pos.shape
    rast = np.zeros((B, h, w, 4), np.float32)
    _lib.orc_rasterize(_ptr(pos), B, V, _ptr(tri), tri.shape[0], h, w, _ptr(rast))
    return rast


def interpolate(attr, rast, tri):
    attr = np.ascontiguousarray(attr, np.float32)
    tri = np.ascontiguousarray(tri, np.int32)
    rast = np.ascontiguousarray(rast, np.float32)
    B, h, w, _ = rast.shape
    out = np.zeros((B, h, w, attr.shape[-1]), np.float32)
    _lib.orc_interpolate(_ptr(attr), attr.shape[0], attr.shape[1], attr.shape[2], _ptr(rast), B, h * w, _ptr(tri), tri.shape[0], _ptr(out))
    return out


def edge_opposites(tri):
    tri = np.ascontiguousarray(tri, np.int32)
    opp = np.empty_like(tri)
    _lib.orc_edge_opposites(_ptr(tri), tri.shape[0], _ptr(opp))
    return opp


def antialias(color, rast, pos, tri, opp=None):
    """dr.antialias(color, rast, pos, tri): color [B,h,w,C], pos [B,V,4] clip space."""
    color = np.ascontiguousarray(color, np.float32)
    rast = np.ascontiguousarray(rast, np.float32)
    pos = np.ascontiguousarray(pos, np.float32)
    tri = np.ascontiguousarray(tri, np.int32)
    opp = edge_opposites(tri) if opp is None else np.ascontiguousarray(opp, np.int32)
    B, h, w, C = color.shape
    out = np.empty_like(color)
    _lib.orc_antialias(_ptr(color), B, h, w, C, _ptr(rast), _ptr(pos), pos.shape[1], _ptr(tri), tri.shape[0], _ptr(opp), _ptr(out))
    return out
