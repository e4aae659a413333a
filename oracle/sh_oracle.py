"""ORACLE (test infrastructure only -- never imported by the product path).

Real spherical harmonics in the reference's convention (lib/ops/shencoder/src/shencoder.cu:28-337: index l^2 + l + m, Condon-Shortley
phase, the polynomial form that equals Y_lm on the unit sphere), built INDEPENDENTLY of csrc/sh_core.h's recurrences: float64, numpy's
Legendre-series derivative for d^m P_l / dz^m and complex powers for (x + i y)^m.  The Jacobian is the derivative of those polynomials.
Pinning: against scipy.special's complex harmonics on the unit sphere (tests/test_shencoder.py, CPU) and against the reference's own
kernel rebuilt for gfx950 (oracle/_ref/_shencoder_ref*.so via oracle/build_ref.py; GPU test)."""
import math

import numpy as np
from numpy.polynomial import legendre as Lg


def sh_encode(xyz, degree, jacobian=False):
    """xyz [B, 3] -> out [B, degree^2] (and jac [B, 3, degree^2]) in float64"""
    p = np.asarray(xyz, np.float64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    C2 = degree * degree
    out = np.zeros((p.shape[0], C2))
    jac = np.zeros((p.shape[0], 3, C2))
    for m in range(degree):
        w = (x + 1j * y) ** m
        dw = m * (x + 1j * y) ** (m - 1) if m else np.zeros_like(w)          # d w / d x;  d w / d y = i d w / d x
        for l in range(m, degree):
            c = np.zeros(l + 1)
            c[l] = 1.0
            d = Lg.legder(c, m) if m else c
            q = Lg.legval(z, d) * (-1) ** m
            dq = Lg.legval(z, Lg.legder(d, 1)) * (-1) ** m if len(d) > 1 else np.zeros_like(z)
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m)) * (math.sqrt(2) if m else 1.0)
            ip, im = l * l + l + m, l * l + l - m
            out[:, ip] = K * w.real * q
            jac[:, 0, ip], jac[:, 1, ip], jac[:, 2, ip] = K * dw.real * q, K * (1j * dw).real * q, K * w.real * dq
            if m:
                out[:, im] = K * w.imag * q
                jac[:, 0, im], jac[:, 1, im], jac[:, 2, im] = K * dw.imag * q, K * (1j * dw).imag * q, K * w.imag * dq
    return (out, jac) if jacobian else out
