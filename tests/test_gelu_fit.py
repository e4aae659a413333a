"""The GEGLU epilogue's erf-GELU (csrc/gemm_shared.h, gelu_erf): the rational fit's coefficients are read from the header and the same
operation sequence is evaluated in float32 on the CPU against scipy's erf.  Reference function: torch's / diffusers' exact GELU as used by
GEGLU (diffusers FeedForward, called from lib/models/architecture/diffusers.py:69-97 of the reference)."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coefficients():
    src = open(os.path.join(ROOT, 'mvedit_amd', 'csrc', 'gemm_shared.h')).read()
    body = src[src.index('float gelu_erf(float x)'):]
    body = body[:body.index('\n}\n')]
    p = re.search(r'const float P = (.*);', body).group(1)
    q = re.search(r'const float Q = (.*);', body).group(1)
    num = lambda s: [np.float32(v.rstrip('f')) for v in re.findall(r'[-+]?\d+\.?\d*(?:e[-+]?\d+)?f', s)]
    clamp = re.search(r'fmed3f\(x, -(\d+\.\d+)f, (\d+\.\d+)f\)', body)
    assert clamp and clamp.group(1) == clamp.group(2)
    return num(p), num(q), np.float32(clamp.group(1))


def test_gelu_rational_fit_error():
    (p3, p2, p1, p0), (q3, q2, q1, q0), cl = _coefficients()
    assert q0 == np.float32(1.0)
    x = np.linspace(-20, 20, 400001).astype(np.float32)
    t = np.clip(x, -cl, cl)
    u = t * t
    P = ((p3 * u + p2) * u + p1) * u + p0
    Q = ((q3 * u + q2) * u + q1) * u + q0
    assert (Q >= 1.0).all()
    got = (x * (np.float32(0.5) + (t * P) / Q)).astype(np.float64)
    ref = x.astype(np.float64) * 0.5 * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    err = np.abs(got - ref)
    assert err.max() < 1.5e-5, err.max()
    # far below the fp16 rounding of the product it feeds: half an ulp of a value of magnitude 1 is 4.9e-4
    big = np.abs(ref) > 0.05
    assert (err[big] / np.abs(ref[big])).max() < 2e-4
