"""Which limiter holds the shader clock inside the conv loop?  (1 257 W drawn under a 1 400 W cap -- "package power cap"
is not what those figures show.)

Runs the level-1 conv of the benchmark back to back for ~4 s per phase while a thread samples `rocm-smi --showpower --showclocks --showtemp
--showperflevel --showvoltage`, and reads the SMU's gpu_metrics table (`rocm-smi --showmetrics`, `amd-smi metric`) before and after each phase:
its throttle-residency accumulators (prochot / ppt / socket thermal / vr thermal / hbm thermal `*_residency_acc`) and throttle_status words
name the limiter that was active.  Everything the tools print is kept verbatim under the summary, so that fields this script does not know are
still on file.  Phases: idle, conv loop (MFMA-bound), a bandwidth-bound copy loop (for contrast), conv loop again."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops  # noqa: E402

dt, dev = torch.float16, 'cuda'
B, H, C1, Cout = 64, 32, 640, 640
x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
conv = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=False)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)
copy = lambda: big2.copy_(big)


def sh(cmd, timeout=20):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout).stdout
    except Exception as e:
        return f'<{" ".join(cmd)} failed: {e!r}>'


def metrics():
    return sh(['rocm-smi', '--showmetrics']), sh(['amd-smi', 'metric', '-g', '0'])


def fields(txt):
    """name -> number for every 'name: value' / 'name (unit): value' line with a numeric value"""
    out = {}
    for ln in txt.splitlines():
        m = re.match(r'^\W*(?:GPU\[\d+\]\s*:\s*)?([A-Za-z_][\w \-\(\)/%.]*?)\s*[:=]\s*(-?\d+(?:\.\d+)?)\b', ln.strip())
        if m:
            out[m.group(1).strip().lower()] = float(m.group(2))
    return out


samples, stop = [], False


def sampler():
    while not stop:
        out = sh(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--showperflevel', '--showvoltage'], 5)
        p = re.search(r'Power \(W\):\s*([\d.]+)', out)
        s = re.search(r'sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)', out)
        tj = re.search(r'junction\)\s*\(C\):\s*([\d.]+)', out)
        th = re.search(r'memory\)\s*\(C\):\s*([\d.]+)', out)
        v = re.search(r'Voltage \(mV\):\s*([\d.]+)', out)
        samples.append((time.time(), float(p.group(1)) if p else -1, int(s.group(1)) if s else -1, float(tj.group(1)) if tj else -1,
                        float(th.group(1)) if th else -1, float(v.group(1)) if v else -1, out if len(samples) == 0 else ''))
        time.sleep(0.05)


print(sh(['rocm-smi', '--showmaxpower', '--showperflevel', '--showsclkrange', '--showclkfrq']))
th_ = threading.Thread(target=sampler)
th_.start()
time.sleep(0.5)
report, raw = [], []
for name, fn, flops in (('idle', None, 0), ('conv loop', conv, 2.0 * B * H * H * Cout * 9 * C1), ('copy loop (1 GiB d2d)', copy, 0), ('conv loop again', conv, 2.0 * B * H * H * Cout * 9 * C1)):
    m0 = metrics()
    t0 = time.time()
    n, ms = 0, 0.0
    if fn is None:
        time.sleep(2.0)
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        while time.time() - t0 < 4.0:
            for _ in range(100 if flops else 10):
                fn()
            n += 100 if flops else 10
            torch.cuda.synchronize()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
    t1 = time.time()
    m1 = metrics()
    mine = [s for s in samples if t0 + 0.5 < s[0] < t1]
    avg = lambda k: sum(s[k] for s in mine if s[k] > 0) / max(1, sum(1 for s in mine if s[k] > 0))
    line = (f'{name:24s}: {len(mine):3d} samples  power avg {avg(1):7.1f} W max {max([s[1] for s in mine] + [0]):7.1f}  sclk avg {avg(2):6.0f} MHz  '
            f'Tj {avg(3):5.1f} C  Thbm {avg(4):5.1f} C  V {avg(5):6.0f} mV')
    if flops:
        line += f'  {ms:.4f} ms/launch = {flops / ms / 1e9:.0f} TFLOP/s'
    elif fn is not None:
        line += f'  {ms:.3f} ms/copy = {2 * big.numel() / ms / 1e6:.0f} GB/s'
    report.append(line)
    f0, f1 = fields(m0[0] + '\n' + m0[1]), fields(m1[0] + '\n' + m1[1])
    moved = {k: (f0[k], f1[k]) for k in f1 if k in f0 and f1[k] != f0[k] and any(t in k for t in ('acc', 'throttl', 'residency', 'violation', 'limit', 'prochot', 'ppt'))}
    report.append('    gpu_metrics fields that moved (throttle / residency / limit): ' + (', '.join(f'{k}: {a:g} -> {b:g}' for k, (a, b) in sorted(moved.items())) or 'none found by name'))
    raw.append((name, m1))
    print(line, flush=True)
stop = True
th_.join()
print('\n==== summary ====')
print('\n'.join(report))
print('\n==== first raw rocm-smi sample ====')
print(samples[0][6] if samples else '')
for name, (a, b) in raw:
    print(f'\n==== rocm-smi --showmetrics after "{name}" ====\n{a}\n==== amd-smi metric after "{name}" ====\n{b}')
