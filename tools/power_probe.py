"""Is the conv / GEMM main loop power-limited?  Runs one conv shape for ~3 s per variant (old / new LDS swizzle: same work, 13 % fewer
cycles per K step with the new one) while sampling `rocm-smi` (average socket power, shader clock) in a background thread."""
import os, subprocess, sys, threading, time, re
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib
tune = _lib.raw('mve_gemm_tune')
dt, dev = torch.float16, 'cuda'
B, H, C1, Cout = 64, 32, 640, 640
x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=True)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp'], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r'Power \(W\):\s*([\d.]+)', out); s = re.search(r'sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)', out)
            samples.append((time.time(), float(p.group(1)) if p else -1, int(s.group(1)) if s else -1, out if len(samples) == 0 else ''))
        except Exception as e:
            samples.append((time.time(), -1, -1, str(e)))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
res = []
for name, mode in (('idle', None), ('old swizzle', 1 | (1 << 25)), ('new swizzle', 1), ('old swizzle', 1 | (1 << 25)), ('new swizzle', 1)):
    t0 = time.time()
    n = 0
    if mode is None:
        time.sleep(1.0)
    else:
        tune(mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        while time.time() - t0 < 3.0:
            for _ in range(200): f()
            n += 200
            torch.cuda.synchronize()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
    t1 = time.time()
    mine = [s for s in samples if t0 + 0.5 < s[0] < t1]
    pw = [s[1] for s in mine if s[1] > 0]; ck = [s[2] for s in mine if s[2] > 0]
    line = f'{name:12s}: {len(mine)} samples, power avg {sum(pw) / max(len(pw), 1):7.1f} W, sclk avg {sum(ck) / max(len(ck), 1):6.0f} MHz'
    if mode is not None:
        line += f', {ms:.4f} ms per launch = {2 * B * H * H * Cout * 9 * C1 / ms / 1e9:.0f} TF'
    print(line, flush=True)
stop = True; th.join()
print('first rocm-smi output:\n' + samples[0][3][:1500])
