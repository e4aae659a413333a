"""Per-kernel durations and inter-kernel gaps of the LAST forward in a rocprofv3 --kernel-trace database (rocpd sqlite).

usage: python tools/trace_gaps.py <dir with *.db> [kernels per forward]
The trace of tools/fwd_small.py holds warm-up + `reps` identical forwards; the last forward is found as the last
repetition of the kernel-name sequence that starts at the last `k_nchw_to_nhwc`-like first kernel."""
import glob
import os
import re
import sqlite3
import sys

root = sys.argv[1]
fs = glob.glob(os.path.join(root, '**', '*.db'), recursive=True)
c = sqlite3.connect(fs[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('kernels')] or [t for t in tabs if 'kernel_dispatch' in t]
rows = c.execute('select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start').fetchall()


def short(name):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', name)
    t = m.group(0) if m else name
    return t.replace('(anonymous namespace)::', '').replace('F16Tag', 'f16')[:70]


names = [short(r[0]) for r in rows]
with open(os.path.join(root, '..', os.path.basename(os.path.normpath(root)) + '_rows.csv'), 'w') as fh:
    for r in rows:
        fh.write(f'{short(r[0])};{r[1]};{r[2]};{r[3] // max(1, r[4])};{r[5]}\n')
first = names[0]
# forwards start at a kernel named like the first kernel of the plan (layout conversion)
starts = [i for i, n in enumerate(names) if 'nchw_to_nhwc' in n or 'to_nhwc' in n]
# keep the starts whose following sequence length equals the modal one
if len(starts) < 2:
    print('could not find forwards', set(names[:5]))
    sys.exit(1)
# conversions also run for ctx etc; use distance between the LAST two candidates with identical name sequence
per = None
for j in range(len(starts) - 2, -1, -1):
    L = starts[-1] - starts[j]
    if L > 50 and names[starts[j]:starts[j] + L] == names[starts[-1] - L:starts[-1]][0:L] and starts[-1] + L <= len(names) and names[starts[-1]:starts[-1] + L] == names[starts[j]:starts[j] + L]:
        per = L
        break
if per is None:
    per = int([a for a in sys.argv[2:] if a.isdigit()][0])
lo = starts[-1]
seg = rows[lo:lo + per]
dur = [(r[2] - r[1]) / 1e3 for r in seg]
gap = [0.0] + [(seg[i][1] - seg[i - 1][2]) / 1e3 for i in range(1, len(seg))]
wall = (seg[-1][2] - seg[0][1]) / 1e3
print(f'kernels in one forward: {per}; wall {wall / 1e3:.3f} ms; sum of kernel durations {sum(dur) / 1e3:.3f} ms; sum of gaps {sum(gap) / 1e3:.3f} ms '
      f'(mean gap {sum(gap) / max(1, per - 1):.2f} us)')
agg = {}
for (n, *_), d, g in zip(seg, dur, gap):
    a = agg.setdefault(short(n), [0, 0.0, 0.0])
    a[0] += 1; a[1] += d; a[2] += g
print(f'{"kernel":72s} {"calls":>5s} {"dur_ms":>8s} {"avg_us":>8s} {"gap_before_us(avg)":>18s}')
for n, (k, d, g) in sorted(agg.items(), key=lambda t: -t[1][1]):
    print(f'{n:72s} {k:5d} {d / 1e3:8.3f} {d / k:8.2f} {g / k:18.2f}')
if '--list' in sys.argv:
    for (n, s, e, gx, wx, lds), d, g in zip(seg, dur, gap):
        print(f'{short(n):72s} dur={d:8.2f} gap={g:7.2f} blocks={gx // max(1, wx):6d} lds={lds}')
