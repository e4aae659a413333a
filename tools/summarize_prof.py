"""Aggregate rocprofv3 (rocpd sqlite) outputs under <dir>/prof_*/ into per-kernel text tables (stdout).

usage: python tools/summarize_prof.py gpurun_out > profiles/rNN_rocprof_summary.txt
"""
import glob
import os
import re
import sqlite3
import sys

root = sys.argv[1]


def short(name):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', name)
    if m:
        t = m.group(0)
        t = t.replace('(anonymous namespace)::', '').replace('F16Tag', 'f16').replace('BF16Tag', 'bf16')
        return t[:64]
    return name[:64]


def db(tag):
    fs = glob.glob(os.path.join(root, tag, '**', '*.db'), recursive=True)
    return sqlite3.connect(fs[0]) if fs else None


# extra arguments: kernel-trace directories to summarise instead of prof_stats (python tools/summarize_prof.py gpurun_out prof_pending ...)
for tag in (sys.argv[2:] or ['prof_stats']):
    c = db(tag)
    if c:
        print(f'== rocprofv3 --kernel-trace --stats ({tag}) : per-kernel totals (all launches of the run)')
        rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
        for name, calls, tot, avg, pct in rows[:(30 if tag == 'prof_stats' else 60)]:
            print(f'{short(name):66s} calls={calls:6d} total_ms={tot / 1e3:10.3f} avg_us={avg:10.2f} pct={pct:6.2f}')
if len(sys.argv) > 2:
    sys.exit(0)

for tag in ('prof_fetch', 'prof_write', 'prof_mfma'):
    c = db(tag)
    if not c:
        continue
    print(f'== rocprofv3 --pmc ({tag}) : per-kernel counter sums over the run (1 step; rows = dispatches x counter instances; FETCH/WRITE_SIZE in KiB)')
    rows = c.execute('select name, counter_name, sum(counter_value), count(*), sum(duration) from pmc_events '
                     'group by name, counter_name order by sum(counter_value) desc').fetchall()
    for name, cn, s, n, dur in rows[:60]:
        print(f'{short(name):66s} {cn:28s} sum={s:14.6g} dispatches={n:5d} mean={s / n:12.6g} rows_dur_ms={(dur or 0) / 1e6:9.3f}')
