"""Dump (kernel, start, end, blocks, lds) rows of a rocprofv3 --kernel-trace rocpd database as CSV (stdout)."""
import glob
import os
import re
import sqlite3
import sys

fs = glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True)
c = sqlite3.connect(fs[0])


def short(name):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', name)
    t = m.group(0) if m else name
    return t.replace('(anonymous namespace)::', '').replace('F16Tag', 'f16')[:70]


for n, s, e, gx, wx, lds in c.execute('select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start'):
    print(f'{short(n)};{s};{e};{gx // max(1, wx)};{lds}')
