"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (rocpd databases under
gpurun_out/prof_fetch, gpurun_out/prof_write): dispatch-weighted KiB means per launch of the dominant kernels of each op class.

usage: python tools/make_pmc_traffic.py gpurun_out profiles/rNN_rocprof_vK_summary.txt > profiles/pmc_traffic.json
"""
import glob
import json
import os
import re
import sqlite3
import sys

root, source = sys.argv[1], sys.argv[2]
CLASSES = {      # op class -> regex on the kernel name (template arguments as rocprofv3 prints them)
    'conv3x3': r'k_gemm_(pp|big)<\w+, 1,',
    'linear': r'k_gemm_(pp|big)<\w+, 0,',
    'attention': r'k_attention3<',
}


def per_kernel(tag, counter):
    fs = glob.glob(os.path.join(root, tag, '**', '*.db'), recursive=True)
    if not fs:
        return {}
    c = sqlite3.connect(fs[0])
    rows = c.execute('select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name = ? group by name',
                     (counter,)).fetchall()
    return {name: (s, n) for name, s, n in rows}


fetch, write = per_kernel('prof_fetch', 'FETCH_SIZE'), per_kernel('prof_write', 'WRITE_SIZE')
out = {
    '_comment': 'Per-launch HBM-side traffic of the dominant kernels from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py '
                'workload (32 views, 1 step): dispatch-weighted KiB means over the kernel instances of each class. bench.py applies the gfx950 '
                'FETCH_SIZE x2 correction of MI355X_MICROARCH.md (wide coalesced 16 B/lane loads). FETCH_SIZE counts L2 misses including '
                'Infinity-Cache hits, so this is an upper bound on true HBM bytes.',
    'source': source,
}
for cls, pat in CLASSES.items():
    fs = [(s, n) for name, (s, n) in fetch.items() if re.search(pat, name)]
    ws = [(s, n) for name, (s, n) in write.items() if re.search(pat, name)]
    if not fs or not ws:
        continue
    nd = sum(n for _, n in fs)
    out[cls] = {'kernel': pat, 'fetch_kib_mean': round(sum(s for s, _ in fs) / nd), 'write_kib_mean': round(sum(s for s, _ in ws) / sum(n for _, n in ws)),
                'dispatches': nd}
print(json.dumps(out, indent=2))
