#!/bin/bash
# rocprofv3 passes over the render / reconstruct / VAE kernels (bench.py --secondary-only): kernel trace, HBM-side bytes, occupancy.
# Output: gpurun_out/prof_render_summary.txt (copy to profiles/rNN_rocprof_render.txt)
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --secondary-only"
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_render_stats -o r -- $B > $REPO/gpurun_out/prof_render_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_render_fetch -o r -- $B > $REPO/gpurun_out/prof_render_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_render_write -o r -- $B > $REPO/gpurun_out/prof_render_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $REPO/gpurun_out/prof_render_occ -o r -- $B > $REPO/gpurun_out/prof_render_occ.log 2>&1
cd $REPO
python - <<PY > gpurun_out/prof_render_summary.txt 2>&1
import glob, sqlite3, re
KEEP = re.compile(r'k_(nerf|render|march|composite|point_decode|decode|raster|resolve|interp|texture|mip|visib|bake|splat|view_weight|antialias|edge|dmtet|scan|box_|xty|adam|hash|grid|pack|near)')
def db(tag):
    fs = glob.glob('gpurun_out/%s/**/*.db' % tag, recursive=True)
    return sqlite3.connect(fs[0]) if fs else None
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', n)
    return (m.group(0) if m else n)[:56]
c = db('prof_render_stats')
dur = {}
if c:
    print('== rocprofv3 --kernel-trace --stats -- python bench.py --secondary-only : render / reconstruct kernels (all launches of the run)')
    for name, calls, tot, avg, pct in c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall():
        if KEEP.search(name):
            dur[short(name)] = avg
            print(f'{short(name):58s} calls={calls:5d} total_ms={tot / 1e3:9.3f} avg_us={avg:10.2f}')
for tag, title in (('prof_render_fetch', 'FETCH_SIZE (KiB; x2 on gfx950 for wide streaming reads)'), ('prof_render_write', 'WRITE_SIZE (KiB)'), ('prof_render_occ', 'occupancy / stall counters')):
    c = db(tag)
    if not c: continue
    print('== rocprofv3 --pmc :', title, '-- mean per dispatch')
    rows = c.execute('select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name').fetchall()
    tab = {}
    for name, cn, s, n in rows:
        if KEEP.search(name): tab.setdefault(short(name), {})[cn] = s / max(n, 1)
    for k, d in sorted(tab.items()):
        extra = ''
        if 'FETCH_SIZE' in d and k in dur: extra = f'  -> {2 * d["FETCH_SIZE"] * 1024 / (dur[k] * 1e-6) / 1e9:8.1f} GB/s fetched (x2)'
        if 'WRITE_SIZE' in d and k in dur: extra = f'  -> {d["WRITE_SIZE"] * 1024 / (dur[k] * 1e-6) / 1e9:8.1f} GB/s written'
        print(f'{k:58s}', {a: round(b, 1) for a, b in sorted(d.items())}, extra)
PY
head -120 gpurun_out/prof_render_summary.txt
find gpurun_out -name "*.db" -size +30M -delete
