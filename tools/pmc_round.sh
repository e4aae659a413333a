#!/bin/bash
# Stall-attribution PMC passes over tools/pmc_probe.py (SQ counters, 8 per pass).  Output: gpurun_out/pmc_probe.txt
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="python $REPO/tools/pmc_probe.py"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $REPO/gpurun_out/pmc_a -o p -- $P > $REPO/gpurun_out/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES -d $REPO/gpurun_out/pmc_b -o p -- $P > $REPO/gpurun_out/pmc_b.log 2>&1
python - <<PY > $REPO/gpurun_out/pmc_probe.txt 2>&1
import glob, sqlite3, re
for tag in ('pmc_a', 'pmc_b'):
    fs = glob.glob('$REPO/gpurun_out/%s/**/*.db' % tag, recursive=True)
    if not fs:
        print(tag, 'no db'); continue
    c = sqlite3.connect(fs[0])
    rows = c.execute('select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name').fetchall()
    tab = {}
    for name, cn, s, n in rows:
        m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', name)
        if not m: continue
        tab.setdefault(m.group(0).replace('(anonymous namespace)::', '')[:48], {})[cn] = s / max(n, 1)
    for k, d in tab.items():
        print(tag, k, {a: round(b) for a, b in sorted(d.items())})
PY
cat $REPO/gpurun_out/pmc_probe.txt | head -40
