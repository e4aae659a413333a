#!/bin/bash
# SQ counter passes over the level-0 self-attention shape (B=16, L=4096, h=8, d=40), one mve_attention_tune variant per run.
#   bash tools/pmc_attention.sh 0 8 11     -> gpurun_out/pmc_attention.txt
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  P="python $REPO/tools/pmc_attention.py $v"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $REPO/gpurun_out/pmca_a$v -o p -- $P > $REPO/gpurun_out/pmca_a$v.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_WAVES -d $REPO/gpurun_out/pmca_b$v -o p -- $P > $REPO/gpurun_out/pmca_b$v.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $REPO/gpurun_out/pmca_c$v -o p -- $P > $REPO/gpurun_out/pmca_c$v.log 2>&1
done
python - "$@" <<PY > $REPO/gpurun_out/pmc_attention.txt 2>&1
import glob, sqlite3, re, sys
for v in sys.argv[1:]:
    for tag in ('pmca_a', 'pmca_b', 'pmca_c'):
        fs = glob.glob('$REPO/gpurun_out/%s%s/**/*.db' % (tag, v), recursive=True)
        if not fs:
            print(tag, v, 'no db'); continue
        c = sqlite3.connect(fs[0])
        rows = c.execute('select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name').fetchall()
        tab = {}
        for name, cn, s, n in rows:
            m = re.search(r'(k_attention[0-9]*)', name)
            if not m: continue
            tab.setdefault(m.group(0), {})[cn] = s / max(n, 1)
        for k, d in tab.items():
            print('variant', v, tag, k, {a: round(b) for a, b in sorted(d.items())})
PY
cat $REPO/gpurun_out/pmc_attention.txt
