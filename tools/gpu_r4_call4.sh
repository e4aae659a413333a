#!/bin/bash
# Round 4, call 4: the default bench command with its extra workloads / outer step; the chain-vs-sliced and norm pair tests again.
mkdir -p gpurun_out
timeout 1500 python bench.py 2>gpurun_out/r04_bench_v1.err | tee gpurun_out/r04_bench_v1.log | tail -1 | cut -c1-600
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04_bench_v1.log') if x.startswith('{')]
d=json.loads(l[-1])
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['per_class_ms'], 'parity', d.get('parity'))
for e in d.get('extra_workloads', []): print('EXTRA', json.dumps(e)[:600])
print('OUTER', json.dumps(d.get('outer_step'))[:1500])
print('SECONDARY keys', list(d.get('secondary', {}).keys()) if isinstance(d.get('secondary'), dict) else d.get('secondary'))
PY
tail -5 gpurun_out/r04_bench_v1.err
timeout 600 python -m pytest tests/test_unet_ops.py -x -q -m gpu -k "chain or norms_read" -s 2>&1 | grep -v "^$" | tail -12
