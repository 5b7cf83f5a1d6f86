O=gpurun_out/r03s; mkdir -p $O
timeout 1200 python scripts/ab_sweep.py 2400,3200,4000,5000,6000,7000,8000 coreacc "" "SKL_SLICED_MAX_PAIRS=0" > $O/ab_sliced_threshold2.jsonl 2>&1
python - <<'PY'
import json,collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/r03s/ab_sliced_threshold2.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); rows[d['n']]['allk' if d['variant'] else 'default(sliced)']=d['step_ms_median']
for n in sorted(rows): print(n, rows[n], '%+.1f%%'%((rows[n]['allk']/rows[n]['default(sliced)']-1)*100))
PY
