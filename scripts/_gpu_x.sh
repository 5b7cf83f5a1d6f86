O=gpurun_out/r03v; mkdir -p $O
C=sketchlib.rust_amd/csrc
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
SKL_LIBRARY=$C/_build/libsketchlib_dist_hip.so timeout 900 python scripts/ab_sweep.py 200,300,500,700,900,1000,1200,1400,1700 coreacc "" "LIB=$C/_build_exp_r02/libsketchlib_dist_hip.so" > $O/ab_final2_vs_r02_coreacc.jsonl 2>&1
SKL_LIBRARY=$C/_build/libsketchlib_dist_hip.so timeout 900 python scripts/ab_sweep.py 500,1000,2000,3000 jaccard "" "LIB=$C/_build_exp_r02/libsketchlib_dist_hip.so" > $O/ab_final2_vs_r02_jaccard.jsonl 2>&1
python - <<'PY'
import json,collections
for f in ('coreacc','jaccard'):
    rows=collections.defaultdict(dict)
    for l in open('gpurun_out/r03v/ab_final2_vs_r02_%s.jsonl'%f):
        if l.startswith('{'):
            d=json.loads(l); rows[d['n']]['r02' if d['variant'] else 'r03']=d['step_ms_median']
    for n in sorted(rows): print(f, n, 'r03 %.4f'%rows[n]['r03'], 'r02 %.4f'%rows[n]['r02'], '%+.1f%%'%((rows[n]['r03']/rows[n]['r02']-1)*100))
PY
