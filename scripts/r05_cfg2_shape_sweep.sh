for e in "X=1" "SKL_TAIL_SLICES=2" "SKL_TAIL_SLICES=8" "SKL_TAIL_MAX_PCT=0" "SKL_TILE32_MIN=0" "SKL_TILE32_MIN=0 SKL_TAIL_SLICES=2" "SKL_TILE32_MIN=0 SKL_TAIL_MAX_PCT=0" "SKL_GROUP_SPAN=1" "SKL_GROUP_SPAN=4"; do
  env $e python bench.py --secondary none --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['value']/1e9,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), d['roofline']['kernel'][-110:])"
done
