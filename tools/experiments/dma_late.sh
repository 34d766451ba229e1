R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do for a in 0 524288; do echo -n "ablate $a clean: "; BT_FUSED_ABLATE=$a python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; echo -n "ablate $a masked: "; BT_FUSED_ABLATE=$a python $R/tools/config_bench_dbg.py --masked16k 2>/dev/null | tail -1; done; done
