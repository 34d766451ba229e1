R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do for parts in 1 2 4 8; do echo -n "parts $parts: "; BT_FUSED_PARTS=$parts python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
