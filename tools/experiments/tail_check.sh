#!/bin/bash
# quick look at the small launches: bench (one stream) twice + the non-headline configs
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do python $R/bench.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], round(d['roofline']['frac'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done
python $R/tools/config_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, v)" | cut -c1-300
