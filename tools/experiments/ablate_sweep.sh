#!/bin/bash
# timing experiments on fused_main with the profiling build: BT_FUSED_ABLATE bit sets (results are NOT valid tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for a in ${@:-0 1 2 8 10 11 16 32}; do echo -n "ablate $a: "; BT_FUSED_ABLATE=$a python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --pipeline 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done
