#!/bin/bash
# work stealing in fused_main on / off (profiling build), alternating inside ONE lease
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for steal in ${STEALS:-0 1}; do echo -n "steal $steal: "; BT_FUSED_STEAL=$steal python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
