#!/bin/bash
# fused_main variants of the profiling build in ONE lease: tools/experiments/mode_ab.sh "<env assignments>" "<env assignments>" ... (ROUNDS=n, STEPS=n)
# e.g. tools/experiments/mode_ab.sh "BT_FUSED_MODE=0" "BT_FUSED_MODE=2" "BT_FUSED_MODE=2 BT_FUSED_ABLATE=70"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in $(seq 1 ${ROUNDS:-2}); do for v in "$@"; do echo -n "$v: "; env $v python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps ${STEPS:-60} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
