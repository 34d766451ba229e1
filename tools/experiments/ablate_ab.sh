#!/bin/bash
# fused_main under BT_FUSED_ABLATE masks (tools/bench_dbg.py, the debug build), all in ONE lease: tools/experiments/ablate_ab.sh 0 70 71 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do for mask in "$@"; do echo -n "ablate $mask: "; BT_FUSED_ABLATE=$mask python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
