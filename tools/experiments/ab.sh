#!/bin/bash
# A/B of two builds of the library inside ONE lease (leases differ by up to 10 %): tools/experiments/ab.sh libA.so libB.so [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${3:-3}); do for lib in "$1" "$2"; do echo -n "$(basename $lib): "; BT_LIB=$R/$lib python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
