#!/bin/bash
# A/B/C of library builds inside ONE lease: tools/experiments/ab3.sh rounds lib1.so lib2.so ... (clean 16k job + the masked one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for i in $(seq 1 $rounds); do for lib in "$@"; do echo -n "$(basename $lib) clean: "; BT_LIB=$R/$lib python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
for lib in "$@"; do echo -n "$(basename $lib) masked: "; BT_LIB=$R/$lib python $R/tools/config_bench_dbg.py --masked16k 2>/dev/null | tail -1; done
