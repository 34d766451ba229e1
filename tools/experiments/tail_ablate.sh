#!/bin/bash
# fused_tail with parts compiled out (profiling build, BT_FUSED_ABLATE bits of the tail; results are NOT valid tiles): what its time is made of
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; }
for a in 0 268435456 33554432 1073741824 536870912 67108864 $((268435456+67108864)) $((268435456+33554432)) $((268435456+33554432+1073741824+67108864)) $((268435456+33554432+1073741824+67108864+2+4+64)); do echo -n "ablate $a: "; BT_FUSED_ABLATE=$a run; done
echo -n "table lookups (BT_FUSED_NO_REGULAR): "; BT_FUSED_NO_REGULAR=1 run
