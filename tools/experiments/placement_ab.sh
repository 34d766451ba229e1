#!/bin/bash
# does the physical placement of the allocations change fused_main's time inside ONE lease?  (device memory taken up front by pads)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for pad in ${PADS:-"" 64 512 1,1 4096 100,3 20000 64}; do echo -n "pad [$pad] MB: "; BT_PAD_MB=$pad BT_LIB=$R/bevy_terrain_amd/libbevy_terrain_amd.so python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done
