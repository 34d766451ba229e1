#!/bin/bash
# fused_direct (config 2 albedo) under the profiling build's switches: BT_FUSED_ABLATE bit sets (1 no pyramid, 2 no finest stores,
# 4 no apron rows, 8 no source loads; results are NOT valid tiles) and BT_FUSED_PARTS (row blocks per workgroup)
R=${GRAFT_REPO_ROOT:-$(pwd)}
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])["config2_albedo_4k"]; print(round(d["ms"]*1e3,1), d["launches"])'
for a in ${@:-0 1 2 4 8 3 7 10 15}; do echo -n "ablate $a: "; BT_FUSED_ABLATE=$a python $R/tools/config_bench_dbg.py --config2 2>/dev/null | python -c "$pick"; done
for g in 1 2 4 8 16; do echo -n "row blocks per workgroup $g: "; BT_FUSED_PARTS=$g python $R/tools/config_bench_dbg.py --config2 2>/dev/null | python -c "$pick"; done
