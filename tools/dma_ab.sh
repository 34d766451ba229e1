#!/bin/bash
# A/B of fused_main's LDS-DMA staging (profiling build): BT_FUSED_DMA=0/1 x BT_FUSED_PARTS
R=${GRAFT_REPO_ROOT:-$(pwd)}
for dma in 0 1; do for parts in ${@:-1}; do echo -n "dma $dma parts $parts: "; BT_FUSED_DMA=$dma BT_FUSED_PARTS=$parts python $R/tools/bench_dbg.py --no-cpu-baseline --no-end-to-end --pipeline 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done; done
