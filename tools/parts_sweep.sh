for p in "" 4 5 10; do echo -n "parts=${p:-auto}: "; BT_FUSED_PARTS=$p python bench.py --no-cpu-baseline --no-end-to-end | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [(l['kind'], round(l['avg_ms']*1e3,1)) for l in d['config']['launches']])"; done
