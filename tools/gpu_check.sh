#!/bin/bash
# gpurun helper: GPU test suite + one bench line + the non-headline configs, results under gpurun_out/<tag>/
#   tools/gpu_check.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-check}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x "$@" > $O/tests.log 2>&1
grep -E "passed|failed|error" $O/tests.log | tail -3
python bench.py --no-cpu-baseline --no-end-to-end 2> $O/bench.err > $O/bench.json
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print(round(d["ms_per_step"], 4), "ms/step", round(d["value"]), "tiles/s", [(l["kind"], round(l["avg_ms"] * 1e3, 1)) for l in d["config"]["launches"]],
      "frac", round(d["roofline"]["frac"], 3))
PY
python tools/config_bench.py --masked16k > $O/masked16k.json 2> $O/masked16k.err; cat $O/masked16k.json
python tools/config_bench.py > $O/config_bench.json 2> $O/config_bench.err; cat $O/config_bench.json
