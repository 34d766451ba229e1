#!/bin/bash
# gpurun helper: GPU test suite + one bench line, results under gpurun_out/ (tools/gpu_check.sh [pytest args])
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/tests.log | tail -3
python bench.py --no-cpu-baseline --no-end-to-end 2> gpurun_out/bench.err > gpurun_out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print(round(d["ms_per_step"], 4), "ms/step", round(d["value"]), "tiles/s", [(l["kind"], round(l["avg_ms"] * 1e3, 1)) for l in d["config"]["launches"]],
      "frac", round(d["roofline"]["frac"], 3))
PY
