#!/bin/bash
# usage: tools/sweep_env.sh "VAR1=a VAR2=b" "VAR1=c" ...   -- one bench line (value, ms/step, median, min) per environment
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > /tmp/sweep.json 2>/tmp/sweep.err || { echo "$cfg FAILED"; tail -3 /tmp/sweep.err; continue; }
  python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open('/tmp/sweep.json').read().strip().splitlines()[-1])
print("%-60s %8.0f img/s  %.4f ms  median %.4f  min %.4f" % (sys.argv[1], d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min']))
PY
done
