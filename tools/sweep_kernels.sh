#!/bin/bash
# usage: tools/sweep_kernels.sh <label-regex> "VAR=a" "VAR=b" ...  -- own durations (ms/step) of the matching kernel labels per environment
cd "$GRAFT_REPO_ROOT"
pat="$1"; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dump-kernels 2>&1 >/dev/null | grep -E "$pat"
done
