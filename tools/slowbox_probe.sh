#!/bin/bash
# Runs the bench; when the box is one of the slow ones (> 2.75 ms/step) also records a kernel-trace timeline and the
# step time of the alternative schedules.  GPU box only.
cd "$GRAFT_REPO_ROOT"
ms=$(timeout -k 10 200 python3 bench.py 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "default ms_per_step $ms"
for cfg in "MV3D_SIDE_STREAMS=0" "MV3D_OVERLAP_ADAM=0" "MV3D_SIDE_STREAMS=0 MV3D_OVERLAP_ADAM=0" "MV3D_ADAM_GATE=none"; do
  v=$(env $cfg timeout -k 10 200 python3 bench.py --steps 20 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$cfg ms_per_step $v"
done
slow=$(python3 -c "print(1 if $ms > 2.75 else 0)")
echo "slow=$slow"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r_prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 > "$GRAFT_REPO_ROOT/gpurun_out/r_bench.json" 2>/dev/null
cd "$GRAFT_REPO_ROOT"
python3 tools/timeline.py gpurun_out/r_prof --all > gpurun_out/r_timeline_slow$slow.txt 2>&1
head -4 gpurun_out/r_timeline_slow$slow.txt
rm -rf gpurun_out/r_prof
