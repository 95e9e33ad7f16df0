#!/bin/bash
# Regenerates the per-round profile set on the GPU box: bench line + hip-event kernel table, rocprofv3 kernel stats,
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), one-step stream timeline.  Outputs under gpurun_out/prof_<tag>/.
tag=${1:-r01_d}
out="$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --dump-kernels > "$out/bench.json" 2> "$out/kernels_hipevents.txt" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$out/trace_bench.json" 2>/dev/null || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
python3 tools/timeline.py "$out/trace" --all > "$out/timeline.txt" 2>&1
python3 tools/make_traffic.py "$out/pmc_fetch" "$out/pmc_write" > "$out/traffic.json" 2> "$out/traffic.err"
cp $(find "$out/trace" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv"
cp $(find "$out/pmc_fetch" -name "*counter_collection.csv" | head -1) "$out/pmc_fetch_size.csv"
cp $(find "$out/pmc_write" -name "*counter_collection.csv" | head -1) "$out/pmc_write_size.csv"
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
ls -la "$out"; head -4 "$out/timeline.txt"; cut -c1-260 "$out/bench.json"
