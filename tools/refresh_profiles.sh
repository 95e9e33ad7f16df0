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
# SQ counters of the pipelined convolution kernels on the layer that carries the step (e0_0 / d1_0: 64 x 64 x 64 x 32 -> 32, 5x5)
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq1" -- python3 "$GRAFT_REPO_ROOT/tools/cconv_stamps.py" > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d "$out/pmc_sq2" -- python3 "$GRAFT_REPO_ROOT/tools/cconv_stamps.py" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
{ echo "# rocprofv3 --pmc (two passes), tools/cconv_stamps.py: conv2d forward 64x64x64x32 -> 32, 5x5 (e0_0 / d1_0 at batch 64), mean per launch";
  python3 tools/pmc_summary.py "$out/pmc_sq1" cconv; python3 tools/pmc_summary.py "$out/pmc_sq2" cconv; } > "$out/pmc_cconv_e0_0.txt" 2>&1
MV3D_DBG=32 timeout -k 10 100 python3 tools/cconv_stamps.py > "$out/stamps_cconv_e0_0.txt" 2>&1
MV3D_DBG=32 timeout -k 10 100 python3 tools/cconv_stamps.py --dgrad >> "$out/stamps_cconv_e0_0.txt" 2>&1
rm -rf "$out/pmc_sq1" "$out/pmc_sq2"
# the same counters for the pipelined filter gradient of that layer (cwgrad_kernel, 128 CUs); its time from an unprofiled run
cd "$GRAFT_REPO_ROOT"
timeout -k 10 100 python3 tools/cconv_stamps.py --wgrad > "$out/cwgrad_time.txt" 2>&1
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq3" -- python3 "$GRAFT_REPO_ROOT/tools/cconv_stamps.py" --wgrad > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d "$out/pmc_sq4" -- python3 "$GRAFT_REPO_ROOT/tools/cconv_stamps.py" --wgrad > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
{ echo "# rocprofv3 --pmc (two passes), tools/cconv_stamps.py --wgrad: conv2d filter gradient 64x64x64x32 -> 32, 5x5 (e0_0 / d1_0 at batch 64), mean per launch";
  grep "us per call" "$out/cwgrad_time.txt"; python3 tools/pmc_summary.py "$out/pmc_sq3" cwgrad; python3 tools/pmc_summary.py "$out/pmc_sq4" cwgrad; } > "$out/pmc_cwgrad_e0_0.txt" 2>&1
rm -rf "$out/pmc_sq3" "$out/pmc_sq4" "$out/cwgrad_time.txt"
ls -la "$out"
