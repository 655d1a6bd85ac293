#!/usr/bin/env bash
# First on-GPU pass: parity tests, bench, sweep, rocprof kernel trace. Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Flags" | cut -c1-400 > gpurun_out/lscpu.txt
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; tail -40 gpurun_out/pytest_gpu.log
echo "=== bench"; timeout 600 python bench.py --steps 1280 --warmup 128 2>&1 | tail -3 | tee gpurun_out/bench_n1.json
echo "=== sweep"; timeout 900 python tools/sweep.py > gpurun_out/sweep.txt 2>&1; tail -80 gpurun_out/sweep.txt
echo "=== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 640 --warmup 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bench -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -15 "$f"
