#!/usr/bin/env bash
# On-GPU pass: parity tests, bench, sweep, rocprof kernel trace. Outputs under gpurun_out/<tag>/.
# usage: bash tools/gpu_run.sh <tag> [tests] [bench] [sweep] [sweepquick] [prof]
set -u
TAG=${1:-run}; shift || true
WHAT=" ${*:-tests bench sweep prof} "
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
if [[ "$WHAT" == *" tests "* ]]; then
  echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
fi
if [[ "$WHAT" == *" bench "* ]]; then
  echo "=== bench"; timeout 600 python bench.py --steps 6400 --warmup 640 --sweep 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/bench_n1.json
fi
if [[ "$WHAT" == *" sweep "* ]]; then
  echo "=== sweep"; timeout 1200 python tools/sweep.py --mfma-only > $OUT/sweep.txt 2>&1; grep -v amdgpu.ids $OUT/sweep.txt | tail -120
fi
if [[ "$WHAT" == *" sweepquick "* ]]; then
  echo "=== sweep quick"; timeout 900 python tools/sweep.py --quick --mfma-only > $OUT/sweep.txt 2>&1; grep -v amdgpu.ids $OUT/sweep.txt | tail -80
fi
if [[ "$WHAT" == *" prof "* ]]; then
  echo "=== rocprof"; R=$PWD; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bench -o bench -- python $R/bench.py --steps 640 --warmup 64 --no-cpu-baseline --no-pmc > $R/$OUT/rocprof_bench.log 2>&1
  cd $R; f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-220
fi
if [[ "$WHAT" == *" stream "* ]]; then
  echo "=== stream bench"; timeout 600 python tools/stream_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_bench.txt
fi
if [[ "$WHAT" == *" exhaustive "* ]]; then
  echo "=== exhaustive encoder check"; timeout 1500 python tests/checks/exhaustive_quantize.py 2>&1 | grep -v amdgpu.ids | tee $OUT/exhaustive_quantize.txt
fi
if [[ "$WHAT" == *" configs "* ]]; then
  echo "=== configs bench"; timeout 900 python tools/configs_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/configs_bench.txt
fi
