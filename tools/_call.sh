set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r7a; mkdir -p $O
echo "=== bench sharded-path (world 1)"; timeout 300 python bench.py --sharded-path --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_sharded.err | tail -1 | tee $O/bench_sharded.json | cut -c1-1500; tail -5 $O/bench_sharded.err
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
echo "=== bench"; timeout 400 python bench.py --steps 100 --warmup 10 --profile-out $O/bench_kernel_stats.csv --span-out $O/bench_kernel_span.json 2>$O/bench.err | tail -1 | tee $O/bench_n1.json | cut -c1-2500
echo "=== configs"; timeout 600 python tools/configs_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/configs_bench.txt
