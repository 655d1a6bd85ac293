set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r8a; mkdir -p $O
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
echo "=== bench"; timeout 400 python bench.py --steps 100 --warmup 10 --profile-out $O/bench_kernel_stats.csv --span-out $O/bench_kernel_span.json 2>$O/bench.err | tail -1 > $O/bench_n1.json; cut -c1-700 $O/bench_n1.json
echo "=== rocprof sharded"; R=$PWD; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sharded -o sharded -- python $R/bench.py --sharded-path --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/rocprof_sharded.log 2>&1; cd $R
f=$(find $O/prof_sharded -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/sharded_kernel_stats.csv && head -8 $O/sharded_kernel_stats.csv | cut -c1-200; rm -rf $O/prof_sharded
tail -1 $O/rocprof_sharded.log | cut -c1-300
