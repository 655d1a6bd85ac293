set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r8c; mkdir -p $O
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
