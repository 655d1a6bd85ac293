set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1 BNB_MI355X_PEER_WAIT_POLLS=1000000
O=gpurun_out/r7g; mkdir -p $O
echo "=== chain bench"; timeout 150 python tools/peer_gather_bench.py chain 1 2 2>&1 | grep "process(es)\|Error" | tee $O/chain_bench.txt
echo "=== tests"; timeout 300 python -m pytest tests/test_gpu_peer.py -q -p no:cacheprovider > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
