set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1 BNB_MI355X_PEER_WAIT_POLLS=1000000
O=gpurun_out/r6h; mkdir -p $O
timeout 100 python tools/peer_gather_bench.py chain 1 2 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tee -a $O/chain_bench.txt
timeout 200 python -m pytest tests/test_gpu_peer.py -q -x -p no:cacheprovider > $O/pytest_peer.log 2>&1; tail -3 $O/pytest_peer.log
