set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r7b; mkdir -p $O
echo "=== chain bench"; BNB_MI355X_PEER_WAIT_POLLS=3000000 timeout 150 python tools/peer_gather_bench.py chain 1 2 4 2>&1 | grep "process(es)" | tee $O/chain_bench.txt
echo "=== tests"; BNB_MI355X_PEER_WAIT_POLLS=3000000 timeout 400 python -m pytest tests/test_gpu_peer.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "peer or bench_multi or stream_kernel_geometry" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
echo "=== bench"; timeout 400 python bench.py --steps 100 --warmup 10 --profile-out $O/bench_kernel_stats.csv --span-out $O/bench_kernel_span.json 2>$O/bench.err | tail -1 > $O/bench_n1.json; cut -c1-1800 $O/bench_n1.json
echo "=== pmc kq C3"; bash tools/pmc_mfma.sh $O/pmc_kq_c3 gemm4_mfma_kq_kernel 0 --n 8192 --k 8192 --m 64 --layers 6 2>&1 | tail -40
