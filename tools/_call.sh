set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "kq or representable or mfma_kernel_variants or config3" > $O/pytest_kq.log 2>&1; tail -5 $O/pytest_kq.log
timeout 600 python tools/route_ab.py small mid tall 2>&1 | grep -v amdgpu.ids | tee $O/route_ab.txt
