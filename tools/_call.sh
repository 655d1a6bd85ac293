set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r8f; mkdir -p $O
timeout 30 python tools/route_ab.py c3 2>&1 | grep -v amdgpu.ids | tee $O/kq_combine_ab.txt
timeout 95 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "kq_kernel_geometries or kq_kernel_is_deterministic or config3 or representable" > $O/pytest_kq.log 2>&1; tail -4 $O/pytest_kq.log
