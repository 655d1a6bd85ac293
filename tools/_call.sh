set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r8d; mkdir -p $O
timeout 170 python tools/route_ab.py small mid tall 2>&1 | grep -v amdgpu.ids | tee $O/route_ab_final.txt
