set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r8e; mkdir -p $O
R=$PWD; cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -o c3 -- python $R/tools/pmc_mfma.py --n 8192 --k 8192 --m 64 --layers 12 > $R/$O/kt.log 2>&1
cd $R; f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c3_kernel_stats.csv && head -6 $O/c3_kernel_stats.csv | cut -c1-60,150-260; rm -rf $O/kt
