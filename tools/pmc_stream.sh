#!/usr/bin/env bash
# PMC passes (one rocprofv3 run per counter group, --kernel-trace only beside --pmc) over tools/pmc_stream.py.
# usage: bash tools/pmc_stream.sh <outdir> <kernel substring> [pmc_stream.py args...]
set -u
OUT=$1; SUB=$2; shift 2
R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$OUT/pass$i -- python $R/tools/pmc_stream.py "$@" > $R/$OUT/pass$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/$OUT/pass$i.log)"
done
cd $R
python tools/pmc_summary.py $OUT "$SUB" | tee $OUT/summary.txt
