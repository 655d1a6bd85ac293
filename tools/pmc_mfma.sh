#!/usr/bin/env bash
# PMC passes over the fused op with a tuning knob (one rocprofv3 run per counter group, --kernel-trace only beside --pmc).
# usage: bash tools/pmc_mfma.sh <outdir> <kernel substring> <knob> [pmc_mfma.py args...]
set -u
OUT=$1; SUB=$2; KNOB=$3; shift 3
R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
G4="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$OUT/pass$i -- python $R/tools/pmc_mfma.py --knob $KNOB "$@" > $R/$OUT/pass$i.log 2>&1 || echo "pass $i failed: $(grep -v amdgpu.ids $R/$OUT/pass$i.log | tail -2)"
done
cd $R
python tools/pmc_summary.py $OUT "$SUB" | tee $OUT/summary.txt
rm -rf $OUT/pass*/
