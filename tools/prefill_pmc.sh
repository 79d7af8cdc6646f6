#!/bin/bash
# Runs ON THE GPU BOX: matrix-pipe / wave-state / LDS counters of the prompt kernels of one prompt length (separate rocprofv3 --pmc passes,
# --kernel-trace/--stats only in other runs).  usage: tools/prefill_pmc.sh r05 512
R=${1:-r05}; N=${2:-512}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
LOG=$OUT/${R}_prefill_pmc_${N}.log
: > $LOG
cd /tmp; export TMPDIR=/tmp
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pq
  rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/prefill_bench.py $N > /dev/null 2> /tmp/pq.err
  echo "== $N rows: $G" >> $LOG
  for K in mc_pf_gemm8_w_bfloat_e3 mc_pf_gemm8_w_bfloat_e2 mc_pf_gemm8_w_bfloat_e0 mc_pf_gemm8_i4_bfloat_e3 mc_pf_gemm8_i4_bfloat_e2 mc_pf_gemm8_i4_bfloat_e0 mc_pf_attn8_bfloat mc_pf_attn4_bfloat; do python3 /root/repo/tools/pmc_summary.py /tmp/pq $K 2>/dev/null >> $LOG; done
done
cat $LOG
