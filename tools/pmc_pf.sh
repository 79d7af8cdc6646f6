# Runs ON THE GPU BOX: matrix-pipe / LDS / wait counters of the tiled prompt GEMM (512-row prompt), old kernel (MC_PF3=0) and mc_pf3
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04_prefill_pmc.log; : > $OUT
for E in MC_PF3=0 MC_PF3=1; do
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  rm -rf /tmp/pq; export $E; timeout -k 5 200 rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/prefill_bench.py 512 > /dev/null 2> /tmp/pq.err
  echo "== $E  $G" >> $OUT
  for K in mc_pf_gemm256_i4_bfloat_d2_e0 mc_pf3_gemm_i4_bfloat_e0; do python3 /root/repo/tools/pmc_summary.py /tmp/pq $K 2>/dev/null | tail -6 >> $OUT; done
done; done
cat $OUT
