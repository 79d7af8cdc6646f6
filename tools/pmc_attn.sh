cd /tmp; export TMPDIR=/tmp
for G in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/pq; timeout -k 5 200 rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/prefill_bench.py 2048 > /dev/null 2> /tmp/pq.err
  echo "== $G"
  for K in mc_pf_attn2_bfloat_hd128 mc_pf_gemm128_i4_bfloat_d2_e0; do python3 /root/repo/tools/pmc_summary.py /tmp/pq $K 2>&1 | tail -5; done
done
