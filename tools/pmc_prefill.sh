#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + MFMA counters of the prompt pass (tools/prefill_bench.py 2048)
# and of the decode attention (tools/long_ctx_profile.py), one rocprofv3 pass each; summaries to
# gpurun_out/.  Counter passes carry no trace domains.
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
rm -rf /tmp/pp; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 /root/repo/tools/prefill_bench.py 2048 > /tmp/pp.out 2>&1
cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $OUT/r01_prefill_kernel_stats.csv
head -12 $OUT/r01_prefill_kernel_stats.csv | cut -c1-130
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pq; timeout -k 5 300 rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/prefill_bench.py 2048 > /dev/null 2> /tmp/pq.err
  echo "== $G"
  for K in mc_pf_gemm128_i4_bfloat_e0 mc_pf_attn_bfloat_hd128; do python3 /root/repo/tools/pmc_summary.py /tmp/pq $K 2>&1 | tail -6; done
  tail -1 /tmp/pq.err | cut -c1-160
done
