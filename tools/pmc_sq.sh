#!/bin/bash
# Runs ON THE GPU BOX: SQ / GRBM counters of the fused GEMVs, one rocprofv3 --pmc pass per group
# (counter passes attach to tools/pmc_gemv.py; never combined with trace domains).
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pq; timeout -k 5 200 rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/pmc_gemv.py > /dev/null 2> /tmp/pq.err
  echo "== $G"; python3 /root/repo/tools/pmc_summary.py /tmp/pq mc_gemv_i4_bfloat_p1_e2 2>&1 | tail -8
  tail -2 /tmp/pq.err | cut -c1-200
done
