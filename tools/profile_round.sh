#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 evidence of one round, written to gpurun_out/ (copied into profiles/ by hand).
#   1. kernel durations of the headline bench (eager launches: hipGraph replay under rocprofv3 segfaults with ROCm 7.2)
#   2. HBM traffic counters (separate --pmc passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; gfx950 correction in
#      tools/pmc_traffic.py) of the GEMVs (tools/pmc_gemv.py) and of the decode attention (tools/pmc_attn_decode.py)
#   3. matrix-pipe counters of the decode attention kernels (north_star: MFMA utilisation for the QK^T / PV bmm)
#   4. kernel durations of the other BASELINE configs (tools/configs_run.py, eager)
# usage (from the build container): tools/gpu_profile.sh r04 [what]   -- writes .build_head, then runs this through gpurun
# usage: tools/profile_round.sh r03 [what: all|stats|traffic|mfma|configs]
set -u
R=${1:-r03}; WHAT=${2:-all}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
stats() { # dir -> csv
  cp $(find $1 -name "*kernel_stats.csv" | head -1) $2 && head -10 $2 | cut -d, -f1-4
}
if [ $WHAT = all ] || [ $WHAT = stats ]; then
  rm -rf /tmp/p_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python3 /root/repo/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs > $OUT/${R}_bench_under_trace.json 2> /tmp/p_trace.err
  stats /tmp/p_trace $OUT/${R}_kernel_stats.csv
fi
if [ $WHAT = all ] || [ $WHAT = traffic ]; then
  # (the names bench.py's pmc_traffic() and profiles/README expect: rNN_pmc_traffic.json = the GEMVs, rNN_pmc_attn.json = the
  #  decode attention; pmc_traffic.py stamps the git head it finds in .build_head -- .git does not travel to the GPU box)
  for P in gemv:traffic attn_decode:attn; do
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/p_$C; rocprofv3 --pmc $C --output-format csv -d /tmp/p_$C -- python3 /root/repo/tools/pmc_${P%%:*}.py > /dev/null 2> /tmp/p_$C.err
    done
    python3 /root/repo/tools/pmc_traffic.py /tmp/p_FETCH_SIZE /tmp/p_WRITE_SIZE > $OUT/${R}_pmc_${P##*:}.json
  done
  grep -A4 "mc_attn_qkv_wo\|mc_attn_wo\|mc_attn_fused\|lin2_p1_e2" $OUT/${R}_pmc_*.json | head -40
fi
if [ $WHAT = all ] || [ $WHAT = mfma ]; then
  : > $OUT/${R}_pmc_attn_mfma.log
  # the three forms of the decode attention: one launch with Wo (default), one launch (MC_ATTN_WO=0), scores + P.V (MC_ATTN_FUSED=0)
  for F in "MC_ATTN_QKV=1" "MC_ATTN_QKV=0" "MC_ATTN_WO=0" "MC_ATTN_FUSED=0"; do
    for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
      rm -rf /tmp/pq; export $F; rocprofv3 --pmc $G --output-format csv -d /tmp/pq -- python3 /root/repo/tools/pmc_attn_decode.py > /dev/null 2> /tmp/pq.err; unset MC_ATTN_WO MC_ATTN_FUSED MC_ATTN_QKV
      echo "== $F  $G" >> $OUT/${R}_pmc_attn_mfma.log
      for K in mc_attn_qkv_wo_i4_bfloat mc_attn_wo_i4_bfloat mc_attn_fused_bfloat mc_attn_scores_bfloat mc_attn_pv_bfloat; do python3 /root/repo/tools/pmc_summary.py /tmp/pq $K 2>/dev/null | tail -6 >> $OUT/${R}_pmc_attn_mfma.log; done
    done
  done
  cat $OUT/${R}_pmc_attn_mfma.log
fi
if [ $WHAT = all ] || [ $WHAT = configs ]; then
  for C in tinyllama gemma 70b int8; do
    rm -rf /tmp/p_c; CASE=$C MC_NO_GRAPH=1 MC_SKIP_FILL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 /root/repo/tools/configs_run.py > $OUT/${R}_config_$C.json 2> /tmp/p_c.err
    cat $OUT/${R}_config_$C.json
    stats /tmp/p_c $OUT/${R}_kernel_stats_$C.csv
  done
fi
