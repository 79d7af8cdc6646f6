# Runs ON THE GPU BOX: prompt-pass tests, then same-box alternating A/B of the round-4 prompt-pass knobs
cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests/test_prefill_gpu.py -x -q > /root/repo/gpurun_out/t_pf2.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_pf2.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_pf2.log
L="8 128 256 512 2048"
for i in 1 2; do
  echo "== default"; python3 /root/repo/tools/prefill_bench.py $L
  echo "== MC_PF_ACT_EPI=0"; MC_PF_ACT_EPI=0 python3 /root/repo/tools/prefill_bench.py $L
  echo "== MC_PF_SPLITS_OLD=1"; MC_PF_SPLITS_OLD=1 python3 /root/repo/tools/prefill_bench.py $L
done 2>&1 | tee /root/repo/gpurun_out/r04_pf_ab.log
