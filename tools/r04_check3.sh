cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests -m gpu -x -q > /root/repo/gpurun_out/t_all.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_all.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_all.log
cd /root/repo
bash tools/ab_env.sh 3 "MC_LAZY_PICK=1" "MC_LAZY_PICK=0" --steps 256 --warmup 32 --no-roofline
for i in 1 2 3; do for E in "MC_HANDOFF_FAST=1" "MC_HANDOFF_FAST=0"; do echo "$E $(env $E CASE=tinyllama MC_SKIP_FILL=1 python3 tools/configs_run.py)"; done; done
