#!/bin/bash
# GPU box: stream lab, lin ablations, RCCL bring-up diagnosis
cd /root/repo
O=gpurun_out
env | grep -i "HSA\|NCCL\|RCCL\|ROCR\|HIP_VIS" > $O/env.log
export DBGS=0 MC_GEMV_LIN=1
timeout -k 10 200 python3 tools/gemv_ab.py metalchat_amd/lib/metalchat.hsaco 512x1 > $O/abm1.log 2> $O/abm1.err
timeout -k 10 200 python3 tools/gemv_ab.py tools/variants/lin_stream.hsaco 512x1 >> $O/abm1.log 2>> $O/abm1.err
timeout -k 10 200 python3 tools/gemv_ab.py tools/variants/lin_noload.hsaco 512x1 >> $O/abm1.log 2>> $O/abm1.err
cat $O/abm1.log
timeout -k 10 300 tools/lds_stream_lab > $O/lab1.log 2> $O/lab1.err
echo lab rc=$?
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV timeout -k 10 200 python3 -m pytest tests/test_pipeline_gpu.py -x -q -k rccl > $O/rccl_dbg.log 2>&1
echo rccl rc=$?
NCCL_DEBUG=INFO timeout -k 10 200 python3 - > $O/rccl_torch.log 2>&1 <<'PY'
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("torch nccl ok", t)
PY
echo torch rc=$?
tail -5 $O/rccl_torch.log
