#!/bin/bash
cd /root/repo
O=gpurun_out
for v in lin_tl lin_tl_stream lin_tl_noload; do
  echo "== $v" >> $O/tl2.log
  MC_HSACO=tools/variants/$v.hsaco GEOMS=512x1 timeout -k 10 120 python3 tools/lin_timeline.py >> $O/tl2.log 2>> $O/tl2.err
done
NCCL_DEBUG=WARN timeout -k 10 300 python3 -m pytest tests/test_pipeline_gpu.py -x -q > $O/rccl_dbg2.log 2>&1
echo rccl rc=$?; tail -3 $O/rccl_dbg2.log
