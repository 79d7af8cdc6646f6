cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests/test_pipeline_gpu.py /root/repo/tests/test_attn_kernels_gpu.py -x -q -k "pipeline or norm_qkv" > /root/repo/gpurun_out/t_new.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_new.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_new.log
cd /root/repo
for A in "--gpus 1" "--gpus 2 --share-device" "--gpus 2 --share-device --no-graph" "--gpus 4 --share-device" "--gpus 4 --share-device --no-graph"; do
  python3 bench.py $A --steps 128 --warmup 16 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$A', round(d['value'],1), d['config'].get('hipgraph'), [ (c.get('config','')[:30], round(c.get('tokens_per_s',0),1)) for c in d.get('other_configs',[])])" | tee -a gpurun_out/r04_share_device.log
done
