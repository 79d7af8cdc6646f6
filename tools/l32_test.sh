cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest /root/repo/tests/test_attn_kernels_gpu.py /root/repo/tests/test_context_gpu.py -x -q -k "norm_qkv or llama32" > /root/repo/gpurun_out/t_l32.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_l32.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_l32.log
bash /root/repo/tools/l32_run.sh
