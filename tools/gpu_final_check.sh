# Runs ON THE GPU BOX (gpurun -- bash tools/gpu_final_check.sh): the whole GPU suite, smoke(), then the default bench line -> gpurun_out/
cd /root/repo
python3 -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; tail -c 600 gpurun_out/r04_bench_default.json
