mkdir -p gpurun_out
timeout 600 python tools/wgrad_ab.py LOTUS_WGRAD_STREAM 0 1 12 16 > gpurun_out/wgrad_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/t_ops.txt 2>&1
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
for i in 1 2; do for m in 0 1 12; do echo "STREAM=$m" >> gpurun_out/ab_stream.txt; LOTUS_WGRAD_STREAM=$m timeout 300 $B 2>&1 | tail -1 | cut -c1-160 >> gpurun_out/ab_stream.txt; done; done
tail -40 gpurun_out/wgrad_ab.txt; tail -5 gpurun_out/t_ops.txt; cat gpurun_out/ab_stream.txt
