mkdir -p gpurun_out
OUT=gpurun_out/bn_ab.txt; : > $OUT
for cfg in "" "LOTUS_BN_FUSED_GRID=1024" "LOTUS_BN_BWD_ROWS=4" "LOTUS_BN_FUSED_GRID=1024 LOTUS_BN_BWD_ROWS=4" "LOTUS_BN_FUSED_GRID=512 LOTUS_BN_BWD_ROWS=4" "LOTUS_BN_FUSED_GRID=1024 LOTUS_BN_BWD_ROWS=4 LOTUS_BN_FUSED_ROWS=4"; do
  echo "=== $cfg" >> $OUT
  env $cfg timeout 300 python tools/membound_bench.py 2>&1 | grep -i "bn" >> $OUT
done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "batchnorm or bn or wgrad" > gpurun_out/t_ops2.txt 2>&1
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
for i in 1 2 3; do for cfg in "X=0" "LOTUS_BN_FUSED_GRID=1024 LOTUS_BN_BWD_ROWS=4" "LOTUS_BN_BWD_ROWS=4"; do echo "$cfg" >> gpurun_out/ab_bn.txt; env $cfg timeout 300 $B 2>&1 | tail -1 | cut -c1-130 >> gpurun_out/ab_bn.txt; done; done
cat $OUT; tail -3 gpurun_out/t_ops2.txt; cat gpurun_out/ab_bn.txt
