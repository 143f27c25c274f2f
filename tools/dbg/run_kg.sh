mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "linear" > gpurun_out/t_kg.txt 2>&1; tail -4 gpurun_out/t_kg.txt
timeout 600 python tools/gemm_ab.py LOTUS_GEMM_KG 1 4 > gpurun_out/gemm_ab_kg.txt 2>&1; grep -v "^fwd 65536\|^dgrad 65536\|^fwd 23894\|^dgrad 23894" gpurun_out/gemm_ab_kg.txt | head -70
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
for i in 1 2 3; do for cfg in "LOTUS_GEMM_KG=1" "LOTUS_GEMM_KG=4" "LOTUS_GEMM_KG=4 LOTUS_GEMM_KG_BLOCKS=512" "LOTUS_GEMM_KG=4 LOTUS_GEMM_KG_MAXK=3072"; do v=$(env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"); echo "$v $cfg"; done; done | tee gpurun_out/ab_kg.txt
