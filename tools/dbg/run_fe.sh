mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_model.py -x -q > gpurun_out/t_fe.txt 2>&1; tail -3 gpurun_out/t_fe.txt
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-side-workloads"
for i in 1 2 3; do for cfg in "LOTUS_FE_FINISH_SIDE=0" "LOTUS_FE_FINISH_SIDE=1"; do env $cfg timeout 300 $B 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('fresh_batches',{}).get('value'), '$cfg')"; done; done | tee gpurun_out/ab_fe.txt
