mkdir -p gpurun_out
python -c "
import ctypes
hip=ctypes.CDLL('libamdhip64.so'); a,b=ctypes.c_int(0),ctypes.c_int(0); print('prio range', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a),ctypes.byref(b)), a.value, b.value)"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "subm_conv_fwd_dgrad_wgrad" > gpurun_out/t_wg.txt 2>&1; tail -2 gpurun_out/t_wg.txt
LOTUS_CONV_WG_CHUNK=512 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "subm_conv_fwd_dgrad_wgrad" > gpurun_out/t_wg2.txt 2>&1; tail -2 gpurun_out/t_wg2.txt
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
for i in 1 2; do for cfg in "X=0" "LOTUS_CONV_WG_CHUNK=1024" "LOTUS_CONV_WG_CHUNK=512" "LOTUS_SIDE_LOWPRIO=1" "LOTUS_SIDE_LOWPRIO=1 LOTUS_CONV_WG_CHUNK=1024"; do v=$(env $cfg timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>&1 | tail -1); echo "$v $cfg"; done; done | tee gpurun_out/ab_wg.txt
