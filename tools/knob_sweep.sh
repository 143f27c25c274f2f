#!/bin/bash
# One-box sweep of the library's tuning switches on the headline step (each setting = one short bench run; the baseline
# is repeated so that the box's own noise is visible).   bash tools/knob_sweep.sh > gpurun_out/knob_sweep.txt
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
run() { v=$(env $1 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null); echo "$v  $1"; }
run "X=base"
for cfg in "LOTUS_CONV_OS_F32=2" "LOTUS_CONV_OS_F32=3" "LOTUS_GEMM_TILE_FWD=1" "LOTUS_GEMM_TILE_FWD=2" "LOTUS_GEMM_TILE_FWD=1 LOTUS_GEMM_TILE_FWD_MINBLOCKS=2048" \
  "LOTUS_WGRAD_BLOCKS=512" "LOTUS_WGRAD_BLOCKS=2048" "LOTUS_WGRAD_MINROWS=512" "LOTUS_GEMM_RING_F32=1" "LOTUS_GEMM_RING_F32=4" "LOTUS_GEMM_RING_WG=1" \
  "X=base2" "LOTUS_WGRAD_STREAM=1" "LOTUS_WGRAD_FUSE_MAX=16" "LOTUS_GEMM_MINK=256" "LOTUS_CONV_TAP_MINC=128" "LOTUS_CONV_TAP_ROWS=65536" "LOTUS_CONV_TAP_ROWS=65536 LOTUS_CONV_TAP_MINC=128" \
  "LOTUS_XQ=2" "LOTUS_SPLITK_FUSED=0" "LOTUS_PAIR=1" "LOTUS_PAIR=0" "LOTUS_BN_FUSED=0" "LOTUS_KV_GROUP=0" "LOTUS_GEMM_RING_F32_BLOCKS=2048" "LOTUS_GEMM_RING_F32_BLOCKS=100000" \
  "X=base3" "LOTUS_SIDE_STREAM=0" "LOTUS_HIPRIO=0" "LOTUS_BN_ROWS=4" "LOTUS_CONV_NZ27_BLOCKS=1" "LOTUS_GEMM_BK=32" "LOTUS_GEMM_BK=64" "X=base4"; do
  run "$cfg"
done
