#!/bin/bash
# upper-bound probe: the step with one kernel family skipped (results are garbage, timing only).  Needs a measurement build:
#   LOTUS_BUILD_DEFINES=LOTUS_EXP_SKIP_PROBE python robot-3dlotus_amd/csrc/build.py --force   (and a plain --force rebuild afterwards)
for k in none ln_fwd_kernel ln_bwd_kernel bn_stat_kernel bn_apply_kernel attn_fwd_kernel attn_bwd_kernel xq_fwd_kernel xq_bwd_kernel conv_pairs_kernel conv_wgrad_kernel conv_smallcin reduce_parts_kernel colpart_reduce_kernel conv_tap_reduce gemm_dma_kernel "gemm_kernel<" "pool_max,unpool" none; do
  LOTUS_EXP_SKIP="$k" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-fresh-batches --no-side-workloads --no-roofline 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip %-24s %8.1f samples/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))
except Exception as e: print('skip', sys.argv[1], 'failed', e)" "$k"
done
