#!/bin/bash
# r03p20: what costs the Color / Tone backward its 4-5 us over the light kernels -- the per-block slope-table staging or
# the 24 x 8 packed accumulations?  Timing-only builds (results wrong by construction).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
Q="--no-cpu-baseline --cold-shape none"
pk() { python -c "
import json,sys
d=json.load(open('/tmp/b.json')); pk=d['per_kernel']; r=d['roofline']
print('$1: step %.4f ms | bwd C %.1f T %.1f BW %.1f E %.1f | ev overhead %.2f us' % (d['ms_per_step'], pk['bwd_C']['ms']*1e3, pk['bwd_T']['ms']*1e3, pk['bwd_BW']['ms']*1e3, pk['bwd_E']['ms']*1e3, r['event_pair_overhead_ms']*1e3))
"; }
for rep in 1 2; do
for v in base probe_NOSTAGE probe_NOACC probe_both; do
  lib=$R/tools/r03/libs/$v.so; [ $v = base ] && lib=$R/exposure_amd/libexposure_hip.so
  EXPO_HIP_LIB=$lib python bench.py $Q > /tmp/b.json 2>/dev/null
  pk "$v r$rep"
done
done
