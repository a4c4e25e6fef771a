#!/bin/bash
# r04p7: Tone / Color backward without the one-deep software prefetch (93 / 64 VGPRs -> occupancy 5 / 8 instead of 4 / 4)
R=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --cold-shape none --no-legs"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['per_kernel']
print('$1 ms %.4f | bwd_C %.1f bwd_T %.1f bwd_E %.1f fwd_C %.1f | sum %.1f' % (d['ms_per_step'], pk['bwd_C']['ms']*1e3, pk['bwd_T']['ms']*1e3, pk['bwd_E']['ms']*1e3, pk['fwd_C']['ms']*1e3, sum(v['ms'] for v in pk.values())*1e3))"; }
for rep in 1 2 3; do
  python $R/bench.py $Q 2>/dev/null | show "C base "
  EXPO_HIP_LIB=$R/tools/r04/libs/nopf.so python $R/bench.py $Q 2>/dev/null | show "C nopf "
done
python $R/bench.py $Q --shape B --steps 50 2>/dev/null | show "B base "
EXPO_HIP_LIB=$R/tools/r04/libs/nopf.so python $R/bench.py $Q --shape B --steps 50 2>/dev/null | show "B nopf "
python $R/bench.py $Q --shape 256,512,512 --steps 6 --warmup 2 --kernel-reps 10 2>/dev/null | show "cold base "
EXPO_HIP_LIB=$R/tools/r04/libs/nopf.so python $R/bench.py $Q --shape 256,512,512 --steps 6 --warmup 2 --kernel-reps 10 2>/dev/null | show "cold nopf "
