#!/bin/bash
# r04p14: does the helper stream of the two-lane chain calls share a hardware queue with its caller?  Cold chain calls
# (256x512x512, eager): the probed pairing (default), the first helper unprobed, 8 hardware queues, one lane.
# (The first version of this script compared helper keys / priorities: profiles/r04_p14_helper_queue.txt.)  Usage (GPU box): bash tools/r04/helper_queue.sh OUT_DIR
OUT=${1:-gpurun_out/r04p14}
mkdir -p $OUT
run() {
  tag=$1; shift
  env "$@" python bench.py --no-legs --no-cpu-baseline --steps 10 > $OUT/hq_$tag.json 2> $OUT/hq_$tag.err
  python - $OUT/hq_$tag.json $tag <<PY
import json, sys
d = json.load(open(sys.argv[1])); c = d['roofline']['hbm_cold']
print('%-22s headline %.4f ms   cold chain call %.4f ms (%.3f)  runs %s' % (sys.argv[2], d['ms_per_step'],
      c['chain_call_ms_per_step'], c['chain_call_frac'], ' '.join('%.3f' % v for v in c['chain_call_ms_runs'])))
PY
}
run probed               EXPO_X=0
run unprobed             EXPO_CHAIN_HELPER_PROBE=0
run probed_8queues       GPU_MAX_HW_QUEUES=8
run one_lane             EXPO_CHAIN_STREAMS=1
