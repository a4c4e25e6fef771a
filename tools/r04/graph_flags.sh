#!/bin/bash
# r04p24: do the HIP runtime's graph tunables (strings libamdhip64.so: DEBUG_CLR_GRAPH_PACKET_CAPTURE,
# DEBUG_HIP_GRAPH_BATCH_SIZE, DEBUG_HIP_FORCE_GRAPH_QUEUES) move the graph-replayed workloads?
OUT=${1:-gpurun_out/r04p24}; mkdir -p $OUT
run() { tag=$1; shift
  t=$(env "$@" python bench.py --workload train --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  c=$(env "$@" python bench.py --no-legs --no-cpu-baseline --cold-shape none --no-per-kernel 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  a=$(env "$@" python bench.py --shape A --no-legs --no-cpu-baseline --cold-shape none --no-per-kernel 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$tag: train $t ms, chain 64x512x512 $c ms, chain 64x64x64 $a ms"; }
run default EXPO_X=0
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run batch_1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch_16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run batch_256 DEBUG_HIP_GRAPH_BATCH_SIZE=256
run batch_4096 DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run queues_1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run queues_2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run default_again EXPO_X=0
