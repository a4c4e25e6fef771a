#!/bin/bash
# r04p22: the alternating image walk (EXPO_CHAIN_TILE_MIB=0) after the parity fix at the forward -> backward boundary.
OUT=${1:-gpurun_out/r04p22}; mkdir -p $OUT
Q="--no-cpu-baseline --cold-shape none --no-legs --no-per-kernel"
for i in 1 2 3; do
  EXPO_CHAIN_TILE_MIB=0 python bench.py --shape 256,512,512 $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('untiled, two lanes: %.4f ms' % d['ms_per_step'])"
  EXPO_CHAIN_TILE_MIB=0 EXPO_CHAIN_STREAMS=1 python bench.py --shape 256,512,512 $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('untiled, one lane:  %.4f ms' % d['ms_per_step'])"
done
python bench.py --shape 256,512,512 $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile-major:         %.4f ms' % d['ms_per_step'])"
