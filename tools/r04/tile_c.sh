#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --cold-shape none --no-legs --no-per-kernel"
for rep in 1 2; do
for cfg in "EXPO_CHAIN_TILE_MIN_MIB=256" "EXPO_CHAIN_TILE_MIN_MIB=64 EXPO_CHAIN_TILE_MIB=48" "EXPO_CHAIN_TILE_MIN_MIB=64 EXPO_CHAIN_TILE_MIB=24" "EXPO_CHAIN_TILE_MIN_MIB=64 EXPO_CHAIN_TILE_MIB=48 EXPO_CHAIN_STREAMS=2" "EXPO_CHAIN_TILE_MIN_MIB=64 EXPO_CHAIN_TILE_MIB=24 EXPO_CHAIN_STREAMS=2" "EXPO_CHAIN_TILE_MIN_MIB=64 EXPO_CHAIN_TILE_MIB=12 EXPO_CHAIN_STREAMS=2"; do
  env $cfg python $R/bench.py $Q --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C $cfg  ms %.4f  GB/s %.0f streams %s' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps'], d['config']['chain_streams']))"
done
done
