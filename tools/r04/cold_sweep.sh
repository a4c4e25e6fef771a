#!/bin/bash
# r04: tile-major chain beyond the Infinity Cache (256x512x512) vs the alternating walk; config-5 shape knob sweep
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04p4; mkdir -p $O
Q="--no-cpu-baseline --cold-shape none --no-legs --no-per-kernel"
run() { # tag, env..., -- args
  tag=$1; shift
  env "$@" > /dev/null 2>&1
}
for t in 0 48 64 96 128 192; do
  EXPO_CHAIN_TILE_MIB=$t python $R/bench.py --shape 256,512,512 $Q --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cold tile_mib=$t  ms %.4f  chain GB/s %.0f  streams %s' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps'], d['config']['chain_streams']))"
done
EXPO_CHAIN_TILE_MIB=96 EXPO_CHAIN_STREAMS=1 python $R/bench.py --shape 256,512,512 $Q --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cold tile 96 one stream  ms %.4f  chain GB/s %.0f' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps']))"
for t in 0 96; do
  EXPO_CHAIN_TILE_MIB=$t python $R/bench.py --shape 128,512,512 $Q --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('128 images tile_mib=$t  ms %.4f  chain GB/s %.0f' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps']))"
done
# config-5 shape (16x512x512): knobs
B="--shape B $Q --steps 50 --warmup 10"
python $R/bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B default ms %.4f GB/s %.0f' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps']))"
for kv in EXPO_FWD_GROUPS_PER_THREAD=2 EXPO_BWD_GROUPS_PER_THREAD=2 EXPO_TONE_GROUPS_PER_THREAD=1 EXPO_COLOR_GROUPS_PER_THREAD=1 EXPO_TONE_GROUPS_PER_THREAD=4 EXPO_COLOR_GROUPS_PER_THREAD=4 EXPO_STREAM_MIN_BYTES=67108864 EXPO_CHAIN_STREAMS=2; do
  env $kv python $R/bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $kv ms %.4f GB/s %.0f' % (d['ms_per_step'], d['config']['chain_algorithmic_GBps']))"
done
python -m pytest $R/tests/test_hip_filters.py $R/tests/test_hip_reduction.py -m gpu -q -x 2>&1 | tail -2
