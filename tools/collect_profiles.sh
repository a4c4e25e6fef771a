#!/bin/bash
# Copies what tools/profile_all.sh wrote under gpurun_out/final/ into profiles/ (tracked) and rebuilds
# profiles/traffic.json; tests/test_profiles_consistency.py then checks the set is consistent.
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=gpurun_out/final
TAG=${1:-r01_final}
for f in bench_chain bench_chain_A bench_chain_B bench_chain_f32 bench_infer_B bench_train bench_extra; do
  [ -s $SRC/$f.json ] && cp $SRC/$f.json profiles/${TAG}_$f.json
done
cp $SRC/kernel_stats.csv profiles/${TAG}_kernel_stats.csv
for c in fetch_size write_size; do
  cp $SRC/pmc_$c.csv profiles/${TAG}_pmc_$c.csv
  cp $SRC/pmc_${c}_calibration.csv profiles/${TAG}_pmc_${c}_calibration.csv
done
cp $SRC/membench.txt profiles/${TAG}_membench.txt
python tools/make_traffic.py $SRC profiles/traffic.json > /dev/null
python -m pytest tests/test_profiles_consistency.py -q
