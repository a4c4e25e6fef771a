#!/bin/bash
# r03p15: the chain's two half-batches on two streams (fork / join inside expo_chain_fwd / _bwd): off vs on by shape
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p15
rm -rf $OUT; mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --cold-shape none --no-per-kernel"
for rep in 1 2; do
for shape in 16,512,512 32,512,512 64,512,512 128,512,512 64,64,64; do
  for st in 1 2; do
    EXPO_CHAIN_STREAMS=$st python bench.py $Q --shape $shape 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shape streams=$st rep$rep: %.4f ms  %.0f Mpx/s' % (d['ms_per_step'], d['value']))"
  done
done
done
python bench.py $Q | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default: %.4f ms  %.0f Mpx/s' % (d['ms_per_step'], d['value']), d['config']['launch'])"
python bench.py $Q --graph off | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default eager: %.4f ms  %.0f Mpx/s' % (d['ms_per_step'], d['value']), d['config']['launch'])"
timeout 900 python -m pytest tests/test_hip_filters.py tests/test_hip_reduction.py -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
