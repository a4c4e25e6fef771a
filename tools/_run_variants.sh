mkdir -p gpurun_out/variants
tools/membench 96 9 40 pol | grep -E "2048 +(rpol L2 L2 S16|rpol3 lds1 pf1)"
for v in base nowork neither nowork_npf; do
  EXPO_HIP_LIB=$PWD/variants_$v.so python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/variants/$v.json 2>/dev/null
  echo $v; python tools/show_bench.py gpurun_out/variants/$v.json | head -3 | tail -1
done
