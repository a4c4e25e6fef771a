#!/bin/bash
# Builds the C restatement (oracle/filters_c.c) twice, next to the source (git-ignored, travels with gpurun):
#   liboracle_c_f64.so   REAL=double, strict IEEE (no contraction), OpenMP over pixel blocks with a fixed-order
#                        final sum: the checker used by tests/ and smoke(); also returns the sums of absolute
#                        terms of every parameter gradient (ORACLE_ABS_TERMS)
#   liboracle_c_f32.so   REAL=float, OpenMP, AVX2+FMA: the cpu_baseline leg of bench.py (the reference's dtype)
set -euo pipefail
cd "$(dirname "$0")"
gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared -DREAL=double -DORACLE_ABS_TERMS filters_c.c -o liboracle_c_f64.so -lm
gcc -O3 -march=x86-64-v3 -fno-math-errno -fopenmp -fPIC -shared -DREAL=float filters_c.c -o liboracle_c_f32.so -lm
echo "built $(pwd)/liboracle_c_f64.so $(pwd)/liboracle_c_f32.so"
