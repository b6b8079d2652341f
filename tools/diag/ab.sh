#!/bin/bash
# A/B of env settings on one box: tools/diag/ab.sh "VAR=a" "VAR=b" ... (each run: bench.py fp32, 10 steps)
for rep in 1 2; do
for cfg in "$@"; do
  v=$(env $cfg python bench.py --no-cpu-baseline --no-extra --no-bf16 --no-prof $BENCH_ARGS | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$cfg -> $v"
done; done
