#!/bin/bash
# round 6: can the allocation kernel's waves live BESIDE the integrate kernel's instead of in their place?  integrate in halves (72 registers) capped at
# 5 / 6 workgroups per CU, with and without a cap on the allocation workgroups
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06l
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --no-single-frame --no-pmc --repeats 3"
run() {
  tag=$1; shift
  ( SF_BENCH_DETAIL=$O/detail_$tag.json timeout 300 python bench.py $Q "$@" ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; j=json.load(open('$O/detail_$tag.json')); print('$tag', j['value'], j['repeats']['value_min'], j['repeats']['value_max'], 'kernel us', j['roofline']['avg_kernel_us'])"
}
run default
run nj2 --tune int_nj=2
run nj2_w6 --tune int_nj=2 --tune int_wgs=6
run nj2_w5 --tune int_nj=2 --tune int_wgs=5
run nj2_w6_a2 --tune int_nj=2 --tune int_wgs=6 --tune alloc_wgs=2
run nj2_w6_a1 --tune int_nj=2 --tune int_wgs=6 --tune alloc_wgs=1
run nj2_w5_a2 --tune int_nj=2 --tune int_wgs=5 --tune alloc_wgs=2
run nj4_w4 --tune int_wgs=4
run nj4_a2 --tune alloc_wgs=2
run nj4_a1 --tune alloc_wgs=1
