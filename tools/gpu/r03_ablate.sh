#!/bin/bash
# what bounds k_alloc_ray: the kernel alone (overlap=0) with parts switched off; durations from a kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k
mkdir -p $O
cd /tmp
for ab in 0 1 4; do
  rm -rf /tmp/kt_$ab
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_$ab -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --no-profile --repeats 1 --steps 640 --warmup 64 --tune overlap=0 --tune alloc_ablate=$ab --teardown > $O/kt_$ab.log 2>&1
  echo "ablate $ab: $(python $R/tools/rocpd_summary.py $(find /tmp/kt_$ab -name '*.db' | head -1) | grep k_alloc_ray | cut -c60-130)"
done
