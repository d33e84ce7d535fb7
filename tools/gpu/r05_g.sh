#!/bin/bash
# (run while the packed pairs were the default build: -DSF_SCALAR_PAIRS selected the plain pairs then; since then plain pairs are the default and -DSF_PACKED_PAIRS selects the round-4 kernels)
# Round 5, call G: the kernel timeline of a pass under two builds of the integrate kernel (who overlaps whom).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
for v in "$@"; do
  name=${v%%:*}; export SCANFUSE_BUILD_FLAGS="${v#*:}"
  cd $R
  python -c "from scannet_amd import build; build.build(force=True)" > $O/build_$name.log 2>&1
  cd /tmp
  rm -rf /tmp/kt_$name; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --no-profile --repeats 1 --teardown > $O/kt_$name.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/kt_$name -name "*.db" | head -1) 2>&1 | head -8 | cut -c1-150 > $O/kt_$name.txt
  python $R/tools/timeline.py $(find /tmp/kt_$name -name "*.db" | head -1) --skip k_synth --skip at:: > $O/timeline_$name.txt 2>&1
  echo "== $name"; cat $O/kt_$name.txt; sed -n 1,30p $O/timeline_$name.txt; tail -2 $O/timeline_$name.txt; grep -o '"value": [0-9.]*' $O/kt_$name.log | head -2
done
