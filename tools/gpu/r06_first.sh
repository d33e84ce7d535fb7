#!/bin/bash
# The first GPU call after round 5: what was written after round 5's GPU minutes were spent and has never run on hardware.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_first.sh'
# 1. bin/depthsensing --ranks 2 --share-gpu against the one-rank file (the xfail-until-measured tests: XPASS = it works), verbosely;
# 2. the same mode on a scan-sized file, timed beside the plain tool (two ranks on ONE device cannot be faster: this is the cost of the orchestration --
#    every rank decodes the whole file, the parts travel through /dev/shm, the parent merges);
# 3. the whole GPU suite, smoke, the default bench line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06first
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_zz_depthsensing_ranks.py -m gpu -v -rxX ) > $O/ranks_tests.log 2>&1
grep -E "XPASS|XFAIL|passed|failed|xfailed|xpassed" $O/ranks_tests.log | tail -5
python - > $O/scan.log 2>&1 <<'PY'
import numpy as np, time, os, subprocess
from scannet_amd import sens, synth
W, H, n = 640, 480, 2000
K = synth.intrinsic_matrix(W, H)
depth = np.stack([synth.render_room_depth(synth.trajectory_pose(i * 2, 5578), W, H, noise_frame=i) for i in range(n)])
poses = np.stack([synth.trajectory_pose(i * 2, 5578) for i in range(n)])
sd = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=1, sensor_name="StructureSensor")
sd.add_depth_frames(depth, poses)
sd.save("/tmp/r06.sens"); sd.close()
open("/tmp/p.txt", "w").write("s_SDFVoxelSize = 0.004f;\ns_SDFTruncation = 0.06f;\ns_SDFTruncationScale = 0.02f;\n")
open("/tmp/t.txt", "w").write("//\n")
for extra, out in (([], "/tmp/one.ply"), (["--ranks", "2", "--share-gpu"], "/tmp/two.ply")):
    t = time.time()
    r = subprocess.run(["bin/depthsensing"] + extra + ["/tmp/p.txt", "/tmp/t.txt", "/tmp/r06.sens", out], capture_output=True, text=True)
    print(extra, "rc", r.returncode, "wall %.2f s" % (time.time() - t), "stderr", repr(r.stderr[-300:]))
    print("\n".join(l for l in r.stdout.splitlines() if "Integrated" in l or "Exchange" in l or "Mesh" in l))
print("same file:", open("/tmp/one.ply", "rb").read() == open("/tmp/two.ply", "rb").read())
PY
tail -12 $O/scan.log
( time timeout 1500 python -m pytest tests -m gpu -q -rs ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1200 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
python -c "
import json; j=json.loads([l for l in open('$O/bench_4mm.json') if l.startswith('{')][0]); print(j['value'], j['unit'], (j.get('roofline') or {}).get('frac'), (j.get('end_to_end') or {}).get('frames_per_s'), (j.get('end_to_end_rgbd') or {}).get('frames_per_s'))"
