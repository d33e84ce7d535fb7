#!/bin/bash
# the N > 1 control flow of bench.py on a one-GPU box: two ranks share GPU 0 over gloo (--share-gpu); rates mean nothing, completion and the line do
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 5 --share-gpu > $O/stream_n2.json 2> $O/stream_n2.err; echo "stream rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --config partition --scan-frames 3000 --share-gpu > $O/partition_n2.json 2> $O/partition_n2.err; echo "partition rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config scans --steps 2 --host-stage gpu --max-scan-frames 600 --share-gpu > $O/scans_n2.json 2> $O/scans_n2.err; echo "scans rc=$?"
python - <<PY
import json
for f in ("stream_n2","partition_n2","scans_n2"):
    try:
        j=json.loads([l for l in open("$O/%s.json"%f) if l.startswith("{")][0]); print(f, j["value"], j["unit"], "n_gpus", j["n_gpus"], j.get("repeats",{}).get("n"), (j.get("roofline") or {}).get("avg_kernel_us"), j.get("exchange"))
    except Exception as e: print(f,"ERR",e, open("$O/%s.err"%f).read()[-600:])
PY
