#!/bin/bash
# copies what tools/gpu/r06_final.sh left under gpurun_out/r06final into the tracked profiles/r06_* files DESIGN.md cites
cd "$(dirname "$0")/../.."
S=gpurun_out/r06final
{ cat $S/pytest.log; cat $S/smoke.log; } > profiles/r06_pytest_gpu.txt
cp $S/bench_4mm.json profiles/r06_bench_4mm.json
cp $S/detail_4mm.json profiles/r06_bench_4mm_detail.json
cp $S/bench_4mm_driver_args.json profiles/r06_bench_4mm_driver_args.json
cp $S/bench_1mm.json profiles/r06_bench_1mm.json
cp $S/bench_scans_gpu.json profiles/r06_bench_scans_gpu.json
cp $S/bench_partition.json profiles/r06_bench_partition.json
cp $S/bench_two_ranks_one_gpu.json profiles/r06_bench_two_ranks_one_gpu.json
cp $S/kt.txt profiles/r06_rocprofv3_kernel_stats.txt
cp $S/timeline.txt profiles/r06_timeline_default_command.txt
cp $S/kt_driver.txt profiles/r06_rocprofv3_kernel_stats_driver_command.txt
cp $S/kt_e2e_rgbd.txt profiles/r06_rocprofv3_kernel_stats_e2e_rgbd.txt
cp $S/timeline_e2e_rgbd.txt profiles/r06_timeline_e2e_rgbd.txt
cp $S/kernel_resources.txt profiles/r06_kernel_resources.txt
ls -la profiles/r06_bench_4mm.json profiles/r06_pytest_gpu.txt
