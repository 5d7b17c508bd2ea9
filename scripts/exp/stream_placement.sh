#!/bin/bash
# builds the CU-mask helper and sweeps stream placements of the default train step (scripts/exp/stream_placement.py); ~20 s per variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libstream_placement.so scripts/exp/stream_placement.hip || exit 1
for q in 4 8 2; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python scripts/exp/stream_placement.py base order:psh order:hps order:shp prio:-1,0 prio:0,-1 2>&1 | grep -v Warning | tee -a $O/stream_placement.txt
done
timeout 600 python scripts/exp/stream_placement.py base mask:64x:p mask:64c:p mask:32x:p mask:128x:p mask:64x:s 2>&1 | grep -v Warning | tee -a $O/stream_placement.txt
# what the default step does with its streams: kernel trace of a short default bench -> scripts/overlap_report.py
export TMPDIR=/tmp; rm -rf $O/prof_overlap
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_overlap -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/overlap_bench.txt 2>&1
f=$(find $O/prof_overlap -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/overlap_report.py "$f" --last-ms 300 | tee $O/overlap_report.txt
find $O/prof_overlap -name "*kernel_trace*" -size +8M -delete
