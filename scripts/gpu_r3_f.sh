#!/bin/bash
# round 3, run F: why is the one-rank data-parallel launch line 12 % slower with 8 hardware queues?  kernel traces of both settings
# through scripts/overlap_report.py (stream -> queue placement, kernels in flight, per-family time) + per-kernel totals
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for q in 4 8; do
  rm -rf $O/prof_dp1_q$q
  GPU_MAX_HW_QUEUES=$q timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dp1_q$q -o t -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $O/r3f_dp1_q$q.txt 2>&1
  grep -o '"ms_per_step": [0-9.]*' $O/r3f_dp1_q$q.txt | tail -1
  f=$(find $O/prof_dp1_q$q -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/overlap_report.py "$f" --last-ms 250 > $O/r3f_overlap_q$q.txt 2>&1 && head -60 $O/r3f_overlap_q$q.txt | cut -c1-170
  s=$(find $O/prof_dp1_q$q -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $O/r3f_kernel_stats_q$q.csv && head -12 "$s" | cut -c1-150
  find $O/prof_dp1_q$q -name "*kernel_trace*" -delete
done
