#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Everything lands under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
STAGE=${1:-all}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt
nproc >> $O/device.txt; grep -m1 "model name" /proc/cpuinfo >> $O/device.txt
if [[ $STAGE == all || $STAGE == test ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
  echo "pytest exit $?" >> $O/pytest_gpu.txt
  tail -60 $O/pytest_gpu.txt
fi
if [[ $STAGE == all || $STAGE == smoke ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt
  tail -5 $O/smoke.txt
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 ${BENCH_ARGS} > $O/bench.txt 2>&1; echo "bench exit $?" >> $O/bench.txt
  tail -5 $O/bench.txt
fi
if [[ $STAGE == perf ]]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_kernels.txt 2>&1
  echo "pytest exit $?" >> $O/pytest_kernels.txt; tail -15 $O/pytest_kernels.txt
  for v in default; do
    if [[ $v == bm256 ]]; then export MUSE_BM256=1; else unset MUSE_BM256; fi
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_$v.txt 2>&1; echo "exit $?" >> $O/bench_$v.txt
    tail -2 $O/bench_$v.txt | cut -c1-2600
  done
  unset MUSE_BM256
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --vq-dtype bf16x3 > $O/bench_vqx3.txt 2>&1; tail -1 $O/bench_vqx3.txt | cut -c1-2600
  timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "vqgan" > $O/pytest_vq.txt 2>&1; tail -15 $O/pytest_vq.txt
fi
if [[ $STAGE == all || $STAGE == prof ]]; then
  rm -rf $O/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/prof.txt 2>&1
  echo "prof exit $?" >> $O/prof.txt
  find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep only the small summaries (the trace itself can be large)
  find $O/prof -name "*kernel_trace*" -size +8M -delete
fi
