#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-run scripts of earlier rounds).  Usage on the box, from the repo root:
#   scripts/gpu.sh <stage> [args]        stages: attn | tests | bench | final | prof
# Everything a stage prints that should survive goes to gpurun_out/<tag>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
stage=$1; tag=${2:-r5}
case $stage in
attn)   # attention2.hip bring-up: parity tests, then old vs new (and build variants) on the config-B shape
  timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "attention" > $O/${tag}_attn_pytest.txt 2>&1
  echo "pytest exit $?" >> $O/${tag}_attn_pytest.txt; tail -15 $O/${tag}_attn_pytest.txt
  : > $O/${tag}_attn_bench.txt
  MUSE_ATTN2=0 timeout 300 python scripts/attn_bench.py 20 0 2>&1 | sed 's/^/old      | /' | tee -a $O/${tag}_attn_bench.txt
  timeout 300 python scripts/attn_bench.py 20 0 2>&1 | sed 's/^/new      | /' | tee -a $O/${tag}_attn_bench.txt
  for v in open-muse_amd/csrc/variants/libmuse_hip_*.so; do
    n=$(basename $v .so); n=${n#libmuse_hip_}
    MUSE_HIP_LIB=$PWD/$v timeout 300 python scripts/attn_bench.py 20 0 2>&1 | sed "s/^/$(printf '%-8s' $n) | /" | tee -a $O/${tag}_attn_bench.txt
  done
  timeout 300 python scripts/attn_bench.py 20 0 2>&1 | sed 's/^/new again| /' | tee -a $O/${tag}_attn_bench.txt
  ;;
attn_stag)  # start stagger of the second workgroup per CU: sweep
  for f in 0 6000 12000 18000 24000; do for b in 0 20000 40000; do
    MUSE_ATTN2_STAGGER_FWD=$f MUSE_ATTN2_STAGGER_BWD=$b timeout 300 python scripts/attn_bench.py 20 0 2>&1 | grep config | sed "s/^/stagger fwd $f bwd $b | /" | tee -a $O/${tag}_attn_stagger.txt
  done; done
  ;;
attn_ts)
  MUSE_HIP_LIB=$PWD/open-muse_amd/csrc/variants/libmuse_hip_ts.so timeout 300 python scripts/exp/attn2_ts.py 2>&1 | grep -v amdgpu.ids | tee $O/${tag}_attn_ts.txt
  ;;
tests)
  timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${tag}_pytest.txt 2>&1; echo "pytest exit $?" >> $O/${tag}_pytest.txt
  grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/${tag}_pytest.txt | tail -5
  ;;
bench)
  timeout 1800 python bench.py ${@:3} > $O/${tag}_bench.json 2> $O/${tag}_bench.err; echo "bench exit $?"
  tail -c 1500 $O/${tag}_bench.json
  ;;
*) echo "unknown stage $stage"; exit 2;;
esac
