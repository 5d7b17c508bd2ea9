#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-run scripts of earlier rounds).  Usage on the box, from the repo root:
#   scripts/gpu.sh <stage> [args]        stages: attn | tests | bench | final | prof
# Everything a stage prints that should survive goes to gpurun_out/<tag>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
stage=$1; tag=${2:-r6}
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
ab)     # same-box A/B of the headline step: scripts/gpu.sh ab <tag> "ENV=a ENV2=b" "ENV=c" ...   (one bench.py --no-extra run per quoted setting)
  for setting in "${@:3}"; do
    for rep in 1 2; do
      env $setting timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 2> $O/${tag}_ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d.get('extra', {})
        print('%-40s rep $rep  %8.1f img/s  %7.2f ms  transformer fwd+bwd %s ms  frac %s' % ('$setting', d['value'], d['ms_per_step'], e.get('transformer_fwd_bwd_ms'), e.get('transformer_mfma_frac')))
" | tee -a $O/${tag}_ab.txt
    done
  done
  ;;
trace)  # concurrency picture of the default multi-stream step: kernel trace of a short bench run -> scripts/overlap_report.py
  rm -rf $O/${tag}_trace; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/${tag}_trace -o t -- python $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OLDPWD/$O/${tag}_trace_bench.txt 2>&1
  cd $OLDPWD
  f=$(find $O/${tag}_trace -name "t_kernel_trace.csv" | head -1)
  python scripts/overlap_report.py $f --last-ms 250 > $O/${tag}_overlap.txt 2>&1; tail -60 $O/${tag}_overlap.txt
  python scripts/step_trace_report.py $f --steps 5 --serial profiles/r05_final_kernel_stats.csv > $O/${tag}_concurrent_step.txt 2>&1
  python scripts/step_timeline.py $f --dump $O/${tag}_step_kernels.csv > $O/${tag}_step_timeline.txt 2>&1
  g=$(find $O/${tag}_trace -name "t_kernel_stats.csv" | head -1); cp $g $O/${tag}_kernel_stats.csv
  find $O/${tag}_trace -name "*.csv" -size +3M -delete
  ;;
g256p)  # persistent-GEMM epilogue modes (MUSE_G256P_EPI: 2 product, 3 no stores, 4 role-split ablation, 5 role split): timing, then check
  export MUSE_GEMM256=1
  run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py $MODE 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /"; }
  MODE=time
  { for m in ${@:3}; do run MUSE_G256P_EPI=$m; done; run MUSE_G256P_EPI=2; } > $O/${tag}_g256p_time.txt 2>&1
  grep -E "total|FFN-in \[T,768\]|QKV    \[T" $O/${tag}_g256p_time.txt | cut -c1-160
  ;;
g256p_check)
  export MUSE_GEMM256=1
  MUSE_G256P_EPI=${3:-5} timeout 600 python scripts/exp/g256p_probe.py check 2>&1 | grep -v amdgpu.ids > $O/${tag}_g256p_check.txt
  grep -c "rerun_identical True" $O/${tag}_g256p_check.txt; grep -E "False|e-0[01]|e\+0|Error|error" $O/${tag}_g256p_check.txt | cut -c1-200 | head -20
  ;;
attn_pmc)
  bash scripts/exp/attn_pmc.sh; cp $O/attn_pmc_summary.txt $O/${tag}_attn_pmc.txt
  ;;
final)  # validation of the committed tree: device, full GPU suite (printed error figures), smoke, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in
        # separate runs), MFMA-busy / clock counters over the GEMM / convolution probes, the default bench line with every leg, rocprof
        # of the serial bench, the driver's launch line at one rank.  scripts/gpu.sh final <tag>
  T=$tag
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/${T}_device.txt
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
  grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/${T}_pytest_gpu.txt | tail -8
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.txt 2>&1; echo "smoke exit $?" >> $O/${T}_smoke.txt; tail -2 $O/${T}_smoke.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c
    MUSE_CONV_PERSIST=0 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prefetch > $O/pmc_$c.log 2>&1
    echo "$c exit $?"
  done
  f=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && [ -n "$w" ] && python scripts/traffic_summary.py "$f" "$w" $O/${T}_traffic.json && cp $O/${T}_traffic.json profiles/r06_traffic.json
  find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +4M -delete
  rm -rf $O/pmc_mfma
  REPS=3 WHICH=nn,nt,grp,conv timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python scripts/gemm_probe.py > $O/pmc_mfma.log 2>&1
  python scripts/pmc_fold.py $O/pmc_mfma | tee $O/${T}_pmc_mfma_clock.txt | tail -14
  find $O/pmc_mfma -name "*.csv" -size +2M -delete
  timeout 1800 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench exit $?"; tail -c 400 $O/${T}_bench.err
  python scripts/bench_summary.py $O/${T}_bench.json
  rm -rf $O/prof_serial
  MUSE_WGRAD_STREAM=0 MUSE_CONV_PERSIST=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-prefetch > $O/${T}_prof.txt 2>&1
  f=$(find $O/prof_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv && head -14 "$f" | cut -c1-170
  find $O/prof_serial -name "*kernel_trace*" -size +8M -delete
  MUSE_BENCH_RCCL_DEBUG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>$O/${T}_dp1.err > $O/${T}_dp1.out
  python -c "
import json
ls=[l for l in open('$O/${T}_dp1.out') if l.strip()]
print('stdout lines', len(ls)); d=json.loads(ls[-1]); print('dp1', d['value'], d['ms_per_step']); open('$O/${T}_dp1_comm_block.json','w').write(json.dumps(d.get('comm'), indent=1)); print(json.dumps(d.get('comm'))[:1500])"
  ;;
x3)     # bf16x3 attention bring-up: kernel + mode tests to a file, then the config-4 leg with and without the fused core
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -rP -k "bf16x3 or three_bf16 or uvit or UViT" > $O/${tag}_x3_pytest.txt 2>&1; echo "pytest exit $?" >> $O/${tag}_x3_pytest.txt
  grep -E "passed|failed|pytest exit|^FAILED|^ERROR|bf16x3 attention|operand images" $O/${tag}_x3_pytest.txt | tail -30
  for v in ${X3_VARS:-MUSE_X3_ATTENTION=0 MUSE_X3_ATTENTION=1}; do env $v timeout 600 python bench.py --uvit-leg 64,256,2,x3 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/$v /" | tee -a $O/${tag}_x3_leg.txt; done
  ;;
c4prof) # config 4 (U-ViT) kernel profile, serial: scripts/gpu.sh c4prof <tag> [leg, default 64,256,2,x3]
  leg=${3:-64,256,2,x3}
  rm -rf $O/${tag}_c4; cd /tmp
  MUSE_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/${tag}_c4 -o c -- python $OLDPWD/bench.py --uvit-leg $leg > $OLDPWD/$O/${tag}_c4_leg.txt 2>&1
  cd $OLDPWD; tail -1 $O/${tag}_c4_leg.txt | cut -c1-300
  f=$(find $O/${tag}_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_c4_kernel_stats.csv && head -40 "$f" | cut -c1-150
  find $O/${tag}_c4 -name "*.csv" -size +3M -delete
  timeout 600 python bench.py --uvit-leg $leg 2>/dev/null | tail -1 | cut -c1-200
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
