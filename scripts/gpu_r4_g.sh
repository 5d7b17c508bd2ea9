#!/bin/bash
# round 4, run G: where the decoder's 16 ms at batch 1 go (rocprof), latency leg with the default switches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rm -rf $O/prof_td
BS=1 REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_td -o td -- python scripts/exp/taming_decode_profile.py > $O/r4g_td_prof.txt 2>&1
tail -1 $O/r4g_td_prof.txt
f=$(find $O/prof_td -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r4g_taming_decode_bs1_kernel_stats.csv && head -14 "$f" | cut -c1-160
BS=1 python scripts/exp/taming_decode_profile.py | tail -1
BS=8 python scripts/exp/taming_decode_profile.py | tail -1
python bench.py --leg latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'ms' in k and 'ref' not in k})"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "small_batch" 2>&1 | tail -2
