#!/bin/bash
# round-4 validation of the committed tree: full GPU suite (with the printed error figures), smoke, PMC traffic passes, MFMA-busy /
# clock counters, the default bench line with every leg, rocprof of the serial bench, of the config-5 and config-4 legs, and the
# driver's launch line at one rank with RCCL's INFO lines captured
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r4_mid}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/${T}_device.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/${T}_pytest_gpu.txt | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.txt 2>&1; echo "smoke exit $?" >> $O/${T}_smoke.txt; tail -2 $O/${T}_smoke.txt
bash scripts/gpu_traffic_bench.sh 2>&1 | tail -4
cp $O/r04_traffic.json profiles/r04_traffic.json 2>/dev/null
rm -rf $O/pmc_r4
REPS=3 WHICH=nn,nt,grp,conv timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r4 -o p -- python scripts/gemm_probe.py > $O/pmc_r4.log 2>&1
sed -e 's#gpurun_out/pmc_r3#gpurun_out/pmc_r4#g' scripts/gpu_r3_pmc.sh | sed -n '/^python - <<.PY./,/^PY$/p' | sed '1d;$d' > /tmp/pmc_fold.py
python /tmp/pmc_fold.py | tee $O/${T}_pmc_mfma_clock.txt | tail -14
grep -E "TFLOP" $O/pmc_r4.log | head -12
find $O/pmc_r4 -name "*.csv" -size +2M -delete
timeout 1800 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench exit $?"; tail -c 600 $O/${T}_bench.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${T}_bench.json") if l.startswith("{")][-1])
    e = d["extra"]
    print("value", d["value"], "ms", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "tr_frac", e.get("transformer_mfma_frac"), e.get("transformer_fwd_bwd_ms"))
    print({k: v for k, v in e.items() if k.startswith("images_per_s") or k.startswith("vqgan_encode_decode_images") or k.startswith("taming")})
    print("config4", e.get("config4_uvit_seq256"), e.get("config4_uvit_seq1024"))
    print("config4 f32 / x3", e.get("config4_uvit_seq256_f32"), e.get("config4_uvit_seq256_bf16x3"))
    print("latency", e.get("inference_latency"))
    print("parity", e.get("measured_parity_bf16_vs_f32_mode"))
    print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind", "cpu_model", "vq_index_mismatches", "vq_index_mismatches_bench_batch")})
    print("per_kernel", {k: v["tflops"] for k, v in d["roofline"]["per_kernel"].items()})
except Exception as ex:
    print("no bench line:", ex)
PY
rm -rf $O/prof_r4
MUSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4 -o r4 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-prefetch > $O/${T}_prof.txt 2>&1
f=$(find $O/prof_r4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_c5b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5b -o c5 -- python bench.py --leg vqgan,64 > $O/${T}_c5_prof.txt 2>&1
f=$(find $O/prof_c5b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_config5_kernel_stats.csv && head -8 "$f" | cut -c1-170
rm -rf $O/prof_c4
MUSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --uvit-leg 128,256,3 > $O/${T}_c4_prof.txt 2>&1
f=$(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_config4_kernel_stats_serial.csv && head -10 "$f" | cut -c1-170
find $O/prof_r4 $O/prof_c5b $O/prof_c4 -name "*kernel_trace*" -size +8M -delete
MUSE_BENCH_RCCL_DEBUG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>$O/${T}_dp1.err > $O/${T}_dp1.out
python -c "
import json
ls=[l for l in open('$O/${T}_dp1.out') if l.strip()]
print('stdout lines', len(ls)); d=json.loads(ls[-1]); print('dp1', d['value'], d['ms_per_step']); print(json.dumps(d.get('comm'))[:2500])"
