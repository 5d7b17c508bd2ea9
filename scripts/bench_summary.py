"""One-screen summary of a bench.py JSON line:  python scripts/bench_summary.py <file holding the line>"""
import json
import sys

try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e = d.get("extra", {})
    r = d["roofline"]
    print("value", d["value"], d["unit"], "ms", d["ms_per_step"], "| roofline", r["bound"], "achieved", r["achieved"], "frac", r["frac"], "traffic", r.get("traffic"))
    print("transformer fwd+bwd ms", e.get("transformer_fwd_bwd_ms"), "mfma frac", e.get("transformer_mfma_frac"), "| vqgan hbm frac", e.get("vqgan_hbm_frac"))
    print({k: v for k, v in e.items() if k.startswith("images_per_s") or k.startswith("vqgan_encode_decode_images") or k.startswith("taming")})
    for k in ("config4_uvit_seq256", "config4_uvit_seq1024", "config4_uvit_seq256_f32", "config4_uvit_seq256_bf16x3", "config4_uvit_seq256_bf16x3_b128"):
        v = e.get(k)
        print(k, None if v is None else {kk: v.get(kk) for kk in ("images_per_s", "ms_per_step", "mfma_frac")})
    print("latency", {k: v for k, v in (e.get("inference_latency") or {}).items() if ("ms" in k or "error" in k) and "ref" not in k})
    print("parity", e.get("measured_parity_bf16_vs_f32_mode"))
    print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind", "cpu_model", "vq_index_mismatches_bench_batch")})
    print("per_kernel", {k: v.get("tflops") for k, v in r.get("per_kernel", {}).items()})
    print("hbm_bound_kernels", r.get("hbm_bound_kernels"))
except Exception as ex:   # noqa: BLE001
    print("no bench line:", type(ex).__name__, ex)
