"""the taming f16-8192 decoder at the batch sizes of the inference-latency leg (decode_code of 1 / 8 images of 256 tokens), repeated,
for `rocprofv3 --kernel-trace --stats`: kernel time against wall time per call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import muse
import weights as W

dev = "cuda"
bs = int(os.environ.get("BS", "1"))
reps = int(os.environ.get("REPS", "20"))
vcfg = dict(W.VQGAN_F16, num_embeddings=8192, attn_resolutions=(16,), no_attn_mid_block=False, resample_with_conv=True)
vae = muse.VQGANModel(**vcfg)
vae.load_state_dict(W.fill_state_dict(W.taming_shapes(vcfg), 4321, "vqgan"))
vae.to(dev).eval().set_compute_dtype("bf16x3")
toks = torch.randint(0, 8192, (bs, 256), device=dev)
with torch.no_grad():
    for _ in range(3):
        vae.decode_code(toks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vae.decode_code(toks)
    torch.cuda.synchronize()
print(f"taming decode_code at batch {bs}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
