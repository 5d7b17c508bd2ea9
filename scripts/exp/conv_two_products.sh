#!/bin/bash
# would TWO bf16 products per f32 product (one cross term dropped) keep the VQ token indices?  Builds the patch-slab
# kernel with -DCDMA_TWO_PRODUCTS=1 (w_lo dropped: weights rounded to bf16) and =2 (x_lo dropped: activations rounded to bf16) and counts
# index mismatches of the f16-256 tokenizer against the f32 oracle over 16 images (4096 tokens); the 3-product build gives 0.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
C=open-muse_amd/csrc; L=/tmp/muse_2p; mkdir -p $L
make -C $C > /dev/null
cat > $L/count.py <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, time
import muse, weights as W
from oracle import maskgit_oracle as O
cfg = W.VQGAN_F16
sd = W.fill_state_dict(W.vqgan_shapes(cfg), 600, "vqgan")
B = 16
px = W.images(B, 256, 611)
torch.set_num_threads(32)
with torch.no_grad():
    idx_o = torch.cat([O.vqgan_encode(sd, cfg, px[i:i + 8])[2] for i in range(0, B, 8)])
v = muse.MaskGitVQGAN(**cfg); v.load_state_dict(sd); v.to("cuda").eval(); v.set_compute_dtype("bf16x3")
idx = v.get_code(px.cuda()).cpu()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): v.get_code(px.cuda())
torch.cuda.synchronize()
print(f"{os.environ.get('VARIANT')}: {int((idx != idx_o).sum())} of {idx.numel()} token indices differ from the f32 oracle; encode {B * 3 / (time.perf_counter() - t0):.0f} img/s")
PY
VARIANT="3 products (product build)" python $L/count.py 2>&1 | tail -1
for v in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCDMA_TWO_PRODUCTS=$v -c $C/conv_dma.hip -o $L/conv_dma_2p.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmuse_2p.so $C/gemm.o $C/rowops.o $C/vqgan.o $C/attention.o $C/conv_split.o $L/conv_dma_2p.o $C/uvit.o $C/sampling.o $C/embed.o
  VARIANT="2 products, variant $v" MUSE_HIP_LIB=$L/libmuse_2p.so python $L/count.py 2>&1 | tail -1
done
