import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
dev = "cuda"
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M = int(os.environ.get("M", "16448"))
tot_m = tot_t = 0.0
for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("w01", 6144, 768), ("wo2", 768, 3072),
                   ("d_qkv", 768, 2304), ("d_w01", 768, 6144), ("d_wo2", 3072, 768), ("logits", 2048, 768)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a = t(lambda: ops.linear(x, w, out=y)); b = t(lambda: torch.matmul(x, w.t()))
    fl = 2.0 * M * N * K
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    tot_m += a; tot_t += b
    print(f"{name:7s} M={M} N={N:5d} K={K:5d} tiles={tiles:5d} ({tiles/768:.2f} rounds)  muse {a:7.1f} us {fl/a/1e6:6.1f} TF | hipBLASLt {b:7.1f} us {fl/b/1e6:6.1f} TF", flush=True)
print(f"sum: muse {tot_m:.0f} us, hipBLASLt {tot_t:.0f} us")
