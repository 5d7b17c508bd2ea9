import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
for rows, cols in [(32768, 1024), (16448, 768), (32768, 4096)]:
    v, dy, dpre = (torch.randn(rows, cols, device="cuda") for _ in range(3))
    w = torch.rand(cols, device="cuda") + 0.5
    for mode in (0, 1):
        fn = lambda: ops.norm_res_bwd(dy, v, w, 1e-6, mode, dpre=dpre, also_bf16=True)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"norm_res_bwd rows {rows} cols {cols} mode {mode}: {us:.0f} us incl. colsum, {rows * cols * 22 / us / 1e6:.2f} TB/s", flush=True)
