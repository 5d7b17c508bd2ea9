"""What a plain streaming kernel reaches on this box: torch copy / add / fill of large f32 tensors (HIP-event timing)"""
import torch
dev = "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (64, 256, 1024, 4096):
    n = mb * 2**20 // 4
    a, b, c = torch.randn(n, device=dev), torch.empty(n, device=dev), torch.randn(n, device=dev)
    tc = timeit(lambda: b.copy_(a)); ta = timeit(lambda: torch.add(a, c, out=b)); tf = timeit(lambda: b.fill_(1.0)); tr = timeit(lambda: a.sum())
    print(f"{mb:5d} MiB  copy {2*n*4/tc/1e9:6.2f} TB/s  add {3*n*4/ta/1e9:6.2f} TB/s  fill {n*4/tf/1e9:6.2f} TB/s  sum(read) {n*4/tr/1e9:6.2f} TB/s")
