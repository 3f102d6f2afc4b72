"""HBM microbench: fill (write-only), sum (read-only), copy (read+write) over 2 GiB, GB/s each."""
import torch, json
dev = "cuda:0"
n = 1 << 29                       # 2 GiB of f32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
res = {"fill_GBs": 4 * n / t(lambda: a.fill_(1.0)) / 1e9,
       "sum_GBs": 4 * n / t(lambda: a.sum()) / 1e9,
       "copy_GBs": 8 * n / t(lambda: b.copy_(a)) / 1e9}
print(json.dumps(res))
