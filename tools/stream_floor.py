"""Streaming floor at the config-2 footprint: how fast can ANY kernel move 117 MB in + 17 MB out when the
working set (134 MB) is Infinity-Cache resident, vs a 1 GB footprint (HBM)."""
import torch, time
dev = torch.device("cuda:0")
def bench(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb_in, mb_out in ((117.4, 16.8), (940, 134)):
    n_in = int(mb_in * 1e6 / 8); n_out = int(mb_out * 1e6 / 8)
    x = torch.rand(n_in, dtype=torch.float64, device=dev)
    y = torch.empty(n_out, dtype=torch.float64, device=dev)
    # read-heavy: sum-reduce 7 chunks into one output chunk (reads n_in, writes n_out)
    xs = x[: (n_in // n_out) * n_out].view(-1, n_out)
    t = bench(lambda: torch.sum(xs, dim=0, out=y), 100 if mb_in > 500 else 300)
    print("reduce %6.1f MB in -> %5.1f MB out: %7.2f us  %.2f TB/s" % (mb_in, mb_out, t, (xs.numel() + n_out) * 8 / t / 1e6))
    z = torch.empty_like(x[: n_in // 2])
    t = bench(lambda: z.copy_(x[: n_in // 2]), 100 if mb_in > 500 else 300)
    print("copy   %6.1f MB -> %6.1f MB        : %7.2f us  %.2f TB/s" % (mb_in / 2, mb_in / 2, t, n_in // 2 * 16 / t / 1e6))
