"""Does a tensor that was JUST written still sit in the 256 MB Infinity Cache when the next kernel reads it?  Times `t2 = t1 * 2` (reads N bytes,
writes N bytes) right after `t1 = x * 2` wrote t1 ("warm") and after a 1.5 GB flush in between ("cold"), for N = 16 MB .. 1 GB."""
import torch
dev = "cuda"
flush = torch.empty(1536 * 2 ** 20 // 4, device=dev)
res = []
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * 2 ** 20 // 4
    x = torch.randn(n, device=dev); t1 = torch.empty_like(x); t2 = torch.empty_like(x)
    def run(cold, reps=20):
        tot = 0.0
        for _ in range(reps):
            torch.mul(x, 2.0, out=t1)
            if cold:
                flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.mul(t1, 2.0, out=t2); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps
    run(False, 3)
    w, c = run(False), run(True)
    # read-only consumer: sum
    def runsum(cold, reps=20):
        tot = 0.0
        for _ in range(reps):
            torch.mul(x, 2.0, out=t1)
            if cold:
                flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s = t1.sum(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps
    ws, cs = runsum(False), runsum(True)
    print(f"N={mb:5d} MB  mul warm {2*mb/1024/w*1e3:7.0f} GB/s  cold {2*mb/1024/c*1e3:7.0f} GB/s | sum warm {mb/1024/ws*1e3:7.0f} GB/s cold {mb/1024/cs*1e3:7.0f} GB/s", flush=True)
