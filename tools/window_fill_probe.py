"""Measurement (round 6): how fast one block of device memory fills as a function of HOW MUCH of it is written at a time - is the
address space spread over the memory stacks finely (a small window reaches the full rate) or coarsely (it reaches a stack's)?

    python tools/window_fill_probe.py [GB of the block]"""
import sys
import torch

gb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
x.zero_()
torch.cuda.synchronize()


def rate(lo, n, reps=5):
    v = x[lo:lo + n]
    v.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        v.zero_()
    b.record()
    torch.cuda.synchronize()
    return n * reps / (a.elapsed_time(b) * 1e-3) / 1e12


for mb in (512, 1024, 2048, 4096, 8192, 16384):
    n = mb << 20
    if n > x.numel():
        break
    offs = [(k * (x.numel() - n) // 7) >> 21 << 21 for k in range(8)] if n < x.numel() else [0]      # (2 MB boundaries)
    print("window %6d MB: %s TB/s" % (mb, " ".join("%.2f" % rate(o, n) for o in offs)), flush=True)
