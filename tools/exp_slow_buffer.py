#!/usr/bin/env python3
"""After 30 GB went through torch's allocator: H2D rate into each freshly allocated 105 MB device buffer (GPU box)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
h = torch.empty(16 * 1024 * 1024, dtype=torch.float32).pin_memory()
def rate(d):
    d[: h.numel()].copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): d[: h.numel()].copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    return 6 * h.numel() * 4 / (time.perf_counter() - t0) / 1e9
def round_(tag):
    bufs = [torch.empty(1024 * 160 * 160, dtype=torch.float32, device=dev) for _ in range(6)]
    print(tag, " ".join(f"{rate(b):.1f}@{b.data_ptr():#x}" for b in bufs), flush=True)
    return bufs
keep = round_("fresh process:")
del keep
if os.environ.get("EMPTY", "1") == "1": torch.cuda.empty_cache()
big = torch.empty(30 * 1024**3 // 4, device=dev); del big; torch.cuda.empty_cache()
keep = round_("after 30 GB + empty_cache:")
del keep; torch.cuda.empty_cache()
keep = round_("again after empty_cache:")
