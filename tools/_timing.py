"""Shared by the probes: best-of-3 mean launch time (us) of a plan, HIP events on torch's current stream."""
import torch


def timeit(p, B, C, n, rounds=3, warm=3):
    s = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for _ in range(rounds):
        for _ in range(warm):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            p.spmm(B.data_ptr(), C.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
