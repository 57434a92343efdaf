"""Steady-state launch timing shared by tools/blas_compare.py and tools/nn_ab.py."""
import torch


def timeit(fn, warm=100, reps=3, n=50):
    """steady state under the kernel's OWN load: the chip manages its clock within milliseconds, so a 20-launch measurement inherits the
    power state the previous kernel left (the same mts kernel read 257 us behind other mts kernels and 272 behind the vendor's in round 4's
    first collection).  `warm` untimed launches (>= 20 ms), then the median of `reps` x `n` timed ones."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]
