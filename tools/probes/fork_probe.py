#!/usr/bin/env python3
"""Cost of a fork/join inside a captured hipGraph: a chain of small dependent launches beside ONE long HBM-bound launch.
Prints replay times (us, median of 30) of: chain alone, big alone, both serial on one stream, big forked beside the chain."""
import statistics, time
import torch

dev = torch.device("cuda")
small = [torch.zeros(1 << 14, device=dev) for _ in range(4)]
big = torch.ones(96 << 20, device=dev)          # 384 MB read + written per pass: ~130 us
N = 60


def chain():
    for i in range(N):
        small[i % 4].add_(1.0)


def big_pass():
    big.mul_(1.0000001)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g


side = torch.cuda.Stream()


def forked(join_after):
    def fn():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            big_pass()
        for i in range(N):
            if i == join_after:
                main.wait_stream(side)
            small[i % 4].add_(1.0)
        if join_after >= N:
            main.wait_stream(side)
    return fn


def serial():
    big_pass()
    chain()


def timeit(g, reps=30):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(1e3 * a.elapsed_time(b))
    return statistics.median(out)


for name, fn in (("chain alone", chain), ("big alone", big_pass), ("serial big+chain", serial),
                 ("forked, join at end", forked(N)), ("forked, join after 30", forked(30)),
                 ("forked, join after 5", forked(5))):
    g = capture(fn)
    print(f"{name:28s} {timeit(g):9.1f} us")
