"""Do the two branches of a captured fork/join run CONCURRENTLY when a hipGraph is replayed on this stack?
Two independent chains of small launches (each far too small to fill the GPU) on two streams: eager on one stream, eager on two streams,
captured as one graph with two branches.  If a replay takes as long as the one-stream eager run, graph branches are serialised and the
decode loop's split chains (generate.py, NS_DECODE_SPLIT) cannot overlap inside a graph."""
import time

import torch

dev = torch.device("cuda:0")
N, CH = 256, 300
a = [torch.randn(N, N, device=dev) * 0.05 for _ in range(2)]
w = torch.randn(N, N, device=dev) * 0.05
side = torch.cuda.Stream(dev)


def chain(x):
    for _ in range(CH):
        x = torch.tanh(x @ w)
    return x


def one_stream():
    chain(a[0]); chain(a[1])


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain(a[1])
    chain(a[0])
    main.wait_stream(side)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"eager, one stream : {timed(one_stream):.2f} ms")
print(f"eager, two streams: {timed(two_streams):.2f} ms")
for name, fn in (("one stream", one_stream), ("two branches", two_streams)):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    print(f"graph, {name:12s}: {timed(g.replay):.2f} ms")
