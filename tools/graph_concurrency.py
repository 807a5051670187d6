"""Does a replayed hipGraph run independent branches (captured on two streams) concurrently on this stack?"""
import time
import torch

dev = torch.device("cuda:0")
a = [torch.randn(256, 256, device=dev) for _ in range(2)]
w = torch.randn(256, 256, device=dev) * 0.05


def chain(x, n=200):
    for _ in range(n):
        x = torch.tanh(x @ w)
    return x


def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / it * 1e3


def one_stream():
    return chain(a[0]), chain(a[1])


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        r1 = chain(a[0])
    with torch.cuda.stream(s2):
        r2 = chain(a[1])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return r1, r2


print("eager 1 stream  %.3f ms" % timeit(one_stream))
print("eager 2 streams %.3f ms" % timeit(two_streams))
for name, fn in (("1 stream", one_stream), ("2 streams", two_streams)):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    print("graph %s %.3f ms" % (name, timeit(g.replay)))
