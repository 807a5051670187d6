import sys, faulthandler
faulthandler.enable()
import torch
dev = torch.device("cuda:0")
variant = sys.argv[1]
n = int(sys.argv[2])
x = torch.randn(512, 512, device=dev)
h = torch.cuda.Stream()
h2 = torch.cuda.Stream()
EVS = [torch.cuda.Event() for _ in range(64)]


def body():
    cur = torch.cuda.current_stream()
    y = x
    if variant.startswith("prejoin"):
        h.wait_stream(cur); h2.wait_stream(cur)
    for i in range(n):
        if variant == "forkjoin":
            h.wait_stream(cur)
            with torch.cuda.stream(h):
                b = y * 2
            a = y + 1
            cur.wait_stream(h)
            y = a + b
        elif variant in ("nested", "prejoin"):       # fork-join inside a second stream that itself forks from cur
            h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                c2 = torch.cuda.current_stream()
                h.wait_stream(c2)
                with torch.cuda.stream(h):
                    b = y * 2
                a = y + 1
                c2.wait_stream(h)
                z = a + b
            cur.wait_stream(h2)
            y = z * 0.5
        elif variant == "cross":        # two first-level forks, one waits on the other's event
            h.wait_stream(cur); h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
            h.wait_stream(h2)
            with torch.cuda.stream(h):
                b = a * 2
            cur.wait_stream(h); cur.wait_stream(h2)
            y = b
        elif variant == "cross_ev":     # same with explicit events
            h.wait_stream(cur); h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
                e = torch.cuda.Event(); e.record(h2)
            h.wait_event(e)
            with torch.cuda.stream(h):
                b = a * 2
            cur.wait_stream(h); cur.wait_stream(h2)
            y = b
        elif variant == "nested_noback":
            h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
                h.wait_stream(h2)
                with torch.cuda.stream(h):
                    b = a * 2
            cur.wait_stream(h); cur.wait_stream(h2)
            y = b
        elif variant == "bidir":
            h.wait_stream(cur); h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
            h.wait_stream(h2)
            with torch.cuda.stream(h):
                b = a * 2
            h2.wait_stream(h)
            with torch.cuda.stream(h2):
                c = b + 1
            cur.wait_stream(h); cur.wait_stream(h2)
            y = c
        elif variant == "bidir_noprefork":   # h enters the capture through h2's event only
            h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
            h.wait_stream(h2)
            with torch.cuda.stream(h):
                b = a * 2
            h2.wait_stream(h)
            with torch.cuda.stream(h2):
                c = b + 1
            cur.wait_stream(h2)
            y = c
        elif variant == "bidir_ev":      # bidirectional between two non-origin streams with long-lived events
            h.wait_stream(cur); h2.wait_stream(cur)
            with torch.cuda.stream(h2):
                a = y + 1
            e1 = EVS[2 * i]; e1.record(h2); h.wait_event(e1)
            with torch.cuda.stream(h):
                b = a * 2
            e2 = EVS[2 * i + 1]; e2.record(h); h2.wait_event(e2)
            with torch.cuda.stream(h2):
                c = b + 1
            cur.wait_stream(h); cur.wait_stream(h2)
            y = c
        elif variant == "emptyjoin":    # helper joins without having launched anything
            h.wait_stream(cur)
            a = y + 1
            cur.wait_stream(h)
            y = a
    if variant.startswith("prejoin"):
        cur.wait_stream(h); cur.wait_stream(h2)
    return y


body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    out = body()
print("captured", variant, n, flush=True)
g.replay(); torch.cuda.synchronize()
print("ok", float(out.sum()))
