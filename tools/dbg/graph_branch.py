"""Do parallel branches of a captured HIP graph overlap on this runtime?  main: 6 streaming kernels; tiny: 3 one-block ops."""
import torch, time
dev = torch.device("cuda:0")
x = torch.zeros(64 << 20, device=dev)        # 256 MB: ~70 us per pass
ys = [torch.zeros(64, device=dev) for _ in range(3)]

def build(branch):
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        cur = torch.cuda.current_stream()
        if branch:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for y in ys:
                    y.add_(1.0)
        for i in range(6):
            x.mul_(1.0001)
        if branch:
            cur.wait_stream(side)
        else:
            for y in ys:
                y.add_(1.0)
    return g

def timeit(g, n=200):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for _ in range(3):
    x.mul_(1.0001)
torch.cuda.synchronize()
ga, gb = build(False), build(True)
for r in range(3):
    print("chain %.1f us   branch %.1f us" % (timeit(ga), timeit(gb)))
