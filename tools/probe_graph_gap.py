"""Is the ~8 us between two replays of one hipGraph a property of re-launching the SAME executable graph?  Two graphs of identical
work on different buffers: replaying one of them back to back vs alternating the two."""
import time, torch
dev = torch.device('cuda', 0)
def make(n=1 << 24):
    x = torch.rand(n, device=dev); y = torch.empty_like(x)
    def f():
        for _ in range(4):
            torch.mul(x, 1.0001, out=y); torch.add(y, 0.5, out=x)
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f()
    return g
g1, g2 = make(), make()
def run(seq, n=300):
    for g in seq * 5: g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for g in seq: g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n * len(seq)) * 1e6
for _ in range(2):
    print(f"same graph back to back: {run([g1]):.1f} us per replay; alternating two graphs: {run([g1, g2]):.1f} us per replay", flush=True)
