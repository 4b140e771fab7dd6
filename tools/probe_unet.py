"""U-Net (caller-side PyTorch module of the event term) fwd + input-gradient bwd at the Replica event resolution under a few
PyTorch settings -- guidance for callers, not part of the path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evennicer_slam_amd as E
dev = torch.device('cuda', 0)
torch.manual_seed(0)
def run(tag, channels_last=False, bench=False, dtype=None):
    torch.backends.cudnn.benchmark = bench
    net = E.event.UNet_2heads(6, 2, 2)
    for p in net.parameters(): p.requires_grad_(False)
    net = net.to(dev).eval()
    if channels_last: net = net.to(memory_format=torch.channels_last)
    x0 = torch.rand(1, 6, 102, 180, device=dev)
    if channels_last: x0 = x0.contiguous(memory_format=torch.channels_last)
    def it():
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=dtype, enabled=dtype is not None):
            e, m = net(x)
        (e.float().sum() + m.float().sum()).backward()
    for _ in range(5): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): it()
    torch.cuda.synchronize()
    print(f"{tag:40s} {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms")
run("fp32 NCHW")
run("fp32 NCHW, cudnn.benchmark", bench=True)
run("fp32 channels_last", channels_last=True)
run("fp32 channels_last, cudnn.benchmark", channels_last=True, bench=True)
run("bf16 autocast NCHW, benchmark", bench=True, dtype=torch.bfloat16)
run("bf16 autocast channels_last, benchmark", channels_last=True, bench=True, dtype=torch.bfloat16)
