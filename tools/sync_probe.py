"""Which of the small tensor idioms used on the host path block until the GPU queue drains?  (needs a GPU)"""
import time
import torch

dev = 'cuda:0'
a = torch.randn(8192, 8192, device=dev)


def busy():
    for _ in range(6):
        a @ a          # ~50 ms of queued GPU work


def probe(name, fn):
    fn()                      # first use loads the kernel's code object (slow, synchronising): not what is probed here
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%-48s %8.3f ms %s' % (name, dt * 1e3, 'BLOCKS' if dt > 5e-3 else ''))


w = torch.full((5,), 0.2, device=dev)
big = torch.zeros(80, 96, 30, 54, device=dev)
src = torch.randn(5, 96, 30, 54, device=dev)
probe('w[0] = python float', lambda: w.__setitem__(0, 0.4))
probe('w[:3] = tensor', lambda: big.__setitem__(slice(0, 5), src))
probe('w / w.sum()', lambda: w / w.sum())
probe('torch.full', lambda: torch.full((5,), 0.2, device=dev))
probe('torch.tensor([..], device)', lambda: torch.tensor([-1, -1], device=dev))
probe('torch.zeros(50MB)', lambda: torch.zeros(80, 96, 30, 54, device=dev))
probe('x.to(uint8)', lambda: src.to(torch.uint8))
probe('(x > 0).float()', lambda: (src > 0).float())
probe('torch.as_tensor(np)', lambda: torch.as_tensor(__import__('numpy').ones((5, 5), dtype='float32'), device=dev))
probe('x.clamp(0,255).floor()', lambda: src.clamp(0, 255).floor())
probe('torch.stack', lambda: torch.stack([src, src]))
probe('F.interpolate', lambda: torch.nn.functional.interpolate(src, size=(60, 108), mode='bilinear', align_corners=False))
probe('F.avg_pool2d', lambda: torch.nn.functional.avg_pool2d(src, 2, ceil_mode=True))
probe('torch.where', lambda: torch.where(src > 0, src, src * 2))
probe('nn.Conv2d(...).to(dev)', lambda: torch.nn.Conv2d(1024, 96, 1, bias=False).to(dev))
probe('torch.random.manual_seed', lambda: torch.random.manual_seed(0))
