import torch, time
x = torch.empty(4096, 704, 720, dtype=torch.float64, device='cuda')
for _ in range(2): x.zero_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): x.zero_()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"zero_ 16.6GB: {ms:.3f} ms -> {x.numel()*8/ms/1e6:.1f} GB/s")
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize()
e0.record()
for _ in range(5): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"copy 16.6GB: {ms:.3f} ms -> {2*x.numel()*8/ms/1e6:.1f} GB/s (r+w)")
