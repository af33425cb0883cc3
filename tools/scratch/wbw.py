import torch, time
x = torch.empty(1280*1024*1024//4, dtype=torch.int32, device="cuda")
for _ in range(3): x.fill_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x.fill_(2)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1)/10
print("fill 1.34GB ms", t, "TB/s", x.numel()*4/t/1e9)
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1)/10
print("copy ms", t, "TB/s (r+w)", 2*x.numel()*4/t/1e9)
