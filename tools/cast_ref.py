"""What a plain u8 -> f32 streaming pass of 256^3 costs on this GPU (the GMM draw moves the same bytes)."""
import torch, time
x = torch.randint(0, 50, (256, 256, 256), dtype=torch.uint8, device="cuda")
y = torch.empty((256, 256, 256), dtype=torch.float32, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("u8 -> f32 copy_ (16.7 MB read, 67 MB written): %.1f us" % t(lambda: y.copy_(x)))
z = torch.empty_like(y)
print("f32 -> f32 copy_ (67 + 67 MB): %.1f us" % t(lambda: z.copy_(y)))
print("f32 fill (67 MB written): %.1f us" % t(lambda: z.fill_(1.0)))
