import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
def bench(M,N,K,ta=False,tb=True):
    A=torch.randn(M,K,device='cuda'); B=torch.randn(N,K,device='cuda') if tb else torch.randn(K,N,device='cuda')
    f=(lambda: A@B.t()) if tb else (lambda: A@B)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*100
    print(f"torch fp32 mm M={M} N={N} K={K} tb={tb}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF")
bench(51200,1024,256); bench(51200,256,1024,tb=False); bench(8192,1024,256); bench(4096,4096,4096)
A=torch.randn(51200,1024,device='cuda'); X=torch.randn(51200,256,device='cuda')
for _ in range(3): A.t()@X
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): A.t()@X
e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*100
print(f"torch fp32 dW (1024x256, K=51200): {us:8.1f} us {2*51200*1024*256/us/1e6:6.1f} TF")
