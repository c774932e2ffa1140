import torch, time
dev=torch.device('cuda:0')
h2d_src=torch.empty(36_864_000,dtype=torch.uint8).pin_memory()
d2h_dst=torch.empty(73_728_000,dtype=torch.uint8).pin_memory()
d_in=torch.empty_like(h2d_src,device=dev); d_out=torch.empty(73_728_000,dtype=torch.uint8,device=dev)
s1=torch.cuda.Stream(); s2=torch.cuda.Stream()
def run(n,both=True,h=True,d=True):
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(n):
        if h:
            with torch.cuda.stream(s1): d_in.copy_(h2d_src,non_blocking=True)
        if d:
            with torch.cuda.stream(s2): d2h_dst.copy_(d_out,non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/n*1e3
for _ in range(2):
    print('h2d only %.3f ms  d2h only %.3f ms  both %.3f ms'%(run(20,h=True,d=False),run(20,h=False,d=True),run(20)))
