import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e-3
N=1<<30
x=torch.empty(N,dtype=torch.uint8,device='cuda'); y=torch.empty(N,dtype=torch.uint8,device='cuda')
xf=x.view(torch.float32)
print("fill  1 GiB: %.2f TB/s written" % (N/t(lambda: xf.zero_())/1e12))
print("sum   1 GiB: %.2f TB/s read" % (N/t(lambda: xf.sum())/1e12))
print("copy  1 GiB: %.2f TB/s read + %.2f written" % ((N/t(lambda: y.copy_(x))/1e12,)*2))
h=N//2
print("copy 0.5 GiB: %.2f TB/s each way" % (h/t(lambda: y[:h].copy_(x[:h]))/1e12))
# 2/3 writes + 1/3 reads (mlp_fb's mix): out[2M] = f(in[M])
a=torch.empty(N//4,dtype=torch.float32,device='cuda'); bq=torch.empty(2,N//4,dtype=torch.float32,device='cuda')
def mix(): torch.add(a, 1.0, out=bq[0]); torch.mul(a, 2.0, out=bq[1])
dt=t(mix)
print("2 x (read 256 MiB, write 256 MiB) elementwise: %.2f TB/s total" % ((4*(N//4)*4)/dt/1e12))
