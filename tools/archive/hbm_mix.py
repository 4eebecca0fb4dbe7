"""HBM bandwidth of four access mixes with stock torch kernels on 512 MB tensors (write only / copy / two reads + one write /
read only): a reference for the write-heavy 1x1 kernels.  python tools/hbm_mix.py"""
import torch, time
dev='cuda:0'
n=128*1024*1024
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev); c=torch.empty(n,device=dev)
def t(f,reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e-3
s=t(lambda: a.zero_()); print('write only  %.2f TB/s' % (n*4/s/1e12))
s=t(lambda: b.copy_(a)); print('copy        %.2f TB/s' % (2*n*4/s/1e12))
s=t(lambda: torch.add(a,b,out=c)); print('2 read 1 wr %.2f TB/s' % (3*n*4/s/1e12))
s=t(lambda: a.sum()); print('read only   %.2f TB/s' % (n*4/s/1e12))
