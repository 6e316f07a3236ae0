import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
def bench(M,N,K,name):
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16)
    b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    for _ in range(5): c=a@b.t()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(20):
        s.record(); c=a@b.t(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); t=ts[len(ts)//2]
    print(f"{name:28s} M={M} N={N} K={K}: {t*1e3:7.1f} us  {2*M*N*K/t/1e9:7.1f} TF")
bench(4608,3072,512,"stage6 expand")
bench(4608,512,3072,"stage6 project")
bench(18432,1536,256,"stage5 expand")
bench(18432,256,1536,"stage5 project")
bench(18432,768,192,"stage4 expand")
bench(4608,3840,640,"stage7 expand")
bench(4608,640,3840,"stage7 project")
bench(294912,64,256,"stage2 project")
bench(294912,192,2304,"fpn L3 as gemm")
