import time, torch
X = (torch.rand(2048, 512*512, device="cuda") < 0.2).float()
def T(name, f):
    f(); torch.cuda.synchronize(); t0=time.perf_counter(); r=f(); torch.cuda.synchronize(); print(f"{name:50s} {1e3*(time.perf_counter()-t0):8.2f} ms")
for nm, A in (("cm", X), ("view .t()", X.t())):
    T(nm+" sum(dtype=f64)", lambda: A.sum(dtype=torch.float64))
    T(nm+" sum()", lambda: A.sum())
    T(nm+" (A!=0).sum()", lambda: (A != 0).sum())
    T(nm+" count_nonzero", lambda: torch.count_nonzero(A))
    T(nm+" sum(dim=0,f64)", lambda: A.sum(dim=0, dtype=torch.float64))
    T(nm+" sum(dim=1,f64)", lambda: A.sum(dim=1, dtype=torch.float64))
    T(nm+" mean(f64)", lambda: A.mean(dtype=torch.float64))
    T(nm+" isfinite.all", lambda: torch.isfinite(A).all())
    T(nm+" (A<0).any", lambda: (A < 0).any())
    T(nm+" bf16 exact", lambda: (A.to(torch.bfloat16).to(A.dtype) - A).abs().max())
    T(nm+" A*s", lambda: A * 0.5)
