import sys, time
sys.path.insert(0,"tests"); sys.path.insert(0,"oracle")
from gpu_common import load_binding
B=load_binding(); ctx=B.Context(0)
for n in (1<<12, 1<<14, 1<<15, 1<<16, 1<<17, 1<<18, 1<<20):
    pts=ctx.gen_g1(n,1); sc=ctx.gen_scalars(n,2,1)
    ctx.msm_g1_dev(pts.ptr, sc.ptr, n)
    ctx.msm_g1_dev(pts.ptr, sc.ptr, n); print(n, ctx.stats()["total_ms"], flush=True)
ctx.close()
