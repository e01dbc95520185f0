import sys, time; sys.path.insert(0,'.')
import numpy as np, ssim_amd
from ssim_amd import synth
ctx=ssim_amd.Context(0, mode=int(sys.argv[1]) if len(sys.argv)>1 else 1)
w=h=8192; n=2
bufs=[]
def mk(stride, step=1, shared=False):
    params=(ssim_amd.Params*n)()
    for i in range(n):
        da,db=ctx.alloc(w*h),ctx.alloc(w*h); ctx.synth_pair(da.ptr,w,db.ptr,w,w,h,0x5EED+i)
        dm=ctx.alloc(4*w*h*max(step,1))
        bufs.extend([da,db,dm])
        params[i]=ssim_amd.make_params(w,h,da.ptr,1,w,db.ptr,1,w,dm.ptr,step,stride)
    return params
sums=ctx.alloc(8*n)
for label,params in (("map stride W (normal)",mk(w)),("map stride 0 (every row onto row 0: cache resident)",mk(0)),("map step 2 stride 2W",mk(2*w,2))):
    for _ in range(20): ctx.enqueue_batch(params,n,sums.ptr)
    ctx.synchronize(); ctx.get_profile(); ctx.set_profiling(True)
    for _ in range(20): ctx.enqueue_batch(params,n,sums.ptr)
    ctx.synchronize(); k,ms=ctx.get_profile(); ctx.set_profiling(False)
    print("%-55s %.4f ms  %.1f Gpix/s"%(label,ms/k,n*w*h/(ms/k)/1e6))
