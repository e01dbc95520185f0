import sys,time; sys.path.insert(0,'.')
import numpy as np, ssim_amd
from ssim_amd import synth
ctx=ssim_amd.Context(0)
for (w,h) in [(256,256),(640,480),(1920,1080),(4096,4096)]:
    a,b=synth.pair_numpy(w,h,0x5EED)
    da,db=ctx.upload(a),ctx.upload(b)
    p=ssim_amd.make_params(w,h,da.ptr,1,w,db.ptr,1,w)
    sums=ctx.alloc(8)
    one=(ssim_amd.Params*1)(p)
    for _ in range(20): ctx.compute_device(p)
    N=200
    t=time.perf_counter()
    for _ in range(N): ctx.compute_device(p)
    t_dev=(time.perf_counter()-t)/N
    t=time.perf_counter()
    for _ in range(N): ctx.enqueue_batch(one,1,sums.ptr); ctx.synchronize()
    t_enq=(time.perf_counter()-t)/N
    t=time.perf_counter()
    for _ in range(N): ctx.enqueue_batch(one,1,sums.ptr)
    ctx.synchronize()
    t_pipe=(time.perf_counter()-t)/N
    for _ in range(5): ssim_amd.compute_ssim(a,b)
    t=time.perf_counter()
    for _ in range(N): ssim_amd.compute_ssim(a,b)
    t_host=(time.perf_counter()-t)/N
    ctx.set_profiling(True)
    for _ in range(20): ctx.enqueue_batch(one,1,sums.ptr)
    ctx.synchronize(); k,ms=ctx.get_profile(); ctx.set_profiling(False)
    print("%dx%d: kernel %.1f us | enqueue pipelined %.1f | enqueue+sync %.1f | compute_device %.1f | host-pointer call %.1f us"%(w,h,ms/k*1e3,t_pipe*1e6,t_enq*1e6,t_dev*1e6,t_host*1e6))
