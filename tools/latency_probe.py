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

# different batches back to back (descriptor-table ring): enqueue cost when every call carries a new table
import ctypes
w, h, n = 640, 480, 16
bufs = []
for k in range(6):
    params = (ssim_amd.Params * n)()
    for i in range(n):
        da, db = ctx.alloc(w * h), ctx.alloc(w * h)
        ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, 100 * k + i)
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
    bufs.append((params, ctx.alloc(8 * n)))
ctx.synchronize()
for label, order in (("same batch re-enqueued", [0] * 600), ("two batches alternating", [0, 1] * 300), ("six batches round robin (ring of 4 tables: every call uploads)", list(range(6)) * 100)):
    for k in order[:12]:
        ctx.enqueue_batch(bufs[k][0], n, bufs[k][1].ptr)
    ctx.synchronize()
    t = time.perf_counter()
    for k in order:
        ctx.enqueue_batch(bufs[k][0], n, bufs[k][1].ptr)
    t_host = (time.perf_counter() - t) / len(order)
    ctx.synchronize()
    t_all = (time.perf_counter() - t) / len(order)
    print("%d x %dx%d, %s: host %.1f us per enqueue, %.1f us per batch end to end" % (n, w, h, label, t_host * 1e6, t_all * 1e6))
