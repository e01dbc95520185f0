import sys, numpy as np
sys.path.insert(0,'.')
import ssim_amd
ctx=ssim_amd.Context(0)
rng=np.random.default_rng(5)
bad=0
for (w,h) in [(256,256),(300,301),(1,1),(17,5),(129,64),(1920,1080),(4096,4096),(1000,37),(130,2049)]:
    for mode in (0,3):
        a=rng.integers(0,256,(h,w),dtype=np.uint8); b=np.clip(a.astype(np.int32)+rng.integers(-40,41,(h,w)),0,255).astype(np.uint8)
        da,db,dm,ds=ctx.upload(a),ctx.upload(b),ctx.alloc(4*w*h),ctx.alloc(8)
        p=ssim_amd.make_params(w,h,da.ptr,1,w,db.ptr,1,w,dm.ptr,1,w)
        one=(ssim_amd.Params*1)(p)
        ctx.set_mode(mode)
        res={}
        for v in (0,4):
            for rows in (0,8,64):
                ctx.set_tuning(rows,v)
                dm.upload(np.zeros((h,w),np.float32))
                ctx.enqueue_batch(one,1,ds.ptr); ctx.synchronize()
                res[(v,rows)]=(ds.download(np.float64,(1,)).copy().view(np.uint64)[0], dm.download(np.float32,(h,w)).copy())
        ref=res[(0,0)]
        for k,(sm,mp) in res.items():
            if sm!=ref[0] or not np.array_equal(mp.view(np.uint32),ref[1].view(np.uint32)):
                bad+=1; print("MISMATCH",w,h,mode,k, int((mp.view(np.uint32)!=ref[1].view(np.uint32)).sum()))
        for d in (da,db,dm,ds): d.free()
print("pair kernel check: mismatches", bad)
ctx.close()
