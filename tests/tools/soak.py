"""Randomized soak of the GPU path against the oracle: random sizes / layouts / modes / kernel variants / strip
heights (run on the GPU box: python tests/tools/soak.py [cases=300] [seed=777]).  Bit-exact modes: every pixel identical
to the oracle; MODE_FAST / MODE_SEPARABLE: identical to the numpy model of their arithmetic (tests/tools/fast_mode_model.py;
their quotient is n * rcp(d) with the hardware's reciprocal: within 3 ulp of the exactly dividing model) and inside their tolerances;
MODE_DOUBLE: 1e-7 per pixel against the naive double oracle.  Every fourth case has no map (the launches plan() may give the balanced
form; tuning variant 6 forces it): its value must have the bits of the same launch with a map under the plain strips."""
import sys, numpy as np, ctypes
import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import ssim_amd, oracle
sys.path.insert(0,os.path.join(ROOT,'tests','tools'))
import fast_mode_model as model
from test_gpu_fuzz import make_layout
CASES=int(sys.argv[1]) if len(sys.argv)>1 else 300
rng=np.random.default_rng(int(sys.argv[2]) if len(sys.argv)>2 else 777)
ctx=ssim_amd.Context(0)
lib=oracle.oracle_lib()
bad=0
for case in range(CASES):
    big = case % 10 == 0
    h,w=(int(rng.integers(200,1400)),int(rng.integers(300,2100))) if big else (int(rng.integers(1,260)),int(rng.integers(1,400)))
    a=rng.integers(0,256,(h,w),dtype=np.uint8)
    b=np.clip(a.astype(np.int32)+rng.integers(-40,41,(h,w)),0,255).astype(np.uint8)
    mode=int(rng.choice([0,0,0,3,1,1,4,4,2]))
    ba,oa,sa,da_=make_layout(rng,a); bb,ob,sb,db_=make_layout(rng,b)
    variant=int(rng.choice([0,0,1,2,3,6,7,103])); rows=int(rng.choice([0,0,0,2,9,31,64,300]))
    nomap = case % 4 == 3          # every fourth case asks for the global value only: the launches without a map are the ones plan() may give the balanced form
    keep=[]
    try:
        da,db=ctx.upload(ba),ctx.upload(bb); dm=ctx.alloc(4*w*h); keep+=[da,db,dm]
        p=ssim_amd.make_params(w,h,da.ptr+oa,sa,da_,db.ptr+ob,sb,db_,None if nomap else dm.ptr,1,w)
        ctx.set_mode(mode); ctx.set_tuning(rows,variant)
        v=ctx.compute_device(p); m=None if nomap else dm.download(np.float32,(h,w))
    finally:
        for d in keep: d.free()
    if nomap:                      # the value alone: against the same launch WITH a map under the plain strips (its map is checked by the other three cases of four)
        keep=[]
        try:
            da,db=ctx.upload(ba),ctx.upload(bb); dm=ctx.alloc(4*w*h); keep+=[da,db,dm]
            p=ssim_amd.make_params(w,h,da.ptr+oa,sa,da_,db.ptr+ob,sb,db_,dm.ptr,1,w)
            ctx.set_tuning(0,2)
            v2=ctx.compute_device(p)
        finally:
            for d in keep: d.free()
        if np.float32(v).view(np.uint32)!=np.float32(v2).view(np.uint32):
            bad+=1; print('FAIL (no map vs map)',case,w,h,mode,variant,rows,sa,da_,sb,db_, float(v), float(v2))
        continue
    if mode in (0,3):
        ov,_,om=oracle.ssim_f32(a,b,want_map=True,fused=(mode==0),threads=8)
        ok=np.array_equal(m.view(np.uint32),om.view(np.uint32)) and abs(int(np.float32(v).view(np.int32))-int(np.float32(ov).view(np.int32)))<=1
    elif mode in (1,4):
        if h*w>400000: continue
        mm=(model.mode_fast if mode==1 else model.mode_separable)(a,b)
        ulps=np.abs(m.view(np.int32).astype(np.int64)-mm.view(np.int32).astype(np.int64))
        # both modes divide as n * rcp(d) (1 ulp reciprocal): up to 3 ulp of any pixel from the exactly dividing model
        ok=int(ulps.max())<=3
        if mode==1:
            ov,_,om=oracle.ssim_f32(a,b,want_map=True,threads=8)
            ok=ok and abs(float(v)-float(ov))<=1.5e-6 and np.abs(m.astype(np.float64)-om).max()<=6.3e-4
        else:
            nv,_,nm=oracle.ssim_naive_f64(a,b,want_map=True,threads=8)
            ok=ok and abs(float(v)-nv)<2e-6 and np.abs(m.astype(np.float64)-nm).max()<1e-3
    else:
        if h*w>60000: continue
        nv,_,nm=oracle.ssim_naive_f64(a,b,want_map=True,threads=8)
        ok=abs(float(v)-nv)<=7e-8 and np.abs(m.astype(np.float64)-nm).max()<=1e-7
    if not ok:
        bad+=1; print('FAIL',case,w,h,mode,variant,rows,sa,da_,sb,db_, float(v))
print('soak done, failures:',bad)
