#!/usr/bin/env python3
"""What the FIRST call of a process costs (the reference's shipped callers make 1-4 calls per process: src/ssim-cli.cpp:197-210,
sample/rmgr-ssim-sample.cpp:84-95).  Run in a FRESH process (no torch, no HIP yet); prints one JSON object.

  plain        dlopen(librmgr-ssim-hip.so) -> rmgr_ssim_compute_ssim() on a 1080p pair in pageable host memory -> return:
               total_ms (the figure a one-shot caller sees), dlopen_ms, first_call_ms, then second_call_ms (steady state)
  split        the same work with the library's own entry points in between, to attribute the first call:
               runtime_init_ms   rmgr_ssim_hip_get_device_count (hipInit + device enumeration)
               context_ms        rmgr_ssim_hip_create (stream, events, first allocations)
               code_object_ms    first launch of ANY kernel of the library (the synthetic-pair generator, 64 x 64): HIP loads the
                                 library's code object -- all strip-kernel instantiations -- at the first launch
               first_ssim_ms     first rmgr_ssim_hip_compute_ssim_device on a device-resident 1080p pair (first launch of the strip kernel
                                 and of the reduction), against steady_ssim_ms for the second
  cli          wall time of ssim_amd/bin/rmgr-ssim on the reference's bbb1080 PNG vs its quality-50 JPEG (decode + 3 channels), if built

usage: python tools/cold_start_probe.py plain|split|cli
"""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RMGR_SSIM_LIB", os.path.join(ROOT, "ssim_amd", "lib", "librmgr-ssim-hip.so"))
W, H = 1920, 1080


class ImgParams(ctypes.Structure):
    _fields_ = [("topLeft", ctypes.c_void_p), ("step", ctypes.c_ssize_t), ("stride", ctypes.c_ssize_t)]


class Params(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("imgA", ImgParams), ("imgB", ImgParams), ("ssimMap", ctypes.c_void_p),
                ("ssimStep", ctypes.c_ssize_t), ("ssimStride", ctypes.c_ssize_t), ("alloc", ctypes.c_void_p), ("dealloc", ctypes.c_void_p)]


def ms(t0):
    return round((time.perf_counter() - t0) * 1e3, 3)


def plain():
    a = (ctypes.c_uint8 * (W * H))()
    b = (ctypes.c_uint8 * (W * H))()
    for i in range(0, W * H, 97):
        a[i] = i & 255
        b[i] = (i * 7) & 255
    p = Params(W, H, ImgParams(ctypes.addressof(a), 1, W), ImgParams(ctypes.addressof(b), 1, W), None, 0, 0, None, None)
    out = ctypes.c_float()
    t0 = time.perf_counter()
    lib = ctypes.CDLL(LIB)
    lib.rmgr_ssim_compute_ssim.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(Params), ctypes.c_void_p]
    d = ms(t0)
    t1 = time.perf_counter()
    rc = lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None)
    first = ms(t1)
    total = ms(t0)
    t2 = time.perf_counter()
    rc2 = lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None)
    second = ms(t2)
    return {"rc": [rc, rc2], "total_ms": total, "dlopen_ms": d, "first_call_ms": first, "second_call_ms": second, "ssim": out.value,
            "library_bytes": os.path.getsize(LIB)}


def split():
    t0 = time.perf_counter()
    lib = ctypes.CDLL(LIB)
    d = ms(t0)
    vp, i32, u32, pd = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_ssize_t
    n = i32()
    t = time.perf_counter()
    rc = [lib.rmgr_ssim_hip_get_device_count(ctypes.byref(n))]
    init = ms(t)
    ctx = vp()
    t = time.perf_counter()
    lib.rmgr_ssim_hip_create.argtypes = [ctypes.POINTER(vp), i32, vp]
    rc.append(lib.rmgr_ssim_hip_create(ctypes.byref(ctx), 0, None))
    cms = ms(t)
    lib.rmgr_ssim_hip_malloc.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_size_t]
    da, db = vp(), vp()
    t = time.perf_counter()
    rc.append(lib.rmgr_ssim_hip_malloc(ctx, ctypes.byref(da), W * H))
    rc.append(lib.rmgr_ssim_hip_malloc(ctx, ctypes.byref(db), W * H))
    alloc = ms(t)
    lib.rmgr_ssim_hip_synth_pair_device.argtypes = [vp, vp, pd, vp, pd, u32, u32, ctypes.c_uint64]
    lib.rmgr_ssim_hip_synchronize.argtypes = [vp]
    t = time.perf_counter()
    rc.append(lib.rmgr_ssim_hip_synth_pair_device(ctx, da, 64, db, 64, 64, 64, 0x5EED))
    rc.append(lib.rmgr_ssim_hip_synchronize(ctx))
    code = ms(t)
    t = time.perf_counter()
    rc.append(lib.rmgr_ssim_hip_synth_pair_device(ctx, da, W, db, W, W, H, 0x5EED))
    rc.append(lib.rmgr_ssim_hip_synchronize(ctx))
    synth = ms(t)
    p = Params(W, H, ImgParams(da.value, 1, W), ImgParams(db.value, 1, W), None, 0, 0, None, None)
    out = ctypes.c_float()
    lib.rmgr_ssim_hip_compute_ssim_device.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(Params)]
    t = time.perf_counter()
    rc.append(lib.rmgr_ssim_hip_compute_ssim_device(ctx, ctypes.byref(out), ctypes.byref(p)))
    first = ms(t)
    t = time.perf_counter()
    rc.append(lib.rmgr_ssim_hip_compute_ssim_device(ctx, ctypes.byref(out), ctypes.byref(p)))
    steady = ms(t)
    return {"rc": rc, "dlopen_ms": d, "runtime_init_ms": init, "context_ms": cms, "device_alloc_ms": alloc, "code_object_ms": code,
            "synth_1080p_ms": synth, "first_ssim_ms": first, "steady_ssim_ms": steady, "ssim_hex": "0x%08x" % ctypes.c_uint32.from_buffer(out).value}


def cli():
    exe = os.path.join(ROOT, "ssim_amd", "bin", "rmgr-ssim")
    img = os.path.join(ROOT, "tests", "golden", "images")
    a, b = os.path.join(img, "big_buck_bunny_1080_07806.png"), os.path.join(img, "big_buck_bunny_1080_07806_50.jpg")
    if not (os.path.exists(exe) and os.path.exists(a) and os.path.exists(b)):
        return {"skipped": "tool or images missing"}
    t0 = time.perf_counter()
    r = subprocess.run([exe, a, b], capture_output=True, text=True)
    wall = ms(t0)
    t0 = time.perf_counter()
    r2 = subprocess.run([exe, "-h"], capture_output=True, text=True)
    return {"rc": r.returncode, "wall_ms": wall, "help_only_wall_ms": ms(t0), "stdout": r.stdout.strip().splitlines()[-1:] if r.stdout else [], "rc_help": r2.returncode}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "plain"
    print(json.dumps({"what": what, **{"plain": plain, "split": split, "cli": cli}[what]()}))
