"""ctypes binding of the C ABI in include/rmgr/ssim-hip.h and include/rmgr/ssim.h.

This module is plumbing for tests and bench.py: the product is the shared library
(ssim_amd/lib/librmgr-ssim-hip.so) and its C/C++ headers.  Struct layouts mirror
include/rmgr/ssim.h field for field (reference include/rmgr/ssim.h:469-533).

There is no fallback of any kind: if the library is missing, or no gfx950 device is usable,
calls raise (ImportError / SsimError with the errno the C ABI returned).
"""
import ctypes
import errno
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMGR_SSIM_LIB") or os.path.join(_HERE, "lib", "librmgr-ssim-hip.so")   # env override: A/B builds in tools/

MODE_EXACT, MODE_FAST, MODE_DOUBLE, MODE_UNFUSED, MODE_SEPARABLE = 0, 1, 2, 3, 4

c_pd = ctypes.c_ssize_t  # ptrdiff_t

AllocFct = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t)
DeallocFct = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
ThreadFct = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint32)
ThreadPoolFct = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ThreadFct, ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.c_uint32, ctypes.c_uint32)


class Version(ctypes.Structure):
    _fields_ = [("major", ctypes.c_uint32), ("minor", ctypes.c_uint32), ("patch", ctypes.c_uint32),
                ("string", ctypes.c_char_p)]


class ImgParams(ctypes.Structure):
    _fields_ = [("topLeft", ctypes.c_void_p), ("step", c_pd), ("stride", c_pd)]


class Params(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("imgA", ImgParams), ("imgB", ImgParams),
                ("ssimMap", ctypes.c_void_p), ("ssimStep", c_pd), ("ssimStride", c_pd),
                ("alloc", AllocFct), ("dealloc", DeallocFct)]


class Plan(ctypes.Structure):
    _fields_ = [("structSize", ctypes.c_uint32), ("stripWidth", ctypes.c_uint32), ("stripRows", ctypes.c_uint32), ("stripsX", ctypes.c_uint32),
                ("stripsY", ctypes.c_uint32), ("wavefronts", ctypes.c_uint32), ("waveSlots", ctypes.c_uint32), ("earlyRowSums", ctypes.c_uint32),
                ("cellRows", ctypes.c_uint32), ("cellsX", ctypes.c_uint32), ("cellsY", ctypes.c_uint32),
                ("balancedChunks", ctypes.c_uint32), ("balancedChunkRows", ctypes.c_uint32), ("balancedInterleave", ctypes.c_uint32)]


class TunedEntry(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("count", ctypes.c_uint32), ("withMap", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("variant", ctypes.c_int32), ("stripRows", ctypes.c_uint32)]


class TuneResult(ctypes.Structure):
    _fields_ = [("structSize", ctypes.c_uint32), ("candidates", ctypes.c_uint32), ("bestVariant", ctypes.c_int32), ("bestStripRows", ctypes.c_uint32),
                ("defaultMs", ctypes.c_double), ("bestMs", ctypes.c_double),
                ("candidateVariant", ctypes.c_int32 * 8), ("candidateStripRows", ctypes.c_uint32 * 8), ("candidateMs", ctypes.c_double * 8)]


ABI_VERSION = 6      # RMGR_SSIM_HIP_ABI_VERSION of include/rmgr/ssim-hip.h this binding was written against


class ThreadPool(ctypes.Structure):
    _fields_ = [("dispatch", ThreadPoolFct), ("context", ctypes.c_void_p), ("threadCount", ctypes.c_uint32)]


class SsimError(RuntimeError):
    def __init__(self, fn, code):
        self.errno = code
        RuntimeError.__init__(self, "%s failed: errno %d (%s)" % (fn, code, errno.errorcode.get(code, "?")))


# every symbol include/rmgr/*.h declares with C linkage (tests check the library exports them all)
C_SYMBOLS = [
    "rmgr_ssim_get_version", "rmgr_ssim_init_interleaved", "rmgr_ssim_init_planar", "rmgr_ssim_use_default_allocator",
    "rmgr_ssim_compute_ssim", "rmgr_ssim_compute_ssim_openmp",
    "rmgr_ssim_hip_get_device_count", "rmgr_ssim_hip_create", "rmgr_ssim_hip_destroy", "rmgr_ssim_hip_set_mode",
    "rmgr_ssim_hip_get_mode", "rmgr_ssim_hip_set_tuning", "rmgr_ssim_hip_get_plan", "rmgr_ssim_hip_compute_ssim_host",
    "rmgr_ssim_hip_compute_ssim_device", "rmgr_ssim_hip_compute_ssim_batch_host", "rmgr_ssim_hip_compute_ssim_batch_host_devices",
    "rmgr_ssim_hip_enqueue_batch", "rmgr_ssim_hip_finalize",
    "rmgr_ssim_hip_synchronize", "rmgr_ssim_hip_malloc", "rmgr_ssim_hip_free", "rmgr_ssim_hip_memcpy_h2d",
    "rmgr_ssim_hip_memcpy_d2h", "rmgr_ssim_hip_set_profiling", "rmgr_ssim_hip_get_profile", "rmgr_ssim_hip_describe",
    "rmgr_ssim_hip_compute_ssim_channels_host", "rmgr_ssim_hip_compute_ssim_luminance_host", "rmgr_ssim_hip_luminance_device",
    "rmgr_ssim_hip_synth_pair_device",
    "rmgr_ssim_hip_comm_get_unique_id", "rmgr_ssim_hip_comm_init", "rmgr_ssim_hip_comm_allreduce_sums", "rmgr_ssim_hip_comm_destroy",
    "rmgr_ssim_hip_comm_rank_count", "rmgr_ssim_hip_comm_describe", "rmgr_ssim_hip_get_abi_version", "rmgr_ssim_hip_get_default_pool", "rmgr_ssim_hip_get_kernel_source_id",
    "rmgr_ssim_hip_enqueue_rows", "rmgr_ssim_hip_reduce_cells", "rmgr_ssim_hip_probe_valu",
    "rmgr_ssim_hip_trim", "rmgr_ssim_hip_trim_default_pool", "rmgr_ssim_hip_get_default_pool_memory", "rmgr_ssim_hip_get_memory_info",
    "rmgr_ssim_hip_tune", "rmgr_ssim_hip_clear_tuned", "rmgr_ssim_hip_get_tuned", "rmgr_ssim_hip_set_tuned", "rmgr_ssim_hip_get_profile_clock",
]
# non-inline C++ entry points of the reference, Itanium-mangled
CXX_SYMBOLS = [
    "_ZN4rmgr4ssim12compute_ssimEPfRK17rmgr_ssim_Params_PK21rmgr_ssim_ThreadPool_",
    "_ZN4rmgr4ssim12compute_ssimERKNS0_6ParamsE",
    "_ZN4rmgr4ssim11select_implENS0_14ImplementationE",
]

_lib = None


def load_library(path=None):
    """dlopen the product library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("%s not found: build it with `make lib` (hipcc --offload-arch=gfx950); "
                          "ssim_amd has no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    vp, i32, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32
    PP = ctypes.POINTER(Params)
    sig = {
        "rmgr_ssim_get_version": [ctypes.POINTER(Version)],
        "rmgr_ssim_init_interleaved": [ctypes.POINTER(ImgParams), vp, c_pd, u32, u32],
        "rmgr_ssim_init_planar": [ctypes.POINTER(ImgParams), ctypes.POINTER(vp), ctypes.POINTER(c_pd), u32],
        "rmgr_ssim_use_default_allocator": [PP],
        "rmgr_ssim_compute_ssim": [ctypes.POINTER(ctypes.c_float), PP, ctypes.POINTER(ThreadPool)],
        "rmgr_ssim_compute_ssim_openmp": [ctypes.POINTER(ctypes.c_float), PP],
        "rmgr_ssim_hip_get_device_count": [ctypes.POINTER(i32)],
        "rmgr_ssim_hip_create": [ctypes.POINTER(vp), i32, vp],
        "rmgr_ssim_hip_destroy": [vp],
        "rmgr_ssim_hip_set_mode": [vp, i32],
        "rmgr_ssim_hip_get_mode": [vp, ctypes.POINTER(i32)],
        "rmgr_ssim_hip_set_tuning": [vp, i32, i32],
        "rmgr_ssim_hip_get_plan": [vp, u32, u32, u32, ctypes.POINTER(Plan)],
        "rmgr_ssim_hip_compute_ssim_host": [vp, ctypes.POINTER(ctypes.c_float), PP, ctypes.POINTER(ThreadPool)],
        "rmgr_ssim_hip_compute_ssim_device": [vp, ctypes.POINTER(ctypes.c_float), PP],
        "rmgr_ssim_hip_enqueue_batch": [vp, u32, PP, vp],
        "rmgr_ssim_hip_compute_ssim_batch_host": [vp, u32, PP, ctypes.POINTER(ctypes.c_float)],
        "rmgr_ssim_hip_compute_ssim_batch_host_devices": [ctypes.POINTER(i32), u32, i32, u32, PP, ctypes.POINTER(ctypes.c_float)],
        "rmgr_ssim_hip_finalize": [u32, ctypes.POINTER(ctypes.c_double), u32, u32, ctypes.POINTER(ctypes.c_float)],
        "rmgr_ssim_hip_synchronize": [vp],
        "rmgr_ssim_hip_malloc": [vp, ctypes.POINTER(vp), ctypes.c_size_t],
        "rmgr_ssim_hip_free": [vp, vp],
        "rmgr_ssim_hip_memcpy_h2d": [vp, vp, vp, ctypes.c_size_t],
        "rmgr_ssim_hip_memcpy_d2h": [vp, vp, vp, ctypes.c_size_t],
        "rmgr_ssim_hip_set_profiling": [vp, i32],
        "rmgr_ssim_hip_get_profile": [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)],
        "rmgr_ssim_hip_compute_ssim_channels_host": [vp, ctypes.POINTER(ctypes.c_float), vp, c_pd, vp, c_pd, u32, u32, u32, vp],
        "rmgr_ssim_hip_compute_ssim_luminance_host": [vp, ctypes.POINTER(ctypes.c_float), vp, c_pd, vp, c_pd, u32, u32, u32, vp],
        "rmgr_ssim_hip_luminance_device": [vp, vp, c_pd, vp, c_pd, c_pd, u32, u32],
        "rmgr_ssim_hip_synth_pair_device": [vp, vp, c_pd, vp, c_pd, u32, u32, ctypes.c_uint64],
        "rmgr_ssim_hip_comm_get_unique_id": [ctypes.c_char_p],
        "rmgr_ssim_hip_comm_init": [vp, ctypes.c_char_p, i32, i32],
        "rmgr_ssim_hip_comm_allreduce_sums": [vp, vp, u32],
        "rmgr_ssim_hip_comm_destroy": [vp],
        "rmgr_ssim_hip_comm_rank_count": [vp, ctypes.POINTER(i32)],
        "rmgr_ssim_hip_enqueue_rows": [vp, PP, u32, u32, vp],
        "rmgr_ssim_hip_reduce_cells": [vp, u32, u32, u32, vp, vp],
        "rmgr_ssim_hip_probe_valu": [vp, i32, i32, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
        "rmgr_ssim_hip_get_profile_clock": [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)],
        "rmgr_ssim_hip_trim": [vp],
        "rmgr_ssim_hip_tune": [vp, u32, u32, u32, i32, ctypes.POINTER(TuneResult)],
        "rmgr_ssim_hip_clear_tuned": [vp],
        "rmgr_ssim_hip_get_tuned": [vp, u32, ctypes.POINTER(TunedEntry)],
        "rmgr_ssim_hip_set_tuned": [vp, u32, u32, u32, i32, i32, u32],
        "rmgr_ssim_hip_trim_default_pool": [],
        "rmgr_ssim_hip_get_default_pool_memory": [ctypes.POINTER(ctypes.c_uint64)] * 3,
        "rmgr_ssim_hip_get_memory_info": [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)],
    }
    for name, args in sig.items():
        if path is None and os.environ.get("RMGR_SSIM_LIB") and not hasattr(lib, name):
            continue                      # A/B runs against an older build of the library (tools/ab_libs.sh)
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    if hasattr(lib, "rmgr_ssim_hip_get_abi_version"):
        lib.rmgr_ssim_hip_get_abi_version.argtypes = []
        lib.rmgr_ssim_hip_get_abi_version.restype = i32
        if path is None and not os.environ.get("RMGR_SSIM_LIB") and lib.rmgr_ssim_hip_get_abi_version() != ABI_VERSION:
            raise ImportError("%s implements rmgr/ssim-hip.h interface version %d, this binding expects %d: rebuild with `make lib`"
                              % (p, lib.rmgr_ssim_hip_get_abi_version(), ABI_VERSION))
    lib.rmgr_ssim_hip_describe.argtypes = [vp]
    lib.rmgr_ssim_hip_describe.restype = ctypes.c_char_p
    if hasattr(lib, "rmgr_ssim_hip_comm_describe"):
        lib.rmgr_ssim_hip_comm_describe.argtypes = []
        lib.rmgr_ssim_hip_comm_describe.restype = ctypes.c_char_p
    if path is None:
        _lib = lib
    return lib


def get_plan(width, height, count=1, ctx=None):
    """Strip geometry of a launch (include/rmgr/ssim-hip.h rmgr_ssim_hip_get_plan); needs no device when ctx is None."""
    lib = load_library()
    out = Plan()
    out.structSize = ctypes.sizeof(Plan)
    _check("rmgr_ssim_hip_get_plan", lib.rmgr_ssim_hip_get_plan(ctx.handle if ctx is not None else None, width, height, count, ctypes.byref(out)))
    return out


def _check(name, rc):
    if rc != 0:
        raise SsimError(name, rc)


def device_count():
    n = ctypes.c_int32(0)
    _check("rmgr_ssim_hip_get_device_count", load_library().rmgr_ssim_hip_get_device_count(ctypes.byref(n)))
    return n.value


def get_version():
    v = Version()
    _check("rmgr_ssim_get_version", load_library().rmgr_ssim_get_version(ctypes.byref(v)))
    return (v.major, v.minor, v.patch, v.string.decode())


def make_params(width, height, a_ptr, a_step, a_stride, b_ptr, b_step, b_stride, map_ptr=None, map_step=1, map_stride=None):
    p = Params()
    p.width, p.height = width, height
    p.imgA = ImgParams(a_ptr, a_step, a_stride)
    p.imgB = ImgParams(b_ptr, b_step, b_stride)
    p.ssimMap = map_ptr
    p.ssimStep = map_step
    p.ssimStride = width if map_stride is None else map_stride
    return p


def finalize(sums, width, height):
    """float(sum / double(width*height)) per entry, computed by the library (src/ssim.cpp:1102)."""
    sums = np.ascontiguousarray(sums, np.float64)
    out = np.empty(sums.shape, np.float32)
    _check("rmgr_ssim_hip_finalize", load_library().rmgr_ssim_hip_finalize(
        sums.size, sums.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), width, height,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    return out


def compute_ssim(a, b, want_map=False, openmp=False, allocator=False, out_map=None):
    """The drop-in entry point rmgr_ssim_compute_ssim() on two host uint8 planes (H x W numpy).
    out_map: reuse this H x W float32 array for the map (avoids first-touch page faults in timing loops)."""
    lib = load_library()
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w = a.shape
    if out_map is not None:
        assert out_map.shape == (h, w) and out_map.dtype == np.float32 and out_map.flags.c_contiguous
        want_map = True
    m = (out_map if out_map is not None else np.empty((h, w), np.float32)) if want_map else None
    p = make_params(w, h, a.ctypes.data, 1, a.strides[0], b.ctypes.data, 1, b.strides[0],
                    m.ctypes.data if want_map else None, 1, w)
    if allocator:
        _check("rmgr_ssim_use_default_allocator", lib.rmgr_ssim_use_default_allocator(ctypes.byref(p)))
    out = ctypes.c_float()
    if openmp:
        _check("rmgr_ssim_compute_ssim_openmp", lib.rmgr_ssim_compute_ssim_openmp(ctypes.byref(out), ctypes.byref(p)))
    else:
        _check("rmgr_ssim_compute_ssim", lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None))
    return np.float32(out.value), m


def kernel_source_id():
    """sha256 of the kernel source the loaded library was compiled from (rmgr_ssim_hip_get_kernel_source_id)."""
    lib = load_library()
    lib.rmgr_ssim_hip_get_kernel_source_id.restype = ctypes.c_char_p
    lib.rmgr_ssim_hip_get_kernel_source_id.argtypes = []
    return lib.rmgr_ssim_hip_get_kernel_source_id().decode()


def default_pool():
    """(contexts that exist, calls that may be in flight at a time) of the ctx == NULL entry points' default contexts."""
    lib = load_library()
    n, lim = ctypes.c_int32(), ctypes.c_int32()
    lib.rmgr_ssim_hip_get_default_pool.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    _check("rmgr_ssim_hip_get_default_pool", lib.rmgr_ssim_hip_get_default_pool(ctypes.byref(n), ctypes.byref(lim)))
    return n.value, lim.value


def default_pool_memory():
    """(device bytes, pinned host bytes) the default contexts held when their last call ended, and the retain cap in bytes (per context)."""
    d, p, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _check("rmgr_ssim_hip_get_default_pool_memory", load_library().rmgr_ssim_hip_get_default_pool_memory(ctypes.byref(d), ctypes.byref(p), ctypes.byref(c)))
    return d.value, p.value, c.value


def trim_default_pool():
    """Releases the staging of every default context that is not inside a call (rmgr_ssim_hip_trim_default_pool)."""
    _check("rmgr_ssim_hip_trim_default_pool", load_library().rmgr_ssim_hip_trim_default_pool())


def memory_info(ctx=None):
    """(free, total) bytes of the context's device (ctx None: the default contexts' device)."""
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    _check("rmgr_ssim_hip_get_memory_info", load_library().rmgr_ssim_hip_get_memory_info(ctx.handle if ctx is not None else None, ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def compute_ssim_batch(pairs, ctx=None):
    """Global SSIM of many host image pairs of one size (rmgr_ssim_hip_compute_ssim_batch_host: pipelined staging).
    pairs: sequence of (a, b) uint8 arrays, H x W (any strides numpy can express along both axes)."""
    lib = load_library()
    n = len(pairs)
    params = (Params * n)()
    for i, (a, b) in enumerate(pairs):
        h, w = a.shape
        params[i] = make_params(w, h, a.ctypes.data, a.strides[1], a.strides[0], b.ctypes.data, b.strides[1], b.strides[0])
    out = (ctypes.c_float * n)()
    _check("rmgr_ssim_hip_compute_ssim_batch_host", lib.rmgr_ssim_hip_compute_ssim_batch_host(ctx.handle if ctx is not None else None, n, params, out))
    return np.array(out[:], np.float32)


def compute_ssim_batch_devices(pairs, devices=None, mode=MODE_EXACT):
    """compute_ssim_batch() sharded by image over `devices` (a list of device indices; None: all visible) from this one
    process (rmgr_ssim_hip_compute_ssim_batch_host_devices)."""
    lib = load_library()
    n = len(pairs)
    params = (Params * max(n, 1))()
    for i, (a, b) in enumerate(pairs):
        h, w = a.shape
        params[i] = make_params(w, h, a.ctypes.data, a.strides[1], a.strides[0], b.ctypes.data, b.strides[1], b.strides[0])
    out = (ctypes.c_float * max(n, 1))()
    devs = (ctypes.c_int32 * len(devices))(*devices) if devices is not None else None
    _check("rmgr_ssim_hip_compute_ssim_batch_host_devices",
           lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(devs, len(devices) if devices is not None else 0, mode, n, params, out))
    return np.array(out[:n], np.float32)


def compute_ssim_channels(a, b, want_map=False):
    """All channels of two interleaved H x W x C uint8 arrays in one launch (host pointers)."""
    lib = load_library()
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w, ch = a.shape
    out = (ctypes.c_float * ch)()
    m = np.empty((h, w, ch), np.float32) if want_map else None
    _check("rmgr_ssim_hip_compute_ssim_channels_host", lib.rmgr_ssim_hip_compute_ssim_channels_host(
        None, out, a.ctypes.data, w * ch, b.ctypes.data, w * ch, w, h, ch, m.ctypes.data if want_map else None))
    return np.array(out[:], np.float32), m


def compute_ssim_luminance(a, b, want_map=False):
    """SSIM of the BT.601 luminance (computed on the GPU) of two interleaved H x W x C (C >= 3) uint8 arrays."""
    lib = load_library()
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w, ch = a.shape
    out = ctypes.c_float()
    m = np.empty((h, w), np.float32) if want_map else None
    _check("rmgr_ssim_hip_compute_ssim_luminance_host", lib.rmgr_ssim_hip_compute_ssim_luminance_host(
        None, ctypes.byref(out), a.ctypes.data, w * ch, b.ctypes.data, w * ch, w, h, ch, m.ctypes.data if want_map else None))
    return np.float32(out.value), m


class DeviceBuffer(object):
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = ctypes.c_void_p()
        _check("rmgr_ssim_hip_malloc", ctx.lib.rmgr_ssim_hip_malloc(ctx.handle, ctypes.byref(p), nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check("rmgr_ssim_hip_memcpy_h2d", self.ctx.lib.rmgr_ssim_hip_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _check("rmgr_ssim_hip_memcpy_d2h", self.ctx.lib.rmgr_ssim_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.rmgr_ssim_hip_free(self.ctx.handle, self.ptr)
            self.ptr = None


class Context(object):
    """rmgr_ssim_hip_Context: one engine bound to one device and one stream."""

    def __init__(self, device=0, stream=None, mode=MODE_EXACT):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check("rmgr_ssim_hip_create", self.lib.rmgr_ssim_hip_create(ctypes.byref(h), device, stream))
        self.handle = h
        self.set_mode(mode)

    def close(self):
        if self.handle:
            self.lib.rmgr_ssim_hip_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def describe(self):
        return self.lib.rmgr_ssim_hip_describe(self.handle).decode()

    def set_mode(self, mode):
        _check("rmgr_ssim_hip_set_mode", self.lib.rmgr_ssim_hip_set_mode(self.handle, mode))

    def tune(self, width, height, count, with_map=False):
        """Times the candidate plans of this launch shape on the device and keeps the winner for this context (rmgr_ssim_hip_tune).
        Returns {"default_ms", "best_ms", "best": (variant, strip_rows), "candidates": [(variant, strip_rows, ms), ...]} (candidates[0]: the default)."""
        r = TuneResult()
        r.structSize = ctypes.sizeof(TuneResult)
        _check("rmgr_ssim_hip_tune", self.lib.rmgr_ssim_hip_tune(self.handle, width, height, count, 1 if with_map else 0, ctypes.byref(r)))
        return {"default_ms": r.defaultMs, "best_ms": r.bestMs, "best": (r.bestVariant, r.bestStripRows),
                "candidates": [(r.candidateVariant[i], r.candidateStripRows[i], r.candidateMs[i]) for i in range(min(r.candidates, 8))]}

    def clear_tuned(self):
        _check("rmgr_ssim_hip_clear_tuned", self.lib.rmgr_ssim_hip_clear_tuned(self.handle))

    def tuned(self):
        """The measured choices the context holds: [(width, height, count, with_map, mode, variant, strip_rows), ...] (rmgr_ssim_hip_get_tuned)."""
        out, e = [], TunedEntry()
        while True:
            rc = self.lib.rmgr_ssim_hip_get_tuned(self.handle, len(out), ctypes.byref(e))
            if rc == errno.ENOENT:
                return out
            _check("rmgr_ssim_hip_get_tuned", rc)
            out.append((e.width, e.height, e.count, bool(e.withMap), e.mode, e.variant, e.stripRows))

    def set_tuned(self, width, height, count, with_map, variant, strip_rows):
        """Installs a choice for a launch shape under the context's current mode without measuring (rmgr_ssim_hip_set_tuned)."""
        _check("rmgr_ssim_hip_set_tuned", self.lib.rmgr_ssim_hip_set_tuned(self.handle, width, height, count, 1 if with_map else 0, variant, strip_rows))

    def trim(self):
        """Gives the context's grow-only staging back to the system (rmgr_ssim_hip_trim)."""
        _check("rmgr_ssim_hip_trim", self.lib.rmgr_ssim_hip_trim(self.handle))

    def set_tuning(self, strip_rows=0, variant=0):
        _check("rmgr_ssim_hip_set_tuning", self.lib.rmgr_ssim_hip_set_tuning(self.handle, strip_rows, variant))

    def alloc(self, nbytes):
        return DeviceBuffer(self, max(int(nbytes), 1))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.alloc(arr.nbytes).upload(arr)

    def download(self, ptr, dtype, shape):
        """Copy device memory at `ptr` into a new numpy array (blocking)."""
        out = np.empty(shape, dtype)
        _check("rmgr_ssim_hip_memcpy_d2h", self.lib.rmgr_ssim_hip_memcpy_d2h(self.handle, out.ctypes.data, ptr, out.nbytes))
        return out

    def synchronize(self):
        _check("rmgr_ssim_hip_synchronize", self.lib.rmgr_ssim_hip_synchronize(self.handle))

    def compute_host(self, params, want_global=True, thread_pool=None):
        out = ctypes.c_float()
        rc = self.lib.rmgr_ssim_hip_compute_ssim_host(self.handle, ctypes.byref(out) if want_global else None,
                                                      ctypes.byref(params), thread_pool)
        _check("rmgr_ssim_hip_compute_ssim_host", rc)
        return np.float32(out.value)

    def compute_device(self, params, want_global=True):
        out = ctypes.c_float()
        rc = self.lib.rmgr_ssim_hip_compute_ssim_device(self.handle, ctypes.byref(out) if want_global else None, ctypes.byref(params))
        _check("rmgr_ssim_hip_compute_ssim_device", rc)
        return np.float32(out.value)

    def enqueue_batch(self, params_array, count, sums_dev_ptr):
        _check("rmgr_ssim_hip_enqueue_batch", self.lib.rmgr_ssim_hip_enqueue_batch(self.handle, count, params_array, sums_dev_ptr))

    def enqueue_rows(self, params, y_begin, y_rows, cells_dev_ptr):
        """Rows [y_begin, y_begin + y_rows) of one device-resident pair: map rows + cell partials into a zeroed cell array."""
        _check("rmgr_ssim_hip_enqueue_rows", self.lib.rmgr_ssim_hip_enqueue_rows(self.handle, ctypes.byref(params), y_begin, y_rows, cells_dev_ptr))

    def reduce_cells(self, width, height, count, cells_dev_ptr, sums_dev_ptr):
        _check("rmgr_ssim_hip_reduce_cells", self.lib.rmgr_ssim_hip_reduce_cells(self.handle, width, height, count, cells_dev_ptr, sums_dev_ptr))

    # ---- native RCCL exchange (rmgr_ssim_hip_comm_*) ----
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(128)
        _check("rmgr_ssim_hip_comm_get_unique_id", load_library().rmgr_ssim_hip_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank_count, rank):
        _check("rmgr_ssim_hip_comm_init", self.lib.rmgr_ssim_hip_comm_init(self.handle, unique_id, rank_count, rank))

    def comm_allreduce_sums(self, sums_dev_ptr, count):
        _check("rmgr_ssim_hip_comm_allreduce_sums", self.lib.rmgr_ssim_hip_comm_allreduce_sums(self.handle, sums_dev_ptr, count))

    def comm_destroy(self):
        _check("rmgr_ssim_hip_comm_destroy", self.lib.rmgr_ssim_hip_comm_destroy(self.handle))

    def comm_rank_count(self):
        """Ranks RCCL counts in this context's communicator (ncclCommCount); 0 without one."""
        n = ctypes.c_int32(0)
        _check("rmgr_ssim_hip_comm_rank_count", self.lib.rmgr_ssim_hip_comm_rank_count(self.handle, ctypes.byref(n)))
        return n.value

    @staticmethod
    def comm_describe():
        return load_library().rmgr_ssim_hip_comm_describe().decode()

    def synth_pair(self, a_ptr, a_stride, b_ptr, b_stride, width, height, seed):
        """Fill two device planes with the synthetic pair of SURVEY.md 8(d) (asynchronous on the context's stream)."""
        _check("rmgr_ssim_hip_synth_pair_device", self.lib.rmgr_ssim_hip_synth_pair_device(
            self.handle, a_ptr, a_stride, b_ptr, b_stride, width, height, seed))

    def probe_valu(self, waves_per_simd, stream_kind=0, launches=5, with_clock=False):
        """T lane-ops/s a pure packed-fp32 stream sustains on this device right now at a forced occupancy (rmgr_ssim_hip_probe_valu);
        with_clock: (T lane-ops/s, mean shader MHz over the XCDs, slowest XCD's MHz) of the timed launches."""
        t, mhz, lo = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check("rmgr_ssim_hip_probe_valu", self.lib.rmgr_ssim_hip_probe_valu(self.handle, waves_per_simd, stream_kind, launches, ctypes.byref(t),
                                                                             ctypes.byref(mhz) if with_clock else None, ctypes.byref(lo) if with_clock else None))
        return (t.value, mhz.value, lo.value) if with_clock else t.value

    def get_profile_clock(self):
        """(mean shader MHz over the XCDs, slowest XCD's MHz, launches) of the profiled strip-kernel launches since the last call (rmgr_ssim_hip_get_profile_clock)."""
        mhz, lo, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
        _check("rmgr_ssim_hip_get_profile_clock", self.lib.rmgr_ssim_hip_get_profile_clock(self.handle, ctypes.byref(mhz), ctypes.byref(lo), ctypes.byref(n)))
        return mhz.value, lo.value, n.value

    def set_profiling(self, on):
        _check("rmgr_ssim_hip_set_profiling", self.lib.rmgr_ssim_hip_set_profiling(self.handle, 1 if on else 0))

    def get_profile(self):
        n = ctypes.c_uint64()
        ms = ctypes.c_double()
        _check("rmgr_ssim_hip_get_profile", self.lib.rmgr_ssim_hip_get_profile(self.handle, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    # ---- convenience used by the tests: planes given as numpy arrays, staged explicitly ----
    def ssim_planes(self, a, b, want_map=False):
        """Upload two H x W uint8 planes, run the device path, return (ssim, map or None)."""
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        h, w = a.shape
        da, db = self.upload(a), self.upload(b)
        dm = self.alloc(4 * w * h) if want_map else None
        try:
            p = make_params(w, h, da.ptr, 1, w, db.ptr, 1, w, dm.ptr if dm else None, 1, w)
            v = self.compute_device(p)
            m = dm.download(np.float32, (h, w)) if dm else None
        finally:
            da.free()
            db.free()
            if dm:
                dm.free()
        return v, m
