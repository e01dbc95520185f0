"""ctypes bindings for the CHECKER libraries (oracle/ssim_oracle.c and oracle/_ref).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (ssim_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libssim_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libssim_ref.so")
REF_DOUBLE_SO = os.path.join(_HERE, "_ref", "libssim_ref_double.so")   # the reference's RMGR_SSIM_USE_DOUBLE build

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_pd = ctypes.c_ssize_t


def build(quiet=True):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_oracle = None
_ref = None
_ref_double = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.oracle_ssim_f32.restype = ctypes.c_int
        lib.oracle_ssim_f32.argtypes = [_f32p, _f64p, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                        ctypes.c_void_p, _pd, _pd, ctypes.c_int, ctypes.c_int]
        lib.oracle_ssim_naive_f64.restype = ctypes.c_int
        lib.oracle_ssim_naive_f64.argtypes = [_f64p, _f64p, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                              ctypes.c_void_p, _pd, _pd, ctypes.c_int]
        lib.oracle_synth_pair.restype = None
        lib.oracle_synth_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                          ctypes.c_uint32, ctypes.c_uint64]
        lib.oracle_kernel21_f32.argtypes = [_f32p]
        lib.oracle_generic_kernel21_f32.argtypes = [_f32p]
        lib.oracle_kernel121_f64.argtypes = [_f64p]
        lib.oracle_max_threads.restype = ctypes.c_int
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_double():
    return os.path.exists(REF_DOUBLE_SO)


def _bind_ref(path):
    lib = ctypes.CDLL(path)
    lib.ref_compute_ssim.restype = ctypes.c_int
    lib.ref_compute_ssim.argtypes = [_f32p, _f64p, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                     ctypes.c_void_p, _pd, _pd, ctypes.c_int, ctypes.c_int]
    lib.ref_naive_f64.restype = ctypes.c_double
    lib.ref_naive_f64.argtypes = [ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                  ctypes.c_void_p, _pd, _pd]
    lib.ref_max_threads.restype = ctypes.c_int
    lib.ref_is_double.restype = ctypes.c_int
    return lib


def ref_double_lib():
    """The reference's fp64 flavour: the same kernels compiled with Float = double (oracle/Makefile)."""
    global _ref_double
    if _ref_double is None:
        _ref_double = _bind_ref(REF_DOUBLE_SO)
        assert _ref_double.ref_is_double() == 1
    return _ref_double


def ref_lib():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(REF_SO)
        lib.ref_compute_ssim.restype = ctypes.c_int
        lib.ref_compute_ssim.argtypes = [_f32p, _f64p, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                         ctypes.c_void_p, _pd, _pd, ctypes.c_int, ctypes.c_int]
        lib.ref_naive_f64.restype = ctypes.c_double
        lib.ref_naive_f64.argtypes = [ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_void_p, _pd, _pd, ctypes.c_void_p, _pd, _pd,
                                      ctypes.c_void_p, _pd, _pd]
        lib.ref_max_threads.restype = ctypes.c_int
        _ref = lib
    return _ref


def _addr(arr, offset=0):
    return ctypes.c_void_p(arr.ctypes.data + offset)


def synth_pair(width, height, seed=0x5EED):
    """SURVEY.md 8(d) synthetic pair (planar, step=1, stride=W)."""
    a = np.empty((height, width), np.uint8)
    b = np.empty((height, width), np.uint8)
    oracle_lib().oracle_synth_pair(_addr(a), _addr(b), width, height, seed)
    return a, b


def synth_pair_numpy(width, height, seed=0x5EED):
    """Same generator in numpy (cross-checks the C one)."""
    y, x = np.meshgrid(np.arange(height, dtype=np.uint64), np.arange(width, dtype=np.uint64), indexing="ij")
    z = np.uint64(seed) ^ ((y << np.uint64(32)) | x)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    r = z ^ (z >> np.uint64(31))
    g = ((3 * x + 5 * y) >> np.uint64(2)) & np.uint64(255)
    a = ((3 * g + (r & np.uint64(255))) >> np.uint64(2)).astype(np.int64)
    n = ((r >> np.uint64(8)) % np.uint64(33)).astype(np.int64) - 16
    b = np.clip(a + n, 0, 255)
    return a.astype(np.uint8), b.astype(np.uint8)


def _views(img, step, stride):
    """Return (array-to-keep-alive, base address of pixel (0,0), step, stride) for a 2-D uint8 view."""
    return img, img.ctypes.data, step, stride


def ssim_f32(a, b, want_map=False, fused=True, threads=1, a_step=None, a_stride=None,
             b_step=None, b_stride=None, width=None, height=None, out_map=None):
    """oracle fp32 SSIM of two uint8 arrays.  By default a, b are 2-D C-contiguous planes.
    Explicit step/stride (bytes) + width/height address pixel (x,y) at base + x*step + y*stride."""
    a = np.ascontiguousarray(a) if a_step is None else a
    b = np.ascontiguousarray(b) if b_step is None else b
    h = a.shape[0] if height is None else height
    w = a.shape[1] if width is None else width
    a_step = 1 if a_step is None else a_step
    b_step = 1 if b_step is None else b_step
    a_stride = a.strides[0] if a_stride is None else a_stride
    b_stride = b.strides[0] if b_stride is None else b_stride
    out = ctypes.c_float()
    s = ctypes.c_double()
    if out_map is not None:
        assert out_map.shape == (h, w) and out_map.dtype == np.float32 and out_map.flags.c_contiguous
        want_map = True
    m = (out_map if out_map is not None else np.empty((h, w), np.float32)) if want_map else None
    rc = oracle_lib().oracle_ssim_f32(ctypes.byref(out), ctypes.byref(s), w, h,
                                      _addr(a), a_step, a_stride, _addr(b), b_step, b_stride,
                                      _addr(m) if want_map else None, 1, w, int(fused), threads)
    if rc:
        raise RuntimeError("oracle_ssim_f32 -> errno %d" % rc)
    return np.float32(out.value), s.value, m


def ssim_naive_f64(a, b, want_map=False, threads=1):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w = a.shape
    out = ctypes.c_double()
    s = ctypes.c_double()
    m = np.empty((h, w), np.float64) if want_map else None
    rc = oracle_lib().oracle_ssim_naive_f64(ctypes.byref(out), ctypes.byref(s), w, h,
                                            _addr(a), 1, a.strides[0], _addr(b), 1, b.strides[0],
                                            _addr(m) if want_map else None, 1, w, threads)
    if rc:
        raise RuntimeError("oracle_ssim_naive_f64 -> errno %d" % rc)
    return out.value, s.value, m


def ref_ssim(a, b, want_map=False, impl=5, threads=1, out_map=None, double=False):
    """REAL reference kernels (oracle/_ref).  impl 5 = FMA, 4 = AVX.  out_map: write the map into this H x W float32 array
    (timing loops: a fresh 268 MB array per call measures the kernel's page faults, not the path).
    double=True: the RMGR_SSIM_USE_DOUBLE flavour (fp64 internals; the map is float there too)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w = a.shape
    out = ctypes.c_float()
    s = ctypes.c_double()
    if out_map is not None:
        assert out_map.shape == (h, w) and out_map.dtype == np.float32 and out_map.flags.c_contiguous
        want_map = True
    m = (out_map if out_map is not None else np.empty((h, w), np.float32)) if want_map else None
    rc = (ref_double_lib() if double else ref_lib()).ref_compute_ssim(ctypes.byref(out), ctypes.byref(s), w, h,
                                    _addr(a), 1, a.strides[0], _addr(b), 1, b.strides[0],
                                    _addr(m) if want_map else None, 1, w, impl, threads)
    if rc:
        raise RuntimeError("ref_compute_ssim -> errno %d" % rc)
    return np.float32(out.value), s.value, m


def ref_naive_f64(a, b, want_map=False):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    h, w = a.shape
    m = np.empty((h, w), np.float64) if want_map else None
    v = ref_lib().ref_naive_f64(w, h, _addr(a), 1, a.strides[0], _addr(b), 1, b.strides[0],
                                _addr(m) if want_map else None, 1, w)
    return v, m
