"""CPU: the drop-in boundary.  The C-ABI library loads without a GPU, exports every symbol the
public headers declare, keeps the reference's struct layouts and validation order, and fails
loudly (ENODEV) instead of computing anything on the CPU."""
import ctypes
import errno
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

import ssim_amd
from conftest import GOLDEN, ROOT

INCLUDE = os.path.join(ROOT, "include")


def declared_c_functions():
    names = set()
    for h in ("ssim.h", "ssim-openmp.h", "ssim-hip.h"):
        txt = open(os.path.join(INCLUDE, "rmgr", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names.update(re.findall(r"\b(rmgr_ssim_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_everything_the_headers_declare(lib):
    declared = declared_c_functions()
    assert declared == set(ssim_amd.C_SYMBOLS), declared ^ set(ssim_amd.C_SYMBOLS)
    for name in sorted(declared) + ssim_amd.CXX_SYMBOLS:
        assert hasattr(lib, name), "missing export " + name


@pytest.mark.parametrize("name", ["librmgr-ssim-hip.so", "librmgr-ssim-hip-double.so"])
def test_library_exports_nothing_but_the_api(name):
    """The other direction: the dynamic symbol table holds the declared C functions and the reference's three non-inline
    C++ entry points (include/rmgr/ssim.h:686, :713, src/ssim_internal.h:53) and NOTHING else -- the ssim_hip:: interface
    between the ABI layer and the kernels, helper functions and the HIP fat-binary bookkeeping stay inside
    (ssim_amd/csrc/exports.map).  The reference's archive exposes only its API as well."""
    path = os.path.join(os.path.dirname(ssim_amd.LIB_PATH), name)
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    allowed = declared_c_functions() | set(ssim_amd.CXX_SYMBOLS)
    assert exported == allowed, (sorted(exported - allowed), sorted(allowed - exported))


def test_static_archive_exposes_nothing_but_the_api():
    """The archive under the reference's name (librmgr-ssim.a): one relocatable object whose global definitions are the
    declared API, as in the shared libraries -- a program that links it must not see (or collide with) ssim_hip::*."""
    libdir = os.path.dirname(ssim_amd.LIB_PATH)

    def exported(path):
        out = subprocess.run(["nm", "--defined-only", "-g", path], capture_output=True, text=True, check=True).stdout
        return set(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3)
    # the reference's split (CMakeLists.txt:205, :229): rmgr_ssim_compute_ssim_openmp is the one function of the second archive
    openmp = {"rmgr_ssim_compute_ssim_openmp"}
    allowed = (declared_c_functions() | set(ssim_amd.CXX_SYMBOLS)) - openmp
    got = exported(os.path.join(libdir, "librmgr-ssim.a"))
    assert got == allowed, (sorted(got - allowed), sorted(allowed - got))
    assert exported(os.path.join(libdir, "librmgr-ssim-openmp.a")) == openmp


@pytest.mark.parametrize("cxx", ["g++", "/opt/rocm/lib/llvm/bin/clang++"])
def test_static_archive_links_into_modern_cxx_programs(tmp_path, cxx):
    """the archive's single relocatable object must survive a client that carries the same COMDAT
    groups (std::make_shared, std::thread, unique_lock, vector, exceptions): built with g++ / ld.bfd and with clang++,
    -O2, and run (no device here: the call reports ENODEV; on the GPU box it computes)."""
    if not os.path.exists(cxx) and shutil.which(cxx) is None:
        pytest.skip(cxx + " not installed")
    archive = os.path.join(os.path.dirname(ssim_amd.LIB_PATH), "librmgr-ssim.a")
    exe = tmp_path / "cxx17_client"
    subprocess.run([cxx, "-std=c++17", "-O2", "-Wall", "-I", INCLUDE, os.path.join(ROOT, "tests", "cxx17_client.cpp"), "-o", str(exe),
                    archive, "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out[1] == "16" and int(out[0]) in (0, errno.ENODEV), out
    if not has_gpu():
        assert int(out[0]) == errno.ENODEV


def test_struct_layouts_match_reference_abi():
    # LP64 layout of include/rmgr/ssim.h:469-533 of the reference
    assert ctypes.sizeof(ssim_amd.Version) == 24
    assert ctypes.sizeof(ssim_amd.ImgParams) == 24
    assert ctypes.sizeof(ssim_amd.Params) == 96
    assert ssim_amd.Params.imgA.offset == 8 and ssim_amd.Params.imgB.offset == 32
    assert ssim_amd.Params.ssimMap.offset == 56 and ssim_amd.Params.ssimStep.offset == 64
    assert ssim_amd.Params.ssimStride.offset == 72 and ssim_amd.Params.alloc.offset == 80 and ssim_amd.Params.dealloc.offset == 88
    assert ctypes.sizeof(ssim_amd.ThreadPool) == 24


def test_struct_layouts_as_compiled(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text("""
#include <rmgr/ssim.h>
#include <rmgr/ssim-openmp.h>
#include <rmgr/ssim-hip.h>
#include <rmgr/ssim-version.h>
#include <stdio.h>
int main(void) {
    printf("%lu %lu %lu %lu %lu %lu %lu\\n", (unsigned long)sizeof(rmgr_ssim_Version), (unsigned long)sizeof(rmgr_ssim_ImgParams),
           (unsigned long)sizeof(rmgr_ssim_Params), (unsigned long)sizeof(rmgr_ssim_ThreadPool),
           (unsigned long)offsetof(rmgr_ssim_Params, ssimMap), (unsigned long)offsetof(rmgr_ssim_Params, alloc),
           (unsigned long)offsetof(rmgr_ssim_ThreadPool, threadCount));
    return 0;
}
""")
    exe = tmp_path / "layout"
    # the public headers are C89-clean (reference: include/rmgr/ssim.h compiles as C)
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Werror", "-I", INCLUDE, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["24", "24", "96", "24", "56", "80", "16"]


@pytest.mark.parametrize("std", ["c++98", "c++11", "c++17"])
def test_headers_compile_as_cxx(tmp_path, std):
    src = tmp_path / "hdr.cpp"
    src.write_text("#include <rmgr/ssim.h>\n#include <rmgr/ssim-openmp.h>\n#include <rmgr/ssim-hip.h>\n"
                   "int main() { rmgr::ssim::GeneralParams p = rmgr::ssim::GeneralParams(); (void)p; return 0; }\n")
    subprocess.run(["g++", "-std=" + std, "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", INCLUDE, str(src)], check=True)


def test_version_kat():
    # tests/rmgr-ssim-tests.cpp:510-517
    assert ssim_amd.get_version() == (2, 1, 0, "2.1.0")


def test_init_helpers_and_their_errors(lib):
    ip = ssim_amd.ImgParams()
    buf = (ctypes.c_uint8 * 64)()
    base = ctypes.addressof(buf)
    assert lib.rmgr_ssim_init_interleaved(ctypes.byref(ip), base, 48, 3, 2) == 0
    assert (ip.topLeft, ip.step, ip.stride) == (base + 2, 3, 48)
    assert lib.rmgr_ssim_init_interleaved(ctypes.byref(ip), base, 48, 3, 3) == errno.EINVAL   # src/ssim.cpp:168
    assert lib.rmgr_ssim_init_interleaved(None, base, 48, 3, 0) == errno.EINVAL
    assert lib.rmgr_ssim_init_interleaved(ctypes.byref(ip), None, 48, 3, 0) == errno.EINVAL
    planes = (ctypes.c_void_p * 2)(base, base + 32)
    strides = (ctypes.c_ssize_t * 2)(16, -16)
    assert lib.rmgr_ssim_init_planar(ctypes.byref(ip), planes, strides, 1) == 0
    assert (ip.topLeft, ip.step, ip.stride) == (base + 32, 1, -16)
    assert lib.rmgr_ssim_init_planar(ctypes.byref(ip), None, strides, 0) == errno.EINVAL
    assert lib.rmgr_ssim_init_planar(ctypes.byref(ip), planes, None, 0) == errno.EINVAL
    p = ssim_amd.Params()
    assert lib.rmgr_ssim_use_default_allocator(ctypes.byref(p)) == 0 and p.alloc and p.dealloc
    assert lib.rmgr_ssim_use_default_allocator(None) == errno.EINVAL
    assert lib.rmgr_ssim_get_version(None) == errno.EINVAL


def test_validation_order_matches_reference(lib):
    """src/ssim.cpp:962-978 and :1147-1151: all EINVAL cases are decided before any device work,
    so they are observable without a GPU."""
    a = np.zeros((8, 8), np.uint8)
    out = ctypes.c_float()
    good = ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), None, None) == errno.EINVAL
    assert lib.rmgr_ssim_compute_ssim(None, ctypes.byref(good), None) == errno.EINVAL          # both outputs NULL
    bad = ssim_amd.make_params(8, 8, None, 1, 8, a.ctypes.data, 1, 8)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(bad), None) == errno.EINVAL
    bad = ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, None, 1, 8)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(bad), None) == errno.EINVAL

    @ssim_amd.api.ThreadPoolFct
    def dispatch(ctx, fct, args, threads, jobs):
        return 0
    tp = ssim_amd.ThreadPool(dispatch, None, 0)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(good), ctypes.byref(tp)) == errno.EINVAL   # threadCount == 0
    assert lib.rmgr_ssim_hip_finalize(1, None, 8, 8, None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_enqueue_batch(None, 1, ctypes.byref(good), None) == errno.EINVAL
    # the pipelined host batch validates before it looks for a device: NULL arrays, maps, mixed sizes
    outs = (ctypes.c_float * 2)()
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 1, None, outs) == errno.EINVAL
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 1, ctypes.byref(good), None) == errno.EINVAL
    m = np.zeros((8, 8), np.float32)
    with_map = ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8, m.ctypes.data, 1, 8)
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 1, ctypes.byref(with_map), outs) == errno.EINVAL
    mixed = (ssim_amd.Params * 2)(good, ssim_amd.make_params(8, 4, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8))
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 2, mixed, outs) == errno.EINVAL
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 0, None, None) == 0


def test_finalize_is_the_reference_mean():
    # float(sum / double(width*height)) with the 32-bit product, src/ssim.cpp:1102
    sums = np.array([14989246.104541957, 0.0, 3.0])
    out = ssim_amd.finalize(sums, 4096, 4096)
    assert out[0].view(np.uint32) == 0x3f64b7be
    assert out[1] == 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        wrapped = ssim_amd.finalize(np.array([65536.0]), 65536, 65537)   # 65536*65537 mod 2^32 = 65536
    assert wrapped[0] == np.float32(1.0)


def has_gpu():
    return ssim_amd.device_count() > 0


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly_never_computes():
    a = np.full((32, 32), 9, np.uint8)
    with pytest.raises(ssim_amd.SsimError) as ei:
        ssim_amd.compute_ssim(a, a)
    assert ei.value.errno == errno.ENODEV
    with pytest.raises(ssim_amd.SsimError) as ei:
        ssim_amd.Context(0)
    assert ei.value.errno == errno.ENODEV
    with pytest.raises(ssim_amd.SsimError) as ei:                  # the one-process / several-devices entry point
        ssim_amd.compute_ssim_batch_devices([(a, a), (a, a)])
    assert ei.value.errno == errno.ENODEV
    lib = ssim_amd.load_library()
    out = (ctypes.c_float * 1)()
    p = (ssim_amd.Params * 1)(ssim_amd.make_params(32, 32, a.ctypes.data, 1, 32, a.ctypes.data, 1, 32))
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(None, 0, 9, 1, p, out) == errno.EINVAL       # bad mode: before anything else
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(None, 0, 0, 1, None, out) == errno.EINVAL


def build_dropin_client(tmp_path, static=False):
    exe = tmp_path / ("dropin_client_static" if static else "dropin_client")
    libdir = os.path.dirname(ssim_amd.LIB_PATH)
    link = ([os.path.join(libdir, "librmgr-ssim-openmp.a"), os.path.join(libdir, "librmgr-ssim.a"), "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"] if static
            else ["-L", libdir, "-lrmgr-ssim-hip", "-Wl,-rpath," + libdir])
    subprocess.run(["g++", "-std=c++98", "-pedantic", "-Wall", "-Werror", "-I", INCLUDE,
                    os.path.join(ROOT, "tests", "dropin_client.cpp"), "-o", str(exe)] + link, check=True)
    return str(exe)


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_reference_style_client_links_and_reports_enodev(tmp_path, manifest):
    """A C++98 program written against the reference's API links against the library unchanged."""
    exe = build_dropin_client(tmp_path)
    il = manifest["_interleaved"]
    out = subprocess.run([exe, os.path.join(GOLDEN, il["a"]), os.path.join(GOLDEN, il["b"]), str(il["width"]), str(il["height"]), "3"],
                         check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "version 2.1.0 2.1.0"
    assert out[1:] == ["channel %d errno %d" % (c, errno.ENODEV) for c in range(3)]


def test_install_layout_and_reference_link_line(tmp_path):
    """`make install` gives the reference's layout (headers in include/rmgr, libs in lib),
    and the reference's own link line (-lrmgr-ssim -lrmgr-ssim-openmp) resolves."""
    prefix = tmp_path / "prefix"
    subprocess.run(["make", "-C", ROOT, "install", "PREFIX=" + str(prefix)], check=True, stdout=subprocess.DEVNULL)
    for rel in ("include/rmgr/ssim.h", "include/rmgr/ssim-openmp.h", "include/rmgr/ssim-version.h", "include/rmgr/ssim-hip.h",
                "lib/librmgr-ssim-hip.so", "lib/librmgr-ssim.so", "lib/librmgr-ssim-openmp.so", "bin/rmgr-ssim", "lib/pkgconfig/rmgr-ssim.pc"):
        assert (prefix / rel).exists(), rel
    exe = tmp_path / "client"
    subprocess.run(["g++", "-std=c++98", "-I", str(prefix / "include"), os.path.join(ROOT, "tests", "dropin_client.cpp"), "-o", str(exe),
                    "-L", str(prefix / "lib"), "-lrmgr-ssim-openmp", "-lrmgr-ssim", "-Wl,-rpath," + str(prefix / "lib")], check=True)
    r = subprocess.run([str(prefix / "bin" / "rmgr-ssim"), "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage: rmgr-ssim" in r.stdout
    # the static archives under the reference's names link too (plus the HIP runtime they depend on); the client calls
    # rmgr_ssim_compute_ssim_openmp, which -- as in the reference -- lives in the second archive
    assert (prefix / "lib" / "librmgr-ssim.a").exists()
    exe_static = tmp_path / "client_static"
    subprocess.run(["g++", "-std=c++98", "-I", str(prefix / "include"), os.path.join(ROOT, "tests", "dropin_client.cpp"), "-o", str(exe_static),
                    str(prefix / "lib" / "librmgr-ssim-openmp.a"), str(prefix / "lib" / "librmgr-ssim.a"), "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    # ... and so does the reference's pair of archives by name (CMakeLists.txt:205, :229), forced static
    assert (prefix / "lib" / "librmgr-ssim-openmp.a").exists()
    exe_pair = tmp_path / "client_static_pair"
    subprocess.run(["g++", "-std=c++98", "-I", str(prefix / "include"), os.path.join(ROOT, "tests", "dropin_client.cpp"), "-o", str(exe_pair),
                    "-L", str(prefix / "lib"), "-Wl,-Bstatic", "-lrmgr-ssim-openmp", "-lrmgr-ssim", "-Wl,-Bdynamic",
                    "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True)


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_select_impl_reports_nothing_without_a_device(lib):
    """rmgr::ssim::select_impl (src/ssim.cpp:808) returns the mask of usable implementations: none when there is no GPU
    (the reference's tests then skip, tests/rmgr-ssim-tests.cpp:231-232); and the NULL-context set_mode it relies on
    says ENODEV."""
    fn = getattr(lib, "_ZN4rmgr4ssim11select_implENS0_14ImplementationE")
    fn.argtypes, fn.restype = [ctypes.c_int], ctypes.c_uint
    for impl in range(9):
        assert fn(impl) == 0
    assert lib.rmgr_ssim_hip_set_mode(None, 0) == errno.ENODEV
    assert lib.rmgr_ssim_hip_set_mode(None, 9) == errno.EINVAL


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_bench_quotes_pmc_traffic_only_for_the_kernels_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC measurement (profiles/traffic.json).  The file names the sha256 of
    the ssim_kernels.hip it was collected from, the library reports the source it was compiled from
    (rmgr_ssim_hip_get_kernel_source_id): any mismatch -> null with a note, never another kernel's bytes."""
    import hashlib
    import json
    sys.path.insert(0, ROOT)
    import bench
    kid = ssim_amd.kernel_source_id()
    src = os.path.join(ROOT, "ssim_amd", "csrc", "ssim_kernels.hip")
    assert kid == hashlib.sha256(open(src, "rb").read()).hexdigest(), "the library in ssim_amd/lib was not built from the kernel source in the tree (make lib)"
    fake = {"kernel_source_sha256": kid, "exact_4096_nomap": {"pairs": 32, "bytes_per_pair": 34000000.0}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(fake))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    v, note = bench.measured_traffic(0, "4k", 32, kid)
    assert v == 32 * 34000000.0 and "this very batch" in note
    v, note = bench.measured_traffic(0, "4k", 8, kid)
    assert v == 8 * 34000000.0 and "scaled per pair" in note
    v, note = bench.measured_traffic(0, "4k", 32, "0" * 64)               # another kernel version
    assert v is None and "not quoted" in note
    assert bench.measured_traffic(4, "4k", 32, kid)[0] is None           # another arithmetic mode: never measured
    assert bench.measured_traffic(0, "8k-map", 2, kid)[0] is None        # a configuration the file does not hold


def test_bench_refuses_to_run_without_the_gpu():
    """bench.py measures the HIP path or nothing: no device -> a loud non-zero exit, never a CPU number."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)
    assert "metric" not in r.stdout


def test_strip_plan_covers_every_image_and_fills_the_gpu():
    """rmgr_ssim_hip_get_plan: pure host arithmetic (no device needed with a NULL context).  The strips must
    tile the image exactly, and the defaults are the ones DESIGN.md section 5 documents."""
    import ssim_amd
    for (w, h, n) in [(1, 1, 1), (7, 3, 1), (128, 64, 1), (129, 65, 2), (256, 256, 1), (1920, 1080, 128), (4096, 4096, 1),
                      (4096, 4096, 32), (8192, 8192, 2), (65537, 3, 1), (3, 65537, 1), (255, 63, 1000)]:
        p = ssim_amd.get_plan(w, h, n)
        assert p.stripWidth in (64, 128) and p.stripRows >= 1
        assert (p.stripsX - 1) * p.stripWidth < w <= p.stripsX * p.stripWidth
        assert (p.stripsY - 1) * p.stripRows < h <= p.stripsY * p.stripRows
        assert p.wavefronts == p.stripsX * p.stripsY * n
    # the discrete wave-slot model (ssim_kernels.hip plan()): 2 x 1024 SIMDs = 2048 slots for the bit-exact kernel
    assert ssim_amd.get_plan(4096, 4096, 32).stripRows == 1024          # 4096 strips = two full rounds (round 5: up to 1024 rows; 512 before)
    assert ssim_amd.get_plan(4096, 4096, 1).stripRows == 64             # 2048 strips: exactly one round
    p = ssim_amd.get_plan(1920, 1080, 16)
    assert (p.stripRows, p.stripsY, p.wavefronts) == (136, 8, 1920)     # one round; 9 strips per column would be 2160 = two rounds
    p = ssim_amd.get_plan(1920, 1080, 128)
    assert p.stripRows % 8 == 0 and p.stripsY * p.stripRows >= 1080 > (p.stripsY - 1) * p.stripRows and p.stripRows <= 1024
    assert ssim_amd.get_plan(256, 256, 1).stripWidth == 64              # tiny launches: twice as many, half as wide strips
    assert ssim_amd.get_plan(1024, 1024, 1).stripWidth == 128 and ssim_amd.get_plan(256, 256, 64).stripWidth == 128
    # strips start on the boundaries of the fp64 reduction cells: 32 rows for images of >= 2048 rows, else 8
    for (w, h, n) in [(4096, 4096, 1), (4096, 4096, 32), (8192, 8192, 2), (5000, 2050, 3), (300, 2048, 1)]:
        assert ssim_amd.get_plan(w, h, n).stripRows % 32 == 0, (w, h, n)
    for (w, h, n) in [(1920, 1080, 1), (1920, 1080, 1024), (640, 480, 7), (300, 2047, 2), (129, 65, 2)]:
        assert ssim_amd.get_plan(w, h, n).stripRows % 8 == 0, (w, h, n)
    assert ssim_amd.get_plan(0, 0, 1).wavefronts == 0
    lib = ssim_amd.load_library()
    assert lib.rmgr_ssim_hip_get_plan(None, 4, 4, 1, None) == errno.EINVAL
    # the struct carries its size: a client compiled against a SHORTER Plan gets its fields and not a byte more
    assert lib.rmgr_ssim_hip_get_abi_version() == ssim_amd.ABI_VERSION == 6
    buf = (ctypes.c_uint32 * 16)(*([0xDEADBEEF] * 16))
    buf[0] = 24                                                         # RMGR_SSIM_HIP_PLAN_MIN_SIZE: structSize .. wavefronts
    assert lib.rmgr_ssim_hip_get_plan(None, 4096, 4096, 1, ctypes.cast(buf, ctypes.POINTER(ssim_amd.Plan))) == 0
    assert list(buf[:6]) == [24, 128, 64, 32, 64, 2048] and all(v == 0xDEADBEEF for v in buf[6:])
    buf[0] = 20
    assert lib.rmgr_ssim_hip_get_plan(None, 4096, 4096, 1, ctypes.cast(buf, ctypes.POINTER(ssim_amd.Plan))) == errno.EINVAL
    buf[0] = 64                                                         # a client from the future: only what this library knows is written
    assert lib.rmgr_ssim_hip_get_plan(None, 4096, 4096, 1, ctypes.cast(buf, ctypes.POINTER(ssim_amd.Plan))) == 0
    assert buf[8] == 32 and buf[9] == 64 and buf[10] == 128 and buf[11] == 2048 and buf[12] == 64 and buf[13] == 1 and all(v == 0xDEADBEEF for v in buf[14:])      # balancedChunks, balancedChunkRows, balancedInterleave: a single 4096^2 pair runs the chunks (which are strips there: no interleave)
    p = ssim_amd.get_plan(1920, 1080, 1)
    assert (p.cellRows, p.cellsX, p.cellsY) == (8, 30, 135)
    # the balanced schedule (one round of equal chunks instead of strips).  Round 5 took it where its model priced the strips 3.5 % above the chunks, for chunks of
    # at most 1100 rows and one strip column; round 6 (images interleaved in the chunk list: neighbouring columns in step, profiles/r06_phase_ab.txt) re-fitted the rule on
    # rmgr_ssim_hip_tune over 148 shapes (profiles/r06_tune_sweep.txt): wherever the model prices the best strips at or above the chunks, long launches included
    p = ssim_amd.get_plan(1920, 1080, 128)
    assert (p.wavefronts, p.balancedChunks, p.balancedChunkRows) == (3840, 2041, 1016)      # strips of 544 rows (reported; the launch runs the chunks)
    # round 6: 16 images interleaved column by column in the chunk list -- 16 x (135 mod 127) = 128 = 127 + 1: neighbouring strip columns one cell row out of step instead of eight (profiles/r06_phase_ab.txt)
    assert p.balancedInterleave == 16 and ssim_amd.get_plan(1920, 1080, 32).balancedInterleave == 9 and ssim_amd.get_plan(4096, 4096, 32).balancedInterleave == 1
    assert p.balancedChunks * p.balancedChunkRows >= 128 * 15 * 1080            # the chunks cover every row of every strip column
    for (w, h, n, chunks, rows, stride) in [(4096, 4096, 24, 2048, 1536, 3), (1920, 1080, 256, 2041, 2032, 32), (8192, 8192, 12, 2048, 3072, 3), (5120, 2880, 32, 2022, 1824, 19)]:
        p = ssim_amd.get_plan(w, h, n)                                         # what round 5's caps kept on multi-round strips: +3...12 % (profiles/r06_tune_sweep.txt)
        assert (p.balancedChunks, p.balancedChunkRows, p.balancedInterleave) == (chunks, rows, stride), (w, h, n)
    # ... and not where a launch is a few short strips per SIMD (ten warm-up rows per 16...64-row chunk: 2 x 1080p -17 %), nor where no interleave brings the
    # neighbouring columns within 16 rows and the model's margin is below 4 % (16 x 5120x2880: 32-row cells; measured -8 %)
    for (w, h, n) in [(1920, 1080, 1), (256, 256, 1), (1920, 1080, 2), (1000, 1000, 4), (1920, 1080, 8), (5120, 2880, 16)]:
        assert ssim_amd.get_plan(w, h, n).balancedChunks == 0, (w, h, n)
    # ... and where the chunk divides the strip column evenly (the chunks are then strips: the tallest that fill the wave slots in one round): the headline batch,
    # a single 4096^2 pair, 8192^2 pairs -- +0.6 ... +2.4 % over the strips on every such shape (profiles/r05_strip_cap_sweep.txt)
    for (w, h, n, chunks, rows) in [(4096, 4096, 32, 2048, 2048), (4096, 4096, 1, 2048, 64), (8192, 8192, 2, 2048, 512), (4096, 4096, 64, 2048, 4096), (512, 512, 16, 2048, 16)]:
        p = ssim_amd.get_plan(w, h, n)
        assert (p.balancedChunks, p.balancedChunkRows) == (chunks, rows), (w, h, n)


def test_kernels_keep_two_waves_per_simd_and_never_spill():
    """Build-time guard for the register cliff: the accumulator rings put every strip kernel
    near the 256-VGPR limit of 2 waves/SIMD; one register more halves the speed (measured 122 vs 208 Gpix/s), a
    spill is far worse.  Compile the kernels with the compiler's resource remarks and check each variant."""
    src = os.path.join(ROOT, "ssim_amd", "csrc", "ssim_kernels.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-I" + INCLUDE, "-I" + os.path.dirname(src), "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
        for key in ("VGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]"):
            m = re.search(r"remark:\s+" + key + r": (\d+)", line)
            if m and name:
                kernels[name][key.split(" ")[0]] = int(m.group(1))
    strip = {k: v for k, v in kernels.items() if "ssim_strip" in k}
    # two-column kernel: 4 fp32 modes x {no map, map, map with 8-byte stores} + the EARLY form of the two bit-exact modes + their
    # balanced-schedule form (no map, EARLY) + MODE_FAST's (no map); one-column kernel: 5 modes x {no map, map} x {64-bit, 32-bit addressing}
    assert len(strip) == 41, sorted(kernels)
    for k, v in strip.items():
        assert v["ScratchSize"] == 0, (k, v)
        assert v["VGPRs"] <= 256 and v["Occupancy"] >= 2, (k, v)
        if "strip2_kernelILi4E" in k or "strip1_kernelILi2E" in k or "strip1_kernelILi4E" in k:      # MODE_SEPARABLE and MODE_DOUBLE: three waves per SIMD
            assert v["VGPRs"] <= 168 and v["Occupancy"] >= 3, (k, v)
        assert v["LDS"] <= 13312, (k, v)                # 12 resident waves per CU (3 per SIMD) must fit the 160 KiB of LDS: row slots + 4 KiB cell batch


# ---- the header's public macro surface (reference include/rmgr/ssim.h:28-376) --------------------------------------
MACRO_TU = os.path.join(ROOT, "tests", "src", "macro_surface.c")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
# every macro name the reference header defines for its includers (collected from the reference header; the names are API)
REFERENCE_MACROS = """RMGR_COMPILER_IS_NOT_DOXYGEN RMGR_COMPILER_IS_DOXYGEN RMGR_COMPILER_IS_CLANG RMGR_COMPILER_IS_MSVC RMGR_COMPILER_IS_GCC
RMGR_COMPILER_IS_GCC_OR_CLANG RMGR_COMPILER_VERSION_MAJOR RMGR_COMPILER_VERSION_MINOR RMGR_COMPILER_VERSION_PATCH
RMGR_COMPILER_VERSION_IS_AT_LEAST RMGR_COMPILER_IS_CLANG_AT_LEAST RMGR_COMPILER_IS_MSVC_AT_LEAST RMGR_COMPILER_IS_GCC_AT_LEAST
RMGR_COMPILER_IS_CLANG_LESS_THAN RMGR_COMPILER_IS_MSVC_LESS_THAN RMGR_COMPILER_IS_GCC_LESS_THAN RMGR_WARNING_PUSH RMGR_WARNING_POP
RMGR_WARNING_MSVC_DISABLE RMGR_WARNING_GCC_DISABLE RMGR_WARNING_CLANG_DISABLE RMGR_DEPRECATED RMGR_DEPRECATED_MSG RMGR_ALIGNED_VAR
RMGR_ARCH_IS_X86_32 RMGR_ARCH_IS_X86_64 RMGR_ARCH_IS_X86_ANY RMGR_ARCH_IS_ARM_32 RMGR_ARCH_IS_ARM_64 RMGR_ARCH_IS_ARM_ANY
RMGR_ARCH_IS_LITTLE_ENDIAN RMGR_ARCH_IS_BIG_ENDIAN RMGR_CPP_VERSION RMGR_NOEXCEPT RMGR_NOEXCEPT_TYPEDEF RMGR_FORCEINLINE RMGR_NOINLINE
RMGR_COMPILER_SUPPORTS_ARM_NEON RMGR_UINT8_MAX RMGR_SSIM_H""".split()


@pytest.mark.parametrize("cc", ["gcc -std=c89 -x c", "g++ -std=c++98 -x c++", "g++ -std=c++17 -x c++",
                                CLANG + " -std=c89 -x c", CLANG + "++ -std=c++98 -x c++", CLANG + "++ -std=c++17 -x c++"])
def test_macro_surface_compiles_pedantic_and_runs(tmp_path, cc):
    """every macro the reference header exports, used the way its callers use them
    (src/ssim-cli.cpp:46-59,:357; sample/rmgr-ssim-sample.cpp:33-38; tests/rmgr-ssim-tests.cpp:35-52,:67,:486,:497;
    tests/ssim_naive.h:28,:60), as C89, C++98 and C++17 under -pedantic -Werror, with gcc and clang."""
    if not os.path.exists(cc.split()[0]) and "/" in cc.split()[0]:
        pytest.skip("no " + cc.split()[0])
    obj = str(tmp_path / "ms.o")
    r = subprocess.run(cc.split() + ["-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + INCLUDE, "-c", MACRO_TU, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    main = tmp_path / "main.c"
    main.write_text("#ifdef __cplusplus\nextern \"C\"\n#endif\nint macro_surface_selftest(void);\nint main(void) { return macro_surface_selftest(); }\n")
    exe = str(tmp_path / "ms")
    r = subprocess.run(cc.split() + [str(main), "-x", "none", obj, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert subprocess.run([exe]).returncode == 0


def _macro_dump(header_dir, cc, tmp_path, tag):
    """the value each macro of REFERENCE_MACROS expands to with the header under header_dir (object-like ones evaluated
    by the preprocessor where they are integer expressions, everything else as expansion text)"""
    src = tmp_path / ("dump_%s.c" % tag)
    lines = ["#include <rmgr/ssim.h>"]
    for m in REFERENCE_MACROS:
        lines.append("#ifdef %s\nDEFINED \"%s\"\n#else\nUNDEFINED \"%s\"\n#endif" % (m, m, m))
    for m in ("RMGR_COMPILER_IS_DOXYGEN", "RMGR_COMPILER_IS_CLANG", "RMGR_COMPILER_IS_MSVC", "RMGR_COMPILER_IS_GCC", "RMGR_COMPILER_IS_GCC_OR_CLANG",
              "RMGR_ARCH_IS_X86_32", "RMGR_ARCH_IS_X86_64", "RMGR_ARCH_IS_X86_ANY", "RMGR_ARCH_IS_ARM_32", "RMGR_ARCH_IS_ARM_64", "RMGR_ARCH_IS_ARM_ANY",
              "RMGR_ARCH_IS_LITTLE_ENDIAN", "RMGR_ARCH_IS_BIG_ENDIAN", "RMGR_COMPILER_SUPPORTS_ARM_NEON", "RMGR_COMPILER_VERSION_IS_AT_LEAST(1,2,3)",
              "RMGR_COMPILER_VERSION_IS_AT_LEAST(99,0,0)", "RMGR_COMPILER_IS_GCC_AT_LEAST(5,1,0)", "RMGR_COMPILER_IS_GCC_LESS_THAN(5,1,0)",
              "RMGR_COMPILER_IS_CLANG_AT_LEAST(5,1,0)", "RMGR_COMPILER_IS_CLANG_LESS_THAN(5,1,0)", "RMGR_COMPILER_IS_MSVC_AT_LEAST(5,1,0)",
              "RMGR_COMPILER_IS_MSVC_LESS_THAN(5,1,0)"):
        lines.append("#if %s\nTRUE \"%s\"\n#else\nFALSE \"%s\"\n#endif" % (m, m, m))
    for m in ("RMGR_COMPILER_VERSION_MAJOR", "RMGR_COMPILER_VERSION_MINOR", "RMGR_COMPILER_VERSION_PATCH", "RMGR_CPP_VERSION", "RMGR_NOEXCEPT",
              "RMGR_NOEXCEPT_TYPEDEF", "RMGR_WARNING_GCC_DISABLE(\"-Wx\")", "RMGR_WARNING_CLANG_DISABLE(\"-Wx\")", "RMGR_WARNING_MSVC_DISABLE(1)",
              "RMGR_DEPRECATED", "RMGR_DEPRECATED_MSG(\"m\")", "RMGR_ALIGNED_VAR(16, int, v)", "RMGR_NOINLINE"):
        lines.append("EXPANDS \"%s\" = %s" % (m.replace("\"", "'"), m))
    src.write_text("\n".join(lines) + "\n")
    r = subprocess.run(cc.split() + ["-E", "-P", "-I" + header_dir, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    keep = [l.strip() for l in r.stdout.splitlines() if l.strip().split(" ")[0] in ("DEFINED", "UNDEFINED", "TRUE", "FALSE", "EXPANDS")]
    return keep


@pytest.mark.parametrize("cc", ["gcc -std=c99 -x c", "g++ -std=c++98 -x c++", "g++ -std=c++17 -x c++", CLANG + "++ -std=c++17 -x c++",
                                # other targets, preprocessor only (freestanding: the compiler's own stddef.h / stdint.h)
                                CLANG + " -std=c99 -x c -ffreestanding --target=aarch64-none-elf",
                                CLANG + " -std=c99 -x c -ffreestanding --target=armv7-none-eabi -mfpu=neon",
                                CLANG + " -std=c99 -x c -ffreestanding --target=i686-none-elf",
                                CLANG + " -std=c99 -x c -ffreestanding --target=powerpc64-none-elf",
                                "g++ -std=c++17 -x c++ -DRMGR_ARCH_IS_BIG_ENDIAN=1 -DRMGR_COMPILER_IS_GCC=0"])
def test_macro_surface_has_the_reference_values(tmp_path, cc):
    """Where the reference tree is present (build container): every macro is defined by both headers or by neither, the
    integer-valued ones evaluate alike and the attribute / pragma ones expand to the same tokens."""
    ref = "/root/reference/include"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    if not os.path.exists(cc.split()[0]) and "/" in cc.split()[0]:
        pytest.skip("no " + cc.split()[0])
    # the reference header needs the cmake-generated <rmgr/ssim-version.h>?  no: only src/ssim.cpp includes it
    ours = _macro_dump(INCLUDE, cc, tmp_path, "ours")
    theirs = _macro_dump(ref, cc, tmp_path, "ref")
    assert len(ours) > 60
    squeeze = lambda l: re.sub(r"\s+", "", l).replace("__inline__", "inline")   # RMGR_FORCEINLINE here is also valid in C89
    diff = [(a, b) for a, b in zip(ours, theirs) if squeeze(a) != squeeze(b)]
    assert not diff and len(ours) == len(theirs), diff[:10]


def test_reference_naive_header_compiles_against_this_header(tmp_path):
    """The reference's header-only test oracle (tests/ssim_naive.h) includes <rmgr/ssim.h> and uses RMGR_COMPILER_IS_GCC /
    RMGR_NOEXCEPT: it must compile, and instantiate, against THIS repository's header (build container only)."""
    if not os.path.exists("/root/reference/tests/ssim_naive.h"):
        pytest.skip("reference tree not present (GPU box)")
    tu = tmp_path / "naive_tu.cpp"
    tu.write_text("#include <ssim_naive.h>\n"
                  "double probe(const rmgr::ssim::GeneralParams& p, double* map)\n"
                  "{ return rmgr::ssim::naive::compute_ssim<double, rmgr::ssim::uint8_t>(p.width, p.height, p.imgA.topLeft, p.imgA.step, p.imgA.stride,\n"
                  "      p.imgB.topLeft, p.imgB.step, p.imgB.stride, map, 1, p.width); }\n")
    # the reference's OpenMP adapter (src/ssim-openmp.c: includes <rmgr/ssim-openmp.h>, builds a rmgr_ssim_ThreadPool and calls
    # rmgr_ssim_compute_ssim) is the one reference-side CALLER that needs no un-vendored dependency: it must compile unchanged
    r = subprocess.run(["gcc", "-std=c99", "-fopenmp", "-fsyntax-only", "-Wall", "-Werror", "-I" + INCLUDE, "/root/reference/src/ssim-openmp.c"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for std in ("gnu++98", "gnu++17"):      # the __float128 literals of ssim_naive.h:72 need the GNU dialect, with the reference header too
        r = subprocess.run(["g++", "-std=" + std, "-D_USE_MATH_DEFINES", "-fsyntax-only", "-Wall", "-I" + INCLUDE, "-I/root/reference/tests", str(tu)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("double", [False, True])
def test_cmake_consumer_links_the_reference_target_names(tmp_path, double):
    """Projects consume the reference with add_subdirectory() + target_link_libraries(app rmgr-ssim rmgr-ssim-openmp)
    (reference CMakeLists.txt:205, :229).  The same two lines pointed at this repository must configure, build the C++98
    client and link it against the HIP library (RMGR_SSIM_USE_DOUBLE, the reference's option, picks the fp64 flavour)."""
    import shutil
    if not shutil.which("cmake"):
        pytest.skip("no cmake")
    src = tmp_path / "src"
    src.mkdir()
    (src / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.16)\nproject(consumer CXX)\n"
        "add_subdirectory(%s ssim)\nadd_executable(client %s)\n"
        "set_target_properties(client PROPERTIES CXX_STANDARD 98 CXX_EXTENSIONS OFF)\n"
        "target_link_libraries(client PRIVATE rmgr-ssim-openmp rmgr-ssim)\n" % (ROOT, os.path.join(ROOT, "tests", "dropin_client.cpp")))
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    r = subprocess.run(["cmake", "-S", str(src), "-B", str(tmp_path / "build")] + gen + (["-DRMGR_SSIM_USE_DOUBLE=ON"] if double else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(["cmake", "--build", str(tmp_path / "build")], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = subprocess.run(["ldd", str(tmp_path / "build" / "client")], capture_output=True, text=True).stdout
    assert ("librmgr-ssim-hip-double.so" if double else "librmgr-ssim-hip.so") in out, out


def test_comm_entry_points_without_a_device(lib):
    """The RCCL exchange's entry points resolve, validate their arguments, and -- with no GPU for RCCL to bootstrap on --
    fail promptly with an errno instead of hanging (the deadline machinery is exercised on the GPU, tests/test_gpu_zz_rccl.py)."""
    import time
    text = ssim_amd.Context.comm_describe()
    assert text.startswith("rccl"), text
    n = ctypes.c_int32(7)
    assert lib.rmgr_ssim_hip_comm_rank_count(None, ctypes.byref(n)) == errno.EINVAL
    assert lib.rmgr_ssim_hip_comm_init(None, b"\0" * 128, 1, 0) == errno.EINVAL
    assert lib.rmgr_ssim_hip_comm_allreduce_sums(None, None, 0) == errno.EINVAL
    assert lib.rmgr_ssim_hip_comm_destroy(None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_comm_get_unique_id(None) == errno.EINVAL
    if ssim_amd.device_count() == 0:
        t = time.time()
        rc = lib.rmgr_ssim_hip_comm_get_unique_id(ctypes.create_string_buffer(128))
        # 0: an RCCL that is already in the process (a test that imported torch ran before this one) hands out an id without touching a device
        assert rc in (0, errno.ECHILD, errno.ENODEV, errno.ENOSYS, errno.ETIMEDOUT) and time.time() - t < 40, rc


def test_plan_invariants_over_random_shapes():
    """rmgr_ssim_hip_get_plan on 3000 random launch shapes (pure host arithmetic): the strips tile the image, start on reduction-cell boundaries and are at most 1024
    rows tall; a balanced plan's chunks cover every cell row of every strip column exactly once in at most waveSlots wavefronts (round 6: of any length -- round 5's
    caps of 1100 rows and one column per chunk went with the over-fetch they were fitted on), and the interleave of the images is within its bounds."""
    import random
    import ssim_amd
    rnd = random.Random(20261007)
    taken = even = long_chunks = 0
    for i in range(3000):
        kind = rnd.randrange(3)
        if kind == 0:
            w, h = 2 ** rnd.randrange(5, 14), 2 ** rnd.randrange(5, 14)
        elif kind == 1:
            w, h = rnd.choice([(640, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320), (1600, 1200), (4096, 2160)])
        else:
            w, h = rnd.randrange(1, 9000), rnd.randrange(1, 9000)
        n = rnd.choice([1, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 1024])
        p = ssim_amd.get_plan(w, h, n)
        cr = p.cellRows
        assert cr == (32 if h >= 2048 else 8) and p.cellsX == (w + 63) // 64 and p.cellsY == (h + cr - 1) // cr
        assert p.stripWidth in (64, 128) and 1 <= p.stripRows <= max(1024, cr) and p.stripRows % cr == 0
        assert (p.stripsX - 1) * p.stripWidth < w <= p.stripsX * p.stripWidth and (p.stripsY - 1) * p.stripRows < h <= p.stripsY * p.stripRows
        assert p.wavefronts == p.stripsX * p.stripsY * n and p.waveSlots in (2048, 3072, 4096)
        if p.balancedChunks:
            taken += 1
            assert p.stripWidth == 128 and p.balancedChunkRows % cr == 0
            col_cells, chunk = p.cellsY, p.balancedChunkRows // cr
            cells = n * p.stripsX * col_cells
            assert cells > p.waveSlots and p.balancedChunks <= p.waveSlots
            assert (p.balancedChunks - 1) * chunk < cells <= p.balancedChunks * chunk          # every cell row of every strip column, once
            assert 1 <= p.balancedInterleave <= min(n, 64), (w, h, n, p.balancedInterleave)
            if col_cells % chunk == 0:
                even += 1
                assert p.balancedInterleave == 1                                                # chunks that ARE strips are in step already
            elif chunk > col_cells:
                long_chunks += 1
        else:
            assert p.balancedInterleave == 0
    assert taken > 300 and even > 100 and long_chunks > 20, (taken, even, long_chunks)           # the draw does exercise every way into the balanced form
