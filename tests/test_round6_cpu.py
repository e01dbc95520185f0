"""Round 6, no GPU: the experiment archive still applies, the RCCL retry policy recognises environmental failures only, the interleaved chunk
list is a bijection, the new entry points validate their arguments without a device."""
import ctypes
import errno
import glob
import importlib.util
import os
import subprocess

import numpy as np

import ssim_amd
from conftest import ROOT


def test_archived_kernel_patches_still_apply():
    """tools/*.patch are kernels that were built, measured and NOT adopted (profiles/README.md says what each measured).  They are kept only while
    they apply to the kernel source in the tree; one that no longer does is deleted (its record stays in profiles/, the patch in the history)."""
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "*.patch")))
    assert patches, "no archived patches: drop this test with the last of them"
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, "%s no longer applies to ssim_kernels.hip: refresh or delete it\n%s" % (os.path.basename(p), r.stderr[-600:])


def test_one_parameterised_sweep_script_replaces_the_round5_scripts():
    assert not glob.glob(os.path.join(ROOT, "tools", "r5_*.sh"))
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "sweep_variants.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "phase_ab.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_rccl_retry_only_on_environmental_signatures():
    """tests/test_gpu_zz_rccl.py retries a failed selftest only when its log says the BOX failed it."""
    spec = importlib.util.spec_from_file_location("zz_rccl", os.path.join(ROOT, "tests", "test_gpu_zz_rccl.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sig = m.environmental_signature
    assert sig("NCCL WARN Call to bind failed: Address already in use", 50)
    assert sig("[rccl_selftest   0.100 s] import ssim_amd\n[rmgr-ssim comm] helper: loading librccl\n", 50)                                   # killed inside the load
    assert sig("[rmgr-ssim comm] helper: loading librccl\n[rmgr-ssim comm] /opt/rocm/lib/librccl.so.1 (31.200 s)\n", 50)                       # the load ate the limit
    assert sig("[rccl_selftest   0.100 s] import torch (bundled HIP runtime + RCCL)\n", 50)
    healthy_load = "[rmgr-ssim comm] helper: loading librccl\n[rmgr-ssim comm] /opt/rocm/lib/librccl.so.1 (1.200 s)\n[rmgr-ssim comm] helper: ncclGetUniqueId returned (1.3 s)\n"
    assert sig(healthy_load + "[rccl_selftest   5.000 s] comm_init as rank 1 of 2 -- rank 0 never arrives; deadline 5 s\n", 50) is None        # a deadline that did not fire: NOT retried
    assert sig(healthy_load + "[rccl_selftest   2.000 s] comm_allreduce_sums x 3 + synchronize\nAssertionError: sums changed", 50) is None


def list_column(col, count, strips_x, stride):
    """ssim_kernels.hip list_column(): list position -> (image, strip column) with `stride` images interleaved column by column."""
    if stride <= 1:
        return col // strips_x, col % strips_x
    span = stride * strips_x
    blk, within = divmod(col, span)
    here = min(stride, count - blk * stride)
    sx, i = divmod(within, here)
    return blk * stride + i, sx


def test_interleaved_chunk_list_is_a_bijection():
    """Every (image, strip column) appears exactly once in the interleaved list, for any interleave -- full blocks, a shorter last block, more
    images interleaved than there are -- and neighbouring columns of an image are `stride` positions apart inside a block."""
    rng = np.random.default_rng(6)
    for _ in range(300):
        count, strips_x, stride = int(rng.integers(1, 70)), int(rng.integers(1, 40)), int(rng.integers(1, 70))
        stride = min(stride, count)                       # plan() caps the interleave at the batch size
        seen = {list_column(p, count, strips_x, stride) for p in range(count * strips_x)}
        assert seen == {(i, x) for i in range(count) for x in range(strips_x)}, (count, strips_x, stride)
    pos = {list_column(p, 32, 15, 9): p for p in range(32 * 15)}
    assert pos[(3, 8)] - pos[(3, 7)] == 9 and pos[(30, 8)] - pos[(30, 7)] == 5      # the last block holds 32 - 27 = 5 images


def test_round6_entry_points_validate_without_a_device():
    lib = ssim_amd.load_library()
    t = ctypes.c_double()
    assert lib.rmgr_ssim_hip_probe_valu(None, 2, 0, 5, ctypes.byref(t), None, None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_get_profile_clock(None, ctypes.byref(t), None, None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_tune(None, 64, 64, 1, 0, None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_clear_tuned(None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_get_tuned(None, 0, ctypes.byref(ssim_amd.TunedEntry())) == errno.EINVAL and lib.rmgr_ssim_hip_set_tuned(None, 64, 64, 1, 0, 2, 64) == errno.EINVAL
    assert lib.rmgr_ssim_hip_trim_default_pool() == 0                     # nothing exists: nothing to trim
    d, p, c = ssim_amd.default_pool_memory()
    assert (d, p) == (0, 0) and c == 256 << 20                            # creates nothing; the default cap
    if ssim_amd.device_count() == 0:
        f, tot = ctypes.c_uint64(), ctypes.c_uint64()
        assert lib.rmgr_ssim_hip_get_memory_info(None, ctypes.byref(f), ctypes.byref(tot)) == errno.ENODEV
        # select_impl's path: set_mode / get_mode on the default contexts answer ENODEV without a device, and do not hang
        m = ctypes.c_int32()
        assert lib.rmgr_ssim_hip_set_mode(None, 0) == errno.ENODEV and lib.rmgr_ssim_hip_get_mode(None, ctypes.byref(m)) == errno.ENODEV


def test_bench_package_power_summary_and_box_fractions():
    """bench.py's round-6 helpers on canned inputs: the rocm-smi summary of the sustained leg, and the box-relative VALU fractions (best probe sample per occupancy; per clock)."""
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ps = bench.PowerSampler()
    mk = lambda p, c: {"Current Socket Graphics Package Power (W)": str(p), "Max Graphics Package Power (W)": "1400.0", "sclk clock speed:": "(%dMhz)" % c, "Temperature (Sensor junction) (C)": "57.0"}
    ps.samples = [mk(300, 111), mk(1350, 2240), mk(1362, 2252), mk(1369, 2246)]          # the first sample predates the load and is dropped
    out = ps.summary()
    assert out["samples"] == 3 and out["power_w_median"] == 1362.0 and out["power_cap_w"] == 1400.0 and out["sclk_mhz_median"] == 2246.0 and out["power_frac_of_cap"] == round(1362.0 / 1400.0, 4)
    assert bench.PowerSampler().summary() == {}                                             # no rocm-smi: no `package` object
    valu = {"achieved": 58.0}
    samples = [{2: (52.0, 2350.0, 2330.0), 8: (73.0, 2340.0, 2320.0)}, {2: (66.0, 2360.0, 2340.0), 8: (73.5, 2350.0, 2330.0)}]      # the first two-wave sample ran degraded
    bench.against_box(valu, 0, samples, (2200.0, 2150.0))
    assert valu["box_peak_2wave"] == 66.0 and valu["box_peak_8wave"] == 73.5 and valu["frac_of_box_peak_at_kernel_occupancy"] == round(58.0 / 66.0, 4)
    k = 58.0e12 / 2200.0e6 / 32768.0
    p = 66.0e12 / 2360.0e6 / 32768.0
    assert valu["frac_of_issue_peak_per_clock"] == round(k, 4) and valu["frac_of_box_peak_per_clock"] == round(k / p, 4) and valu["slowest_xcd_mhz_during_timed_launches"] == 2150.0
    v2 = {"achieved": 30.0}
    assert bench.against_box(v2, 2, samples, (2200.0, 2150.0)) == {"achieved": 30.0}       # fp64 internals: the packed-fp32 stream is not its yardstick
