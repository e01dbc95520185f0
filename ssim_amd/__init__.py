"""ssim_amd -- MI355X-native SSIM engine, drop-in for rmgr::ssim::compute_ssim (romigrou/ssim 2.1.0).

The product is the C/C++ library ssim_amd/lib/librmgr-ssim-hip.so (sources in ssim_amd/csrc,
public headers in include/rmgr).  This Python package only binds its C ABI for tests and bench.py.
"""
from .api import (  # noqa: F401
    C_SYMBOLS, CXX_SYMBOLS, LIB_PATH, MODE_DOUBLE, MODE_EXACT, MODE_FAST, MODE_SEPARABLE, MODE_UNFUSED,
    ABI_VERSION, Context, DeviceBuffer, ImgParams, Params, Plan, SsimError, ThreadPool, Version,
    TuneResult, TunedEntry, compute_ssim, compute_ssim_batch, compute_ssim_batch_devices, compute_ssim_channels, compute_ssim_luminance, default_pool, default_pool_memory, device_count, finalize,
    get_plan, get_version, kernel_source_id, load_library, make_params, memory_info, trim_default_pool,
)
