"""GPU: luminance_kernel (BT.601 integer luminance of the reference's CLI, src/ssim-cli.cpp:158-186)
for every pixel layout rmgr_ssim_hip_luminance_device accepts: RGB and RGBA through the four-pixels-per-thread form at
EVERY byte alignment of base address and row pitch (dword accesses at odd addresses), widths that are not multiples of
four, bottom-up rows, and other steps through the per-pixel form.  Integer arithmetic: exact equality with numpy."""
import ctypes

import numpy as np
import pytest

import ssim_amd

pytestmark = pytest.mark.gpu


def bt601(px):
    px = px.astype(np.uint32)
    return ((px[..., 0] * 19595 + px[..., 1] * 38470 + px[..., 2] * 7471 + 32768) >> 16).astype(np.uint8)


def run(ctx, host, base, step, stride, w, h, dst_pad, dst_off):
    """host: the 1-D source buffer; pixel (x, y) at host[base + y * stride + x * step + c]."""
    src = ctx.upload(host)
    dpitch = w + dst_pad
    dst = ctx.alloc(dst_off + dpitch * h + 8)
    try:
        dst.upload(np.full(dst_off + dpitch * h + 8, 0xA5, np.uint8))
        ssim_amd.api._check("rmgr_ssim_hip_luminance_device", ctx.lib.rmgr_ssim_hip_luminance_device(
            ctx.handle, ctypes.c_void_p(dst.ptr + dst_off), dpitch, ctypes.c_void_p(src.ptr + base), step, stride, w, h))
        ctx.synchronize()
        out = dst.download(np.uint8, (dst_off + dpitch * h + 8,))
    finally:
        src.free(); dst.free()
    return out, dpitch


@pytest.mark.parametrize("step", [3, 4, 5, 7])
def test_luminance_every_alignment_width_and_pitch(gpu_ctx, step):
    rng = np.random.default_rng(step)
    for w, h in ((1, 3), (3, 2), (4, 5), (5, 1), (64, 7), (257, 9), (1023, 4), (1920, 16)):
        for src_off in (0, 1, 2, 3):
            for pad in (0, 1, 2, 3, 13):
                stride = w * step + pad
                host = rng.integers(0, 256, src_off + stride * h + 16, dtype=np.uint8)
                dst_off, dst_pad = int(rng.integers(0, 4)), int(rng.integers(0, 4))
                out, dpitch = run(gpu_ctx, host, src_off, step, stride, w, h, dst_pad, dst_off)
                idx = src_off + np.arange(h)[:, None, None] * stride + np.arange(w)[None, :, None] * step + np.arange(3)[None, None, :]
                want = bt601(host[idx])
                got = out[dst_off:dst_off + dpitch * h].reshape(h, dpitch)
                assert np.array_equal(got[:, :w], want), (step, w, h, src_off, pad, dst_off, dst_pad)
                # nothing outside the destination rows' pixels is written
                assert np.all(out[:dst_off] == 0xA5) and np.all(out[dst_off + dpitch * h:] == 0xA5)
                assert np.all(got[:, w:] == 0xA5), (step, w, h, src_off, pad)


@pytest.mark.parametrize("step", [3, 4])
def test_luminance_bottom_up_rows(gpu_ctx, step):
    rng = np.random.default_rng(40 + step)
    w, h = 333, 21
    stride = w * step + 2
    host = rng.integers(0, 256, stride * h + 16, dtype=np.uint8)
    base = (h - 1) * stride                                  # first row of the image = last row of the buffer
    out, dpitch = run(gpu_ctx, host, base, step, -stride, w, h, 0, 0)
    idx = base - np.arange(h)[:, None, None] * stride + np.arange(w)[None, :, None] * step + np.arange(3)[None, None, :]
    assert np.array_equal(out[:w * h].reshape(h, w), bt601(host[idx]))


def test_rgba_source_offset_into_the_pixel_reads_nothing_past_its_three_bytes(gpu_ctx):
    """rmgr_ssim_hip_luminance_device may read bytes 0..2 of a pixel only.  With srcStep == 4 and the source pointer offset
    into the pixel (ARGB: src = base + 1) the image's last pixel ends at the buffer's last byte: the four-pixels-per-thread form must
    not load that pixel as a dword.  The allocation is EXACTLY the image (a stray read past it is at best invisible: so the values are
    checked for widths that end on a full quad, the case that used to take the dword path, and the test mainly documents the contract)."""
    rng = np.random.default_rng(4242)
    for w, h in ((4, 1), (8, 3), (64, 5), (1920, 2)):
        host = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)          # A R G B per pixel, exactly sized
        src = gpu_ctx.upload(host)
        dst = gpu_ctx.alloc(w * h)
        try:
            # pixels are (R, G, B) at base + 1: the last pixel's three bytes are the buffer's last three
            ssim_amd.api._check("rmgr_ssim_hip_luminance_device", gpu_ctx.lib.rmgr_ssim_hip_luminance_device(
                gpu_ctx.handle, ctypes.c_void_p(dst.ptr), w, ctypes.c_void_p(src.ptr + 1), 4, 4 * w, w, h))
            gpu_ctx.synchronize()
            got = dst.download(np.uint8, (h, w))
        finally:
            src.free(); dst.free()
        px = host.reshape(h, w, 4)[..., 1:4]
        assert np.array_equal(got, bt601(px)), (w, h)
