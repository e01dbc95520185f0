"""The rmgr-ssim command-line tool (reference src/ssim-cli.cpp).

CPU: the built-in image codecs decode what PIL encodes (PNG incl. the reference's own test images
when present, PNM, BMP, TGA) and the argument handling mirrors the reference.
GPU: printed values and written maps agree with the golden vectors / the oracle.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, f32_hex

CLI = os.environ.get("RMGR_SSIM_CLI") or os.path.join(ROOT, "ssim_amd", "bin", "rmgr-ssim")     # RMGR_SSIM_CLI: e.g. an ASan/UBSan build of the tool


def run(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True)


def write_png(path, arr, filter_type=1):
    """Minimal PNG encoder (zlib from the stdlib) with a chosen row filter, to exercise the CLI's defilter code."""
    arr = np.ascontiguousarray(arr)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    h, w, c = arr.shape
    rows = []
    prev = np.zeros((w, c), np.int16)
    for y in range(h):
        cur = arr[y].astype(np.int16)
        left = np.vstack([np.zeros((1, c), np.int16), cur[:-1]])
        upleft = np.vstack([np.zeros((1, c), np.int16), prev[:-1]])
        if filter_type == 0:
            pred = 0
        elif filter_type == 1:
            pred = left
        elif filter_type == 2:
            pred = prev
        elif filter_type == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        rows.append(bytes([filter_type]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(b"".join(rows), 6)) + chunk(b"IEND", b""))


def rgb_fixture(manifest):
    il = manifest["_interleaved"]
    w, h = il["width"], il["height"]
    a = np.fromfile(os.path.join(GOLDEN, il["a"]), np.uint8).reshape(h, w, 3)
    b = np.fromfile(os.path.join(GOLDEN, il["b"]), np.uint8).reshape(h, w, 3)
    return a, b


def test_cli_exists_and_prints_help():
    r = run("-h")
    assert r.returncode == 0 and r.stdout.startswith("Usage: rmgr-ssim [options] img1 img2 [map]")
    r = run()
    assert r.returncode != 0 and "Usage:" in r.stderr


def test_codecs_roundtrip(tmp_path, manifest):
    a, _ = rgb_fixture(manifest)
    h, w, _ = a.shape
    cases = {}
    for ft in range(5):
        p = str(tmp_path / ("f%d.png" % ft))
        write_png(p, a, ft)
        cases[p] = a
    p = str(tmp_path / "gray.png")
    write_png(p, a[:, :, 1], 4)
    cases[p] = a[:, :, 1:2]
    p = str(tmp_path / "rgba.png")
    rgba = np.dstack([a, 255 - a[:, :, :1]])
    write_png(p, rgba, 3)
    cases[p] = rgba
    p = str(tmp_path / "a.ppm")
    open(p, "wb").write(b"P6\n# comment\n%d %d\n255\n" % (w, h) + a.tobytes())
    cases[p] = a
    p = str(tmp_path / "a.pgm")
    open(p, "wb").write(b"P5 %d %d 255\n" % (w, h) + a[:, :, 0].tobytes())
    cases[p] = a[:, :, :1]
    p = str(tmp_path / "a_ascii.ppm")
    open(p, "w").write("P3\n%d %d\n255\n" % (w, 2) + " ".join(str(v) for v in a[:2].ravel()) + "\n")
    cases[p] = a[:2]
    try:
        from PIL import Image
        for ext in ("png", "bmp", "tga"):
            p = str(tmp_path / ("pil." + ext))
            Image.fromarray(a).save(p)
            cases[p] = a
        p = str(tmp_path / "pil_gray.png")
        Image.fromarray(a[:, :, 0]).save(p, optimize=True)
        cases[p] = a[:, :, :1]
    except ImportError:
        pass
    for path, want in cases.items():
        out = str(tmp_path / "dump.raw")
        r = run("--decode", path, out)
        assert r.returncode == 0, (path, r.stderr)
        ww, hh, cc = map(int, r.stdout.split())
        got = np.fromfile(out, np.uint8).reshape(hh, ww, cc)
        assert got.shape == want.shape and np.array_equal(got, want), path


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/images"), reason="reference tree not present")
def test_decodes_reference_test_pngs(tmp_path):
    from PIL import Image
    for name in ("einstein.png", "blur.png", "big_buck_bunny_360_07806.png"):
        src = "/root/reference/tests/images/" + name
        out = str(tmp_path / "d.raw")
        r = run("--decode", src, out)
        assert r.returncode == 0, r.stderr
        w, h, c = map(int, r.stdout.split())
        want = np.array(Image.open(src))
        got = np.fromfile(out, np.uint8).reshape(h, w, c)
        assert np.array_equal(got.reshape(want.shape), want), name


def decode_with_cli(tmp_path, path):
    out = str(tmp_path / "jd.raw")
    r = run("--decode", path, out)
    assert r.returncode == 0, (path, r.stderr)
    w, h, c = map(int, r.stdout.split())
    return np.fromfile(out, np.uint8).reshape((h, w, c) if c > 1 else (h, w))


def test_jpeg_decoder_matches_libjpeg(tmp_path, manifest):
    """Baseline and progressive Huffman JPEG, 4:4:4 / 4:2:2 / 4:2:0, greyscale, odd sizes, restart intervals,
    optimised tables, extreme qualities: every decoded sample equals libjpeg's (PIL), i.e. the slow-integer
    IDCT, fancy upsampling and fixed-point colour conversion conventions are reproduced exactly.  The
    reference's BBB test images are progressive 4:4:4 JPEGs (stb_image decodes them there)."""
    Image = pytest.importorskip("PIL.Image")
    a, _ = rgb_fixture(manifest)
    src = Image.fromarray(a)
    n = 0
    for size in [a.shape[1::-1], (a.shape[1] - 3, a.shape[0] - 5), (33, 17), (8, 8), (1, 1), (17, 40)]:
        im = src.crop((0, 0) + tuple(size))
        for prog in (False, True):
            for ss in (0, 1, 2):
                for q, extra in ((30, {}), (90, {"optimize": True}), (3, {}), (100, {}), (75, {"restart_marker_blocks": 3})):
                    p = str(tmp_path / "t.jpg")
                    try:
                        im.save(p, "JPEG", quality=q, progressive=prog, subsampling=ss, **extra)
                    except (TypeError, ValueError, OSError):
                        continue                                  # an option this PIL does not know
                    assert np.array_equal(decode_with_cli(tmp_path, p), np.asarray(Image.open(p))), (size, prog, ss, q, extra)
                    n += 1
        p = str(tmp_path / "g.jpg")
        im.convert("L").save(p, "JPEG", quality=80, progressive=True)
        assert np.array_equal(decode_with_cli(tmp_path, p), np.asarray(Image.open(p))), size
    assert n >= 100
    if os.path.isdir("/root/reference/tests/images"):
        for name in ("big_buck_bunny_360_07806_00.jpg", "big_buck_bunny_360_07806_50.jpg", "big_buck_bunny_360_07806_100.jpg"):
            p = "/root/reference/tests/images/" + name
            assert np.array_equal(decode_with_cli(tmp_path, p), np.asarray(Image.open(p))), name
    # refused, not mis-decoded: arithmetic coding / 12-bit / CMYK are outside the decoder
    bad = bytearray(open(str(tmp_path / "g.jpg"), "rb").read())
    i = bad.index(b"\xff\xc2")
    bad[i + 1] = 0xCA
    open(str(tmp_path / "arith.jpg"), "wb").write(bytes(bad))
    r = run("--decode", str(tmp_path / "arith.jpg"), str(tmp_path / "x.raw"))
    assert r.returncode == 1 and "JPEG" in r.stderr


def test_argument_errors(tmp_path, manifest):
    a, b = rgb_fixture(manifest)
    pa, pb, pg = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm"), str(tmp_path / "g.pgm")
    h, w, _ = a.shape
    open(pa, "wb").write(b"P6\n%d %d\n255\n" % (w, h) + a.tobytes())
    open(pb, "wb").write(b"P6\n%d %d\n255\n" % (w - 1, h) + np.ascontiguousarray(b[:, :w - 1]).tobytes())
    open(pg, "wb").write(b"P5\n%d %d\n255\n" % (w, h) + a[:, :, 0].tobytes())
    assert "Unknown option" in run("-x", pa, pa).stderr
    r = run(pa, pb)
    assert r.returncode != 0 and "Images do not have the same dimensions: 257x65 vs 256x65" in r.stderr
    r = run(pa, pg)
    assert r.returncode != 0 and "Images do not have the same number of channels: 3 vs 1" in r.stderr
    r = run("-3", pa, pa)
    assert r.returncode != 0 and "Cannot compute SSIM for channel 3, images have only 3 channels" in r.stderr
    r = run(str(tmp_path / "missing.png"), pa)
    assert r.returncode != 0 and "Failed to open file" in r.stderr


def test_decoders_reject_mutated_files_without_crashing(tmp_path, manifest):
    """The built-in codecs parse untrusted files: truncated, bit-flipped and extended PNG / PNM / BMP / TGA inputs
    must end in exit code 0 or 1 (decoded, or refused with a message) -- never a signal, and never an attempt to
    allocate what a hostile header claims.  (The same loop was run 4000x under ASan + UBSan: clean.)"""
    import random
    import struct
    a, _ = rgb_fixture(manifest)
    small = np.ascontiguousarray(a[:24, :40])
    png = str(tmp_path / "seed.png")
    write_png(png, small, 4)
    h, w, _ = small.shape
    stride = (w * 3 + 3) // 4 * 4
    seeds = {
        ".png": open(png, "rb").read(),
        ".ppm": b"P6\n%d %d\n255\n" % (w, h) + small.tobytes(),
        ".pgm": b"P2\n3 2\n15\n1 2 3 4 5 6\n",
        ".bmp": b"BM" + struct.pack("<IHHI", 54 + stride * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 0, 0, 0, 0) + bytes(stride * h),
        ".tga": bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w, 0, h, 0, 24, 0]) + small.tobytes(),
    }
    try:
        import io
        from PIL import Image
        for k, (prog, ss) in enumerate(((False, 2), (True, 0), (True, 2))):
            b = io.BytesIO()
            Image.fromarray(small).save(b, "JPEG", quality=60, progressive=prog, subsampling=ss)
            seeds[".%d.jpg" % k] = b.getvalue()
    except ImportError:
        pass
    rng = random.Random(7)
    out = str(tmp_path / "out.raw")
    for it in range(400):
        ext = rng.choice(sorted(seeds))
        s = bytearray(seeds[ext])
        for _ in range(rng.randint(1, 6)):
            k = rng.randrange(4)
            if k == 0 and s:
                s[rng.randrange(len(s))] = rng.randrange(256)
            elif k == 1 and len(s) > 8:
                del s[rng.randrange(len(s)):]
            elif k == 2 and s:
                s[rng.randrange(min(len(s), 64))] = rng.choice([0, 255, 128, 1])      # header fields
            else:
                s += bytes(rng.randrange(256) for _ in range(rng.randint(1, 16)))
        path = str(tmp_path / ("m" + ext))
        open(path, "wb").write(bytes(s))
        r = subprocess.run([CLI, "--decode", path, out], capture_output=True, timeout=30)
        assert r.returncode in (0, 1), (it, ext, r.returncode, r.stderr[-200:])
    # a header that claims 2^31 x 2^31 pixels is refused up front
    huge = bytearray(seeds[".png"])
    huge[16:24] = struct.pack(">II", 0x7FFFFFFF, 0x7FFFFFFF)
    path = str(tmp_path / "huge.png")
    open(path, "wb").write(bytes(huge))
    r = subprocess.run([CLI, "--decode", path, out], capture_output=True, timeout=30)
    assert r.returncode == 1 and b"PNG" in r.stderr


def test_png_bit_depths_outside_the_specification_are_refused(tmp_path):
    """depth 0 sized a row of 0 bytes and divided by (1 << 0) - 1 (SIGSEGV / SIGFPE), depths 3, 5, 6, 7
    formed negative shifts.  Only the depths the PNG specification allows per colour type may pass the header check;
    the legal sub-byte depths must still decode."""
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    def png(depth, ctype, w=8, h=2, plte=False):
        row = bytes([0]) + bytes(max(1, (w * {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype, 1) * max(depth, 1) + 7) // 8))
        body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
        if plte:
            body += chunk(b"PLTE", bytes(range(48)))
        return body + chunk(b"IDAT", zlib.compress(row * h)) + chunk(b"IEND", b"")

    out = str(tmp_path / "o.raw")
    bad = [(0, 0), (3, 0), (5, 0), (6, 0), (7, 0), (32, 0), (0, 3), (3, 3), (16, 3), (1, 2), (4, 2), (0, 2), (2, 4), (4, 6), (0, 6), (8, 1), (8, 5)]
    for depth, ctype in bad:
        path = str(tmp_path / ("bad_%d_%d.png" % (depth, ctype)))
        open(path, "wb").write(png(depth, ctype, plte=ctype == 3))
        r = subprocess.run([CLI, "--decode", path, out], capture_output=True, timeout=30)
        assert r.returncode == 1 and b"PNG" in r.stderr, (depth, ctype, r.returncode, r.stderr[-200:])
    for depth, ctype in [(1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (1, 3), (2, 3), (4, 3), (8, 3), (8, 2), (16, 2), (8, 4), (16, 6)]:
        path = str(tmp_path / ("ok_%d_%d.png" % (depth, ctype)))
        open(path, "wb").write(png(depth, ctype, plte=ctype == 3))
        r = subprocess.run([CLI, "--decode", path, out], capture_output=True, timeout=30)
        assert r.returncode == 0, (depth, ctype, r.stderr[-200:])


def bt601(img):
    x = img.astype(np.uint32)
    return ((x[:, :, 0] * 19595 + x[:, :, 1] * 38470 + x[:, :, 2] * 7471 + 32768) >> 16).astype(np.uint8)


@pytest.mark.gpu
def test_cli_values_and_maps_on_gpu(tmp_path, manifest, oracle):
    a, b = rgb_fixture(manifest)
    h, w, _ = a.shape
    pa, pb = str(tmp_path / "a.png"), str(tmp_path / "b.ppm")
    write_png(pa, a, 4)
    open(pb, "wb").write(b"P6\n%d %d\n255\n" % (w, h) + b.tobytes())
    want = [np.uint32(int(manifest[n]["fma"]["ssim_hex"], 16)).view(np.float32) for n in manifest["_interleaved"]["per_channel"]]

    # all channels + average (src/ssim-cli.cpp:197-210), PFM map with 3 channels
    pfm = str(tmp_path / "map.pfm")
    r = run(pa, pb, pfm)
    assert r.returncode == 0, r.stderr
    avg = np.float32(0)
    for v in want:
        avg = np.float32(avg + v)
    lines = r.stdout.splitlines()
    assert lines == ["Channel %u: % 7.4f" % (c, want[c]) for c in range(3)] + ["Average  : % 7.4f" % (avg / np.float32(3))]
    raw = open(pfm, "rb").read()
    header = b"PF\n%d %d\n-1.0\n" % (w, h)
    assert raw.startswith(header)
    m = np.frombuffer(raw[len(header):], np.float32).reshape(h, w, 3)[::-1]
    for c, name in enumerate(manifest["_interleaved"]["per_channel"]):
        ref = np.load(os.path.join(GOLDEN, manifest[name]["fma"]["map"])) if "map" in manifest[name]["fma"] else oracle.ssim_f32(a[:, :, c], b[:, :, c], want_map=True)[2]
        assert np.array_equal(np.ascontiguousarray(m[:, :, c]).view(np.uint32), ref.view(np.uint32)), c

    # single channel, 8-bit PGM map = uint8(max(0, v) * 255)  (src/ssim-cli.cpp:341-342)
    pgm = str(tmp_path / "map.pgm")
    r = run("-1", pa, pb, pgm)
    assert r.returncode == 0 and r.stdout == "% 7.4f\n" % want[1]
    ref = np.load(os.path.join(GOLDEN, manifest["bbb257x65_q50_ch1"]["fma"]["map"]))
    raw = open(pgm, "rb").read()
    header = b"P5\n%d %d\n255\n" % (w, h)
    assert raw.startswith(header)
    got8 = np.frombuffer(raw[len(header):], np.uint8).reshape(h, w)
    assert np.array_equal(got8, (np.maximum(ref, 0) * np.float32(255)).astype(np.uint8))

    # luminance: integer BT.601 on the GPU, then SSIM; PNG map read back through the CLI's own decoder
    ya, yb = bt601(a), bt601(b)
    ov, _, om = oracle.ssim_f32(ya, yb, want_map=True)
    png = str(tmp_path / "map.png")
    r = run("-y", pa, pb, png)
    assert r.returncode == 0 and r.stdout == "% 7.4f\n" % ov, (r.stdout, float(ov), r.stderr)
    out = str(tmp_path / "map.raw")
    r = run("--decode", png, out)
    assert r.returncode == 0 and r.stdout.split() == [str(w), str(h), "1"]
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(h, w), (np.maximum(om, 0) * np.float32(255)).astype(np.uint8))

    # the same through the C ABI bindings, bit-exact
    import ssim_amd
    lv, lm = ssim_amd.compute_ssim_luminance(a, b, want_map=True)
    assert f32_hex(lv) == f32_hex(ov) and np.array_equal(lm.view(np.uint32), om.view(np.uint32))
    cv, cm = ssim_amd.compute_ssim_channels(a, b, want_map=True)
    assert [f32_hex(x) for x in cv] == [f32_hex(x) for x in want]
    # bottom-up interleaved input addresses the same pixels
    import ctypes
    lib = ssim_amd.load_library()
    af, bf = np.ascontiguousarray(a[::-1]), np.ascontiguousarray(b[::-1])
    out3 = (ctypes.c_float * 3)()
    rc = lib.rmgr_ssim_hip_compute_ssim_channels_host(None, out3, af.ctypes.data + (h - 1) * w * 3, -w * 3, bf.ctypes.data + (h - 1) * w * 3, -w * 3, w, h, 3, None)
    assert rc == 0 and [f32_hex(x) for x in out3] == [f32_hex(x) for x in want]
    one = ctypes.c_float()
    rc = lib.rmgr_ssim_hip_compute_ssim_luminance_host(None, ctypes.byref(one), af.ctypes.data + (h - 1) * w * 3, -w * 3, bf.ctypes.data + (h - 1) * w * 3, -w * 3, w, h, 3, None)
    assert rc == 0 and f32_hex(one.value) == f32_hex(ov)


@pytest.mark.gpu
def test_cli_png_versus_jpeg_on_gpu(tmp_path, manifest, oracle):
    """The reference tool's everyday use: an original against its JPEG-compressed copy (here: progressive 4:2:0)."""
    Image = pytest.importorskip("PIL.Image")
    a, _ = rgb_fixture(manifest)
    pa, pj = str(tmp_path / "a.png"), str(tmp_path / "a_q40.jpg")
    write_png(pa, a, 4)
    Image.fromarray(a).save(pj, "JPEG", quality=40, progressive=True, subsampling=2)
    j = np.asarray(Image.open(pj))
    want = [oracle.ssim_f32(np.ascontiguousarray(a[:, :, c]), np.ascontiguousarray(j[:, :, c]))[0] for c in range(3)]
    r = run(pa, pj)
    assert r.returncode == 0, r.stderr
    avg = np.float32(0)
    for v in want:
        avg = np.float32(avg + v)
    assert r.stdout.splitlines() == ["Channel %u: % 7.4f" % (c, want[c]) for c in range(3)] + ["Average  : % 7.4f" % (avg / np.float32(3))]
    ov = oracle.ssim_f32(bt601(a), bt601(j))[0]
    r = run("-y", pa, pj)
    assert r.returncode == 0 and r.stdout == "% 7.4f\n" % ov


def test_own_decoders_reproduce_the_fixture_pixels_of_every_reference_image(tmp_path, refsets):
    """Every image file of the reference's bbb sets (tests/golden/images: 2 PNG frames + 22 progressive JPEGs) through the
    tool's own PNG / JPEG readers: the decoded planes hash to the values recorded when the expected outputs were generated
    (refsets.json: PIL / libjpeg decode), i.e. the tool sees exactly the pixels the parity fixtures are defined on."""
    import hashlib
    seen = {}
    for set_name in ("bbb360", "bbb1080"):
        for key, ent in sorted(refsets[set_name]["pairs"].items()):
            for f, h in ((ent["a_file"], ent["a_sha256"]), (ent["b_file"], ent["b_sha256"])):
                if f not in seen:
                    seen[f] = decode_with_cli(tmp_path, os.path.join(GOLDEN, "images", f))
                    assert seen[f].shape == (ent["height"], ent["width"], 3), f
                plane = np.ascontiguousarray(seen[f][:, :, ent["channel"]])
                assert hashlib.sha256(plane.tobytes()).hexdigest() == h, (f, ent["channel"])
    assert len(seen) == 24


@pytest.mark.gpu
@pytest.mark.parametrize("res", ["360", "1080"])
def test_cli_on_the_reference_image_files(tmp_path, refsets, res):
    """`rmgr-ssim frame.png frame_qNN.jpg [map.pfm]` on the reference's own files, end to end (own decoders -> all channels in
    one launch -> the reference's printf formats): the three channel lines and the average are the FMA path's values
    (refsets.json), the PFM map its per-pixel values."""
    import hashlib
    pairs = refsets["bbb" + res]["pairs"]
    png = os.path.join(GOLDEN, "images", "big_buck_bunny_%s_07806.png" % res)
    for q in ("00", "50", "100") if res == "1080" else ("00", "10", "20", "30", "40", "50", "60", "70", "80", "90", "100"):
        jpg = os.path.join(GOLDEN, "images", "big_buck_bunny_%s_07806_%s.jpg" % (res, q))
        want = [np.array([int(pairs["q%s_ch%d" % (q, c)]["fma"]["ssim_hex"], 16)], np.uint32).view(np.float32)[0] for c in range(3)]
        with_map = q == "50"
        pfm = str(tmp_path / "m.pfm")
        r = run(png, jpg, pfm) if with_map else run(png, jpg)
        assert r.returncode == 0, r.stderr
        avg = np.float32(0)
        for v in want:
            avg = np.float32(avg + v)
        assert r.stdout.splitlines() == ["Channel %u: % 7.4f" % (c, want[c]) for c in range(3)] + ["Average  : % 7.4f" % (avg / np.float32(3))], (res, q)
        if with_map:
            e = pairs["q50_ch0"]
            raw = open(pfm, "rb").read()
            header = b"PF\n%d %d\n-1.0\n" % (e["width"], e["height"])
            assert raw.startswith(header)
            m = np.frombuffer(raw[len(header):], np.float32).reshape(e["height"], e["width"], 3)[::-1]
            for c in range(3):
                assert hashlib.sha256(np.ascontiguousarray(m[:, :, c]).tobytes()).hexdigest() == pairs["q50_ch%d" % c]["fma"]["map_sha256"], (res, c)
