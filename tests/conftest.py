import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "rccl: depends on an RCCL communicator (collected last: a stalled bootstrap must not mask parity tests)")


def pytest_collection_modifyitems(config, items):
    """Tests that need an RCCL communicator run after everything else, whatever the file order: under `pytest -x` a
    communication-library problem of the box must never hide a parity test (round 3's GPU record stopped at test 43 of 96
    for exactly that reason)."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("rccl") else 0)      # stable: the rest keeps its order


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_pair(entry):
    w, h = entry["width"], entry["height"]
    a = np.fromfile(os.path.join(GOLDEN, entry["a"]), np.uint8).reshape(h, w)
    b = np.fromfile(os.path.join(GOLDEN, entry["b"]), np.uint8).reshape(h, w)
    return a, b


def image_entries(manifest):
    return sorted(k for k in manifest if not k.startswith("_"))


def f32_hex(v):
    return "0x%08x" % np.float32(v).view(np.uint32)


def ulp_diff(a, b):
    """Distance in float32 ulps (both finite, same sign expected)."""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.oracle_lib()  # builds on demand
    return o


@pytest.fixture(scope="session")
def lib():
    import ssim_amd
    return ssim_amd.load_library()


@pytest.fixture(scope="session")
def gpu_ctx():
    """A context on cuda:0.  Fails (never skips) when the extension or the device is missing."""
    import ssim_amd
    assert ssim_amd.device_count() > 0, "no HIP device visible: -m gpu tests need the MI355X"
    ctx = ssim_amd.Context(0)
    yield ctx
    ctx.close()


# ---- the reference's full-size test sets (tests/golden/images + refsets.json; tests/tools/make_refset_fixtures.py) --
IMAGES = os.path.join(GOLDEN, "images")
_decoded = {}


@pytest.fixture(scope="session")
def refsets():
    with open(os.path.join(GOLDEN, "refsets.json")) as f:
        return json.load(f)["sets"]


def _decode_rgb(name):
    if name not in _decoded:
        from PIL import Image
        _decoded[name] = np.array(Image.open(os.path.join(IMAGES, name)).convert("RGB"))
        if len(_decoded) > 4:                      # a 1080p RGB frame is 6 MB: keep the cache small
            _decoded.pop(next(k for k in _decoded if k != name and not k.endswith(".png")))
    return _decoded[name]


def refset_pair(entry):
    """Decoded planes of one (pair, channel) of refsets.json.  The pixels must be the ones the expected outputs were
    generated from: a decoder that produces anything else fails the test loudly (nothing is compared on other pixels)."""
    import hashlib
    a = _decode_rgb(entry["a_file"])[:entry["height"], :entry["width"], entry["channel"]]
    b = _decode_rgb(entry["b_file"])[:entry["height"], :entry["width"], entry["channel"]]
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert hashlib.sha256(a.tobytes()).hexdigest() == entry["a_sha256"], "PNG decode differs from the fixture generator's: " + entry["a_file"]
    assert hashlib.sha256(b.tobytes()).hexdigest() == entry["b_sha256"], "JPEG decode differs from the fixture generator's (PIL/libjpeg version?): " + entry["b_file"]
    return a, b
