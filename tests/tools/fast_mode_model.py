#!/usr/bin/env python3
"""CPU model (numpy) of the arithmetic of the two non-bit-exact fp32 modes of ssim_kernels.hip, and the study that chose
them, over ALL of the reference's test image sets (einstein, bbb255, bbb257, bbb360, bbb1080: 138 pairs).

  MODE_FAST       the three E[.] planes in the reference FMA path's exact operation order (blur_exact: bit-identical to
                  the reference's planes), the two mu planes separable (blur_separable_pair, centre-first row pass, fused
                  column pass), the reference's unfused epilogue (ssim_px2_head/_tail)
  MODE_SEPARABLE  four planes on centred pixels a' = a - 128: mu_a', mu_b', E[a'^2 + b'^2], E[a'b'], all separable,
                  centre first (ssim_px2_sep)

fma(a,b,c) is modelled as float32(float64(a)*float64(b) + float64(c)) (the product of two floats is exact in double).
`ref` -- the reference's own operation order in this model -- reproduces the oracle's FMA maps bit for bit (last column
of the table: 0 differing pixels), which is what validates the model; the GPU tests assert that the kernels reproduce the
model's numbers (tests/test_gpu_modes.py).

What the study shows (DESIGN.md section 2 has the table):
  * on bbb1080 the reference's FMA path is up to 6.46e-4 away from the exact per-pixel value, MORE than north_star's
    FMA-relative tolerance of 6.3e-4.  Arithmetic that is not correlated with the reference's rounding -- exact arithmetic
    included (`exact mu + E`) -- therefore cannot be inside that tolerance on every pixel of that set: all purely
    separable forms land at 6.8e-4 ... 8.6e-4 there, whatever the tap order, plane count or centring.
  * almost all of the reference's per-pixel error is the rounding of its three E[.] planes (reference mu + exact E:
    6.5e-4 from the reference; exact mu + reference E: 2.2e-4), while its global bias sits in the mu planes on some images
    and in the E planes on others.  Reproducing the E planes bit for bit and approximating only the mu planes (MODE_FAST)
    stays within 2.3e-4 per pixel / 1.02e-6 global of the FMA reference on all 138 pairs.
  * the three E planes have to be treated ALIKE: with only E[ab], or only E[a^2] and E[b^2], in the reference's order the
    errors of numerator and denominator stop cancelling and the GLOBAL value moves by 2.7e-6 ... 3.1e-6 -- worse than any
    all-separable form.  There is no cheaper hybrid than "all three E planes exact".

Reference values come from the oracle's C restatement (bit-identical to the real reference kernels on every one of these
pairs: tests/test_oracle_golden.py), so the tool runs wherever the repository does; maps are cached under /tmp.

usage: python tests/tools/fast_mode_model.py [form ...] [--sets einstein,bbb360,...] [--jobs 8]      (CPU only, ~3 min for all)
       python tests/tools/fast_mode_model.py --adversarial      (synthetic flat / saturated stress images, profiles/r03_adversarial.md)
"""
import hashlib
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("SSIM_MODEL_CACHE", "/tmp/ssim_model_cache")
f32, f64 = np.float32, np.float64

# the 21 taps of the reference's SIMD paths (src/ssim_fma.cpp:169-174), as ktab() in ssim_kernels.hip
K21 = [f32(v) for v in (
    7.07622393965721130e-02,
    5.66619709134101868e-02, 4.53713610768318176e-02,
    2.90912277996540070e-02, 2.32944320887327194e-02, 1.19597595185041428e-02,
    9.57662798464298248e-03, 7.66836293041706085e-03, 3.93706932663917542e-03, 1.29605561960488558e-03,
    2.02135881409049034e-03, 1.61857774946838617e-03, 8.31005279906094074e-04, 2.73561221547424793e-04, 5.77411265112459660e-05,
    2.73561221547424793e-04, 2.19050692976452410e-04, 1.12464345875196159e-04, 3.70224843209143728e-05, 7.81441485742107034e-06,
    1.05756600987660931e-06)]
C1 = f32((0.01 * 255.0) ** 2)
C2 = f32((0.03 * 255.0) ** 2)
CENTRE, SMALL, INNER = (0, 1, 2, 3, 4, 5), (5, 4, 3, 2, 1, 0), (2, 1, 0, 3, 4, 5)


def kc(i, j):
    if i > j:
        i, j = j, i
    return K21[j * (j + 1) // 2 + i]


def fma(a, b, c):
    return (a.astype(f64) * f64(b) + c.astype(f64)).astype(f32)


def taps_true():
    """launch() in ssim_kernels.hip: the true 1-D Gaussian, sigma 1.5, normalised over the 11 taps, in double"""
    g = np.exp(-(np.arange(6, dtype=f64) ** 2) / (2 * 1.5 * 1.5))
    return g / (g[0] + 2 * g[1:].sum())


GT = taps_true().astype(f32)


def folds(P):
    H, W = P.shape
    pad = np.pad(P, 5, mode="edge")
    return [pad[:, 5:5 + W]] + [pad[:, 5 + i:5 + i + W] + pad[:, 5 - i:5 - i + W] for i in range(1, 6)]


def blur_ref(P):
    """blur_exact<true>: the reference FMA path's operation order (src/ssim_fma.cpp:196-257)"""
    H = P.shape[0]
    s = folds(P.astype(f32))
    d = None
    for r in range(11):                      # source row offset dy = r - 5, top row first
        j = abs(r - 5)
        S = (s[0][r:r + H] * kc(0, j)).astype(f32)
        for i in range(1, 6):
            S = fma(s[i][r:r + H], kc(i, j), S)
        d = S if d is None else (S + d).astype(f32)
    return d


def blur_sep(P, g=GT, order=CENTRE):
    """blur_separable: row pass on the folded sums in `order`, column pass in ring order (top row first), all fused"""
    H = P.shape[0]
    s = folds(P.astype(f32))
    g = [f32(v) for v in g]
    h = (s[order[0]] * g[order[0]]).astype(f32)
    for i in order[1:]:
        h = fma(s[i], g[i], h)
    acc = (h[0:H] * g[5]).astype(f32)
    for k in range(1, 11):
        acc = fma(h[k:k + H], g[abs(k - 5)], acc)
    return acc


def blur_f64(P):
    g = taps_true()
    H = P.shape[0]
    s = folds(P.astype(f64))
    h = sum(s[i] * g[i] for i in range(6))
    return sum(h[k:k + H] * g[abs(k - 5)] for k in range(11))


def px5(muA, muB, eAA, eBB, eAB):
    """ssim_px / ssim_px2_head + _tail: the reference epilogue (src/ssim_avx.cpp:342-352), float32 unfused"""
    muA2, muB2, muAB = muA * muA, muB * muB, muA * muB
    num = (f32(2) * muAB + C1) * (f32(2) * (eAB - muAB) + C2)
    den = ((muA2 + muB2) + C1) * (((eAA - muA2) + (eBB - muB2)) + C2)
    return (num / den).astype(f32)


def px4(muA, muB, sS, sAB):
    """the four-plane epilogue from mu and the variance sum / covariance, float32 unfused"""
    muAB = muA * muB
    tm = muA * muA + muB * muB
    num = (f32(2) * muAB + C1) * (f32(2) * sAB + C2)
    den = (tm + C1) * (sS + C2)
    return (num / den).astype(f32)


# ---- the two shipped modes ------------------------------------------------------------------------------------------
def mode_fast(a, b, order=CENTRE):
    a = a.astype(f32); b = b.astype(f32)
    return px5(blur_sep(a, GT, order), blur_sep(b, GT, order), blur_ref(a * a), blur_ref(b * b), blur_ref(a * b))


def mode_separable(a, b, c=128.0, om=CENTRE, oe=CENTRE):
    """round 5: the kernels form the quotient as n * rcp(d) with the hardware's 1-ulp reciprocal, which is not reproducible here:
    this model divides exactly and the kernels are compared with it to 3 ulp (tests/test_gpu_modes.py)"""
    a = a.astype(f32) - f32(c); b = b.astype(f32) - f32(c)
    mA, mB = blur_sep(a, GT, om), blur_sep(b, GT, om)
    sS = blur_sep(a * a + b * b, GT, oe) - (mA * mA + mB * mB)
    sAB = blur_sep(a * b, GT, oe) - mA * mB
    return px4(mA + f32(c), mB + f32(c), sS, sAB)


def px4_centred_expanded(mA, mB, eS, eX, c=128.0):
    """round 5, tried and DROPPED: luminance terms expanded around the centre -- 2 mu_a mu_b + c1 = fma(2c, s, fma(2, pc, K1)),
    mu_a^2 + mu_b^2 + c1 = fma(2c, s, tc + K1) with s = m_a + m_b, pc = m_a m_b, tc = m_a^2 + m_b^2, K1 = 2 c^2 + c1: two packed and
    two scalar operations fewer per pixel pair, but in dark areas 2 pc + 2c s + 2 c^2 cancels catastrophically: 4.1e-4 per pixel
    on bbb360 (the shipped form: 1.5e-4)."""
    k1 = f32(np.float64(2.0 * c * c) + np.float64(0.01 * 255.0) ** 2)
    tc = mA * mA + mB * mB
    pc = mA * mB
    ss = mA + mB
    sS, sAB = eS - tc, eX - pc
    nl = fma(f32(2 * c), ss, fma(f32(2), pc, k1))
    dl = fma(f32(2 * c), ss, tc + k1)
    n = nl * fma(f32(2), sAB, C2)
    d = dl * (sS + C2)
    return (n / d).astype(f32)


def form_separable_expanded(a, b, c=128.0):
    a = a.astype(f32) - f32(c); b = b.astype(f32) - f32(c)
    return px4_centred_expanded(blur_sep(a), blur_sep(b), blur_sep(a * a + b * b), blur_sep(a * b), c)


def form_separable_difference(a, b, c=128.0):
    """round 5, tried and DROPPED: mu_a^2 + mu_b^2 as (m_a - m_b)^2 + 2 m_a m_b and the luminance denominator as numerator +
    (m_a - m_b)^2: four packed operations fewer per pixel pair, per-pixel error unchanged, but the global value picks up a
    bias (1.4e-6 on the bbb crops against 8.7e-7)."""
    a = a.astype(f32) - f32(c); b = b.astype(f32) - f32(c)
    mA, mB, eS, eX = blur_sep(a), blur_sep(b), blur_sep(a * a + b * b), blur_sep(a * b)
    d = mA - mB
    dd = d * d
    pc = mA * mB
    tc = fma(f32(2), pc, dd)
    ln = fma(f32(2), (mA + f32(c)) * (mB + f32(c)), C1)
    n = ln * fma(f32(2), eX - pc, C2)
    den = (ln + dd) * ((eS - tc) + C2)
    return (n / den).astype(f32)


# ---- the forms that were studied and not shipped ----------------------------------------------------------------------
def form_ref(a, b):
    a = a.astype(f32); b = b.astype(f32)
    return px5(blur_ref(a), blur_ref(b), blur_ref(a * a), blur_ref(b * b), blur_ref(a * b))


def make_five(om, oe):
    def f(a, b):
        a = a.astype(f32); b = b.astype(f32)
        return px5(blur_sep(a, GT, om), blur_sep(b, GT, om), blur_sep(a * a, GT, oe), blur_sep(b * b, GT, oe), blur_sep(a * b, GT, oe))
    return f


def make_four(om, oe):
    def f(a, b):
        a = a.astype(f32); b = b.astype(f32)
        muA, muB = blur_sep(a, GT, om), blur_sep(b, GT, om)
        return px4(muA, muB, blur_sep(a * a + b * b, GT, oe) - (muA * muA + muB * muB), blur_sep(a * b, GT, oe) - muA * muB)
    return f


def form_centred_five(a, b, c=128.0):
    a = a.astype(f32) - f32(c); b = b.astype(f32) - f32(c)
    mA, mB = blur_sep(a), blur_sep(b)
    sS = (blur_sep(a * a) - mA * mA) + (blur_sep(b * b) - mB * mB)
    return px4(mA + f32(c), mB + f32(c), sS, blur_sep(a * b) - mA * mB)


def form_refmu_exactE(a, b):
    """the reference's mu planes, exact (double) E planes: what the reference's mu rounding alone does"""
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur_ref(a), blur_ref(b)
    eAA, eBB, eAB = blur_f64(a * a), blur_f64(b * b), blur_f64(a * b)
    sS = ((eAA - muA.astype(f64) ** 2) + (eBB - muB.astype(f64) ** 2)).astype(f32)
    return px4(muA, muB, sS, (eAB - muA.astype(f64) * muB.astype(f64)).astype(f32))


def form_exactmu_refE(a, b):
    """exact (double) mu planes, the reference's E planes: what the reference's E-plane rounding alone does"""
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur_f64(a), blur_f64(b)
    eAA, eBB, eAB = blur_ref(a * a).astype(f64), blur_ref(b * b).astype(f64), blur_ref(a * b).astype(f64)
    sS = ((eAA - muA * muA) + (eBB - muB * muB)).astype(f32)
    return px4(muA.astype(f32), muB.astype(f32), sS, (eAB - muA * muB).astype(f32))


def make_partial_hybrid(aa_bb_ref, ab_ref):
    """separable mu planes, and only some of the E planes in the reference's order"""
    def f(a, b):
        a = a.astype(f32); b = b.astype(f32)
        bl = lambda P, r: blur_ref(P) if r else blur_sep(P)
        return px5(blur_sep(a), blur_sep(b), bl(a * a, aa_bb_ref), bl(b * b, aa_bb_ref), bl(a * b, ab_ref))
    return f


def form_hybrid_four(a, b):
    """round 4 study: MODE_FAST with E[a^2] and E[b^2] merged into ONE reference-order plane E[a^2 + b^2] (four planes, 169 lane-ops)"""
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur_sep(a), blur_sep(b)
    muA2, muB2, muAB = muA * muA, muB * muB, muA * muB
    eS, eAB = blur_ref(a * a + b * b), blur_ref(a * b)
    num = (f32(2) * muAB + C1) * (f32(2) * (eAB - muAB) + C2)
    den = ((muA2 + muB2) + C1) * ((eS - (muA2 + muB2)) + C2)
    return (num / den).astype(f32)


def form_hybrid_four_split(a, b):
    """the same with the reference's association of the variance sum: (E[a^2 + b^2] - mu_a^2) - mu_b^2"""
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur_sep(a), blur_sep(b)
    muA2, muB2, muAB = muA * muA, muB * muB, muA * muB
    eS, eAB = blur_ref(a * a + b * b), blur_ref(a * b)
    num = (f32(2) * muAB + C1) * (f32(2) * (eAB - muAB) + C2)
    den = ((muA2 + muB2) + C1) * (((eS - muA2) - muB2) + C2)
    return (num / den).astype(f32)


FORMS = {
    "ref": (form_ref, "the reference's order in this model (validates the model: 0 pixels differ)"),
    "MODE_FAST": (mode_fast, "shipped: reference-order E planes, separable mu planes (centre first)"),
    "MODE_SEPARABLE": (mode_separable, "shipped: four planes, centred pixels, all separable (centre first)"),
    "fast, mu inner first": (lambda a, b: mode_fast(a, b, INNER), "MODE_FAST with round 2's mu order 2,1,0,3,4,5"),
    "fast, mu small first": (lambda a, b: mode_fast(a, b, SMALL), "MODE_FAST with mu order 5,4,3,2,1,0"),
    "five planes": (make_five(CENTRE, CENTRE), "round 1: five planes separable, centre first"),
    "five planes, r2 orders": (make_five(INNER, SMALL), "five planes with round 2's orders"),
    "four planes, r2 orders": (make_four(INNER, SMALL), "round 2's MODE_FAST"),
    "five planes centred": (form_centred_five, "five planes on centred pixels"),
    "separable, luminance expanded": (form_separable_expanded, "round 5 (dropped): MODE_SEPARABLE with the luminance terms expanded around the centre"),
    "separable, difference form": (form_separable_difference, "round 5 (dropped): MODE_SEPARABLE with mu_a^2 + mu_b^2 = (m_a - m_b)^2 + 2 m_a m_b"),
    "fast, E[ab] separable": (make_partial_hybrid(True, False), "MODE_FAST with only E[a^2], E[b^2] in the reference's order"),
    "fast, E[a^2] E[b^2] separable": (make_partial_hybrid(False, True), "MODE_FAST with only E[ab] in the reference's order"),
    "hybrid, E[a^2 + b^2] one plane": (form_hybrid_four, "round 4: MODE_FAST with the two variance planes merged into one reference-order plane"),
    "hybrid, one plane, split subtraction": (form_hybrid_four_split, "the same, (E - mu_a^2) - mu_b^2"),
    "ref mu + exact E": (form_refmu_exactE, "attribution: only the reference's mu rounding"),
    "exact mu + ref E": (form_exactmu_refE, "attribution: only the reference's E rounding"),
}
DEFAULT = ["ref", "MODE_FAST", "MODE_SEPARABLE", "five planes", "five planes, r2 orders", "four planes, r2 orders", "five planes centred",
           "fast, mu inner first", "fast, mu small first", "fast, E[ab] separable", "fast, E[a^2] E[b^2] separable", "ref mu + exact E", "exact mu + ref E"]


# ---- data: the 18 small fixtures (manifest.json) + the four bbb sets (refsets.json, decoded with PIL) -------------------
def all_pairs():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for n in sorted(k for k in man if k.startswith("einstein_")):
        yield "einstein", n
    ref = json.load(open(os.path.join(GOLDEN, "refsets.json")))["sets"]
    for s in ("bbb255", "bbb257", "bbb360", "bbb1080"):
        for k in sorted(ref[s]["pairs"]):
            yield s, s + "_" + k


def load(set_name, name):
    """inputs + oracle maps of one pair, cached"""
    p = os.path.join(CACHE, name + ".npz")
    if os.path.exists(p):
        z = np.load(p)
        return z["a"], z["b"], z["fma"], z["fma_map"], z["naive"], z["naive_map"]
    import oracle
    if set_name == "einstein":
        e = json.load(open(os.path.join(GOLDEN, "manifest.json")))[name]
        a = np.fromfile(os.path.join(GOLDEN, e["a"]), np.uint8).reshape(e["height"], e["width"])
        b = np.fromfile(os.path.join(GOLDEN, e["b"]), np.uint8).reshape(e["height"], e["width"])
    else:
        from PIL import Image
        e = json.load(open(os.path.join(GOLDEN, "refsets.json")))["sets"][set_name]["pairs"][name[len(set_name) + 1:]]
        img = lambda f: np.array(Image.open(os.path.join(GOLDEN, "images", f)).convert("RGB"))
        a = np.ascontiguousarray(img(e["a_file"])[:e["height"], :e["width"], e["channel"]])
        b = np.ascontiguousarray(img(e["b_file"])[:e["height"], :e["width"], e["channel"]])
        assert hashlib.sha256(a.tobytes()).hexdigest() == e["a_sha256"] and hashlib.sha256(b.tobytes()).hexdigest() == e["b_sha256"], "decoder differs"
    fma_v, _, fma_map = oracle.ssim_f32(a, b, want_map=True, fused=True)
    nv, _, nmap = oracle.ssim_naive_f64(a, b, want_map=True)
    os.makedirs(CACHE, exist_ok=True)
    np.savez(p, a=a, b=b, fma=f32(fma_v), fma_map=fma_map, naive=f64(nv), naive_map=nmap)
    return a, b, f32(fma_v), fma_map, f64(nv), nmap


def evaluate(fn, a, b, fma_v, fma_map, nv, nmap):
    m = fn(a, b)
    g = f32(m.astype(f64).sum() / f64(m.size))
    return (float(np.abs(m.astype(f64) - fma_map.astype(f64)).max()), abs(float(g) - float(fma_v)),
            float(np.abs(m.astype(f64) - nmap).max()), abs(float(g) - float(nv)), int((m.view(np.uint32) != fma_map.view(np.uint32)).sum()))


def work(task):
    form, set_name, name = task
    return (form, set_name, name) + evaluate(FORMS[form][0], *load(set_name, name))


def adversarial_pairs(H=384, W=512, seed=7):
    """Synthetic stress images for the non-bit-exact modes: large flat (bright / mid / dark) areas with +-1...2 of noise,
    where every pixel's rounding error has the same sign -- the regime in which a global value cannot average per-pixel
    deviations away -- plus gradients, saturated patterns and uncorrelated / anti-correlated noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    u8 = lambda v: np.clip(v, 0, 255).astype(np.uint8)
    for v in (255, 250, 240, 200, 128, 30):
        a = u8(v + rng.integers(-1, 2, (H, W)))
        yield "flat %d, both +-1 noise" % v, a, u8(a.astype(int) + rng.integers(-1, 2, (H, W)))
        a = np.full((H, W), v, np.uint8)
        yield "flat %d vs +-2 noise" % v, a, u8(a.astype(int) + rng.integers(-2, 3, (H, W)))
    a = u8(180 + 70 * xx / W + rng.integers(-1, 2, (H, W)))
    yield "bright gradient", a, u8(a.astype(int) + rng.integers(-3, 4, (H, W)))
    a = u8(np.where(((xx // 32) + (yy // 32)) % 2 == 0, 255, 235))
    yield "bright checker, 32 px", a, u8(a.astype(int) + rng.integers(-2, 3, (H, W)))
    a = u8(255 - ((xx * 7 + yy * 3) % 5))
    yield "saturated sawtooth", a, u8(a.astype(int) - rng.integers(0, 3, (H, W)))
    a = u8(245 + 8 * np.sin(xx / 3.0) * np.cos(yy / 5.0))
    yield "bright ripple", a, u8(a.astype(int) + rng.integers(-1, 2, (H, W)))
    yield "uncorrelated 250..255", u8(rng.integers(250, 256, (H, W))), u8(rng.integers(250, 256, (H, W)))
    a = u8(rng.integers(0, 256, (H, W)))
    yield "anti-correlated noise", a, u8(255 - a.astype(int))


def adversarial():
    """python tests/tools/fast_mode_model.py --adversarial: the table of profiles/r03_adversarial.md"""
    import oracle
    print("| image (384 x 512) | reference FMA vs exact: pixel / global | MODE_FAST vs FMA: pixel / global | MODE_FAST vs exact: pixel / global | MODE_SEPARABLE vs exact: pixel / global |")
    print("|---|---|---|---|---|")
    for name, a, b in adversarial_pairs():
        fv, _, fmap = oracle.ssim_f32(a, b, want_map=True, threads=8)
        nv, _, nmap = oracle.ssim_naive_f64(a, b, want_map=True, threads=8)
        cells = ["%.2e / %.2e" % (np.abs(fmap.astype(f64) - nmap).max(), abs(float(fv) - nv))]
        rf = evaluate(mode_fast, a, b, fv, fmap, nv, nmap)
        rs = evaluate(mode_separable, a, b, fv, fmap, nv, nmap)
        cells += ["%.2e / %.2e" % (rf[0], rf[1]), "%.2e / %.2e" % (rf[2], rf[3]), "%.2e / %.2e" % (rs[2], rs[3])]
        print("| %s | %s |" % (name, " | ".join(cells)))


def main():
    argv = sys.argv[1:]
    if "--adversarial" in argv:
        return adversarial()
    sets, jobs, forms = None, 8, []
    i = 0
    while i < len(argv):
        if argv[i] == "--sets":
            sets = argv[i + 1].split(","); i += 2
        elif argv[i] == "--jobs":
            jobs = int(argv[i + 1]); i += 2
        else:
            forms.append(argv[i]); i += 1
    forms = forms or DEFAULT
    pairs = [(s, n) for s, n in all_pairs() if not sets or s in sets]
    with Pool(jobs) as pool:
        res = pool.map(work, [(f, s, n) for f in forms for s, n in pairs], chunksize=1)
    print("taps", " ".join("%.9g" % v for v in GT), " sum-1 = %.2e" % (float(GT[0].astype(f64) + 2 * GT[1:].astype(f64).sum()) - 1))
    print("| form | set | pixel vs FMA (6.3e-4) | global vs FMA (1.5e-6) | pixel vs naive (1e-3) | global vs naive (2e-6) | worst pair (pixel vs FMA) | pixels != FMA |")
    print("|---|---|---|---|---|---|---|---|")
    order = ["einstein", "bbb255", "bbb257", "bbb360", "bbb1080"]
    for f in forms:
        rows_all = [r for r in res if r[0] == f]
        for s in [x for x in order if any(r[1] == x for r in rows_all)] + ["ALL"]:
            rows = [r for r in rows_all if s == "ALL" or r[1] == s]
            w = max(rows, key=lambda r: r[3])
            mark = lambda v, tol: "%.3e%s" % (v, " ✗" if v > tol else "")
            print("| %s | %s (%d) | %s | %s | %s | %s | %s | %d |" % (f, s, len(rows), mark(max(r[3] for r in rows), 6.3e-4), mark(max(r[4] for r in rows), 1.5e-6),
                  mark(max(r[5] for r in rows), 1e-3), mark(max(r[6] for r in rows), 2e-6), w[2], sum(r[7] for r in rows)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
