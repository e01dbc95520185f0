#!/usr/bin/env python3
"""Exploration of candidate arithmetic for the separable mode against the REAL reference maps cached by
fast_study_data.py (all five of the reference's test image sets).  CPU only, numpy; build-container tool.

usage: python tests/tools/fast_study.py [form ...] [--sets einstein,bbb360,...] [--jobs 8]
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

CACHE = "/tmp/faststudy"
f32, f64 = np.float32, np.float64

K21 = [7.07622393965721130e-02,
       5.66619709134101868e-02, 4.53713610768318176e-02,
       2.90912277996540070e-02, 2.32944320887327194e-02, 1.19597595185041428e-02,
       9.57662798464298248e-03, 7.66836293041706085e-03, 3.93706932663917542e-03, 1.29605561960488558e-03,
       2.02135881409049034e-03, 1.61857774946838617e-03, 8.31005279906094074e-04, 2.73561221547424793e-04, 5.77411265112459660e-05,
       2.73561221547424793e-04, 2.19050692976452410e-04, 1.12464345875196159e-04, 3.70224843209143728e-05, 7.81441485742107034e-06,
       1.05756600987660931e-06]
K21 = [f32(v) for v in K21]


def kc(i, j):
    if i > j:
        i, j = j, i
    return K21[j * (j + 1) // 2 + i]


def fma(a, b, c):
    return (a.astype(f64) * f64(b) + c.astype(f64)).astype(f32)


def taps_true():
    g = np.exp(-(np.arange(6, dtype=f64) ** 2) / (2 * 1.5 * 1.5))
    return g / (g[0] + 2 * g[1:].sum())


def folds(P):
    H, W = P.shape
    pad = np.pad(P, 5, mode="edge")
    return [pad[:, 5:5 + W]] + [pad[:, 5 + i:5 + i + W] + pad[:, 5 - i:5 - i + W] for i in range(1, 6)]


def blur_ref(P):
    """the reference FMA path's operation order (src/ssim_fma.cpp:196-257), float32"""
    H = P.shape[0]
    s = folds(P.astype(f32))
    d = None
    for r in range(11):                      # source row offset dy = r-5, top first
        j = abs(r - 5)
        S = (s[0][r:r + H] * kc(0, j)).astype(f32)
        for i in range(1, 6):
            S = fma(s[i][r:r + H], kc(i, j), S)
        d = S if d is None else (S + d).astype(f32)
    return d


def blur_sep(P, g, order=(0, 1, 2, 3, 4, 5)):
    """separable fp32: row pass on folded sums in `order`, column pass in ring order (top row first), all fma"""
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


def blur_f64(P, g=None):
    g = taps_true() if g is None else g
    H = P.shape[0]
    s = folds(P.astype(f64))
    h = sum(s[i] * g[i] for i in range(6))
    return sum(h[k:k + H] * g[abs(k - 5)] for k in range(11))


C1 = f32((0.01 * 255.0) ** 2)
C2 = f32((0.03 * 255.0) ** 2)


def px5(muA, muB, eAA, eBB, eAB):
    """reference epilogue (src/ssim_avx.cpp:342-352), float32 unfused"""
    muA2, muB2, muAB = muA * muA, muB * muB, muA * muB
    num = (f32(2) * muAB + C1) * (f32(2) * (eAB - muAB) + C2)
    den = ((muA2 + muB2) + C1) * (((eAA - muA2) + (eBB - muB2)) + C2)
    return (num / den).astype(f32)


def px_generic(muA, muB, sS, sAB):
    """epilogue from mu (uncentred) and the centred-form variance sum / covariance, float32"""
    muAB = muA * muB
    tm = muA * muA + muB * muB
    num = (f32(2) * muAB + C1) * (f32(2) * sAB + C2)
    den = (tm + C1) * (sS + C2)
    return (num / den).astype(f32)


# ---- forms: name -> fn(a, b) -> float32 map -------------------------------------------------------------------------
CENTRE, SMALL, INNER = (0, 1, 2, 3, 4, 5), (5, 4, 3, 2, 1, 0), (2, 1, 0, 3, 4, 5)
GT = taps_true().astype(f32)


def form_ref(a, b):      # must reproduce the cached FMA map bit for bit (checks the model)
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
        eS, eX = blur_sep(a * a + b * b, GT, oe), blur_sep(a * b, GT, oe)
        muAB = muA * muB
        tm = muA * muA + muB * muB
        return px_generic(muA, muB, eS - tm, eX - muAB)
    return f


def make_centred(planes, om, oe, c=128.0, mu="sep"):
    """everything blurred on a' = a - c; mu restored as mu' + c.  mu = 'sep' | 'ref' (reference-order blur of a', exact
    order on the centred plane -- NOT the reference's mu)"""
    def f(a, b):
        a = a.astype(f32) - f32(c); b = b.astype(f32) - f32(c)
        bl = (lambda P: blur_sep(P, GT, om))
        mA, mB = bl(a), bl(b)
        if planes == 5:
            sS = (blur_sep(a * a, GT, oe) - mA * mA) + (blur_sep(b * b, GT, oe) - mB * mB)
        else:
            sS = blur_sep(a * a + b * b, GT, oe) - (mA * mA + mB * mB)
        sAB = blur_sep(a * b, GT, oe) - mA * mB
        return px_generic(mA + f32(c), mB + f32(c), sS, sAB)
    return f


def make_hybrid(e_kind, planes=5, oe=CENTRE, c=128.0):
    """mu planes in the reference's exact order (bit-identical to the reference's mu); E planes: 'f64' exact double,
    'sep' separable fp32 uncentred, 'sepc' separable centred at c (variances formed from centred moments)"""
    def f(a, b):
        a = a.astype(f32); b = b.astype(f32)
        muA, muB = blur_ref(a), blur_ref(b)
        if e_kind == "f64":
            eAA, eBB, eAB = blur_f64(a * a), blur_f64(b * b), blur_f64(a * b)
            sS = ((eAA - muA.astype(f64) ** 2) + (eBB - muB.astype(f64) ** 2)).astype(f32)
            sAB = (eAB - muA.astype(f64) * muB.astype(f64)).astype(f32)
            return px_generic(muA, muB, sS, sAB)
        if e_kind == "sep":
            return px5(muA, muB, blur_sep(a * a, GT, oe), blur_sep(b * b, GT, oe), blur_sep(a * b, GT, oe))
        if e_kind == "sep_refmu2":
            # uncentred E planes, but the epilogue exactly as the reference: isolates E-plane rounding
            return px5(muA, muB, blur_sep(a * a, GT, oe), blur_sep(b * b, GT, oe), blur_sep(a * b, GT, oe))
        if e_kind == "sepc":
            ac, bc = a - f32(c), b - f32(c)
            mA, mB = muA - f32(c), muB - f32(c)
            if planes == 5:
                sS = (blur_sep(ac * ac, GT, oe) - mA * mA) + (blur_sep(bc * bc, GT, oe) - mB * mB)
            else:
                sS = blur_sep(ac * ac + bc * bc, GT, oe) - (mA * mA + mB * mB)
            sAB = blur_sep(ac * bc, GT, oe) - mA * mB
            return px_generic(muA, muB, sS, sAB)
        raise ValueError(e_kind)
    return f


def form_mu64_eref(a, b):
    """mu planes exact (double), E planes in the reference's order: what the reference's E-plane rounding alone does"""
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur_f64(a), blur_f64(b)
    eAA, eBB, eAB = blur_ref(a * a).astype(f64), blur_ref(b * b).astype(f64), blur_ref(a * b).astype(f64)
    sS = ((eAA - muA * muA) + (eBB - muB * muB)).astype(f32)
    sAB = (eAB - muA * muB).astype(f32)
    return px_generic(muA.astype(f32), muB.astype(f32), sS, sAB)


def make_eref(om, g=None):
    """E planes in the reference's exact order (bit-identical), mu planes separable fp32 in tap order `om`; reference epilogue"""
    def f(a, b):
        a = a.astype(f32); b = b.astype(f32)
        gg = GT if g is None else g
        return px5(blur_sep(a, gg, om), blur_sep(b, gg, om), blur_ref(a * a), blur_ref(b * b), blur_ref(a * b))
    return f


def taps_fit_k():
    """1-D taps whose outer product follows the reference's float-computed 2-D table: g_i = K(i,0) / sqrt(K(0,0))"""
    k0 = np.array([f64(kc(i, 0)) for i in range(6)])
    return (k0 / np.sqrt(k0[0])).astype(f32)


FORMS = {
    "eref_cc": make_eref(CENTRE),
    "eref_is": make_eref(INNER),
    "eref_sf": make_eref(SMALL),
    "eref_cc_k": make_eref(CENTRE, taps_fit_k()),
    "eref_sf_k": make_eref(SMALL, taps_fit_k()),
    "ref": form_ref,
    "five_cc": make_five(CENTRE, CENTRE),
    "five_is": make_five(INNER, SMALL),
    "four_is": make_four(INNER, SMALL),
    "c5_cc": make_centred(5, CENTRE, CENTRE),
    "c5_is": make_centred(5, INNER, SMALL),
    "c4_cc": make_centred(4, CENTRE, CENTRE),
    "hyb_f64": make_hybrid("f64"),
    "hyb_sep5": make_hybrid("sep", 5),
    "hyb_sepc5": make_hybrid("sepc", 5),
    "hyb_sepc4": make_hybrid("sepc", 4),
    "mu64_eref": form_mu64_eref,
}


def work(args):
    form, name = args
    z = np.load(os.path.join(CACHE, name + ".npz"))
    m = FORMS[form](z["a"], z["b"])
    g = f32(m.astype(f64).sum() / f64(m.size))
    fm, nm = z["fma_map"], z["naive_map"]
    return (form, name, float(np.abs(m.astype(f64) - fm.astype(f64)).max()), abs(float(g) - float(z["fma"])),
            float(np.abs(m.astype(f64) - nm).max()), abs(float(g) - float(z["naive"])), int((m.view(np.uint32) != fm.view(np.uint32)).sum()))


def main():
    argv = sys.argv[1:]
    sets = None
    jobs = 8
    forms = []
    i = 0
    while i < len(argv):
        if argv[i] == "--sets":
            sets = argv[i + 1].split(","); i += 2
        elif argv[i] == "--jobs":
            jobs = int(argv[i + 1]); i += 2
        else:
            forms.append(argv[i]); i += 1
    forms = forms or list(FORMS)
    index = [l.split() for l in open(os.path.join(CACHE, "index.txt"))]
    if sets:
        index = [(s, n) for s, n in index if s in sets]
    tasks = [(f, n) for f in forms for s, n in index]
    set_of = dict((n, s) for s, n in index)
    with Pool(jobs) as pool:
        res = pool.map(work, tasks, chunksize=1)
    print("| form | set | px vs FMA (6.3e-4) | glob vs FMA (1.5e-6) | px vs naive (1e-3) | glob vs naive (2e-6) | worst pair (px vs FMA) | px != FMA bits |")
    print("|---|---|---|---|---|---|---|---|")
    for f in forms:
        for s in sorted(set(set_of.values())):
            rows = [r for r in res if r[0] == f and set_of[r[1]] == s]
            if not rows:
                continue
            w = max(rows, key=lambda r: r[2])
            print("| %s | %s | %.3e | %.3e | %.3e | %.3e | %s | %d |" % (f, s, max(r[2] for r in rows), max(r[3] for r in rows),
                  max(r[4] for r in rows), max(r[5] for r in rows), w[1], sum(r[6] for r in rows)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
