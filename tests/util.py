"""Shared helpers for the test-suite (tests may use oracle/, the product may not)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_meta(g):
    return json.loads(bytes(g["meta"]).decode())


def logits_close(got, ref, rtol=1e-4, atol=1e-6):
    """BASELINE north_star bar: fp32 within 1e-4 relative (plus 1e-6 absolute floor for values near 0)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    bar = rtol * np.abs(ref) + atol
    return bool((err <= bar).all()), float((err / bar).max()) if err.size else 0.0


def assert_close(got, ref, rtol=1e-4, atol=1e-6, what=""):
    ok, worst = logits_close(got, ref, rtol, atol)
    assert ok, "%s: max err / bar = %.3g (rtol=%g atol=%g)" % (what, worst, rtol, atol)


def sigmoid_inv(y):
    y = np.clip(np.asarray(y, dtype=np.float64), 1e-300, 1 - 1e-16)
    return np.log(y) - np.log1p(-y)


def assert_fm_close(got, x, what="fm"):
    """FM = 0.5*sum_d(S_d^2 - Q_d) cancels, so besides the 1e-4 relative bar the fp32 floor is a few ulp of the
    TERMS (0.5*sum_d(S_d^2 + Q_d)), not of the result; x [B,F,E] are the embeddings fed to FM."""
    x = np.asarray(x, dtype=np.float64)
    ref = 0.5 * (np.square(x.sum(1)) - np.square(x).sum(1)).sum(-1)
    scale = 0.5 * (np.square(x.sum(1)) + np.square(x).sum(1)).sum(-1)
    err = np.abs(np.asarray(got, dtype=np.float64).reshape(-1) - ref)
    bar = 1e-4 * np.abs(ref) + 1e-6 + 4 * np.finfo(np.float32).eps * scale
    assert (err <= bar).all(), "%s: max err/bar %.3g" % (what, float((err / bar).max()))


TERMS_REPORT = []     # (what, elements, elements outside 1e-4 * |ref|, worst err / (1e-4 * |ref|), worst err / bar)


def assert_close_terms(got, ref, terms, rtol=1e-4, rtol_terms=2e-6, what=""):
    """fp32 bar for a result that is a SUM of terms which may cancel: 1e-4 of the result (north_star) plus a few fp32 ulp
    (2e-6 ~ 17 * 2^-23: summation orders of the MFMA path and the reference differ over tens to hundreds of terms) of the
    magnitude the terms were summed at.  `terms` >= sum of |summands| per output element, from the float64 oracle run on
    absolute values (an upper bound of every intermediate magnitude for products, sums and ReLU); it replaces blanket
    absolute tolerances: where nothing cancels, terms ~ |ref| and the bar stays 1e-4 relative."""
    got, ref, terms = (np.asarray(a, dtype=np.float64) for a in (got, ref, terms))
    err = np.abs(got - ref)
    bar = rtol * np.abs(ref) + rtol_terms * np.abs(terms) + 1e-30
    worst = float((err / bar).max()) if err.size else 0.0
    # how much of the comparison leaned on the extra term: elements outside north_star's literal bar (1e-4 of the result), and the
    # worst error in units of that literal bar — collected here, printed at the end of the session (tests/conftest.py)
    lit = rtol * np.abs(ref) + 1e-30
    TERMS_REPORT.append((what, int(err.size), int((err > lit).sum()), float((err / lit).max()) if err.size else 0.0, worst))
    assert (err <= bar).all(), "%s: max err / bar = %.3g (rtol=%g, %g of the summed magnitude)" % (what, worst, rtol, rtol_terms)
