"""Bjontegaard-delta rate between rate-distortion curves.

What the reference does: compare_bd_rate.py:193-225 (`retrieve_data`) collects, per method and per
dataset (or sequence), the lists of `ave_<frame_type>_frame_bpp` / `ave_<frame_type>_frame_<metric>`
over the rate points of a result JSON (the format test_video.py writes and
anchors/vtm_17.0_yuv420_LB_allf_ip0.json has) and calls `BD_RATE(R_anchor, D_anchor, R_test, D_test, 1)`
of the third-party package `bd_metric` (github.com/Anserw/Bjontegaard_metric, not vendored, not
installed here) for every dataset both methods cover with >= 3 points; class-level numbers average
the sequences of a class per rate point first (compare_bd_rate.py:98-146, weights = frame counts).

This module restates that pipeline without the dependency: the published piecewise-cubic variant of
the Bjontegaard metric (log-rate as a PCHIP interpolant of the distortion, both curves sampled at
100 points of the common distortion interval, trapezoid rule, exp of the mean log-rate difference)
with its own Fritsch-Carlson PCHIP (tests/test_bd_rate.py checks it against scipy's).

    python -m dcvc_amd.bd_rate --base_method VTM --log_paths VTM anchors/vtm.json DCVC-UF out.json
"""
import argparse
import json
import sys

import numpy as np


# ------------------------------------------------------------------------------------ PCHIP
def _pchip_slopes(x, y):
    """Fritsch-Carlson derivatives (the scheme of scipy.interpolate.PchipInterpolator): weighted
    harmonic mean of the neighbouring secants where they agree in sign, 0 otherwise; three-point
    shape-preserving formula at the ends."""
    h = np.diff(x)
    d = np.diff(y) / h
    n = len(x)
    m = np.zeros(n)
    if n == 2:
        m[:] = d[0]
        return m
    for k in range(1, n - 1):
        if d[k - 1] * d[k] > 0:
            w1 = 2 * h[k] + h[k - 1]
            w2 = h[k] + 2 * h[k - 1]
            m[k] = (w1 + w2) / (w1 / d[k - 1] + w2 / d[k])

    def end(h0, h1, d0, d1):
        s = ((2 * h0 + h1) * d0 - h0 * d1) / (h0 + h1)
        if np.sign(s) != np.sign(d0):
            return 0.0
        if np.sign(d0) != np.sign(d1) and abs(s) > 3 * abs(d0):
            return 3 * d0
        return s

    m[0] = end(h[0], h[1], d[0], d[1])
    m[-1] = end(h[-1], h[-2], d[-1], d[-2])
    return m


def pchip(x, y, xs):
    """Values at xs of the PCHIP interpolant through (x, y); x strictly increasing."""
    x, y, xs = np.asarray(x, float), np.asarray(y, float), np.asarray(xs, float)
    if len(x) < 2 or np.any(np.diff(x) <= 0):
        raise ValueError("pchip: need >= 2 strictly increasing abscissae")
    m = _pchip_slopes(x, y)
    k = np.clip(np.searchsorted(x, xs, side="right") - 1, 0, len(x) - 2)
    h = x[k + 1] - x[k]
    t = (xs - x[k]) / h
    h00 = (1 + 2 * t) * (1 - t) ** 2
    h10 = t * (1 - t) ** 2
    h01 = t * t * (3 - 2 * t)
    h11 = t * t * (t - 1)
    return h00 * y[k] + h10 * h * m[k] + h01 * y[k + 1] + h11 * h * m[k + 1]


# ------------------------------------------------------------------------------------ BD-rate
def bd_rate(rate_anchor, dist_anchor, rate_test, dist_test, samples=100):
    """Average rate difference in percent of `test` against `anchor` at equal distortion (negative =
    the test codec needs fewer bits). Piecewise-cubic Bjontegaard metric as called at
    compare_bd_rate.py:218-222 (`BD_RATE(..., 1)`)."""
    ra, da = np.asarray(rate_anchor, float), np.asarray(dist_anchor, float)
    rt, dt = np.asarray(rate_test, float), np.asarray(dist_test, float)
    if len(ra) != len(da) or len(rt) != len(dt) or len(ra) < 2 or len(rt) < 2:
        raise ValueError("bd_rate: need two curves of >= 2 (rate, distortion) points each")
    if np.any(ra <= 0) or np.any(rt <= 0):
        raise ValueError("bd_rate: rates must be positive")
    oa, ot = np.argsort(da), np.argsort(dt)
    lo = max(da.min(), dt.min())
    hi = min(da.max(), dt.max())
    if not hi > lo:
        raise ValueError("bd_rate: the curves share no distortion interval")
    xs, step = np.linspace(lo, hi, num=samples, retstep=True)
    va = pchip(da[oa], np.log(ra[oa]), xs)
    vt = pchip(dt[ot], np.log(rt[ot]), xs)
    trapz = lambda v: step * (v.sum() - 0.5 * (v[0] + v[-1]))
    avg = (trapz(vt) - trapz(va)) / (hi - lo)
    return float((np.exp(avg) - 1) * 100)


# ------------------------------------------------------------------------------------ result files
def class_average(results, metric="psnr"):
    """{dataset: {sequence: {rate_point: entry}}} -> {dataset: {rate_point: entry}} with the
    frame-count-weighted means of compare_bd_rate.py:98-146."""
    out = {}
    for ds, seqs in results.items():
        by_point = {}
        for seq in seqs.values():
            for rp, e in seq.items():
                by_point.setdefault(rp, []).append(e)
        out[ds] = {}
        for rp, entries in by_point.items():
            acc = {"i_frame_num": 0, "p_frame_num": 0}
            sums = {}
            for e in entries:
                ni, np_ = e["i_frame_num"], e["p_frame_num"]
                acc["i_frame_num"] += ni
                acc["p_frame_num"] += np_
                for ft, w in (("i", ni), ("p", np_), ("all", ni + np_)):
                    for key in ("bpp", metric):
                        name = "ave_%s_frame_%s" % (ft, key)
                        sums[name] = sums.get(name, 0.0) + (e.get(name) or 0.0) * w
            ni, np_ = max(acc["i_frame_num"], 1), max(acc["p_frame_num"], 1)
            na = acc["i_frame_num"] + acc["p_frame_num"]
            for name, v in sums.items():
                ft = name.split("_")[1]
                acc[name] = v / {"i": ni, "p": np_, "all": max(na, 1)}[ft]
            out[ds][rp] = acc
    return out


def per_sequence(results):
    """{dataset: {sequence: ...}} -> {sequence: ...} (compare_bd_rate.py:47-55)."""
    return {seq: pts for seqs in results.values() for seq, pts in seqs.items()}


def curves(by_unit, frame_type="all", metric="psnr"):
    """{unit: {rate_point: entry}} -> {unit: (bpp list, distortion list)} in file order."""
    out = {}
    for unit, pts in by_unit.items():
        out[unit] = ([e["ave_%s_frame_bpp" % frame_type] for e in pts.values()],
                     [e["ave_%s_frame_%s" % (frame_type, metric)] for e in pts.values()])
    return out


def compare(files, base_method, between="class", frame_type="all", metric="psnr"):
    """files: {method: result dict}. Returns {method: {unit: BD-rate %}} against `base_method`,
    with the reference's admission rule (compare_bd_rate.py:212-217): the unit exists for the
    anchor, the test curve has >= 3 points, positive first rate and distortion."""
    prep = {}
    for method, res in files.items():
        units = class_average(res, metric) if between == "class" else per_sequence(res)
        prep[method] = curves(units, frame_type, metric)
    base = prep[base_method]
    out = {}
    for method, cs in prep.items():
        if method == base_method:
            continue
        out[method] = {}
        for unit, (bpp, dist) in cs.items():
            if unit in base and len(bpp) >= 3 and base[unit][0][0] > 0 and dist[0] is not None and dist[0] > 0:
                out[method][unit] = bd_rate(base[unit][0], base[unit][1], bpp, dist)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--base_method", required=True)
    ap.add_argument("--log_paths", required=True, nargs="+", help="method name followed by its result JSON, repeated")
    ap.add_argument("--compare_between", default="class", choices=["class", "sequence"])
    ap.add_argument("--frame_type", default="all", choices=["i", "p", "all"])
    ap.add_argument("--distortion_metrics", nargs="+", default=["psnr"])
    a = ap.parse_args(argv)
    if len(a.log_paths) % 2:
        ap.error("--log_paths takes pairs: method name, file")
    files = {}
    for name, path in zip(a.log_paths[::2], a.log_paths[1::2]):
        with open(path) as f:
            files[name] = json.load(f)
    if a.base_method not in files:
        ap.error("base method %r is not among the log paths" % a.base_method)
    for metric in a.distortion_metrics:
        res = compare(files, a.base_method, a.compare_between, a.frame_type, metric)
        for method, units in res.items():
            print("BD-rate (%s, %s frames) of %s against %s" % (metric, a.frame_type, method, a.base_method))
            for unit in sorted(units):
                print("  %-48s %+8.2f %%" % (unit, units[unit]))
            if units:
                print("  %-48s %+8.2f %%" % ("* Average", float(np.mean(list(units.values())))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
