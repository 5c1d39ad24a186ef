"""BD-rate tooling (compare_bd_rate.py:193-225 without the `bd_metric` dependency):
the restated metric against known answers made by an independent scipy-based restatement on the
reference's own VTM-17.0 anchor points (tests/golden/make_bd_golden.py)."""
import copy
import json
import os

import numpy as np
import pytest

from dcvc_amd import bd_rate as bd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bd_rate_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_pchip_matches_scipy():
    scipy_interp = pytest.importorskip("scipy.interpolate")
    rng = np.random.default_rng(3)
    for n in (2, 3, 4, 7, 10):
        x = np.cumsum(rng.uniform(0.1, 2.0, n))
        for y in (rng.standard_normal(n), np.sort(rng.standard_normal(n)), -np.sort(rng.standard_normal(n))):
            xs = np.linspace(x[0], x[-1], 57)
            assert np.allclose(bd.pchip(x, y, xs), scipy_interp.pchip_interpolate(x, y, xs), rtol=1e-12, atol=1e-12)
    assert np.allclose(bd.pchip([0, 1, 2], [0, 1, 4], [0, 1, 2]), [0, 1, 4])       # interpolates its knots


def test_known_answers_on_the_anchor_points(gold):
    for seq, pts in bd.per_sequence(gold["anchor"]).items():
        rps = sorted(pts)
        bpp = [pts[r]["ave_all_frame_bpp"] for r in rps]
        psnr = [pts[r]["ave_all_frame_psnr"] for r in rps]
        assert bd.bd_rate(bpp[1::2], psnr[1::2], bpp[0::2], psnr[0::2]) == pytest.approx(gold["expected"]["split"][seq], abs=1e-9)
        assert bd.bd_rate(bpp, psnr, [0.9 * b for b in bpp], psnr) == pytest.approx(gold["expected"]["cheaper"][seq], abs=1e-9)
        assert bd.bd_rate(bpp, psnr, [0.9 * b for b in bpp], psnr) == pytest.approx(-10.0, abs=1e-9)
        assert bd.bd_rate(bpp, psnr, bpp, psnr) == pytest.approx(0.0, abs=1e-12)
        # order of the points does not matter
        assert bd.bd_rate(bpp[::-1], psnr[::-1], bpp[0::2], psnr[0::2]) == pytest.approx(
            bd.bd_rate(bpp, psnr, bpp[0::2], psnr[0::2]), abs=1e-12)


def test_compare_pipeline_class_and_sequence(gold):
    anchor = gold["anchor"]
    test = copy.deepcopy(anchor)
    for seqs in test.values():
        for pts in seqs.values():
            for e in pts.values():
                for k in ("ave_i_frame_bpp", "ave_all_frame_bpp"):
                    e[k] *= 0.8                         # a codec that needs 20 % fewer bits everywhere
    files = {"VTM": anchor, "better": test}
    by_class = bd.compare(files, "VTM", "class")
    assert set(by_class["better"]) == {"UVG", "HEVC_B"}
    for v in by_class["better"].values():
        assert v == pytest.approx(-20.0, abs=1e-9)
    by_seq = bd.compare(files, "VTM", "sequence")
    assert len(by_seq["better"]) == sum(len(s) for s in anchor.values())
    assert all(v == pytest.approx(-20.0, abs=1e-9) for v in by_seq["better"].values())
    # class averaging is weighted by frame counts (compare_bd_rate.py:98-146)
    avg = bd.class_average(anchor)["UVG"]
    rp = sorted(avg)[0]
    seqs = anchor["UVG"]
    w = np.array([s[rp]["i_frame_num"] + s[rp]["p_frame_num"] for s in seqs.values()], float)
    v = np.array([s[rp]["ave_all_frame_bpp"] for s in seqs.values()])
    assert avg[rp]["ave_all_frame_bpp"] == pytest.approx(float((w * v).sum() / w.sum()))
    # the admission rule: fewer than 3 rate points -> no number
    short = {"UVG": {seq: dict(list(pts.items())[:2]) for seq, pts in anchor["UVG"].items()}}
    assert bd.compare({"VTM": anchor, "short": short}, "VTM", "class")["short"] == {}


def test_cli(tmp_path, gold, capsys):
    a, b = tmp_path / "a.json", tmp_path / "b.json"
    a.write_text(json.dumps(gold["anchor"]))
    b.write_text(json.dumps(gold["anchor"]))
    assert bd.main(["--base_method", "VTM", "--log_paths", "VTM", str(a), "same", str(b)]) == 0
    out = capsys.readouterr().out
    assert "UVG" in out and "+0.00 %" in out


def test_rejects_disjoint_curves():
    with pytest.raises(ValueError):
        bd.bd_rate([1, 2, 3], [30, 31, 32], [1, 2, 3], [40, 41, 42])
    with pytest.raises(ValueError):
        bd.bd_rate([1, 0, 3], [30, 31, 32], [1, 2, 3], [30, 31, 32])
