"""dcvc_amd/rate_control.py: the rate-control hook of the coding loop on a stand-in codec whose
bits follow the measured rate law of the intra model (log2 bits linear in the q_index)."""
import math

import numpy as np
import pytest

from dcvc_amd import rate_control as rc


def _fake_codec(pixels, seed=0):
    rng = np.random.default_rng(seed)

    def bits(qp, pictures, intra):
        base = (0.45 if intra else 0.02) * pixels            # bits per picture at q 0 (P: 0.02 .. 0.18 bpp over q 0 .. 63)
        noise = 2.0 ** rng.normal(0.0, 0.08)
        return int(pictures * base * 2.0 ** (0.05 * qp) * noise)

    return bits


def test_constant_qp_is_the_reference_behaviour():
    c = rc.ConstantQP(30, 36)
    units = rc.code_sequence(11, 8, lambda i, q: b"x" * 10, lambda i, n, q, r: b"y" * n, c, intra_period=-1, reset_interval=32)
    assert [(u[0], u[1]) for u in units] == [(True, 30), (False, 36), (False, 36)]
    assert [len(u[3]) for u in units] == [10, 8, 2]          # the last chunk holds 2 source pictures


@pytest.mark.parametrize("intra_period,delay,want", [
    (-1, 1, [0]), (1, 1, list(range(10))), (4, 1, [0, 5, 9]), (8, 8, [0, 9]),
])
def test_picture_types_follow_the_reference_loop(intra_period, delay, want):
    """test_video.py:204-213: frame 0 is intra; intra_period 1 = all intra; intra_period > 1: every frame with
    index % intra_period == 1 other than frame 1."""
    seen = []
    rc.code_sequence(10 if delay == 1 else 18, delay, lambda i, q: seen.append(i) or b"", lambda i, n, q, r: b"", rc.ConstantQP(1),
                     intra_period=intra_period)
    assert seen == (want if delay == 1 else [0, 9, 17][:len(seen)])


def test_reset_rule():
    resets = []

    def inter(i, n, q, r):
        if r:
            resets.append(i)
        return b""

    rc.code_sequence(70, 1, lambda i, q: b"", inter, rc.ConstantQP(1), reset_interval=32)
    assert resets == [32, 64]                                 # (frame_idx + 1) % 32 == 1
    del resets[:]
    rc.code_sequence(100, 8, lambda i, q: b"", inter, rc.ConstantQP(1), reset_interval=32)
    assert resets == [25, 57, 89]                             # (frame_idx + 8) % 32 == 1


@pytest.mark.parametrize("target", [0.03, 0.06, 0.12])
def test_target_bpp_controller_lands_on_the_budget(target):
    pixels = 1920 * 1080
    bits = _fake_codec(pixels)
    ctl = rc.TargetBpp(target, pixels, qp0=32)
    units = rc.code_sequence(
        241, 8,
        lambda i, q: b"\0" * (bits(q, 1, True) // 8),
        lambda i, n, q, r: b"\0" * (bits(q, n, False) // 8),
        ctl, intra_period=-1)
    total = sum(8 * len(u[3]) for u in units)
    got = total / 241 / pixels
    assert got == pytest.approx(target, rel=0.08), (got, [u[1] for u in units])
    assert len({u[1] for u in units}) > 1                    # the q_index really changes from unit to unit ...
    assert all(0 <= u[1] <= 63 for u in units)               # ... inside the model's range
    assert ctl.spent_bits_per_picture == pytest.approx(total / 241)


def test_units_carry_what_the_container_stores():
    """every unit = (picture type, q_index, reset flag, payload): exactly the fields write_ip() puts in front of the
    payload (stream_helper.py:130-141), so a decoder reads the controller's decisions from the stream itself"""
    import io
    from dcvc_amd import stream_helper as sh
    units = rc.code_sequence(17, 8, lambda i, q: bytes([q]) * 5, lambda i, n, q, r: bytes([q]) * 7,
                             rc.TargetBpp(0.01, 64 * 64, qp0=40), reset_interval=16)
    out = io.BytesIO()
    for intra, qp, reset, payload in units:
        sh.write_ip(out, intra, 0, qp, 1, reset, payload)
    f = io.BytesIO(out.getvalue())
    for intra, qp, reset, payload in units:
        h = sh.read_header(f)
        assert (h["nal_type"] == sh.NalType.NAL_I) == intra
        got_qp, ec, got_reset, got = sh.read_ip_remaining(f)
        assert (got_qp, bool(got_reset), bytes(got)) == (qp, bool(reset), payload)


def test_an_intra_picture_does_not_kick_the_controller():
    """advisor, round 3: an I picture costs ~ 20x a P picture here; its bits enter the budget, not the proportional term - the
    q_index of the P units behind every I picture stays within the bounded step of the one in front of it, and the sequence
    still lands near the budget."""
    pixels = 1920 * 1080
    bits = _fake_codec(pixels, seed=3)
    target = 0.06
    c = rc.TargetBpp(target, pixels, qp0=30)
    units = rc.code_sequence(161, 1, lambda i, q: b"\0" * (bits(q, 1, True) // 8), lambda i, n, q, r: b"\0" * (bits(q, n, False) // 8), c,
                             intra_period=32)
    qps = [u[1] for u in units]
    kinds = [u[0] for u in units]
    for k in range(2, len(units) - 1):
        if kinds[k]:                                           # P in front of the I picture, P behind it
            assert abs(qps[k + 1] - qps[k - 1]) <= 2 * c.max_step + 1, (k, qps[k - 1:k + 2])
    steps = [abs(qps[k + 1] - qps[k]) for k in range(len(units) - 1) if not kinds[k] and not kinds[k + 1]]
    assert max(steps) <= c.max_step + 1
    assert abs(c.spent_bits_per_picture / pixels - target) / target < 0.2


def test_the_update_behind_an_intra_picture_projects_the_last_p_unit_to_the_current_q_index():
    """advisor, round 4: behind an I picture the controller compares the budget with the last P unit's size - which was
    measured at THAT unit's q_index, while the update behind the P unit has already moved the controller. Comparing the raw
    size again corrects the same error twice; the size is projected along the model's slope to the q_index the controller
    stands at now."""
    import math
    pixels = 1920 * 1080
    c = rc.TargetBpp(0.05, pixels, qp0=30, horizon=8)
    p_bits = 4.0 * c.target_bits                      # a P picture four times over budget at q 30
    assert c.next_qp(False) == 30
    c.update(p_bits, 1, False)
    q_after_p = c.qp
    assert q_after_p == 30 - c.max_step              # the bounded step down
    spent, pictures = c.spent, c.pictures
    i_bits = 10.0 * c.target_bits
    c.update(i_bits, 1, True)
    budget = c.target_bits * (pictures + 1 + c.horizon) - (spent + i_bits)
    want = max(budget / c.horizon, c.target_bits / 64.0)
    projected = math.log2(p_bits) + c.slope * (q_after_p - 30)
    step = (math.log2(want) - projected) / c.slope
    assert c.qp == pytest.approx(min(63.0, max(0.0, q_after_p + min(c.max_step, max(-c.max_step, step)))))
    # and the projection matters: against the raw size the step would have been larger by (q_after_p - 30) model steps
    raw_step = (math.log2(want) - math.log2(p_bits)) / c.slope
    assert raw_step == pytest.approx(step + (q_after_p - 30))
