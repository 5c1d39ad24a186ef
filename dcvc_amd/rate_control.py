"""Rate-control hook of the coding loop.

The reference harness codes a whole sequence with ONE q_index per rate point (test_video.py:512-514:
`linspace(0, 63, rate_num)`), but its stream container already carries the q_index of every picture
(stream_helper.py:134-135 write_ip / read_ip_remaining), so a decoder needs no side information when the encoder
changes it from picture to picture. This module is the encoder-side hook for that: a controller object the coding
loop asks for the q_index of the next coded unit and tells the bits the unit took, plus `code_sequence`, the
loop of test_video.py:204-257 (picture types, reset rule, chunk padding) with the q_index taken from the
controller instead of a constant. The native tool (csrc/cli/dcvc_cli.hip) keeps the reference's constant
--qp-i / --qp-p; this is the Python-surface hook.

In DCVC-UF a HIGHER q_index means finer quantisation = more bits (q 0 .. 63).
"""
import math


class ConstantQP:
    """The reference behaviour: one q_index for I pictures, one for P units."""

    def __init__(self, qp_i, qp_p=None):
        self.qp_i, self.qp_p = int(qp_i), int(qp_i if qp_p is None else qp_p)

    def next_qp(self, is_intra):
        return self.qp_i if is_intra else self.qp_p

    def update(self, bits, pictures, is_intra):
        pass


class TargetBpp:
    """One-pass control towards an average of `target_bpp` bits per pixel over the sequence.

    Model: log2(bits of a unit) is roughly linear in the q_index (measured on the synthetic 1080p intra pictures:
    115 KB at q 0, 971 KB at q 63 -> 0.049 per step); the slope is re-estimated from the units seen so far. After
    every unit the controller compares the bits spent with the budget of the pictures coded and moves the q_index
    by the step count that would cancel the error over the next `horizon` pictures. I pictures get `intra_bonus`
    steps (they anchor a whole GOP)."""

    def __init__(self, target_bpp, pixels_per_picture, qp0=32, horizon=8, intra_bonus=0, qp_min=0, qp_max=63,
                 slope=0.049):
        if target_bpp <= 0 or pixels_per_picture <= 0:
            raise ValueError("target_bpp and pixels_per_picture must be positive")
        self.target_bits = float(target_bpp) * pixels_per_picture          # per picture
        self.qp = float(qp0)
        self.horizon, self.intra_bonus = max(1, int(horizon)), int(intra_bonus)
        self.qp_min, self.qp_max = int(qp_min), int(qp_max)
        self.slope = float(slope)
        self.max_step = 4.0                                                # q_index steps per update
        self.spent, self.pictures = 0.0, 0
        self._last = None                                                  # (qp, log2 bits per picture) of the last P unit

    def next_qp(self, is_intra):
        q = self.qp + (self.intra_bonus if is_intra else 0)
        return int(min(self.qp_max, max(self.qp_min, round(q))))

    def update(self, bits, pictures, is_intra):
        if pictures <= 0:
            return
        self.spent += bits
        self.pictures += pictures
        per_picture = max(bits / pictures, 1.0)
        used_qp = self.next_qp(is_intra)
        if not is_intra:
            if self._last is not None and used_qp != self._last[0]:
                s = (math.log2(per_picture) - self._last[1]) / (used_qp - self._last[0])
                if 0.005 < s < 0.5:                                        # keep a sane, positive slope
                    self.slope = 0.75 * self.slope + 0.25 * s
            self._last = (used_qp, math.log2(per_picture))
        # bits the next `horizon` pictures may take so that the running average lands on the target
        budget = self.target_bits * (self.pictures + self.horizon) - self.spent
        want = max(budget / self.horizon, self.target_bits / 64.0)
        if is_intra:
            # An I picture costs several times the per-picture target by design: its bits count in `spent` (the budget above
            # already pays them back over the horizon), but its SIZE says nothing about what the P units cost at this qp, so it
            # does not enter the proportional term (advisor, round 3: it used to drop qp by ~34 steps behind every I picture).
            # Only the budget it leaves moves qp, measured against the last P unit if there is one.
            if self._last is None:
                return
            # ... projected to the q_index the controller stands at NOW: the last P unit was coded at self._last[0], and the
            # update behind it has already moved self.qp - comparing its raw size again would correct the same error twice
            # (advisor, round 4: q saw-toothed between 0 and 12 with intra period 32 in a simulation)
            per_picture = 2.0 ** (self._last[1] + self.slope * (self.qp - self._last[0]))
        step = (math.log2(want) - math.log2(per_picture)) / self.slope
        self.qp += min(self.max_step, max(-self.max_step, step))           # bounded step: no oscillation with the intra period
        self.qp = min(float(self.qp_max), max(float(self.qp_min), self.qp))

    @property
    def spent_bits_per_picture(self):
        return self.spent / max(self.pictures, 1)


def code_sequence(frame_count, frames_per_p, code_intra, code_inter, controller, intra_period=-1, reset_interval=32,
                  force_intra=False):
    """The encode loop of test_video.py:204-257 with the q_index from `controller`.

    code_intra(frame_idx, qp) -> bytes-like          one I picture
    code_inter(frame_idx, n, qp, reset) -> bytes-like a P unit of frames_per_p pictures starting at frame_idx, of which
                                                      the first n exist in the source (the caller pads the rest by
                                                      repeating the last picture, test_video.py:104-110)
    Returns [(is_intra, qp, reset, payload)] in coding order - what write_ip stores per unit."""
    units, idx = [], 0
    while idx < frame_count:
        # test_video.py:204-213: frame 0; every frame when intra_period == 1 (or --force_intra); with intra_period > 1
        # every frame with index % intra_period == 1 other than frame 1
        intra = idx == 0 or force_intra or intra_period == 1 or (intra_period > 1 and idx != 1 and idx % intra_period == 1)
        if intra:
            qp = controller.next_qp(True)
            payload = code_intra(idx, qp)
            controller.update(8 * len(payload), 1, True)
            units.append((True, qp, False, payload))
            idx += 1
            continue
        n = min(frames_per_p, frame_count - idx)
        reset = reset_interval > 0 and (idx + frames_per_p) % reset_interval == 1
        qp = controller.next_qp(False)
        payload = code_inter(idx, n, qp, reset)
        controller.update(8 * len(payload), n, False)
        units.append((False, qp, reset, payload))
        idx += frames_per_p
    return units
