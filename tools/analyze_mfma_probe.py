"""Offline analysis of gpurun_out/mfma_probe.bin: which arithmetic model does
v_mfma_f32_32x32x16_f16 follow? (see tools/mfma_probe.hip)"""
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe.bin"
raw = open(path, "rb").read()
magic, n_des, n_rand = np.frombuffer(raw[:12], np.int32)
assert magic == 0x4d464d41
des_dt = np.dtype([("a", np.float16, 16), ("b", np.float16, 16), ("c", np.float32), ("d", np.float32),
                   ("mism", np.int32)])
des = np.frombuffer(raw[12:12 + n_des * des_dt.itemsize], des_dt)
off = 12 + n_des * des_dt.itemsize
rnd_dt = np.dtype([("A", np.float16, (32, 16)), ("Bt", np.float16, (32, 16)), ("C", np.float32, (32, 32)),
                   ("D", np.float32, (32, 32))])
rnd = np.frombuffer(raw[off:off + n_rand * rnd_dt.itemsize], rnd_dt)
print("designed", n_des, "random", n_rand, "lane-nonuniform designed trials:", int((des["mism"] != 0).sum()))


def f32(x):
    return np.float32(x)


def model_exact(a, b, c):
    """exact sum of products + c, one rounding (RNE)"""
    s = float(c)
    from fractions import Fraction
    t = Fraction(float(c))
    for x, y in zip(a, b):
        t += Fraction(float(x)) * Fraction(float(y))
    return np.float32(float(t)) if abs(t) < 1e300 else np.float32(s)


def rne32(fr):
    from fractions import Fraction
    # exact Fraction -> float32 RNE via float64 is unsafe (double rounding); do it exactly
    if fr == 0:
        return np.float32(0.0)
    import math
    sign = -1 if fr < 0 else 1
    fr = abs(fr)
    e = math.floor(math.log2(fr)) if fr > 0 else 0
    # adjust e so that 2^e <= fr < 2^(e+1)
    while Fraction(2) ** e > fr:
        e -= 1
    while Fraction(2) ** (e + 1) <= fr:
        e += 1
    e = max(e, -126)
    q = fr / (Fraction(2) ** (e - 23))
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(sign * float(n) * 2.0 ** (e - 23))


def model_exact_fr(a, b, c):
    from fractions import Fraction
    t = Fraction(float(c))
    for x, y in zip(a, b):
        t += Fraction(float(x)) * Fraction(float(y))
    return rne32(t)


def model_seq(a, b, c, order=range(16)):
    acc = np.float32(c)
    for k in order:
        acc = np.float32(np.float64(acc) + np.float64(a[k]) * np.float64(b[k]))
    return acc


def model_groups(a, b, c, g, c_first=True):
    """exact sum inside groups of g products (+ running acc), rounded once per group"""
    from fractions import Fraction
    acc = Fraction(float(c))
    for s in range(0, 16, g):
        t = acc
        for k in range(s, s + g):
            t += Fraction(float(a[k])) * Fraction(float(b[k]))
        acc = Fraction(float(rne32(t)))
    return np.float32(float(acc))


models = {
    "exact16": model_exact_fr,
    "seq0..15": lambda a, b, c: model_seq(a, b, c),
    "groups of 4": lambda a, b, c: model_groups(a, b, c, 4),
    "groups of 8": lambda a, b, c: model_groups(a, b, c, 8),
    "groups of 2": lambda a, b, c: model_groups(a, b, c, 2),
}

# designed trials
fam = [("F1 single product", 0, 16 * 5 * 2 * 2)]
res = {m: 0 for m in models}
bad_examples = {m: [] for m in models}
for i, t in enumerate(des):
    for m, fn in models.items():
        w = fn(t["a"], t["b"], t["c"])
        if w.tobytes() == t["d"].tobytes() or (w == 0 and t["d"] == 0):
            res[m] += 1
        elif len(bad_examples[m]) < 6:
            bad_examples[m].append((i, float(w), float(t["d"])))
print("designed trials matched per model (of %d):" % n_des, res)
for m in models:
    print(" ", m, "first mismatches:", bad_examples[m][:4])

# random trials: sample outputs
rs = np.random.default_rng(0)
cnt = {m: 0 for m in models}
N = 0
for t in rnd[:: max(1, n_rand // 96)]:
    for _ in range(40):
        i, j = rs.integers(0, 32, 2)
        a, b, c, d = t["A"][i], t["Bt"][j], t["C"][i, j], t["D"][i, j]
        N += 1
        for m, fn in models.items():
            w = fn(a, b, c)
            if w.tobytes() == d.tobytes() or (w == 0 and d == 0):
                cnt[m] += 1
print("random samples matched (of %d):" % N, cnt)
