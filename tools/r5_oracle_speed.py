"""Round-5 probe (CPU only): seconds per compress() of the bit-exact oracle at 64x64 for the four codecs, for the oracle
library in place (oracle/liboracle.so) - run once per library variant / ORACLE_THREADS value on the host in question."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from codec_util import chunk, dmc_ht_model, dmc_ld_model, dmci_model, oracle_for, picture  # noqa: E402

out = []
m = dmci_model(skip_thres=0.15)
o = oracle_for(m)
o.compress(picture(64, 64), 5)
t = time.time()
for q in (3, 30, 60):
    r = o.compress(picture(64, 64), q)
    o.decompress(r["bit_stream"], q, 64, 64, r["ec_parallel"])
out.append("dmci %.2f" % ((time.time() - t) / 3))
for name in ("ld", "hts", "htl"):
    m = dmc_ld_model(skip_thres=0.15) if name == "ld" else dmc_ht_model(name, skip_thres=0.15)
    o = oracle_for(m)
    o.add_ref_feature_from_frame(picture(64, 64), True)
    x = picture(64, 64, index=1) if name == "ld" else chunk(64, 64, 1)
    o.compress(x, 5, False)
    t = time.time()
    for q in (3, 30, 60):
        o.compress(x, q, False)
    out.append("%s %.2f" % (name, (time.time() - t) / 3))
print(os.environ.get("ORACLE_VARIANT", "?"), "ORACLE_THREADS=%s" % os.environ.get("ORACLE_THREADS"), "  ".join(out), "(s per call)")
