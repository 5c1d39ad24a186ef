#!/usr/bin/env python
"""Do two hosts generate the same synthetic weights / pictures? Prints sha256 digests of the pieces
the full-size oracle digests depend on, under the default CPU dispatch of torch / numpy and with the
dispatch pinned (ATEN_CPU_CAPABILITY, NPY_DISABLE_CPU_FEATURES). Run here and on the GPU box."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
print("  torch capability:", torch.backends.cpu.get_cpu_capability(), " threads", torch.get_num_threads())
g = torch.Generator().manual_seed(0)
print("  randn fp32       :", sha(torch.randn((1000, 37), generator=g).numpy()))
print("  rand fp32        :", sha(torch.rand((1000, 37), generator=g).numpy()))
print("  exp fp32         :", sha(torch.exp((torch.arange(64, dtype=torch.float32)[:, None] - 32.0) / 48.0).numpy()))
from dcvc_amd import arch, models, synthetic
sd = synthetic.synthetic_state_dict(arch.dmci_spec(), 0)
h = hashlib.sha256()
for k in sorted(sd):
    h.update(sd[k].numpy().astype(np.float16).tobytes())
print("  dmci weights fp16:", h.hexdigest()[:16])
m = models.DMCI(); m.load_state_dict(sd); m.update(0.15)
print("  cdf tables       :", [sha(np.asarray(t)) for t in m.get_cdf_info()])
y, uv = synthetic.synthetic_frame_yuv420(256, 256, 3, 0)
print("  picture u8       :", sha(y), sha(uv))
rng = np.random.default_rng(0); n = rng.standard_normal((300, 400)).astype(np.float32)
print("  np rfft2 / mean / std:", sha(np.fft.rfft2(n)), sha(np.array(n.mean())), sha(np.array(n.std())))
''' % (ROOT, ROOT)

for label, env in (("default", {}),
                   ("ATEN_CPU_CAPABILITY=avx2", {"ATEN_CPU_CAPABILITY": "avx2"}),
                   ("ATEN_CPU_CAPABILITY=default", {"ATEN_CPU_CAPABILITY": "default"}),
                   ("avx2 + numpy without AVX512", {"ATEN_CPU_CAPABILITY": "avx2",
                                                    "NPY_DISABLE_CPU_FEATURES": "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"}),
                   ("1 thread", {"OMP_NUM_THREADS": "1"})):
    print(label, flush=True)
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, "-c", CHILD], env=e)
