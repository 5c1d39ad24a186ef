"""state_dict -> .dcvw, the flat weight file the standalone encoder / decoder (dcvc_amd/bin/dcvc)
loads - everything DMC*Proxy.set_param receives (the module's state_dict in fp16 plus the four
int32 CDF tensors of update(), common_model.py:64-70), so the native tool needs no Python.

  python -m dcvc_amd.export_weights --model dmci|ld|hts|htl (--checkpoint ckpt.pth.tar | --synthetic SEED)
                                    [--skip-thres 0.15] -o model.dcvw

File: "DCVW1\\0\\0\\0", u32 kind (0 dmci, 1 ld, 2 hts, 3 htl), f32 skip_thres, u32 tensor count, then per
tensor: u16 name length, name, u8 dtype (0 fp16, 1 fp32, 2 int32), u8 ndim, i64 dims[ndim],
u64 byte count, data (each record padded to 8 bytes). Little endian.
"""
import argparse
import struct
import sys

import numpy as np
import torch

KINDS = {"dmci": 0, "ld": 1, "hts": 2, "htl": 3}


def build_model(kind, checkpoint=None, synthetic_seed=None):
    from dcvc_amd import arch, models, synthetic
    if kind == "dmci":
        net, spec = models.DMCI(), arch.dmci_spec()
    elif kind == "ld":
        net, spec = models.DMC(), arch.dmc_ld_spec()
    else:
        net, spec = models.DMCHT(kind), arch.dmc_ht_spec(kind == "hts")
    if checkpoint is not None:
        ckpt = torch.load(checkpoint, map_location="cpu")
        for key in ("state_dict", "net"):
            if isinstance(ckpt, dict) and key in ckpt:
                ckpt = ckpt[key]
        net.load_state_dict({k.replace("module.", ""): v for k, v in ckpt.items()}, strict=True)
    else:
        net.load_state_dict(synthetic.synthetic_state_dict(spec, synthetic_seed or 0))
    return net


def write_dcvw(path, kind, net, skip_thres):
    net.update(skip_thres)
    sd = net.add_cdf_to_state_dict(net.state_dict())
    with open(path, "wb") as f:
        f.write(b"DCVW1\0\0\0")
        f.write(struct.pack("<IfI", KINDS[kind], float(skip_thres), len(sd)))
        for name, t in sd.items():
            a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
            if a.dtype.kind == "f":
                a, code = a.astype(np.float16), 0
            elif a.dtype == np.int32:
                code = 2
            else:
                raise TypeError("%s: unsupported dtype %s" % (name, a.dtype))
            a = np.ascontiguousarray(a)
            nb = name.encode()
            rec = struct.pack("<H", len(nb)) + nb + struct.pack("<BB", code, a.ndim)
            rec += struct.pack("<%dq" % a.ndim, *a.shape) + struct.pack("<Q", a.nbytes)
            f.write(rec)
            f.write(a.tobytes())
            pad = (-(len(rec) + a.nbytes)) % 8
            f.write(b"\0" * pad)
    return len(sd)


def read_dcvw(path):
    """-> (kind name, skip_thres, {name: numpy array}) - the inverse of write_dcvw (tests, inspection)"""
    names = {v: k for k, v in KINDS.items()}
    dtypes = {0: np.float16, 1: np.float32, 2: np.int32}
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"DCVW1\0\0\0":
        raise ValueError("not a .dcvw file")
    kind, skip_thres, count = struct.unpack_from("<IfI", data, 8)
    pos, out = 20, {}
    for _ in range(count):
        start = pos
        (nlen,) = struct.unpack_from("<H", data, pos)
        name = data[pos + 2:pos + 2 + nlen].decode()
        pos += 2 + nlen
        code, ndim = struct.unpack_from("<BB", data, pos)
        pos += 2
        dims = struct.unpack_from("<%dq" % ndim, data, pos)
        pos += 8 * ndim
        (nbytes,) = struct.unpack_from("<Q", data, pos)
        pos += 8
        out[name] = np.frombuffer(data, dtype=dtypes[code], count=nbytes // np.dtype(dtypes[code]).itemsize, offset=pos).reshape(dims)
        pos += nbytes
        pos += (-(pos - start)) % 8
    if pos != len(data):
        raise ValueError("trailing bytes in .dcvw file")
    return names[kind], skip_thres, out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", required=True, choices=tuple(KINDS))
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--checkpoint")
    g.add_argument("--synthetic", type=int, metavar="SEED")
    ap.add_argument("--skip-thres", type=float, default=0.0)
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args(argv)
    net = build_model(a.model, a.checkpoint, a.synthetic)
    n = write_dcvw(a.output, a.model, net, a.skip_thres)
    print("wrote %s: %d tensors (%s, skip_thres %g)" % (a.output, n, a.model, a.skip_thres))
    return 0


if __name__ == "__main__":
    sys.exit(main())
