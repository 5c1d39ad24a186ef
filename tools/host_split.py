"""Where does a step's wall time go? compress / decompress call durations (host view) with and
without a device synchronisation in between, per workload. Usage: python tools/host_split.py ld|hts|htl|intra"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "ld"
    device = torch.device("cuda", 0)
    import __graft_entry__
    __graft_entry__.build()
    cpu_net, gpu_net = bench.build_model(device)
    pics = bench.make_pictures(3, 0, device)
    pad_r, pad_b = gpu_net.get_padding_size(bench.HEIGHT, bench.WIDTH, 16)
    sps = {"height": bench.HEIGHT, "width": bench.WIDTH}
    if kind == "intra":
        enc = dec = gpu_net
        comp = lambda x, qp: enc.compress(x, qp, pad_b, pad_r)
        decomp = lambda r, qp: dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"])
        inputs = pics
    else:
        w = bench.InterWorkload(kind, device, pics, gpu_net, pad_b, pad_r)
        w.enc.add_ref_feature_from_frame(w.ref)
        w.dec.add_ref_feature_from_frame(w.ref, apply_feature_adaptor=False)
        comp = lambda x, qp: w.enc.compress(x, qp, 0, pad_b, pad_r)
        decomp = lambda r, qp: w.dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], 0)
        inputs = w.inputs
    for sync_between in (True, False):
        tc = td = ts = 0.0
        n = 0
        for i in range(14):
            x, qp = inputs[i % len(inputs)], 32
            torch.cuda.synchronize()
            a = time.perf_counter()
            r = comp(x, qp)
            b = time.perf_counter()
            if sync_between:
                torch.cuda.synchronize()
            c = time.perf_counter()
            decomp(r, qp)
            d = time.perf_counter()
            torch.cuda.synchronize()
            e = time.perf_counter()
            if i >= 4:
                tc += b - a; ts += c - b; td += d - c; n += 1
                tail = e - d
        print("%s sync_between=%s: compress call %.2f ms, gpu tail after compress %.2f ms, decompress call %.2f ms, "
              "gpu tail after decompress %.2f ms, bytes %d" % (kind, sync_between, 1e3 * tc / n, 1e3 * ts / n,
                                                              1e3 * td / n, 1e3 * tail, len(r["bit_stream"])))


if __name__ == "__main__":
    main()
