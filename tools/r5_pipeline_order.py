"""Round-5 probe: does the ORDER of the loops inside one process change what the intra two-stage pipeline reaches?
  python tools/r5_pipeline_order.py pipe-first | plain-first | plain-long-first | fresh-encoder
One process per order (bench.py's own functions and objects)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")      # this probe is about the runtime's DEFAULT; bench.py itself sets 1
import bench  # noqa: E402


def main():
    order = sys.argv[1]
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device))
    import __graft_entry__
    __graft_entry__.build()
    cpu_net, gpu_net = bench.build_model(device)
    pics = bench.make_pictures(5, 0, device)
    pad_r, pad_b = gpu_net.get_padding_size(bench.HEIGHT, bench.WIDTH, 16)

    def plain(steps):
        w = bench.IntraWorkload(gpu_net, pics, pad_b, pad_r)
        bench.run_steps(w, 0, 10)
        dt, _ = bench.timed_region(w, bench.run_steps, 10, steps)
        return steps / dt

    def pipe(enc_net):
        if os.environ.get("FRESH_ENC"):          # what bench.py does since: an encoder object of its own, streams at low priority
            enc_net = bench._to_gpu(cpu_net, device)
        w = bench.IntraWorkload(enc_net, pics, pad_b, pad_r, bench._to_gpu(cpu_net, device), True)
        r = bench.pipelined_block(w, 0, 40, 10, 1.0)
        return r["value"], r["sustained"]["value"]

    if order == "pipe-first":
        print(order, "pipelined %.1f (sustained %.1f)" % pipe(gpu_net), "then plain %.1f" % plain(40))
    elif order == "plain-first":
        print(order, "plain %.1f" % plain(40), "then pipelined %.1f (sustained %.1f)" % pipe(gpu_net))
    elif order == "plain-long-first":
        print(order, "plain %.1f" % plain(400), "then pipelined %.1f (sustained %.1f)" % pipe(gpu_net))
    elif order == "fresh-encoder":
        # the plain loop on gpu_net as before, the pipeline with an encoder object that never decoded
        print(order, "plain %.1f" % plain(40), "then pipelined with a fresh encoder object %.1f (sustained %.1f)" % pipe(bench._to_gpu(cpu_net, device)))
    elif order == "seq-two":
        # ONE thread, compress then decompress, but on two objects (encoder streams low, decoder streams high): the
        # decoder's chain of short kernels and host round trips can run under the encoder's reconstruction tail
        a = plain(40)
        w = bench.IntraWorkload(bench._to_gpu(cpu_net, device), pics, pad_b, pad_r, bench._to_gpu(cpu_net, device), True)
        bench.run_steps(w, 0, 10)
        dt, _ = bench.timed_region(w, bench.run_steps, 10, 60)
        ok = bench.closure_ok(w, 70)
        print(order, "plain (one object) %.1f then two prioritised objects, same sequential loop %.1f closure %s" % (a, 60 / dt, ok))
    elif order == "seq-two-ld":
        w = bench.InterWorkload("ld", device, pics, gpu_net, pad_b, pad_r, False)
        bench.run_steps(w, 0, 12)
        dt, _ = bench.timed_region(w, bench.run_steps, 12, 96)
        a = 96 / dt
        w2 = bench.InterWorkload("ld", device, pics, gpu_net, pad_b, pad_r, True)
        bench.run_steps(w2, 0, 12)
        dt, _ = bench.timed_region(w2, bench.run_steps, 12, 96)
        print(order, "LD sequential loop, plain objects %.1f, prioritised objects %.1f" % (a, 96 / dt))
    elif order == "idle-between":
        a = plain(400)
        time.sleep(3.0)
        print(order, "plain %.1f" % a, "3 s idle, then pipelined %.1f (sustained %.1f)" % pipe(gpu_net))


if __name__ == "__main__":
    main()
