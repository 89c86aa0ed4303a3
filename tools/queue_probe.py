#!/usr/bin/env python3
"""Which stream creation disturbs the stage pipeline?    Builds a small pipeline (batch 4), then adds the process's other
stream users one at a time - a torch stream, torch.distributed's RCCL process group, libfrt's own communicator (frt_comm_create) - and after
each step prints frt_pipeline_check_overlap's ratio and the measured ms per 4-frame step (frt_pipeline_submit / wait, 3 in flight).  GPU box only.

    python tools/queue_probe.py [order]      order: letters t (torch stream), d (torch.distributed nccl), c (frt comm), g (first all-gather), m (create the caller's stream here, not first); default "tdc"
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    order = sys.argv[1] if len(sys.argv) > 1 else "tdc"
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    frt = entry.load_pkg()
    s = frt.synth
    tmp = tempfile.mkdtemp(prefix="frt_qp_")
    dpath = frt.write_weights(os.path.join(tmp, "det.frtw"), s.retinaface_state(1), 1)
    rpath = frt.write_weights(os.path.join(tmp, "rec.frtw"), s.arcface_state(2, "ir", calib=s.load_calibration("ir")), 2)
    B, K, H, W = 4, 4, 640, 640
    torch.cuda.set_device(0)
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, (3, 112, 112), 512, B * K, K, 0.65)
    rec.setGallery(s.make_gallery(100000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    if "m" not in order:  # 'm' in the order string: create the caller's stream at that point instead of here
        main_s = torch.cuda.Stream()
        torch.cuda.set_stream(main_s)
        pipe.set_stream(main_s.cuda_stream)
    frames = [torch.from_numpy(s.make_frames(B, H, W, start=i * 8)).pin_memory() for i in range(2)]
    res = [torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(4)]

    def step_ms(n=300):
        t = []
        for i in range(20 + n):
            if i == 20:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if len(t) >= 3:
                pipe.wait(t.pop(0))
            t.append(pipe.submit(frames[i & 1].numpy(), res[i % 4].numpy().view(frt.RESULT_DTYPE)))
        for x in t:
            pipe.wait(x)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    def report(tag):
        r, w = pipe.check_overlap()
        print("%-28s overlap ratio %.2f   %.3f ms per %d-frame step%s" % (tag, r, step_ms(), B, ("   WARNING: " + w[:110]) if w else ""), flush=True)

    if "m" not in order:
        report("pipeline only")
    keep = []
    for ch in order:
        if ch == "m":
            main_s = torch.cuda.Stream()
            torch.cuda.set_stream(main_s)
            pipe.set_stream(main_s.cuda_stream)
            report("+ caller's stream created")
        elif ch == "x":  # a dummy stream, no report: shifts the round-robin position of whatever is created next
            keep.append(torch.cuda.Stream())
        elif ch == "t":
            keep.append(torch.cuda.Stream())
            report("+ torch stream")
        elif ch == "d":
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            x = torch.zeros(4, device="cuda")
            dist.all_reduce(x)
            torch.cuda.synchronize()
            report("+ torch.distributed nccl")
        elif ch == "c":
            keep.append(frt.Comm(frt.comm_unique_id(), 0, 1, 0))
            report("+ frt_comm_create")
        elif ch == "g":
            a = torch.zeros(64, device="cuda")
            b = torch.zeros(64, device="cuda")
            keep[-1].all_gather(a.data_ptr(), b.data_ptr(), 256)
            keep[-1].sync()
            report("+ first frt_comm_all_gather")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
