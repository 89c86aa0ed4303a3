#!/usr/bin/env python3
"""4-frame calls through frt_pipeline_submit / wait in the library's modes: ms per call, calls per pass, merged tickets (GPU box).
    python tools/proxy_only.py [depth ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
frt = ge.load_pkg()
s = frt.synth
import tempfile
tmp = tempfile.mkdtemp()
dp = frt.write_weights(os.path.join(tmp, "d.frtw"), s.retinaface_state(1), 1)
rp = frt.write_weights(os.path.join(tmp, "r.frtw"), s.arcface_state(2, "ir", calib=s.load_calibration("ir")), 2)
B, K, H, W, nb = 32, 4, 640, 640, 4
det = frt.RetinaFace(dp, W, H, (3, H, W), B, K, 0.4, 0.6)
rec = frt.ArcFaceIR50(rp, W, H, (3, 112, 112), 512, B * K, K, 0.65)
rec.setGallery(s.make_gallery(1_000_000)); rec.initMatMul()
pipe = frt.Pipeline(det, rec, B)
st = torch.cuda.Stream(); pipe.set_stream(st.cuda_stream)
frames = [torch.from_numpy(s.make_frames(nb, H, W, start=10 * i)).pin_memory() for i in range(2)]
res = [torch.zeros(nb * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(12)]
frt.lib.frt_set_wait_spin_us(50000)
def leg(n, depth):
    tk = []
    for i in range(n):
        if len(tk) >= depth: pipe.wait(tk.pop(0))
        tk.append(pipe.submit(frames[i & 1].numpy(), res[i % 12].numpy().view(frt.RESULT_DTYPE)))
    for t in tk: pipe.wait(t)
for depth in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 8, 11]:
    for mode in (-1, -3, 0):
        pipe.set_pairing(mode)
        leg(24, depth)
        p0, s0 = pipe.pairing_stats(); m0 = pipe.merge_stats()
        t0 = time.perf_counter(); n = 600; leg(n, depth); dt = time.perf_counter() - t0
        p1, s1 = pipe.pairing_stats(); m1 = pipe.merge_stats()
        print("depth %2d mode %2d: %.4f ms/call  passes: %d shared %d single; merged calls %d carrying %d tickets" % (depth, mode, 1e3 * dt / n, p1 - p0, s1 - s0, m1[0] - m0[0], m1[1] - m0[1]), flush=True)
