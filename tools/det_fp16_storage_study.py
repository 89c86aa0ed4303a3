#!/usr/bin/env python3
"""Would fp16 activation STORAGE (fp32 accumulation) keep the detector inside its parity tests?  (round-1 VERDICT, task 6)

Emulation on the CPU oracle: the RetinaFace-mnet0.25 forward of oracle/nets.py with the outputs of the first k body layers rounded to
fp16 and widened again (what storing them as fp16 in HBM does), compared with the all-fp32 forward on the same frames - head
differences and, after the reference's decode + NMS (oracle/postproc.c), the number of box coordinates whose integer truncation flips.
Runs anywhere (no GPU).   python tools/det_fp16_storage_study.py [frames]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

frt = g.load_pkg()
s = frt.synth
import oracle  # noqa: E402
from oracle import nets  # noqa: E402

sd = s.retinaface_state(1)


def forward(x, n_round):
    with torch.no_grad():
        x = torch.as_tensor(x, dtype=torch.float32)
        feats, k = [], 0
        for st in ("stage1", "stage2", "stage3"):
            for i, (kind, stride) in enumerate(nets._STAGES[st]):
                p = "body.%s.%d" % (st, i)
                x = nets._conv_bn(x, sd, p, stride) if kind == "bn" else nets._conv_dw(x, sd, p, stride)
                if k < n_round:
                    x = x.half().float()
                k += 1
            feats.append(x)
        o1 = F.relu(nets._bn(F.conv2d(feats[0], nets._t(sd, "fpn.output1.0.weight")), sd, "fpn.output1.1"))
        o2 = F.relu(nets._bn(F.conv2d(feats[1], nets._t(sd, "fpn.output2.0.weight")), sd, "fpn.output2.1"))
        o3 = F.relu(nets._bn(F.conv2d(feats[2], nets._t(sd, "fpn.output3.0.weight")), sd, "fpn.output3.1"))
        o2 = nets._conv_bn(o2 + F.interpolate(o3, size=[o2.size(2), o2.size(3)], mode="nearest"), sd, "fpn.merge2")
        o1 = nets._conv_bn(o1 + F.interpolate(o2, size=[o1.size(2), o1.size(3)], mode="nearest"), sd, "fpn.merge1")
        f = [nets._ssh(o1, sd, "ssh1"), nets._ssh(o2, sd, "ssh2"), nets._ssh(o3, sd, "ssh3")]
        locs, confs = [], []
        for i, t in enumerate(f):
            b = F.conv2d(t, nets._t(sd, "BboxHead.%d.conv1x1.weight" % i), nets._t(sd, "BboxHead.%d.conv1x1.bias" % i))
            c = F.conv2d(t, nets._t(sd, "ClassHead.%d.conv1x1.weight" % i), nets._t(sd, "ClassHead.%d.conv1x1.bias" % i))
            locs.append(b.permute(0, 2, 3, 1).contiguous().view(b.shape[0], -1, 4))
            confs.append(c.permute(0, 2, 3, 1).contiguous().view(c.shape[0], -1, 2))
        return torch.cat(locs, 1).numpy(), F.softmax(torch.cat(confs, 1), dim=-1).numpy()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    frames = s.make_frames(n, 640, 640, start=500)
    x = np.stack([oracle.det_preprocess(f, 640, 640) for f in frames])
    l0, c0 = forward(x, 0)
    print("RetinaFace-mnet0.25, %d synthetic 640x640 frames, heads and boxes vs the all-fp32 forward (test tolerances: |dloc| < 2e-4, |dconf| < 2e-5)" % n)
    for nr in (1, 2, 4, 14):
        l, c = forward(x, nr)
        flips = tot = cnt = 0
        for i in range(n):
            a = oracle.postprocess(l0[i], c0[i], 640, 640, 640, 640, 0.4, 0.6, 4)
            b = oracle.postprocess(l[i], c[i], 640, 640, 640, 640, 0.4, 0.6, 4)
            if len(a) != len(b):
                cnt += 1
                continue
            for k in ("x1", "y1", "x2", "y2"):
                flips += int((a[k] != b[k]).sum())
                tot += len(a)
        print("fp16 storage of the first %2d body layers: max|dloc| %.2e  max|dconf| %.2e  flipped box coordinates %d/%d  frames with a different box count %d"
              % (nr, np.abs(l - l0).max(), np.abs(c - c0).max(), flips, tot, cnt))


if __name__ == "__main__":
    main()
