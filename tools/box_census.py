#!/usr/bin/env python3
"""Box-flip census (round-1 VERDICT, parity item 3): how often does the HIP detector's fp32 summation order flip an `int`
truncation of the box decode (src/retinaface.cpp:171-187) relative to the fp32 oracle?  Runs findFace on N synthetic frames per
geometry on the GPU and the oracle (torch-CPU fp32 + oracle/postproc.c) on the same frames, and writes the histogram of coordinate
differences as JSON.  GPU box only (needs libfrt + a device); the oracle is the checker.

    python tools/box_census.py --frames 512 --out gpurun_out/r02_box_census.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


GEOMETRIES = {"640x640->640x640": (640, 640, 640, 640), "1920x1080->640x640": (1920, 1080, 640, 640), "640x480->320x288": (640, 480, 320, 288)}


def census(frt, dsd, dpath, geometry, n_frames, K=4, CH=16, start=1000):
    """One geometry (frame_w, frame_h, in_w, in_h): GPU findFace vs fp32 oracle + oracle/postproc.c on ``n_frames`` synthetic frames."""
    import oracle
    from oracle import nets
    s = frt.synth
    fw, fh, in_w, in_h = geometry
    det = frt.RetinaFace(dpath, fw, fh, (3, in_h, in_w), CH, K, 0.4, 0.6)
    hist, score_max, n_boxes, count_mismatch, t0 = {}, 0.0, 0, 0, time.time()
    for c0 in range(0, n_frames, CH):
        frames = s.make_frames(CH, fh, fw, start=start + c0)
        got = det.findFaceBatch(frames)
        x = np.stack([oracle.det_preprocess(fr, in_h, in_w) for fr in frames])
        loc, conf = nets.retinaface_forward(dsd, x)
        for f in range(CH):
            want = oracle.postprocess(loc[f], conf[f], in_w, in_h, fw, fh, 0.4, 0.6, K)
            if len(want) != len(got[f]):
                count_mismatch += 1
                continue
            for k in ("x1", "y1", "x2", "y2"):
                for d in np.abs(got[f][k].astype(np.int64) - want[k].astype(np.int64)):
                    hist[int(d)] = hist.get(int(d), 0) + 1
            if len(want):
                score_max = max(score_max, float(np.abs(got[f]["score"] - want["score"]).max()))
            n_boxes += len(want)
    det.close()
    coords = sum(hist.values())
    return {"frames": n_frames, "boxes": n_boxes, "coordinates": coords, "abs_diff_histogram": {str(k): v for k, v in sorted(hist.items())},
            "coordinates_differing": coords - hist.get(0, 0), "fraction_differing": (coords - hist.get(0, 0)) / max(coords, 1),
            "max_abs_diff_px": max(hist) if hist else 0, "frames_with_different_box_count": count_mismatch,
            "max_abs_score_diff": score_max, "seconds": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    import __graft_entry__ as entry
    frt = entry.load_pkg()
    s = frt.synth
    tmp = tempfile.mkdtemp(prefix="frt_census_")
    dsd = s.retinaface_state(1)
    dpath = frt.write_weights(os.path.join(tmp, "det.frtw"), dsd, 1)
    report = {"faces_per_frame_cap": 4, "threads": int(torch.get_num_threads()), "geometries": {}}
    for name, geo in GEOMETRIES.items():
        report["geometries"][name] = census(frt, dsd, dpath, geo, args.frames)
        print(name, json.dumps(report["geometries"][name]), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
