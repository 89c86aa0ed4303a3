#!/usr/bin/env python3
"""Dynamic-range sweep of the recogniser (round-2 VERDICT, robustness item 7a): where does the fp16 activation storage of the HIP
ArcFace path stop matching the fp32 oracle, and does anything saturate silently?

All parity evidence sits on synthetic weights built to keep activations O(1) (synth.py).  A trained backbone_ir50_asia.pth may keep its
residual stream or its branch activations orders of magnitude away from that.  This tool moves them there WITHOUT changing the function the
network computes (so the fp32 oracle's embedding is the same up to rounding and cosines are comparable across scales):

  stream scale s   the residual stream x_l (tensors Y / Z / SC of csrc, stored as fp16) becomes s * x_l: input BN and PReLU output * s, every
                   unit's leading BN and shortcut BN take mean * s, var' = s^2 (var + eps) - eps, every unit's closing BN (and shortcut
                   BN) gamma, beta * s, output_layer.0 un-scales.  PReLU, MaxPool, conv are positively homogeneous.
  branch scale t   the activation between conv1 and conv2 (tensor T, fp16; conv1's accumulators) becomes t * (...): conv1 weights * t,
                   conv2 weights / t (both are fp16 in the product: the weights leave their trained range too, as they would in such a net).

Round 5: libfrt conditions every unit's branch at load (powers of two on conv1 rows / conv2 columns + rows / the closing BatchNorm's scale, the same
function exactly - csrc/frt_embedder.cpp), so the branch sweep is flat now (5.5e-6 from 1e-4 to 1e4, profiles/r05e_dynamic_range.json; round 3:
4.8e-4 at 1e-4, non-finite at 1e4).  The tuning build's FRT_ARC_CONDITION=0 restores the unconditioned load for an A/B.

    python tools/dynamic_range_sweep.py --out gpurun_out/r03_dynamic_range.json        (GPU box)
"""
import argparse
import json
import os
import sys
import tempfile
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EPS = 1e-5


def rescale(sd, s=1.0, t=1.0, mode="ir"):
    """function-preserving rescaling of an IR / IR-SE state dict (see module docstring)"""
    o = OrderedDict((k, np.array(v, np.float64)) for k, v in sd.items())

    def bn_in(p, f):      # BN whose INPUT got f times larger: same output
        o[p + ".running_mean"] *= f
        o[p + ".running_var"] = f * f * (o[p + ".running_var"] + EPS) - EPS

    def bn_out(p, f):     # BN whose OUTPUT shall be f times larger
        o[p + ".weight"] *= f
        o[p + ".bias"] *= f

    bn_out("input_layer.1", s)
    n_units = sum(1 for k in sd if k.endswith(".res_layer.1.weight"))
    for i in range(n_units):
        p = "body.%d" % i
        bn_in(p + ".res_layer.0", s)
        bn_out(p + ".res_layer.4", s)
        if p + ".shortcut_layer.0.weight" in o:
            bn_in(p + ".shortcut_layer.1", s)
            bn_out(p + ".shortcut_layer.1", s)
        o[p + ".res_layer.1.weight"] *= t
        o[p + ".res_layer.3.weight"] /= t
        # (IR-SE: the SE gate sees mean(BN4 output) = s * mean: fc1 takes 1/s so that the gate is unchanged)
        if p + ".res_layer.5.fc1.weight" in o:
            o[p + ".res_layer.5.fc1.weight"] /= s
    bn_in("output_layer.0", s)
    return OrderedDict((k, v.astype(np.float32)) for k, v in o.items())


def sweep(frt, mode, scales, n_faces=8, which="stream"):
    from oracle import nets
    sy = frt.synth
    base = sy.arcface_state(2, mode, calib=sy.load_calibration(mode))
    faces = sy.make_faces(n_faces)
    x = np.ascontiguousarray(((faces[..., ::-1].astype(np.float32) - 127.5) * 0.0078125).transpose(0, 3, 1, 2))
    ref = nets.arcface_forward(base, x)
    tmp = tempfile.mkdtemp(prefix="frt_range_")
    rows = []
    for sc in scales:
        sd = rescale(base, s=sc if which == "stream" else 1.0, t=sc if which == "branch" else 1.0, mode=mode)
        o32 = nets.arcface_forward(sd, x)                      # the oracle on the rescaled weights (fp32): must equal `ref`
        path = frt.write_weights(os.path.join(tmp, "w.frtw"), sd, 2 if mode == "ir" else 3)
        rec = frt.ArcFaceIR50(path, 640, 640, maxBatchSize=n_faces, maxFacesPerScene=4)
        got = rec.doInference(x)
        rec.close()
        finite = bool(np.isfinite(got).all())
        cos = (got * o32).sum(1) if finite else np.full(n_faces, np.nan)
        rows.append({"scale": sc, "min_cos_vs_fp32_oracle": float(np.nanmin(cos)) if finite else None, "one_minus_min_cos": float(1 - np.nanmin(cos)) if finite else None,
                     "finite": finite, "nonfinite_values": int((~np.isfinite(got)).sum()),
                     "zero_embeddings": int((np.abs(got).sum(1) == 0).sum()),
                     "oracle_self_consistency": float(1 - (o32 * ref).sum(1).min())})
        print(mode, which, json.dumps(rows[-1]), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--faces", type=int, default=8)
    args = ap.parse_args()
    import __graft_entry__ as entry
    frt = entry.load_pkg()
    scales = [1e-4, 1e-3, 1e-2, 1e-1, 1.0, 1e1, 1e2, 1e3, 1e4]
    report = {"what": __doc__.split("\n\n")[0], "tolerance": "north_star: embeddings cosine-equal within 1e-4", "faces": args.faces, "runs": {}}
    for mode in ("ir", "ir_se"):
        for which in ("stream", "branch"):
            report["runs"]["%s/%s" % (mode, which)] = sweep(frt, mode, scales, args.faces, which)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
