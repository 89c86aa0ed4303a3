#!/usr/bin/env python3
"""Builds and runs tests/cpp/dropin_bench.cpp: the reference's /inference call sequence (src/app.cpp:304-310) through the C++ drop-in
shells at a given gallery size, single-threaded and with T threads (Crow's model, src/app.cpp:367), one set of objects per thread on the
listed devices.  GPU box only.

    python tools/dropin_bench.py --gallery 1000000 --threads 1 8 --iters 200 --out gpurun_out/r03_dropin_bench.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")


def build(outdir):
    exe = os.path.join(outdir, "dropin_bench")
    obj = exe + ".o"
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "cpp", "dropin_bench.cpp"), "-o", obj])
    subprocess.check_call(["g++", "-o", exe, obj, os.path.join(PKG, "libfrt.so"), "-lpthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, dpath, rpath, frames_path, n_frames, H, W, gallery, threads, iters, devices="0", shared=False, coalesce=0, window_us=100):
    cmd = [exe, dpath, rpath, frames_path, str(n_frames), str(H), str(W), str(gallery), str(threads), str(iters), devices]
    if shared or coalesce:
        cmd += ["shared", str(coalesce), str(window_us)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=3000)
    if out.returncode != 0:
        raise RuntimeError("dropin_bench failed (%d): %s %s" % (out.returncode, out.stdout[-2000:], out.stderr[-2000:]))
    return json.loads(next(l for l in out.stdout.splitlines() if l.startswith("{")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gallery", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--devices", default="0")
    ap.add_argument("--out", default=None)
    ap.add_argument("--shared", action="store_true", help="one detector + one recogniser shared by all threads (the reference's shape)")
    ap.add_argument("--coalesce", type=int, default=0, help="recognizer.coalesceWith(detector, N frames) (implies --shared)")
    ap.add_argument("--window-us", type=int, default=100)
    args = ap.parse_args()
    import __graft_entry__ as entry
    frt = entry.load_pkg()
    s = frt.synth
    tmp = tempfile.mkdtemp(prefix="frt_dropin_")
    dpath = frt.write_weights(os.path.join(tmp, "det.frtw"), s.retinaface_state(1), 1)
    rpath = frt.write_weights(os.path.join(tmp, "rec.frtw"), s.arcface_state(2, "ir", calib=s.load_calibration("ir")), 2)
    H = W = 640
    frames = s.make_frames(args.frames, H, W)
    fpath = os.path.join(tmp, "frames.bin")
    frames.tofile(fpath)
    exe = build(tmp)
    report = {"what": "src/app.cpp:304-310 through the drop-in shells, one 640x640 frame per call, K = 4 faces per frame, %d-row fp32 gallery" % args.gallery,
              "runs": []}
    for t in args.threads:
        r = run(exe, dpath, rpath, fpath, args.frames, H, W, args.gallery, t, args.iters, args.devices, args.shared, args.coalesce, args.window_us)
        report["runs"].append(r)
        print(json.dumps(r), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
