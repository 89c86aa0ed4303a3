#!/usr/bin/env python3
"""Builds and runs tests/cpp/multi_device_pipeline.cpp at the benchmark's size: one process, one host thread per device, 32 frames per
step per device, 1M-row gallery replica per device, grouped ncclAllGather of every step's records (weak scaling without a launcher).

    python tools/multi_device_bench.py --devices all --steps 100 --out gpurun_out/r04_multi_device.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="all")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--gallery", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import __graft_entry__ as entry
    frt = entry.load_pkg()
    from test_cpp_shells import build_multi_device
    s = frt.synth
    tmp = tempfile.mkdtemp(prefix="frt_multi_")
    dpath = frt.write_weights(os.path.join(tmp, "det.frtw"), s.retinaface_state(1), 1)
    rpath = frt.write_weights(os.path.join(tmp, "rec.frtw"), s.arcface_state(2, "ir", calib=s.load_calibration("ir")), 2)
    s.make_frames(64, 640, 640).tofile(os.path.join(tmp, "frames.bin"))
    exe = build_multi_device(tmp)
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")  # a communicator lives in the process (DESIGN: streams and hardware queues)
    out = subprocess.run([exe, dpath, rpath, os.path.join(tmp, "frames.bin"), str(args.batch), str(args.gallery), str(args.steps), args.devices],
                         capture_output=True, text=True, timeout=3000, env=env)
    sys.stderr.write(out.stderr[-3000:])
    if out.returncode != 0:
        raise SystemExit("multi_device_pipeline failed (%d): %s" % (out.returncode, out.stdout[-2000:]))
    line = next(l for l in out.stdout.splitlines() if l.startswith("{"))
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
