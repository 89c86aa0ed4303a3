import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
frt = entry.load_pkg(); s = frt.synth
tmp = tempfile.mkdtemp()
dp = frt.write_weights(os.path.join(tmp, "d.frtw"), s.retinaface_state(1), 1)
rp = frt.write_weights(os.path.join(tmp, "r.frtw"), s.arcface_state(2, "ir", calib=s.load_calibration("ir")), 2)
B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4
_pre = [torch.cuda.Stream() for _ in range(int(os.environ.get("N_PRE", "0")))]
det = frt.RetinaFace(dp, 640, 640, (3, 640, 640), B, K, 0.4, 0.6)
rec = frt.ArcFaceIR50(rp, 640, 640, (3, 112, 112), 512, B * K, K, 0.65)
rec.setGallery(s.make_gallery(int(sys.argv[2]) if len(sys.argv) > 2 else 100000)); rec.initMatMul()
pipe = frt.Pipeline(det, rec, B)
fr = torch.from_numpy(s.make_frames(B, 640, 640)).cuda()
res = torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
_mid = [torch.cuda.Stream() for _ in range(int(os.environ.get("N_MID", "0")))]
st = torch.cuda.Stream(); torch.cuda.set_stream(st); pipe.set_stream(st.cuda_stream)
for _ in range(5): pipe.run_dev(fr.data_ptr(), B, res.data_ptr(), None)
torch.cuda.synchronize()
for graph in (0,):
    pipe.set_graph(graph)
    for _ in range(6): pipe.run_dev(fr.data_ptr(), B, res.data_ptr(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(50):
        a = time.perf_counter(); pipe.run_dev(fr.data_ptr(), B, res.data_ptr(), None); per.append(time.perf_counter() - a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("graph", graph, "host enqueue ms/step %.3f (median %.3f)  total ms/step %.3f" % ((t1 - t0) * 20, np.median(per) * 1e3, (t2 - t0) * 20))
