"""Generate the committed golden vectors by running the REFERENCE's own PyTorch modules (build container only).

    python tests/golden/make_golden.py --calib     # once: BatchNorm1d statistics for the synthetic recogniser
    python tests/golden/make_golden.py             # goldens -> tests/golden/*.npz

Reads /root/reference in place (never copied).  The recogniser module (conversion/arcface/model_irse.py) imports with
stock PyTorch.  The detector modules (conversion/retina/models/{net,retinaface_trim,retinaface}.py) ``import
torchvision`` which is not installed here: an in-memory stand-in supplies ``IntermediateLayerGetter`` (run the children
in order, collect the return layers) - every convolution / BN / FPN / SSH / head / softmax still executes from the
reference's source.  DESIGN.md states this limitation next to the parity claims.

Fixtures hold inputs by SEED (frames/weights are regenerated from face-recognition-cpp-tensorrt_amd/synth.py) and
expected OUTPUTS (sub-sampled where large); nothing from /root/reference is stored.
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/conversion"


def load_pkg():
    name = "frt_synth_only"
    d = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = load_pkg()


def torchvision_standin():
    class IntermediateLayerGetter(nn.ModuleDict):
        def __init__(self, model, return_layers):
            layers, rl = {}, dict(return_layers)
            for name, module in model.named_children():
                layers[name] = module
                rl.pop(name, None)
                if not rl:
                    break
            super().__init__(layers)
            self.return_layers = dict(return_layers)

        def forward(self, x):
            from collections import OrderedDict
            out = OrderedDict()
            for name, module in self.items():
                x = module(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    names = ["torchvision", "torchvision.models", "torchvision.models._utils", "torchvision.models.detection",
             "torchvision.models.detection.backbone_utils"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["torchvision.models._utils"].IntermediateLayerGetter = IntermediateLayerGetter
    mods["torchvision"].models = mods["torchvision.models"]
    mods["torchvision.models"]._utils = mods["torchvision.models._utils"]
    mods["torchvision.models"].detection = mods["torchvision.models.detection"]
    mods["torchvision.models.detection"].backbone_utils = mods["torchvision.models.detection.backbone_utils"]
    sys.modules.update(mods)


def load_sd(module, sd):
    ref = {k: v for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert set(ref) == set(sd), sorted(set(ref) ^ set(sd))[:8]
    for k in ref:
        assert tuple(ref[k].shape) == sd[k].shape, k
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return module.eval()


def face_input(faces):
    x = (faces[..., ::-1].astype(np.float32) - 127.5) * 0.0078125  # arcface.cpp:105-114
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def arcface_module(mode):
    sys.path.insert(0, os.path.join(REF, "arcface"))
    import model_irse
    return (model_irse.IR_50 if mode == "ir" else model_irse.IR_SE_50)([112, 112])


def calibrate():
    torch.manual_seed(0)
    for mode in ("ir", "ir_se"):
        m = load_sd(arcface_module(mode), synth.arcface_state(2, mode))
        feats = []
        h = m.output_layer[3].register_forward_hook(lambda mod, i, o: feats.append(o.detach().numpy().copy()))
        with torch.no_grad():
            for s in range(0, 96, 16):
                m(torch.from_numpy(face_input(synth.make_faces(96)[s:s + 16])))
        h.remove()
        f = np.concatenate(feats).astype(np.float64)
        out = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd", "synth_calib_%s.npz" % mode)
        np.savez(out, mean=f.mean(0).astype(np.float32), var=f.var(0).astype(np.float32))
        print("wrote", out, "pre-BN1d mean |mu| %.3f, mean var %.3f" % (np.abs(f.mean(0)).mean(), f.var(0).mean()))


def golden_arcface():
    for mode, nf in (("ir", 8), ("ir_se", 4)):
        sd = synth.arcface_state(2, mode, calib=synth.load_calibration(mode))
        m = load_sd(arcface_module(mode), sd)
        x = face_input(synth.make_faces(nf))
        blocks = []
        hooks = [m.input_layer.register_forward_hook(lambda mod, i, o: blocks.append(o.detach().numpy().copy()))]
        for u in m.body:
            hooks.append(u.register_forward_hook(lambda mod, i, o: blocks.append(o.detach().numpy().copy())))
        with torch.no_grad():
            emb = m(torch.from_numpy(x)).numpy()
        for h in hooks:
            h.remove()
        # per-block pins: a fixed pseudo-random sample of 64 positions of face 0 + block statistics
        r = np.random.Generator(np.random.PCG64(7))
        samp_idx, samp_val, stats = [], [], []
        for b in blocks:
            flat = b[0].reshape(-1)
            idx = r.integers(0, flat.size, 64)
            samp_idx.append(idx)
            samp_val.append(flat[idx])
            stats.append([float(b.mean()), float(b.std()), float(np.abs(b).max())])
        np.savez(os.path.join(HERE, "arcface_%s.npz" % mode), seed=2, n_faces=nf, embeddings=emb, block_idx=np.array(samp_idx),
                 block_val=np.array(samp_val, np.float32), block_stats=np.array(stats, np.float32),
                 cos=(emb @ emb.T).astype(np.float32))
        print("arcface", mode, "emb", emb.shape, "off-diagonal cos: mean %.3f max %.3f" %
              ((emb @ emb.T)[~np.eye(nf, dtype=bool)].mean(), (emb @ emb.T)[~np.eye(nf, dtype=bool)].max()))


def golden_retinaface():
    torchvision_standin()
    sys.path.insert(0, os.path.join(REF, "retina"))
    from config import cfg_mnet
    from models.retinaface_trim import RetinaFace
    sd = synth.retinaface_state(1)
    m = load_sd(RetinaFace(cfg_mnet, "test"), sd)
    out = {}
    for tag, (h, w) in (("640", (640, 640)), ("288x320", (288, 320)), ("96x160", (96, 160))):
        fr = synth.make_frames(2, h, w)
        x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
        with torch.no_grad():
            loc, conf = m(torch.from_numpy(x))
        loc, conf = loc.numpy(), conf.numpy()
        step = 7 if loc.shape[1] > 4000 else 1
        out["loc_" + tag] = loc[:, ::step]
        out["conf_" + tag] = conf[:, ::step]
        out["sum_" + tag] = np.array([loc.astype(np.float64).sum(), np.abs(loc).astype(np.float64).sum(), conf[..., 1].astype(np.float64).sum()])
        out["step_" + tag] = step
        out["npass_" + tag] = (conf[..., 1] > 0.6).sum(1)
        print("retinaface", tag, loc.shape, "anchors > 0.6:", out["npass_" + tag])
    np.savez(os.path.join(HERE, "retinaface_mnet.npz"), seed=1, **out)


def golden_retinaface_landmarks():
    """Full model WITH LandmarkHead (conversion/retina/models/retinaface.py) - pins the raw landmark output of the optional
    alignment mode.  Same torchvision stand-in as above."""
    torchvision_standin()
    sys.path.insert(0, os.path.join(REF, "retina"))
    from config import cfg_mnet
    from models.retinaface import RetinaFace
    cfg = dict(cfg_mnet)
    cfg["pretrain"] = False
    sd = synth.retinaface_state(1, landmarks=True)
    m = load_sd(RetinaFace(cfg, "test"), sd)
    out = {}
    for tag, (h, w) in (("288x320", (288, 320)), ("96x160", (96, 160))):
        fr = synth.make_frames(2, h, w)
        x = np.ascontiguousarray((fr.astype(np.float32) - np.array([104, 117, 123], np.float32)).transpose(0, 3, 1, 2))
        with torch.no_grad():
            loc, conf, ldm = m(torch.from_numpy(x))
        out["loc_" + tag], out["conf_" + tag], out["ldm_" + tag] = loc.numpy()[:, ::3], conf.numpy()[:, ::3], ldm.numpy()[:, ::3]
        print("retinaface+landmarks", tag, ldm.shape, "|ldm| mean %.3f" % np.abs(ldm.numpy()).mean())
    np.savez(os.path.join(HERE, "retinaface_mnet_ldm.npz"), seed=1, **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--calib", action="store_true")
    a = ap.parse_args()
    if a.calib:
        calibrate()
    else:
        golden_arcface()
        golden_retinaface()
        golden_retinaface_landmarks()
