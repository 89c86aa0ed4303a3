"""oracle/imgops.c: OpenCV semantics (restated; OpenCV itself is absent) cross-checked against torch interpolate."""
import numpy as np
import torch
import torch.nn.functional as F


def torch_resize(img, dh, dw, mode):
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(x, size=(dh, dw), mode=mode, align_corners=False)
    return np.clip(np.rint(y[0].permute(1, 2, 0).numpy()), 0, 255).astype(np.int32)


def test_bilinear_and_bicubic_agree_with_float_reference_within_one_lsb(orc, synth):
    img = synth.make_frame(5, 97, 131)
    for (dh, dw) in ((112, 112), (50, 77), (200, 300), (97, 131)):
        lin = orc.resize_linear(img, dh, dw).astype(np.int32)
        cub = orc.resize_cubic(img, dh, dw).astype(np.int32)
        assert np.abs(lin - torch_resize(img, dh, dw, "bilinear")).max() <= 1
        assert np.abs(cub - torch_resize(img, dh, dw, "bicubic")).max() <= 1
    assert np.array_equal(orc.resize_cubic(img, 97, 131), img)  # same size == copy
    assert np.array_equal(orc.resize_linear(img, 97, 131), img)


def test_det_preprocess_letterbox_geometry_and_normalisation(orc, synth):
    fr = synth.make_frame(1, 480, 640)
    out = orc.det_preprocess(fr, 288, 320)  # config.json default: h = int(0.5*480) = 240, y = 24 (retinaface.cpp:112-116)
    assert out.shape == (3, 288, 320)
    mean = np.array([104, 117, 123], np.float32)
    for c in range(3):
        assert np.all(out[c, :24] == 128 - mean[c]) and np.all(out[c, 264:] == 128 - mean[c])
    inner = out[:, 24:264].transpose(1, 2, 0) + mean
    assert np.array_equal(inner.astype(np.uint8), orc.resize_linear(fr, 240, 320))
    # identity geometry: pure mean subtraction, BGR order kept, planar
    fr2 = synth.make_frame(2, 64, 64)
    out2 = orc.det_preprocess(fr2, 64, 64)
    assert np.array_equal(out2, fr2.astype(np.float32).transpose(2, 0, 1) - mean[:, None, None])
    # portrait: else branch, x offset (in_w - int(scale_h*cols))/2
    fr3 = synth.make_frame(3, 640, 480)
    out3 = orc.det_preprocess(fr3, 640, 640)
    assert np.all(out3[0, :, :80] == 24) and np.all(out3[0, :, 560:] == 24)


def test_crop_roi_excludes_far_corner_and_normalise_layout(orc, synth):
    fr = synth.make_frame(4, 300, 400)
    boxes = np.zeros(2, orc.BBOX_DTYPE)
    boxes[0] = (10, 20, 122, 132, 0.9)   # rows 10..121, cols 20..131 -> exactly 112x112: a copy
    boxes[1] = (50, 60, 250, 300, 0.8)
    crops = orc.crop_faces(fr, boxes)
    assert np.array_equal(crops[0], fr[10:122, 20:132])
    want = torch_resize(fr[50:250, 60:300], 112, 112, "bicubic")
    assert np.abs(crops[1].astype(np.int32) - want).max() <= 1
    x = orc.face_normalize(crops)
    assert x.shape == (2, 3, 112, 112)
    assert np.array_equal(x[0, 0], (crops[0][..., 2].astype(np.float32) - 127.5) * 0.0078125)  # channel 0 = R
    boxes[1] = (50, 60, 50, 300, 0.8)
    try:
        orc.crop_faces(fr, boxes)
        assert False, "empty ROI must raise (OpenCV throws)"
    except ValueError:
        pass
