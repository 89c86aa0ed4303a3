"""Device pre-processing / crop kernels vs oracle/imgops.c: byte- and bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geom", [(640, 640, 640, 640), (320, 288, 640, 480), (640, 640, 1920, 1080), (640, 640, 480, 640), (160, 96, 200, 100)])
def test_det_preprocess_bit_exact(frt, orc, synth, blobs, geom):
    in_w, in_h, fw, fh = geom
    path, _ = blobs("det")
    det = frt.RetinaFace(path, fw, fh, (3, in_h, in_w), 1, 4)
    fr = synth.make_frame(7, fh, fw)
    got = det.preprocess(fr)
    want = orc.det_preprocess(fr, in_h, in_w)
    assert got.shape == want.shape and np.array_equal(got, want)
    # a padded row stride must not change anything
    wide = np.zeros((fh, fw + 13, 3), np.uint8)
    wide[:, :fw] = fr
    assert np.array_equal(det.preprocess(wide[:, :fw]), want)
    det.close()


def test_crop_faces_byte_exact_and_errors(frt, orc, synth):
    fr = synth.make_frame(9, 480, 640)
    boxes = np.zeros(6, frt.BBOX_DTYPE)
    boxes[0] = (10, 20, 122, 132, 0.9)     # 112x112 ROI: copy
    boxes[1] = (0, 0, 479, 639, 0.8)       # whole frame minus far corner, strong downscale
    boxes[2] = (100, 200, 110, 215, 0.7)   # 10x15 ROI: strong upscale, border taps clamp to the ROI
    boxes[3] = (5, 600, 300, 639, 0.6)
    boxes[4] = (470, 0, 479, 5, 0.5)
    boxes[5] = (17, 33, 18, 34, 0.4)       # 1x1 ROI
    got = frt.getCroppedFaces(fr, boxes)
    want = orc.crop_faces(fr, boxes)
    assert got.shape == (6, 112, 112, 3) and np.array_equal(got, want)
    bad = boxes.copy()
    bad[2] = (100, 200, 100, 215, 0.7)     # zero-height ROI: OpenCV would throw
    with pytest.raises(frt.FrtError) as e:
        frt.getCroppedFaces(fr, bad)
    assert e.value.code == frt.FRT_ERR_EMPTY_ROI


def test_face_normalise_bit_exact(frt, orc, synth, blobs):
    path, _ = blobs("ir")
    rec = frt.ArcFaceIR50(path)
    crop = synth.make_faces(1)[0]
    assert np.array_equal(rec.preprocessFace(crop), orc.face_normalize(crop[None])[0])
    rec.close()


@pytest.mark.parametrize("src,dst", [((480, 640), (640, 640)), ((1080, 1920), (480, 640)), ((1280, 1280), (640, 640)), ((97, 131), (640, 480)),
                                     ((640, 480), (640, 480)), ((333, 500), (112, 112))])
def test_frame_resize_matches_opencv_restatement(frt, orc, synth, src, dst):
    """Frame ingest (app.cpp:301): device resize == the CPU restatement of cv::resize INTER_LINEAR, bit for bit (also the exact-2x
    case OpenCV routes through its area path, which has the same value)."""
    img = synth.make_frames(1, src[0], src[1])[0]
    out = frt.resizeFrame(img, dst[1], dst[0])
    assert out.shape == (dst[0], dst[1], 3)
    assert np.array_equal(out, orc.resize_linear(img, dst[0], dst[1]))


def test_frame_resize_device_batch(frt, orc, synth):
    import torch
    fr = synth.make_frames(3, 360, 480)
    src = torch.from_numpy(fr).cuda()
    dst = torch.empty((3, 640, 640, 3), dtype=torch.uint8, device="cuda")
    frt._check(frt.lib.frt_resize_frames_dev(src.data_ptr(), 3, 360, 480, 480 * 3, 360 * 480 * 3, dst.data_ptr(), 640, 640,
                                             torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], orc.resize_linear(fr[i], 640, 640))
