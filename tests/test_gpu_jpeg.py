"""JPEG frame ingest / reply step, device half (kernels_jpeg.hip) against the libjpeg fixtures, the NumPy restatement and PIL, and the
ingest chain JPEG bytes -> decode -> resize -> pipeline (src/app.cpp:296-310) against the same chain fed with raw frames."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLDEN, "jpeg_vectors.npz"))


@pytest.fixture(scope="module")
def codec(frt):
    c = frt.JpegCodec(max_images=8, max_width=1920, max_height=1088)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["444_q95", "422_q80", "420_q95", "420_odd_q60", "420_rst", "420_tiny", "420_q100", "gray_q85",
                                  "prog_444_q85", "prog_420_odd_q60", "prog_422_q80", "prog_420_rst", "prog_gray_q85"])
def test_device_decode_equals_libjpeg(codec, vec, name):
    got = codec.decode(vec["dec_%s_jpg" % name].tobytes())
    want = vec["dec_%s_bgr" % name]
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("i", range(4))
def test_device_encode_equals_libjpeg(codec, vec, i):
    img, q = vec["enc_%d_bgr" % i], int(vec["enc_%d_q" % i])
    assert codec.encode(img, quality=q)[0] == vec["enc_%d_jpg" % i].tobytes()


def test_unsupported_stream_is_reported(frt, codec, vec):
    with pytest.raises(frt.FrtError) as e:
        codec.decode(vec["unsupported_arithmetic_jpg"].tobytes())
    assert e.value.code == frt.FRT_ERR_FORMAT


def test_frame_sized_batch_against_pil_and_the_oracle(frt, orc, codec, synth):
    """A batch of 640x640 and 1280x720 camera-style JPEGs: decode == PIL == NumPy restatement, the 1280x720 ones additionally resized
    to the 640x640 frame like cv::resize(rawInput, frame, Size(640, 640)) (src/app.cpp:301) == oracle.resize_linear."""
    import torch
    Image = pytest.importorskip("PIL.Image")
    from oracle import jpegops
    rng = np.random.default_rng(4)
    blobs, want = [], []
    for i, (h, w, sub, q) in enumerate([(640, 640, 2, 90), (720, 1280, 2, 85), (640, 640, 0, 95), (720, 1280, 1, 70), (641, 639, 2, 80)]):
        img = np.clip(synth.make_frame(i, h, w).astype(int) + rng.integers(-12, 12, (h, w, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img[..., ::-1]).save(b, "JPEG", quality=q, subsampling=sub)
        blobs.append(b.getvalue())
        dec = np.ascontiguousarray(np.array(Image.open(io.BytesIO(b.getvalue())))[..., ::-1])
        if i < 2:
            assert np.array_equal(jpegops.decode_from_coefficients(*frt.jpeg_read_coefficients(b.getvalue())), dec)
        assert np.array_equal(codec.decode(b.getvalue()), dec), i
        want.append(dec if (h, w) == (640, 640) else orc.resize_linear(dec, 640, 640))
    d = torch.zeros(len(blobs), 640, 640, 3, dtype=torch.uint8, device="cuda")
    for rep in range(3):  # both staging sets
        d.zero_()
        codec.decode_batch_dev(blobs, d.data_ptr(), 640, 640)
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        for i in range(len(blobs)):
            assert np.array_equal(got[i], want[i]), (rep, i)
    # a batch in which every image already has the frame size takes the direct path (no intermediate buffer)
    same = [blobs[0], blobs[2], blobs[0]]
    d3 = torch.zeros(3, 640, 640, 3, dtype=torch.uint8, device="cuda")
    codec.decode_batch_dev(same, d3.data_ptr(), 640, 640)
    torch.cuda.synchronize()
    g3 = d3.cpu().numpy()
    assert np.array_equal(g3[0], want[0]) and np.array_equal(g3[1], want[2]) and np.array_equal(g3[2], want[0])


def test_jpeg_ingest_feeds_the_pipeline_and_the_reply_comes_back_as_jpeg(frt, codec, synth, blobs):
    """/inference end to end (src/app.cpp:293-340): JPEG bytes -> decode (+resize) on a producer stream -> pipeline ordered behind the
    producer's event -> best crop -> JPEG -> base64.  Equals the same calls fed with the decoded frames from host memory."""
    import torch
    Image = pytest.importorskip("PIL.Image")
    dpath, _ = blobs("det")
    rpath, _ = blobs("ir")
    B, K, H, W = 4, 4, 640, 640
    det = frt.RetinaFace(dpath, W, H, (3, H, W), B, K, 0.4, 0.6)
    rec = frt.ArcFaceIR50(rpath, W, H, maxBatchSize=B * K, maxFacesPerScene=K)
    rec.setGallery(synth.make_gallery(5000))
    rec.initMatMul()
    pipe = frt.Pipeline(det, rec, B)
    jpgs, frames = [], []
    for i in range(B):
        b = io.BytesIO()
        Image.fromarray(synth.make_frame(20 + i, H, W)[..., ::-1]).save(b, "JPEG", quality=92, subsampling=2)
        jpgs.append(b.getvalue())
        frames.append(np.ascontiguousarray(np.array(Image.open(io.BytesIO(b.getvalue())))[..., ::-1]))
    frames = np.stack(frames)
    want, want_emb = pipe.run(frames)
    main, prod = torch.cuda.Stream(), torch.cuda.Stream()
    pipe.set_stream(main.cuda_stream)
    d_frames = torch.zeros(B, H, W, 3, dtype=torch.uint8, device="cuda")
    d_res = torch.zeros(B * K * frt.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    codec.decode_batch_dev(jpgs, d_frames.data_ptr(), H, W, hip_stream=prod.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(prod)
    pipe.run_dev(d_frames.data_ptr(), B, d_res.data_ptr(), None, ready_event=ev.cuda_event)
    torch.cuda.synchronize()
    got = np.frombuffer(d_res.cpu().numpy().tobytes(), frt.RESULT_DTYPE)
    assert np.array_equal(got, want) and got["valid"].sum() > 0
    # reply step for frame 0: crop of the best-matching face -> JPEG -> base64 (src/app.cpp:313-331)
    boxes = det.findFace(frames[0])
    rec.forward(frames[0], boxes)
    names, sims = rec.matchTop1()
    best = int(np.argmax(sims))
    crop = rec.croppedFaces[best]["face"]
    jpg = codec.encode(crop)[0]
    b = io.BytesIO()
    Image.fromarray(crop[..., ::-1]).save(b, "JPEG", quality=95, subsampling=2)
    assert jpg == b.getvalue()
    import base64
    assert frt.base64_encode(jpg) == base64.b64encode(jpg).decode()
    pipe.set_stream(None)
    pipe.close()
    det.close()
    rec.close()


def test_encode_of_an_asynchronously_produced_device_input_waits_for_its_event(frt, codec, vec):
    """frt_jpeg_encode_batch_after (round-2 advisor finding): the crops are written on a producer stream behind a long-running kernel; the
    codec's stream waits for the producer's event on the device, so the stream comes out identical to the host-input encode."""
    import torch
    img = vec["enc_0_bgr"]
    want = codec.encode(img, quality=95)[0]
    prod = torch.cuda.Stream()
    d = torch.zeros(img.shape, dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(img).pin_memory()
    with torch.cuda.stream(prod):
        torch.cuda._sleep(200_000_000)            # ~0.1 s of device time in front of the copy
        d.copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(prod)
    got = codec.encode(None, quality=95, device_ptr=d.data_ptr(), shape=(1, img.shape[0], img.shape[1]), ready_event=ev.cuda_event)[0]
    assert got == want == vec["enc_0_jpg"].tobytes()
