"""JPEG frame ingest / reply step, CPU half: the host entropy coder of libfrt (Huffman decode, JFIF writer, base64) and the NumPy
restatement of the integer transforms (oracle/jpegops.py) against fixtures written by PIL's libjpeg-turbo - the library family behind
cv::imdecode / cv::imencode in the reference (src/app.cpp:296,328) - and, when PIL is importable, against PIL live.  No GPU needed."""
import base64
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLDEN, "jpeg_vectors.npz"))


DEC = ["444_q95", "422_q80", "420_q95", "420_odd_q60", "420_rst", "420_tiny", "420_q100", "gray_q85",
       "prog_444_q85", "prog_420_odd_q60", "prog_422_q80", "prog_420_rst", "prog_gray_q85"]  # prog_*: progressive scan scripts (SOF2)


@pytest.mark.parametrize("name", DEC)
def test_entropy_decode_plus_numpy_transforms_equal_libjpeg(frt, vec, name):
    from oracle import jpegops
    data = vec["dec_%s_jpg" % name].tobytes()
    want = vec["dec_%s_bgr" % name]
    w, h, c = frt.jpeg_info(data)
    assert (h, w) == want.shape[:2] and c == (1 if "gray" in name else 3)
    geo, coef = frt.jpeg_read_coefficients(data)
    got = jpegops.decode_from_coefficients(geo, coef)
    assert np.array_equal(got, want), np.abs(got.astype(int) - want).max()


@pytest.mark.parametrize("i", range(4))
def test_numpy_forward_transforms_plus_jfif_writer_equal_libjpeg(frt, vec, i):
    from oracle import jpegops
    img, q = vec["enc_%d_bgr" % i], int(vec["enc_%d_q" % i])
    got = frt.jpeg_write_jfif(q, img.shape[1], img.shape[0], jpegops.encode_blocks_420(img, q))
    assert got == vec["enc_%d_jpg" % i].tobytes()  # the whole stream: JFIF header, tables, scan, EOI


def test_live_against_pil_when_available(frt, synth):
    Image = pytest.importorskip("PIL.Image")
    from oracle import jpegops
    rng = np.random.default_rng(0)
    for h, w in [(96, 128), (45, 77), (8, 8), (120, 16)]:
        img = np.clip(synth.make_frame(h + w, h, w).astype(int) + rng.integers(-20, 20, (h, w, 3)), 0, 255).astype(np.uint8)
        for sub in (0, 1, 2):
            b = io.BytesIO()
            Image.fromarray(img[..., ::-1]).save(b, "JPEG", quality=int(rng.integers(40, 100)), subsampling=sub)
            want = np.array(Image.open(io.BytesIO(b.getvalue())))[..., ::-1]
            got = jpegops.decode_from_coefficients(*frt.jpeg_read_coefficients(b.getvalue()))
            assert np.array_equal(got, want), (h, w, sub)
        b = io.BytesIO()
        Image.fromarray(img[..., ::-1]).save(b, "JPEG", quality=95, subsampling=2)
        assert frt.jpeg_write_jfif(95, w, h, jpegops.encode_blocks_420(img, 95)) == b.getvalue(), (h, w)


def test_bad_streams_are_errors_not_crashes(frt, vec):
    good = vec["dec_420_q95_jpg"].tobytes()
    with pytest.raises(frt.FrtError) as e:
        frt.jpeg_info(vec["unsupported_arithmetic_jpg"].tobytes())
    assert e.value.code == frt.FRT_ERR_FORMAT and "arithmetic" in str(e.value)
    for bad in (b"", b"\xff\xd8", good[:40], good[:200], b"\x00" * 64, good[:2] + b"\xff\xc0\x00\x02" + good[2:]):
        with pytest.raises(frt.FrtError):
            frt.jpeg_read_coefficients(bad)
    # a truncated scan must not read out of bounds (zero bits are supplied past the end, like libjpeg's fill-with-EOI behaviour)
    geo, coef = frt.jpeg_read_coefficients(good[:len(good) // 2])
    assert coef.shape[0] == sum(c["bw"] * c["bh"] for c in geo["comps"])
    rng = np.random.default_rng(1)
    for _ in range(50):  # random corruption inside the entropy-coded segment: error or garbage, never a crash
        b = bytearray(good)
        for k in rng.integers(len(good) // 2, len(good) - 2, 4):
            b[k] = int(rng.integers(0, 256))
        try:
            frt.jpeg_read_coefficients(bytes(b))
        except frt.FrtError:
            pass


def test_base64_matches_the_standard_alphabet(frt):
    rng = np.random.default_rng(2)
    for n in (0, 1, 2, 3, 4, 57, 1000, 10368):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert frt.base64_encode(data) == base64.b64encode(data).decode()


def test_progressive_coefficients_equal_the_sequential_coding_of_the_same_image(frt):
    """A progressive stream carries the SAME quantised coefficients as the baseline coding of the image, spread over DC / AC first and
    refinement scans; the progressive entropy decoder (csrc/frt_jpeg.cpp: decode_progressive, ITU T.81 Annex G) must reassemble them
    exactly.  Needs PIL to write the two codings (skipped without it; the committed prog_* vectors above cover the decoder regardless)."""
    import io
    Image = pytest.importorskip("PIL.Image")
    r = np.random.default_rng(3)
    for (h, w), kw in [((40, 40), dict(quality=85)), ((33, 71), dict(quality=60, subsampling=2)), ((120, 200), dict(quality=95, subsampling=0)),
                       ((5, 3), dict(quality=90)), ((48, 80), dict(quality=90, subsampling=2, restart_marker_blocks=3))]:
        img = r.integers(0, 256, (h, w, 3)).astype(np.uint8)

        def enc(**k):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", **k)
            return b.getvalue()
        g0, c0 = frt.jpeg_read_coefficients(enc(**kw))
        g1, c1 = frt.jpeg_read_coefficients(enc(progressive=True, **kw))
        assert np.array_equal(c0, c1) and g0["width"] == g1["width"] and [c["bw"] for c in g0["comps"]] == [c["bw"] for c in g1["comps"]]
    # truncated / damaged progressive streams: an error or partial data, never a crash
    good = enc(progressive=True, quality=90)
    for cut in range(0, len(good), max(1, len(good) // 60)):
        try:
            frt.jpeg_read_coefficients(good[:cut])
        except frt.FrtError:
            pass
    for _ in range(200):
        b = bytearray(good)
        for k in r.integers(2, len(good) - 2, 3):
            b[k] = int(r.integers(0, 256))
        try:
            frt.jpeg_read_coefficients(bytes(b))
        except frt.FrtError:
            pass
