"""ORACLE (test infrastructure only) - baseline JPEG transforms restated in NumPy integer arithmetic.

The reference decodes request images with ``cv::imdecode`` (``/root/reference/src/app.cpp:296``) and encodes the reply crop with
``cv::imencode(".jpg")`` (``src/app.cpp:328``).  OpenCV 4.5.5 is an un-vendored third-party dependency that hands both to
libjpeg / libjpeg-turbo with the library defaults: ``JDCT_ISLOW`` (Loeffler-Ligtenberg-Moschytz integer DCT, 13-bit constants),
"fancy" triangle-filter chroma upsampling, 16-bit fixed-point colour conversion, quality 95, 4:2:0, Annex-K tables.  This file
restates that published arithmetic (IJG ``jidctint.c`` / ``jfdctint.c`` / ``jdsample.c`` / ``jdcolor.c`` / ``jccolor.c`` /
``jcsample.c`` / ``jcparam.c``; rounding points as documented there) on whole arrays.

PIN: ``tests/test_jpeg_oracle.py`` checks every function here against PIL's bundled libjpeg-turbo (the same library family the
reference's OpenCV wraps) - decoded pixels and encoded byte streams must be identical - and against the committed fixtures in
``tests/golden/jpeg_vectors.npz`` (made with PIL by ``tests/golden/make_jpeg_golden.py``).  The product's device kernels
(``kernels_jpeg.hip``) are then compared with this restatement and with the same fixtures.
"""
import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F = dict(f0_298631336=2446, f0_390180644=3196, f0_541196100=4433, f0_765366865=6270, f0_899976223=7373, f1_175875602=9633,
         f1_501321110=12299, f1_847759065=15137, f1_961570560=16069, f2_053119869=16819, f2_562915447=20995, f3_072711026=25172)

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
                   57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

STD_LUM_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                      18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
STD_CHR_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """v: int64 [..., 8] -> [..., 8]; one pass of the islow inverse transform with a final descale by ``shift``."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["f0_541196100"]
    tmp2 = z1 - z3 * F["f1_847759065"]
    tmp3 = z1 + z2 * F["f0_765366865"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1_175875602"]
    tmp0 = tmp0 * F["f0_298631336"]
    tmp1 = tmp1 * F["f2_053119869"]
    tmp2 = tmp2 * F["f3_072711026"]
    tmp3 = tmp3 * F["f1_501321110"]
    z1 = -z1 * F["f0_899976223"]
    z2 = -z2 * F["f2_562915447"]
    z3 = -z3 * F["f1_961570560"] + z5
    z4 = -z4 * F["f0_390180644"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], -1)
    return _descale(out, shift)


def idct_blocks(coef, q):
    """coef int16 [n, 64] (natural order, not dequantised), q [64] -> u8 samples [n, 8, 8]."""
    x = (coef.astype(np.int64) * q.astype(np.int64)).reshape(-1, 8, 8)
    ws = _idct_1d(x.transpose(0, 2, 1), CONST_BITS - PASS1_BITS).transpose(0, 2, 1)  # pass 1 works on columns
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3) + 128
    return np.clip(out, 0, 255).astype(np.uint8)


def plane_from_blocks(samples, bw, bh):
    return samples.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2v1(p):
    """fancy 2:1 horizontal upsampling of a [h, w] plane (w > 2) -> [h, 2w]."""
    p = p.astype(np.int32)
    out = np.empty((p.shape[0], p.shape[1] * 2), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def upsample_h2v2(p):
    """fancy 2x2 upsampling of a [h, w] plane (w > 2) -> [2h, 2w]."""
    p = p.astype(np.int32)
    h, w = p.shape
    out = np.empty((2 * h, 2 * w), np.int32)
    up = np.concatenate([p[:1], p[:-1]], 0)
    dn = np.concatenate([p[1:], p[-1:]], 0)
    for v, far in ((0, up), (1, dn)):
        col = 3 * p + far                                   # vertical blend, x4 scale
        last = np.concatenate([col[:, :1], col[:, :-1]], 1)
        nxt = np.concatenate([col[:, 1:], col[:, -1:]], 1)
        even = (3 * col + last + 8) >> 4
        odd = (3 * col + nxt + 7) >> 4
        even[:, 0] = (col[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (col[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def ycc_to_bgr(y, cb, cr):
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def decode_from_coefficients(geo, coef):
    """geometry + coefficient blocks (frt.jpeg_read_coefficients) -> u8 BGR [h, w, 3] exactly as libjpeg would decode the stream."""
    W, H = geo["width"], geo["height"]
    planes = []
    for c, q in zip(geo["comps"], geo["q"]):
        n = c["bw"] * c["bh"]
        s = idct_blocks(coef[c["block0"]:c["block0"] + n], q)
        planes.append(plane_from_blocks(s, c["bw"], c["bh"])[:c["dh"], :c["dw"]])
    if geo["ncomp"] == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], -1)
    hs, vs = geo["hmax"] // geo["comps"][1]["h"], geo["vmax"] // geo["comps"][1]["v"]
    ch = []
    for p in planes[1:]:
        if hs == 1 and vs == 1:
            u = p
        elif p.shape[1] <= 2:
            u = np.repeat(np.repeat(p, vs, 0), hs, 1)
        elif vs == 1:
            u = upsample_h2v1(p)
        else:
            u = upsample_h2v2(p)
        ch.append(u[:H, :W])
    return ycc_to_bgr(planes[0][:H, :W], ch[0], ch[1])


# ---------------------------------------------------------------------------------------------------------------- encoder
def quant_tables(quality):
    quality = min(max(int(quality), 1), 100)
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    return [np.clip((t * scale + 50) // 100, 1, 255).astype(np.int64) for t in (STD_LUM_Q, STD_CHR_Q)]


def bgr_to_ycc(img):
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    return y, cb, cr


def _fdct_1d(d, first):
    tmp0, tmp7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
    tmp1, tmp6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
    tmp2, tmp5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
    tmp3, tmp4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    sh = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS
    o = [None] * 8
    if first:
        o[0], o[4] = (tmp10 + tmp11) << PASS1_BITS, (tmp10 - tmp11) << PASS1_BITS
    else:
        o[0], o[4] = _descale(tmp10 + tmp11, PASS1_BITS), _descale(tmp10 - tmp11, PASS1_BITS)
    z1 = (tmp12 + tmp13) * F["f0_541196100"]
    o[2] = _descale(z1 + tmp13 * F["f0_765366865"], sh)
    o[6] = _descale(z1 - tmp12 * F["f1_847759065"], sh)
    z1, z2, z3, z4 = tmp4 + tmp7, tmp5 + tmp6, tmp4 + tmp6, tmp5 + tmp7
    z5 = (z3 + z4) * F["f1_175875602"]
    t4, t5, t6, t7 = tmp4 * F["f0_298631336"], tmp5 * F["f2_053119869"], tmp6 * F["f3_072711026"], tmp7 * F["f1_501321110"]
    z1, z2 = -z1 * F["f0_899976223"], -z2 * F["f2_562915447"]
    z3, z4 = -z3 * F["f1_961570560"] + z5, -z4 * F["f0_390180644"] + z5
    o[7], o[5], o[3], o[1] = _descale(t4 + z1 + z3, sh), _descale(t5 + z2 + z4, sh), _descale(t6 + z2 + z3, sh), _descale(t7 + z1 + z4, sh)
    return np.stack(o, -1)


def _blocks(plane, bw, bh):
    return plane.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3).reshape(-1, 8, 8)


def encode_blocks_420(img, quality=95):
    """u8 BGR [h, w, 3] -> quantised blocks int16 [6*mcux*mcuy, 64] in ZIGZAG order: luma plane blocks, then Cb, then Cr."""
    h, w = img.shape[:2]
    mcux, mcuy = (w + 15) // 16, (h + 15) // 16
    pad = np.pad(img, ((0, mcuy * 16 - h), (0, mcux * 16 - w), (0, 0)), mode="edge")
    y, cb, cr = bgr_to_ycc(pad)
    bias = np.tile(np.array([1, 2], np.int64), mcux * 4)[None, :]
    dh = (h + 1) // 2
    # right edge: the last full-resolution column is repeated BEFORE the box filter (== edge-padded pixels); bottom edge: the last
    # real downsampled row is repeated AFTER it
    down = [np.pad(((c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + bias) >> 2)[:dh], ((0, mcuy * 8 - dh), (0, 0)), mode="edge")
            for c in (cb, cr)]
    ql, qc = quant_tables(quality)
    out = []
    for plane, bw, bh, q in ((y, 2 * mcux, 2 * mcuy, ql), (down[0], mcux, mcuy, qc), (down[1], mcux, mcuy, qc)):
        d = _blocks(plane - 128, bw, bh)
        ws = _fdct_1d(d, True)                                        # rows
        res = _fdct_1d(ws.transpose(0, 2, 1), False).transpose(0, 2, 1)  # columns
        qv = (q << 3).reshape(1, 8, 8)
        mag = (np.abs(res) + (qv >> 1)) // qv
        out.append((np.sign(res) * mag).reshape(-1, 64)[:, ZIGZAG])
    # luma blocks entirely outside the image ("dummy" blocks that only fill the last MCU column / row): AC = 0, DC = the previous
    # block of the same MCU in coding order
    yb = out[0].reshape(2 * mcuy, 2 * mcux, 64)
    wr, hr = (w + 7) // 8, (h + 7) // 8
    for my in range(mcuy):
        for mx in range(mcux):
            prev = 0
            for v in range(2):
                for hh in range(2):
                    r, c = my * 2 + v, mx * 2 + hh
                    if r >= hr or c >= wr:
                        yb[r, c] = 0
                        yb[r, c, 0] = prev
                    prev = yb[r, c, 0]
    return np.concatenate(out).astype(np.int16)
