"""CPU oracle of the OPTIONAL 5-point alignment mode.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED, and necessarily so: the reference has no landmark decode and no alignment - it exports the detector
without ``LandmarkHead`` (/root/reference/conversion/retina/torch2trt.py:7-9, /root/reference/src/retinaface.cpp:58-60 binds two
outputs) and crops by bbox (/root/reference/src/arcface.cpp:3-17).  What exists in the reference is the head itself
(/root/reference/conversion/retina/models/retinaface.py:37-46,117-128: 1x1 conv 64 -> num_anchors*10, NHWC view (-1, 10)); its raw
output IS pinned, by the ``retinaface_mnet_ldm`` golden generated from that unmodified module.  Everything after the raw
head output restates public algorithms:

  * landmark decode: upstream RetinaFace ``decode_landm``: point_k = prior_centre + pre[2k:2k+2] * variance[0] * prior_size
    (variance[0] = 0.1 as in retinaface.cpp:166-167), then the same un-letterboxing as the boxes (retinaface.cpp:171-183)
    in float32 without the int truncation;
  * similarity fit: least-squares scale*rotation + translation (Umeyama 1991; identical whenever the optimum is not a
    reflection) onto the public ArcFace 112x112 template;
  * warp: inverse-mapped bilinear, zero border, round-half-up to u8.

All arithmetic is float32 in the same operation order as the HIP kernels (kernels_post.hip landmark_decode_kernel,
kernels_image.hip align_faces_kernel), so the GPU tests can use tight tolerances (1e-3 px, 1 LSB).
"""
import numpy as np

ARC_TEMPLATE = np.array([38.2946, 51.6963, 73.5318, 51.5014, 56.0252, 71.7366, 41.5493, 92.3655, 70.7299, 92.2041], np.float32)
STEPS = (8, 16, 32)
MIN_SIZES = ((10, 20), (32, 64), (128, 256))  # retinaface.cpp:214
f32 = np.float32


def anchor_geometry(a, in_h, in_w):
    """anchor index -> (cx, cy, sx, sy) normalised float32, same enumeration as retinaface.cpp:213-239."""
    base = 0
    for lv, st in enumerate(STEPS):
        fh, fw = -(-in_h // st), -(-in_w // st)
        n = fh * fw * 2
        if a < base + n:
            rel = a - base
            l, cell = rel & 1, rel >> 1
            i, j = divmod(cell, fw)
            ms = MIN_SIZES[lv][l]
            return (f32(f32(f32(j) + f32(0.5)) * f32(st)) / f32(in_w), f32(f32(f32(i) + f32(0.5)) * f32(st)) / f32(in_h),
                    f32(ms) / f32(in_w), f32(ms) / f32(in_h))
        base += n
    raise IndexError(a)


def decode_landmarks(ldm_raw, anchors_idx, in_h, in_w, frame_h, frame_w):
    """ldm_raw [A][10], anchors_idx [n] -> [n][5][2] (x=col, y=row) in frame pixels, float32."""
    scale_h, scale_w = f32(in_h) / f32(frame_h), f32(in_w) / f32(frame_w)
    out = np.zeros((len(anchors_idx), 5, 2), np.float32)
    for n, a in enumerate(anchors_idx):
        cx, cy, sx, sy = anchor_geometry(int(a), in_h, in_w)
        for k in range(5):
            p0, p1 = f32(ldm_raw[a, 2 * k]), f32(ldm_raw[a, 2 * k + 1])
            x = f32(cx + f32(f32(p0 * f32(0.1)) * sx)) * f32(in_w)
            y = f32(cy + f32(f32(p1 * f32(0.1)) * sy)) * f32(in_h)
            if scale_h > scale_w:
                x = x / scale_w
                y = f32(y - f32(f32(in_h) - f32(scale_w * f32(frame_h))) / f32(2)) / scale_w
            else:
                x = f32(x - f32(f32(in_w) - f32(scale_h * f32(frame_w))) / f32(2)) / scale_h
                y = y / scale_h
            out[n, k] = (x, y)
    return out


def similarity_inverse(lm):
    """5 landmarks [5][2] -> (ok, ia, ib, tx, ty): src = [ia ib; -ib ia] * (dst - t).  float32, kernel operation order."""
    lm = np.asarray(lm, np.float32).reshape(5, 2)
    t = ARC_TEMPLATE.reshape(5, 2)
    msx = msy = mdx = mdy = f32(0)
    for k in range(5):
        msx = f32(msx + lm[k, 0]); msy = f32(msy + lm[k, 1])
        mdx = f32(mdx + t[k, 0]); mdy = f32(mdy + t[k, 1])
    msx, msy, mdx, mdy = f32(msx * f32(0.2)), f32(msy * f32(0.2)), f32(mdx * f32(0.2)), f32(mdy * f32(0.2))
    sa = sb = den = f32(0)
    for k in range(5):
        sx, sy = f32(lm[k, 0] - msx), f32(lm[k, 1] - msy)
        dx, dy = f32(t[k, 0] - mdx), f32(t[k, 1] - mdy)
        sa = f32(sa + f32(f32(sx * dx) + f32(sy * dy)))
        sb = f32(sb + f32(f32(sx * dy) - f32(sy * dx)))
        den = f32(den + f32(f32(sx * sx) + f32(sy * sy)))
    ok = bool(den > f32(1e-6) and f32(f32(sa * sa) + f32(sb * sb)) > f32(1e-12))
    if not ok:
        return False, f32(0), f32(0), f32(0), f32(0)
    a, b = f32(sa / den), f32(sb / den)
    tx = f32(mdx - f32(f32(a * msx) - f32(b * msy)))
    ty = f32(mdy - f32(f32(b * msx) + f32(a * msy)))
    n2 = f32(f32(a * a) + f32(b * b))
    return True, f32(a / n2), f32(b / n2), tx, ty


def similarity_matrix(lm):
    """Forward 2x3 matrix (float64, for documentation / cross-checks): dst = M @ [src; 1]."""
    lm = np.asarray(lm, np.float64).reshape(5, 2)
    t = ARC_TEMPLATE.astype(np.float64).reshape(5, 2)
    sc, dc = lm - lm.mean(0), t - t.mean(0)
    a = (sc * dc).sum() / (sc ** 2).sum()
    b = (sc[:, 0] * dc[:, 1] - sc[:, 1] * dc[:, 0]).sum() / (sc ** 2).sum()
    A = np.array([[a, -b], [b, a]])
    return np.hstack([A, (t.mean(0) - A @ lm.mean(0))[:, None]])


def align_faces(frame, landmarks):
    """frame u8 [H][W][3] BGR, landmarks [n][5][2] -> (crops u8 [n][112][112][3], valid [n])."""
    frame = np.asarray(frame, np.uint8)
    H, W = frame.shape[:2]
    lms = np.asarray(landmarks, np.float32).reshape(-1, 5, 2)
    crops = np.zeros((len(lms), 112, 112, 3), np.uint8)
    valid = np.zeros(len(lms), np.int32)
    oy, ox = np.meshgrid(np.arange(112, dtype=np.float32), np.arange(112, dtype=np.float32), indexing="ij")
    fpad = np.zeros((H + 2, W + 2, 3), np.float32)  # zero border: taps outside the frame read 0
    fpad[1:-1, 1:-1] = frame
    for n, lm in enumerate(lms):
        ok, ia, ib, tx, ty = similarity_inverse(lm)
        valid[n] = int(ok)
        if not ok:
            continue
        ux, uy = (ox - tx).astype(np.float32), (oy - ty).astype(np.float32)
        sx = ((ia * ux).astype(np.float32) + (ib * uy).astype(np.float32)).astype(np.float32)
        sy = ((-ib * ux).astype(np.float32) + (ia * uy).astype(np.float32)).astype(np.float32)
        fx0, fy0 = np.floor(sx), np.floor(sy)
        wx, wy = (sx - fx0).astype(np.float32)[..., None], (sy - fy0).astype(np.float32)[..., None]
        x0, y0 = fx0.astype(np.int64), fy0.astype(np.int64)

        def tap(dy, dx):
            xx, yy = x0 + dx, y0 + dy
            inside = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(inside[..., None], fpad[np.clip(yy, -1, H) + 1, np.clip(xx, -1, W) + 1], f32(0)).astype(np.float32)

        t00, t01, t10, t11 = tap(0, 0), tap(0, 1), tap(1, 0), tap(1, 1)
        top = (t00 + (wx * (t01 - t00)).astype(np.float32)).astype(np.float32)
        bot = (t10 + (wx * (t11 - t10)).astype(np.float32)).astype(np.float32)
        val = (top + (wy * (bot - top)).astype(np.float32)).astype(np.float32)
        crops[n] = np.clip(np.floor(val + f32(0.5)), 0, 255).astype(np.uint8)
    return crops, valid
