"""FRTW weight blob: the build's replacement for the reference's serialized TensorRT ``.engine`` file.

The reference loads an opaque TensorRT engine from ``engineFile`` (``/root/reference/src/retinaface.cpp:31-55``,
``src/arcface.cpp:45-69``) that was produced offline by ``conversion/*/torch2trt.py``.  This build keeps the
"one file per network, path passed to the constructor" contract, but the file is a flat little-endian dump of the
PyTorch ``state_dict`` (names after ``module.`` prefix stripping, as ``conversion/retina/torch2trt.py:41-45`` does);
all folding (BatchNorm), layout permutes and fp16 casts happen inside ``libfrt.so`` at load time.

Layout::

    char[8]  magic  = b"FRTW0001"
    u32      kind   (1 = retinaface mobilenet0.25 trimmed, 2 = arcface IR-50, 3 = arcface IR-SE-50)
    u32      n_tensors
    n_tensors x { u16 name_len; char name[name_len]; u8 ndim; u32 dims[ndim]; u64 offset; u64 n_elem }
    ... padding to a 64-byte boundary ...
    float32 data, every tensor 64-byte aligned; ``offset`` is relative to the start of the file.
"""
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"FRTW0001"
KIND_RETINAFACE_MNET025 = 1
KIND_ARCFACE_IR50 = 2
KIND_ARCFACE_IR_SE50 = 3


def _align(n, a=64):
    return (n + a - 1) // a * a


def write_blob(path, state, kind):
    """Write ``state`` (ordered mapping name -> float32 ndarray) as an FRTW blob."""
    items = []
    for name, arr in state.items():
        if name.endswith("num_batches_tracked"):
            continue
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        items.append((name, a))
    header_len = 8 + 4 + 4
    for name, a in items:
        header_len += 2 + len(name.encode()) + 1 + 4 * a.ndim + 8 + 8
    off = _align(header_len)
    offsets = []
    for _, a in items:
        offsets.append(off)
        off = _align(off + a.size * 4)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", kind, len(items)))
        for (name, a), o in zip(items, offsets):
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<B", a.ndim))
            for d in a.shape:
                f.write(struct.pack("<I", d))
            f.write(struct.pack("<QQ", o, a.size))
        pos = f.tell()
        for (name, a), o in zip(items, offsets):
            if o > pos:
                f.write(b"\0" * (o - pos))
            f.write(a.tobytes())
            pos = o + a.size * 4
    return path


def read_blob(path):
    """Read an FRTW blob back -> (kind, OrderedDict name -> ndarray).  Used by tests only."""
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:8] == MAGIC, "not an FRTW blob"
    kind, n = struct.unpack_from("<II", buf, 8)
    p = 16
    out = OrderedDict()
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, p)
        p += 2
        name = buf[p:p + ln].decode()
        p += ln
        ndim = buf[p]
        p += 1
        dims = struct.unpack_from("<" + "I" * ndim, buf, p)
        p += 4 * ndim
        off, ne = struct.unpack_from("<QQ", buf, p)
        p += 16
        out[name] = np.frombuffer(buf, dtype=np.float32, count=ne, offset=off).reshape(dims).copy()
    return kind, out


def export_pth(pth_path, out_path, kind):
    """``.pth`` -> FRTW blob; the replacement for ``conversion/*/torch2trt.py``.

    Strips the ``module.`` prefix and unwraps a ``state_dict`` key exactly like
    ``/root/reference/conversion/retina/torch2trt.py:41-61``.
    """
    import torch

    sd = torch.load(pth_path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    clean = OrderedDict()
    for k, v in sd.items():
        k = k.split("module.", 1)[-1] if k.startswith("module.") else k
        clean[k] = v.detach().cpu().float().numpy()
    return write_blob(out_path, clean, kind)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="Export a PyTorch .pth checkpoint to an FRTW weight blob")
    ap.add_argument("pth")
    ap.add_argument("out")
    ap.add_argument("--kind", choices=["retinaface", "ir50", "ir_se50"], required=True)
    a = ap.parse_args()
    export_pth(a.pth, a.out, {"retinaface": 1, "ir50": 2, "ir_se50": 3}[a.kind])
