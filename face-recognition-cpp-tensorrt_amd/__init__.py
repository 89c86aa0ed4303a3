"""MI355X-native detect -> crop -> embed -> match hot path of nghiapq77/face-recognition-cpp-tensorrt.

The product is ``libfrt.so`` (hand-written HIP for gfx950 behind the C ABI of ``include/frt.h``).  This package is the
Python-side mirror of the reference's class surface - ``RetinaFace`` (``/root/reference/src/retinaface.h:18-23``),
``ArcFaceIR50`` (``src/arcface.h:19-39``), ``MatMul`` (``src/matmul.h:6-21``), ``getCroppedFaces`` (``src/arcface.h:17``)
- used by the tests and by ``bench.py``; the C++ drop-in shells live in ``include/frt/*.h``.  Every call goes through the
C ABI; there is no Python/CPU fallback: if ``libfrt.so`` is missing the import fails.

The directory name is not an importable identifier; load it with::

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("frt_amd", ".../face-recognition-cpp-tensorrt_amd/__init__.py",
                                                  submodule_search_locations=[".../face-recognition-cpp-tensorrt_amd"])
    frt_amd = importlib.util.module_from_spec(spec); sys.modules["frt_amd"] = frt_amd; spec.loader.exec_module(frt_amd)
"""
import ctypes
import os
import sys

import numpy as np

from . import dist, synth, weights_io  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRT_LIB") or os.path.join(_HERE, "libfrt.so")  # FRT_LIB: measurement builds (make TUNING=1), tools only

if not os.path.exists(LIB_PATH):
    raise ImportError("libfrt.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(or `make -C face-recognition-cpp-tensorrt_amd/csrc`).  There is no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

# ----------------------------------------------------------------------------------------------------------------------
# C ABI declarations (include/frt.h)
# ----------------------------------------------------------------------------------------------------------------------
FRT_OK, FRT_ERR_INVALID, FRT_ERR_NOT_FOUND, FRT_ERR_FORMAT, FRT_ERR_DEVICE, FRT_ERR_EMPTY, FRT_ERR_EMPTY_ROI, FRT_ERR_CAPACITY = range(8)

BBOX_DTYPE = np.dtype([("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"), ("score", "<f4")])  # == struct Bbox, common.h:13-16
RESULT_DTYPE = np.dtype([("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"), ("score", "<f4"), ("frame", "<i4"),
                         ("match_idx", "<i4"), ("match_sim", "<f4"), ("valid", "<i4")])
assert BBOX_DTYPE.itemsize == 20 and RESULT_DTYPE.itemsize == 36

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
ABI = {
    "frt_last_error": (ctypes.c_char_p, []),
    "frt_version": (ctypes.c_char_p, []),
    "frt_device_count": (_i, []),
    "frt_set_wait_spin_us": (ctypes.c_long, [ctypes.c_long]),
    "frt_detector_create": (_i, [ctypes.c_char_p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, ctypes.POINTER(_vp)]),
    "frt_detector_destroy": (None, [_vp]),
    "frt_detector_num_anchors": (_i, [_vp]),
    "frt_detector_find_faces": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp]),
    "frt_detector_find_faces_batch": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _vp]),
    "frt_detector_preprocess": (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    "frt_detector_infer": (_i, [_vp, _vp, _i, _vp, _vp]),
    "frt_detector_postprocess": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "frt_crop_faces": (_i, [_vp, _i, _i, _sz, _vp, _i, _i, _i, _vp, _i]),
    "frt_resize_frame": (_i, [_vp, _i, _i, _sz, _vp, _i, _i, _i]),
    "frt_resize_frames_dev": (_i, [_vp, _i, _i, _i, _sz, _sz, _vp, _i, _i, _vp]),
    "frt_embedder_create": (_i, [ctypes.c_char_p, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_vp)]),
    "frt_embedder_destroy": (None, [_vp]),
    "frt_embedder_preprocess_face": (_i, [_vp, _vp, _vp]),
    "frt_embedder_infer": (_i, [_vp, _vp, _i, _vp]),
    "frt_embedder_forward": (_i, [_vp, _vp, _i, _i, _sz, _vp, _i, _vp, _vp]),
    "frt_matcher_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "frt_matcher_destroy": (None, [_vp]),
    "frt_matcher_init": (_i, [_vp, _vp, _i, _i]),
    "frt_matcher_set_row_offset": (_i, [_vp, _i]),
    "frt_probe_sustained_mfma": (_i, [_i, _i, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]),
    "frt_matcher_set_storage": (_i, [_vp, _i]),
    "frt_matcher_set_screening": (_i, [_vp, _i]),
    "frt_matcher_scan_bytes": (ctypes.c_size_t, [_vp]),
    "frt_matcher_generation": (ctypes.c_uint, [_vp]),
    "frt_matcher_gallery_begin": (_i, [_vp, _i, _i]),
    "frt_matcher_gallery_append": (_i, [_vp, _vp, _i]),
    "frt_matcher_gallery_commit": (_i, [_vp]),
    "frt_matcher_num_rows": (_i, [_vp]),
    "frt_matcher_top1_dev": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "frt_matcher_calculate": (_i, [_vp, _vp, _i, _vp]),
    "frt_matcher_top1": (_i, [_vp, _vp, _i, _vp, _vp]),
    "frt_matcher_calculate_top1": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "frt_pinned_alloc": (_i, [_sz, _i, ctypes.POINTER(_vp)]),
    "frt_pinned_free": (None, [_vp]),
    "frt_merge_top1": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "frt_matcher_topk": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "frt_matcher_topk_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "frt_merge_topk": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "frt_merge_topk_dev": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "frt_embeds_to_half_dev": (_i, [_vp, _sz, _vp, _vp]),
    "frt_comm_get_unique_id": (_i, [_vp]),
    "frt_comm_set_bootstrap_timeout": (ctypes.c_double, [ctypes.c_double]),
    "frt_comm_create": (_i, [_vp, _i, _i, _i, ctypes.POINTER(_vp)]),
    "frt_comm_create_all": (_i, [_i, _vp, _vp]),
    "frt_comm_destroy": (None, [_vp]),
    "frt_comm_rank": (_i, [_vp]),
    "frt_comm_world": (_i, [_vp]),
    "frt_comm_stream": (_vp, [_vp]),
    "frt_comm_all_gather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "frt_comm_all_gather_multi": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "frt_comm_sync": (_i, [_vp]),
    "frt_embedder_set_se_fused": (_i, [_vp, _i]),
    "frt_embedder_set_precision": (_i, [_vp, _i]),
    "frt_jpeg_encode_batch_after": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "frt_detector_geometry": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "frt_coalescer_create": (_i, [_vp, _vp, _vp, _i, _i, ctypes.POINTER(_vp)]),
    "frt_coalescer_destroy": (None, [_vp]),
    "frt_coalescer_infer": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _vp]),
    "frt_coalescer_infer_crops": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _vp, _vp]),
    "frt_pipeline_submit_crops": (_i, [_vp, _vp, _i, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_long)]),
    "frt_coalescer_stats": (_i, [_vp, _vp, _vp]),
    "frt_pipeline_create": (_i, [_vp, _vp, _vp, _i, ctypes.POINTER(_vp)]),
    "frt_pipeline_destroy": (None, [_vp]),
    "frt_pipeline_run": (_i, [_vp, _vp, _i, _vp, _vp]),
    "frt_pipeline_run_dev": (_i, [_vp, _vp, _i, _vp, _vp]),
    "frt_pipeline_run_dev_after": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "frt_pipeline_set_input_sync": (_i, [_vp, _i]),
    "frt_pipeline_check_overlap": (_i, [_vp, _vp]),
    "frt_pipeline_submit": (_i, [_vp, _vp, _i, _vp, _vp, ctypes.POINTER(ctypes.c_long)]),
    "frt_pipeline_wait": (_i, [_vp, ctypes.c_long]),
    "frt_pipeline_sync": (_i, [_vp]),
    "frt_pipeline_set_stream": (_i, [_vp, _vp]),
    "frt_pipeline_set_overlap": (_i, [_vp, _i]),
    "frt_pipeline_set_graph": (_i, [_vp, _i]),
    "frt_pipeline_set_pairing": (_i, [_vp, _i]),
    "frt_pipeline_pairing_stats": (_i, [_vp, _vp, _vp]),
    "frt_pipeline_graph_stats": (_i, [_vp, _vp, _vp]),
    "frt_pipeline_merge_stats": (_i, [_vp, _vp, _vp]),
    "frt_detector_has_landmarks": (_i, [_vp]),
    "frt_detector_find_faces_landmarks": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _vp]),
    "frt_detector_infer_landmarks": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "frt_align_faces": (_i, [_vp, _i, _i, _sz, _vp, _i, _vp, _i]),
    "frt_embedder_forward_aligned": (_i, [_vp, _vp, _i, _i, _sz, _vp, _i, _vp, _vp]),
    "frt_pipeline_set_align": (_i, [_vp, _i]),
    "frt_jpeg_info": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "frt_jpeg_decoder_create": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(_vp)]),
    "frt_jpeg_decoder_destroy": (None, [_vp]),
    "frt_jpeg_decode_batch_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "frt_jpeg_decode": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "frt_jpeg_read_coefficients": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "frt_jpeg_encode_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "frt_jpeg_write_jfif": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp]),
    "frt_base64_encode": (_sz, [_vp, _sz, _vp, _sz]),
    "frt_profile_enable": (_i, [_i]),
    "frt_profile_collect": (_i, [_vp, _sz, _vp, _vp, _i]),
}
for _name, (_res, _args) in ABI.items():
    try:
        _fn = getattr(lib, _name)  # AttributeError here == a symbol declared in frt.h is not exported
    except AttributeError:
        # The import is strict: a library that lacks a declared symbol is a broken build.  The one exception is not reachable from the environment:
        # a measurement driver that A/Bs an OLDER build on the same box registers the module name below BEFORE importing this package
        # (bench.py --ab-old-lib); the skipped symbols are listed on stderr.
        if "frt_amd_ab_old_library" in sys.modules:
            sys.stderr.write("[frt_amd] WARNING: %s is missing from %s (A/B against an older build: measurement only)\n" % (_name, LIB_PATH))
            continue
        raise
    _fn.restype = _res
    _fn.argtypes = _args


class FrtError(RuntimeError):
    """Raised on a non-zero frt_status; mirrors the reference's ``std::logic_error`` / ``throw const char*``."""

    def __init__(self, code, msg):
        super().__init__("frt status %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc != FRT_OK:
        raise FrtError(rc, lib.frt_last_error().decode(errors="replace"))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def device_count():
    return lib.frt_device_count()


def set_wait_spin_us(us):
    """How long blocking entry points busy-poll before backing off to sleeping polls (frt_set_wait_spin_us); returns the previous value."""
    return int(lib.frt_set_wait_spin_us(int(us)))


def probe_sustained_mfma(device=0, mix=2, seconds=0.25):
    """TFLOP/s the device sustains on back-to-back fp16 MFMAs with random operands (mix 0), + one LDS read per MFMA (1), + the dominant conv
    kernel's global loads (2): frt_probe_sustained_mfma."""
    out = ctypes.c_double(0.0)
    _check(lib.frt_probe_sustained_mfma(int(device), int(mix), float(seconds), ctypes.byref(out)))
    return float(out.value)


def write_weights(path, state, kind):
    return weights_io.write_blob(path, state, kind)


# ----------------------------------------------------------------------------------------------------------------------
# class MatMul (src/matmul.h)
# ----------------------------------------------------------------------------------------------------------------------
class MatMul:
    def __init__(self, device=0):
        self._h = _vp()
        _check(lib.frt_matcher_create(device, ctypes.byref(self._h)))
        self.m = self.k = 0

    def init(self, knownEmbeds, numRow=None, numCol=None):
        g = np.ascontiguousarray(knownEmbeds, np.float32)
        numRow = g.shape[0] if numRow is None else numRow
        numCol = g.shape[1] if numCol is None else numCol
        _check(lib.frt_matcher_init(self._h, _ptr(g), int(numRow), int(numCol)))
        self.m, self.k = int(numRow), int(numCol)

    def setStorage(self, fp16):
        """Rows of the NEXT init / galleryBegin are stored as fp16 on the device (BASELINE config 5) when ``fp16`` is true."""
        _check(lib.frt_matcher_set_storage(self._h, 1 if fp16 else 0))

    def setScreening(self, on):
        """``False``: every top-1 call takes the exact fp32 scan of the whole gallery (same answers, the path's worst case)."""
        _check(lib.frt_matcher_set_screening(self._h, 1 if on else 0))

    def scanBytes(self):
        """Gallery bytes one top-1 call reads in the current mode (shadow copy when screened, stored rows otherwise)."""
        return int(lib.frt_matcher_scan_bytes(self._h))

    # streaming load == initKnownEmbeds / addEmbedding x n / initMatMul (src/db.cpp:316-346)
    def galleryBegin(self, rowCapacity, numCol=512):
        _check(lib.frt_matcher_gallery_begin(self._h, int(rowCapacity), int(numCol)))
        self._pending_k = int(numCol)

    def galleryAppend(self, rows):
        """``rows``: float32 array [n, numCol] or a bytes-like blob of raw little-endian float32 rows (SQLite's EMBEDDING column)."""
        if isinstance(rows, (bytes, bytearray, memoryview)):
            a = np.frombuffer(rows, dtype="<f4")
        else:
            a = np.ascontiguousarray(rows, np.float32).reshape(-1)
        if a.size % self._pending_k:
            raise ValueError("galleryAppend: %d floats is not a whole number of %d-float rows" % (a.size, self._pending_k))
        _check(lib.frt_matcher_gallery_append(self._h, _ptr(a), a.size // self._pending_k))

    def galleryCommit(self):
        _check(lib.frt_matcher_gallery_commit(self._h))
        self.m, self.k = int(lib.frt_matcher_num_rows(self._h)), self._pending_k

    def top1_dev(self, embeds_ptr, n, idx_ptr, sim_ptr, hip_stream=None):
        """Asynchronous, raw device addresses (queries fp32 [n][k], idx int32 [n], sim fp32 [n])."""
        _check(lib.frt_matcher_top1_dev(self._h, _vp(embeds_ptr), int(n), _vp(idx_ptr), _vp(sim_ptr), _vp(hip_stream) if hip_stream else None))

    def topk(self, embeds, k):
        """Exact top-k lists [n, k] (entry 0 == top1; ties: lower index first; -1 / -inf when the gallery has fewer than k rows)."""
        e = np.ascontiguousarray(embeds, np.float32).reshape(-1, self.k)
        idx = np.empty((e.shape[0], int(k)), np.int32)
        sim = np.empty((e.shape[0], int(k)), np.float32)
        _check(lib.frt_matcher_topk(self._h, _ptr(e), e.shape[0], int(k), _ptr(idx), _ptr(sim)))
        return idx, sim

    def topk_dev(self, embeds_ptr, n, k, idx_ptr, sim_ptr, hip_stream=None, fp16=False):
        """Asynchronous, raw device addresses: queries fp32 (or IEEE fp16 with ``fp16=True``) [n][k_dim], idx int32 [n][k], sim fp32 [n][k]."""
        _check(lib.frt_matcher_topk_dev(self._h, _vp(embeds_ptr), 1 if fp16 else 0, int(n), int(k), _vp(idx_ptr), _vp(sim_ptr),
                                        _vp(hip_stream) if hip_stream else None))

    def setRowOffset(self, row_offset):
        """Sharded gallery: local row 0 is global row ``row_offset`` (top-1 indices become global)."""
        _check(lib.frt_matcher_set_row_offset(self._h, int(row_offset)))

    def calculate(self, embeds, embedCount=None):
        e = np.ascontiguousarray(embeds, np.float32).reshape(-1, self.k)
        n = e.shape[0] if embedCount is None else embedCount
        out = np.empty((n, self.m), np.float32)
        _check(lib.frt_matcher_calculate(self._h, _ptr(e), int(n), _ptr(out)))
        return out

    def calculate_top1(self, embeds, materialize=True):
        """calculate + row-wise first maximum in one call -> (matrix [n, m] or None, idx [n], sim [n])."""
        e = np.ascontiguousarray(embeds, np.float32).reshape(-1, self.k)
        out = np.empty((e.shape[0], self.m), np.float32) if materialize else None
        idx = np.empty(e.shape[0], np.int32)
        sim = np.empty(e.shape[0], np.float32)
        _check(lib.frt_matcher_calculate_top1(self._h, _ptr(e), e.shape[0], _ptr(out), _ptr(idx), _ptr(sim)))
        return out, idx, sim

    def top1(self, embeds):
        e = np.ascontiguousarray(embeds, np.float32).reshape(-1, self.k)
        idx = np.empty(e.shape[0], np.int32)
        sim = np.empty(e.shape[0], np.float32)
        _check(lib.frt_matcher_top1(self._h, _ptr(e), e.shape[0], _ptr(idx), _ptr(sim)))
        return idx, sim

    def close(self):
        if self._h:
            lib.frt_matcher_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_top1(idx_a, sim_a, idx_b, sim_b):
    n = len(idx_a)
    ia, ib = np.ascontiguousarray(idx_a, np.int32), np.ascontiguousarray(idx_b, np.int32)
    sa, sb = np.ascontiguousarray(sim_a, np.float32), np.ascontiguousarray(sim_b, np.float32)
    io, so = np.empty(n, np.int32), np.empty(n, np.float32)
    _check(lib.frt_merge_top1(n, _ptr(ia), _ptr(sa), _ptr(ib), _ptr(sb), _ptr(io), _ptr(so)))
    return io, so


def merge_topk(idx_all, sim_all):
    """Host k-way merge of per-shard top-k lists [shards, n, k] with global indices -> ([n, k], [n, k])."""
    ia = np.ascontiguousarray(idx_all, np.int32)
    sa = np.ascontiguousarray(sim_all, np.float32)
    shards, n, k = ia.shape
    io, so = np.empty((n, k), np.int32), np.empty((n, k), np.float32)
    _check(lib.frt_merge_topk(shards, n, k, _ptr(ia), _ptr(sa), _ptr(io), _ptr(so)))
    return io, so


def merge_topk_dev(shards, n, k, idx_all_ptr, sim_all_ptr, idx_out_ptr, sim_out_ptr, hip_stream=None):
    _check(lib.frt_merge_topk_dev(int(shards), int(n), int(k), _vp(idx_all_ptr), _vp(sim_all_ptr), _vp(idx_out_ptr), _vp(sim_out_ptr),
                                  _vp(hip_stream) if hip_stream else None))


def embeds_to_half_dev(src_ptr, n_values, dst_ptr, hip_stream=None):
    _check(lib.frt_embeds_to_half_dev(_vp(src_ptr), int(n_values), _vp(dst_ptr), _vp(hip_stream) if hip_stream else None))


# ----------------------------------------------------------------------------------------------------------------------
# RCCL communicator behind the C ABI (frt_comm_*): the exchange step of the multi-GPU path from C++, no torch.distributed
# ----------------------------------------------------------------------------------------------------------------------
COMM_ID_BYTES = 128


def comm_unique_id():
    b = np.zeros(COMM_ID_BYTES, np.uint8)
    _check(lib.frt_comm_get_unique_id(_ptr(b)))
    return b.tobytes()


class Comm:
    def __init__(self, unique_id, rank, world, device=0):
        self._h = _vp()
        b = np.frombuffer(bytes(unique_id), np.uint8)
        assert b.size == COMM_ID_BYTES
        _check(lib.frt_comm_create(_ptr(b), int(rank), int(world), int(device), ctypes.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    @property
    def stream(self):
        """raw hipStream_t of the communicator's exchange stream"""
        return int(lib.frt_comm_stream(self._h) or 0)

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, hip_stream=None):
        _check(lib.frt_comm_all_gather(self._h, _vp(send_ptr), _vp(recv_ptr), int(bytes_per_rank), _vp(hip_stream) if hip_stream else None))

    def sync(self):
        _check(lib.frt_comm_sync(self._h))

    def close(self):
        if self._h:
            lib.frt_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# class RetinaFace (src/retinaface.h)
# ----------------------------------------------------------------------------------------------------------------------
class RetinaFace:
    def __init__(self, engineFile, frameWidth, frameHeight, inputShape=(3, 640, 640), maxBatchSize=1, maxFacesPerScene=4,
                 nms_threshold=0.4, bbox_threshold=0.6, inputName="input_det", outputNames=("output_det0", "output_det1"), device=0):
        assert len(inputShape) == 3 and len(outputNames) == 2  # retinaface.cpp:8,86 (binding names are accepted and ignored)
        self._h = _vp()
        self.frameWidth, self.frameHeight = int(frameWidth), int(frameHeight)
        self.inputShape = tuple(int(x) for x in inputShape)
        self.maxBatchSize, self.maxFacesPerScene = int(maxBatchSize), int(maxFacesPerScene)
        _check(lib.frt_detector_create(os.fsencode(engineFile), self.frameWidth, self.frameHeight, *self.inputShape, self.maxBatchSize,
                                       self.maxFacesPerScene, nms_threshold, bbox_threshold, device, ctypes.byref(self._h)))
        self.numAnchors = lib.frt_detector_num_anchors(self._h)
        self.hasLandmarks = bool(lib.frt_detector_has_landmarks(self._h))

    # ---- optional alignment mode (no counterpart in the reference, see include/frt.h)
    def findFaceLandmarks(self, img):
        """-> (boxes, landmarks [n][5][2] as (x=col, y=row) frame pixels)."""
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros(self.maxFacesPerScene, BBOX_DTYPE)
        ldm = np.zeros((self.maxFacesPerScene, 5, 2), np.float32)
        n = ctypes.c_int(0)
        _check(lib.frt_detector_find_faces_landmarks(self._h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(out), _ptr(ldm),
                                                     ctypes.byref(n)))
        return out[:n.value].copy(), ldm[:n.value].copy()

    def doInferenceLandmarks(self, chw):
        x = np.ascontiguousarray(chw, np.float32).reshape((-1,) + self.inputShape)
        b = x.shape[0]
        loc = np.empty((b, self.numAnchors, 4), np.float32)
        conf = np.empty((b, self.numAnchors, 2), np.float32)
        ldm = np.empty((b, self.numAnchors, 10), np.float32)
        _check(lib.frt_detector_infer_landmarks(self._h, _ptr(x), b, _ptr(loc), _ptr(conf), _ptr(ldm)))
        return loc, conf, ldm

    def findFace(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros(self.maxFacesPerScene, BBOX_DTYPE)
        n = ctypes.c_int(0)
        _check(lib.frt_detector_find_faces(self._h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(out), ctypes.byref(n)))
        return out[:n.value].copy()

    def findFaceBatch(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        nf = frames.shape[0]
        out = np.zeros((nf, self.maxFacesPerScene), BBOX_DTYPE)
        n = np.zeros(nf, np.int32)
        _check(lib.frt_detector_find_faces_batch(self._h, _ptr(frames), nf, frames.shape[1], frames.shape[2], frames.strides[1],
                                                 frames.strides[0], _ptr(out), _ptr(n)))
        return [out[i, :n[i]].copy() for i in range(nf)]

    def preprocess(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty(self.inputShape, np.float32)
        _check(lib.frt_detector_preprocess(self._h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(out)))
        return out

    def doInference(self, chw):
        x = np.ascontiguousarray(chw, np.float32).reshape((-1,) + self.inputShape)
        b = x.shape[0]
        loc = np.empty((b, self.numAnchors, 4), np.float32)
        conf = np.empty((b, self.numAnchors, 2), np.float32)
        _check(lib.frt_detector_infer(self._h, _ptr(x), b, _ptr(loc), _ptr(conf)))
        return loc, conf

    def postprocessing(self, loc, conf):
        loc = np.ascontiguousarray(loc, np.float32).reshape(self.numAnchors, 4)
        conf = np.ascontiguousarray(conf, np.float32).reshape(self.numAnchors, 2)
        out = np.zeros(self.maxFacesPerScene, BBOX_DTYPE)
        n = ctypes.c_int(0)
        _check(lib.frt_detector_postprocess(self._h, _ptr(loc), _ptr(conf), _ptr(out), ctypes.byref(n)))
        return out[:n.value].copy()

    def close(self):
        if self._h:
            lib.frt_detector_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resizeFrame(img, width, height, device=0):
    """``cv::resize(img, img, Size(width, height))`` (src/app.cpp:166,301: default INTER_LINEAR) on the device -> u8 [height][width][3]."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((int(height), int(width), 3), np.uint8)
    _check(lib.frt_resize_frame(_ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(out), int(height), int(width), device))
    return out


def getCroppedFaces(frame, outputBbox, resize_w=112, resize_h=112, device=0):
    """``getCroppedFaces`` (src/arcface.cpp:3-17) -> u8 BGR [n][resize_h][resize_w][3]."""
    frame = np.ascontiguousarray(frame, np.uint8)
    boxes = np.ascontiguousarray(outputBbox, BBOX_DTYPE)
    out = np.zeros((len(boxes), resize_h, resize_w, 3), np.uint8)
    _check(lib.frt_crop_faces(_ptr(frame), frame.shape[0], frame.shape[1], frame.strides[0], _ptr(boxes), len(boxes), resize_w, resize_h,
                              _ptr(out), device))
    return out


def alignFaces(frame, landmarks, device=0):
    """Optional alignment mode: 5-point similarity warp to the ArcFace template -> u8 BGR [n][112][112][3]."""
    frame = np.ascontiguousarray(frame, np.uint8)
    lm = np.ascontiguousarray(landmarks, np.float32).reshape(-1, 10)
    out = np.zeros((len(lm), 112, 112, 3), np.uint8)
    _check(lib.frt_align_faces(_ptr(frame), frame.shape[0], frame.shape[1], frame.strides[0], _ptr(lm), len(lm), _ptr(out), device))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# class ArcFaceIR50 (src/arcface.h)
# ----------------------------------------------------------------------------------------------------------------------
class ArcFaceIR50:
    classCount = 0  # the reference keeps this as a process-wide static (arcface.cpp:19); per-instance here

    def __init__(self, engineFile, frameWidth=640, frameHeight=480, inputShape=(3, 112, 112), outputDim=512, maxBatchSize=1,
                 maxFacesPerScene=4, knownPersonThreshold=0.65, inputName="input", outputName="output", device=0):
        assert len(inputShape) == 3
        self._h = _vp()
        self.outputDim, self.maxBatchSize, self.maxFacesPerScene = int(outputDim), int(maxBatchSize), int(maxFacesPerScene)
        self.knownPersonThresh = knownPersonThreshold
        _check(lib.frt_embedder_create(os.fsencode(engineFile), *[int(x) for x in inputShape], self.outputDim, self.maxBatchSize, device,
                                       ctypes.byref(self._h)))
        self.matmul = MatMul(device)
        self.croppedFaces = []  # list of dicts: face (u8 BGR crop), x1, y1, x2, y2  (struct CroppedFace, arcface.h:11-15)
        self.classNames = []
        self.classCount = 0
        self._known = None
        self._embeds = np.zeros((0, self.outputDim), np.float32)

    def setPrecision(self, fp32):
        """``True``: fp32 activations / weights / products end to end (BASELINE configs[1]'s "fp32"; slow, a few faces per call);
        ``False`` (default): fp16 on the matrix cores with fp32 accumulation."""
        _check(lib.frt_embedder_set_precision(self._h, 1 if fp32 else 0))

    def setSeFused(self, enable):
        """IR-SE only: SE tail inside conv2's epilogue (default) or as stand-alone launches (equal to float rounding)."""
        _check(lib.frt_embedder_set_se_fused(self._h, 1 if enable else 0))

    def preprocessFace(self, face):
        face = np.ascontiguousarray(face, np.uint8)
        out = np.empty((3, 112, 112), np.float32)
        _check(lib.frt_embedder_preprocess_face(self._h, _ptr(face), _ptr(out)))
        return out

    def doInference(self, chw, batchSize=None):
        x = np.ascontiguousarray(chw, np.float32).reshape(-1, 3, 112, 112)
        b = x.shape[0] if batchSize is None else batchSize
        out = np.empty((b, self.outputDim), np.float32)
        _check(lib.frt_embedder_infer(self._h, _ptr(x), b, _ptr(out)))
        return out

    def initKnownEmbeds(self, num):
        self._known = np.zeros((int(num), self.outputDim), np.float32)

    def addEmbedding(self, className, embedding):
        self.classNames.append(className)
        self._known[self.classCount] = np.asarray(embedding, np.float32)
        self.classCount += 1

    def addEmbeddings(self, classNames, embeddings):
        """Bulk variant."""
        e = np.asarray(embeddings, np.float32)
        self._known[self.classCount:self.classCount + len(e)] = e
        self.classNames.extend(classNames)
        self.classCount += len(e)

    def setGallery(self, embeddings, classNames=None):
        """initKnownEmbeds + N x addEmbedding without per-row copies; ``classNames`` defaults to the row indices."""
        self._known = np.ascontiguousarray(embeddings, np.float32)
        self.classCount = len(self._known)
        self.classNames = classNames if classNames is not None else range(self.classCount)

    def resetEmbeddings(self):
        self.classCount = 0
        self.classNames = []

    def initMatMul(self):
        self.matmul.init(self._known[:self.classCount], self.classCount, self.outputDim)

    def forward(self, image, outputBbox):
        image = np.ascontiguousarray(image, np.uint8)
        boxes = np.ascontiguousarray(outputBbox, BBOX_DTYPE)
        n = len(boxes)
        embeds = np.zeros((n, self.outputDim), np.float32)
        crops = np.zeros((n, 112, 112, 3), np.uint8)
        if n:
            _check(lib.frt_embedder_forward(self._h, _ptr(image), image.shape[0], image.shape[1], image.strides[0], _ptr(boxes), n,
                                            _ptr(embeds), _ptr(crops)))
        self._embeds = embeds
        self.croppedFaces = [dict(face=crops[i], x1=int(b["x1"]), y1=int(b["y1"]), x2=int(b["x2"]), y2=int(b["y2"])) for i, b in enumerate(boxes)]
        return embeds

    def forwardAligned(self, image, outputBbox, landmarks):
        """``forward`` with the optional aligned crop (boxes are only carried into ``croppedFaces``)."""
        image = np.ascontiguousarray(image, np.uint8)
        boxes = np.ascontiguousarray(outputBbox, BBOX_DTYPE)
        lm = np.ascontiguousarray(landmarks, np.float32).reshape(-1, 10)
        n = len(lm)
        assert len(boxes) == n
        embeds = np.zeros((n, self.outputDim), np.float32)
        crops = np.zeros((n, 112, 112, 3), np.uint8)
        if n:
            _check(lib.frt_embedder_forward_aligned(self._h, _ptr(image), image.shape[0], image.shape[1], image.strides[0], _ptr(lm), n,
                                                    _ptr(embeds), _ptr(crops)))
        self._embeds = embeds
        self.croppedFaces = [dict(face=crops[i], x1=int(b["x1"]), y1=int(b["y1"]), x2=int(b["x2"]), y2=int(b["y2"])) for i, b in enumerate(boxes)]
        return embeds

    def featureMatching(self):
        if not len(self.classNames) or not self.croppedFaces:
            raise FrtError(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found")  # arcface.cpp:198
        return self.matmul.calculate(self._embeds, len(self.croppedFaces))

    def getOutputs(self, output_sims):
        sims = np.asarray(output_sims, np.float32).reshape(len(self.croppedFaces), self.classCount)
        arg = sims.argmax(1)  # first maximum == std::max_element (arcface.cpp:210)
        return [self.classNames[a] for a in arg], [float(sims[i, a]) for i, a in enumerate(arg)]

    def matchTop1(self):
        """Fused featureMatching + getOutputs on the device (never materialises the F x N matrix)."""
        if not len(self.classNames) or not self.croppedFaces:
            raise FrtError(FRT_ERR_EMPTY, "Feature matching: No faces in database or no faces found")
        idx, sim = self.matmul.top1(self._embeds)
        return [self.classNames[a] for a in idx], [float(s) for s in sim]

    def close(self):
        if self._h:
            lib.frt_embedder_destroy(self._h)
            self._h = _vp()
        self.matmul.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# request coalescer (opt-in): concurrent one-frame requests -> pipeline batches (frt_coalescer_*)
# ----------------------------------------------------------------------------------------------------------------------
class Coalescer:
    def __init__(self, detector, recognizer, max_frames, window_us=100, match=True):
        self._h = _vp()
        self.det, self.rec = detector, recognizer
        self.max_faces = detector.maxFacesPerScene
        _check(lib.frt_coalescer_create(detector._h, recognizer._h, recognizer.matmul._h if match else None, int(max_frames), int(window_us),
                                        ctypes.byref(self._h)))

    def infer(self, frame, want_embeds=True, want_crops=False):
        """One frame (u8 BGR [H, W, 3]) -> (records of its boxes, embeddings [n, 512] or None[, u8 crops [n, 112, 112, 3]]).  Blocks; call
        it from many threads."""
        frame = np.ascontiguousarray(frame, np.uint8)
        res = np.zeros(self.max_faces, RESULT_DTYPE)
        emb = np.zeros((self.max_faces, 512), np.float32) if want_embeds else None
        crops = np.zeros((self.max_faces, 112, 112, 3), np.uint8) if want_crops else None
        n = ctypes.c_int(0)
        _check(lib.frt_coalescer_infer_crops(self._h, _ptr(frame), frame.shape[0], frame.shape[1], frame.shape[1] * 3, _ptr(res), _ptr(emb), _ptr(crops),
                                             ctypes.byref(n)))
        out = (res[:n.value], (emb[:n.value] if want_embeds else None))
        return out + (crops[:n.value],) if want_crops else out

    def stats(self):
        b, f = ctypes.c_long(0), ctypes.c_long(0)
        _check(lib.frt_coalescer_stats(self._h, ctypes.byref(b), ctypes.byref(f)))
        return int(b.value), int(f.value)

    def close(self):
        if self._h:
            lib.frt_coalescer_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# batched device-resident pipeline (new surface)
# ----------------------------------------------------------------------------------------------------------------------
class Pipeline:
    def __init__(self, detector, recognizer, max_frames, match=True):
        """``match=False``: no matcher stage (records carry match_idx = -1); the caller matches the embeddings itself, e.g. against a
        sharded gallery (dist.py)."""
        self._h = _vp()
        self.det, self.rec = detector, recognizer
        self.max_frames = int(max_frames)
        self.max_faces = detector.maxFacesPerScene
        _check(lib.frt_pipeline_create(detector._h, recognizer._h, recognizer.matmul._h if match else None, self.max_frames, ctypes.byref(self._h)))

    def run(self, frames, want_embeds=True):
        frames = np.ascontiguousarray(frames, np.uint8)
        n = frames.shape[0]
        res = np.zeros(n * self.max_faces, RESULT_DTYPE)
        emb = np.zeros((n * self.max_faces, 512), np.float32) if want_embeds else None
        _check(lib.frt_pipeline_run(self._h, _ptr(frames), n, _ptr(res), _ptr(emb)))
        return res, emb

    def submit(self, frames, results, embeds=None):
        """Asynchronous host boundary: queue one batch (``frames`` u8 [n, H, W, 3], ``results`` a RESULT_DTYPE array of
        n*max_faces records, optional ``embeds`` float32 [n*max_faces, 512]; all C-contiguous, ideally pinned) and return a
        ticket for :meth:`wait`.  The arrays must stay alive and untouched until then."""
        if not (frames.flags.c_contiguous and frames.dtype == np.uint8):
            raise ValueError("frames must be a C-contiguous uint8 array")
        n = frames.shape[0]
        if results.dtype != RESULT_DTYPE or results.size < n * self.max_faces or not results.flags.c_contiguous:
            raise ValueError("results must be a C-contiguous RESULT_DTYPE array of n*max_faces records")
        if embeds is not None and (embeds.dtype != np.float32 or embeds.size < n * self.max_faces * 512 or not embeds.flags.c_contiguous):
            raise ValueError("embeds must be a C-contiguous float32 [n*max_faces, 512] array")
        t = ctypes.c_long(-1)
        _check(lib.frt_pipeline_submit(self._h, _ptr(frames), n, _ptr(results), _ptr(embeds), ctypes.byref(t)))
        return int(t.value)

    def wait(self, ticket):
        _check(lib.frt_pipeline_wait(self._h, int(ticket)))

    def run_dev(self, frames_ptr, n_frames, results_ptr, embeds_ptr=None, ready_event=None):
        """Asynchronous; arguments are raw device addresses (e.g. ``torch.Tensor.data_ptr()``).  ``ready_event``: raw hipEvent_t the
        producer of the frames recorded (``torch.cuda.Event.cuda_event``) - the stages wait for it on the device."""
        if ready_event:
            _check(lib.frt_pipeline_run_dev_after(self._h, _vp(frames_ptr), int(n_frames), _vp(results_ptr), _vp(embeds_ptr) if embeds_ptr else None,
                                                  _vp(ready_event)))
        else:
            _check(lib.frt_pipeline_run_dev(self._h, _vp(frames_ptr), int(n_frames), _vp(results_ptr), _vp(embeds_ptr) if embeds_ptr else None))

    def set_input_sync(self, enable):
        """Safe mode: order every run_dev call behind the work already queued on the pipeline stream (see include/frt.h)."""
        _check(lib.frt_pipeline_set_input_sync(self._h, 1 if enable else 0))

    def check_overlap(self):
        """-> (ratio, warning): ratio ~1 = the stage streams (and the caller's) run side by side, ~n = n of them share a hardware queue;
        warning = the library's message when ratio > 1.5, else ''."""
        r = ctypes.c_float(0)
        _check(lib.frt_pipeline_check_overlap(self._h, ctypes.byref(r)))
        return float(r.value), lib.frt_last_error().decode(errors="replace")

    def sync(self):
        _check(lib.frt_pipeline_sync(self._h))

    def set_overlap(self, enable):
        """Two-stream software pipelining of consecutive calls (detector of call b+1 under embed/match of call b)."""
        _check(lib.frt_pipeline_set_overlap(self._h, 1 if enable else 0))

    def set_graph(self, enable):
        """hipGraph replay of repeated calls (default on)."""
        _check(lib.frt_pipeline_set_graph(self._h, 1 if enable else 0))

    def set_pairing(self, enable):
        """frt_pipeline_set_pairing: -1 (the default) adaptive - submit() calls share a recogniser pass with the next call's only while the
        recogniser is busy anyway; -2 the same for run_dev calls too; 0 / False off; True / 2, 3, 4: always groups of that many consecutive
        calls (results complete when the group is, or at a flush)."""
        _check(lib.frt_pipeline_set_pairing(self._h, int(enable)))

    def merge_stats(self):
        """-> (calls that carried several submit tickets, tickets they carried): adaptive merging at the host boundary."""
        a, b = ctypes.c_long(0), ctypes.c_long(0)
        _check(lib.frt_pipeline_merge_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def graph_stats(self):
        """-> (stage graphs captured, stage graphs replayed) since the pipeline was created (set_graph)."""
        a, b = ctypes.c_long(0), ctypes.c_long(0)
        _check(lib.frt_pipeline_graph_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def pairing_stats(self):
        """-> (recogniser passes that served two calls, passes that served one)."""
        a, b = ctypes.c_long(0), ctypes.c_long(0)
        _check(lib.frt_pipeline_pairing_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def set_align(self, enable):
        """Optional 5-point aligned crop instead of the reference's bbox crop (needs a detector blob with LandmarkHead)."""
        _check(lib.frt_pipeline_set_align(self._h, 1 if enable else 0))

    def set_stream(self, hip_stream):
        """``hip_stream``: raw hipStream_t value (e.g. ``torch.cuda.current_stream().cuda_stream``) or None."""
        _check(lib.frt_pipeline_set_stream(self._h, _vp(hip_stream) if hip_stream else None))

    def close(self):
        if self._h:
            lib.frt_pipeline_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# JPEG frame ingest / reply step (src/app.cpp:296, 328-340)
# ----------------------------------------------------------------------------------------------------------------------
def jpeg_info(data):
    b = np.frombuffer(bytes(data), np.uint8)
    w, h, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _check(lib.frt_jpeg_info(_ptr(b), b.size, ctypes.byref(w), ctypes.byref(h), ctypes.byref(c)))
    return w.value, h.value, c.value


def jpeg_read_coefficients(data):
    """Host half of the decoder: -> (geometry dict, coef int16 [total_blocks, 64] natural order, not dequantised)."""
    b = np.frombuffer(bytes(data), np.uint8)
    g = np.zeros(219, np.int32)
    _check(lib.frt_jpeg_read_coefficients(_ptr(b), b.size, None, 0, _ptr(g)))
    coef = np.zeros((int(g[26]), 64), np.int16)
    _check(lib.frt_jpeg_read_coefficients(_ptr(b), b.size, _ptr(coef), coef.shape[0], _ptr(g)))
    comps = [dict(zip(("h", "v", "bw", "bh", "dw", "dh", "block0"), (int(x) for x in g[5 + 7 * i:12 + 7 * i]))) for i in range(int(g[2]))]
    geo = dict(width=int(g[0]), height=int(g[1]), ncomp=int(g[2]), hmax=int(g[3]), vmax=int(g[4]), comps=comps,
               q=[g[27 + 64 * i:27 + 64 * (i + 1)].astype(np.int32).copy() for i in range(int(g[2]))])
    return geo, coef


def jpeg_write_jfif(quality, width, height, coef_zigzag):
    c = np.ascontiguousarray(coef_zigzag, np.int16)
    out = np.zeros(c.size * 3 + 4096, np.uint8)
    n = ctypes.c_size_t(0)
    _check(lib.frt_jpeg_write_jfif(int(quality), int(width), int(height), _ptr(c), _ptr(out), out.size, ctypes.byref(n)))
    return out[:n.value].tobytes()


def base64_encode(data):
    b = np.frombuffer(bytes(data), np.uint8)
    need = lib.frt_base64_encode(_ptr(b) if b.size else None, b.size, None, 0)
    out = ctypes.create_string_buffer(need)
    lib.frt_base64_encode(_ptr(b) if b.size else None, b.size, out, need)
    return out.value.decode()


class JpegCodec:
    """``cv::imdecode`` in front of the hot path and ``cv::imencode('.jpg')`` behind it: Huffman coding on host threads, transforms on the device."""

    def __init__(self, max_images=32, max_width=1920, max_height=1088, n_threads=0, device=0):
        self._h = _vp()
        self.max_images = int(max_images)
        _check(lib.frt_jpeg_decoder_create(self.max_images, int(max_width), int(max_height), int(n_threads), device, ctypes.byref(self._h)))

    def decode(self, data):
        """bytes -> u8 BGR [h, w, 3] at the image's own size."""
        w, h, _ = jpeg_info(data)
        b = np.frombuffer(bytes(data), np.uint8)
        out = np.zeros((h, w, 3), np.uint8)
        _check(lib.frt_jpeg_decode(self._h, _ptr(b), b.size, _ptr(out), out.size, None, None))
        return out

    def decode_batch_dev(self, blobs, frames_ptr, out_h, out_w, hip_stream=None):
        """list of bytes -> device frames [n][out_h][out_w][3] at raw address ``frames_ptr`` (asynchronous on ``hip_stream``)."""
        keep = [np.frombuffer(bytes(b), np.uint8) for b in blobs]
        n = len(keep)
        ptrs = (ctypes.c_void_p * n)(*[k.ctypes.data for k in keep])
        sizes = (ctypes.c_size_t * n)(*[k.size for k in keep])
        _check(lib.frt_jpeg_decode_batch_dev(self._h, ptrs, sizes, n, _vp(frames_ptr), int(out_h), int(out_w), _vp(hip_stream) if hip_stream else None))

    def encode(self, images, quality=95, device_ptr=None, shape=None, ready_event=None):
        """u8 BGR [n, rows, cols, 3] (host array, or raw device address + ``shape``) -> list of JFIF byte strings.  ``ready_event``: raw
        hipEvent_t recorded behind the producer of a device input (the codec's stream waits for it on the device)."""
        if device_ptr is None:
            a = np.ascontiguousarray(images, np.uint8)
            if a.ndim == 3:
                a = a[None]
            n, rows, cols = a.shape[:3]
            src, dev = _ptr(a), 0
        else:
            n, rows, cols = shape
            src, dev = _vp(device_ptr), 1
        stride = rows * cols * 3 + 4096
        out = np.zeros((n, stride), np.uint8)
        sizes = (ctypes.c_size_t * n)()
        _check(lib.frt_jpeg_encode_batch_after(self._h, src, dev, n, rows, cols, int(quality), _ptr(out), stride, sizes,
                                               _vp(ready_event) if ready_event else None))
        return [out[i, :sizes[i]].tobytes() for i in range(n)]

    def close(self):
        if self._h:
            lib.frt_jpeg_decoder_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def profile_enable(kind):
    _check(lib.frt_profile_enable(int(kind)))


def profile_collect(cap=65536):
    names = ctypes.create_string_buffer(cap * 24)
    ms = np.zeros(cap, np.float64)
    work = np.zeros(cap, np.float64)
    n = lib.frt_profile_collect(names, len(names), _ptr(ms), _ptr(work), cap)
    labels = names.value.decode().split("\n")[:n]
    return labels, ms[:n].copy(), work[:n].copy()
