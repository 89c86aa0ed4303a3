import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import face-recognition-cpp-tensorrt_amd/ (not a valid identifier) as module ``frt_amd``."""
    name = "frt_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


@pytest.fixture(scope="session")
def frt():
    return load_pkg()


@pytest.fixture(scope="session")
def synth(frt):
    return frt.synth


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def blobs(frt, tmp_path_factory):
    """Synthetic weight blobs + the state dicts they were written from (regenerated from seeds, never committed)."""
    d = tmp_path_factory.mktemp("weights")
    out = {}

    def get(kind):
        if kind in out:
            return out[kind]
        s = frt.synth
        if kind == "det":
            sd = s.retinaface_state(1)
            path = frt.write_weights(str(d / "retina.frtw"), sd, frt.weights_io.KIND_RETINAFACE_MNET025)
        elif kind == "det_ldm":  # full export WITH the landmark head (optional alignment mode)
            sd = s.retinaface_state(1, landmarks=True)
            path = frt.write_weights(str(d / "retina_ldm.frtw"), sd, frt.weights_io.KIND_RETINAFACE_MNET025)
        elif kind == "ir":
            sd = s.arcface_state(2, "ir", calib=s.load_calibration("ir"))
            path = frt.write_weights(str(d / "ir50.frtw"), sd, frt.weights_io.KIND_ARCFACE_IR50)
        elif kind == "ir_se":
            sd = s.arcface_state(2, "ir_se", calib=s.load_calibration("ir_se"))
            path = frt.write_weights(str(d / "irse50.frtw"), sd, frt.weights_io.KIND_ARCFACE_IR_SE50)
        else:
            raise KeyError(kind)
        out[kind] = (path, sd)
        return out[kind]

    return get


def face_input(faces):
    """u8 BGR [F,112,112,3] -> float32 planar RGB [F,3,112,112] (arcface.cpp:105-114), in numpy."""
    x = (faces[..., ::-1].astype(np.float32) - 127.5) * 0.0078125
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))
