"""Build-time checks on the code objects inside libfrt.so (no GPU: reads hipcc's kernel metadata).

dwpw_wave_kernel (csrc/kernels_det_wave.hip) keeps its depthwise weights in FIXED scalar registers s[56:95], above the kernel's
amdgpu_num_sgpr(48) cap, from inline asm.  That rests on two properties of the compiled kernel which a toolchain bump could change
silently (round-4 advisor finding): the kernel descriptor must allocate scalar registers up to s95, and the kernel must not use scratch
(a spill of an in-flight asm register to scratch would be a wrong sum).  Checked here for every instantiation on every build."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "face-recognition-cpp-tensorrt_amd", "libfrt.so")


def _meta():
    spec = importlib.util.spec_from_file_location("codeobj_meta", os.path.join(ROOT, "tools", "codeobj_meta.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libfrt.so not built")
    ks = _meta().kernels(LIB)
    assert len(ks) > 100, "no gfx950 kernel metadata found in libfrt.so"
    return ks


def test_wave_kernels_allocate_their_hand_assigned_scalar_registers(kernels):
    wave = {k: v for k, v in kernels.items() if "dwpw_wave_kernel" in k}
    assert len(wave) >= 3
    for name, m in wave.items():
        assert m["sgpr_count"] >= 96, (name, m)               # s[56:95] lie inside the wave's allocation
        assert m["private_segment_fixed_size"] == 0, (name, m)  # no scratch: nothing in flight can be spilled to memory
        assert m["vgpr_spill_count"] == 0, (name, m)


def test_scratch_users_are_the_known_ones(kernels):
    """Kernels that touch scratch memory are few and known (a new one is a regression worth a look before it ships): the JPEG block
    encoder's private coefficient array, the two PRE = 4 small-grid conv_dw variants (25 spilled registers, used at <= 256 workgroups where
    the loads in flight matter more) and conv_s2c64_kernel (6)."""
    known = ("jpeg_encode_blocks_kernel", "dwpw_mfma_kernelILi1ELi2ELi32ELi2ELi2ELb0ELb1ELi4E", "dwpw_mfma_kernelILi2ELi2ELi32ELi2ELi2ELb0ELb1ELi4E",
             "conv_s2c64_kernel")
    bad = {k: (m.get("private_segment_fixed_size"), m.get("vgpr_spill_count")) for k, m in kernels.items()
           if (m.get("private_segment_fixed_size", 0) or m.get("vgpr_spill_count", 0)) and not any(n in k for n in known)}
    assert not bad, bad
