#!/usr/bin/env python3
"""Kernel resource metadata of a HIP object / shared library (gfx950 code objects inside its .hip_fatbin clang offload bundle).

    python tools/codeobj_meta.py face-recognition-cpp-tensorrt_amd/libfrt.so [name-substring ...]

Prints, per kernel: VGPRs (+ AGPRs), SGPRs, spills, scratch bytes, LDS bytes, waves per SIMD the registers allow.  Used by
tests/test_codeobj.py (the build-time check the hand-allocated scalar registers of dwpw_wave_kernel rest on) and while tuning kernels.
No GPU needed: it reads what hipcc wrote."""
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    """-> list of ELF images (bytes) for `arch` found in any clang offload bundle inside the file."""
    data = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos += 24
    return out


def kernels(path, arch="gfx950"):
    """-> {kernel symbol: {field: value}} from the amdhsa.kernels notes of every code object."""
    res = {}
    for img in code_objects(path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if line.lstrip().startswith("- .") and k in ("agpr_count", "args"):
                cur = {}
            if cur is None:
                continue
            if k in ("agpr_count", "sgpr_count", "vgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
                     "group_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size"):
                cur[k] = int(v)
            elif k == "name":
                cur["name"] = v
            elif k == "symbol":
                res[v[:-3] if v.endswith(".kd") else v] = cur
    return res


def demangle(names):
    try:
        o = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, o))
    except OSError:
        return {n: n for n in names}


def waves_per_simd(vgpr_total):
    alloc = max(8, -(-vgpr_total // 8) * 8)
    return min(8, 512 // alloc)


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    dm = demangle(list(ks))
    pats = sys.argv[2:]
    for sym, m in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        name = dm[sym].replace("(anonymous namespace)::", "").replace("void ", "")
        if pats and not any(p in name for p in pats):
            continue
        # on gfx950 .vgpr_count is the unified total (arch VGPRs + AGPRs)
        tot = m.get("vgpr_count", 0)
        print("%-110s vgpr %3d (agpr %3d) sgpr %3d  spill v%d s%d  scratch %4d  lds %6d  waves/SIMD %d" % (
            name[:110], tot, m.get("agpr_count", 0), m.get("sgpr_count", 0), m.get("vgpr_spill_count", 0), m.get("sgpr_spill_count", 0),
            m.get("private_segment_fixed_size", 0), m.get("group_segment_fixed_size", 0), waves_per_simd(tot)))
