#!/usr/bin/env python3
"""Scan the gfx950 code objects of a HIP object / shared library for the "load -> s_waitcnt vmcnt(0) -> store" chains the compiler emits when an
epilogue interleaves parameter loads (or per-channel branches) with its output stores: every store then waits for ALL earlier memory operations,
previous stores included - a dependent chain of memory round trips at the end of a thread (round 5: dwpw_row4_kernel, dwpw_mfma_kernel,
pw_mfma_kernel, conv3x3_split_kernel lost 10 - 30 us per launch to it).

    python tools/isa_store_chains.py face-recognition-cpp-tensorrt_amd/libfrt.so [min_count=4]

Prints, per kernel, (a) how many `s_waitcnt vmcnt(N)` retire at least one store (the memory operations of a wave retire in order and vmcnt counts loads
and stores alike, so a wait for a load issued behind a store waits for the store's acknowledgement too; read linearly through the function, loops
counted once), (b) how many global / buffer stores sit directly behind an `s_waitcnt vmcnt(0)` (only scalar / branch instructions in between)."""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import codeobj_meta as m  # noqa: E402


def scan(path):
    out = {}
    for img in m.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        for fn in re.split(r"\n[0-9a-f]+ <", txt)[1:]:
            name = fn.split(">")[0]
            n = stores = 0
            waited = False
            pending = []  # outstanding vector-memory operations in issue order (True = store), read linearly through the function
            swaits = 0  # waits that retire at least one store: the thread stalls for a store's acknowledgement
            for line in fn.splitlines()[1:]:
                op = line.strip().split(" ")[0] if line.strip() else ""
                mm = re.search(r"vmcnt\((\d+)\)", line) if op == "s_waitcnt" else None
                if mm:
                    keep = int(mm.group(1))
                    gone = pending[: max(0, len(pending) - keep)]
                    pending = pending[len(gone):]
                    swaits += any(gone)
                elif op.startswith(("global_load", "buffer_load", "flat_load", "global_atomic", "buffer_atomic")):
                    pending.append(False)
                elif op.startswith(("global_store", "buffer_store", "flat_store")):
                    pending.append(True)
                if op == "s_waitcnt" and "vmcnt(0)" in line:
                    waited = True
                elif op.startswith(("global_store", "buffer_store", "flat_store")):
                    stores += 1
                    if waited:
                        n += 1
                    waited = False
                elif op.startswith(("s_", "v_mov", "v_cndmask", "v_add", "v_max", "v_lshl", "v_mad")) or not op:
                    pass  # (address / select arithmetic between the wait and the store does not end the pattern)
                else:
                    waited = False
            out[name] = (n, stores, swaits)
    return out


if __name__ == "__main__":
    res = scan(sys.argv[1])
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dm = m.demangle(list(res))
    for k, (n, st, sw) in sorted(res.items(), key=lambda kv: -kv[1][2]):
        if sw >= lo:
            print("%4d waits that retire a store, %4d of %4d stores behind vmcnt(0)   %s" % (sw, n, st, dm[k].replace("(anonymous namespace)::", "").replace("void ", "")[:120]))
