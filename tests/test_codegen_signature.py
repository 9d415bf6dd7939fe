"""Guard against silent code-generation regressions of the synthesis kernels (CPU test: reads the built library).

The kernels sit at their register limit and their timing depends on what the register allocator spills: in round 3 a
dead reference to one variable inside the helper loop took the all-double kernel from 95 to 119 spilled registers and from
21.5 to 25.5 ms per 4096 x 2 s, unnoticed for half a round ("The helper loop's text").  This test compares the
scratch bytes and spilled registers of the product's main kernel shapes, read from the code object inside
gama_tts_amd/lib/libgama_vtm.so, with tests/golden/codegen_signature.json.  A failure is not a wrong result -- it says "the
generated code of this kernel changed: measure it against the previous build (tools/ab.py) before accepting", after which
the file is refreshed with    python tests/test_codegen_signature.py --write

Fewer spills are not automatically faster (the float kernel is 20 % slower in the build that spills least), so the test
bounds growth only and prints the whole table."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gama_tts_amd", "lib", "libgama_vtm.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "codegen_signature.json")
LLVM = "/opt/rocm/lib/llvm/bin"
# the shapes the product picks for batch > 512 / <= 256 on the 10 + 6 tube: <CT, ST, SectionDelay, rows, chunk, helpers, layout>
WATCHED = [
    "float, float, 1, 4, 48, 11, 0", "float, float, 2, 4, 48, 11, 0", "float, float, 1, 1, 144, 3, 0", "float, float, 2, 1, 144, 3, 0",
    "double, float, 1, 4, 32, 7, 0", "double, float, 2, 4, 32, 7, 0", "double, float, 1, 1, 96, 3, 0",
    "double, double, 1, 4, 32, 7, 0", "double, double, 2, 4, 32, 7, 0", "double, double, 1, 1, 84, 3, 0",
]


def signature():
    """{template arguments: {"scratch": bytes per lane, "vgpr_spill": n, "vgprs": n}} of every vtm_synth_kernel in the library."""
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        return None
    with tempfile.TemporaryDirectory() as wd:
        fat, co = os.path.join(wd, "fat.bin"), os.path.join(wd, "k.co")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
        subprocess.run([tools[1], "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or "vtm_synth_kernel" not in name.group(1):
            continue
        # _ZN4gvtm2v216vtm_synth_kernelIffLi2ELi4ELi48ELi11ELi0EEEv...: two type letters, five integers
        m = re.search(r"vtm_synth_kernelI([fd])([fd])Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEE", name.group(1))
        if not m:
            continue
        ty = {"f": "float", "d": "double"}
        args = ", ".join([ty[m.group(1)], ty[m.group(2)]] + [m.group(i) for i in range(3, 8)])
        num = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1))  # noqa: E731
        out[args] = {"scratch": num("private_segment_fixed_size"), "vgpr_spill": num("vgpr_spill_count"), "vgprs": num("vgpr_count")}
    return out


def test_main_kernels_do_not_spill_more_than_recorded():
    sig = signature()
    if sig is None:
        pytest.skip("library or LLVM binary tools not present")
    golden = json.load(open(GOLDEN))["kernels"]
    lines, grown = [], []
    for name in WATCHED:
        assert name in sig, "kernel shape %s is not in the library any more: update WATCHED and the golden file" % name
        now, then = sig[name], golden[name]
        lines.append("%-34s scratch %4d B (recorded %4d)  spilled registers %3d (recorded %3d)" % (name, now["scratch"], then["scratch"], now["vgpr_spill"], then["vgpr_spill"]))
        if now["scratch"] > then["scratch"] * 1.15 + 16:
            grown.append(name)
    print("\n".join(lines))
    assert not grown, ("the generated code of %s spills more than recorded: measure against the previous build (tools/ab.py) and, if "
                       "accepted, refresh tests/golden/codegen_signature.json with  python tests/test_codegen_signature.py --write\n%s"
                       % (grown, "\n".join(lines)))


if __name__ == "__main__":
    sig = signature()
    if sig is None:
        sys.exit("library or LLVM binary tools not present")
    if "--write" in sys.argv:
        json.dump({"_comment": "scratch bytes per lane / spilled registers / registers of every vtm_synth_kernel shape in gama_tts_amd/lib/libgama_vtm.so "
                               "(tests/test_codegen_signature.py --write); hipcc " + (shutil.which("hipcc") or "/opt/rocm/bin/hipcc"),
                   "kernels": dict(sorted(sig.items()))}, open(GOLDEN, "w"), indent=1)
        print("wrote", GOLDEN, len(sig), "kernels")
    else:
        for k, v in sorted(sig.items()):
            print(k, v)
