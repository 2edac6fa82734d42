"""Register / scratch figures of the search kernels' code objects, pinned (advisor r05: the chunk-of-32 instance's win rests
on how much the compiler spills, and nothing held that down).

Reads the notes of the built objects (kbmod_amd/_obj, what tools/kernel_resources.py prints): no GPU needed, only the LLVM
binutils of the ROCm install that built them.  The bounds are today's numbers with a little air: a change that pushes the
headline instance's lists or sums into scratch memory -- every C++ re-arrangement of the chunk finish tried in rounds 5 and 6
did, by hundreds of bytes -- fails here before it reaches a GPU."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _kernels():
    import kernel_resources as kr

    if not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")):
        pytest.skip("no LLVM binutils on this machine")
    obj = os.path.join(kr.OBJ, "search_lds.o")
    if not os.path.exists(obj):
        from kbmod_amd import build

        build.build_hip()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cur = None
        for line in kr.notes_of(obj, tmp).split("\n"):
            if line.startswith("  - .agpr_count"):
                cur = {}
            m = re.match(r"\s+-?\s*\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(.*)", line)
            if m and cur is not None and line.startswith("    ."):
                cur[m.group(1)] = m.group(2)
                if m.group(1) == "name":
                    out[m.group(2)] = cur
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"\(.*", "", n).replace("void ", ""): v for n, v in zip(names, out.values())}


@pytest.mark.parametrize("instance, max_scratch, max_spills", [
    ("kb::kb_search_lds<8, 16, 16, 4, true, false, 3>", 32, 8),      # the headline: packed lists, chunks of 16, 64 x 16 tiles
    ("kb::kb_search_lds<8, 16, 8, 4, true, false, 3>", 32, 8),
    ("kb::kb_search_lds<8, 16, 16, 4, true, true, 0>", 64, 16),      # the sigma-G emit (configs[2])
    ("kb::kb_search_lds<16, 16, 16, 4, true, false, 4>", 128, 48),   # pooled stable lists (what every rank of the exchange runs)
    ("kb::kb_search_lds<8, 32, 16, 4, true, false, 3>", 450, 210),   # chunks of 32: 64 sums next to the lists (396 B, 194 registers)
])
def test_scratch_and_spills_of_the_search_instances_stay_where_they_are(instance, max_scratch, max_spills):
    k = _kernels()
    assert instance in k, sorted(k)[:5]
    notes = k[instance]
    assert int(notes["vgpr_count"]) <= 128, notes   # four waves per SIMD
    assert int(notes.get("private_segment_fixed_size", 0)) <= max_scratch, notes
    assert int(notes.get("vgpr_spill_count", 0)) <= max_spills, notes
