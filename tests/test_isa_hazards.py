"""The shipped kernels' ISA holds no read of an MFMA result with too few wait states behind the MFMA.

hipcc (ROCm 7.2) normally separates an MFMA from a VALU / LDS read of its accumulator by `s_nop`s.  Twice in this repository
it put them BEHIND the first reads, where an MFMA chain ended a conditional block (DESIGN.md section 3a''): accumulator
element 3 -- rows fk + 12 of a tile -- came out stale, only in builds without the timeline stamps.  The kernels carry
hand-written wait states there (`mfma_result_guard`); `tools/mfma_hazard_scan.py` walks the assembly (through branches)
and this test keeps it clean, and checks that the scanner does see the hazard when the guard is compiled out."""
import pathlib
import shutil
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CSRC = ROOT / "pnmol-experiments_amd" / "csrc"


def _scan(src, tmp_path, *defines):
    out = tmp_path / (src.stem + "".join(defines) + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", f"-I{ROOT / 'include'}", "-S",
                    "--cuda-device-only", *defines, str(src), "-o", str(out)], check=True)
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "mfma_hazard_scan.py"), str(out), "10"],
                         check=True, capture_output=True, text=True).stdout
    _scan.sc1x2 = int(res.strip().splitlines()[-2].split()[0])     # "N 8-byte sc1 loads" (second check of the scanner)
    return int(res.strip().splitlines()[-1].split()[0]), res


@pytest.mark.skipif(not pathlib.Path(HIPCC).exists(), reason="hipcc not available")
@pytest.mark.parametrize("name", ["pnmol_hip.hip", "pnmol_sqrt.hip"])
def test_no_mfma_result_is_read_too_early(tmp_path, name):
    hits, report = _scan(CSRC / name, tmp_path)
    assert hits == 0, report
    # hand-over data is never read with 8-byte sc1 loads (they were served stale L2 lines: DESIGN.md section 3a'')
    assert _scan.sc1x2 == 0, report


@pytest.mark.skipif(not pathlib.Path(HIPCC).exists(), reason="hipcc not available")
def test_the_scanner_sees_the_hazard_without_the_guard(tmp_path):
    hits, report = _scan(CSRC / "pnmol_hip.hip", tmp_path, "-DPNMOL_NO_MFMA_GUARD")
    assert hits > 0, "the compiler no longer produces the hazard: the guard (and this test) can go"
