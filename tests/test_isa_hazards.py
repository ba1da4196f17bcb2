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


SYNTHETIC = """
	.text
kernel_a:
	v_mfma_f64_16x16x4_f64 a[64:71], v[114:115], v[118:119], a[64:71]
	v_mfma_f64_16x16x4_f64 a[64:71], v[102:103], v[136:137], a[64:71]
	s_cbranch_vccnz .LBB0_2
	v_mfma_f64_16x16x4_f64 a[0:7], v[112:113], v[84:85], a[8:15]
	v_mfma_f64_16x16x4_f64 a[0:7], v[102:103], v[98:99], a[0:7]
.LBB0_2:
	s_nop 4
	v_accvgpr_read_b32 v10, a64
	s_nop 11
	v_accvgpr_read_b32 v11, a65
	s_endpgm
kernel_b:
	v_mfma_f64_16x16x4_f64 a[64:71], v[102:103], v[136:137], a[64:71]
	s_cbranch_vccnz .LBB1_2
	v_mfma_f64_16x16x4_f64 a[0:7], v[102:103], v[98:99], a[0:7]
.LBB1_2:
	s_nop 15
	s_nop 7
	v_accvgpr_read_b32 v10, a64
	global_load_dwordx2 v[2:3], v[4:5], off sc1
	s_endpgm
"""


def test_the_scanner_sees_a_hazard_through_a_branch(tmp_path):
    """The shape hipcc produced in k_sweep_rl (twice in round 2, once in round 3): the last MFMA of a chain, a taken branch,
    `s_nop 4`, then the read of the chain's first register -- 6 wait states on the taken path although the fall-through path
    has seven more MFMAs in between.  kernel_b: the same with the hand-written wait states (clean), and an 8-byte sc1 load
    for the scanner's second check.  (Until round 3 this test compiled the library with -DPNMOL_NO_MFMA_GUARD and expected
    hits; with the bulk loop's exit on a `break` the compiler no longer produces the hazard there, so the scanner is checked
    against the recorded shape instead.)"""
    src = tmp_path / "synthetic.s"
    src.write_text(SYNTHETIC)
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "mfma_hazard_scan.py"), str(src), "10"],
                         check=True, capture_output=True, text=True).stdout
    lines = res.strip().splitlines()
    assert int(lines[-1].split()[0]) == 1, res
    assert "a64" in lines[0] and "after 6 wait states" in lines[0], res
    assert int(lines[-2].split()[0]) == 1, res
