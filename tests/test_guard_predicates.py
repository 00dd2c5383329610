"""Integer predicates of the guarded fp64 shortcuts (dj_brdf_amd/csrc/djb_device.hpp) whose short forms are argued in
comments: checked exhaustively on the host (tools/near_midpoint_check.c), for the two widths the kernels use."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_near_f32_midpoint_two_instruction_form(tmp_path):
    exe = str(tmp_path / "nmc")
    subprocess.run(["gcc", "-O3", "-o", exe, os.path.join(ROOT, "tools", "near_midpoint_check.c")], check=True)
    out = subprocess.run([exe, "256", "1024"], check=True, capture_output=True, text=True).stdout
    rows = re.findall(r"width (\d+): (\d+) mismatches over 2\^32 low words \((\d+) inside the band\)", out)
    assert [(int(w), int(m)) for w, m, _ in rows] == [(256, 0), (1024, 0)], out
    # the band is 2 width + 1 values of the 29 low bits, times the 8 settings of the three bits above them
    assert [int(h) for _, _, h in rows] == [8 * (2 * 256 + 1), 8 * (2 * 1024 + 1)]
    # the expression checked is the one the header carries
    src = open(os.path.join(ROOT, "dj_brdf_amd", "csrc", "djb_device.hpp")).read()
    assert "<< 3) + (0u - (((unsigned int)0x10000000 - (unsigned int)width) << 3))" in src
    assert "return d <= ((unsigned int)width << 4);" in src
