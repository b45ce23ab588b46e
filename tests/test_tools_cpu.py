"""The measurement tools that back DESIGN.md / profiles/r01_d stay buildable: hipcc cross-compiles them for gfx950, and
the library itself builds with each measurement-aid macro the docs name."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(ROOT, "profiles", "tools", "*.hip"))))
def test_profile_tool_compiles(src, tmp_path):
    out = tmp_path / "tool.o"
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-c", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0


@pytest.mark.parametrize("flag", ["-DVC_STREAM_ONLY", "-DVC_DBG_TIMES", "-DVC_PF=2", "-DVC_EPI_ROWS=1", "-DVC_RCP_MERGE=0",
                                  "-DVC_SWAP_REDUCE=1", "-DVC_LDS_REDUCE=0", "-DVC_LB_SINGLE=3"])
def test_measurement_aid_builds(flag, tmp_path):
    """One translation unit of the likelihood kernel per macro (the full library takes too long for the CPU suite)."""
    src = os.path.join(ROOT, "velocycle_amd", "csrc", "vc_main_vfull_poisson_u16.hip" if "REDUCE" in flag or "RCP" in flag
                       else "vc_main_vu_poisson.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", flag, "-c", src, "-o", str(tmp_path / "k.o")],
                       capture_output=True, text=True, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr[-2000:]
