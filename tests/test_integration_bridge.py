"""integration/lajolla_hip_bridge.cpp is the binding a maintainer of the reference would add: it must compile
against the reference's own headers and the C ABI header.  Checked where the reference exists (the build container)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LJ_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference's headers are not on this machine")
def test_bridge_compiles_against_the_reference_headers():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(REF, "src"), "-I" + os.path.join(REF, "embree", "include"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "lajolla_hip_bridge.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_bridge_has_no_elisions():
    src = open(os.path.join(ROOT, "integration", "lajolla_hip_bridge.cpp")).read()
    assert "/* ..." not in src and "..." not in src.replace("...)", "")
