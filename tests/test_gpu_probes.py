"""The wave-level primitives the one-launch kernels are built from (common.hpp: sums, scans, min / max / or of a wave's 64
lanes as data-parallel-primitive adds) checked in isolation against serial loops: tools/probes/wave_sum_probe.hip, compiled
here with hipcc for gfx950 (the prebuilt build/wave_sum_probe is used when the sources have not changed since)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_primitives_against_serial_loops():
    src = os.path.join(ROOT, "tools", "probes", "wave_sum_probe.hip")
    exe = os.path.join(ROOT, "build", "wave_sum_probe")
    hdr = os.path.join(ROOT, "graphblast_amd", "csrc", "common.hpp")
    stale = not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if stale:
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-I" + os.path.join(ROOT, "graphblast_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 of 4096 waves wrong" in out.stdout, out.stdout
