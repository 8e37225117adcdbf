"""The pipelined kernels issue their global loads from inline asm; hipcc does not track those, so it could copy,
spill or read a destination register before the data has landed (tools/audit_asm_loads.py).  This test
cross-compiles the two sources to gfx950 assembly (no GPU needed) and requires a clean audit for every shipped
kernel that uses the protocol -- a guard against a compiler or code change breaking it silently."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-face-detection-from-scratch_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

KERNELS = {
    "fdet_wgrad3x3_x3.hip": ["_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi4ELi0ELi16E", "_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi1ELi0ELi16E",
                             "_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi1ELi0ELi32E"],
    "fdet_stem_x3.hip": ["_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipe", "_ZN12_GLOBAL__N_120k_stem_wgrad_x3_pipe"],
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", sorted(KERNELS))
def test_no_instruction_touches_in_flight_asm_loads(src, tmp_path):
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    for k in KERNELS[src]:
        nload, bad = mod.audit(k, str(out), verbose=False)
        assert nload > 0, f"{k}: no asm buffer loads found (kernel renamed?)"
        assert bad == 0, f"{k}: {bad} instructions touch an in-flight asm load destination"
