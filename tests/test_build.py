"""CPU: properties of the compiled kernels that the product relies on (hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_persistent_gemm_kernels_do_not_spill(tmp_path):
    """csrc/gemm256p.h keeps its MFMA fragments in registers that inline-asm `ds_read`s fill asynchronously: a compiler spill of such
    a register right behind the asm statement would store bytes that have not landed yet (cdna_hip_programming.md, section 5.7).  Every
    instantiated form of the persistent kernel must therefore compile WITHOUT scratch and inside the 256-register budget of two
    waves per SIMD; the layout pairs that do spill are not instantiated (gemm_p.hip: eligibility)."""
    src = os.path.join(ROOT, "open-muse_amd", "csrc", "gemm_p.hip")
    mk = open(os.path.join(ROOT, "open-muse_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^FLAGS_gemm_p\s*=\s*(.*)$", mk, flags=re.M).group(1).split()
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + flags + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "gemm_p.o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = re.findall(r"Function Name: (\S*g256p\S*kernel\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, flags=re.S)
    assert len(kernels) >= 3, r.stderr[-2000:]
    for name, vgprs, scratch in kernels:
        assert int(scratch) == 0 and int(vgprs) <= 256, (name, vgprs, scratch)
