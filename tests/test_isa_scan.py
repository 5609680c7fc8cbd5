"""CPU: no kernel of the library contains the packed-fp32 operand form that proved irreproducible on gfx950 (csrc/common.h, HPFG_NO_PK_F32).
tools/pk_opsel_scan.sh compiles every translation unit to gfx950 assembly (no GPU needed) and lists offending kernels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_kernel_uses_a_cross_half_op_sel_on_packed_fp32():
    out = subprocess.run([os.path.join(ROOT, "tools", "pk_opsel_scan.sh")], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip() and not ln.startswith("#")]
    assert not lines, "kernels with v_pk_*_f32 op_sel cross-half operands (give them HPFG_NO_PK_F32 or hpfg_own_vgpr):\n" + "\n".join(lines)
