"""Static guard on the built device code (no GPU needed): no packed-FP32 VALU instruction may carry an `op_sel:` modifier.

Why: on gfx950 with ROCm 7.2, `v_pk_fma_f32 ... op_sel:[0,1,0]` (the low result lane reading the HIGH source register)
returned wrong low-lane values in lanes 48-63 in 20/20 runs whenever a kernel with MFMA -> VALU-read sequences shared the SIMD
(two streams), and 0/20 with the high element first copied to a low register (tools/race_mixed.py).  The
kernels without matrix instructions are therefore built with packed FP32 disabled (EAE_NO_PK) and the MFMA kernels, which use
packed FP32 for the BatchNorm transforms, must not contain the form."""
import os

import pytest

from eae_amd import build as B

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(B.LIB), reason="libeae.so not built")
@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_no_op_sel_on_packed_fp32():
    cos = B.device_code_objects()
    assert cos, "no gfx950 code object found in libeae.so (compressed fat binary?)"
    total, opsel = B.scan_packed_fp32()
    assert total > 1000, "the MFMA kernels are expected to use packed FP32 for the BatchNorm transforms"
    assert opsel == 0, f"{opsel} packed-FP32 instructions with op_sel found"


@pytest.mark.skipif(not os.path.exists(B.LIB), reason="libeae.so not built")
@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_no_wide_buffer_store_with_an_sgpr_soffset():
    """buffer_store_dwordx3/x4 with soffset in an SGPR: hipcc omits the wait state in front of a VALU write of the data registers, and
    gfx950 then stores the overwritten dword now and then (round 4, found as a run-to-run difference of the stored dy tensor)."""
    total, sgpr = B.scan_wide_stores_sgpr_soffset()
    assert total > 0, "the backward-data kernels are expected to store dy with 16-byte buffer stores"
    assert sgpr == 0, f"{sgpr} wide buffer stores with an SGPR soffset found"
