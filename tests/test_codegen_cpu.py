"""What the hand-scheduled split-f16 ALS loop (csrc/als_kernels.hpp: `fused`) relies on, checked on the compiler's output -- no GPU:
hipcc cross-compiles the one kernel to gfx950 assembly (~10 s) and scripts/als_asm_stats.py reads it.  The loop was laid out by
hand because a lone wave per SIMD overlaps the matrix pipe with the vector pipe only if the two alternate in the instruction stream;
a compiler or source change that lets the matrix instructions clump again, or makes the register allocator shuttle the accumulators
(both seen while the kernel was written: DESIGN.md 4.5), costs 10-40 % and no parity test notices."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_split_f16_loop_keeps_its_shape():
    import als_asm_stats as A
    st = A.stats(A.compile_split_kernel(4))
    loop = st.pop("loop")
    print(st)
    assert st["loop_mfma"] == 60, st                  # two 16-entry groups x (10 tiles x 3 products) in one basic block
    assert st["loop_scratch"] == 0, st                # nothing spilled inside the loop
    assert st["loop_accvgpr"] <= 4, st                # the accumulators stay where they are
    assert st["longest_mfma_run"] <= 2, (st, loop)    # matrix instructions alternate with the preparation ...
    assert st["longest_gap"] <= 40, (st, loop)        # ... and no stretch of other work is much longer than one matrix instruction covers
    assert st["loop_valu"] <= 520, st                 # ~230 VALU per group (measured 472 per pair)
    assert st["vgpr_spill"] <= 48, st                 # whole kernel (the per-row phases spill a few registers)
    assert st["lds"] <= 80 * 1024, st                 # FF tiles (64 KB) + per-wave vectors: one block per CU by registers, two by LDS
