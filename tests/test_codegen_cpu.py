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


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_pc_kernel_keeps_its_wait_counts_and_registers():
    """als_pc_kernel (csrc/als_pc.hpp) depends on two properties of the compiler's output that no parity test sees and that were
    both lost and found while it was written (profiles/r04_als_pc_steps.txt):
      * the producer's loop waits for the group it loaded THREE steps ago with s_waitcnt vmcnt(N), N >= 16 -- one control-flow path
        without the unconditional loads and every wait becomes vmcnt(0), i.e. no memory overlap at all (1.7 us per group);
      * the consumer's matrix instructions run from registers: no scratch traffic in their blocks (a second copy of the stream
        behind a branch made the allocator shuttle the accumulators: 900 spilled registers)."""
    import re
    import subprocess
    import tempfile
    src = ('#include "als_kernels.hpp"\nnamespace bfh {\ntemplate __global__ void als_pc_kernel<4, false, false>(AlsParams, const AlsWork*, int, float*, '
           'const float*, const int*, int*);\n}\n')
    with tempfile.TemporaryDirectory() as d:
        hip, asm = os.path.join(d, "one.hip"), os.path.join(d, "one.s")
        open(hip, "w").write(src)
        cmd = [HIPCC, "-DBFH_ALS_KERNELS_ONLY", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "buffalo_amd", "csrc"), "-S", "--cuda-device-only", hip, "-o", asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    i = text.index("als_pc_kernelILi4ELb0ELb0E")
    body = text[text.index(":", i):text.index(".Lfunc_end", i)]
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    prep, gather, matrix = [], [], []
    for b in blocks:
        ins = [l.strip() for l in b.split("\n") if l.strip() and not l.strip().startswith(";")]
        if sum("v_fma_mix" in l for l in ins) >= 32:
            prep.append(ins)
        if sum(l.startswith("global_load_dwordx4") for l in ins) >= 8:
            gather.append(ins)
        if sum("v_mfma" in l for l in ins) >= 10:
            matrix.append(ins)
    assert len(prep) == 4 and len(gather) == 4 and len(matrix) >= 2, (len(prep), len(gather), len(matrix))
    for ins in gather:      # the eight row loads of a step are never preceded by a drain of everything in flight
        waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in ins if "vmcnt" in l]
        assert all(w >= 8 for w in waits), waits
    for ins in prep:        # the group being prepared was loaded three steps ago: two newer groups (16 loads) may stay in flight
        waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in ins if "vmcnt" in l]
        assert all(w >= 16 for w in waits), waits
    for ins in matrix + prep + gather:
        assert not any(l.startswith("scratch_") for l in ins), [l for l in ins if l.startswith("scratch_")][:4]
    spill = int(re.search(r"als_pc_kernelILi4ELb0ELb0E.*?\.vgpr_spill_count:\s+(\d+)", text, re.S).group(1))
    assert spill <= 48, spill   # the per-row solve spills a few registers; the loops none


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_wide_split_kernel_fits_three_blocks_per_cu():
    """als_wide_kernel<5, ., SPLIT> (d = 160) runs at three blocks per CU, which it owes to properties of the compiler's output that no parity
    test sees (profiles/r05_als_wide_d160.txt): 168 registers; the matrix loops of the three consumer roles (18 / 18 / 9 instructions per
    group) and the producer's store blocks all but free of scratch traffic (at most one reload per group, nothing stored); no flat loads (a pointer laundered through an asm statement is a FLAT pointer
    until it is cast back: the FF tiles were read that way); the spills that remain sit in per-row code and stay few -- the row end lost 1.7 ms
    per epoch when every lane constant was reloaded from scratch before each use (als_fresh_lane)."""
    import re
    import subprocess
    import tempfile
    src = ('#include "als_kernels.hpp"\nnamespace bfh {\ntemplate __global__ void als_wide_kernel<5, false, true>(AlsParams, const AlsWork*, int, float*, int);\n}\n')
    with tempfile.TemporaryDirectory() as d:
        hip, asm = os.path.join(d, "one.hip"), os.path.join(d, "one.s")
        open(hip, "w").write(src)
        cmd = [HIPCC, "-DBFH_ALS_KERNELS_ONLY", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "buffalo_amd", "csrc"), "-S", "--cuda-device-only", hip, "-o", asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    name = "als_wide_kernelILi5ELb0ELb1E"
    i = text.index(name)
    body = text[text.index(":", i):text.index(".Lfunc_end", i)]
    meta = text[text.index("amdhsa.kernels"):]
    vg = int(re.search(name + r".*?\.vgpr_count:\s+(\d+)", meta, re.S).group(1))
    spill = int(re.search(name + r".*?\.vgpr_spill_count:\s+(\d+)", meta, re.S).group(1))
    assert vg <= 168, vg              # three waves per SIMD
    assert spill <= 260, spill        # (226 when written; 350 - 450 cost the row end 1 - 1.5 ms per epoch)
    assert "flat_load" not in body
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    mfma, stores, scratch_total = [], 0, 0
    for b in blocks:
        ins = [l.strip() for l in b.split("\n") if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
        n_scr = sum(l.startswith("scratch_") for l in ins)
        scratch_total += n_scr
        n_mf = sum("v_mfma" in l for l in ins)
        if n_mf:
            mfma.append(n_mf)
            assert n_scr <= 1, (n_mf, n_scr)     # (one lane constant reloaded per group in the 18-instruction roles -- the measured state; no stores)
            assert not any(l.startswith("scratch_store") for l in ins)
        if sum(l.startswith("ds_write") for l in ins) >= 20:      # the producer's slot stores
            stores += 1
            assert n_scr <= 2, n_scr
    assert sorted(mfma) == [9, 18, 18, 18] or sorted(mfma) == [9, 18, 18], mfma   # l h + h l + h h per tile: rows {0, 4}, {1, 3} (6 tiles each), {2} (3)
    assert stores >= 3, stores        # one per row set of the producer's pipeline
    assert scratch_total <= 230, scratch_total
