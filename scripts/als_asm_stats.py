"""Compile ONE instantiation of als_gram_kernel out of csrc/als_kernels.hpp to gfx950 assembly (hipcc -S, ~10 s, no GPU needed) and
report what the hand-scheduled split-f16 loop depends on: registers, spills, and the instruction stream of the block that holds the
matrix instructions (one letter per instruction: M matrix, v VALU, a accvgpr move, S scratch, g global load, d LDS, s SALU, w waitcnt,
n nop).
    python scripts/als_asm_stats.py [T=4] [--show]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def letter(line):
    t = line.strip().split(" ")[0] if line.strip() else ""
    if "accvgpr" in t: return "a"
    if "mfma" in t: return "M"
    if t.startswith("scratch"): return "S"
    if t.startswith("global_load"): return "g"
    if t.startswith("v_"): return "v"
    if t.startswith("s_waitcnt"): return "w"
    if t.startswith("s_nop"): return "n"
    if t.startswith("s_"): return "s"
    if t.startswith("ds_"): return "d"
    return ""


def compile_split_kernel(T=4, loss=False, big=False):
    src = '#include "als_kernels.hpp"\nnamespace bfh {\ntemplate __global__ void als_gram_kernel<%d, true, true, %s, %s, true>(AlsParams, const AlsWork*, int, float*, int);\n}\n' % (
        T, "true" if big else "false", "true" if loss else "false")
    with tempfile.TemporaryDirectory() as d:
        hip, asm = os.path.join(d, "one.hip"), os.path.join(d, "one.s")
        open(hip, "w").write(src)
        cmd = [HIPCC, "-DBFH_ALS_KERNELS_ONLY", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "buffalo_amd", "csrc"), "-S", "--cuda-device-only", hip, "-o", asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-4000:])
        return open(asm).read()


def stats(text):
    out = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", text, re.S):
        if "als_gram_kernel" in m.group(3):
            out.update(agpr=int(m.group(1)), lds=int(m.group(2)), sgpr_spill=int(m.group(4)), vgpr=int(m.group(5)), vgpr_spill=int(m.group(6)))
    i = text.index("als_gram_kernel")
    body = text[text.index(":", i):text.index(".Lfunc_end", i)]
    blocks, cur = [], []
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            blocks.append(cur)
            cur = []
        else:
            cur.append(ln)
    blocks.append(cur)
    best = max(("".join(letter(l) for l in b) for b in blocks), key=lambda s: s.count("M"))
    out["loop"] = best
    out["loop_mfma"] = best.count("M")
    out["loop_valu"] = best.count("v")
    out["loop_scratch"] = best.count("S")
    out["loop_accvgpr"] = best.count("a")
    out["longest_mfma_run"] = max((len(r) for r in re.findall(r"M+", best)), default=0)
    gaps = [len(g) for g in re.split(r"M", best)[1:-1]]
    out["longest_gap"] = max(gaps, default=0)
    return out


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
    st = stats(compile_split_kernel(T))
    loop = st.pop("loop")
    print(st)
    if "--show" in sys.argv:
        for x in range(0, len(loop), 160):
            print("   ", loop[x:x + 160])
