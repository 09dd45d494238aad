#!/usr/bin/env python3
"""Static instruction histograms of the hot kernels (no GPU needed: hipcc -S cross-compiles gfx950).

    python bench/tools/isa_histogram.py > profiles/r04_isa_histograms.txt

For every kernel named below: the resource lines the compiler prints (VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy), the
instruction count of the whole function and of its loop blocks (the blocks llvm marks "in Loop" / "Loop Header"), split
into the classes the issue-rate microbenchmark prices (bench/ubench_valu.hip): v_mad_i64_i32 / other VALU / LDS / global /
scalar.  The counts are STATIC: a block that is skipped at run time (the rare P = +-Q path of the mixed addition, the
store-factor multiplications of a plain transform) is still counted, so per-path figures quoted in DESIGN.md (the
~1710-instruction common path of the mixed addition) are sums over the blocks that path executes, listed here by label.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "halo2_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# (source file, demangled-name substring, what it is)
KERNELS = [
    ("msm_accumulate.hip", "msm_accumulate<0, false, true, 256>", "registered-table accumulate, Pallas (the roofline kernel)"),
    ("msm_accumulate.hip", "msm_accumulate<1, false, true, 256>", "the same, Vesta"),
    ("msm_sort.hip", "msm_s1_count<1, false>", "bucket sort pass 1: count (Pallas scalars = Fq)"),
    ("msm_sort.hip", "msm_s1_scatter<1, false>", "bucket sort pass 1: scatter"),
    ("msm_sort.hip", "msm_s2_bins", "bucket sort pass 2 (a workgroup per bin)"),
    ("msm_fold.hip", "fold9_finish<0>", "fold: range heads into their buckets"),
    ("msm_fold.hip", "fold9_rowcol<0>", "fold: row / column sums"),
    ("msm_fold.hip", "fold9_planes<0>", "fold: bit planes + final point"),
    ("ntt.hip", "ntt_pass9<0, 10, true>", "NTT first pass of a 2^20 transform (10 stages, bit-reversed gather)"),
    ("ntt.hip", "ntt_pass9<0, 10, false>", "NTT second pass of a 2^20 transform"),
    ("ntt.hip", "ntt_pass9<0, 8, true>", "NTT first pass of the batched / 2^22 plans (8 stages)"),
    ("ntt.hip", "ntt_pass9<0, 8, false>", "NTT later 8-stage pass"),
    ("ntt.hip", "ntt_pass9<0, 6, false>", "NTT last pass of a 2^22 transform (6 stages)"),
]


def classify(op):
    if op.startswith("v_mad_i64_i32") or op.startswith("v_mad_u64_u32"):
        return "mad64"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "global"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith("v_"):
        return "valu"
    return "other"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.split("\n")


def compile_s(src, outdir):
    dst = os.path.join(outdir, src.replace(".hip", ".s"))
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", dst], check=True, stderr=subprocess.DEVNULL)
    return open(dst).read().split("\n")


def functions(lines):
    """name -> (start, end) of every kernel function in the listing"""
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*; @", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[m.group(1)] = (i, j)
            i = j
        i += 1
    return out


def histogram(body):
    """per basic block: (label, in_loop, Counter of opcodes)"""
    blocks = [["entry", False, collections.Counter()]]
    for ln in body:
        s = ln.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", s)
        if m:
            blocks.append([m.group(1), "Loop" in m.group(2), collections.Counter()])
            continue
        m = re.match(r"^; %bb\.\d+:(.*)$", s)
        if m:
            blocks.append([s.split(":")[0].lstrip("; "), "Loop" in m.group(1), collections.Counter()])
            continue
        if s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        blocks[-1][2][s.split()[0]] += 1
    return blocks


def resources(lines, start, end):
    keys = ("NumSgprs", "NumVgprs", "NumAgprs", "TotalNumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "codeLenInByte")
    got = {}
    for ln in lines[end:end + 80]:
        m = re.match(r"^; (\w+): (\d+)", ln.strip())
        if m and m.group(1) in keys and m.group(1) not in got:
            got[m.group(1)] = int(m.group(2))
    return got


def fmt_classes(c):
    cls = collections.Counter()
    for op, k in c.items():
        cls[classify(op)] += k
    order = ("mad64", "valu", "lds", "global", "scratch", "scalar", "wait", "other")
    return "  ".join(f"{k} {cls[k]}" for k in order if cls[k])


def main():
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.split("\n")[0]
    print(f"# static instruction histograms, hipcc -O3 --offload-arch=gfx950 -S (tree {head}; {ver})")
    print("# classes: mad64 = v_mad_i64_i32 / v_mad_u64_u32 (4.85 cycles per wave and SIMD), valu = every other v_* (2.7-4.9),")
    print("#          lds = ds_*, global = global_/buffer_/flat_, scalar = s_* but waits; see profiles/r04_ubench_valu.txt for the rates")
    with tempfile.TemporaryDirectory() as td:
        cache = {}
        for src, want, what in KERNELS:
            if src not in cache:
                lines = compile_s(src, td)
                fn = functions(lines)
                names = list(fn)
                cache[src] = (lines, fn, dict(zip(demangle(names), names)))
            lines, fn, dem = cache[src]
            hits = [d for d in dem if want in d and d.startswith(("void h2::", "h2::"))]
            if not hits:
                print(f"\n## {want}: not found in {src}")
                continue
            name = dem[hits[0]]
            start, end = fn[name]
            blocks = histogram(lines[start + 1:end])
            total = collections.Counter()
            loop = collections.Counter()
            for _, in_loop, c in blocks:
                total.update(c)
                if in_loop:
                    loop.update(c)
            res = resources(lines, start, end)
            print(f"\n## {want}  --  {what}")
            print("   resources: " + "  ".join(f"{k} {v}" for k, v in res.items()))
            print(f"   whole function: {sum(total.values())} instructions   {fmt_classes(total)}")
            print(f"   loop blocks:    {sum(loop.values())} instructions   {fmt_classes(loop)}")
            print("   top opcodes (whole function): " + ", ".join(f"{op} {k}" for op, k in total.most_common(14)))
            big = [(lbl, il, c) for lbl, il, c in blocks if sum(c.values()) >= 60]
            print("   blocks of >= 60 instructions (label, in a loop?, instructions, classes):")
            for lbl, il, c in big:
                print(f"     {lbl:<12} {'loop' if il else '    '} {sum(c.values()):>5}   {fmt_classes(c)}")


if __name__ == "__main__":
    sys.exit(main())
