"""Static guard on what the compiler makes of the hot kernels (no GPU: hipcc -S cross-compiles gfx950; bench/tools/isa_histogram.py does the
parsing): the occupancy each kernel is designed for and the absence of spills on its hot path are properties a source change can lose
silently -- results stay bit-exact and only the clock notices.  DESIGN.md sections 4.2 / 5.5 state the figures asserted here."""
import collections
import concurrent.futures
import importlib.util
import os
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_histogram", os.path.join(ROOT, "bench", "tools", "isa_histogram.py"))
ih = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ih)


@pytest.fixture(scope="module")
def listings():
    if not os.path.exists(ih.HIPCC):
        pytest.skip("hipcc not installed")
    with tempfile.TemporaryDirectory() as td:
        with concurrent.futures.ThreadPoolExecutor(2) as ex:
            got = list(ex.map(lambda src: ih.compile_s(src, td), ["msm_accumulate.hip", "ntt.hip"]))
    out = {}
    for src, lines in zip(["msm_accumulate.hip", "ntt.hip"], got):
        fn = ih.functions(lines)
        names = list(fn)
        out[src] = (lines, fn, dict(zip(ih.demangle(names), names)))
    return out


def _kernel(listings, src, want):
    lines, fn, dem = listings[src]
    hit = [d for d in dem if want in d and d.startswith(("void h2::", "h2::"))][0]
    start, end = fn[dem[hit]]
    return ih.resources(lines, start, end), ih.histogram(lines[start + 1:end])


@pytest.mark.parametrize("block", [256, 512])
@pytest.mark.parametrize("curve", [0, 1])
def test_accumulate_keeps_its_registers_and_its_instruction_count(listings, curve, block):
    """msm_accumulate<FB, false, true>: sized so that the sort / fold kernels of other streams fit beside two of its waves per SIMD
    (<= 168 VGPRs, three waves by the register file); the mixed addition's common path carries no scratch traffic (the only spills
    belong to the out-of-line P = +-Q path); 1151 multiply-adds per addition (8 products, 2 squares, one fused pair)."""
    # (block 512: the one-workgroup-per-CU launch shape of the grouped generic multiexp, csrc/msm_generic.hip -- the same body and the same
    # 168-register budget, so that the fold kernels of the previous group still fit beside its two waves per SIMD)
    res, blocks = _kernel(listings, "msm_accumulate.hip", f"msm_accumulate<{curve}, false, true, {block}>")
    assert res["NumVgprs"] <= 168 and res["NumAgprs"] == 0 and res["Occupancy"] >= 3, res
    assert res["ScratchSize"] <= 512, res
    big = [(lbl, c) for lbl, in_loop, c in blocks if in_loop and sum(c.values()) >= 200]
    assert len(big) == 2, [(lbl, sum(c.values())) for lbl, c in big]       # the two straight-line halves of the addition
    mads = sum(c["v_mad_i64_i32"] for _, c in big)
    total = sum(sum(c.values()) for _, c in big)
    assert mads == 1151, mads
    assert total <= 1640, total                                             # (1607 when this guard was written; the whole common path ~1710)
    for lbl, c in big:
        assert not any(op.startswith("scratch_") for op in c), (lbl, [op for op in c if op.startswith("scratch_")])


@pytest.mark.parametrize("r,first", [(10, "true"), (10, "false"), (8, "true"), (8, "false"), (6, "false"), (7, "false"), (11, "true"), (11, "false"), (12, "false")])
def test_ntt_passes_keep_four_waves_per_simd(listings, r, first):
    """ntt_pass9<F, R, FIRST>: 1024 lanes per tile = four waves per SIMD needs <= 128 VGPRs, and nothing may spill.  (7: an odd stage count
    of the default plans, 2^21 = 8 + 6 + 7; 11 and 12: the round-5 passes behind H2_NTT_MAXR, which open with a radix-2 round when odd.)"""
    for field in (0, 1):
        res, blocks = _kernel(listings, "ntt.hip", f"ntt_pass9<{field}, {r}, {first}>")
        assert res["NumVgprs"] <= 128 and res["ScratchSize"] == 0 and res["Occupancy"] >= 4, (field, res)
        total = collections.Counter()
        for _, _, c in blocks:
            total.update(c)
        assert not any(op.startswith("scratch_") for op in total)
        # every radix-4 round is four products: a 10-stage pass carries 5 x 504 multiply-adds on its plain path (the store-factor and
        # load-factor branches add theirs on top, statically)
        assert total["v_mad_i64_i32"] >= (r // 2) * 504 - (378 if first == "true" else 0), total["v_mad_i64_i32"]
