"""The documents carry the DRIVER's numbers (round-5 review, item 6): BASELINE.md's headline names the BENCH_rNN.json it quotes and must equal that
file's parsed value; it must be the driver's latest as of the round that wrote it (the newest file, or the one before it when the driver has since
added the current round's); and the two documents stay readable as documents (no regrowth into a log -- the log is DESIGN_LOG.md)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baseline_headline_is_the_drivers_number():
    text = open(os.path.join(ROOT, "BASELINE.md")).read()
    m = re.search(r"\*\*Headline \(the driver's own run, `BENCH_r(\d+)\.json`[^)]*\): ([0-9.]+) M scalar-mults/s\*\*", text)
    assert m, "BASELINE.md section 4 must open with the driver's headline and the BENCH file it comes from"
    named, quoted = int(m.group(1)), float(m.group(2))
    rounds = sorted(int(re.search(r"BENCH_r(\d+)\.json", p).group(1)) for p in glob.glob(os.path.join(ROOT, "BENCH_r*.json")))
    assert rounds and named in rounds[-2:], (named, rounds)          # the latest when written; at most one round has been added since
    value = json.load(open(os.path.join(ROOT, f"BENCH_r{named:02d}.json")))["parsed"]["value"]
    assert abs(quoted - value) / value <= 0.005, (quoted, value)
    # the same number in the table's driver column and in the README
    assert f"**{quoted}**" in text or f"**{value:.1f}**" in text
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert f"{quoted}" in readme, "README.md quotes the driver's headline too"


def test_documents_stay_documents():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 45 * 1024
    assert os.path.getsize(os.path.join(ROOT, "BASELINE.md")) <= 16 * 1024
    base = open(os.path.join(ROOT, "BASELINE.md")).read()
    sec4 = base[base.index("## 4. Results"):]
    assert sec4.count("\n| Workload") == 1, "ONE results table"
    prose = [l for l in sec4.splitlines() if l.strip() and not l.startswith("|") and not l.startswith("#")]
    assert len(prose) <= 32, len(prose)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "## 0. Round 6 at a glance" in design and "Round 5 at a glance" not in design.split("## 1.")[0]
