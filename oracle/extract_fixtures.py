#!/usr/bin/env python3
"""Extract the reference's own known-answer fixtures for the hot path into tests/golden/.

Reads /root/reference (present only in the build container), writes small JSON
files that travel with the repo.  Every record carries the reference file:line
it came from.  Run:  python oracle/extract_fixtures.py

TEST INFRASTRUCTURE ONLY - nothing under halo2_amd/ uses this.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def lineno(text, pos):
    return text.count("\n", 0, pos) + 1


def parse_from_raw_block(text, start, end):
    """All `from_raw([a,b,c,d])` 4x64 little-endian limb constants in text[start:end]."""
    vals = []
    for mt in re.finditer(r"from_raw\(\[\s*([^\]]+?)\]\)", text[start:end], re.S):
        limbs = [int(x.strip().replace("_", ""), 16) for x in mt.group(1).split(",") if x.strip()]
        assert len(limbs) == 4
        vals.append(sum(l << (64 * i) for i, l in enumerate(limbs)))
    return vals


def poseidon_constants(name):
    path = f"halo2_poseidon/src/{name}.rs"
    text = open(os.path.join(REF, path)).read()
    i_rc = text.index("const ROUND_CONSTANTS")
    i_mds = text.index("const MDS:")
    i_inv = text.index("const MDS_INV")
    rc = parse_from_raw_block(text, i_rc, i_mds)
    mds = parse_from_raw_block(text, i_mds, i_inv)
    assert len(rc) == 192 and len(mds) == 9
    return {
        "source": f"{path}:{lineno(text, i_rc)} (ROUND_CONSTANTS), :{lineno(text, i_mds)} (MDS)",
        "round_constants": [[hex(v) for v in rc[3 * i:3 * i + 3]] for i in range(64)],
        "mds": [[hex(v) for v in mds[3 * i:3 * i + 3]] for i in range(3)],
    }


def poseidon_vectors(mod):
    path = "halo2_poseidon/src/test_vectors.rs"
    text = open(os.path.join(REF, path)).read()
    i_mod = text.index(f"pub mod {mod} ")
    i_perm = text.index("pub fn permute()", i_mod)
    i_hash = text.index("pub fn hash()", i_perm)
    body = text[i_perm:i_hash]
    arrays = re.findall(r"\[\s*((?:0x[0-9a-fA-F]{2},\s*){31}0x[0-9a-fA-F]{2},?\s*)\]", body)
    vals = []
    for a in arrays:
        bs = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", a))
        assert len(bs) == 32
        vals.append(int.from_bytes(bs, "little"))
    assert len(vals) % 6 == 0 and vals
    vecs = []
    for i in range(0, len(vals), 6):
        vecs.append({"initial_state": [hex(v) for v in vals[i:i + 3]],
                     "final_state": [hex(v) for v in vals[i + 3:i + 6]]})
    return {"source": f"{path}:{lineno(text, i_perm)}", "permute": vecs}


def pinned_vk(path):
    text = open(os.path.join(REF, path)).read()
    rec = {"source": path}
    m = re.search(r"base_modulus: \"(0x[0-9a-f]+)\"", text)
    rec["base_modulus"] = m.group(1)
    rec["base_modulus_line"] = lineno(text, m.start())
    m = re.search(r"scalar_modulus: \"(0x[0-9a-f]+)\"", text)
    rec["scalar_modulus"] = m.group(1)
    m = re.search(r"k: (\d+),\s*extended_k: (\d+),\s*omega: (0x[0-9a-f]+)", text)
    rec["k"], rec["extended_k"], rec["omega"] = int(m.group(1)), int(m.group(2)), m.group(3)
    rec["omega_line"] = lineno(text, m.start(3))
    pts = []
    for mt in re.finditer(r"\((0x[0-9a-f]{64}), (0x[0-9a-f]{64})\)", text):
        pts.append({"x": mt.group(1), "y": mt.group(2), "line": lineno(text, mt.start())})
    rec["points"] = pts
    return rec


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are already committed under tests/golden/")
    os.makedirs(OUT, exist_ok=True)
    pose = {}
    for name, mod in (("fp", "fp"), ("fq", "fq")):
        rec = poseidon_constants(name)
        rec.update(poseidon_vectors(mod))
        pose[name] = rec
    json.dump(pose, open(os.path.join(OUT, "poseidon_kat.json"), "w"), indent=0)

    vks = [pinned_vk("halo2_proofs/tests/plonk_api.rs")]
    d = "halo2_gadgets/src/test_circuits/circuit_data"
    for f in sorted(os.listdir(os.path.join(REF, d))):
        if f.startswith("vk_") and f.endswith(".rdata"):
            vks.append(pinned_vk(f"{d}/{f}"))
    json.dump(vks, open(os.path.join(OUT, "pinned_vk.json"), "w"), indent=0)
    # the whole pretty-printed pinned verifying key (tests/plonk_api.rs:585-984: its compact Debug form is what
    # VerifyingKey::transcript_repr hashes, plonk.rs:75-89) and the stored proof the reference's test verifies against it
    text = open(os.path.join(REF, "halo2_proofs/tests/plonk_api.rs")).read()
    a, b = text.index('r#####"') + 7, text.index('"#####')
    open(os.path.join(OUT, "plonk_api_pinned_vk.txt"), "w").write(text[a:b])
    proof = open(os.path.join(REF, "halo2_proofs/tests/plonk_api_proof.bin"), "rb").read()
    open(os.path.join(OUT, "plonk_api_proof.bin"), "wb").write(proof)
    npts = sum(len(v["points"]) for v in vks)
    print(f"poseidon: {len(pose['fp']['permute'])}+{len(pose['fq']['permute'])} permute vectors; "
          f"{len(vks)} pinned VKs with {npts} Vesta points")


if __name__ == "__main__":
    main()
