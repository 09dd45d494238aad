"""INTEGRATION.md section 1 shows the `extern "C"` block a maintainer of the reference would add (a `-sys` crate; the reference is Rust and
this image has no cargo to compile it).  Checked here against include/halo2_mi355x.h instead: every function of the block exists in the
header with the same argument count, the same argument types position by position (c_int / c_uint / usize / *const u64 / *mut u64 /
h2_bases_t ...), the same return type; every constant of the block equals the header's #define; and the library exports each of them."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST_TO_C = {"c_int": "int", "c_uint": "unsigned", "usize": "size_t", "*const u64": "const uint64_t *", "*mut u64": "uint64_t *",
             "h2_bases_t": "h2_bases_t", "*mut h2_bases_t": "h2_bases_t *", "*const c_char": "const char *", "*const c_void": "const void *",
             "*mut c_void": "void *", "*const u8": "const uint8_t *", "*mut u8": "uint8_t *",
             "WritePoint": "h2_ipa_write_point_fn", "Squeeze": "h2_ipa_squeeze_fn", "*const h2_bases_t": "const h2_bases_t *",
             "*const c_int": "const int *", "*const *const u64": "const uint64_t *const *", "*const *mut u64": "uint64_t *const *",
             "*const *const c_void": "const void *const *", "*const *mut c_void": "void *const *", "*mut usize": "size_t *", "*mut c_int": "int *"}


def _norm_c(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", " *", t)
    return t.replace("unsigned int", "unsigned")


def _header():
    src = open(os.path.join(ROOT, "include", "halo2_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^([A-Za-z_][\w \*]*?)\b(h2_\w+)\s*\(([^;{]*?)\)\s*;", src, re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")]
        types = []
        for a in params:
            is_array = bool(re.search(r"\[\d*\]$", a))               # uint8_t id[128] -> pointer
            a = re.sub(r"\[\d*\]$", "", a)
            mm = re.match(r"(.*?)(\w+)$", a.strip())
            ty = mm.group(1) if mm and mm.group(1).strip() else a
            if is_array and "*" not in ty:
                ty += "*"
            types.append(_norm_c(ty))
        protos[name] = (_norm_c(ret), types)
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define (H2_\w+) (\d+|0x[0-9a-fA-F]+)\b", src)}
    return protos, defines


def test_rust_extern_block_matches_the_header():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```rust\n// halo2_mi355x-sys/src/lib.rs(.*?)```", md, re.S).group(1)
    protos, defines = _header()
    consts = re.findall(r"pub const (H2_\w+): c_int = (\d+);", block)
    assert len(consts) >= 10
    for name, value in consts:
        assert defines[name] == int(value), name
    fns = re.findall(r"pub fn (h2_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, re.S)
    assert len(fns) >= 12
    lib = None
    so = os.path.join(ROOT, "halo2_amd", "libhalo2_mi355x.so")
    if os.path.exists(so):
        try:
            lib = ctypes.CDLL(so)
        except OSError:
            lib = None
    for name, args, ret in fns:
        assert name in protos, f"{name} is not declared in include/halo2_mi355x.h"
        c_ret, c_types = protos[name]
        rust_types = [a.split(":", 1)[1].strip() for a in re.sub(r"\s+", " ", args).split(",") if a.strip()]
        assert len(rust_types) == len(c_types), (name, rust_types, c_types)
        for rt, ct in zip(rust_types, c_types):
            assert _norm_c(RUST_TO_C[rt]) == ct, (name, rt, ct)
        assert _norm_c(RUST_TO_C[(ret or "").strip()]) == c_ret, (name, ret, c_ret)
        if lib is not None:
            assert hasattr(lib, name), f"{name} is not exported by the library"


def test_every_rust_block_of_the_document_matches_the_header():
    """The later sections show more `extern "C"` items beside the edited function bodies (the opening argument as one call, its round
    loop, the collapsed generators, several GPUs ...): every `pub fn h2_*` of EVERY rust block is held to the header the same way --
    argument count, types position by position (the two callback types by their typedef), return type -- and must be exported."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    protos, _ = _header()
    so = os.path.join(ROOT, "halo2_amd", "libhalo2_mi355x.so")
    lib = ctypes.CDLL(so) if os.path.exists(so) else None
    seen = set()
    for block in re.findall(r"```rust\n(.*?)```", md, re.S):
        for name, args, ret in re.findall(r"pub fn (h2_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, re.S):
            assert name in protos, f"{name} is not declared in include/halo2_mi355x.h"
            c_ret, c_types = protos[name]
            rust_types = [a.split(":", 1)[1].strip() for a in re.sub(r"\s+", " ", args).split(",") if a.strip()]
            assert len(rust_types) == len(c_types), (name, rust_types, c_types)
            for rt, ct in zip(rust_types, c_types):
                assert _norm_c(RUST_TO_C[rt]) == ct, (name, rt, ct)
            assert _norm_c(RUST_TO_C[(ret or "").strip()]) == c_ret, (name, ret, c_ret)
            if lib is not None:
                assert hasattr(lib, name), f"{name} is not exported by the library"
            seen.add(name)
    assert {"h2_open", "h2_open_device", "h2_ipa_rounds_device", "h2_msm", "h2_commit", "h2_ntt"} <= seen
