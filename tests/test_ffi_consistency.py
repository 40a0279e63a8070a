"""include/nttx.h and rust/src/ffi.rs declare the same C ABI: every function of the header appears
in the Rust `extern "C"` block with the same number of arguments and matching integer widths and
pointer kinds, and the block declares nothing the header lacks.  (No Rust toolchain exists in the
build image, so this is the check that keeps the unbuilt binding honest.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "int": "c_int", "uint64_t": "u64", "size_t": "usize",
    "const uint64_t *": "*const u64", "uint64_t *": "*mut u64",
    "const int *": "*const c_int", "int *": "*mut c_int",
    "void *": "*mut c_void", "char *": "*mut c_char", "const char *": "*const c_char",
    "const nttx_plan *": "*const NttxPlan", "nttx_plan *": "*mut NttxPlan", "nttx_plan **": "*mut *mut NttxPlan",
    "void": None,
}


def norm_c(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*", " *", t)
    return re.sub(r"\* \*", "**", t)


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nttx.h")).read(), flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"^([A-Za-z_][\w \t\*]*?)\s*\b(nttx_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        params = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                m = re.match(r"^(.*?)(\w+)$", a.strip())           # type, then the parameter name
                params.append((norm_c(m.group(1)), m.group(2)))
        out[name] = (norm_c(ret), params)
    return out


def rust_functions():
    text = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    block = re.sub(r"//[^\n]*", "", block)
    out = {}
    for name, args, ret in re.findall(r"pub fn (nttx_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            pname, ptype = a.split(":", 1)
            params.append((re.sub(r"\s+", " ", ptype.strip()), pname.strip()))
        out[name] = (ret.strip() if ret else None, params)
    return out


def test_every_header_function_is_bound_with_matching_types():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 28
    assert sorted(h) == sorted(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name, (cret, cparams) in h.items():
        rret, rparams = r[name]
        assert C_TO_RUST[cret] == rret, (name, cret, rret)
        assert len(cparams) == len(rparams), name
        for (ctype, cname), (rtype, rname) in zip(cparams, rparams):
            assert C_TO_RUST[ctype] == rtype, (name, cname, ctype, rtype)
            assert cname == rname, (name, cname, rname)


def test_ctypes_binding_lists_the_same_symbols():
    from ntt_amd import _lib
    assert sorted(_lib.SYMBOLS) == sorted(header_functions())
