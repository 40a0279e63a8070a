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


def test_shim_sources_call_only_methods_the_binding_has():
    """rust/src/dft_gpu.rs (the bodies that replace the CPU loops) and rust/benches/ntt.rs are unbuilt
    here; at least every `Plan` method and `ffi::` item they use must exist in rust/src/ffi.rs, and the
    bench must name the crate as Cargo.toml does."""
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    plan_impl = ffi[ffi.index("impl Plan {"):]
    plan_methods = set(re.findall(r"pub fn (\w+)\s*\(", plan_impl))
    ffi_items = set(re.findall(r"pub (?:const|fn|struct) (\w+)", ffi))
    shim = open(os.path.join(ROOT, "rust", "src", "dft_gpu.rs")).read()
    shim = re.sub(r"//[^\n]*", "", shim)
    used_methods = set(re.findall(r"\.expect\(\"[^\"]*\"\)\s*\.(\w+)\(", shim)) | set(re.findall(r"ffi::Plan::(\w+)\(", shim))
    assert used_methods >= {"create", "forward", "backward", "forward_lazy", "backward_lazy"}
    assert used_methods <= plan_methods, used_methods - plan_methods
    used_items = set(re.findall(r"ffi::(\w+)", shim))
    assert used_items <= ffi_items | {"Plan"}, used_items - ffi_items
    # one forward and one backward core per table type, each ignoring LAZY
    assert len(re.findall(r"pub fn forward_inplace_core<const LAZY: bool>", shim)) == 3
    assert len(re.findall(r"pub fn backward_inplace_core<const LAZY: bool>", shim)) == 3
    bench = open(os.path.join(ROOT, "rust", "benches", "ntt.rs")).read()
    assert "use ntt::dft::ntt::Table;" in bench and "app::" not in re.sub(r"//[^\n]*", "", bench)
