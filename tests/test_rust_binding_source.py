"""The Rust binding crate under bindings/rust cannot be compiled where this library is
built (no Rust toolchain), so this test keeps its source in step with the C header by
reading both: every struct with the same fields in the same order and of the matching
type, every entry point with the same parameters, every constant with the same value."""
import os
import re

from pathtracing_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pathtracing_amd.h")
CRATE = os.path.join(ROOT, "bindings", "rust", "pathtracing-amd-sys", "src", "lib.rs")

SCALARS = {"float": "f32", "double": "f64", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "size_t": "usize",
           "int": "c_int", "char": "c_char", "void": "c_void", "PtStatus": "PtStatus"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _rust_type(c_type, stars, dims):
    c_type = c_type.strip()
    const = c_type.startswith("const ")
    base = c_type[6:].strip() if const else c_type
    t = SCALARS.get(base, base)
    for _ in range(stars):
        t = ("*const " if const else "*mut ") + t
        const = False  # only the innermost level is const in this header
    for d in reversed(dims):
        t = f"[{t}; {d}]"
    return t


def _c_declarators(decl):
    """'const float* positions' / 'float x, y' / 'float m[4][4]' / 'PtScene** out' -> [(name, rust type)]"""
    decl = " ".join(decl.split())
    m = re.match(r"((?:const )?\w+)\s*(.*)", decl)
    c_type, rest = m.group(1), m.group(2)
    out = []
    for item in rest.split(","):
        item = item.strip()
        stars = item.count("*")
        item = item.replace("*", "").strip()
        dims = re.findall(r"\[(\d+)\]", item)
        name = re.sub(r"\[.*", "", item)
        out.append((name, _rust_type(c_type, stars, dims)))
    return out


def c_structs(text):
    structs = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*\1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            if decl.strip():
                fields += _c_declarators(decl)
        structs[m.group(1)] = fields
    return structs


def c_functions(text):
    funcs = {}
    for m in re.finditer(r"^(?:const )?\w+\*?\s+\*?(pt_\w+)\(([^)]*)\);", text, flags=re.M):
        params = m.group(2).strip()
        plist = []
        if params and params != "void":
            for p in params.split(","):
                plist += _c_declarators(p)
        funcs[m.group(1)] = plist
    return funcs


def rust_structs(text):
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for f in re.finditer(r"(?:pub )?(?:r#)?(\w+):\s*([^,\n]+),", m.group(2)):
            fields.append((f.group(1), f.group(2).strip()))
        structs[m.group(1)] = fields
    return structs


def rust_functions(text):
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (pt_\w+)\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        funcs[m.group(1)] = [(p.split(":")[0].strip(), p.split(":", 1)[1].strip()) for p in params]
    return funcs


def test_rust_structs_follow_the_header():
    header = _strip_comments(open(HEADER).read())
    crate = _strip_comments(open(CRATE).read())
    c, r = c_structs(header), rust_structs(crate)
    assert len(c) == 18
    for name, fields in c.items():
        assert name in r, f"struct {name} is missing from the crate"
        assert r[name] == fields, f"struct {name}: header {fields} / crate {r[name]}"
    for opaque in ("PtScene", "PtHostScene", "PtFrame", "PtWorker", "PtServer"):
        assert r[opaque] == [("_private", "[u8; 0]")]


def test_rust_functions_follow_the_header():
    header = _strip_comments(open(HEADER).read())
    crate = _strip_comments(open(CRATE).read())
    c, r = c_functions(header), rust_functions(crate)
    assert set(c) == set(_native.SYMBOLS), "header parser lost an entry point"
    assert set(r) == set(c)
    for name, params in c.items():
        assert r[name] == params, f"{name}: header {params} / crate {r[name]}"


def test_rust_constants_follow_the_header():
    header = open(HEADER).read()
    crate = open(CRATE).read()
    want = {}
    for m in re.finditer(r"#define (PT_\w+) (0x[0-9A-Fa-f]+|\d+)u\b", header):
        want[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"\b(PT_[A-Z_]+) = (\d+)", _strip_comments(header)):
        want[m.group(1)] = int(m.group(2))
    assert {"PT_ABI_VERSION", "PT_NO_INDEX", "PT_FLAG_PROFILE", "PT_ERR_UNSUPPORTED", "PT_NODE_KD_TREE", "PT_BOKEH_SQUARE"} <= set(want)
    got = {m.group(1): int(m.group(2).replace("_", ""), 0)
           for m in re.finditer(r"pub const (PT_\w+): \w+ = (0x[0-9A-Fa-f_]+|\d+);", crate)}
    assert got == want
