#!/usr/bin/env python3
"""include/bevy_terrain_amd.h -> integration/hip.rs: the complete Rust binding a maintainer of bevy_terrain would add as `src/hip.rs`
(the reference has no FFI: src/lib.rs:60-86 re-exports plain Rust modules; INTEGRATION.md shows where the calls go).

Every `extern "C"` prototype, every `#[repr(C)]` struct (fields in declaration order, arrays as `[T; N]`), every opaque handle, every
`#define` / enumerator as a `pub const`.  The header is the library's own and regular (one declaration per statement, no function
pointers, no bit fields, no unions), so a small declaration parser is enough; anything it does not understand is an error, not a guess.

    python tools/gen_rust_ffi.py            # writes integration/hip.rs
    python tools/gen_rust_ffi.py --check    # exits 1 when the committed file is stale

tests/test_host_logic.py parses the generated file back and compares symbol set, argument counts, field order and struct sizes with
the header, the ctypes mirror and the layouts the C99 consumer (tests/abi_consumer.c) prints."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bevy_terrain_amd.h")
OUT = os.path.join(ROOT, "integration", "hip.rs")

SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16", "int32_t": "i32",
           "int64_t": "i64", "float": "f32", "double": "f64", "size_t": "usize", "char": "c_char", "void": "c_void", "bt_status": "bt_status"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def parse_int(tok, consts):
    tok = tok.strip()
    if tok in consts:
        return consts[tok]
    m = re.fullmatch(r"(-?)(0[xX][0-9a-fA-F]+|\d+)[uUlL]*", tok)
    if not m:
        raise ValueError(f"not an integer constant: {tok!r}")
    return (-1 if m.group(1) else 1) * int(m.group(2), 0)


def rust_type(ctype, structs, opaque):
    """C type text (without the declarator's name) -> Rust type"""
    t = ctype.strip()
    stars = 0
    while t.endswith("*"):
        stars += 1
        t = t[:-1].strip()
    const = False
    toks = [x for x in t.split() if x != "struct"]
    if "const" in toks:
        const = True
        toks = [x for x in toks if x != "const"]
    if len(toks) != 1:
        raise ValueError(f"type not understood: {ctype!r}")
    base = toks[0]
    if base in SCALARS:
        r = SCALARS[base]
    elif base in structs or base in opaque:
        r = base
    else:
        raise ValueError(f"unknown type {base!r} in {ctype!r}")
    for k in range(stars):
        # `const T*` is a pointer to const; further levels (T**) are mutable pointers to the pointer
        r = ("*const " if (const and k == 0) else "*mut ") + r
    if stars == 0 and r == "c_void":
        return None  # a bare `void` (return type)
    return r


def split_declarators(decl):
    """`uint32_t a, b[3], c` -> (type text, [(name, [dims])...])"""
    decl = decl.strip()
    m = re.match(r"^(.*?[\s\*])([A-Za-z_]\w*(?:\s*\[[^\]]*\])*(?:\s*,\s*[A-Za-z_]\w*(?:\s*\[[^\]]*\])*)*)$", decl, flags=re.S)
    if not m:
        raise ValueError(f"declaration not understood: {decl!r}")
    ctype, names = m.group(1), m.group(2)
    out = []
    for piece in names.split(","):
        piece = piece.strip()
        name = re.match(r"[A-Za-z_]\w*", piece).group(0)
        dims = re.findall(r"\[([^\]]*)\]", piece)
        out.append((name, dims))
    return ctype, out


def parse_header(text):
    text = strip_comments(text)
    consts, const_order = {}, []   # name -> int ; order with a kind
    for m in re.finditer(r"^[ \t]*#define[ \t]+(BT_[A-Z0-9_]+)[ \t]+([^\n]+)$", text, flags=re.M):
        name, val = m.group(1), m.group(2).strip()
        try:
            consts[name] = parse_int(val, consts)
            const_order.append((name, "define"))
        except ValueError:
            pass  # not a numeric macro
    body = re.sub(r"^[ \t]*#[^\n]*$", "", text, flags=re.M)          # preprocessor lines
    body = re.sub(r'extern\s+"C"\s*\{', "", body)
    # enums
    for m in re.finditer(r"\benum\s*\{(.*?)\}\s*;", body, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [x.strip() for x in item.split("=", 1)]
                nxt = parse_int(val, consts)
            else:
                name = item
            consts[name] = nxt
            const_order.append((name, "enum"))
            nxt += 1
    body = re.sub(r"\benum\s*\{.*?\}\s*;", "", body, flags=re.S)
    # opaque handles and plain typedefs
    opaque = re.findall(r"\btypedef\s+struct\s+(bt_\w+)\s+\1\s*;", body)
    body = re.sub(r"\btypedef\s+struct\s+(bt_\w+)\s+\1\s*;", "", body)
    typedefs = re.findall(r"\btypedef\s+(\w+)\s+(bt_\w+)\s*;", body)
    body = re.sub(r"\btypedef\s+\w+\s+bt_\w+\s*;", "", body)
    # structs (in declaration order: a later struct may embed an earlier one)
    structs = {}
    for m in re.finditer(r"\btypedef\s+struct\s+(bt_\w+)\s*\{(.*?)\}\s*(bt_\w+)\s*;", body, flags=re.S):
        name, fields_text, alias = m.group(1), m.group(2), m.group(3)
        if name != alias:
            raise ValueError(f"struct tag {name} != typedef name {alias}")
        fields = []
        for stmt in fields_text.split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            ctype, decls = split_declarators(stmt)
            for fname, dims in decls:
                fields.append((fname, ctype.strip(), [parse_int(d, consts) for d in dims]))
        structs[name] = fields
    body = re.sub(r"\btypedef\s+struct\s+bt_\w+\s*\{.*?\}\s*bt_\w+\s*;", "", body, flags=re.S)
    # functions: what is left are prototypes `ret name(args);`
    functions = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split()).strip()
        if not stmt or stmt in ("}",):
            continue
        m = re.match(r"^(.*?[\s\*])(bt_[a-z0-9_]+)\s*\((.*)\)$", stmt)
        if not m:
            if stmt.strip("} ") == "":
                continue
            raise ValueError(f"statement not understood: {stmt!r}")
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                ctype, decls = split_declarators(a.strip())
                (pname, dims), = decls
                params.append((pname, ctype.strip() + ("*" * len(dims))))  # an array parameter is a pointer to its element
        functions.append((name, ret, params))
    return {"consts": consts, "const_order": const_order, "opaque": opaque, "typedefs": typedefs, "structs": structs, "functions": functions}


RUST_KEYWORDS = {"type", "box", "ref", "move", "in", "loop", "match", "fn", "self", "super", "where", "use", "mod", "pub", "impl", "trait", "as", "dyn", "async", "await", "yield", "final", "override"}


def ident(name):
    return "r#" + name if name in RUST_KEYWORDS else name


def generate(header_text):
    H = parse_header(header_text)
    structs, opaque = H["structs"], set(H["opaque"])
    out = []
    w = out.append
    w("// GENERATED by tools/gen_rust_ffi.py from include/bevy_terrain_amd.h — do not edit; regenerate after every header change")
    w("// (tests/test_host_logic.py fails when this file is stale).  The complete binding of libbevy_terrain_amd.so: it would live")
    w("// at `src/hip.rs` of bevy_terrain; INTEGRATION.md shows where src/preprocess/, src/terrain_data/ and src/render/ call into it.")
    w(f"// {len(H['functions'])} functions, {len(structs)} #[repr(C)] structs, {len(opaque)} opaque handles, {len(H['const_order'])} constants, ABI version {H['consts']['BT_ABI_VERSION']}.")
    w("#![allow(non_camel_case_types, non_snake_case, dead_code)]")
    w("")
    w("use core::ffi::{c_char, c_void};")
    w("")
    for alias_of, alias in H["typedefs"]:
        w(f"pub type {alias} = {SCALARS[alias_of]};")
    w("")
    status_names = {n for n, _ in H["const_order"] if n == "BT_OK" or n.startswith("BT_ERR_")}
    for name, kind in H["const_order"]:
        v = H["consts"][name]
        if name in status_names:
            w(f"pub const {name}: bt_status = {v};")
        elif v < 0:
            w(f"pub const {name}: i32 = {v};")
        elif v > 0xFFFFFFFF:
            w(f"pub const {name}: u64 = {v};")
        else:
            w(f"pub const {name}: u32 = {v};" if v < 0x10000 else f"pub const {name}: u32 = {v:#x};")
    w("")
    for name in H["opaque"]:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        w("    _private: [u8; 0],")
        w("}")
    w("")
    for name, fields in structs.items():
        w("#[repr(C)]")
        w("#[derive(Clone, Copy)]")
        w(f"pub struct {name} {{")
        for fname, ctype, dims in fields:
            t = rust_type(ctype, structs, opaque)
            for d in reversed(dims):
                t = f"[{t}; {d}]"
            w(f"    pub {ident(fname)}: {t},")
        w("}")
        w("")
    w('#[link(name = "bevy_terrain_amd")]')
    w('extern "C" {')
    for name, ret, params in H["functions"]:
        ps = ", ".join(f"{ident(p)}: {rust_type(t, structs, opaque)}" for p, t in params)
        r = rust_type(ret, structs, opaque)
        w(f"    pub fn {name}({ps})" + (f" -> {r};" if r else ";"))
    w("}")
    return "\n".join(out) + "\n"


def main():
    text = generate(open(HEADER).read())
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("integration/hip.rs is", "current" if ok else "STALE: run python tools/gen_rust_ffi.py")
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {os.path.relpath(OUT, ROOT)}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()
