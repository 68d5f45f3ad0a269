#!/usr/bin/env python3
"""Generates the Rust `extern "C"` mirror of include/rsv.h (the `rsv-sys` crate of INTEGRATION.md §1).

There is no Rust toolchain in the build image, so the mirror cannot be compiled here; what CAN be guaranteed is that it
never drifts from the header: this script parses rsv.h (constants, enums, structs, opaque handles, prototypes) and emits
the block; `python tools/gen_rust_ffi.py --write` rewrites the marked region of INTEGRATION.md, and
tests/test_abi.py::test_rust_mirror_matches_header fails when the committed block differs from what the header gives.

The header is deliberately plain C (no macros in declarations, no function pointers, no unions, no bit-fields), which is
what makes a small declaration parser sufficient; anything it does not understand is an error, not a skipped line."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rsv.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN = "<!-- BEGIN GENERATED: tools/gen_rust_ffi.py from include/rsv.h -->"
END = "<!-- END GENERATED -->"

SCALARS = {"int": "c_int", "size_t": "usize", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "long long": "i64",
           "float": "f32", "char": "c_char", "void": "c_void"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def section_titles(text):
    """`/* ---- a3: Poseidon2 ... ----` banners -> (offset, title): carried into the Rust block as comments."""
    out = []
    for m in re.finditer(r"/\* ---- (.*?)(?: -+)?\s*\n", text):
        out.append((m.start(), m.group(1).strip(" -")))
    return out


def rust_type(ctype, structs):
    """C type (without the declarator name) -> Rust."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split(" ")
    # base type: leading qualifiers + (struct)? name
    i, const_base = 0, False
    if toks[i] == "const":
        const_base, i = True, i + 1
    if toks[i] == "struct":
        i += 1
    if toks[i] == "long" and i + 1 < len(toks) and toks[i + 1] == "long":
        base, i = "long long", i + 2
    else:
        base, i = toks[i], i + 1
    if base in SCALARS:
        r = SCALARS[base]
    elif base in structs:
        r = base
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    const_next = const_base
    rest = toks[i:]
    k = 0
    while k < len(rest):
        if rest[k] != "*":
            raise ValueError(f"cannot parse {ctype!r}")
        r = ("*const " if const_next else "*mut ") + r
        const_next = False
        k += 1
        if k < len(rest) and rest[k] == "const":
            const_next = True   # `T* const*`: the NEXT pointer level points at a const pointer
            k += 1
    if r == "c_void":
        return None  # plain `void`
    return r


def split_decl(decl):
    """'const uint32_t* in16' -> ('const uint32_t*', 'in16'); 'uint32_t value[4]' -> ('uint32_t', 'value', 4)."""
    decl = " ".join(decl.split())
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\d+)\s*\])?$", decl)
    if not m:
        raise ValueError(f"cannot split {decl!r}")
    return m.group(1).strip(), m.group(2), int(m.group(4)) if m.group(4) else None


def parse_header(path=HEADER):
    raw = open(path).read()
    titles = section_titles(raw)
    # keep offsets comparable: blank out comments without changing length
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), raw, flags=re.S)
    items = []  # (offset, kind, payload)
    structs = set()
    for m in re.finditer(r"^#define\s+(RSV_[A-Z0-9_]+)\s+(\S+)\s*$", text, flags=re.M):
        if m.group(1) == "RSV_H_":
            continue
        items.append((m.start(), "define", (m.group(1), m.group(2))))
    for m in re.finditer(r"typedef\s+struct\s+([a-z_0-9]+)\s+([a-z_0-9]+)\s*;", text):
        structs.add(m.group(2))
        items.append((m.start(), "opaque", m.group(2)))
    for m in re.finditer(r"typedef\s+struct\s*([a-z_0-9]*)\s*\{(.*?)\}\s*([a-z_0-9]+)\s*;", text, flags=re.S):
        structs.add(m.group(3))
        items.append((m.start(), "struct", (m.group(3), m.group(2))))
    for m in re.finditer(r"(typedef\s+)?enum\s*([a-z_0-9]*)\s*\{(.*?)\}\s*([a-z_0-9]*)\s*;", text, flags=re.S):
        items.append((m.start(), "enum", (m.group(4) or m.group(2), m.group(3))))
    body = text
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(rsv_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.M | re.S):
        items.append((m.start(), "fn", (m.group(1).strip(), m.group(2), m.group(3))))
    items.sort(key=lambda it: it[0])
    return items, structs, titles


def emit(items, structs, titles):
    out = ["#![allow(non_camel_case_types)]", "use std::os::raw::{c_char, c_int, c_void};", ""]
    consts, types, fns = [], [], []
    for off, kind, p in items:
        if kind == "define":
            name, val = p
            v = val.rstrip("u")
            ty = "u32" if name == "RSV_M31_P" else ("c_int" if name == "RSV_ABI_VERSION" else "usize")
            consts.append(f"pub const {name}: {ty} = {v};")
        elif kind == "enum":
            name, body = p
            cur = -1
            lines = []
            for ent in [e.strip() for e in body.split(",") if e.strip()]:
                if "=" in ent:
                    k, v = [x.strip() for x in ent.split("=")]
                    cur = int(v, 0)
                else:
                    k, cur = ent, cur + 1
                lines.append(f"pub const {k}: c_int = {cur};")
            consts.append(f"// enum {name or '(anonymous)'}")
            consts.extend(lines)
        elif kind == "opaque":
            types.append(f"#[repr(C)] pub struct {p} {{ _private: [u8; 0] }}")
        elif kind == "struct":
            name, body = p
            fields = []
            for decl in [d.strip() for d in body.split(";") if d.strip()]:
                # `uint32_t a, b, c` -> three fields
                first_type = None
                for part in [x.strip() for x in decl.split(",")]:
                    if first_type is None:
                        ctype, fname, arr = split_decl(part)
                        first_type = ctype
                    else:
                        m = re.match(r"^(\**)\s*([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\d+)\s*\])?$", part)
                        ctype, fname, arr = first_type.rstrip("* ") + m.group(1), m.group(2), int(m.group(4)) if m.group(4) else None
                    rt = rust_type(ctype, structs)
                    fields.append(f"pub {fname}: " + (f"[{rt}; {arr}]" if arr else rt))
            types.append(f"#[repr(C)] #[derive(Clone, Copy)]\npub struct {name} {{ " + ", ".join(fields) + " }")
        elif kind == "fn":
            ret, name, args = p
            title = None
            for toff, t in titles:
                if toff < off:
                    title = t
            params = []
            if args.strip() and args.strip() != "void":
                for a in [x.strip() for x in args.split(",")]:
                    ctype, pname, arr = split_decl(a)
                    if arr:
                        raise ValueError(f"array parameter in {name}")
                    params.append(f"{pname}: {rust_type(ctype, structs)}")
            rret = rust_type(ret, structs)
            fns.append((title, f"    pub fn {name}(" + ", ".join(params) + ")" + (f" -> {rret}" if rret else "") + ";"))
    out += consts + [""] + types + ["", '#[link(name = "rsv_hip")]', 'extern "C" {']
    last = None
    for title, line in fns:
        if title != last and title:
            out.append(f"    // ---- {title}")
            last = title
        out.append(line)
    out.append("}")
    return "\n".join(out) + "\n"


def wrap(line, width=118):
    """break a long declaration at commas, continuation lines indented behind the opening parenthesis / brace"""
    if len(line) <= width:
        return line
    open_at = min([i for i in (line.find("("), line.find("{")) if i >= 0])
    indent = " " * min(open_at + 1 + (1 if line[open_at] == "{" else 0), 40)
    out, cur = [], ""
    for part in line.split(", "):
        cand = part if not cur else cur + ", " + part
        if len(cand) > width and cur:
            out.append(cur + ",")
            cur = indent + part
        else:
            cur = cand
    out.append(cur)
    return "\n".join(out)


def generate():
    items, structs, titles = parse_header()
    text = emit(items, structs, titles)
    return "\n".join(wrap(l) for l in text.split("\n"))


def function_names():
    items, _, _ = parse_header()
    return sorted(p[1] for _, kind, p in items if kind == "fn")


def committed_block():
    doc = open(DOC).read()
    a, b = doc.index(BEGIN), doc.index(END)
    block = doc[a + len(BEGIN):b]
    m = re.search(r"```rust\n(.*?)```", block, flags=re.S)
    return m.group(1)


def main():
    text = generate()
    if "--write" in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN), doc.index(END)
        doc = doc[:a + len(BEGIN)] + "\n```rust\n" + text + "```\n" + doc[b:]
        open(DOC, "w").write(doc)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
