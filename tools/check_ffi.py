#!/usr/bin/env python3
"""Keeps provers/hip/driver/src/ffi.rs (the Rust `extern "C"` declarations of the provers/hip crate)
in step with include/raiko_hip.h.

  python tools/check_ffi.py           compare: every struct field, enum value, function name,
                                      argument type and return type of the header against ffi.rs
  python tools/check_ffi.py --emit    print the ffi.rs body generated from the header

The C header is the source of truth; ffi.rs is its mechanical translation (plus comments), so
the check is an exact comparison of the normalised declarations.  Runs in the CPU test suite
(tests/test_ffi_crate.py): the Rust toolchain is absent from this image, this is what stands in
for `cargo check` on the FFI boundary."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raiko_hip.h")
FFI_RS = os.path.join(ROOT, "provers", "hip", "driver", "src", "ffi.rs")

SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "float": "f32",
           "double": "f64", "char": "c_char", "void": "c_void"}


OPAQUE = set()   # `typedef struct X X;` handles, filled by parse_header


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_type(ctype, structs_enums, fn_types):
    """C type (declarator removed) -> Rust type"""
    t = " ".join(ctype.split())
    t = t.replace("unsigned int", "uint32_t")
    m = re.match(r"^(const )?([A-Za-z_0-9]+)( const)?((?: ?\*(?: const)?)*)$", t)
    if not m:
        raise ValueError("cannot translate C type %r" % ctype)
    const, base, _, stars = m.group(1), m.group(2), m.group(3), m.group(4)
    if base in fn_types and not stars:
        return "Option<%s>" % base
    if base in SCALARS:
        rbase = SCALARS[base]
    elif base in structs_enums or base in OPAQUE:
        rbase = base
    else:
        raise ValueError("unknown C type %r" % base)
    ptrs = re.findall(r"\*( const)?", stars)
    out = rbase
    # innermost pointer's constness comes from the leading `const`; outer levels from `* const`
    for level, pc in enumerate(ptrs):
        is_const = bool(const) if level == 0 else bool(ptrs[level - 1])
        out = ("*const " if is_const else "*mut ") + out
    if not ptrs and rbase == "c_void":
        return "()"
    return out


def split_decl(decl):
    """'const uint32_t* h_seals[3]' -> (type, name, array_len or None)"""
    decl = " ".join(decl.split())
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)\s*(\[(\w+)\])?$", decl)
    if not m:
        raise ValueError("cannot split %r" % decl)
    return m.group(1).strip(), m.group(2), m.group(4)


def parse_header(src):
    src = strip_comments(src)
    enums, structs, fn_types, funcs = {}, {}, {}, []
    OPAQUE.clear()
    OPAQUE.update(re.findall(r"typedef struct (\w+) \1;", src))
    # anonymous `enum { A = 0, B = 1 };` blocks: plain constants
    for k, m in enumerate(re.finditer(r"^enum \{(.*?)\};", src, flags=re.S | re.M)):
        vals, nxt = [], 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                nxt = int(v, 0)
            else:
                name = item
            vals.append((name, nxt))
            nxt += 1
        enums["__anon%d" % k] = vals
    for m in re.finditer(r"typedef enum \{(.*?)\}\s*(\w+);", src, flags=re.S):
        vals, nxt = [], 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                nxt = int(v, 0)
            else:
                name = item
            vals.append((name, nxt))
            nxt += 1
        enums[m.group(2)] = vals
    for m in re.finditer(r"typedef (\w+) \(\*(\w+)\)\((.*?)\);", src, flags=re.S):
        fn_types[m.group(2)] = (m.group(1), m.group(3))
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(\w+);", src, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        # function-pointer members: `int (*name)(args);`
        for fm in re.finditer(r"(\w+) \(\*(\w+)\)\((.*?)\);", body, flags=re.S):
            fields.append(("fnptr", fm.group(2), fm.group(1), fm.group(3), fm.start()))
        plain = re.sub(r"\w+ \(\*\w+\)\(.*?\);", lambda mm: " " * len(mm.group(0)), body, flags=re.S)
        pos = 0
        for stmt in plain.split(";"):
            start = pos + (len(stmt) - len(stmt.lstrip()))   # where the declaration itself begins
            pos += len(stmt) + 1
            stmt = stmt.strip()
            if not stmt:
                continue
            # `float a, b, c` style lists
            first_type, first_name, arr = split_decl(stmt.split(",")[0])
            fields.append(("field", first_name, first_type, arr, start))
            for extra in stmt.split(",")[1:]:
                fields.append(("field", extra.strip(), first_type, None, start + 1))
        fields.sort(key=lambda f: f[4])
        structs[name] = fields
    body = re.sub(r"typedef (enum|struct) \{.*?\}\s*\w+;", "", src, flags=re.S)
    body = re.sub(r"^enum \{.*?\};", "", body, flags=re.S | re.M)
    body = re.sub(r"typedef .*?;", "", body, flags=re.S)
    for m in re.finditer(r"^([A-Za-z_][A-Za-z_0-9 \*]*?)\b(rk_[a-z0-9_]+)\s*\((.*?)\);", body, flags=re.S | re.M):
        funcs.append((m.group(2), m.group(1).strip(), m.group(3)))
    return enums, structs, fn_types, funcs


def params_to_rust(params, known, fn_types):
    params = " ".join(params.split())
    if params in ("void", ""):
        return []
    out = []
    for p in params.split(","):
        ctype, name, arr = split_decl(p.strip())
        rt = rust_type(ctype + ("*" if arr else ""), known, fn_types)  # `T x[4]` decays to `T*`
        if name in ("pub", "type", "in", "ref", "use", "match"):
            name += "_"
        out.append((name, rt))
    return out


def emit(header_src):
    enums, structs, fn_types, funcs = parse_header(header_src)
    known = {e for e in enums if not e.startswith('__anon')} | set(structs)
    lines = []
    for name, vals in enums.items():
        tname = "c_int" if name.startswith("__anon") else name
        if tname != "c_int":
            lines.append("pub type %s = c_int;" % name)
        for vn, v in vals:
            lines.append("pub const %s: %s = %d;" % (vn, tname, v))
        lines.append("")
    for m in re.finditer(r"^#define (RK_[A-Z0-9_]+) (\d+)\s*$", strip_comments(header_src), flags=re.M):
        lines.append("pub const %s: u32 = %s;" % (m.group(1), m.group(2)))
    lines.append("")
    for name in sorted(OPAQUE):
        lines.append("#[repr(C)]\npub struct %s {\n    _private: [u8; 0],\n}\n" % name)
    for name, (ret, params) in fn_types.items():
        ps = ", ".join("%s: %s" % p for p in params_to_rust(params, known, fn_types))
        lines.append("pub type %s = unsafe extern \"C\" fn(%s) -> %s;" % (name, ps, rust_type(ret, known, fn_types)))
    lines.append("")
    for name, fields in structs.items():
        lines.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct %s {" % name)
        for f in fields:
            if f[0] == "fnptr":
                ps = ", ".join("%s: %s" % p for p in params_to_rust(f[3], known, fn_types))
                lines.append("    pub %s: Option<unsafe extern \"C\" fn(%s) -> %s>," % (f[1], ps, rust_type(f[2], known, fn_types)))
            else:
                rt = rust_type(f[2], known, fn_types)
                if f[3]:
                    rt = "[%s; %s]" % (rt, f[3])
                fname = f[1] + "_" if f[1] in ("type", "ref") else f[1]
                lines.append("    pub %s: %s," % (fname, rt))
        lines.append("}\n")
    lines.append("#[link(name = \"raiko_hip\")]\nextern \"C\" {")
    for name, ret, params in funcs:
        ps = ", ".join("%s: %s" % p for p in params_to_rust(params, known, fn_types))
        r = rust_type(ret, known, fn_types)
        lines.append("    pub fn %s(%s)%s;" % (name, ps, "" if r == "()" else " -> " + r))
    lines.append("}")
    return "\n".join(lines) + "\n"


def normalise(rs):
    rs = re.sub(r"//[^\n]*", "", rs)
    rs = re.sub(r"/\*.*?\*/", "", rs, flags=re.S)
    rs = re.sub(r"#!\[.*?\]\n", "", rs)
    rs = re.sub(r"^use [^;]+;\n", "", rs, flags=re.M)
    return " ".join(rs.split())


def main():
    src = open(HEADER).read()
    gen = emit(src)
    if "--emit" in sys.argv:
        sys.stdout.write(gen)
        return 0
    have = open(FFI_RS).read()
    a, b = normalise(gen), normalise(have)
    if a != b:
        import difflib
        ga = re.sub(r"([;{}])", r"\1\n", a).split("\n")
        gb = re.sub(r"([;{}])", r"\1\n", b).split("\n")
        sys.stdout.write("\n".join(difflib.unified_diff(ga, gb, "from raiko_hip.h", "ffi.rs", lineterm="", n=1)) + "\n")
        print("ffi.rs does not match include/raiko_hip.h")
        return 1
    enums, structs, fn_types, funcs = parse_header(src)
    print("ffi.rs matches include/raiko_hip.h: %d functions, %d structs, %d enums, %d callback types"
          % (len(funcs), len(structs), len(enums), len(fn_types)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
