"""Parsers for the two spellings of the drop-in boundary — the C header (include/rapier_hip.h) and the ```rust block of INTEGRATION.md —
and a generator that prints the Rust `#[repr(C)]` structs + `extern "C"` block from the header.

    python tools/abi_parse.py            # prints the generated Rust FFI layer (paste between the markers of INTEGRATION.md)
    python tools/abi_parse.py --check    # exit 1 when INTEGRATION.md's block and the header disagree

tests/test_abi_shim.py uses the same parsers: every struct's field list (name, scalar type, element count, offset) and every export
(name, return type, argument types) must agree between the header, the Rust block, and the numpy descriptor dtypes."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rapier_hip.h")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")

SCALARS = {"int32_t": ("i32", 4), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8), "int64_t": ("i64", 8), "float": ("f32", 4), "char": ("c_char", 1), "void": ("c_void", 0)}
# the reference-side names of the ABI's structs (what a rapier maintainer would call them)
RUST_NAME = {"rp_integration_params": "IntegrationParameters", "rp_body_desc": "BodyDesc", "rp_collider_desc": "ColliderDesc",
             "rp_collision_event": "CollisionEventRaw", "rp_contact_force_event": "ContactForceEventRaw", "rp_joint_motor": "JointMotor",
             "rp_joint_desc": "GenericJoint", "rp_counters": "Counters", "rp_world": "RpWorld", "rp_comm": "RpComm", "rp_comm_id": "CommId"}


def _strip_c_comments(txt: str) -> str:
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", txt, flags=re.S))


def parse_header(path: str = HEADER):
    """-> (structs, functions): structs = {c_name: [(field, c_type, count)]} in declaration order, functions = {name: (ret, [arg types])}
    with types spelt the Rust way ('i32', '*const f32', '*mut *mut RpWorld', ...)."""
    txt = _strip_c_comments(open(path).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, rest = decl.split(" ", 1)
            for item in rest.split(","):
                item = item.strip()
                dims = [int(d) for d in re.findall(r"\[(\d+)\]", item)]
                fname = re.sub(r"\[.*", "", item).strip()
                count = 1
                for d in dims:
                    count *= d
                fields.append((fname, ctype, count, tuple(dims)))
        structs[name] = fields
    functions = {}
    proto = re.compile(r"(?:^|\n)\s*((?:const\s+)?\w+\s*\**)\s*(rp_\w+)\s*\(([^;{]*?)\)\s*;", flags=re.S)
    for m in proto.finditer(txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        functions[name] = (_rust_type(ret), [_rust_type(a) for a in _split_args(args)])
    return structs, functions


def _split_args(args: str):
    args = " ".join(args.split())
    return [] if args in ("", "void") else [a.strip() for a in args.split(",")]


def _rust_type(c: str) -> str:
    """A C parameter / return declaration ('const float gravity[3]', 'rp_world **out', 'rp_world *const *worlds') as a Rust FFI type."""
    c = " ".join(c.replace("*", " * ").split())
    is_array = "[" in c
    c = re.sub(r"\[[^\]]*\]", "", c)
    toks = c.split()
    # drop the parameter name (an identifier that is neither a qualifier nor a type) when one is there
    if toks and toks[-1] not in ("*", "const") and len([t for t in toks if t not in ("*", "const")]) > 1:
        toks = toks[:-1]
    base = [t for t in toks if t not in ("*", "const")][0]
    rbase = RUST_NAME.get(base) or SCALARS.get(base, (base,))[0]
    # pointer levels with their pointee constness, left to right:  const T *  -> *const T ;  T *const *  -> *const *mut T
    i = toks.index(base)
    const_pending = "const" in toks[:i]
    out = rbase
    rest = toks[i + 1:]
    k = 0
    while k < len(rest):
        if rest[k] == "*":
            out = ("*const " if const_pending else "*mut ") + out
            const_pending = False
        elif rest[k] == "const":
            const_pending = True
        k += 1
    if is_array:
        out = ("*const " if const_pending or "const" in toks[:i] else "*mut ") + rbase
    return "()" if out == "c_void" else out


def struct_layout(structs, name):
    """[(field, rust scalar or struct name, count, offset, size)] with C layout rules; returns (fields, total size, alignment)."""
    off, align, out = 0, 1, []
    for fname, ctype, count, _ in structs[name]:
        if ctype in SCALARS:
            rt, sz = SCALARS[ctype]
            al = sz
        else:
            _, sz, al = struct_layout(structs, ctype)
            rt = RUST_NAME[ctype]
        off = (off + al - 1) // al * al
        out.append((fname, rt, count, off, sz * count))
        off += sz * count
        align = max(align, al)
    return out, (off + align - 1) // align * align, align


def rust_block(path: str = INTEGRATION) -> str:
    txt = open(path).read()
    m = re.search(r"```rust\n(.*?)```", txt, flags=re.S)
    if not m:
        raise ValueError("INTEGRATION.md holds no ```rust block")
    return m.group(1)


def parse_rust(block: str):
    """-> (structs, functions) of the ```rust block: #[repr(C)] structs as [(field, type, count)], extern "C" fns as (ret, [arg types])."""
    code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", block, flags=re.S))
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", code, flags=re.S):
        fields = []
        for f in re.finditer(r"pub\s+(\w+)\s*:\s*([^,}]+)", m.group(2)):
            t = " ".join(f.group(2).split())
            count = 1
            while True:
                a = re.fullmatch(r"\[(.+);\s*(\d+)\]", t)
                if not a:
                    break
                count *= int(a.group(2)); t = a.group(1).strip()
            fields.append((f.group(1), t, count))
        structs[m.group(1)] = fields
    functions = {}
    ext = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', code, flags=re.S)
    if ext:
        for m in re.finditer(r"fn\s+(rp_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext.group(1), flags=re.S):
            args = [" ".join(a.split(":", 1)[1].split()) for a in m.group(2).split(",") if ":" in a]
            functions[m.group(1)] = (" ".join((m.group(3) or "()").split()), args)
    return structs, functions


def compare(header=HEADER, integration=INTEGRATION):
    """list of human-readable differences between the header and the Rust block (empty = in agreement)"""
    hs, hf = parse_header(header)
    rs, rf = parse_rust(rust_block(integration))
    diffs = []
    for cname, fields in hs.items():
        rname = RUST_NAME[cname]
        if rname not in rs:
            diffs.append(f"struct {cname}: no #[repr(C)] {rname} in the Rust block"); continue
        want = [(f, RUST_NAME.get(t) or SCALARS[t][0], c) for f, t, c, _ in fields]
        if want != rs[rname]:
            diffs.append(f"struct {cname} / {rname}: fields differ\n  header: {want}\n  rust:   {rs[rname]}")
    for rname in rs:
        if rname not in RUST_NAME.values():
            diffs.append(f"Rust #[repr(C)] struct {rname} has no counterpart in the header")
    if set(hf) != set(rf):
        diffs.append(f"exports differ: only in the header {sorted(set(hf) - set(rf))}, only in the Rust block {sorted(set(rf) - set(hf))}")
    for name in sorted(set(hf) & set(rf)):
        if hf[name] != rf[name]:
            diffs.append(f"{name}: header {hf[name]} != rust {rf[name]}")
    return diffs


def generate() -> str:
    """the FFI layer of the Rust shim, generated from the header"""
    hs, hf = parse_header()
    lines = ["use std::os::raw::{c_char, c_void};", ""]
    for cname, fields in hs.items():
        rname = RUST_NAME[cname]
        derive = "#[repr(C)] #[derive(Clone, Copy)]"
        lines.append(f"{derive}\npub struct {rname} {{ // {cname}")
        for f, t, c, dims in fields:
            rt = RUST_NAME.get(t) or SCALARS[t][0]
            for d in reversed(dims):
                rt = f"[{rt}; {d}]"
            lines.append(f"    pub {f}: {rt},")
        lines.append("}")
    lines += ["#[repr(C)] pub struct RpWorld { _private: [u8; 0] }", "#[repr(C)] pub struct RpComm { _private: [u8; 0] }", "", '#[link(name = "rapier_hip")]', 'extern "C" {']
    txt = _strip_c_comments(open(HEADER).read())
    for name, (ret, args) in hf.items():
        m = re.search(r"\b" + name + r"\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
        names = []
        for a in _split_args(m.group(1)):
            a = re.sub(r"\[[^\]]*\]", "", a).replace("*", " ").split()
            names.append(a[-1])
        sig = ", ".join(f"{n}: {t}" for n, t in zip(names, args))
        lines.append(f"    pub fn {name}({sig})" + ("" if ret == "()" else f" -> {ret}") + ";")
    lines.append("}")
    return "\n".join(lines)


if __name__ == "__main__":
    if "--check" in sys.argv:
        d = compare()
        print("\n".join(d) if d else "INTEGRATION.md's Rust block agrees with include/rapier_hip.h")
        sys.exit(1 if d else 0)
    print(generate())
