"""risc0's word serde -- `risc0_zkvm::serde::to_vec` / `from_slice`, the framing of everything the
host hands to the guest and of the journal (reference call sites: provers/risc0/driver/src/lib.rs:71
`to_vec(&input)`, bonsai.rs:101 `to_vec(expected_output)`, bonsai.rs:157 `journal.decode()`).

RECALLED from risc0-zkvm 1.0.1 (serde/serializer.rs; the crate is not in the reference tree and
no test of the reference pins a serialised value, so this layout is unverifiable here):
a value becomes a stream of little-endian u32 words --
  bool, u8, u16, u32, char, i8..i32     one word
  u64 / i64                             two words, low first;  u128: four
  str / bytes (serialize_bytes)         length word, then the bytes packed 4 per word, zero padded
  seq / map                             length word, then the elements (map: key, value pairs)
  tuple, fixed array [T; N], struct     the fields in order, no length ([u8; 32] = 32 words)
  Option                                0, or 1 followed by the value
  enum                                  variant index word, then the variant's fields

A schema is a small tree of the constructors below; `to_vec(schema, value)` walks value and schema
together.  The GuestInput schema itself is raiko's tree of reth / alloy types
(lib/src/input.rs:28-45) and is not reproduced here: the executor that would consume those words
is outside this backend."""
import struct
from typing import Any, List, Sequence, Tuple

U8, U16, U32, BOOL = ("prim", 1), ("prim", 2), ("prim", 4), ("bool",)
U64, U128 = ("wide", 2), ("wide", 4)
STR, BYTES = ("str",), ("bytes",)


def Seq(elem):
    return ("seq", elem)


def Array(elem, n: int):
    return ("array", elem, n)


def Struct(*fields):
    """fields: (name, schema) pairs; values are dicts or objects with those attributes"""
    return ("struct", tuple(fields))


def Tup(*elems):
    return ("tuple", tuple(elems))


def Option(elem):
    return ("option", elem)


def Enum(*variants):
    """variants: (name, schema or None); values are (name, payload) pairs"""
    return ("enum", tuple(variants))


B256 = Array(U8, 32)


def _pack_bytes(out: List[int], b: bytes):
    out.append(len(b))
    pad = (-len(b)) % 4
    b = bytes(b) + b"\0" * pad
    out.extend(struct.unpack("<%dI" % (len(b) // 4), b))


def _ser(out: List[int], schema, v: Any):
    kind = schema[0]
    if kind == "prim":
        x = int(v)
        if not 0 <= x < (1 << (8 * schema[1])):
            raise ValueError("value %r out of range for a %d-byte integer" % (v, schema[1]))
        out.append(x)
    elif kind == "bool":
        out.append(1 if v else 0)
    elif kind == "wide":
        x = int(v)
        if not 0 <= x < (1 << (32 * schema[1])):
            raise ValueError("value out of range")
        for i in range(schema[1]):
            out.append((x >> (32 * i)) & 0xFFFFFFFF)
    elif kind == "str":
        _pack_bytes(out, v.encode("utf-8"))
    elif kind == "bytes":
        _pack_bytes(out, bytes(v))
    elif kind == "seq":
        out.append(len(v))
        for e in v:
            _ser(out, schema[1], e)
    elif kind == "array":
        if len(v) != schema[2]:
            raise ValueError("array length %d, expected %d" % (len(v), schema[2]))
        for e in v:
            _ser(out, schema[1], e)
    elif kind == "struct":
        for name, sub in schema[1]:
            _ser(out, sub, v[name] if isinstance(v, dict) else getattr(v, name))
    elif kind == "tuple":
        if len(v) != len(schema[1]):
            raise ValueError("tuple arity")
        for sub, e in zip(schema[1], v):
            _ser(out, sub, e)
    elif kind == "option":
        if v is None:
            out.append(0)
        else:
            out.append(1)
            _ser(out, schema[1], v)
    elif kind == "enum":
        name, payload = v
        for idx, (vn, sub) in enumerate(schema[1]):
            if vn == name:
                out.append(idx)
                if sub is not None:
                    _ser(out, sub, payload)
                return
        raise ValueError("unknown variant %r" % (name,))
    else:
        raise ValueError("bad schema node %r" % (kind,))


def to_vec(schema, value) -> List[int]:
    """`risc0_zkvm::serde::to_vec`: the u32 words of `value`"""
    out: List[int] = []
    _ser(out, schema, value)
    return out


def words_to_bytes(words: Sequence[int]) -> bytes:
    """`bytemuck::cast_slice::<u32, u8>` on a little-endian host"""
    return struct.pack("<%dI" % len(words), *words)


class _Reader:
    def __init__(self, words):
        self.w, self.pos = list(words), 0

    def word(self):
        if self.pos >= len(self.w):
            raise ValueError("unexpected end of words")
        self.pos += 1
        return self.w[self.pos - 1]


def _de(r: _Reader, schema):
    kind = schema[0]
    if kind == "prim":
        x = r.word()
        if x >= (1 << (8 * schema[1])):
            raise ValueError("word %#x does not fit a %d-byte integer" % (x, schema[1]))
        return x
    if kind == "bool":
        x = r.word()
        if x > 1:
            raise ValueError("bad bool")
        return bool(x)
    if kind == "wide":
        return sum(r.word() << (32 * i) for i in range(schema[1]))
    if kind in ("str", "bytes"):
        n = r.word()
        raw = words_to_bytes([r.word() for _ in range((n + 3) // 4)])
        if any(raw[n:]):
            raise ValueError("non-zero padding")
        return raw[:n].decode("utf-8") if kind == "str" else raw[:n]
    if kind == "seq":
        return [_de(r, schema[1]) for _ in range(r.word())]
    if kind == "array":
        return [_de(r, schema[1]) for _ in range(schema[2])]
    if kind == "struct":
        return {name: _de(r, sub) for name, sub in schema[1]}
    if kind == "tuple":
        return tuple(_de(r, sub) for sub in schema[1])
    if kind == "option":
        tag = r.word()
        if tag > 1:
            raise ValueError("bad option tag")
        return _de(r, schema[1]) if tag else None
    if kind == "enum":
        idx = r.word()
        if idx >= len(schema[1]):
            raise ValueError("bad variant index")
        name, sub = schema[1][idx]
        return (name, _de(r, sub) if sub is not None else None)
    raise ValueError("bad schema node")


def from_slice(schema, words: Sequence[int]):
    """`risc0_zkvm::serde::from_slice`; trailing words are an error"""
    r = _Reader(words)
    v = _de(r, schema)
    if r.pos != len(r.w):
        raise ValueError("trailing words")
    return v
