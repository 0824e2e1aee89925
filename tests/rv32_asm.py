"""TEST INFRASTRUCTURE: a minimal RV32IM assembler and ELF32 writer for the executor tests (the
image has no RISC-V toolchain, and the prebuilt guest ELFs inside the reference are not run).
Programs are lists of (mnemonic, operands...) with labels as strings; `assemble` resolves
branch / jump targets, `elf` wraps code + data into an executable with two PT_LOAD segments."""
import struct

REG = {n: i for i, n in enumerate(
    ["zero", "ra", "sp", "gp", "tp", "t0", "t1", "t2", "s0", "s1", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7",
     "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "t3", "t4", "t5", "t6"])}
REG.update({"x%d" % i: i for i in range(32)})

R_OPS = {"add": (0, 0), "sub": (0, 0x20), "sll": (1, 0), "slt": (2, 0), "sltu": (3, 0), "xor": (4, 0), "srl": (5, 0),
         "sra": (5, 0x20), "or": (6, 0), "and": (7, 0), "mul": (0, 1), "mulh": (1, 1), "mulhsu": (2, 1), "mulhu": (3, 1),
         "div": (4, 1), "divu": (5, 1), "rem": (6, 1), "remu": (7, 1)}
I_OPS = {"addi": 0, "slti": 2, "sltiu": 3, "xori": 4, "ori": 6, "andi": 7}
SH_OPS = {"slli": (1, 0), "srli": (5, 0), "srai": (5, 0x20)}
LOADS = {"lb": 0, "lh": 1, "lw": 2, "lbu": 4, "lhu": 5}
STORES = {"sb": 0, "sh": 1, "sw": 2}
BRANCHES = {"beq": 0, "bne": 1, "blt": 4, "bge": 5, "bltu": 6, "bgeu": 7}


def _r(x):
    return REG[x] if isinstance(x, str) else int(x)


def encode(op, args, pc, labels):
    def target(t):
        return (labels[t] if isinstance(t, str) else int(t)) - pc

    if op in R_OPS:
        f3, f7 = R_OPS[op]
        return (f7 << 25) | (_r(args[2]) << 20) | (_r(args[1]) << 15) | (f3 << 12) | (_r(args[0]) << 7) | 0x33
    if op in I_OPS:
        return ((int(args[2]) & 0xFFF) << 20) | (_r(args[1]) << 15) | (I_OPS[op] << 12) | (_r(args[0]) << 7) | 0x13
    if op in SH_OPS:
        f3, f7 = SH_OPS[op]
        return (f7 << 25) | ((int(args[2]) & 31) << 20) | (_r(args[1]) << 15) | (f3 << 12) | (_r(args[0]) << 7) | 0x13
    if op in LOADS:      # lw rd, imm(rs1) as ("lw", rd, imm, rs1)
        return ((int(args[1]) & 0xFFF) << 20) | (_r(args[2]) << 15) | (LOADS[op] << 12) | (_r(args[0]) << 7) | 0x03
    if op in STORES:     # sw rs2, imm(rs1) as ("sw", rs2, imm, rs1)
        imm = int(args[1]) & 0xFFF
        return ((imm >> 5) << 25) | (_r(args[0]) << 20) | (_r(args[2]) << 15) | (STORES[op] << 12) | ((imm & 31) << 7) | 0x23
    if op in BRANCHES:
        off = target(args[2]) & 0x1FFF
        return (((off >> 12) & 1) << 31) | (((off >> 5) & 0x3F) << 25) | (_r(args[1]) << 20) | (_r(args[0]) << 15) | \
            (BRANCHES[op] << 12) | (((off >> 1) & 0xF) << 8) | (((off >> 11) & 1) << 7) | 0x63
    if op == "lui":
        return ((int(args[1]) & 0xFFFFF) << 12) | (_r(args[0]) << 7) | 0x37
    if op == "auipc":
        return ((int(args[1]) & 0xFFFFF) << 12) | (_r(args[0]) << 7) | 0x17
    if op == "jal":
        off = target(args[1]) & 0x1FFFFF
        return (((off >> 20) & 1) << 31) | (((off >> 1) & 0x3FF) << 21) | (((off >> 11) & 1) << 20) | \
            (((off >> 12) & 0xFF) << 12) | (_r(args[0]) << 7) | 0x6F
    if op == "jalr":     # ("jalr", rd, imm, rs1)
        return ((int(args[1]) & 0xFFF) << 20) | (_r(args[2]) << 15) | (_r(args[0]) << 7) | 0x67
    if op == "ecall":
        return 0x00000073
    if op == "ebreak":
        return 0x00100073
    if op == "fence":
        return 0x0000000F
    if op == "word":
        return int(args[0]) & 0xFFFFFFFF
    raise ValueError("unknown mnemonic %r" % op)


def li(rd, value):
    """load a 32-bit constant: lui + addi"""
    value &= 0xFFFFFFFF
    lo = value & 0xFFF
    if lo >= 0x800:
        lo -= 0x1000
    hi = ((value - lo) >> 12) & 0xFFFFF
    out = []
    if hi:
        out.append(("lui", rd, hi))
        if lo:
            out.append(("addi", rd, rd, lo))
    else:
        out.append(("addi", rd, "zero", lo))
    return out


def assemble(program, base=0x00200800):
    """program: list of instruction tuples, label strings ("name:") and lists from li()"""
    flat = []
    for item in program:
        if isinstance(item, list):
            flat.extend(item)
        else:
            flat.append(item)
    labels, pc = {}, base
    for item in flat:
        if isinstance(item, str):
            labels[item.rstrip(":")] = pc
        else:
            pc += 4
    words, pc = [], base
    for item in flat:
        if isinstance(item, str):
            continue
        words.append(encode(item[0], item[1:], pc, labels))
        pc += 4
    return struct.pack("<%dI" % len(words), *words), labels


def elf(code, entry=0x00200800, data=b"", data_addr=0x00300000):
    """ELF32 little-endian RISC-V executable: code at `entry` (risc0 guests are linked at
    -Ttext=0x00200800, pipeline/src/executor.rs of the reference), optional data segment"""
    segs = [(entry, code)] + ([(data_addr, data)] if data else [])
    ehsize, phentsize = 52, 32
    off = ehsize + phentsize * len(segs)
    hdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8)
    hdr += struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, entry, ehsize, 0, 0, ehsize, phentsize, len(segs), 40, 0, 0)
    ph, body = b"", b""
    for vaddr, blob in segs:
        ph += struct.pack("<IIIIIIII", 1, off + len(body), vaddr, vaddr, len(blob), len(blob), 7, 4)
        body += blob
    return hdr + ph + body
