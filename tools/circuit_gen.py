#!/usr/bin/env python3
"""Turns a circuit's constraint list (risc0-zkp 1.0.1 adapter.rs `PolyExtStepDef`; RECALLED) into
straight-line HIP: what risc0's own build does for its CUDA / Metal `eval_check` kernels, here for
gfx950 against libraiko_hip.so.  The step list is the same operand rk_program_create takes
(include/raiko_hip.h); the library's interpreter needs no build step, this generator removes the
interpreter's per-step costs (op fetch, slot traffic through LDS) -- the compiler keeps the values in
VGPRs and schedules the tap loads.

Output: one .hip file with
  int  <name>_eval_check(void* user, const rk_circuit_view*, const uint32_t poly_mix[4], uint32_t* d_check)
       -- an rk_circuit_hooks.eval_check (CircuitHal::eval_check);
  int  <name>_poly_ext(void* user, const rk_segment* pub, ...)   -- an rk_poly_ext_fn (CircuitDef::poly_ext).
Dead steps are dropped, every mix state's `mul` becomes a compile-time power of poly_mix (one table
per proof, read through the scalar cache), constants are emitted as Montgomery literals.

    python tools/circuit_gen.py --toy 8,4,8 --n-mix 8 --name toy_gen -o examples/toy_circuit/_build/toy_gen.hip
    python tools/circuit_gen.py --steps steps.npy --ret 123 --taps taps.npz --name rv32im -o rv32im_check.hip
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

P = 2013265921
CONST, GET, GET_GLOBAL, ADD, SUB, MUL, TRUE, AND_EQZ, AND_COND = range(9)


def mont(v: int) -> int:
    return (v % P) * (1 << 32) % P


def flatten_taps(taps):
    """[(group, offset, back)] in eval_u order"""
    out = []
    for r in range(taps.n_regs):
        for b in taps.combo(int(taps.reg_combo[r])):
            out.append((int(taps.reg_group[r]), int(taps.reg_offset[r]), int(b)))
    return out


def analyse(steps, ret):
    """the compile of raiko_amd/csrc/circuit_program.hip, minus slots: liveness and mix exponents"""
    fp, mx, where = [], [], []
    for op, a, b, c in steps:
        op, a, b, c = int(op), int(a), int(b), int(c)
        if op <= MUL:
            if op in (ADD, SUB, MUL) and (a >= len(fp) or b >= len(fp)):
                raise ValueError("operand not yet pushed")
            where.append((False, len(fp)))
            fp.append(dict(op=op, a=a, b=b, live=False))
        elif op <= AND_COND:
            m = dict(op=op, x=a, v=b, inner=c, k=0, zero=op == TRUE, live=False)
            if op != TRUE:
                if a >= len(mx) or b >= len(fp) or (op == AND_COND and c >= len(mx)):
                    raise ValueError("operand not yet pushed")
                if op == AND_COND:
                    m["k"] = mx[a]["k"] + mx[c]["k"]
                    m["zero"] = mx[a]["zero"] and mx[c]["zero"]
                else:
                    m["k"] = mx[a]["k"] + 1
            where.append((True, len(mx)))
            mx.append(m)
        else:
            raise ValueError("unknown op %d" % op)
    if ret >= len(mx):
        raise ValueError("ret out of range")
    mx[ret]["live"] = True
    for is_mix, i in reversed(where):
        if is_mix:
            m = mx[i]
            if not m["live"] or m["op"] == TRUE:
                continue
            mx[m["x"]]["live"] = True
            if m["op"] == AND_EQZ:
                fp[m["v"]]["live"] = True
            elif not mx[m["inner"]]["zero"]:
                mx[m["inner"]]["live"] = True
                fp[m["v"]]["live"] = True
        else:
            v = fp[i]
            if v["live"] and v["op"] in (ADD, SUB, MUL):
                fp[v["a"]]["live"] = fp[v["b"]]["live"] = True
    return fp, mx, where


def generate(steps, ret, taps, name, chunk=300, host_inline_limit=2000) -> str:
    """chunk: statements per device function.  One basic block of tens of thousands of statements
    costs the compiler quadratic time (10^4 statements: half an hour), so the list is cut into
    `__noinline__` functions of `chunk` statements; a value used outside the function that made it
    travels through a per-lane carry array (scratch memory), leaves are re-read where they are used."""
    fp, mx, where = analyse(steps, ret)
    flat = flatten_taps(taps)
    powers = sorted({mx[m["x"]]["k"] for m in mx if m["live"] and m["op"] != TRUE and not m["zero"]})
    pw_idx = {k: i for i, k in enumerate(powers)}
    need = [0, 0]
    group_min = [0, 0, 0]
    for v in fp:
        if v["live"] and v["op"] == GET_GLOBAL:
            need[v["a"]] = max(need[v["a"]], v["b"] + 1)
        if v["live"] and v["op"] == GET:
            g, off, _ = flat[v["a"]]
            group_min[g] = max(group_min[g], off + 1)

    def is_leaf(i):
        return fp[i]["op"] in (CONST, GET, GET_GLOBAL)

    def emitted_mix(m):
        return m["live"] and m["op"] != TRUE and not m["zero"]

    # the statements, in list order; copies (AND_COND with a zero inner block) are aliases, not statements
    alias = {}

    def mix_name(i):
        while i in alias:
            i = alias[i]
        return i

    items = []
    for is_mix, i in where:
        if not is_mix:
            if fp[i]["live"] and not is_leaf(i):
                items.append(("fp", i))
        else:
            m = mx[i]
            if not emitted_mix(m):
                continue
            if m["op"] == AND_COND and mx[m["inner"]]["zero"]:
                alias[i] = m["x"]
                continue
            items.append(("mx", i))
    n_chunks = max(1, (len(items) + chunk - 1) // chunk)
    chunk_of = {it: k // chunk for k, it in enumerate(items)}
    END = n_chunks
    # operands of every statement and the chunks each value is used in
    uses = {}

    def note(kind, j, c):
        uses.setdefault((kind, j), set()).add(c)

    def operands(it):
        kind, i = it
        if kind == "fp":
            return [("fp", fp[i]["a"]), ("fp", fp[i]["b"])]
        m = mx[i]
        ops = [("fp", m["v"])]
        if not mx[m["x"]]["zero"]:
            ops.append(("mx", mix_name(m["x"])))
        if m["op"] == AND_COND:
            ops.append(("mx", mix_name(m["inner"])))
        return ops

    for it in items:
        for kind, j in operands(it):
            if kind == "mx" or not is_leaf(j):
                note(kind, j, chunk_of[it])
    ret_name = None if mx[ret]["zero"] else mix_name(ret)
    if ret_name is not None:
        note("mx", ret_name, END)
    # carry slots for values that leave their function (freed after the last chunk that reads them)
    slot = {}
    free = {"fp": [], "mx": []}
    nxt = {"fp": 0, "mx": 0}
    release_at = {}
    by_chunk = [[] for _ in range(n_chunks)]
    for it in items:
        by_chunk[chunk_of[it]].append(it)
    bodies = []
    for c in range(n_chunks):
        lines = []
        have = set()                           # names defined in this function

        def name_of(kind, j):
            nm = ("f%d" if kind == "fp" else "x%d") % j
            if (kind, j) in have:
                return nm
            have.add((kind, j))
            if kind == "fp" and is_leaf(j):
                v = fp[j]
                if v["op"] == CONST:
                    lines.append("    const uint32_t %s = 0x%08xu;" % (nm, mont(v["a"])))
                elif v["op"] == GET:
                    g, off, back = flat[v["a"]]
                    idx = "i" if back == 0 else "((i + a.d - ((size_t)%du << a.blow)) & (a.d - 1))" % back
                    lines.append("    const uint32_t %s = a.lde[%d][(size_t)%du * a.d + %s];" % (nm, g, off, idx))
                else:
                    lines.append("    const uint32_t %s = tab[a.%s_base + %du];" % (nm, "glob" if v["a"] == 0 else "mix", v["b"]))
            elif kind == "fp":
                lines.append("    const uint32_t %s = c[%d];" % (nm, slot[(kind, j)]))
            else:
                lines.append("    const Ext %s = cx[%d];" % (nm, slot[(kind, j)]))
            return nm

        for it in by_chunk[c]:
            kind, i = it
            if kind == "fp":
                v = fp[i]
                fn = {ADD: "add", SUB: "sub", MUL: "mul"}[v["op"]]
                a_, b_ = name_of("fp", v["a"]), name_of("fp", v["b"])
                lines.append("    const uint32_t f%d = bb::%s(%s, %s);" % (i, fn, a_, b_))
            else:
                m = mx[i]
                pw = pw_idx[mx[m["x"]]["k"]]
                term = "bb::scale(load_pw(tab, a.pw_base, %d), %s)" % (pw, name_of("fp", m["v"]))
                if m["op"] == AND_COND:
                    term = "bb::mul(%s, %s, a.wm)" % (term, name_of("mx", mix_name(m["inner"])))
                if not mx[m["x"]]["zero"]:
                    term = "bb::add(%s, %s)" % (name_of("mx", mix_name(m["x"])), term)
                lines.append("    const Ext x%d = %s;" % (i, term))
            have.add(it)
            later = [u for u in uses.get(it, ()) if u > c]
            if later:
                k = free[kind].pop() if free[kind] else nxt[kind]
                if k == nxt[kind]:
                    nxt[kind] += 1
                slot[it] = k
                release_at.setdefault(max(later), []).append(it)
                lines.append("    %s[%d] = %s%d;" % ("c" if kind == "fp" else "cx", k, "f" if kind == "fp" else "x", i))
        for it in release_at.get(c, ()):
            free[it[0]].append(slot[it])
        bodies.append("\n".join(lines))
    single = n_chunks == 1
    qual = "__forceinline__" if single else "__noinline__"
    funcs = "\n\n".join(
        "__device__ %s void %s_part%d(const Args& a, const_u32 tab, size_t i, uint32_t* c, Ext* cx) {\n%s\n}" % (qual, name, k, body)
        for k, body in enumerate(bodies))
    calls = "\n".join("    %s_part%d(a, tab, i, c, cx);" % (name, k) for k in range(n_chunks))
    result = "bb::ext_zero()" if ret_name is None else "cx[%d]" % slot[("mx", ret_name)]

    # host half: straight-line for small lists (an independent reading of the list), the list itself for large ones
    n_live = len(items)
    if n_live <= host_inline_limit:
        host = []
        for is_mix, i in where:
            if not is_mix:
                v = fp[i]
                if not v["live"]:
                    continue
                if v["op"] == CONST:
                    host.append("    const Ext f%d = bb::ext_from(0x%08xu);" % (i, mont(v["a"])))
                elif v["op"] == GET:
                    host.append("    const Ext f%d = u[%d];" % (i, v["a"]))
                elif v["op"] == GET_GLOBAL:
                    host.append("    const Ext f%d = bb::ext_from(%s[%d]);" % (i, "pub->globals" if v["a"] == 0 else "mix", v["b"]))
                else:
                    fn = {ADD: "add", SUB: "sub", MUL: "mul"}[v["op"]]
                    host.append("    const Ext f%d = bb::%s(f%d, f%d%s);" % (i, fn, v["a"], v["b"], ", wm" if v["op"] == MUL else ""))
            else:
                m = mx[i]
                if not emitted_mix(m) or i in alias:
                    continue
                term = "bb::mul(pw[%d], f%d, wm)" % (pw_idx[mx[m["x"]]["k"]], m["v"])
                if m["op"] == AND_COND:
                    term = "bb::mul(%s, x%d, wm)" % (term, mix_name(m["inner"]))
                if not mx[m["x"]]["zero"]:
                    term = "bb::add(x%d, %s)" % (mix_name(m["x"]), term)
                host.append("    const Ext x%d = %s;" % (i, term))
        host.append("    const Ext result = %s;" % ("bb::ext_zero()" if ret_name is None else "x%d" % ret_name))
        host_src = "\n".join(host)
        table = ""
    else:
        table = "const uint32_t STEPS[][4] = {\n%s\n};\nconstexpr uint32_t RET = %du;\n" % (
            ",\n".join("    {%d, %d, %d, %d}" % tuple(int(x) for x in st) for st in steps), ret)
        host_src = HOST_INTERPRETER
    return TEMPLATE.format(name=name, funcs=funcs, calls=calls, host=host_src, table=table, result=result, n_pw=len(powers),
                           powers=", ".join("%du" % k for k in powers) or "0u", need_glob=need[0], need_mix=need[1],
                           gmin0=group_min[0], gmin1=group_min[1], gmin2=group_min[2], n_taps=len(flat),
                           n_steps=len(steps), n_live=n_live, n_chunks=n_chunks, n_c=max(1, nxt["fp"]), n_cx=max(1, nxt["mx"]))


HOST_INTERPRETER = r"""    // a list this long is interpreted on the host (once per proof), not unrolled
    struct Mix { Ext tot, mul; };
    std::vector<Ext> f;
    std::vector<Mix> x;
    for (const auto& st : STEPS) {
        switch (st[0]) {
            case 0: f.push_back(bb::ext_from(bb::encode(st[1]))); break;
            case 1: f.push_back(u[st[1]]); break;
            case 2: f.push_back(bb::ext_from(st[1] == 0 ? (st[2] < pub->n_globals ? pub->globals[st[2]] : 0u) : (st[2] < n_mix ? mix[st[2]] : 0u))); break;
            case 3: f.push_back(bb::add(f[st[1]], f[st[2]])); break;
            case 4: f.push_back(bb::sub(f[st[1]], f[st[2]])); break;
            case 5: f.push_back(bb::mul(f[st[1]], f[st[2]], wm)); break;
            case 6: x.push_back(Mix{bb::ext_zero(), bb::ext_one()}); break;
            case 7: { const Mix m = x[st[1]]; x.push_back(Mix{bb::add(m.tot, bb::mul(m.mul, f[st[2]], wm)), bb::mul(m.mul, pm, wm)}); break; }
            default: { const Mix m = x[st[1]], in = x[st[3]];
                       x.push_back(Mix{bb::add(m.tot, bb::mul(bb::mul(f[st[2]], in.tot, wm), m.mul, wm)), bb::mul(m.mul, in.mul, wm)}); }
        }
    }
    const Ext result = x[RET].tot;"""


TEMPLATE = r"""// GENERATED by tools/circuit_gen.py from a {n_steps}-step constraint list ({n_live} live statements in {n_chunks} function(s)).
// CircuitHal::eval_check / CircuitDef::poly_ext of one circuit as straight-line code against
// libraiko_hip.so (include/raiko_hip.h rk_circuit_hooks.eval_check, rk_poly_ext_fn).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "bb.hpp"
#include "raiko_hip.h"

namespace {{

using bb::Ext;
constexpr uint32_t N_PW = {n_pw};
const uint32_t POWERS[] = {{{powers}}};
{table}
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint32_t* const_u32;
#else
typedef const uint32_t* const_u32;
#endif

struct Args {{
    const uint32_t* lde[3];
    uint64_t tab;        // globals | accum mix | powers of poly_mix (4 words each), wave-uniform
    uint32_t* check;
    size_t d;
    uint32_t glob_base, mix_base, pw_base, wm;
    uint32_t blow;       // log2 of the blow-up (rk_params.blowup_log2)
    uint32_t inv_den[16];
}};

__device__ __forceinline__ Ext load_pw(const_u32 tab, uint32_t base, uint32_t j) {{
    const_u32 p = tab + base + 4 * j;
    return Ext{{{{p[0], p[1], p[2], p[3]}}}};
}}

{funcs}

__global__ __launch_bounds__(256) void {name}_kernel(Args a) {{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.d) return;
    const const_u32 tab = (const_u32)a.tab;
    uint32_t c[{n_c}];   // values that travel between the functions
    Ext cx[{n_cx}];
{calls}
    const Ext tot = bb::scale({result}, a.inv_den[i & ((1u << a.blow) - 1)]);
#pragma unroll
    for (int e = 0; e < 4; e++) a.check[(size_t)e * a.d + i] = tot.c[e];
}}

}}  // namespace

extern "C" {{

int {name}_eval_check(void*, const rk_circuit_view* v, const uint32_t poly_mix[4], uint32_t* d_check) {{
    if (!v || !v->ctx || v->n_globals < {need_glob}u || v->n_mix < {need_mix}u) return 1;
    const uint32_t gmin[3] = {{{gmin0}u, {gmin1}u, {gmin2}u}};
    for (int g = 0; g < 3; g++)
        if (gmin[g] && (!v->d_lde[g] || v->group_size[g] < gmin[g])) return 1;
    rk_params prm;
    if (rk_get_params(v->ctx, &prm) != RK_OK) return 2;
    const uint32_t wm = bb::encode(prm.ext_w);
    const size_t n = (size_t)1 << v->po2, d = n << prm.blowup_log2;
    Args a{{}};
    a.blow = prm.blowup_log2;
    for (int g = 0; g < 3; g++) a.lde[g] = v->d_lde[g];
    a.check = d_check;
    a.d = d;
    a.wm = wm;
    a.glob_base = 0;
    a.mix_base = v->n_globals;
    a.pw_base = (v->n_globals + v->n_mix + 3u) & ~3u;
    std::vector<uint32_t> tab(a.pw_base + 4 * (size_t)N_PW + 4, 0);
    if (v->n_globals) std::memcpy(tab.data(), v->globals, (size_t)v->n_globals * 4);
    if (v->n_mix) std::memcpy(tab.data() + a.mix_base, v->mix, (size_t)v->n_mix * 4);
    Ext pm, cur = bb::ext_one();
    std::memcpy(pm.c, poly_mix, 16);
    uint32_t at = 0;
    for (uint32_t j = 0; j < N_PW; j++) {{
        cur = bb::mul(cur, bb::pow(pm, POWERS[j] - at, wm), wm);
        at = POWERS[j];
        std::memcpy(&tab[a.pw_base + 4 * j], cur.c, 16);
    }}
    const uint32_t sn = bb::pow(bb::encode(prm.coset_shift), n);
    const uint32_t wb = bb::pow(bb::encode(prm.root_2_27), (uint64_t)1 << (27 - prm.blowup_log2));
    for (uint32_t r = 0; r < (1u << prm.blowup_log2); r++) a.inv_den[r] = bb::inv(bb::sub(bb::mul(sn, bb::pow(wb, r)), bb::ONE));
    void* d_tab = nullptr;
    if (rk_alloc(v->ctx, tab.size() * 4, &d_tab) != RK_OK) return 3;
    int rc = rk_h2d(v->ctx, d_tab, tab.data(), tab.size() * 4) == RK_OK ? 0 : 4;   // synchronises: `tab` may go
    if (!rc) {{
        a.tab = (uint64_t)(uintptr_t)d_tab;
        hipLaunchKernelGGL({name}_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, (hipStream_t)v->stream, a);
        if (hipGetLastError() != hipSuccess) rc = 5;
    }}
    (void)rk_free(v->ctx, d_tab);   // drains the stream first
    return rc;
}}

int {name}_poly_ext(void* user, const rk_segment* pub, const uint32_t poly_mix[4], const uint32_t* eval_u_ext, size_t n_taps,
                    const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]) {{
    // user: optional pointer to the canonical W of the extension (uint32_t), NULL = risc0's
    if (!pub || n_taps != {n_taps}u || pub->n_globals < {need_glob}u || n_mix < {need_mix}u) return 1;
    const uint32_t wm = user ? bb::encode(*(const uint32_t*)user) : bb::WM_RISC0;
    const Ext* u = reinterpret_cast<const Ext*>(eval_u_ext);
    (void)u;
    Ext pm;
    std::memcpy(pm.c, poly_mix, 16);
    std::vector<Ext> pw(N_PW + 1);
    Ext cur = bb::ext_one();
    uint32_t at = 0;
    for (uint32_t j = 0; j < N_PW; j++) {{
        cur = bb::mul(cur, bb::pow(pm, POWERS[j] - at, wm), wm);
        at = POWERS[j];
        pw[j] = cur;
    }}
{host}
    std::memcpy(out_ext, result.c, 16);
    return 0;
}}

}}  // extern "C"
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--toy", help="Wa,Wc,Wd: the toy circuit's list (raiko_amd.circuit_program.toy_program)")
    ap.add_argument("--n-mix", type=int, default=8)
    ap.add_argument("--steps", help=".npy (n, 4) uint32")
    ap.add_argument("--ret", type=int)
    ap.add_argument("--taps", help=".npz with group_size, reg_group, reg_offset, reg_combo, combo_off, combo_backs")
    ap.add_argument("--chunk", type=int, default=300, help="statements per device function")
    ap.add_argument("--name", required=True)
    ap.add_argument("-o", "--out", required=True)
    args = ap.parse_args()
    # the builder and TapSet are plain Python + numpy: no GPU library is touched here
    from raiko_amd.segment import TapSet, synthetic_tapset
    if args.toy:
        from raiko_amd.circuit_program import toy_program
        taps = synthetic_tapset(*[int(x) for x in args.toy.split(",")])
        steps, ret = toy_program(taps, args.n_mix)
    else:
        z = np.load(args.taps)
        taps = TapSet(tuple(int(x) for x in z["group_size"]), z["reg_group"], z["reg_offset"], z["reg_combo"], z["combo_off"],
                      z["combo_backs"])
        steps, ret = np.load(args.steps), args.ret
    src = generate(steps, ret, taps, args.name, chunk=args.chunk)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(src)
    print("%s: %d steps -> %d lines" % (args.out, len(steps), src.count("\n")))


if __name__ == "__main__":
    main()
