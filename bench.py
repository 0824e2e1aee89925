#!/usr/bin/env python3
"""bench.py -- proven RISC-V cycles/sec on synthetic 2^20-cycle segments (BASELINE.json config 2).

A "step" is one full segment proof on one GPU: three trace groups (16 / 16 / 224 columns x 2^20
rows) through iNTT -> zk-shift -> 4x LDE -> Poseidon2 Merkle commit, the 16-column check group,
DEEP mixing + division, FRI (arity 16 down to degree 256) and 50 query openings, producing the
seal.  The timed region is ONE call of the drop-in entry point, `rk_prove_session` (what replaces
`session.prove()`, reference provers/risc0/driver/src/bonsai.rs:271): `steps` segments, `--inflight`
of them in flight, every seal verified by the library's host-side verifier inside the region.
`value` has the inputs resident in HBM when the clock starts; `value_with_h2d` is the same session
from host-resident arrays through the library's staging ring (PCIe upload inside the region).
With N > 1 ranks every rank proves `steps` segments of its own (segments are independent: weak
scaling) and the seals are gathered to rank 0 with one RCCL all_gather inside the timed region;
`--total-segments S` fixes the total instead (strong scaling: one block's segment list).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--widths", type=str, default="16,16,224", help="accum,code,data column counts")
    ap.add_argument("--cpu-po2", type=int, default=None,
                    help="segment size of the all-threads CPU sample (default: --po2, the GPU line's size)")
    ap.add_argument("--cpu-po2-1t", type=int, default=16, help="segment size of the 1-thread CPU sample")
    ap.add_argument("--inflight", type=int, default=3,
                    help="segments proven concurrently per GPU (one prover context + HIP stream each); "
                         "the latency-bound parts of one proof (Merkle tops, transcript round trips) "
                         "overlap the throughput-bound parts of the other")
    ap.add_argument("--upload-ahead", type=int, default=2)
    ap.add_argument("--total-segments", type=int, default=0,
                    help="strong scaling: this many segments in all, sharded round-robin over the ranks "
                         "(--steps is then ignored); default 0 = every rank proves --steps segments (weak)")
    ap.add_argument("--preset", choices=["risc0", "sp1", "sp1-p3"], default="risc0",
                    help="parameter set of the proofs (rk_session_opts.params): sp1 = SP1 core's RECALLED set -- x^4 - 11, "
                         "Poseidon2 width 16, blow-up 2, FRI fold 2 down to a constant, 100 queries, 16 proof-of-work bits; "
                         "the contract line is risc0's.  sp1-p3: not a segment proof at all but SP1's own proof system as far as "
                         "it is built (rk_p3_prove: Plonky3-style uni-stark over the two-adic FRI PCS, --p3-shape tables of a "
                         "chip-shaped synthetic AIR under SP1's parameter set); its line has its own metric")
    ap.add_argument("--p3-shape", default="20x256,19x128,16x64,10x32", help="--preset sp1-p3: log2 rows x columns per table")
    ap.add_argument("--p3-jit", action="store_true", help="--preset sp1-p3: quotient through the hiprtc-generated kernel")
    ap.add_argument("--p3-lookups", type=int, default=0, help="--preset sp1-p3: every table sends and receives this many tuples "
                                                              "(2x interactions) through the permutation argument")
    ap.add_argument("--circuit", type=str, default="8000",
                    help="comma-separated op counts, e.g. 8000,33000: for each, the same S20 session again with the circuit's "
                         "two stages inside the timed region -- eval_check from a synthetic step list of that many ops over "
                         "the 256 columns (rk_program, compiled with hiprtc), the check group computed inside the proof; "
                         "reported as `with_circuit`, never as `value` (the rv32im list itself is outside the tree); "
                         "default 8000 (5 s of hiprtc), '' to skip, 33000 costs ~30 s of compilation")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-resident (value_with_h2d) run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-small", action="store_true",
                    help="skip the extra line for segments of 2^18 cycles (script/prove-block.sh's execution_po2)")
    ap.add_argument("--no-verify", action="store_true", help="leave rk_verify_segment out of the timed region")
    return ap.parse_args()


def device_segment(torch, seg_mod, po2, widths, seed, device, blowup_log2=2):
    """A synthetic segment whose O(trace) inputs live in HBM; small metadata on the host."""
    import numpy as np
    P = seg_mod.P
    n = 1 << po2
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    groups = [torch.randint(0, P, (w, n), dtype=torch.int32, device=device, generator=g) for w in widths]
    check = torch.randint(0, P, (4, n << blowup_log2), dtype=torch.int32, device=device, generator=g)
    rng = np.random.Generator(np.random.PCG64(seed))
    seg = seg_mod.Segment(po2=po2, taps=seg_mod.synthetic_tapset(*widths), groups=[None, None, None], check=None,
                          globals_=rng.integers(0, P, size=(32,), dtype=np.uint32))
    return seg, groups, check


def main_p3(args):
    """BASELINE config 5 as far as it is built: one shard-shaped rk_p3_prove per step (tools/bench_p3.py), verified on the host"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_p3
    ns = argparse.Namespace(shape=args.p3_shape, jit=args.p3_jit, reps=min(max(args.steps, 1), 5), preset=1, no_verify=False,
                            shards=max(args.steps, 1), batch=args.inflight, lookups=args.p3_lookups, host_traces=True)
    r = bench_p3.run(ns)
    out = {"metric": "proven trace cells/sec (Plonky3-style uni-stark, SP1 parameter set; NOT the contract metric)",
           "value": r["cells_per_s"], "unit": "cells/s", "n_gpus": 1, "steps": ns.reps, "warmup": 1, "ms_per_step": r["wall_ms"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (BabyBear Montgomery)", "data": "synthetic",
           "config": {"workload": "one rk_p3_prove of tables %s (log2 rows x columns, chip-shaped degree-3 AIR, %s ops per quotient "
                                  "point%s), blow-up 2, 100 queries, 16 proof-of-work bits, Poseidon2 width 16"
                                  % (r["shape"], r["ops_per_point"], ", %d interactions per table" % (2 * r["lookups"]) if r["lookups"] else ""),
                      "entry_point": "rk_p3_prove (on_device traces) + rk_p3_verify", "quotient": "generated kernel" if r["jit"] else "interpreter"},
           "stages_ms": r["stages_ms"], "proof_words": r["proof_words"], "verify_rc": r["verify_rc"], "verify_ms": r["verify_ms"],
           "one_proof_at_a_time": {"cells_per_s": r["cells_per_s"], "ms": r["wall_ms"]}}
    if "shards" in r:   # the headline of this line: `steps` shards through rk_p3_prove_shards, --inflight of them in flight, each verified
        out["value"] = r["shards"]["cells_per_s"]
        out["ms_per_step"] = r["shards"]["ms_per_shard"]
        out["shards"] = r["shards"]
    if "host_traces" in r:   # the PCIe-inclusive rate: the same shards with their traces in pageable host memory (never `value`)
        out["value_with_h2d"] = r["host_traces"]["cells_per_s"]
        out["host_traces"] = r["host_traces"]
        out["config"]["entry_point"] = "rk_p3_prove_shards (batch = %d in flight, on_device traces, every proof verified by rk_p3_verify)" % r["shards"]["batch"]
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    if args.preset == "sp1-p3":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--preset sp1-p3 is a one-GPU line")
        return main_p3(args)
    widths = tuple(int(x) for x in args.widths.split(","))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist
    from raiko_amd import segment as seg_mod
    from raiko_amd.dist import gather_seals, shard_indices
    from raiko_amd.hal import HipHal, make_params, prove_session, session_kernel_stats, session_set_kernel_timing, verify_segment
    blob = make_params(1) if args.preset == "sp1" else None
    blow = blob.blowup_log2 if blob is not None else 2

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # RAIKO_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share
    # GPUs (LOCAL_RANK modulo the device count) and the collectives run on CPU tensors.  The
    # driver's runs use the default: one rank per GPU over RCCL ("nccl" on ROCm).
    backend = os.environ.get("RAIKO_BENCH_BACKEND", "nccl")
    gpu_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    coll_device = torch.device("cpu") if backend == "gloo" else device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    strong = args.total_segments > 0
    total_segments = args.total_segments if strong else args.steps * world
    my_steps = len(shard_indices(total_segments, rank, world))
    inflight = max(1, min(args.inflight, max(my_steps, 1)))

    # two distinct resident segments per rank, alternated, so no step sees data it just proved
    segs = [device_segment(torch, seg_mod, args.po2, widths, 20240807 + 1000 * rank + i, device, blow) for i in range(2)]
    torch.cuda.synchronize()

    def session(n, host=None, verify=not args.no_verify):
        """n segments through rk_prove_session: device-resident inputs (left untouched: on_device = 1)
        or the host copies `host` through the staging ring"""
        if n == 0:
            return []
        if host is not None:
            return prove_session([host[i % 2] for i in range(n)], device=gpu_index, inflight=inflight,
                                 upload_ahead=args.upload_ahead, verify=verify, params=blob)
        return prove_session([segs[i % 2][0] for i in range(n)], device=gpu_index, inflight=inflight,
                             upload_ahead=args.upload_ahead, verify=verify, params=blob,
                             device_inputs=[(segs[i % 2][1], segs[i % 2][2]) for i in range(n)])

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # Serial calibration proofs on a context of its own (also a warm-up): with a single stream
    # the hipEvent brackets are pure kernel time, which picks the dominant kernel class; under
    # concurrency a bracket also contains time spent queued behind the other contexts' kernels.
    hal = HipHal(gpu_index)
    if blob is not None:
        hal.set_params(1)
    hal.prove_segment(segs[0][0], device_inputs=(segs[0][1], segs[0][2]))  # cold: first-touch allocations, tables
    hal.set_kernel_timing(True)
    CALIB_PROOFS = 3
    serial_stage = None
    for i in range(CALIB_PROOFS):
        hal.prove_segment(segs[(i + 1) % 2][0], device_inputs=(segs[(i + 1) % 2][1], segs[(i + 1) % 2][2]))
        t = hal.last_timing()
        serial_stage = t if serial_stage is None else {k: serial_stage[k] + t[k] for k in t}
    serial_stage = {k: v / CALIB_PROOFS for k, v in serial_stage.items()}
    calib = hal.kernel_stats()          # totals over the CALIB_PROOFS serial proofs
    hal.set_kernel_timing(False)
    hal.close()
    dom_name = max(calib.items(), key=lambda kv: kv[1]["ms"])[0]
    if args.warmup:
        warm = session(max(args.warmup, inflight), verify=False)
        if world > 1:
            # the first collective of each kind builds RCCL's channels: keep that out of the timed
            # region, like the other one-time costs (same shapes as the timed gather)
            pad = [warm[i % len(warm)] for i in range(my_steps)]
            gather_seals(pad, total_segments, device=coll_device)
    session_set_kernel_timing(gpu_index, True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    seals = session(my_steps)
    if world > 1:
        # rank r proved global segments r, r+world, ...: gather in that order (one collective)
        gather_seals(seals, total_segments, device=coll_device)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    kstats = session_kernel_stats(gpu_index)
    session_set_kernel_timing(gpu_index, False)

    # the same session from host-resident arrays: PCIe upload through the staging ring inside the region
    elapsed_h2d = None
    if not args.no_h2d:
        host = []
        for seg, groups, check in segs:
            host.append(seg_mod.Segment(po2=seg.po2, taps=seg.taps,
                                        groups=[g.cpu().numpy().view(np.uint32) for g in groups],
                                        check=check.cpu().numpy().view(np.uint32), globals_=seg.globals_))
        session(min(my_steps, inflight + args.upload_ahead), host=host, verify=False)  # staging ring warm-up
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        seals_h = session(my_steps, host=host)
        if world > 1:
            gather_seals(seals_h, total_segments, device=coll_device)
        torch.cuda.synchronize()
        barrier()
        elapsed_h2d = max_over_ranks(time.perf_counter() - t1)
        same = all(np.array_equal(a, b) for a, b in zip(seals, seals_h))
        del host

    # the reference's own script proves with execution_po2 = 18 (script/prove-block.sh:71): the same session at that
    # segment size, reported next to the headline (one GPU, default workload only; never part of `value`)
    small = None
    if world == 1 and not args.no_small and args.po2 == 20 and blob is None and args.widths == "16,16,224":
        try:
            s18 = [device_segment(torch, seg_mod, 18, widths, 20240807 + i, device, blow) for i in range(2)]

            def session18(n, verify):
                return prove_session([s18[i % 2][0] for i in range(n)], device=gpu_index, inflight=inflight,
                                     upload_ahead=args.upload_ahead, verify=verify,
                                     device_inputs=[(s18[i % 2][1], s18[i % 2][2]) for i in range(n)])
            session18(2 * inflight, False)
            torch.cuda.synchronize()
            n18 = 2 * args.steps
            t2 = time.perf_counter()
            session18(n18, not args.no_verify)
            torch.cuda.synchronize()
            e18 = time.perf_counter() - t2
            small = {"po2": 18, "segments": n18, "value": round(n18 * (1 << 18) / e18, 1), "unit": "cycles/s",
                     "ms_per_segment": round(e18 / n18 * 1e3, 3),
                     "why": "script/prove-block.sh proves with execution_po2 = 18; same entry point, inputs in HBM, verify as above"}
            del s18
        except Exception as e:  # the headline line must not depend on this
            small = {"po2": 18, "error": repr(e)}

    # the same session with a constraint list of rv32im-like size evaluated inside every proof (ADVICE r2 / VERDICT r2 #3):
    # what `session.prove()` costs per segment once eval_check is in the region (bonsai.rs:271)
    with_circuit = []
    if args.circuit and world == 1 and blob is None:
        from raiko_amd import circuit_program as cp
        chal = HipHal(gpu_index)
        for n_ops in (int(x) for x in args.circuit.split(",")):
            try:
                rng = np.random.default_rng(1)
                taps = segs[0][0].taps
                steps_, ret = cp.synthetic_program(rng, taps, 32, segs[0][0].n_accum_mix, n_fp_ops=n_ops, n_live=0, depth=2,
                                                   n_constraints=max(8, n_ops // 10), local=True)
                prog = cp.Program(steps_, ret, taps)
                t0 = time.perf_counter()
                prog.compile(chal)
                compile_s = time.perf_counter() - t0
                csegs = []
                for seg, groups, _check in segs:
                    cs = seg_mod.Segment(po2=seg.po2, taps=seg.taps, groups=[None, None, None], check=None, globals_=seg.globals_)
                    cs.program = prog
                    csegs.append((cs, groups))
                chal.prove_segment(csegs[0][0], device_inputs=(csegs[0][1], None))      # warm
                chal.prove_segment(csegs[1][0], device_inputs=(csegs[1][1], None))
                cstage = chal.last_timing()

                def csession(n):
                    return prove_session([csegs[i % 2][0] for i in range(n)], device=gpu_index, inflight=inflight,
                                         upload_ahead=args.upload_ahead, verify=not args.no_verify,
                                         device_inputs=[(csegs[i % 2][1], None) for i in range(n)])
                csession(inflight)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                csession(my_steps)
                torch.cuda.synchronize()
                ce = time.perf_counter() - t0
                with_circuit.append({"ops": prog.info()["n_ops"], "steps_in_list": prog.info()["n_steps"],
                                     "value": round(my_steps * (1 << args.po2) / ce, 1), "unit": "cycles/s",
                                     "ms_per_step": round(ce / max(my_steps, 1) * 1e3, 3),
                                     "serial_stage_ms": {k: round(v, 3) for k, v in cstage.items()},
                                     "hiprtc_compile_s": round(compile_s, 2),
                                     "what": "S%d with eval_check of a synthetic %d-op constraint list over the 256 columns inside every "
                                             "proof (rk_circuit_hooks.program, generated kernel); accum still given" % (args.po2, prog.info()["n_ops"])})
                prog.close()
            except Exception as e:  # the headline line must not depend on this
                with_circuit.append({"ops": n_ops, "error": repr(e)})
        chal.close()

    if rank == 0:
        cycles = total_segments * (1 << args.po2)
        value = cycles / elapsed
        per_step = elapsed / max(my_steps, 1)
        # dominant kernel = the class with the most device time in the serial calibration proofs.  Every per-kernel
        # figure of this line (roofline.* and pipeline.kernels) is taken from THOSE proofs: one stream, so a hipEvent
        # bracket is the kernel's own duration -- the number `rocprofv3 --kernel-trace --stats` reports
        # (profiles/rNN_final_kernel_stats_inflight1.csv).  Brackets taken inside the timed region run under
        # `inflight`-way concurrency and also contain time queued behind the other contexts' kernels: they are kept
        # only as the labelled extra `inflight_bracket_avg_ms`.
        dom = calib[dom_name]
        dom_launches = max(dom["launches"], 1)
        achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom_launches, 4),
            "algorithmic_bytes_per_launch": round(dom["bytes"] / dom_launches),
            "source": "%d serial proofs (one context, one stream) before the timed region: kernel durations, "
                      "comparable with rocprofv3 --kernel-trace --stats" % CALIB_PROOFS,
            "inflight_bracket_avg_ms": round(kstats[dom_name]["ms"] / max(kstats[dom_name]["launches"], 1), 4),
            "note": "Poseidon2 hashing is integer-ALU-bound (about 1.36k modular multiplies per 64 B absorbed, "
                    "~6.7k VALU instructions per permutation at 16 lanes/clk/SIMD: see `alu`); the HBM fraction is "
                    "reported because it is the contract figure",
        }
        if dom_name == "hash_rows_kernel" and dom["ms"] > 0:
            # the bound that actually applies: VALU issue.  peak = 256 CUs x 4 SIMDs x 16 lanes/clk x
            # 2.4 GHz (profiles/r01_ubench_isa.txt: every VALU op except plain add/sub issues at that rate)
            perms = seg_mod.poseidon2_permutations(args.po2, widths)["hash_rows"] * CALIB_PROOFS
            alu_peak = 256 * 4 * 16 * 2.4e9 / 1e12
            alu = perms * seg_mod.P2_VALU_PER_PERMUTATION / (dom["ms"] * 1e-3) / 1e12
            roofline["alu"] = {"achieved": round(alu, 2), "peak": round(alu_peak, 2), "unit": "T lane-instr/s",
                               "frac": round(alu / alu_peak, 4),
                               "permutations_per_s": round(perms / (dom["ms"] * 1e-3) / 1e9, 3),
                               "valu_per_permutation": seg_mod.P2_VALU_PER_PERMUTATION}
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                with open(traffic_file) as f:
                    tr = json.load(f)
                if tr.get("kernel") == dom_name:
                    roofline["traffic"] = tr.get("bytes_per_launch")
                    roofline["traffic_source"] = ("profiles/hbm_traffic.json: PMC FETCH_SIZE/WRITE_SIZE of an earlier "
                                                  "rocprofv3 run of this kernel (%s), not measured in this run" % tr.get("source", "see file"))
            except Exception:
                pass
        kernels = {k: {"ms_per_step": round(v["ms"] / CALIB_PROOFS, 3), "launches_per_step": v["launches"] / CALIB_PROOFS,
                       "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                   for k, v in calib.items()}
        # invariants of the block: kernel time of a serial proof cannot exceed the proof's own stage clock, and a
        # kernel class cannot take longer per segment than a serial segment does
        kernel_sum = sum(v["ms_per_step"] for v in kernels.values())
        inv_ok = kernel_sum <= serial_stage["total"] * 1.02 + 0.05 and \
            all(v["ms_per_step"] <= min(serial_stage["total"], per_step * 1e3 * inflight) for v in kernels.values())
        if not inv_ok:
            print("bench.py: per-kernel figures break their invariants: sum %.3f ms vs serial segment %.3f ms" %
                  (kernel_sum, serial_stage["total"]), file=sys.stderr)
        sp1 = blob is not None
        algo = seg_mod.algorithmic_bytes(args.po2, widths, 1, 1, 1) if sp1 else seg_mod.algorithmic_bytes(args.po2, widths)
        shape_text = ("%d check columns, blow-up 2, Poseidon2 width 16 Merkle, FRI arity 2 to a constant, 100 queries, "
                      "16 proof-of-work bits, x^4 - 11 (SP1 core's RECALLED parameter set on risc0's flow)" % (4 << blow)) if sp1 else \
            "16 check columns, blow-up 4, Poseidon2 Merkle, FRI arity 16, 50 queries"
        out = {
            "metric": "proven RISC-V cycles/sec", "value": round(value, 1), "unit": "cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32 (BabyBear Montgomery)", "data": "synthetic",
            "config": {"workload": "S%d: one 2^%d-cycle segment proof, W=%s (accum/code/data) + %s" % (args.po2, args.po2, args.widths, shape_text),
                       "params": args.preset,
                       "entry_point": "rk_prove_session (on_device = 1 inputs, verify = %d)" % (0 if args.no_verify else 1),
                       "segments_per_gpu_per_step": 1, "segments_in_flight_per_gpu": inflight,
                       "total_segments": total_segments, "parallelism": "segment-parallel x%d" % world},
            "roofline": roofline,
            "pipeline": {"algorithmic_bytes_per_segment": algo["total"],
                         "hbm_frac_end_to_end": round(algo["total"] / per_step / 1e9 / HBM_PEAK_GBS, 5),
                         "serial_stage_ms": {k: round(v, 3) for k, v in serial_stage.items()},
                         "kernels_source": "serial calibration proofs (kernel durations per segment)",
                         "kernels_sum_ms": round(kernel_sum, 3), "kernels_invariants_ok": bool(inv_ok),
                         "kernels": kernels},
        }
        if elapsed_h2d is not None:
            out["value_with_h2d"] = round(cycles / elapsed_h2d, 1)
            out["ms_per_step_with_h2d"] = round(elapsed_h2d / max(my_steps, 1) * 1e3, 3)
            out["h2d_seals_identical"] = bool(same)
        if small is not None:
            out["segments_of_2^18_cycles"] = small
        if with_circuit:
            out["with_circuit"] = with_circuit
        out["config"]["circuit_stages"] = ("excluded from `value`: accum and the check evaluations are resident inputs (BASELINE.md's S20); "
                                           "see `with_circuit` (--circuit N) for the same session with eval_check inside the region")
        if seals:
            # every seal was verified inside the timed region unless --no-verify; check the last one here too
            out["seal_verified"] = verify_segment(segs[(my_steps - 1) % 2][0], seals[-1], params=blob) == 0
            out["seal_words"] = int(seals[-1].size)
        if not args.no_cpu and world == 1:
            # the cpu_baseline leg is the only place bench.py touches oracle/ (test infrastructure):
            # its optimised operator forms (oracle/or_fast.c), the same seal as the plain restatement
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            if sp1:
                oracle_lib.oracle_set_params(1)
            cores = oracle_lib.usable_cores()
            cpu_po2 = args.cpu_po2 if args.cpu_po2 is not None else args.po2
            cseg = seg_mod.synthetic_segment(cpu_po2, widths, seed=20240807, blowup_log2=blow)
            t1 = time.perf_counter()
            oracle_lib.oracle_prove(cseg, threads=cores, fast=True)
            dt = time.perf_counter() - t1
            cseg1 = seg_mod.synthetic_segment(args.cpu_po2_1t, widths, seed=20240807, blowup_log2=blow)
            t1 = time.perf_counter()
            oracle_lib.oracle_prove(cseg1, threads=1, fast=True)
            dt1 = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": round((1 << cpu_po2) / dt, 1), "unit": "cycles/s", "cores": cores, "kind": "port",
                "value_1_thread": round((1 << args.cpu_po2_1t) / dt1, 1),
                "sample": "one 2^%d-cycle segment of the same column layout proven by oracle/ with its AVX2 / table-driven "
                          "operator forms (or_fast.c; OpenMP, %d threads = the cores this process may use) in %.2f s; "
                          "1 thread: one 2^%d-cycle segment in %.2f s.  The risc0 binary itself cannot be built here "
                          "(no Rust toolchain)" % (cpu_po2, cores, dt, args.cpu_po2_1t, dt1),
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
