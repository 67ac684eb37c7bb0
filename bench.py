#!/usr/bin/env python3
"""bench.py -- aggregate simulated instructions/sec of the batched RISC-V hart path on MI355X.

Workload (BASELINE.json configs[2]): 4096 RV64IMAFD CoreMark-style compute-kernel instances per
GPU, per-instance seed, each run to its tohost stop.  One "step" = rewind the batch to its loaded
state (device snapshot restore), run every hart to completion (wsp_batch_run), compute the
per-hart state/memory digests (wsp_batch_results) and, for N > 1, all-gather the result records
over RCCL.  value = instructions executed by all harts of all ranks / wall time (weak scaling:
every rank runs its own 4096 instances with its own seeds).  After the timed steps the same steps
run once more with the commit-trace digest on (`value_traced`: north_star's "trace hashes on every
instance").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Labelled diagnostics (not the BASELINE headline): --workload sv39 | fuzz_sv39 | rvv; --split strong cuts ONE list of
--harts programs over the N ranks (whisper_amd.shard.partition) instead of giving every rank its own --harts.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_INST = 8.0        # SURVEY.md 8(d) model: 4 B fetch + ~0.3 x 8 B data + slice state swap, rounded up
N_SIMD, CLK_HZ = 256 * 4, 2.4e9  # MI355X: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 clocks
PROFILE_TAG = "r03_d"            # profiles/<tag>_pmc_per_launch.json + <tag>_bench.json: rocprofv3 --pmc passes of this same command
PMC_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_per_launch.json")
CSRC = [os.path.join(ROOT, "whisper_amd", "csrc", f) for f in ("wsp_lib.hip", "wsp_core.h", "wsp_conv_core.h", "wsp_conv_handlers.inc", "wsp_softfloat.h")]


def csrc_digest():
    """Digest of the kernel sources: a committed counter profile describes the kernel only while this is unchanged."""
    h = hashlib.sha256()
    for f in CSRC:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_profile(insts_per_launch):
    """Per-launch figures of wsp_step_kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ
    instruction counts: separate runs of this same command).  Counters cannot be read from inside the bench, so the
    profiled per-simulated-instruction figures are scaled to this launch.  None when no profile is committed or when the
    kernel sources have changed since it was taken (the profile records csrc_digest())."""
    try:
        allk = json.load(open(PMC_FILE))
        if allk.get("_csrc_digest") != csrc_digest():
            return None
        rec = allk["void wsp_step_kernel<false, true>"]
        ref = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_bench.json")))["config"]["instructions_per_step_per_gpu"]
        k = insts_per_launch / ref
        return {"traffic": (rec["FETCH_SIZE_per_launch"] + rec["WRITE_SIZE_per_launch"]) * 1024.0 * k,
                "valu_per_inst": rec["SQ_INSTS_VALU_per_launch"] / ref, "salu_per_inst": rec["SQ_INSTS_SALU_per_launch"] / ref,
                "branch_per_inst": rec["SQ_INSTS_BRANCH_per_launch"] / ref, "lds_per_inst": rec["SQ_INSTS_LDS_per_launch"] / ref,
                "vmem_per_inst": rec["SQ_INSTS_VMEM_per_launch"] / ref, "commit": allk.get("_commit")}
    except Exception:
        return None


def _cpu_worker(args):
    """One oracle process: run instances until the deadline; returns (instructions, seconds)."""
    idx, iters, budget_s = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import rvoracle
    from whisper_amd import workloads as WL
    prog = WL.compute_kernel(iters=iters)
    n = 0
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s:
        i = idx * 100000 + k
        h = rvoracle.Hart(isa=prog.isa, mem_base=prog.mem_base, mem_size=prog.mem_size, tohost=prog.tohost, reset_pc=prog.entry, hart_id=i)
        for paddr, data, _ in prog.segments:
            h.write_mem(paddr, data)
        h.set_x(10, WL.compute_seed(i))
        h.write_mem(prog.symbols["seed"], i.to_bytes(8, "little"))
        h.run(prog.max_inst)
        n += h.exec_count
        k += 1
    return n, time.perf_counter() - t0, k


def host_cores():
    """(workers to start, CPUs the machine shows).  The GPU box shows the whole machine's CPUs to every tenant but
    gives a one-GPU job a share of them, exported as OMP_NUM_THREADS; without that variable every CPU this process
    may run on is used."""
    shown = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = shown
    share = os.environ.get("OMP_NUM_THREADS")
    return max(1, min(allowed, int(share)) if share and share.isdigit() else allowed), shown


def cpu_baseline(iters, budget_s=12.0):
    """The CPU oracle (a scalar port of the reference's run loop, reference SoftFloat inside) on the
    host cores of this box: one process per core, same compute-kernel instances.  Must run before
    this process touches HIP (it forks)."""
    import multiprocessing as mp
    cores, shown = host_cores()
    with mp.get_context("fork").Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(i, iters, budget_s) for i in range(cores)])
    rate = sum(n / s for n, s, _ in out)
    return {"value": rate / 1e9, "unit": "GIPS", "cores": cores, "kind": "port",
            "host_cpus_shown": shown, "per_core_mips": rate / cores / 1e6,
            "whole_host_linear_extrapolation_gips": rate / cores * shown / 1e9,
            "reference_note": "the reference's own Hart::run cannot be built here (Boost headers absent, DESIGN.md section 4), so no "
                              "reference/port ratio is measured; SURVEY.md section 6 measured the reference at 61.8 M inst/s/core "
                              "(integer loop) and 11.4 M (50 % SoftFloat) in the survey container",
            "sample": f"{sum(k for _, _, k in out)} compute-kernel instances (iters={iters}) in {budget_s:.0f} s, "
                      f"{sum(n for n, _, _ in out)} instructions, one oracle process per worker ({cores} workers = this job's CPU share; "
                      f"the machine shows {shown} CPUs), untraced-equivalent"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run before this
    process touches a GPU, pass their output through and leave with their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


WORKLOAD_LABEL = {
    "compute": "BASELINE configs[2]: 4096 RV64IMAFD compute-kernel instances per GPU (CoreMark-style, per-instance seed), run to tohost, state+memory digests",
    "sv39": "DIAGNOSTIC (not the BASELINE headline): the compute kernel in S mode under Sv39 (PMP + page table set up by M-mode boot code)",
    "fuzz_sv39": "DIAGNOSTIC (not the BASELINE headline): BASELINE configs[4]'s per-GPU shape - every hart its own Sv39 fuzz program "
                 "(generated programs x code variants, 8-entry TLBs), --maxinst as generated",
    "rvv": "DIAGNOSTIC (not the BASELINE headline): BASELINE configs[3]'s shape - RVV kernels (fill, convert, saxpy / daxpy, integer pipeline, "
           "masked ops, dot products) on every hart, per-instance seed",
}


def build_batch(args, W, WL, shard, rank, world, local_rank):
    """Create and load this rank's batch.  Returns (batch, n harts of this rank, first global hart, oracle spot-check closure or None)."""
    strong = args.split == "strong"
    if strong:
        lo, hi = shard.partition(args.harts, world)[rank]          # one list of --harts systems cut over the ranks
    else:
        lo, hi = shard.weak_range(args.harts, rank)                # every rank its own --harts systems
    n, base = hi - lo, lo
    dev = local_rank if world > 1 else -1
    if args.workload in ("compute", "sv39"):
        prog = WL.compute_kernel(iters=args.iters, fp=not args.nofp, paged=args.workload == "sv39")
        b = W.Batch(n, isa=prog.isa, mem_base=prog.mem_base, mem_size=prog.mem_size, lanes_per_wave=args.lanes, device=dev, hart_id_base=base)
        b.load_elf_bytes(prog.elf)
        b.set_stop(max_inst=prog.max_inst)
        b.poke_each(W.RES_X, 10, [WL.compute_seed(base + i) for i in range(n)])
        b.poke_each(W.RES_MEM, prog.symbols["seed"], [base + i for i in range(n)])

        def spot(S, i):
            return S.run_oracle(prog, seed_a0=WL.compute_seed(base + i), seed_mem=base + i, hart_id=base + i)[0]
        return b, n, base, spot
    if args.workload == "rvv":
        prog = WL.rvv_kernel(n=1024, iters=args.iters, vlenb=args.vlenb)
        b = W.Batch(n, isa=prog.isa, mem_base=prog.mem_base, mem_size=prog.mem_size, lanes_per_wave=args.lanes, device=dev, hart_id_base=base,
                    vlen_bytes=args.vlenb)
        b.load_elf_bytes(prog.elf)
        b.set_stop(max_inst=prog.max_inst)
        b.poke_each(W.RES_X, 10, [WL.compute_seed(base + i) for i in range(n)])

        def spot(S, i):
            return S.run_oracle(prog, seed_a0=WL.compute_seed(base + i), hart_id=base + i)[0]
        return b, n, base, spot
    # fuzz_sv39: global program g = variant (g % nv) of generated program (g // nv)
    nv = 50
    bases = {}

    def prog_of(g):
        s = g // nv
        if s not in bases:
            bases[s] = WL.sv39_program(s)
        return WL.variant(bases[s], g % nv)
    p0 = prog_of(0)
    b = W.Batch(n, isa=p0.isa, mem_base=p0.mem_base, mem_size=p0.mem_size, lanes_per_wave=args.lanes, device=dev, hart_id_base=base,
                tlb_entries=p0.meta.get("tlb_entries", 0))
    chunk = 3200
    for c in range(0, n, chunk):
        b.load_elfs([prog_of(base + i).elf for i in range(c, min(n, c + chunk))], lo=c)
        for s in [s for s in bases if s < (base + c) // nv]:
            del bases[s]
    b.set_stop(max_inst=p0.max_inst)

    def spot(S, i):
        return S.run_oracle(prog_of(base + i), hart_id=base + i)[0]
    return b, n, base, spot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--harts", type=int, default=4096, help="instances per GPU (--split weak) or in total (--split strong)")
    ap.add_argument("--iters", type=int, default=16, help="outer iterations of the compute kernel (~56k instructions each)")
    ap.add_argument("--lanes", type=int, default=0, help="harts per wavefront (0 = auto)")
    ap.add_argument("--trace", action="store_true", help="the timed steps themselves fold the commit-trace digest (default: untraced steps, then a traced leg)")
    ap.add_argument("--no-traced-leg", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nofp", action="store_true", help="diagnostic: integer-only variant of the kernel (not the BASELINE config)")
    ap.add_argument("--vlenb", type=int, default=64, help="rvv: bytes per vector register")
    ap.add_argument("--workload", default="compute", choices=sorted(WORKLOAD_LABEL))
    ap.add_argument("--split", default="weak", choices=["weak", "strong"])
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))              # n_gpus in the line below is always the number of live ranks
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and args.workload == "compute":
        cpu = cpu_baseline(args.iters)          # before any HIP call (forks)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import whisper_amd as W
    from whisper_amd import shard
    from whisper_amd import workloads as WL

    gdev = torch.device("cuda", local_rank) if world > 1 else None
    b, n, base, spot = build_batch(args, W, WL, shard, rank, world, local_rank)
    b.snapshot()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    def run_leg(trace, steps, warmup):
        kernel_s = []

        def step(timed):
            b.restore()
            left = b.run(trace_hash=trace)
            res = b.results()
            gathered = shard.gather_results(bytes(res), dist, gdev) if dist is not None else None   # final gather of pass/fail + digests (RCCL)
            if left:
                raise RuntimeError(f"{left} harts still running")
            if timed:
                kernel_s.append(b.run_stats()[0])
            return res, gathered

        for _ in range(warmup):
            res, gathered = step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res, gathered = step(True)
        barrier()
        dt = time.perf_counter() - t0
        per_step = sum(r.exec_count for r in res)
        dt, tot = shard.reduce_totals(dt, float(per_step), dist, gdev)
        return dict(res=res, gathered=gathered, dt=dt, insts_all_ranks=tot, insts_this_rank=per_step, kernel_s=kernel_s)

    main_leg = run_leg(args.trace, args.steps, args.warmup)
    traced_leg = None
    if not args.trace and not args.no_traced_leg:
        traced_leg = run_leg(True, args.steps, min(1, args.warmup))
    res = main_leg["res"]
    per_step_insts = main_leg["insts_this_rank"]
    dt = main_leg["dt"]
    total_insts = main_leg["insts_all_ranks"] * args.steps
    if args.workload in ("compute", "sv39", "rvv"):
        ok = all(r.status == W.ST_STOP_TOHOST and r.stop_value == 1 for r in res)
    else:
        ok = all(r.status != W.ST_RUNNING for r in res)

    if rank == 0:
        # cheap end-to-end sanity against the oracle for a few harts of rank 0 (parity proper: tests/)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        parity = None
        try:
            import simutil as S
            parity = True
            for i in (0, 1, n - 1):
                ro = spot(S, i)
                rd = res[i].as_dict()
                if not args.trace:
                    ro = dict(ro); ro["trace_hash"] = 0
                parity = parity and (ro == rd)
                if traced_leg is not None:
                    parity = parity and (spot(S, i) == traced_leg["res"][i].as_dict())
        except Exception as e:   # oracle not built: report, do not fail the bench
            parity = f"not checked: {e}"
        kernel_s = main_leg["kernel_s"]
        kavg = sum(kernel_s) / max(1, len(kernel_s))
        gips = total_insts / dt / 1e9
        krate = per_step_insts / kavg if kavg > 0 else 0.0
        achieved = krate * ALGO_BYTES_PER_INST / 1e9
        hs, hb = b.hash_stats()
        pmc = pmc_profile(per_step_insts) if args.workload == "compute" and not args.trace else None
        roof = {"bound": "hbm", "kernel": "wsp_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic"] if pmc else None,
                "algorithmic_bytes_per_launch": per_step_insts * ALGO_BYTES_PER_INST,
                "algorithmic_model": "8 B per simulated instruction (SURVEY.md 8d): a model figure, small against HBM by construction",
                "kernel_ms": kavg * 1e3, "kernel_gips": krate / 1e9,
                "clocks_per_simulated_step_per_wave": (kavg * CLK_HZ / (per_step_insts / n)) if per_step_insts else None,
                "limiter": "dependent-issue latency of one wave's interpreter step (DESIGN.md section 7), not HBM",
                "memhash_kernel": {"achieved": hb / hs / 1e9 if hs > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": (hb / hs / 1e9 / HBM_PEAK_GBS) if hs > 0 else 0.0, "bytes": hb, "ms": hs * 1e3}}
        if pmc:
            # the primary efficiency figure.  Wave-level VALU instructions issued per second against the chip's
            # VALU issue slots (1024 SIMDs, one wave64 VALU instruction per 4 clocks each); SALU against one per clock per CU.
            roof.update({"measured_traffic_GBps": pmc["traffic"] / kavg / 1e9 if kavg > 0 else None,
                         "valu_issue_frac": pmc["valu_per_inst"] * krate / (N_SIMD * CLK_HZ / 4),
                         "salu_issue_frac": (pmc["salu_per_inst"] + pmc["branch_per_inst"]) * krate / (256 * CLK_HZ),
                         "wave_instructions_per_simulated_instruction": {k: pmc[k + "_per_inst"] for k in ("valu", "salu", "branch", "lds", "vmem")},
                         "traffic_note": f"per-launch HBM bytes and instruction mix scaled from profiles/{PROFILE_TAG}_pmc_per_launch.json "
                                         f"(rocprofv3 --pmc passes of this command at commit {pmc['commit']}, kernel sources unchanged since)"})
        else:
            roof["traffic_note"] = ("no counter profile of these kernel sources is committed (profiles/%s_pmc_per_launch.json is absent or was "
                                    "taken from other sources): counter-derived fields left out" % PROFILE_TAG)
        line = {
            "metric": "aggregate simulated instructions/sec (GIPS)", "value": gips, "unit": "GIPS",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.split, "vs_baseline": None, "dtype": "u64/softfloat-f64", "data": "synthetic",
            "config": {"workload": WORKLOAD_LABEL[args.workload],
                       "harts_per_gpu": n if args.split == "weak" else None, "harts_total": args.harts * (world if args.split == "weak" else 1),
                       "iters": args.iters,
                       "instructions_per_step_per_gpu": per_step_insts, "trace_hash": bool(args.trace), "lanes_per_wave": args.lanes,
                       "all_harts_pass": ok, "oracle_spot_check": parity, "csrc_digest": csrc_digest()},
            "roofline": roof,
        }
        if traced_leg is not None:
            tk = traced_leg["kernel_s"]
            line["value_traced"] = traced_leg["insts_all_ranks"] * args.steps / traced_leg["dt"] / 1e9
            line["traced"] = {"what": "the same steps with WSP_RUN_TRACE_HASH: every commit record folded into the per-hart trace digest",
                              "ms_per_step": traced_leg["dt"] / args.steps * 1e3, "kernel_ms": sum(tk) / max(1, len(tk)) * 1e3,
                              "kernel_gips": traced_leg["insts_this_rank"] / (sum(tk) / max(1, len(tk))) / 1e9 if tk else None}
        if main_leg["gathered"] is not None:
            line["config"]["gathered_records_per_rank"] = [len(x) // 64 for x in main_leg["gathered"]]
        if cpu is not None:
            line["cpu_baseline"] = cpu
            # not BASELINE.md's published number (there is none: vs_baseline stays null) - this run's own CPU leg
            line["vs_cpu_baseline"] = {"ratio": gips / cpu["value"] if cpu["value"] else None,
                                       "what": f"{args.gpus} GPU(s) / the oracle port on this job's {cpu['cores']} host cores (same instances)"}
        print(json.dumps(line))
    b.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
