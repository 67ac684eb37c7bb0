"""`whisper`-compatible command line over the batch C ABI (the subset of /root/reference/Args.cpp / README.md:259-403 that drives
the execution path): what arch_test_target/riscof_whisper.py:152 invokes - `whisper --isa=<s> --testsignature=<f> --target=<elf>` -
works unchanged, and one invocation can run many ELFs at once, one hart each, on the GPU.

    python -m whisper_amd.cli --isa rv64imc --target prog.elf [--configfile cfg.json] [--maxinst N] [--tohost A] [--startpc A]
                              [--endpc A] [--setreg x5=0x10 ...] [--memorysize N] [--logfile F] [--testsignature F] [--harts N]
    python -m whisper_amd.cli --isa rv64imc --batch list.txt      # lines: <elf> [<signature file> [<log file>]]

Exit status as whisper.cpp:118: 0 iff every hart's run succeeded (a tohost stop with value >> 1 == 0, the stop address, or - unless
--failoninstlimit - the instruction limit, Hart.cpp:6224-6231; an UNSUPPORTED or consecutive-illegal stop fails).
--harts N replicates the one target over N independent harts (mhartid 0..N-1): the batch's reading of the flag, since shared-memory
multi-hart systems are out of scope."""
from __future__ import annotations

import argparse
import sys

import whisper_amd as W
from whisper_amd import elfio, tracefmt


def _num(s):
    return int(s, 0)


def build_parser():
    ap = argparse.ArgumentParser(prog="whisper_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("--isa", default=None)
    ap.add_argument("--xlen", type=int, default=0)
    ap.add_argument("--configfile", default=None)
    ap.add_argument("--target", "-t", action="append", default=[], help="ELF file (repeat for one hart each)")
    ap.add_argument("--batch", default=None, help="file with one '<elf> [<signature> [<log>]]' per line")
    ap.add_argument("--tohost", type=_num, default=None)
    ap.add_argument("--maxinst", "-m", type=_num, default=None)
    ap.add_argument("--maxretinst", type=_num, default=None)
    ap.add_argument("--startpc", "-s", type=_num, default=None)
    ap.add_argument("--endpc", "-e", type=_num, default=None)
    ap.add_argument("--setreg", action="append", default=[], help="x<n>=<value>, repeatable")
    ap.add_argument("--memorysize", type=_num, default=0)
    ap.add_argument("--logfile", "-f", default=None)
    ap.add_argument("--logsize", type=_num, default=1 << 20, help="commit-ring words per hart behind --logfile (a record takes 5 .. 40)")
    ap.add_argument("--testsignature", default=None)
    ap.add_argument("--harts", type=int, default=1)
    ap.add_argument("--failoninstlimit", action="store_true")
    ap.add_argument("--tlbsize", type=_num, default=0)
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--quiet", action="store_true")
    return ap


def arena_for(elfs, memorysize=0):
    """One arena for every hart of the batch: from the lowest loaded page, a power of two that holds the highest one plus slack
    for a stack (the reference's flat 4 GiB default memory, System.cpp:50, cut down to what the programs touch)."""
    lo, hi = None, 0
    for e in elfs:
        _, segs, _ = elfio.read_elf(e)
        for pa, data, _ in segs:
            lo = pa if lo is None else min(lo, pa)
            hi = max(hi, pa + len(data))
    if lo is None:
        raise SystemExit("Error: no loadable segment in the target(s)")
    base = lo & ~0xfff
    size = 1 << 16
    while base + size < hi + (16 << 10):
        size <<= 1
    return base, max(size, (memorysize + 0xfff) & ~0xfff)


def write_signature(path, words):
    """System::produceTestSignatureFile (System.cpp:1586-1633): one 32-bit word per line, 8 hex digits."""
    with open(path, "w") as f:
        for w in words:
            f.write(f"{w:08x}\n")


def _with_hart_column(text, hart):
    """The log's second column is the hart's index in its system (printTrace.cpp:71: `#tag hart`); the library formats every hart
    as hart 0 of a system of its own, --harts N makes them hart 0 .. N-1 of one target."""
    if hart == 0:
        return text
    out = []
    for line in text.splitlines(True):
        if line.startswith("#"):
            tag, _, rest = line.partition(" ")
            rest = rest.partition(" ")[2]
            line = f"{tag} {hart} {rest}"
        out.append(line)
    return "".join(out)


def run_entries(entries, isa=None, config=None, xlen=0, tohost=None, maxinst=None, maxret=None, startpc=None, endpc=None, setreg=(),
                memorysize=0, harts_per_entry=1, tlbsize=0, device=-1, fail_on_limit=False, quiet=False, disasm=None, logsize=1 << 20):
    """entries: [(elf_path, signature_path | None, log_path | None)].  Runs them all in one batch; returns [ok per entry]."""
    elfs = [open(e[0], "rb").read() for e in entries]
    base, size = arena_for(elfs, memorysize)
    n = len(entries) * harts_per_entry
    want_log = any(e[2] for e in entries)
    b = W.Batch(n, isa=isa, mem_base=base, mem_size=size, device=device, tlb_entries=tlbsize, config=config, xlen=xlen)
    try:
        b.load_elfs(elfs, harts_per_elf=harts_per_entry)
        for spec in setreg:
            reg, _, val = spec.partition("=")
            if not reg.startswith("x"):
                raise SystemExit(f"Error: --setreg {spec}: only integer registers (x<n>=<value>)")
            for h in range(n):
                b.poke(h, W.RES_X, int(reg[1:]), _num(val))
        if startpc is not None:
            for h in range(n):
                b.poke(h, W.RES_PC, 0, startpc)
        b.set_stop(tohost=tohost, max_inst=W.NO_LIMIT if maxinst is None else maxinst, max_ret=W.NO_LIMIT if maxret is None else maxret,
                   stop_pc=endpc)
        if want_log:
            b.trace_ring(logsize)
        left = b.run(trace_hash=want_log)
        res = b.results()
        oks = []
        for i, (elf, sig, log) in enumerate(entries):
            ok = True
            for h in range(i * harts_per_entry, (i + 1) * harts_per_entry):
                r = res[h]
                good = (r.status == W.ST_STOP_TOHOST and (r.stop_value >> 1) == 0) or r.status == W.ST_STOP_ADDR or \
                       (r.status == W.ST_STOP_LIMIT and not fail_on_limit)
                ok = ok and good and left == 0
                if not quiet:
                    what = {W.ST_STOP_TOHOST: "Successful stop: write to to-host" if good else f"Failed stop: write to to-host ({r.stop_value:#x})",
                            W.ST_STOP_LIMIT: "Stopped -- Reached instruction limit", W.ST_STOP_ADDR: "Stopped -- Reached end address",
                            W.ST_STOP_ILLEGAL: "Failed stop: 64 consecutive illegal instructions",
                            W.ST_STOP_UNSUPPORTED: f"Failed stop: left the modelled subset ({r.stop_value:#x})"}.get(r.status, "Running")
                    print(f"Info: {what} hart={h}  Retired {r.retired_count} instructions  ({elf})", file=sys.stderr)
            if sig:
                write_signature(sig, b.signature(i * harts_per_entry))
            if log:
                with open(log, "w") as f:
                    for k in range(harts_per_entry):           # the reference's --logfile text (wsp_batch_trace_text)
                        h = i * harts_per_entry + k
                        text = b.trace_text(h)
                        got = sum(1 for ln in text.splitlines() if ln.startswith("#") and not ln.rstrip().endswith("+"))
                        if got < res[h].exec_count:
                            print(f"Warning: {log}: hart {h}: the commit ring held {got} of {res[h].exec_count} instructions "
                                  f"(--logsize {logsize} words per hart): the log ends early", file=sys.stderr)
                        f.write(_with_hart_column(text, k))
            oks.append(ok)
        return oks
    finally:
        b.close()


def main(argv=None):
    a = build_parser().parse_args(argv)
    entries = [(t, a.testsignature if len(a.target) == 1 else None, a.logfile if len(a.target) == 1 else None) for t in a.target]
    if a.batch:
        for line in open(a.batch):
            parts = line.split()
            if parts and not parts[0].startswith("#"):
                entries.append((parts[0], parts[1] if len(parts) > 1 else None, parts[2] if len(parts) > 2 else None))
    if not entries:
        print("Error: no target program (--target / --batch)", file=sys.stderr)
        return 1
    config = open(a.configfile).read() if a.configfile else None
    if a.isa is None and config is None:
        a.isa = "rv64imafdc_zicsr" if a.xlen != 32 else "rv32imac_zicsr"
    try:
        oks = run_entries(entries, isa=a.isa, config=config, xlen=a.xlen, tohost=a.tohost, maxinst=a.maxinst, maxret=a.maxretinst,
                          startpc=a.startpc, endpc=a.endpc, setreg=a.setreg, memorysize=a.memorysize, harts_per_entry=a.harts, logsize=a.logsize,
                          tlbsize=a.tlbsize, device=a.device, fail_on_limit=a.failoninstlimit, quiet=a.quiet)
    except W.WspError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 1
    return 0 if all(oks) else 1


if __name__ == "__main__":
    sys.exit(main())
