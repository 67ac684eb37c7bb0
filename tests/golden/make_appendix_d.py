#!/usr/bin/env python3
"""Transcribe the reference traces captured in SURVEY.md Appendix D into JSON fixtures.

The traces were produced by the real reference (Whisper v1.861 System/Hart driven directly) in the
survey session; they are the only outputs of the reference run loop available offline (Hart.cpp needs
Boost, which this image lacks, so the run loop itself is unbuildable here - DESIGN.md section 3).
This script only extracts text: program words come from the opcode column of the log itself.

  python tests/golden/make_appendix_d.py [SURVEY.md]   ->  tests/golden/appendix_d.json
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LINE = re.compile(r"^#(\d+) (\d+) +([MSU]) ([0-9a-f]{16}) +([0-9a-f]{4,8}) ([rfmcv]) ([0-9a-f]{16}) ([0-9a-f]{16}) ?(.*?)(  \+)?$")


def blocks(text):
    app = text[text.index("## Appendix D"):]
    return re.findall(r"```\n(.*?)```", app, flags=re.S)


def parse(block):
    recs, extra = [], []
    for ln in block.splitlines():
        m = LINE.match(ln.rstrip())
        if m:
            tag, hart, priv, pc, op, kind, key, val, txt, cont = m.groups()
            recs.append(dict(tag=int(tag), hart=int(hart), priv=priv, pc=int(pc, 16), op=op, kind=kind, key=int(key, 16),
                             val=int(val, 16), text=txt.rstrip(), cont=bool(cont)))
        else:
            extra.append(ln)
    return recs, extra


def main():
    survey = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "..", "SURVEY.md")
    bl = blocks(open(survey).read())
    fp_recs, fp_extra = parse(bl[0])
    # program image = opcode column of the records at their pcs (first trace only; all 4-byte)
    words = {}
    for r in fp_recs:
        if r["pc"] >= 0x80000000 and len(r["op"]) == 8:
            words[r["pc"]] = int(r["op"], 16)
    final = {}
    for ln in fp_extra:
        if ln.startswith("final:"):
            for k, v in re.findall(r"(\w+)=([0-9a-fx]+)", ln):
                final[k] = int(v, 16)
        m = re.match(r"Info: Executed (\d+) instructions and retired (\d+)", ln)
        if m:
            final["executed"], final["retired"] = int(m.group(1)), int(m.group(2))
    sv_recs, _ = parse(bl[2])
    # RVV sample: r / c lines parse as usual; the two v lines are elided in the capture ("head…tail", most significant byte first)
    rvv_recs, rvv_extra = parse(bl[1])
    vlines = []
    for ln in rvv_extra:
        m = re.match(r"^#(\d+) (\d+) +([MSU]) ([0-9a-f]{16}) +([0-9a-f]{8}) v ([0-9a-f]{2}) ([0-9a-f]+)…([0-9a-f]+) (.*)$", ln.rstrip())
        if m:
            vlines.append(dict(tag=int(m.group(1)), pc=int(m.group(4), 16), op=m.group(5), reg=int(m.group(6), 16), head=m.group(7), tail=m.group(8),
                               text=m.group(9)))
    # RV32 sample (System<uint32_t>, rv32imc): the RV32 column layout - 8-digit pc, 2-digit register, 8-digit value
    rv32_recs = []
    for ln in bl[3].splitlines():
        m = re.match(r"^#(\d+) (\d+) +([MSU]) ([0-9a-f]{8}) +([0-9a-f]{4,8}) ([rm]) ([0-9a-f]{2,8}) +([0-9a-f]{8})  (.*)$", ln.rstrip())
        if m:
            rv32_recs.append(dict(tag=int(m.group(1)), priv=m.group(3), pc=int(m.group(4), 16), op=m.group(5), kind=m.group(6), key=int(m.group(7), 16),
                                  val=int(m.group(8), 16), text=m.group(9).strip(), line=ln.rstrip()))
    rv32_final = {}
    for ln in bl[3].splitlines():
        if "final:" in ln:
            for k, v in re.findall(r"(\w+)=([0-9a-f]+)", ln[ln.index("final:"):]):
                rv32_final[k] = int(v, 16)
    out = {
        "source": "SURVEY.md Appendix D (reference Whisper v1.861, captured by the survey session)",
        "fp_trap": {"isa": "rv64imafdcv_zicsr", "base": 0x80000000, "words": [words[a] for a in sorted(words)],
                    "records": fp_recs, "final": final,
                    "stop": "Failed stop: Hart 0: 16 consecutive illegal instructions"},
        "sv39_first": {"isa": "rv64imafdcsu_zicsr", "records": [r for r in sv_recs if r["tag"] == 1],
                       "pokes": {"mstatus_mpp": 1, "mepc": 0x40000100}, "next_pc": 0x40000100, "next_priv": "S"},
        # RVV sample (default VLEN = 512): csrrs x0, mstatus, x5 (= 0x2200) precedes it as #3; x5 is poked here
        "rvv": {"isa": "rv64imafdcv_zicsr", "records": rvv_recs, "vlines": vlines, "x5": 0x2200, "first_pc": 0x80000008},
        # the whole Sv39 sample: set-up as the survey's caption states it (root PTE[1] = 0x200000cf maps VA 0x40000000 -> PA
        # 0x80000000 as a 1 GiB V R W X A D leaf; satp = 8<<60 | 0x80002; pmpaddr0 = ~0, pmpcfg0 = 0x1f; tohost = PA 0x80003000)
        "sv39": {"isa": "rv64imafdcsu_zicsr", "records": sv_recs,
                 "pokes": {"mstatus_mpp": 1, "mepc": 0x40000100, "satp": (8 << 60) | 0x80002, "pmpaddr0": (1 << 54) - 1, "pmpcfg0": 0x1f},
                 "mem": {"0x80002008": 0x200000cf}, "tohost": 0x80003000, "va_to_pa": 0x40000000,
                 "stop": "Successful stop: Hart 0: write to to-host"},
        # RV32 sample: 16-bit parcels as the survey's caption lists them, poked at 0x80000000; the store goes to tohost = 0x1000
        "rv32": {"isa": "rv32imc", "base": 0x80000000, "parcels": [0x4505, 0x4589, 0x952e, 0x12b7, 0x0000, 0x4305, 0xa023, 0x0062],
                 "tohost": 0x1000, "records": rv32_recs, "final": rv32_final, "stop": "Successful stop: Hart 0: write to to-host"},
    }
    with open(os.path.join(HERE, "appendix_d.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("records:", len(fp_recs), "words:", [hex(w) for w in out["fp_trap"]["words"]], "final:", final)


if __name__ == "__main__":
    main()
