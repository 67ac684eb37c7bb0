"""Host-side formatter for the device commit ring (Batch.trace): prints records in the column layout of the reference's
--logfile (printTrace.cpp:380-495, formatInstTrace):

    #<tag> <hart>  <priv> <pc:016x> <inst:08x> <kind> <key:016x> <value:016x> <disassembly>[  +]

one line per changed resource, all but the last of a record ending in "  +".  Kept for callers that post-process the records
in Python; the product formatter is the library's own wsp_batch_trace_text (Batch.trace_text), which prints the complete
reference text: disassembly column (wsp_disassemble), load / store address suffixes, vector register bytes, both widths."""

KIND_CHAR = {1: "r", 2: "f", 3: "v", 4: "m", 5: "c"}
PRIV_CHAR = {0: "U", 1: "S", 3: "M"}


def format_record(tag, hart, rec, disasm=None):
    pc, inst, size, priv, trap, pairs = rec
    op = f"{inst:08x}" if size == 4 else f"{inst & 0xffff:04x}"
    text = disasm(inst) if disasm else ""
    # printTrace.cpp:338-346 appends the data address of scalar loads / stores; the ring carries it for stores (the m change)
    # only, so load lines lack the suffix.  Traps and interrupts are marked as the reference marks them.
    for kind, key, _ in pairs:
        if kind == 4:
            text += f" [0x{key:x}]"
    text += " (exception)" if trap == 1 else " (interrupted)" if trap == 2 else ""
    lines = []
    for k, (kind, key, val) in enumerate(pairs):
        more = "  +" if k + 1 < len(pairs) else ""
        lines.append(f"#{tag} {hart}  {PRIV_CHAR.get(priv, '?')} {pc:016x} {op:>8s} {KIND_CHAR[kind]} {key:016x} {val:016x} {text}".rstrip() + more)
    return lines


def format_log(records, hart=0, first_tag=1, disasm=None):
    out = []
    for i, rec in enumerate(records):
        out.extend(format_record(first_tag + i, hart, rec, disasm))
    return out
