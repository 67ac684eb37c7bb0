"""Host-side formatter for the device commit ring (Batch.trace): prints records in the column layout of the reference's
--logfile (printTrace.cpp:380-495, formatInstTrace):

    #<tag> <hart>  <priv> <pc:016x> <inst:08x> <kind> <key:016x> <value:016x> <disassembly>[  +]

one line per changed resource, all but the last of a record ending in "  +".  The library carries no disassembler: pass
`disasm` (a callable inst -> str) to fill the last column, otherwise it is left empty.  Vector-register lines carry the
digest of the register the ring holds, not its bytes (read those with Batch.peek(hart, RES_V, reg))."""

KIND_CHAR = {1: "r", 2: "f", 3: "v", 4: "m", 5: "c"}
PRIV_CHAR = {0: "U", 1: "S", 3: "M"}


def format_record(tag, hart, rec, disasm=None):
    pc, inst, size, priv, trap, pairs = rec
    op = f"{inst:08x}" if size == 4 else f"{inst & 0xffff:04x}"
    text = disasm(inst) if disasm else ""
    lines = []
    for k, (kind, key, val) in enumerate(pairs):
        more = "  +" if k + 1 < len(pairs) else ""
        lines.append(f"#{tag} {hart}  {PRIV_CHAR.get(priv, '?')} {pc:016x} {op} {KIND_CHAR[kind]} {key:016x} {val:016x} {text}{more}".rstrip())
    return lines


def format_log(records, hart=0, first_tag=1, disasm=None):
    out = []
    for i, rec in enumerate(records):
        out.extend(format_record(first_tag + i, hart, rec, disasm))
    return out
