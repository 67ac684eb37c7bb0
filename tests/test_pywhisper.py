"""Row f4: the pywhisper-shaped Python surface (whisper_amd/pywhisper.py) against the reference's pybind11 module as
py-bindings.cpp:74-486 defines it - class / method / attribute names, the shape of step()'s result - and, on the GPU, against the
oracle: stepping a program instruction by instruction reproduces the oracle's commit log change for change."""
import json

import pytest

import simutil as S
import whisper_amd as W
from whisper_amd import pywhisper, workloads as WL


def test_module_layout_matches_py_bindings():
    assert pywhisper.system.System64 and pywhisper.system.System32
    assert pywhisper.system.hart.Hart64 and pywhisper.system.hart.Hart32 and pywhisper.system.memory.Memory
    assert pywhisper.decoded_inst.DecodedInst and pywhisper.file.open and pywhisper.file.close
    assert int(pywhisper.enums.PrivilegeMode.Machine) == 3 and pywhisper.enums.Supervisor == 1 and pywhisper.enums.User == 0
    for name in ("step", "run", "disass_inst", "__getattr__", "__setattr__"):
        assert hasattr(pywhisper.system.hart.Hart64, name)
    for name in ("load_hex_files", "load_bin_files", "load_elf_files", "harts", "memory"):
        assert hasattr(pywhisper.system.System64, name)
    for name in ("peek8", "peek16", "peek32", "peek64", "poke8", "poke16", "poke32", "poke64"):
        assert hasattr(pywhisper.system.memory.Memory, name)
    for name in ("inst", "address", "phys_address", "ith_operand", "ith_operand_type", "ith_operand_mode"):
        assert hasattr(pywhisper.decoded_inst.DecodedInst, name)
    names = pywhisper._csr_names()
    assert names["mstatus"] == 0x300 and names["satp"] == 0x180 and names["fcsr"] == 3 and names["mhartid"] == 0xF14


@pytest.mark.gpu
def test_step_run_and_attributes_against_the_oracle(tmp_path):
    p = WL.fuzz_program(12, n=120)                      # ends with a tohost write (Hart.run has no instruction limit of its own)
    cfg = tmp_path / "cfg.json"; cfg.write_text(json.dumps({"isa": p.isa, "xlen": 64}))
    elf = tmp_path / "p.elf"; elf.write_bytes(p.elf)
    with pytest.raises(ValueError):
        pywhisper.system.System64(str(tmp_path / "missing.json"))
    with pytest.raises(ValueError):
        pywhisper.system.System64(str(cfg), core_count=2, memory_size=p.mem_size, memory_base=p.mem_base)
    sy = pywhisper.system.System64(str(cfg), memory_size=p.mem_size, memory_base=p.mem_base, instances=2)
    try:
        assert sy.load_elf_files([str(elf)])
        h0, h1 = sy.harts()
        assert type(h0).__name__ == "Hart64" and h0.pc == p.entry and h0.priv_mode == pywhisper.enums.Machine and h0.virt_mode is False
        ro, oh, olog = S.run_oracle(p, log_cap=100000)
        names = {v: k for k, v in pywhisper._csr_names().items()}
        finished, n = False, 0
        while not finished and n < len(olog):
            finished, di, changes = h0.step()
            c = olog[n]
            pc, inst, pairs = c.pc, c.inst, c.changes()
            assert di.address() == pc and di.inst() == inst, (n, hex(pc))
            want = ["x%d" % i for k, i, v in pairs if k == 1 and i > 0] + ["f%d" % i for k, i, v in pairs if k == 2] + \
                   [names[i] for k, i, v in pairs if k == 5]
            got = [c for c in changes if isinstance(c, str)]
            assert got == want, (n, hex(pc), got, want)
            mem = [c for c in changes if isinstance(c, tuple)]
            assert [int(m[0][1:]) for m in mem] == [i for k, i, v in pairs if k == 4], (n, hex(pc))
            n += 1
        assert n == len(olog) == ro["exec"]
        assert finished == (ro["status"] == W.ST_STOP_TOHOST)
        assert h0.pc == ro["pc"] and h0.int_regs == [oh.x(i) for i in range(32)] and h0.fp_regs == [oh.f(i) for i in range(32)]
        assert h0.mstatus == oh.csr(0x300) and h0.cs_regs["mepc"] == oh.csr(0x341)
        # the second instance was not touched by the stepping; run() takes it to the same end
        assert h1.pc == p.entry
        h1.run()
        assert h1.pc == ro["pc"] and h1.int_regs == h0.int_regs
        # attribute writes, type errors, memory
        h1.x5 = 0x1234; h1.a0 = 7; h1.f3 = 0x3ff0000000000000; h1.mscratch = 0xabc; h1.pc = p.entry
        assert (h1.t0, h1.x10, h1.f3, h1.mscratch, h1.pc) == (0x1234, 7, 0x3ff0000000000000, 0xabc, p.entry)
        with pytest.raises(TypeError):
            h1.x5 = "1"
        with pytest.raises(AttributeError):
            h1.nonsense
        with pytest.raises(AttributeError):
            h1.nonsense = 1
        m = sy.memory(1)
        m.poke32(p.mem_base + 0x100, 0xdeadbeef); m.poke8(p.mem_base + 0x104, 0x7f)
        assert m.peek32(p.mem_base + 0x100) == 0xdeadbeef and m.peek16(p.mem_base + 0x102) == 0xdead and m.peek64(p.mem_base + 0x100) & 0xffffffffff == 0x7fdeadbeef
        with pytest.raises(RuntimeError):
            m.peek32(p.mem_base - 4)
        assert h0.disass_inst(0x00a58533) == W.disassemble(0x00a58533)
    finally:
        sy.close()


@pytest.mark.gpu
def test_hex_and_binary_loaders(tmp_path):
    p = WL.readme_program()
    cfg = tmp_path / "cfg.json"; cfg.write_text(json.dumps({"isa": p.isa}))
    hexf, binf = tmp_path / "p.hex", tmp_path / "p.bin"
    with open(hexf, "w") as f:
        for pa, data, _ in p.segments:
            f.write("@%x\n" % pa)
            for k in range(0, len(data), 16):
                f.write(" ".join("%02x" % x for x in data[k:k + 16]) + "\n")
    binf.write_bytes(p.segments[0][1])
    ro, _, _ = S.run_oracle(p)
    for how in ("hex", "bin"):
        sy = pywhisper.system.System64(str(cfg), memory_size=p.mem_size, memory_base=p.mem_base)
        try:
            if how == "hex":
                assert sy.load_hex_files([str(hexf)])
            else:
                assert sy.load_bin_files([str(binf)], p.segments[0][0])
            h = sy.harts()[0]
            h.pc = p.entry
            sy._batch.set_stop(tohost=p.tohost)
            h.run()
            assert h.pc == ro["pc"] and h.int_regs[10] == 3
        finally:
            sy.close()
