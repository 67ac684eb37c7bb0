import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import whisper_amd as W
import test_config_stop as T
import simutil as S
p = T._lr_trap_sc_program()
cfg = {"isa": p.isa, "memmap": {"size": hex(p.mem_size)}, "reset_vec": hex(p.entry + 0), "cancel_lr_on_trap": True}
for n, lanes in ((3, 0), (3, 64), (1, 0), (40, 0)):
    b = W.Batch(n, isa=None, mem_base=p.mem_base, mem_size=0, config=json.dumps(cfg), lanes_per_wave=lanes)
    b.load_elf_bytes(p.elf)
    left = b.run()
    res = b.results()
    print(n, lanes, left, [(r.status, hex(r.pc), r.exec_count, b.peek(i, W.RES_X, 11)) for i, r in enumerate(res)][:4])
    b.close()
ro, _, _ = S.run_oracle(p, features=T.F_CANCEL_LR)
print(ro)
