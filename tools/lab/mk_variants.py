"""Timing variants of the hand-written loop for tools/asm_lab.hip (tuning aid; variants may be functionally wrong on purpose:
they answer 'what does this piece cost')."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
src = open(os.path.join(ROOT, "whisper_amd/csrc/wsp_conv_core.h")).read()

def sub(s, old, new, count=1):
    assert old in s, old
    return s.replace(old, new, count)

V = {}
V["v0"] = src
# no register-index mode on the operand reads (reads x12 / x13 whatever the record says)
ISS_OLD = '''  "s_set_gpr_idx_on " Q4 ", gpr_idx(SRC0)\\n"                                            \\
  "v_mov_b32 v194, v128\\n"                                                              \\
  "v_mov_b32 v195, v129\\n"                                                              \\
  "s_set_gpr_idx_idx " Q5 "\\n"                                                          \\
  "v_mov_b32 v196, v128\\n"                                                              \\
  "v_mov_b32 v197, v129\\n"                                                              \\
  "s_set_gpr_idx_off\\n"                                                                 \\
'''
ISS_FIXED = '''  "v_mov_b32 v194, v152\\n"                                                              \\
  "v_mov_b32 v195, v153\\n"                                                              \\
  "v_mov_b32 v196, v154\\n"                                                              \\
  "v_mov_b32 v197, v155\\n"                                                              \\
'''
V["v_noidx_read"] = sub(src, ISS_OLD, ISS_FIXED)
WR_OLD = '''  "s_set_gpr_idx_on s38, gpr_idx(DST)\\n"                                                \\
  "v_mov_b32 v128, v200\\n"                                                              \\
  "v_mov_b32 v129, v201\\n"                                                              \\
  "s_set_gpr_idx_off\\n"
#define CC_RET'''
WR_FIXED = '''  "v_mov_b32 v160, v200\\n"                                                              \\
  "v_mov_b32 v161, v201\\n"
#define CC_RET'''
V["v_noidx_wr"] = sub(src, WR_OLD, WR_FIXED)
V["v_noidx_both"] = sub(V["v_noidx_read"], WR_OLD, WR_FIXED)
# no s_load of the successor (reuses whatever is in the buffer: only valid for timing of uniform streams)
pass
for k in list(V):
    if V[k] is None: del V[k]
# no wait for the successor's record at the end of a converged step (timing only: stale records)
V["v_nowait"] = sub(src, '''  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \\
  "s_waitcnt lgkmcnt(0)\\n"                                                              \\
  "s_mov_b32 s34, s35\\n"''', '''  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \\
  "s_mov_b32 s34, s35\\n"''')
# the add handler inline in the step: no dispatch, no return (timing of the two taken branches)
V["v_inline_add"] = sub(src, '''  "s_swappc_b64 s[54:55], s[90:91]\\n"''', '''  "v_lshl_add_u64 v[200:201], v[194:195], 0, v[198:199]\\n" "s_set_gpr_idx_on s38, gpr_idx(DST)\\n" "v_mov_b32 v128, v200\\n" "v_mov_b32 v129, v201\\n" "s_set_gpr_idx_off\\n"''')
V["v_inline_add_nowait"] = sub(V["v_inline_add"], '''  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \\
  "s_waitcnt lgkmcnt(0)\\n"                                                              \\
  "s_mov_b32 s34, s35\\n"''', '''  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \\
  "s_mov_b32 s34, s35\\n"''')
NOW_OLD = '''  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \\
  "s_waitcnt lgkmcnt(0)\\n"                                                              \\
  "s_mov_b32 s34, s35\\n"'''
V["v_snop8"] = sub(src, NOW_OLD, NOW_OLD + '''  "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n" "s_nop 0\\n"''')
V["v_vmov8"] = sub(src, NOW_OLD, NOW_OLD + '''  "v_mov_b32 v202, v203\\n" "v_mov_b32 v204, v203\\n" "v_mov_b32 v205, v203\\n" "v_mov_b32 v206, v203\\n" "v_mov_b32 v207, v203\\n" "v_mov_b32 v208, v203\\n" "v_mov_b32 v209, v203\\n" "v_mov_b32 v238, v203\\n"''')
V["v_smov8"] = sub(src, NOW_OLD, NOW_OLD + '''  "s_mov_b32 s88, s89\\n" "s_mov_b32 s92, s89\\n" "s_mov_b32 s93, s89\\n" "s_mov_b32 s94, s89\\n" "s_mov_b32 s95, s89\\n" "s_mov_b32 s88, s89\\n" "s_mov_b32 s92, s89\\n" "s_mov_b32 s93, s89\\n"''')
V["v_noload"] = sub(V["v_nowait"], '''  "s_load_dwordx8 " QN ", s[80:81], s88\\n"''', '''  "s_nop 0\\n"''')
for k, text in V.items():
    path = os.path.join(HERE, k + ".h")
    open(path, "w").write(text)
    if "--build" in sys.argv:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", f'-DLAB_CORE="{path}"',
                               os.path.join(ROOT, "tools/asm_lab.hip"), "-o", os.path.join(HERE, k + ".bin")], stderr=subprocess.DEVNULL)
        print("built", k)
