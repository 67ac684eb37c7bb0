// The converged hot loop as hand-written CDNA (gfx950) code: one scheduled pc per wave, threaded dispatch, x registers in VGPRs.
//
// What it replaces: the C++ converged loop of wsp_step_kernel for untraced runs (same contract, see the comment there), which is
// the reference's Hart::untilAddress body (Hart.cpp:6025-6201: fetch, decode-cache hit, execute switch Hart.cpp:7544) for the
// inline subset.  Measured on the C++ loop (tools/ubench.py, 4096 harts = one wave per SIMD, profiles/r03_ubench_*.jsonl): an
// `add` step took 487 clocks for ~60 wave instructions, and every extra level of the compiler's compare-chain dispatch cost ~38
// clocks (`xor`, 15 levels down: 870 clocks) - with one wave per SIMD a taken branch is a full instruction-fetch restart and
// nothing hides it; an F / D instruction cost 7 000 clocks, almost all of it the way out of the loop and back.  So:
//
//  * threaded dispatch, two taken branches per simulated instruction: the loop reads its own 32-byte "hot records" (HRec,
//    wsp_core.h: handler offset, size, 64-bit immediate, VGPR indices of the operands, branch target - nothing left to decode),
//    handlers sit at a fixed 512-byte stride, and the dispatch is `s_add / s_addc / s_swappc_b64`; the handler returns with
//    `s_setpc_b64` into the step's epilogue, which falls through into the next step's code (three record buffers A / B / T in
//    rotation, so records are never copied);
//  * the x register file of every lane lives in VGPRs v[128:193] (slot s = x[s] for 1..31, slot 0 = write sink for rd = x0,
//    slot 32 = constant zero which reads of x0 and immediate forms are redirected to), indexed with s_set_gpr_idx_on by the
//    wave-uniform register numbers - no LDS round trip on the dependent chain; LDS <-> VGPR copies happen at entry and exit only;
//  * two modes.  Converged: every eligible lane stands at the scheduled pc - no membership test, no per-lane pc or retire-count
//    update (both are implied by the scalar state and written back when the mode is left).  Diverged: exactly the lanes whose pc
//    is the scheduled one execute; after every control transfer the minimum pc over all lanes is scheduled (min-pc
//    reconvergence), and the loop is back in converged mode as soon as all lanes agree;
//  * records come through the scalar cache, one step ahead; branches and jal request their target's record while the compare runs;
//  * F / D computational instructions, mulh* / div* / rem* and Zb* never leave the loop: their handlers marshal the operands and
//    call the compiled C++ semantics (wsp_pure_fp / wsp_pure_int in wsp_lib.hip: pure functions of their arguments) through the
//    standard calling convention.  Everything this loop keeps lives in registers that convention preserves or that the callees do
//    not reach (VGPRs >= 128: the build checks the callees' register counts; callee-saved SGPRs); the record buffers are re-read after a call;
//  * a step whose lanes are not all fine (access outside the arena, tohost, text, misaligned, target outside the text, an op
//    outside the inline subset, a reserved rounding mode) changes nothing for ANY lane of the group and ends the loop with ncv = 1
//    on those lanes: the general path executes that instruction (same contract as the C++ loop, coarser grain: rare events only).
//
//  * measured on gfx950 (tools/lab/valu_rate.hip): a wave that has its SIMD to itself issues a vector ALU instruction every
//    ~1.9 ns with 15 or more lanes enabled in exec, but only every ~8.6 ns with 4 or fewer - and a 4096-hart batch puts 4 harts in
//    a wave.  So the vector ALU always runs with EVERY lane of the wave enabled (lanes without a hart, and lanes outside the group
//    in diverged mode, compute on garbage); results are committed with v_cndmask under the mask of the lanes that execute
//    (s[84:85]: every hart's lane in converged mode, the group in diverged mode), exec is narrowed to that mask only around
//    memory and LDS instructions, and checks (`all lanes fine`, `who takes the branch`) are and-ed with it;
//
// gfx940-family wait states observed by hand (nothing inside an asm statement is padded by the compiler): VALU write of an
// SGPR / VCC -> VALU read of it: 2; VALU write of a VGPR -> v_readlane / v_readfirstlane / DPP read: 1 / 1 / 2.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)

// ---- fixed registers (all in the clobber list, together with everything a callee may clobber: s0-s31, vcc, the caller-saved VGPRs).
// The calling convention keeps s34-s39, s48-s55, s64-s71, s80-s87 (and s96+) across a call: what must survive the calls of
// wsp_pure_* lives there.  The record buffers and scratch sit in the blocks a callee may overwrite: after such a call the loop
// re-reads the next record (L_resume).
//  callee-saved: s34 upc  s35 spc  s36 steps left - 1  s37 its value when converged mode was entered  s38 record word 6  s39 word 7
//                s48 mode (0 converged, 1 diverged)  s49 cfg (bit 1 rv32, bit 2 FS dirty, bit 3 D)  s50 lpw shift  s51 f block offset
//                s52 isa  s53 text length  s[54:55] return address of the step  s[64:65] handler table
//                s[66:67] s[86:87] the threaded copies A / B of the handler table  s[68:69] exec at entry (= while the loop runs)  s[70:71] eligible lanes
//                s[80:81] hot records  s[82:83] text_lo  s[84:85] the lanes that execute the current instruction
//                traced runs: s[96:97] record image  s98 instruction bits  s99 size  (s[64:65]: the traced copy of the handler table)
//  clobbered by calls: s[40:47] record buffer A  s[56:63] buffer B  s[72:79] buffer T (branch / jump target)
//                s88 s89 scratch  s[90:91] dispatch target  s[92:93] s[94:95] scratch pairs
//  v[128:193] x file  v[194:195] a  v[196:197] b  v[198:199] b + imm  v[200:201] result  v[202:209] scratch  v[210:211] tohost + 4
//  v[212:213] -mem_base  v[214:221] mem_size - 1 / 2 / 4 / 8  v[222:223] text_lo  v[224:225] text_hi  v[226:227] -text_lo
//  v228 high address mask  v229 xb  v[230:231] arena  v[232:233] tohost  v234 pc  v235 retired  v236 ncv  v237 fcsr  v[238:241] scratch
//  traced runs: v[242:243] commit digest  v[244:245] v[246:247] the record's (key, value)  v[248:249] digest being folded

#define CC_DPPMIN(src, dst_s)                                                           \
  "v_mov_b32 v202, " src "\n"                                                           \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_shr:1 row_mask:0xf bank_mask:0xf\n"               \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_shr:2 row_mask:0xf bank_mask:0xf\n"               \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_shr:4 row_mask:0xf bank_mask:0xf\n"               \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_shr:8 row_mask:0xf bank_mask:0xf\n"               \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_bcast:15 row_mask:0xa bank_mask:0xf\n"            \
  "s_nop 1\n"                                                                           \
  "v_min_u32_dpp v202, v202, v202 row_bcast:31 row_mask:0xc bank_mask:0xf\n"            \
  "s_nop 1\n"                                                                           \
  "v_readlane_b32 " dst_s ", v202, 63\n"

// one step's front end for the record in buffer Q (words Q0..Q7, immediate pair Q23): request the sequential successor into QN,
// read both operands, dispatch.  The handler comes back to the instruction after the s_swappc_b64.
#define CC_FETCH(Q1, QN)                                                                \
  "s_add_u32 s35, s34, " Q1 "\n"                                                        \
  "s_lshl_b32 s88, s35, 4\n"                                                            \
  "s_load_dwordx8 " QN ", s[80:81], s88\n"
#define CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7) CC_ISSUE_B(Q0, Q23, Q4, Q5, Q6, Q7, "s64", "s65")
#define CC_ISSUE_B(Q0, Q23, Q4, Q5, Q6, Q7, BLO, BHI)                                   \
  "s_set_gpr_idx_on " Q4 ", gpr_idx(SRC0)\n"                                            \
  "v_mov_b32 v194, v128\n"                                                              \
  "v_mov_b32 v195, v129\n"                                                              \
  "s_set_gpr_idx_idx " Q5 "\n"                                                          \
  "v_mov_b32 v196, v128\n"                                                              \
  "v_mov_b32 v197, v129\n"                                                              \
  "s_set_gpr_idx_off\n"                                                                 \
  "s_mov_b32 s38, " Q6 "\n"                                                             \
  "s_mov_b32 s39, " Q7 "\n"                                                             \
  "s_add_u32 s90, " BLO ", " Q0 "\n"                                                    \
  "s_addc_u32 s91, " BHI ", 0\n"                                                        \
  "v_lshl_add_u64 v[198:199], v[196:197], 0, " Q23 "\n"                                 \
  "s_swappc_b64 s[54:55], s[90:91]\n"
// converged mode: every eligible lane is in; the epilogue only moves the scalar state on
#define CC_CSTEP(Q0, Q1, Q23, Q4, Q5, Q6, Q7, QN)                                       \
  CC_FETCH(Q1, QN) CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                    \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  "s_mov_b32 s34, s35\n"                                                                \
  "s_sub_u32 s36, s36, 1\n"                                                             \
  "s_cbranch_scc1 L_exit_conv_%=\n"
// diverged mode: the group = lanes whose pc is the scheduled one; they retire and move on in the epilogue
#define CC_DSTEP(Q0, Q1, Q23, Q4, Q5, Q6, Q7, QN)                                       \
  CC_FETCH(Q1, QN)                                                                      \
  "v_cmp_eq_u32_e64 s[84:85], s34, v234\n"                                              \
  CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                                     \
  CC_ADVANCE("s35")                                                                     \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  "s_mov_b32 s34, s35\n"                                                                \
  "s_sub_u32 s36, s36, 1\n"                                                             \
  "s_cbranch_scc1 L_exit_%=\n"
// Converged mode runs THREADED: two more copies of the handler table (A: the record sits in buffer A, B: in buffer B) whose handlers
// end with the next step's front end and dispatch straight into the other copy - one taken branch per simulated instruction instead
// of two (each ~11 ns with one wave on the SIMD).  A handler that leaves through shared code and returns with s[54:55] lands behind
// the dispatch that called it, which branches to a stand-alone copy of the tail.
#define CC_TSTEP(Q0, Q1, Q23, Q4, Q5, Q6, Q7, QN, BLO, BHI, NEXT)                       \
  CC_FETCH(Q1, QN) CC_ISSUE_B(Q0, Q23, Q4, Q5, Q6, Q7, BLO, BHI)                        \
  "s_branch " NEXT "\n"
#define CC_TSTEP_(...) CC_TSTEP(__VA_ARGS__)
#define CC_TEPI                                                                         \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  "s_mov_b32 s34, s35\n"                                                                \
  "s_sub_u32 s36, s36, 1\n"                                                             \
  "s_cbranch_scc1 L_exit_conv_%=\n"
#define CC_TAIL_A CC_TEPI CC_TSTEP_(CC_QB, "s86", "s87", "L_tail_B_%=")     /* record A done: the one in B next, through table B */
#define CC_TAIL_B CC_TEPI CC_TSTEP_(CC_QA, "s66", "s67", "L_tail_A_%=")
#define CC_QA "s40", "s41", "s[42:43]", "s44", "s45", "s46", "s47", "s[56:63]"
#define CC_QB "s56", "s57", "s[58:59]", "s60", "s61", "s62", "s63", "s[40:47]"
#define CC_QT "s72", "s73", "s[74:75]", "s76", "s77", "s78", "s79", "s[40:47]"
#define CC_CSTEP_(...) CC_CSTEP(__VA_ARGS__)
#define CC_DSTEP_(...) CC_DSTEP(__VA_ARGS__)

// x[rd] = v[200:201] for the lanes of the group; back to the step's epilogue
#define CC_WR                                                                           \
  "s_set_gpr_idx_on s38, gpr_idx(SRC0,DST)\n"                                           \
  "v_cndmask_b32_e64 v128, v128, v200, s[84:85]\n"                                      \
  "v_cndmask_b32_e64 v129, v129, v201, s[84:85]\n"                                      \
  "s_set_gpr_idx_off\n" CC_TRH_REG
// the lanes that execute move to pc NPC (an SGPR) and retire one instruction
#define CC_ADVANCE(NPC)                                                                 \
  "v_mov_b32 v202, " NPC "\n"                                                           \
  "v_addc_co_u32_e64 v235, s[92:93], 0, v235, s[84:85]\n"                               \
  "v_cndmask_b32_e64 v234, v234, v202, s[84:85]\n"
#define CC_RET "s_setpc_b64 s[54:55]\n"             /* (redefined around the threaded copies of the handler table) */
#define CC_WR_RET CC_WR CC_RET
#define CC_H(k) ".org " CC_HTAB " + 512 * " #k "\n"
// leaving converged mode: the steps taken in it reach the lanes' retire counts
#define CC_FLUSH                                                                        \
  "s_sub_u32 s88, s37, s36\n"                                                           \
  "v_add_u32 v235, s88, v235\n"

// address of a load / store: va = (a + imm) & am -> v[202:203], arena offset -> v[204:205]; s[92:93] = lanes that are fine so far
// (inside the arena with the access size to spare, aligned, not tohost)
#define CC_ADDR(LIM, SZM1)                                                              \
  "v_lshl_add_u64 v[202:203], v[194:195], 0, v[198:199]\n"                              \
  "v_and_b32 v203, v228, v203\n"                                                        \
  "v_lshl_add_u64 v[204:205], v[202:203], 0, v[212:213]\n"                              \
  "v_cmp_ne_u64_e64 s[92:93], v[202:203], v[232:233]\n"                                 \
  "v_and_b32 v206, " SZM1 ", v202\n"                                                    \
  "v_cmp_ge_u64_e64 s[94:95], " LIM ", v[204:205]\n"                                    \
  "v_cmp_eq_u32_e32 vcc, 0, v206\n"                                                     \
  "s_and_b64 s[92:93], s[92:93], s[94:95]\n"                                            \
  "s_and_b64 s[92:93], s[92:93], vcc\n"
#define CC_ALLOK                                                                        \
  "s_andn2_b64 s[92:93], s[84:85], s[92:93]\n"                                          \
  "s_cbranch_scc1 L_bad_%=\n"
#define CC_MEMEXEC "s_mov_b64 exec, s[84:85]\n"      /* side effects (memory, LDS, pc) only for the lanes that execute */
#define CC_ALUEXEC "s_mov_b64 exec, s[68:69]\n"
// lane address of arena offset v[204:205] -> v[206:207] (8-byte interleave across the wave's harts, DevMem::p)
#define CC_LANEPTR                                                                      \
  "v_and_b32 v206, -8, v204\n"                                                          \
  "v_mov_b32 v207, v205\n"                                                              \
  "v_lshlrev_b64 v[206:207], s50, v[206:207]\n"                                         \
  "v_and_or_b32 v206, v204, 7, v206\n"                                                  \
  "v_lshl_add_u64 v[206:207], v[206:207], 0, v[230:231]\n"
#define CC_LOAD(LIM, SZM1, LDOP, EXT)                                                   \
  CC_ADDR(LIM, SZM1) CC_ALLOK CC_LANEPTR                                                \
  CC_MEMEXEC LDOP " v[206:207], off\n" CC_ALUEXEC                                       \
  "s_waitcnt vmcnt(0)\n"                                                                \
  EXT CC_WR_RET
// stores: the data register's index is record word 6; also not tohost + 4 and not into the text
#define CC_STDATA                                                                       \
  "s_set_gpr_idx_on s38, gpr_idx(SRC0)\n"                                               \
  "v_mov_b32 v196, v128\n"                                                              \
  "v_mov_b32 v197, v129\n"                                                              \
  "s_set_gpr_idx_off\n"
#define CC_STCHK(SZ)                                                                    \
  "v_lshl_add_u64 v[208:209], v[202:203], 0, " SZ "\n"                                  \
  "v_cmp_ne_u64_e64 s[94:95], v[202:203], v[210:211]\n"                                 \
  "v_cmp_gt_u64_e64 s[88:89], v[208:209], v[222:223]\n"                                 \
  "v_cmp_lt_u64_e32 vcc, v[202:203], v[224:225]\n"                                      \
  "s_and_b64 s[92:93], s[92:93], s[94:95]\n"                                            \
  "s_and_b64 s[88:89], s[88:89], vcc\n"                                                 \
  "s_andn2_b64 s[92:93], s[92:93], s[88:89]\n"
#define CC_STORE(LIM, SZ, SZM1, STOP, TRH)                                              \
  CC_STDATA CC_ADDR(LIM, SZM1) CC_STCHK(SZ) CC_ALLOK CC_LANEPTR                         \
  CC_MEMEXEC STOP "\n" CC_ALUEXEC TRH CC_RET
#define CC_FADDR(dst) "s_add_u32 s88, s38, s51\n" "v_add_u32 " dst ", s88, v229\n"      /* LDS address of the f register of word 6 */
#define CC_FPON "s_bitcmp1_b32 s49, 2\n" "s_cbranch_scc0 L_bad_%=\n"

// branches: vcc = lanes that take it (the compare sits in front), s39 = target; its record is requested into buffer T
#define CC_BR_HEAD                                                                      \
  "s_lshl_b32 s88, s39, 4\n"                                                            \
  "s_load_dwordx8 s[72:79], s[80:81], s88\n"                                            \
  "s_cmp_lg_u32 s48, 0\n"                                                               \
  "s_cbranch_scc1 L_br_div_%=\n"
#define CC_TAKEN_ALL                                                                    \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  "s_mov_b32 s34, s39\n"                                                                \
  "s_sub_u32 s36, s36, 1\n"                                                             \
  "s_cbranch_scc1 L_exit_conv_%=\n"                                                     \
  "s_branch L_CT_T_%=\n"
#define CC_BR_FWD(CMP)                                                                  \
  CMP " vcc, v[194:195], v[196:197]\n" "s_and_b64 vcc, vcc, s[84:85]\n" CC_BR_HEAD                                       \
  "s_cmp_lg_u64 vcc, 0\n"                                                               \
  "s_cbranch_scc1 L_br_some_%=\n"                                                       \
  CC_RET
#define CC_BR_BACK(CMP)                                                                 \
  CMP " vcc, v[194:195], v[196:197]\n" "s_and_b64 vcc, vcc, s[84:85]\n" CC_BR_HEAD                                       \
  "s_cmp_eq_u64 vcc, s[70:71]\n"                                                        \
  "s_cbranch_scc0 L_br_notall_%=\n"                                                     \
  CC_TAKEN_ALL
#define CC_BR_OUT(CMP)                                                                  \
  CMP " vcc, v[194:195], v[196:197]\n"                                                  \
  "s_and_b64 vcc, vcc, s[84:85]\n"                                                      \
  "s_cmp_lg_u64 vcc, 0\n"                                                               \
  "s_cbranch_scc1 L_bad_%=\n"                                                           \
  CC_RET

// a call of compiled C++ (standard calling convention; the callee keeps s34+ and never reaches v128+)
#define CC_CALL(fn)                                                                     \
  "s_getpc_b64 s[92:93]\n"                                                              \
  "s_add_u32 s92, s92, " fn "@rel32@lo+4\n"                                             \
  "s_addc_u32 s93, s93, " fn "@rel32@hi+12\n"                                           \
  "s_swappc_b64 s[30:31], s[92:93]\n"
#define CC_FREG_READ(FIELD, dst)                                                        \
  "s_bfe_u32 s88, s39, " FIELD "\n"                                                     \
  "s_lshl_b32 s88, s88, 3\n"                                                            \
  "s_add_u32 s88, s88, s51\n"                                                           \
  "v_add_u32 v238, s88, v229\n"                                                         \
  "ds_read_b64 " dst ", v238\n"
#define CC_PURE_FP_CALL                                                                 \
  CC_FPON                                                                               \
  CC_FREG_READ("0x5000d", "v[4:5]") CC_FREG_READ("0x50012", "v[6:7]") CC_FREG_READ("0x50017", "v[8:9]")   \
  "s_bfe_u32 s88, s39, 0x50008\n"                                                       \
  "s_lshl_b32 s88, s88, 3\n"                                                            \
  "s_add_u32 s88, s88, s51\n"                                                           \
  "v_add_u32 v241, s88, v229\n"                                                         \
  "v_mov_b32 v0, s39\n"                                                                 \
  "v_sub_u32 v1, v198, v196\n"                                                          \
  "v_mov_b32 v2, v194\n"                                                                \
  "v_mov_b32 v3, v195\n"                                                                \
  "v_mov_b32 v10, v237\n"                                                               \
  "v_mov_b32 v11, s52\n"                                                                \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  CC_CALL("wsp_pure_fp")                                                                \
  "v_cmp_ne_u32_e32 vcc, 0, v3\n"                                                       \
  "s_andn2_b64 s[92:93], s[84:85], vcc\n"                                               \
  "s_cbranch_scc1 L_bad_%=\n"                                                           \
  "v_cndmask_b32_e64 v237, v237, v2, s[84:85]\n"

// ---- traced runs (cfg bit 17): the commit digest of every executed instruction, as the C++ loop folds it (wsp_lib.hip: pc, meta =
// raw | M << 32 | size << 48, then one (key, value) pair: x register / f register / memory / none).  One more copy of the handler
// table whose handlers leave the pair in v[244:245], v[246:247] (the CC_TRX_* hooks); the step folds the four words into the lane's
// digest v[242:243] under the mask of the lanes that execute.  Traced runs use diverged-mode steps throughout (per-lane pc and
// counters), call / return dispatch, and leave F / D arithmetic to the general path (its record can carry an fcsr pair too).
#define CC_TRX_REG                                                                      \
  "s_lshr_b32 s88, s38, 1\n" "s_cmp_lg_u32 s38, 0\n" "s_cselect_b32 s89, -1, 0\n"       \
  "v_mov_b32 v244, s88\n" "v_mov_b32 v245, 0x10000000\n"                                \
  "v_and_b32 v246, s89, v200\n" "v_and_b32 v247, v228, v201\n" "v_and_b32 v247, s89, v247\n"
#define CC_TRX_STK "v_mov_b32 v244, v202\n" "v_xor_b32 v245, 0x40000000, v203\n"
#define CC_TRX_ST1 CC_TRX_STK "v_and_b32 v246, 0xff, v196\n" "v_mov_b32 v247, 0\n"
#define CC_TRX_ST2 CC_TRX_STK "v_and_b32 v246, 0xffff, v196\n" "v_mov_b32 v247, 0\n"
#define CC_TRX_ST4 CC_TRX_STK "v_mov_b32 v246, v196\n" "v_mov_b32 v247, 0\n"
#define CC_TRX_ST8 CC_TRX_STK "v_mov_b32 v246, v196\n" "v_mov_b32 v247, v197\n"
#define CC_TRX_FLD "s_lshr_b32 s88, s38, 3\n" "v_mov_b32 v244, s88\n" "v_mov_b32 v245, 0x20000000\n" "v_mov_b32 v246, v200\n" "v_mov_b32 v247, v201\n"
#define CC_TRX_NONE "v_mov_b32 v244, 0\n" "v_mov_b32 v245, 0x10000000\n" "v_mov_b32 v246, 0\n" "v_mov_b32 v247, 0\n"
#define CC_BR_TR(CMP, OUT)                                                              \
  CMP " vcc, v[194:195], v[196:197]\n" "s_and_b64 vcc, vcc, s[84:85]\n" OUT              \
  "v_mov_b32 v202, s35\n" "v_mov_b32 v203, s39\n" CC_TRX_NONE                            \
  "v_cndmask_b32_e32 v202, v202, v203, vcc\n" "s_branch L_tr_ctl_%=\n"
// h = fold(h, w) on the working copy v[248:249]: h ^= w; h = rotl(h, 27); h = h * 5 + 0x52dce729 (wsp_core.h)
#define CC_FOLD(WLO, WHI)                                                               \
  "v_xor_b32 v238, " WLO ", v248\n" "v_xor_b32 v239, " WHI ", v249\n"                    \
  "v_alignbit_b32 v240, v238, v239, 5\n" "v_alignbit_b32 v241, v239, v238, 5\n"         \
  "v_lshl_add_u64 v[238:239], v[240:241], 2, v[240:241]\n"                              \
  "v_add_co_u32_e32 v248, vcc, 0x52dce729, v238\n" "v_addc_co_u32_e32 v249, vcc, 0, v239, vcc\n"
#define CC_FOLD4                                                                        \
  "v_mov_b32 v248, v242\n" "v_mov_b32 v249, v243\n"                                     \
  "s_add_u32 s92, s82, s34\n" "s_addc_u32 s93, s83, 0\n" CC_FOLD("s92", "s93")          \
  "s_lshl_b32 s89, s99, 16\n" "s_or_b32 s89, s89, 3\n" CC_FOLD("s98", "s89")            \
  CC_FOLD("v244", "v245") CC_FOLD("v246", "v247")                                       \
  "v_cndmask_b32_e64 v242, v242, v248, s[84:85]\n" "v_cndmask_b32_e64 v243, v243, v249, s[84:85]\n"
#define CC_RSTEP(Q0, Q1, Q23, Q4, Q5, Q6, Q7, QN)                                       \
  CC_FETCH(Q1, QN)                                                                      \
  "s_lshl_b32 s88, s34, 3\n" "s_add_u32 s88, s88, 12\n" "s_load_dword s98, s[96:97], s88\n"   \
  "s_mov_b32 s99, " Q1 "\n"                                                             \
  "v_cmp_eq_u32_e64 s[84:85], s34, v234\n"                                              \
  CC_ISSUE(Q0, Q23, Q4, Q5, Q6, Q7)                                                     \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  CC_FOLD4                                                                              \
  CC_ADVANCE("s35")                                                                     \
  "s_mov_b32 s34, s35\n"                                                                \
  "s_sub_u32 s36, s36, 1\n"                                                             \
  "s_cbranch_scc1 L_exit_%=\n"
#define CC_RSTEP_(...) CC_RSTEP(__VA_ARGS__)
// hooks and endings of the untraced copies
#define CC_TRH_REG
#define CC_TRH_ST1
#define CC_TRH_ST2
#define CC_TRH_ST4
#define CC_TRH_ST8
#define CC_TRH_FLD
#define CC_TRH_NONE
#define CC_TRH_NOFP
#define CC_JAL_END "s_cmp_lg_u32 s48, 0\n" "s_cbranch_scc1 L_jal_div_%=\n" CC_TAKEN_ALL
#define CC_JALR_END "s_branch L_jalr_tail_%=\n"

// The compiler prices an inline-asm block at 8 bytes per statement and does not see the 512-byte alignment of the handler slots, so it
// would place short branches across this block (~122 KB of code from ~6500 statements) that cannot reach.  10240 re-definable local
// labels emit nothing and bring its estimate above the real size, so that it relaxes those branches.
#define CC_PAD4 "1:\n1:\n1:\n1:\n"
#define CC_PAD32 CC_PAD4 CC_PAD4 CC_PAD4 CC_PAD4 CC_PAD4 CC_PAD4 CC_PAD4 CC_PAD4
#define CC_PAD256 CC_PAD32 CC_PAD32 CC_PAD32 CC_PAD32 CC_PAD32 CC_PAD32 CC_PAD32 CC_PAD32
#define CC_PAD2K CC_PAD256 CC_PAD256 CC_PAD256 CC_PAD256 CC_PAD256 CC_PAD256 CC_PAD256 CC_PAD256
#define CC_SIZE_PAD CC_PAD2K CC_PAD2K CC_PAD2K CC_PAD2K CC_PAD2K

#define CC_CLOBBERS \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
  "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
  "v38", "v39", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", \
  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v112", \
  "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", \
  "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", \
  "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", \
  "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", \
  "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", \
  "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", \
  "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", \
  "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", \
  "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", \
  "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s34", "s35", "s36", "s37", "s38", "s39", \
  "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", \
  "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", \
  "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", \
  "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc", "memory"

namespace wsp {

// One run of the converged loop.  Arguments that are the same for the whole wave may sit in any lane's registers (they are read
// with v_readfirstlane).  hrecs: hot-record image of the wave's program; cfg: bit 1 rv32, bit 16 never enter converged mode (tests of the diverged steps), bit 2 mstatus.FS dirty in every lane that
// takes part (flw / fld / fsw / fsd and F / D arithmetic inline), bit 3 D extension (flw boxes), bit 17 traced run (thash: the lane's
// commit digest, updated for every instruction it executes; recs: the program's 16-byte record image, for the instruction bits); xb: LDS byte offset of the lane's x
// block; arena: the lane's arena base (DevMem::arena); vpc: the lane's pc as a text offset, 0xffffffff for lanes that take no part;
// fcsr: the lane's fcsr; n: iterations (>= 1).  Out: vit = instructions retired by the lane, ncv != 0 = the lane stands at an
// instruction for the general path.
__device__ __forceinline__ void conv_core(const void* hrecs, const void* recs, u64 mbase, u64 msize, u64 tlo, u64 thi, u32 tlen, u32 cfg,
                                          u32 lpw_shift, u32 foff, u32 n, u32 isa, u32 xb, const void* arena, u64 tohost, u32& vpc, u32& fcsr,
                                          u64& thash, u32& vit, u32& ncv)
{
  u32 vit_ = 0, ncv_ = 0;
  asm volatile(
    // ------------------------------------------------------------------ entry (exec = all lanes of the wave)
    "s_mov_b64 s[68:69], exec\n"
    "v_mov_b32 v229, %[xb]\n"          "v_mov_b64 v[230:231], %[arena]\n"   "v_mov_b64 v[232:233], %[tohost]\n"
    "v_mov_b32 v234, %[vpc]\n"         "v_mov_b32 v237, %[fcsr]\n"          "v_mov_b32 v235, 0\n"               "v_mov_b32 v236, 0\n"
    "v_mov_b64 v[202:203], %[recs]\n"  "v_mov_b64 v[204:205], %[mbase]\n"   "v_mov_b64 v[206:207], %[msize]\n"
    "v_mov_b64 v[222:223], %[tlo]\n"   "v_mov_b64 v[224:225], %[thi]\n"
    "v_mov_b32 v208, %[cfg]\n"         "v_mov_b32 v209, %[lpwsh]\n"         "v_mov_b32 v238, %[foff]\n"         "v_mov_b32 v239, %[n]\n"
    "v_mov_b32 v240, %[isa]\n"         "v_mov_b32 v241, %[tlen]\n"
    "v_mov_b64 v[242:243], %[thash]\n" "v_mov_b64 v[244:245], %[rimg]\n"
    "ds_read_b64 v[130:131], v229 offset:8\n"    "ds_read_b64 v[132:133], v229 offset:16\n"   "ds_read_b64 v[134:135], v229 offset:24\n"
    "ds_read_b64 v[136:137], v229 offset:32\n"   "ds_read_b64 v[138:139], v229 offset:40\n"   "ds_read_b64 v[140:141], v229 offset:48\n"
    "ds_read_b64 v[142:143], v229 offset:56\n"   "ds_read_b64 v[144:145], v229 offset:64\n"   "ds_read_b64 v[146:147], v229 offset:72\n"
    "ds_read_b64 v[148:149], v229 offset:80\n"   "ds_read_b64 v[150:151], v229 offset:88\n"   "ds_read_b64 v[152:153], v229 offset:96\n"
    "ds_read_b64 v[154:155], v229 offset:104\n"  "ds_read_b64 v[156:157], v229 offset:112\n"  "ds_read_b64 v[158:159], v229 offset:120\n"
    "ds_read_b64 v[160:161], v229 offset:128\n"  "ds_read_b64 v[162:163], v229 offset:136\n"  "ds_read_b64 v[164:165], v229 offset:144\n"
    "ds_read_b64 v[166:167], v229 offset:152\n"  "ds_read_b64 v[168:169], v229 offset:160\n"  "ds_read_b64 v[170:171], v229 offset:168\n"
    "ds_read_b64 v[172:173], v229 offset:176\n"  "ds_read_b64 v[174:175], v229 offset:184\n"  "ds_read_b64 v[176:177], v229 offset:192\n"
    "ds_read_b64 v[178:179], v229 offset:200\n"  "ds_read_b64 v[180:181], v229 offset:208\n"  "ds_read_b64 v[182:183], v229 offset:216\n"
    "ds_read_b64 v[184:185], v229 offset:224\n"  "ds_read_b64 v[186:187], v229 offset:232\n"  "ds_read_b64 v[188:189], v229 offset:240\n"
    "ds_read_b64 v[190:191], v229 offset:248\n"
    "v_mov_b32 v192, 0\n"  "v_mov_b32 v193, 0\n"  "v_mov_b32 v128, 0\n"  "v_mov_b32 v129, 0\n"
    "v_readfirstlane_b32 s80, v202\n"  "v_readfirstlane_b32 s81, v203\n"  "v_readfirstlane_b32 s82, v222\n"  "v_readfirstlane_b32 s83, v223\n"
    "v_readfirstlane_b32 s49, v208\n"  "v_readfirstlane_b32 s50, v209\n"  "v_readfirstlane_b32 s51, v238\n"  "v_readfirstlane_b32 s36, v239\n"
    "v_readfirstlane_b32 s52, v240\n"  "v_readfirstlane_b32 s53, v241\n"  "v_readfirstlane_b32 s96, v244\n"  "v_readfirstlane_b32 s97, v245\n"
    "v_sub_co_u32_e32 v212, vcc, 0, v204\n"
    "v_lshl_add_u64 v[214:215], v[206:207], 0, -1\n"
    "v_lshl_add_u64 v[216:217], v[206:207], 0, -2\n"
    "v_subb_co_u32_e32 v213, vcc, 0, v205, vcc\n"
    "v_lshl_add_u64 v[218:219], v[206:207], 0, -4\n"
    "v_lshl_add_u64 v[220:221], v[206:207], 0, -8\n"
    "v_sub_co_u32_e32 v226, vcc, 0, v222\n"
    "v_lshl_add_u64 v[210:211], v[232:233], 0, 4\n"
    "s_bitcmp1_b32 s49, 1\n"
    "s_nop 0\n"
    "v_subb_co_u32_e32 v227, vcc, 0, v223, vcc\n"
    "s_cselect_b32 s88, 0, -1\n"
    "v_mov_b32 v228, s88\n"
    "s_sub_u32 s36, s36, 1\n"
    "s_getpc_b64 s[64:65]\n"
    "L_anchor_%=:\n"
    "s_add_u32 s64, s64, L_htab_%=-L_anchor_%=\n"
    "s_addc_u32 s65, s65, 0\n"
    "s_add_u32 s66, s64, L_htabA_%=-L_htab_%=\n"
    "s_addc_u32 s67, s65, 0\n"
    "s_add_u32 s86, s64, L_htabB_%=-L_htab_%=\n"
    "s_addc_u32 s87, s65, 0\n"
    "s_bitcmp1_b32 s49, 17\n"                     // a traced run dispatches through the traced copy only: s[64:65] is that one then
    "s_cbranch_scc0 L_untraced_%=\n"
    "s_add_u32 s64, s64, L_htabR_%=-L_htab_%=\n"
    "s_addc_u32 s65, s65, 0\n"
    "L_untraced_%=:\n"
    "v_cmp_ne_u32_e64 s[70:71], -1, v234\n"
    "s_branch L_sched_%=\n"
    // ------------------------------------------------------------------ a control transfer in diverged mode (or one that splits the
    // wave), then the schedule: the minimum pc over all lanes; converged if all agree.  Laid out so that the diverged path falls
    // through from here into its steps: with one wave on the SIMD every taken branch is ~11 ns of instruction refetch.
    "L_br_mixed_%=:\n"                             // the wave diverges here
    CC_FLUSH
    "L_br_div_%=:\n"                               // the lanes that execute go their ways (vcc: those that take the branch)
    "v_mov_b32 v202, s35\n"
    "v_mov_b32 v203, s39\n"
    "v_addc_co_u32_e64 v235, s[92:93], 0, v235, s[84:85]\n"
    "v_cndmask_b32_e32 v202, v202, v203, vcc\n"
    "v_cndmask_b32_e64 v234, v234, v202, s[84:85]\n"
    "L_sched_count_%=:\n"
    "s_sub_u32 s36, s36, 1\n"
    "s_cbranch_scc1 L_exit_%=\n"
    "L_sched_%=:\n"
    "s_cmp_gt_u32 s50, 2\n"                       // up to 4 harts per wave (lanes 0..3; the others hold -1): four lane reads, else
    "s_cbranch_scc1 L_sched_dpp_%=\n"             // the six-step DPP reduction
    "v_readlane_b32 s34, v234, 0\n"
    "v_readlane_b32 s88, v234, 1\n"
    "v_readlane_b32 s89, v234, 2\n"
    "v_readlane_b32 s92, v234, 3\n"
    "s_min_u32 s34, s34, s88\n"
    "s_min_u32 s89, s89, s92\n"
    "s_min_u32 s34, s34, s89\n"
    "L_sched_have_%=:\n"
    "s_lshl_b32 s88, s34, 4\n"
    "s_waitcnt lgkmcnt(0)\n"                      // a successor record may still be on its way into buffer A: scalar loads land in any order
    "s_load_dwordx8 s[40:47], s[80:81], s88\n"
    "v_cmp_eq_u32_e64 s[84:85], s34, v234\n"      // the lanes that stand there (lanes without a hart hold -1)
    "s_cmp_eq_u64 s[84:85], s[70:71]\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_cbranch_scc1 L_go_conv_%=\n"
    "L_go_div_%=:\n"
    "s_mov_b32 s48, 1\n"
    "s_bitcmp1_b32 s49, 17\n"
    "s_cbranch_scc1 L_RT_A_%=\n"
    // ------------------------------------------------------------------ the steps, diverged mode
    "L_DT_A_%=:\n" CC_DSTEP_(CC_QA) CC_DSTEP_(CC_QB) CC_DSTEP_(CC_QA) CC_DSTEP_(CC_QB)
    "s_branch L_DT_A_%=\n"
    "L_go_conv_%=:\n"
    "s_and_b32 s88, s49, 0x30000\n"               // never converged mode: test hook (bit 16), traced run (bit 17)
    "s_cbranch_scc1 L_go_div_%=\n"
    "s_mov_b32 s48, 0\n"
    "s_mov_b32 s37, s36\n"
    "s_branch L_CT_A_%=\n"
    "L_sched_dpp_%=:\n"
    CC_DPPMIN("v234", "s34")
    "s_branch L_sched_have_%=\n"
    // ------------------------------------------------------------------ the steps of a traced run
    "L_RT_A_%=:\n" CC_RSTEP_(CC_QA) CC_RSTEP_(CC_QB) CC_RSTEP_(CC_QA) CC_RSTEP_(CC_QB)
    "s_branch L_RT_A_%=\n"
    "L_tr_ctl_%=:\n"                               // traced control transfer: lanes' next pcs in v202, pair set
    "s_waitcnt lgkmcnt(0)\n"
    CC_FOLD4
    "v_addc_co_u32_e64 v235, s[92:93], 0, v235, s[84:85]\n"
    "v_cndmask_b32_e64 v234, v234, v202, s[84:85]\n"
    "s_branch L_sched_count_%=\n"
    // ------------------------------------------------------------------ the steps, converged mode
    "L_CT_T_%=:\n" CC_TSTEP_(CC_QT, "s86", "s87", "L_tail_B_%=")     // after a taken branch: the record in T, its successor into A = what
                                                                        // table B's tails continue with
    "L_CT_A_%=:\n" CC_TSTEP_(CC_QA, "s66", "s67", "L_tail_A_%=")
    "L_tail_A_%=:\n" CC_TAIL_A
    "L_tail_B_%=:\n" CC_TAIL_B
    // ------------------------------------------------------------------ shared pieces
    "L_bad_%=:\n"                                  // nobody of the group executes: general path
    "v_cndmask_b32_e64 v236, v236, 1, s[84:85]\n"
    "s_cmp_eq_u32 s48, 0\n"
    "s_cbranch_scc1 L_exit_conv_%=\n"
    "s_branch L_exit_%=\n"
    "L_exit_conv_%=:\n"                            // converged mode ends: pc and retire count of every lane from the scalar state
    CC_FLUSH
    "v_mov_b32 v202, s34\n"
    "v_cndmask_b32_e64 v234, v234, v202, s[70:71]\n"
    "s_branch L_exit_%=\n"
    "L_br_some_%=:\n"                              // converged, forward branch, somebody takes it
    "s_cmp_eq_u64 vcc, s[70:71]\n"
    "s_cbranch_scc0 L_br_mixed_%=\n"
    CC_TAKEN_ALL
    "L_br_notall_%=:\n"                            // converged, backward branch, not everybody takes it
    "s_cmp_eq_u64 vcc, 0\n"
    "s_cbranch_scc0 L_br_mixed_%=\n"
    CC_RET
    "L_resume_%=:\n"                               // after a call: the step's epilogue by hand, then the next record again
    "s_cmp_lg_u32 s48, 0\n"
    "s_cbranch_scc0 L_resume2_%=\n"
    "s_bitcmp1_b32 s49, 17\n"
    "s_cbranch_scc0 L_resume1_%=\n"
    CC_FOLD4
    "L_resume1_%=:\n"
    CC_ADVANCE("s35")
    "L_resume2_%=:\n"
    "s_mov_b32 s34, s35\n"
    "s_lshl_b32 s88, s34, 4\n"
    "s_load_dwordx8 s[40:47], s[80:81], s88\n"
    "s_sub_u32 s36, s36, 1\n"
    "s_cbranch_scc1 L_exit_any_%=\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_bitcmp1_b32 s49, 17\n"
    "s_cbranch_scc1 L_RT_A_%=\n"
    "s_cmp_lg_u32 s48, 0\n"
    "s_cbranch_scc1 L_DT_A_%=\n"
    "s_branch L_CT_A_%=\n"
    "L_exit_any_%=:\n"
    "s_cmp_lg_u32 s48, 0\n"
    "s_cbranch_scc1 L_exit_%=\n"
    "s_branch L_exit_conv_%=\n"
    "L_jal_div_%=:\n"
    CC_ADVANCE("s39")
    "s_branch L_sched_count_%=\n"
    "L_jalr_tail_%=:\n"                            // lanes' targets in v202, link written
    "v_addc_co_u32_e64 v235, s[92:93], 0, v235, s[84:85]\n"
    "v_cndmask_b32_e64 v234, v234, v202, s[84:85]\n"
    "s_cmp_lg_u32 s48, 0\n"
    "s_cbranch_scc1 L_sched_count_%=\n"
    CC_FLUSH
    "s_branch L_sched_count_%=\n"
    // ------------------------------------------------------------------ handlers, 512 bytes apart (class order: HC_* in wsp_core.h)
#define CC_HTAB "L_htab_%="
#define CC_RET "s_setpc_b64 s[54:55]\n"
#include "wsp_conv_handlers.inc"
#undef CC_HTAB
#undef CC_RET
#define CC_HTAB "L_htabA_%="
#define CC_RET CC_TAIL_A
#include "wsp_conv_handlers.inc"
#undef CC_HTAB
#undef CC_RET
#define CC_HTAB "L_htabB_%="
#define CC_RET CC_TAIL_B
#include "wsp_conv_handlers.inc"
#undef CC_HTAB
#undef CC_RET
#define CC_RET "s_setpc_b64 s[54:55]\n"
    // the traced copy: call / return dispatch, hooks on, control transfers through L_tr_ctl
#define CC_HTAB "L_htabR_%="
#undef CC_TRH_REG
#undef CC_TRH_ST1
#undef CC_TRH_ST2
#undef CC_TRH_ST4
#undef CC_TRH_ST8
#undef CC_TRH_FLD
#undef CC_TRH_NONE
#undef CC_TRH_NOFP
#undef CC_JAL_END
#undef CC_JALR_END
#undef CC_BR_FWD
#undef CC_BR_BACK
#undef CC_BR_OUT
#define CC_TRH_REG CC_TRX_REG
#define CC_TRH_ST1 CC_TRX_ST1
#define CC_TRH_ST2 CC_TRX_ST2
#define CC_TRH_ST4 CC_TRX_ST4
#define CC_TRH_ST8 CC_TRX_ST8
#define CC_TRH_FLD CC_TRX_FLD
#define CC_TRH_NONE CC_TRX_NONE
#define CC_TRH_NOFP "s_branch L_bad_%=\n"
#define CC_JAL_END "v_mov_b32 v202, s39\n" "s_branch L_tr_ctl_%=\n"
#define CC_JALR_END "s_branch L_tr_ctl_%=\n"
#define CC_BR_FWD(CMP) CC_BR_TR(CMP, "")
#define CC_BR_BACK(CMP) CC_BR_TR(CMP, "")
#define CC_BR_OUT(CMP) CC_BR_TR(CMP, "s_cmp_lg_u64 vcc, 0\n" "s_cbranch_scc1 L_bad_%=\n")
#include "wsp_conv_handlers.inc"
#undef CC_HTAB
    // ------------------------------------------------------------------ exit: x file back to LDS
    "L_exit_%=:\n"
    "s_mov_b64 exec, s[70:71]\n"                   // only lanes that took part own an x block (the others alias lane 0's / their owner's)
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
    "ds_write_b64 v229, v[130:131] offset:8\n"    "ds_write_b64 v229, v[132:133] offset:16\n"   "ds_write_b64 v229, v[134:135] offset:24\n"
    "ds_write_b64 v229, v[136:137] offset:32\n"   "ds_write_b64 v229, v[138:139] offset:40\n"   "ds_write_b64 v229, v[140:141] offset:48\n"
    "ds_write_b64 v229, v[142:143] offset:56\n"   "ds_write_b64 v229, v[144:145] offset:64\n"   "ds_write_b64 v229, v[146:147] offset:72\n"
    "ds_write_b64 v229, v[148:149] offset:80\n"   "ds_write_b64 v229, v[150:151] offset:88\n"   "ds_write_b64 v229, v[152:153] offset:96\n"
    "ds_write_b64 v229, v[154:155] offset:104\n"  "ds_write_b64 v229, v[156:157] offset:112\n"  "ds_write_b64 v229, v[158:159] offset:120\n"
    "ds_write_b64 v229, v[160:161] offset:128\n"  "ds_write_b64 v229, v[162:163] offset:136\n"  "ds_write_b64 v229, v[164:165] offset:144\n"
    "ds_write_b64 v229, v[166:167] offset:152\n"  "ds_write_b64 v229, v[168:169] offset:160\n"  "ds_write_b64 v229, v[170:171] offset:168\n"
    "ds_write_b64 v229, v[172:173] offset:176\n"  "ds_write_b64 v229, v[174:175] offset:184\n"  "ds_write_b64 v229, v[176:177] offset:192\n"
    "ds_write_b64 v229, v[178:179] offset:200\n"  "ds_write_b64 v229, v[180:181] offset:208\n"  "ds_write_b64 v229, v[182:183] offset:216\n"
    "ds_write_b64 v229, v[184:185] offset:224\n"  "ds_write_b64 v229, v[186:187] offset:232\n"  "ds_write_b64 v229, v[188:189] offset:240\n"
    "ds_write_b64 v229, v[190:191] offset:248\n"
    "s_mov_b64 exec, s[68:69]\n"
    "v_mov_b32 %[vpc], v234\n"  "v_mov_b32 %[vit], v235\n"  "v_mov_b32 %[ncv], v236\n"  "v_mov_b32 %[fcsr], v237\n"  "v_mov_b64 %[thash], v[242:243]\n"
    "s_waitcnt lgkmcnt(0)\n"
    CC_SIZE_PAD
    : [vpc] "+v"(vpc), [fcsr] "+v"(fcsr), [thash] "+v"(thash), [vit] "+v"(vit_), [ncv] "+v"(ncv_)
    : [recs] "v"(hrecs), [rimg] "v"(recs), [mbase] "v"(mbase), [msize] "v"(msize), [tlo] "v"(tlo), [thi] "v"(thi), [tlen] "v"(tlen), [cfg] "v"(cfg),
      [lpwsh] "v"(lpw_shift), [foff] "v"(foff), [n] "v"(n), [isa] "v"(isa), [xb] "v"(xb), [arena] "v"(arena), [tohost] "v"(tohost)
    : CC_CLOBBERS);
  vit = vit_; ncv = ncv_;
}

}  // namespace wsp

#else   // host pass of the same translation unit: the kernels' bodies are parsed but never run there
namespace wsp {
__device__ void conv_core(const void* hrecs, const void* recs, u64 mbase, u64 msize, u64 tlo, u64 thi, u32 tlen, u32 cfg, u32 lpw_shift, u32 foff,
                          u32 n, u32 isa, u32 xb, const void* arena, u64 tohost, u32& vpc, u32& fcsr, u64& thash, u32& vit, u32& ncv);
}
#endif  // __HIP_DEVICE_COMPILE__
