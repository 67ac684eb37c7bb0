// The converged hot loop as hand-written CDNA (gfx950) code: one scheduled pc per wave, threaded dispatch, x registers in VGPRs.
//
// What it replaces: the C++ converged loop of wsp_step_kernel for untraced runs (same contract, see the comment there), which is
// the reference's Hart::untilAddress body (Hart.cpp:6025-6201: fetch, decode-cache hit, execute switch Hart.cpp:7544) for the
// inline subset.  Measured on the C++ loop (tools/ubench.py, 4096 harts = one wave per SIMD, profiles/r03_ubench_*.jsonl): an
// `add` step took 487 clocks for ~60 wave instructions, and every extra level of the compiler's compare-chain dispatch cost ~38
// clocks (`xor`, 15 levels down: 870 clocks) - with one wave per SIMD a taken branch is a full instruction-fetch restart and
// nothing hides it.  So this loop has exactly two taken branches per simulated instruction (the dispatch jump and the back edge):
//
//  * threaded dispatch: the pre-decoded record carries a handler class (Rec.pad bits 15:9, HC_*), handlers sit at a fixed
//    512-byte stride and the dispatch is `s_and / s_add / s_addc / s_setpc_b64`;
//  * the x register file of every lane lives in VGPRs v[128:193] (slot s = x[s] for 1..31, slot 0 = write sink for rd = x0,
//    slot 32 = constant zero which reads of x0 and immediate forms are redirected to), indexed with s_set_gpr_idx_on by the
//    wave-uniform register numbers - no LDS round trip on the dependent chain; LDS <-> VGPR copies happen at entry and exit only;
//  * records come through the scalar cache, one step ahead; taken branches prefetch their target record while the compare runs;
//  * a step whose lanes are not all fine (access outside the arena, tohost, text, misaligned, target outside the text, an op
//    outside the inline subset) changes nothing for ANY lane of the group and ends the loop with ncv = 1 on those lanes: the
//    general path executes that instruction (same contract as the C++ loop, coarser grain: rare events only).
//
// gfx940-family wait states observed by hand (nothing inside an asm statement is padded by the compiler): VALU write of an
// SGPR / VCC -> VALU read of it: 2; VALU write of a VGPR -> v_readlane / DPP read: 1 / 2; v_cmp -> s_and_saveexec: interlocked.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)

// ---- fixed registers (all in the clobber list)
//  s[40:43] q (current record)   s[44:47] qn (sequential successor)   s[48:51] qt (branch / jump target)
//  s52 upc  s53 spc  s54 iterations left  s55 s56 scratch  s57 rd * 2  s[58:59] sign-extended immediate
//  s[60:61] handler table base  s[62:63] exec at loop top  s[64:65] dispatch target  s[66:71] scratch pairs
//  s[72:73] exec at entry  s[74:75] -mem_base  s[76:77] mem_size  s[78:79] text_lo  s[80:81] text_hi  s[82:83] -text_lo
//  s[84:85] eligible lanes  s[86:87] record base  s88 text length (0 when control transfers cannot be inlined)
//  s89 high address mask (rv32: 0)  s90 cfg (bit 1 rv32, bit 2 FS dirty, bit 3 D)  s91 lpw shift  s92 f block offset  s93 target  s98 next pc
//  v[128:193] x file  v[194:195] a  v[196:197] b  v[198:199] b + imm  v[200:201] result  v[202:209] scratch  v[210:211] tohost + 4

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

// x[rd] = v[200:201] for the lanes of the group
#define CC_WR                                                                           \
  "s_set_gpr_idx_on s57, gpr_idx(DST)\n"                                                \
  "v_mov_b32 v128, v200\n"                                                              \
  "v_mov_b32 v129, v201\n"                                                              \
  "s_set_gpr_idx_off\n"

// the lanes of the group retire this instruction and go on to its sequential successor
#define CC_LANES_SEQ                                                                    \
  "v_mov_b32 %[vpc], s53\n"                                                             \
  "v_add_u32 %[vit], 1, %[vit]\n"

// back to the loop top with the sequential successor's record (already requested)
#define CC_NEXT_SEQ                                                                     \
  "s_mov_b64 exec, s[62:63]\n"                                                          \
  "s_waitcnt lgkmcnt(0)\n"                                                              \
  "s_mov_b64 s[40:41], s[44:45]\n"                                                      \
  "s_mov_b64 s[42:43], s[46:47]\n"                                                      \
  "s_mov_b32 s52, s53\n"                                                                \
  "s_sub_u32 s54, s54, 1\n"                                                             \
  "s_cmp_lg_u32 s54, 0\n"                                                               \
  "s_cbranch_scc1 L_top_%=\n"                                                           \
  "s_branch L_exit_%=\n"

#define CC_TAIL_WR CC_WR CC_LANES_SEQ CC_NEXT_SEQ
#define CC_H(k) ".org L_htab_%= + 512 * " #k "\n"

// address of a load / store: va = (a + imm) & am -> v[202:203], arena offset -> v[204:205]; s[68:69] = lanes that are fine so far
// (inside the arena with `SZ` bytes to spare, aligned, not tohost)
#define CC_ADDR(SZ, SZM1)                                                               \
  "v_lshl_add_u64 v[202:203], v[194:195], 0, s[58:59]\n"                                \
  "s_sub_u32 s66, s76, " SZ "\n"                                                        \
  "s_subb_u32 s67, s77, 0\n"                                                            \
  "v_and_b32 v203, s89, v203\n"                                                         \
  "v_lshl_add_u64 v[204:205], v[202:203], 0, s[74:75]\n"                                \
  "v_cmp_ne_u64_e64 s[70:71], v[202:203], %[tohost]\n"                                  \
  "v_and_b32 v206, " SZM1 ", v202\n"                                                    \
  "v_cmp_ge_u64_e64 s[68:69], s[66:67], v[204:205]\n"                                   \
  "v_cmp_eq_u32_e32 vcc, 0, v206\n"                                                     \
  "s_and_b64 s[68:69], s[68:69], s[70:71]\n"                                            \
  "s_and_b64 s[68:69], s[68:69], vcc\n"

// lane address of arena offset v[204:205] -> v[206:207] (8-byte interleave across the wave's harts, DevMem::p)
#define CC_LANEPTR                                                                      \
  "v_and_b32 v206, -8, v204\n"                                                          \
  "v_mov_b32 v207, v205\n"                                                              \
  "v_lshlrev_b64 v[206:207], s91, v[206:207]\n"                                         \
  "v_and_or_b32 v206, v204, 7, v206\n"                                                  \
  "v_lshl_add_u64 v[206:207], v[206:207], 0, %[arena]\n"

// integer load: LDOP the global_load mnemonic, EXT the instruction(s) that make v201 from v200
#define CC_LOAD(SZ, SZM1, LDOP, EXT)                                                    \
  CC_ADDR(SZ, SZM1)                                                                     \
  "s_xor_b64 s[68:69], s[68:69], exec\n"                                                \
  "s_cbranch_scc1 L_bad_%=\n"                                                           \
  CC_LANEPTR                                                                            \
  LDOP " v[206:207], off\n"                                                             \
  CC_LANES_SEQ                                                                          \
  "s_waitcnt vmcnt(0)\n"                                                                \
  EXT                                                                                   \
  CC_WR CC_NEXT_SEQ

// the extra conditions of a store: not tohost + 4, not into the text -> s[68:69] updated
#define CC_STCHK(SZ)                                                                    \
  "v_lshl_add_u64 v[208:209], v[202:203], 0, " SZ "\n"                                  \
  "v_cmp_ne_u64_e64 s[70:71], v[202:203], v[210:211]\n"                                 \
  "v_cmp_gt_u64_e64 s[66:67], v[208:209], s[78:79]\n"                                   \
  "v_cmp_lt_u64_e64 vcc, v[202:203], s[80:81]\n"                                        \
  "s_and_b64 s[68:69], s[68:69], s[70:71]\n"                                            \
  "s_and_b64 s[66:67], s[66:67], vcc\n"                                                 \
  "s_andn2_b64 s[68:69], s[68:69], s[66:67]\n"

#define CC_STORE(SZ, SZM1, STOP)                                                        \
  CC_ADDR(SZ, SZM1)                                                                     \
  CC_STCHK(SZ)                                                                          \
  "s_xor_b64 s[68:69], s[68:69], exec\n"                                                \
  "s_cbranch_scc1 L_bad_%=\n"                                                           \
  CC_LANEPTR                                                                            \
  STOP "\n"                                                                             \
  CC_LANES_SEQ CC_NEXT_SEQ

// conditional branch: vcc = lanes that take it (set by the compare in front of this)
#define CC_BRANCH                                                                       \
  "s_add_u32 s93, s52, s41\n"                                                           \
  "s_cmp_ge_u32 s93, s88\n"                                                             \
  "s_cbranch_scc1 L_br_out_%=\n"                                                        \
  "s_lshl_b32 s55, s93, 3\n"                                                            \
  "s_load_dwordx4 s[48:51], s[86:87], s55\n"                                            \
  "s_branch L_br_%=\n"

#define CC_CLOBBERS                                                                                                                     \
  "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143",     \
  "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159",     \
  "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175",     \
  "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191",     \
  "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",     \
  "v208", "v209", "v210", "v211",                                                                                                     \
  "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57",       \
  "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75",       \
  "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93",       \
  "s98", "vcc", "scc", "memory"

namespace wsp {

// One run of the converged loop.  Uniform inputs must be provably wave-uniform (kernel arguments / readfirstlane).
//   recs: record image of the wave's program; tlen_ct: text length in bytes, or 0 when taken control transfers cannot be inlined
//   (no C extension); cfg: bit 1 rv32, bit 2 mstatus.FS dirty (flw / fld / fsw / fsd inline), bit 3 D extension (flw boxes);
//   xb: LDS byte offset of the lane's x block; arena: the lane's arena base (DevMem::arena); vpc: the lane's pc as a text offset,
//   0xffffffff for lanes that take no part; n: iterations (>= 1).  Out: vit = instructions retired by the lane, ncv != 0 = the lane
//   stands at an instruction for the general path.
__device__ __forceinline__ void conv_core(const void* recs, u64 nmbase, u64 msize, u64 tlo, u64 thi, u64 ntlo, u32 tlen_ct, u32 amhi,
                                          u32 cfg, u32 lpw_shift, u32 foff, u32 n, u32 xb, const void* arena, u64 tohost,
                                          u32& vpc, u32& vit, u32& ncv)
{
  u32 vit_ = 0, ncv_ = 0;
  asm volatile(
    // ------------------------------------------------------------------ entry
    "s_mov_b64 s[72:73], exec\n"
    "ds_read_b64 v[130:131], %[xb] offset:8\n"    "ds_read_b64 v[132:133], %[xb] offset:16\n"   "ds_read_b64 v[134:135], %[xb] offset:24\n"
    "ds_read_b64 v[136:137], %[xb] offset:32\n"   "ds_read_b64 v[138:139], %[xb] offset:40\n"   "ds_read_b64 v[140:141], %[xb] offset:48\n"
    "ds_read_b64 v[142:143], %[xb] offset:56\n"   "ds_read_b64 v[144:145], %[xb] offset:64\n"   "ds_read_b64 v[146:147], %[xb] offset:72\n"
    "ds_read_b64 v[148:149], %[xb] offset:80\n"   "ds_read_b64 v[150:151], %[xb] offset:88\n"   "ds_read_b64 v[152:153], %[xb] offset:96\n"
    "ds_read_b64 v[154:155], %[xb] offset:104\n"  "ds_read_b64 v[156:157], %[xb] offset:112\n"  "ds_read_b64 v[158:159], %[xb] offset:120\n"
    "ds_read_b64 v[160:161], %[xb] offset:128\n"  "ds_read_b64 v[162:163], %[xb] offset:136\n"  "ds_read_b64 v[164:165], %[xb] offset:144\n"
    "ds_read_b64 v[166:167], %[xb] offset:152\n"  "ds_read_b64 v[168:169], %[xb] offset:160\n"  "ds_read_b64 v[170:171], %[xb] offset:168\n"
    "ds_read_b64 v[172:173], %[xb] offset:176\n"  "ds_read_b64 v[174:175], %[xb] offset:184\n"  "ds_read_b64 v[176:177], %[xb] offset:192\n"
    "ds_read_b64 v[178:179], %[xb] offset:200\n"  "ds_read_b64 v[180:181], %[xb] offset:208\n"  "ds_read_b64 v[182:183], %[xb] offset:216\n"
    "ds_read_b64 v[184:185], %[xb] offset:224\n"  "ds_read_b64 v[186:187], %[xb] offset:232\n"  "ds_read_b64 v[188:189], %[xb] offset:240\n"
    "ds_read_b64 v[190:191], %[xb] offset:248\n"
    "v_mov_b32 v192, 0\n"  "v_mov_b32 v193, 0\n"  "v_mov_b32 v128, 0\n"  "v_mov_b32 v129, 0\n"
    "s_mov_b64 s[86:87], %[recs]\n"   "s_mov_b64 s[74:75], %[nmbase]\n"  "s_mov_b64 s[76:77], %[msize]\n"  "s_mov_b64 s[78:79], %[tlo]\n"
    "s_mov_b64 s[80:81], %[thi]\n"    "s_mov_b64 s[82:83], %[ntlo]\n"    "s_mov_b32 s88, %[tlen]\n"        "s_mov_b32 s89, %[amhi]\n"
    "s_mov_b32 s90, %[cfg]\n"         "s_mov_b32 s91, %[lpwsh]\n"        "s_mov_b32 s92, %[foff]\n"        "s_mov_b32 s54, %[n]\n"
    "s_mov_b32 s93, -1\n"
    "s_getpc_b64 s[60:61]\n"
    "L_anchor_%=:\n"
    "s_add_u32 s60, s60, L_htab_%=-L_anchor_%=\n"
    "s_addc_u32 s61, s61, 0\n"
    "v_cmp_ne_u32_e64 s[84:85], -1, %[vpc]\n"
    "v_lshl_add_u64 v[210:211], %[tohost], 0, 4\n"
    CC_DPPMIN("%[vpc]", "s52")
    "s_lshl_b32 s55, s52, 3\n"
    "s_load_dwordx4 s[40:43], s[86:87], s55\n"
    "s_waitcnt lgkmcnt(0)\n"
    // ------------------------------------------------------------------ one step: s[40:43] = record at upc (s52)
    "L_top_%=:\n"
    "s_bfe_u32 s55, s40, 0x1001c\n"               // F_SIZE4
    "s_lshl1_add_u32 s53, s55, s52\n"
    "s_add_u32 s53, s53, 2\n"                     // spc
    "s_lshl_b32 s55, s53, 3\n"
    "s_load_dwordx4 s[44:47], s[86:87], s55\n"    // sequential successor, in flight for the rest of the step
    "v_cmp_eq_u32_e32 vcc, s52, %[vpc]\n"
    "s_and_saveexec_b64 s[62:63], vcc\n"          // the group: lanes whose pc is the scheduled one
    "s_bfe_u32 s55, s42, 0x70002\n"               // slot(rs1) * 2
    "s_bfe_u32 s56, s42, 0x70012\n"               // slot(rs2) * 2 (zero slot for immediate forms)
    "s_set_gpr_idx_on s55, gpr_idx(SRC0)\n"
    "v_mov_b32 v194, v128\n"
    "v_mov_b32 v195, v129\n"
    "s_set_gpr_idx_idx s56\n"
    "v_mov_b32 v196, v128\n"
    "v_mov_b32 v197, v129\n"
    "s_set_gpr_idx_off\n"
    "s_mov_b32 s58, s41\n"
    "s_ashr_i32 s59, s41, 31\n"
    "s_bfe_u32 s57, s40, 0x60007\n"               // rd * 2
    "s_and_b32 s55, s42, 0xfe00\n"                // handler class * 512
    "s_add_u32 s64, s60, s55\n"
    "s_addc_u32 s65, s61, 0\n"
    "v_lshl_add_u64 v[198:199], v[196:197], 0, s[58:59]\n"
    "s_setpc_b64 s[64:65]\n"
    // ------------------------------------------------------------------ shared pieces
    "L_bad_%=:\n"                                  // nobody of the group executes: general path
    "v_mov_b32 %[ncv], 1\n"
    "s_mov_b64 exec, s[62:63]\n"
    "s_branch L_exit_%=\n"
    // conditional branch, target inside the text: vcc = takers, s93 = target, its record requested into s[48:51]
    "L_br_%=:\n"
    "v_mov_b32 v202, s53\n"
    "v_mov_b32 v203, s93\n"
    "s_mov_b64 s[66:67], vcc\n"
    "s_xor_b64 s[68:69], exec, s[84:85]\n"        // scc: the group is not every eligible lane
    "v_cndmask_b32_e32 %[vpc], v202, v203, vcc\n"
    "v_add_u32 %[vit], 1, %[vit]\n"
    "s_mov_b64 exec, s[62:63]\n"
    "s_cbranch_scc1 L_sched_%=\n"
    "s_cmp_eq_u64 s[66:67], 0\n"
    "s_cbranch_scc1 L_nseq_%=\n"                   // nobody took it
    "s_cmp_eq_u64 s[66:67], s[84:85]\n"
    "s_cbranch_scc0 L_sched_%=\n"                  // some did, some did not
    "L_ntgt_%=:\n"                                 // everybody goes to the target
    "s_waitcnt lgkmcnt(0)\n"
    "s_mov_b64 s[40:41], s[48:49]\n"
    "s_mov_b64 s[42:43], s[50:51]\n"
    "s_mov_b32 s52, s93\n"
    "s_sub_u32 s54, s54, 1\n"
    "s_cmp_lg_u32 s54, 0\n"
    "s_cbranch_scc1 L_top_%=\n"
    "s_branch L_exit_%=\n"
    "L_nseq_%=:\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_mov_b64 s[40:41], s[44:45]\n"
    "s_mov_b64 s[42:43], s[46:47]\n"
    "s_mov_b32 s52, s53\n"
    "s_sub_u32 s54, s54, 1\n"
    "s_cmp_lg_u32 s54, 0\n"
    "s_cbranch_scc1 L_top_%=\n"
    "s_branch L_exit_%=\n"
    // conditional branch whose target lies outside the text (or no C extension): fine while nobody takes it
    "L_br_out_%=:\n"
    "s_cmp_eq_u64 vcc, 0\n"
    "s_cbranch_scc0 L_bad_%=\n"
    CC_LANES_SEQ CC_NEXT_SEQ
    // after a control transfer: the next scheduled pc is the minimum over all eligible lanes (exec = all lanes here)
    "L_sched_%=:\n"
    "s_sub_u32 s54, s54, 1\n"
    "s_cmp_eq_u32 s54, 0\n"
    "s_cbranch_scc1 L_exit_%=\n"
    CC_DPPMIN("%[vpc]", "s98")
    "s_waitcnt lgkmcnt(0)\n"
    "s_cmp_eq_u32 s98, s53\n"
    "s_cbranch_scc1 L_sched_seq_%=\n"
    "s_cmp_eq_u32 s98, s93\n"
    "s_cbranch_scc1 L_sched_tgt_%=\n"
    "s_lshl_b32 s55, s98, 3\n"
    "s_load_dwordx4 s[40:43], s[86:87], s55\n"
    "s_mov_b32 s52, s98\n"
    "s_waitcnt lgkmcnt(0)\n"
    "s_branch L_top_%=\n"
    "L_sched_seq_%=:\n"
    "s_mov_b64 s[40:41], s[44:45]\n"
    "s_mov_b64 s[42:43], s[46:47]\n"
    "s_mov_b32 s52, s98\n"
    "s_branch L_top_%=\n"
    "L_sched_tgt_%=:\n"
    "s_mov_b64 s[40:41], s[48:49]\n"
    "s_mov_b64 s[42:43], s[50:51]\n"
    "s_mov_b32 s52, s98\n"
    "s_branch L_top_%=\n"
    // jal / jalr tail: link in s[66:67] (64-bit), lanes' new pc in v203, s93 = uniform target or -1; the target's record may be in flight
    "L_jump_%=:\n"
    "s_ashr_i32 s55, s66, 31\n"
    "s_bitcmp1_b32 s90, 1\n"
    "s_cselect_b32 s67, s55, s67\n"               // rv32: the link register holds the sign-extended 32-bit pc
    "v_mov_b32 v200, s66\n"
    "v_mov_b32 v201, s67\n"
    CC_WR
    "v_mov_b32 %[vpc], v203\n"
    "v_add_u32 %[vit], 1, %[vit]\n"
    "s_xor_b64 s[68:69], exec, s[84:85]\n"
    "s_mov_b64 exec, s[62:63]\n"
    "s_cbranch_scc1 L_sched_%=\n"
    "s_cmp_eq_u32 s93, -1\n"
    "s_cbranch_scc1 L_sched_%=\n"
    "s_branch L_ntgt_%=\n"
    // ------------------------------------------------------------------ handlers, 512 bytes apart
    ".p2align 9\n"
    "L_htab_%=:\n"
    "s_branch L_bad_%=\n"                                                                                   // HC_BAD
    CC_H(1)  "v_lshl_add_u64 v[200:201], v[194:195], 0, v[198:199]\n" CC_TAIL_WR                            // add
    CC_H(2)  "v_sub_co_u32_e32 v200, vcc, v194, v198\n" "s_nop 1\n" "v_subb_co_u32_e32 v201, vcc, v195, v199, vcc\n" CC_TAIL_WR
    CC_H(3)  "v_lshlrev_b64 v[200:201], v198, v[194:195]\n" CC_TAIL_WR                                      // sll
    CC_H(4)  "v_cmp_lt_i64_e32 vcc, v[194:195], v[198:199]\n" "v_mov_b32 v201, 0\n" "s_nop 0\n" "v_cndmask_b32_e64 v200, 0, 1, vcc\n" CC_TAIL_WR
    CC_H(5)  "v_cmp_lt_u64_e32 vcc, v[194:195], v[198:199]\n" "v_mov_b32 v201, 0\n" "s_nop 0\n" "v_cndmask_b32_e64 v200, 0, 1, vcc\n" CC_TAIL_WR
    CC_H(6)  "v_xor_b32 v200, v194, v198\n" "v_xor_b32 v201, v195, v199\n" CC_TAIL_WR
    CC_H(7)  "v_lshrrev_b64 v[200:201], v198, v[194:195]\n" CC_TAIL_WR                                      // srl
    CC_H(8)  "v_ashrrev_i64 v[200:201], v198, v[194:195]\n" CC_TAIL_WR                                      // sra
    CC_H(9)  "v_or_b32 v200, v194, v198\n" "v_or_b32 v201, v195, v199\n" CC_TAIL_WR
    CC_H(10) "v_and_b32 v200, v194, v198\n" "v_and_b32 v201, v195, v199\n" CC_TAIL_WR
    CC_H(11) "v_add_u32 v200, v194, v198\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                     // addw
    CC_H(12) "v_sub_u32 v200, v194, v198\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                     // subw
    CC_H(13) "v_lshlrev_b32 v200, v198, v194\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                 // sllw
    CC_H(14) "v_lshrrev_b32 v200, v198, v194\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                 // srlw
    CC_H(15) "v_ashrrev_i32 v200, v198, v194\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                 // sraw
    CC_H(16) "v_mad_u64_u32 v[200:201], s[66:67], v194, v198, 0\n" "v_mul_lo_u32 v202, v195, v198\n" "v_mul_lo_u32 v203, v194, v199\n"
             "v_add3_u32 v201, v201, v202, v203\n" CC_TAIL_WR                                               // mul
    CC_H(17) "v_mul_lo_u32 v200, v194, v198\n" "v_ashrrev_i32 v201, 31, v200\n" CC_TAIL_WR                  // mulw
    CC_H(18) "s_add_u32 s66, s78, s52\n" "s_addc_u32 s67, s79, 0\n" "s_add_u32 s66, s66, s58\n" "s_addc_u32 s67, s67, s59\n"   // auipc
             "s_ashr_i32 s55, s66, 31\n" "s_bitcmp1_b32 s90, 1\n" "s_cselect_b32 s67, s55, s67\n"
             "v_mov_b32 v200, s66\n" "v_mov_b32 v201, s67\n" CC_TAIL_WR
    CC_H(19) "s_add_u32 s93, s52, s41\n"                                                                    // jal
             "s_cmp_ge_u32 s93, s88\n"
             "s_cbranch_scc1 L_bad_%=\n"
             "s_lshl_b32 s55, s93, 3\n"
             "s_load_dwordx4 s[48:51], s[86:87], s55\n"
             "s_add_u32 s66, s78, s53\n" "s_addc_u32 s67, s79, 0\n"
             "v_mov_b32 v203, s93\n"
             "s_branch L_jump_%=\n"
    CC_H(20) "v_lshl_add_u64 v[202:203], v[194:195], 0, s[58:59]\n"                                         // jalr
             "v_and_b32 v203, s89, v203\n"
             "v_and_b32 v202, -2, v202\n"
             "v_lshl_add_u64 v[202:203], v[202:203], 0, s[82:83]\n"
             "s_add_u32 s66, s78, s53\n" "s_addc_u32 s67, s79, 0\n"
             "v_cmp_eq_u32_e64 s[68:69], 0, v203\n"
             "v_cmp_gt_u32_e32 vcc, s88, v202\n"
             "s_and_b64 s[68:69], s[68:69], vcc\n"
             "s_xor_b64 s[68:69], s[68:69], exec\n"
             "s_cbranch_scc1 L_bad_%=\n"
             "v_mov_b32 v203, v202\n"
             "s_mov_b32 s93, -1\n"
             "s_branch L_jump_%=\n"
    CC_H(21) "v_cmp_eq_u64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(22) "v_cmp_ne_u64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(23) "v_cmp_lt_i64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(24) "v_cmp_ge_i64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(25) "v_cmp_lt_u64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(26) "v_cmp_ge_u64_e32 vcc, v[194:195], v[196:197]\n" CC_BRANCH
    CC_H(27) CC_LOAD("1", "0", "global_load_sbyte v200,", "v_ashrrev_i32 v201, 31, v200\n")
    CC_H(28) CC_LOAD("2", "1", "global_load_sshort v200,", "v_ashrrev_i32 v201, 31, v200\n")
    CC_H(29) CC_LOAD("4", "3", "global_load_dword v200,", "v_ashrrev_i32 v201, 31, v200\n")
    CC_H(30) CC_LOAD("8", "7", "global_load_dwordx2 v[200:201],", "")
    CC_H(31) CC_LOAD("1", "0", "global_load_ubyte v200,", "v_mov_b32 v201, 0\n")
    CC_H(32) CC_LOAD("2", "1", "global_load_ushort v200,", "v_mov_b32 v201, 0\n")
    CC_H(33) CC_LOAD("4", "3", "global_load_dword v200,", "v_mov_b32 v201, 0\n")
    CC_H(34) CC_STORE("1", "0", "global_store_byte v[206:207], v196, off")
    CC_H(35) CC_STORE("2", "1", "global_store_short v[206:207], v196, off")
    CC_H(36) CC_STORE("4", "3", "global_store_dword v[206:207], v196, off")
    CC_H(37) CC_STORE("8", "7", "global_store_dwordx2 v[206:207], v[196:197], off")
    CC_H(38) CC_LANES_SEQ CC_NEXT_SEQ                                                                       // nop
    CC_H(39) "s_bitcmp1_b32 s90, 2\n" "s_cbranch_scc0 L_bad_%=\n"                                           // flw
             CC_ADDR("4", "3")
             "s_xor_b64 s[68:69], s[68:69], exec\n" "s_cbranch_scc1 L_bad_%=\n"
             CC_LANEPTR
             "global_load_dword v200, v[206:207], off\n"
             "s_lshl_b32 s55, s57, 2\n" "s_add_u32 s55, s55, s92\n" "v_add_u32 v202, s55, %[xb]\n"
             "s_bitcmp1_b32 s90, 3\n" "s_cselect_b32 s56, -1, 0\n" "v_mov_b32 v201, s56\n"
             CC_LANES_SEQ
             "s_waitcnt vmcnt(0)\n"
             "ds_write_b64 v202, v[200:201]\n"
             CC_NEXT_SEQ
    CC_H(40) "s_bitcmp1_b32 s90, 2\n" "s_cbranch_scc0 L_bad_%=\n"                                           // fld
             CC_ADDR("8", "7")
             "s_xor_b64 s[68:69], s[68:69], exec\n" "s_cbranch_scc1 L_bad_%=\n"
             CC_LANEPTR
             "global_load_dwordx2 v[200:201], v[206:207], off\n"
             "s_lshl_b32 s55, s57, 2\n" "s_add_u32 s55, s55, s92\n" "v_add_u32 v202, s55, %[xb]\n"
             CC_LANES_SEQ
             "s_waitcnt vmcnt(0)\n"
             "ds_write_b64 v202, v[200:201]\n"
             CC_NEXT_SEQ
    CC_H(41) "s_bitcmp1_b32 s90, 2\n" "s_cbranch_scc0 L_bad_%=\n"                                           // fsw
             "s_lshr_b32 s55, s40, 15\n" "s_and_b32 s55, s55, 0xf8\n" "s_add_u32 s55, s55, s92\n" "v_add_u32 v209, s55, %[xb]\n"
             "ds_read_b64 v[196:197], v209\n"
             CC_ADDR("4", "3")
             CC_STCHK("4")
             "s_xor_b64 s[68:69], s[68:69], exec\n" "s_cbranch_scc1 L_bad_%=\n"
             CC_LANEPTR
             "s_waitcnt lgkmcnt(0)\n"
             "global_store_dword v[206:207], v196, off\n"
             CC_LANES_SEQ CC_NEXT_SEQ
    CC_H(42) "s_bitcmp1_b32 s90, 2\n" "s_cbranch_scc0 L_bad_%=\n"                                           // fsd
             "s_lshr_b32 s55, s40, 15\n" "s_and_b32 s55, s55, 0xf8\n" "s_add_u32 s55, s55, s92\n" "v_add_u32 v209, s55, %[xb]\n"
             "ds_read_b64 v[196:197], v209\n"
             CC_ADDR("8", "7")
             CC_STCHK("8")
             "s_xor_b64 s[68:69], s[68:69], exec\n" "s_cbranch_scc1 L_bad_%=\n"
             CC_LANEPTR
             "s_waitcnt lgkmcnt(0)\n"
             "global_store_dwordx2 v[206:207], v[196:197], off\n"
             CC_LANES_SEQ CC_NEXT_SEQ
    CC_H(43)
    // ------------------------------------------------------------------ exit: x file back to LDS
    "L_exit_%=:\n"
    "s_mov_b64 exec, s[84:85]\n"                   // only lanes that took part own an x block (the others alias lane 0's / their owner's)
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
    "ds_write_b64 %[xb], v[130:131] offset:8\n"    "ds_write_b64 %[xb], v[132:133] offset:16\n"   "ds_write_b64 %[xb], v[134:135] offset:24\n"
    "ds_write_b64 %[xb], v[136:137] offset:32\n"   "ds_write_b64 %[xb], v[138:139] offset:40\n"   "ds_write_b64 %[xb], v[140:141] offset:48\n"
    "ds_write_b64 %[xb], v[142:143] offset:56\n"   "ds_write_b64 %[xb], v[144:145] offset:64\n"   "ds_write_b64 %[xb], v[146:147] offset:72\n"
    "ds_write_b64 %[xb], v[148:149] offset:80\n"   "ds_write_b64 %[xb], v[150:151] offset:88\n"   "ds_write_b64 %[xb], v[152:153] offset:96\n"
    "ds_write_b64 %[xb], v[154:155] offset:104\n"  "ds_write_b64 %[xb], v[156:157] offset:112\n"  "ds_write_b64 %[xb], v[158:159] offset:120\n"
    "ds_write_b64 %[xb], v[160:161] offset:128\n"  "ds_write_b64 %[xb], v[162:163] offset:136\n"  "ds_write_b64 %[xb], v[164:165] offset:144\n"
    "ds_write_b64 %[xb], v[166:167] offset:152\n"  "ds_write_b64 %[xb], v[168:169] offset:160\n"  "ds_write_b64 %[xb], v[170:171] offset:168\n"
    "ds_write_b64 %[xb], v[172:173] offset:176\n"  "ds_write_b64 %[xb], v[174:175] offset:184\n"  "ds_write_b64 %[xb], v[176:177] offset:192\n"
    "ds_write_b64 %[xb], v[178:179] offset:200\n"  "ds_write_b64 %[xb], v[180:181] offset:208\n"  "ds_write_b64 %[xb], v[182:183] offset:216\n"
    "ds_write_b64 %[xb], v[184:185] offset:224\n"  "ds_write_b64 %[xb], v[186:187] offset:232\n"  "ds_write_b64 %[xb], v[188:189] offset:240\n"
    "ds_write_b64 %[xb], v[190:191] offset:248\n"
    "s_mov_b64 exec, s[72:73]\n"
    "s_waitcnt lgkmcnt(0)\n"
    : [vpc] "+v"(vpc), [vit] "+v"(vit_), [ncv] "+v"(ncv_)
    : [recs] "s"(recs), [nmbase] "s"(nmbase), [msize] "s"(msize), [tlo] "s"(tlo), [thi] "s"(thi), [ntlo] "s"(ntlo), [tlen] "s"(tlen_ct),
      [amhi] "s"(amhi), [cfg] "s"(cfg), [lpwsh] "s"(lpw_shift), [foff] "s"(foff), [n] "s"(n), [xb] "v"(xb), [arena] "v"(arena), [tohost] "v"(tohost)
    : CC_CLOBBERS);
  vit = vit_; ncv = ncv_;
}

}  // namespace wsp

#else   // host pass of the same translation unit: the kernels' bodies are parsed but never run there
namespace wsp {
__device__ void conv_core(const void* recs, u64 nmbase, u64 msize, u64 tlo, u64 thi, u64 ntlo, u32 tlen_ct, u32 amhi, u32 cfg, u32 lpw_shift,
                          u32 foff, u32 n, u32 xb, const void* arena, u64 tohost, u32& vpc, u32& vit, u32& ncv);
}
#endif  // __HIP_DEVICE_COMPILE__
