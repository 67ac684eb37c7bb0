// Per-hart step semantics of the batched RISC-V simulator (one hart per GPU lane).
//
// This header is the product's restatement of the reference hot path
//   Hart<URV>::untilAddress / singleStep   /root/reference/Hart.cpp:6003-6204, 7417-7529
//   execute() switch and exec* bodies       Hart.cpp:7544-11480, float.cpp, amo.cpp
//   traps / xRET / CSR instructions         Hart.cpp:3324-3929, 12247-12560, 12723-13748
// re-designed for lock-step SIMT execution: instructions are pre-decoded once per program into
// 16-byte records (replacing Decoder::decode + the per-hart 128Ki-entry decode cache,
// Hart.cpp:6113-6117), integer/FP registers live in LDS as [reg][lane], cold CSRs and the RAM
// arena live in HBM, and everything a hart touches every step sits in registers (struct Hot).
//
// Everything is WSP_HD so tests can also compile the decoder and softfloat for the host
// (tests never run a CPU build of the step loop as the product: the C-ABI run entry point is
// HIP-only and fails loudly without a device).
#pragma once
#include <stdint.h>
#include "wsp_softfloat.h"

#ifndef WSP_PROF_ADD
#define WSP_PROF_ADD(k, v) do { } while (0)
#define WSP_PROF_T() 0ULL
#endif

namespace wsp {

typedef uint64_t u64;
typedef int64_t  i64;
typedef uint32_t u32;
typedef int32_t  i32;
typedef uint16_t u16;
typedef uint8_t  u8;

// ---------------------------------------------------------------------------------- constants
enum { ISA_M = 1, ISA_A = 2, ISA_F = 4, ISA_D = 8, ISA_C = 16, ISA_S = 32, ISA_U = 64, ISA_V = 128, ISA_ZICNTR = 256,
       // System<uint32_t> (Session.cpp:1373-1424).  An RV32 hart keeps its x registers SIGN-EXTENDED in the 64-bit register file and
       // its pc zero-extended; the decoder lowers the RV32 integer operations onto the RV64 W-forms, which compute exactly the RV32
       // results in that representation (add -> addw, sll -> sllw, mul -> mulw, div -> divw, ...; slt / sltu / and / or / xor are
       // the same on sign-extended values), addresses wrap at 32 bits, and the CSR file is kept in its RV64 shape behind 32-bit
       // views (csr_peek / csr_store).  Modelled for M / U mode with I M A F D C Zicsr Zicntr: no S mode (Sv32) and no V.
       ISA_RV32 = 512,
       // bit manipulation and conditional zero (bitmanip.cpp, Hart.cpp:14728-14760): --isa ..._zba_zbb_zbs_zbc_zicond
       ISA_ZBA = 1 << 10, ISA_ZBB = 1 << 11, ISA_ZBS = 1 << 12, ISA_ZICOND = 1 << 13, ISA_ZBC = 1 << 14,
       ISA_ZB_ANY = ISA_ZBA | ISA_ZBB | ISA_ZBS | ISA_ZICOND | ISA_ZBC };
// Configuration switches (HartConfig::applyConfig tags, HartConfig.cpp:2085-2958) travel in the same word as the ISA bits: they are
// fixed for the lifetime of a batch exactly as the ISA is.  All zero = the reference's defaults.
enum : u32 {
  CFG_NO_MISAL = 1u << 16,          // enable_misaligned_data = false: misaligned loads / stores trap (Hart.cpp:2013-2016, 13810-13813)
  CFG_CLR_MTVAL_ILL = 1u << 17,     // clear_mtval_on_illegal_instruction (Hart.cpp:3256)
  CFG_CLR_MTVAL_EBREAK = 1u << 18,  // clear_mtval_on_ebreak (Hart.cpp:12097)
  CFG_CANCEL_LR_TRAP = 1u << 19,    // cancel_lr_on_trap (Hart.cpp:3570, 4311)
  CFG_KEEP_MPRV = 1u << 20,         // clear_mprv_on_ret = false (Hart.cpp:12283, 12499)
  CFG_NO_SV39 = 1u << 21            // address_translation_modes without sv39: a satp write selecting it is dropped (Hart.cpp:13284-13294)
};
enum { ST_RUNNING = 0, ST_STOP_TOHOST = 1, ST_STOP_ILLEGAL = 2, ST_STOP_LIMIT = 3, ST_STOP_UNSUPPORTED = 4, ST_STOP_ADDR = 5 };
enum { PRV_U = 0, PRV_S = 1, PRV_M = 3 };

// Unsupported-reason codes reported in wsp_result.stop_value when status == ST_STOP_UNSUPPORTED
enum { UNS_PAGING = 0x180, UNS_PMP_STALE = 0x181, UNS_CSR_BASE = 0x10000, UNS_VECTOR = 0x20000 /* | instruction << 20 */ };

enum Op : u32 {
  OP_ILLEGAL = 0,
  // ALU: second operand = imm when F_IMM is set, else x[rs2]
  OP_ADD, OP_SUB, OP_SLL, OP_SLT, OP_SLTU, OP_XOR, OP_SRL, OP_SRA, OP_OR, OP_AND,
  OP_ADDW, OP_SUBW, OP_SLLW, OP_SRLW, OP_SRAW,
  OP_MUL, OP_MULH, OP_MULHSU, OP_MULHU, OP_DIV, OP_DIVU, OP_REM, OP_REMU,
  OP_MULW, OP_DIVW, OP_DIVUW, OP_REMW, OP_REMUW,
  OP_AUIPC, OP_JAL, OP_JALR, OP_BEQ, OP_BNE, OP_BLT, OP_BGE, OP_BLTU, OP_BGEU,
  OP_LB, OP_LH, OP_LW, OP_LD, OP_LBU, OP_LHU, OP_LWU, OP_SB, OP_SH, OP_SW, OP_SD,
  OP_NOP, OP_ECALL, OP_EBREAK, OP_MRET, OP_SRET, OP_WFI, OP_SFENCE_VMA,
  OP_CSRRW, OP_CSRRS, OP_CSRRC,
  OP_LR, OP_SC, OP_AMO,                 // F_DBL = 64-bit form; AMO function (funct5) in imm
  OP_FLW, OP_FLD, OP_FSW, OP_FSD,
  // F/D: F_DBL selects double precision; rounding mode field in imm[2:0]
  OP_FADD, OP_FSUB, OP_FMUL, OP_FDIV, OP_FSQRT, OP_FMA /* imm[4:3]: 0 madd 1 msub 2 nmsub 3 nmadd */,
  OP_FSGNJ /* imm: 0 j 1 jn 2 jx */, OP_FMINMAX /* imm: 0 min 1 max */, OP_FCMP /* imm: 0 le 1 lt 2 eq */,
  OP_FCVT_F2I /* rs2: 0 w 1 wu 2 l 3 lu */, OP_FCVT_I2F, OP_FCVT_FF /* to F_DBL ? d.s : s.d */,
  OP_FMV_F2X, OP_FCLASS, OP_FMV_X2F,
  OP_ZB,                                // Zba / Zbb / Zbs / Zbc / Zicond: function (ZB_*) in the rs3 field, a parameter in imm for register forms
  OP_VEC, OP_VLD, OP_VST,               // RVV: OP-V (0x57), vector load, vector store; fields are taken from Rec.raw
  OP_SLOWFETCH,                         // predecoded slot cannot be trusted: fetch + decode from RAM
  OP_STOPCHK,                           // some hart's stop address (wsp_batch_set_stop): never in the inline subset, so the general
                                        // path sees it, compares its own stop address and otherwise decodes Rec.raw again
  OP_COUNT
};

// Pre-decoded instruction record (replaces DecodedInst, DecodedInst.hpp:640-657).
struct Rec {
  u32 w0;     // op[7:0] rd[12:8] rs1[17:13] rs2[22:18] rs3[27:23] F_SIZE4[28] F_IMM[29] F_DBL[30]
  i32 imm;
  u32 pad;    // hot-loop operand offsets, filled by finish_rec(): LDS byte offset of x[rs1] | x[rs2] << 16 (x0 -> zero slot 32)
  u32 raw;    // instruction bits as fetched (16-bit masked when compressed): trace + mtval.  Last: the untraced hot loop
              // loads only the first three words
};
enum : u32 { F_SIZE4 = 1u << 28, F_IMM = 1u << 29, F_DBL = 1u << 30 };
WSP_HD u32 rec_op(const Rec& r) { return r.w0 & 0xff; }
WSP_HD u32 rec_rd(const Rec& r) { return (r.w0 >> 8) & 31; }
WSP_HD u32 rec_rs1(const Rec& r) { return (r.w0 >> 13) & 31; }
WSP_HD u32 rec_rs2(const Rec& r) { return (r.w0 >> 18) & 31; }
WSP_HD u32 rec_rs3(const Rec& r) { return (r.w0 >> 23) & 31; }
// hot-loop operand fields: LDS byte offsets of x[rs1] (low half) and x[rs2] (high half; x0 and immediate forms -> the zero slot 32)
WSP_HD void finish_rec(Rec& r)
{
  const u32 s1 = rec_rs1(r), s2 = (r.w0 & F_IMM) ? 0 : rec_rs2(r);
  r.pad = ((s1 ? s1 : 32u) * 8u) | (((s2 ? s2 : 32u) * 8u) << 16);
}
WSP_HD Rec mk(u32 op, u32 rd, u32 rs1, u32 rs2, i32 imm, u32 flags = 0, u32 rs3 = 0)
{ Rec r; r.w0 = op | (rd << 8) | (rs1 << 13) | (rs2 << 18) | (rs3 << 23) | flags; r.imm = imm; r.raw = 0; r.pad = 0; return r; }

// ---------------------------------------------------------------------------------- hashing
// The commit-trace / state / memory digests; oracle/rv_oracle.c computes the same folds.
WSP_HD u64 rotl64(u64 x, unsigned r) { return (x << r) | (x >> (64 - r)); }
WSP_HD u64 fold(u64 h, u64 w) { h ^= w; h = rotl64(h, 27); return h * 5 + 0x52dce729u; }
WSP_HD u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
static constexpr u64 GOLDEN = 0x9E3779B97F4A7C15ULL;

// ---------------------------------------------------------------------------------- decode
WSP_HD i32 sx(u32 v, unsigned bits) { return (i32)(v << (32 - bits)) >> (32 - bits); }

// RV32: the integer operations that differ between the two widths are executed as the RV64 W-forms (see ISA_RV32)
WSP_HD Rec lower_rv32(Rec r)
{
  u32 op = r.w0 & 0xff, n = op;
  switch (op) {
    case OP_ADD: n = OP_ADDW; break; case OP_SUB: n = OP_SUBW; break; case OP_SLL: n = OP_SLLW; break; case OP_SRL: n = OP_SRLW; break;
    case OP_SRA: n = OP_SRAW; break; case OP_MUL: n = OP_MULW; break; case OP_DIV: n = OP_DIVW; break; case OP_DIVU: n = OP_DIVUW; break;
    case OP_REM: n = OP_REMW; break; case OP_REMU: n = OP_REMUW; break; default: break;
  }
  r.w0 = (r.w0 & ~0xffu) | n;
  return r;
}

// ---------------------------------------------------------------------------------- hot records
// The hand-written converged loop (wsp_conv_core.h) reads its own image of the program: one 32-byte record per 2-byte slot, every
// field already in the form the loop uses it, so that a step spends no scalar instruction on decoding:
//   w0 handler offset (class * 512)   w1 instruction size   w2:w3 immediate, sign-extended to 64 bits
//   w4 / w5 VGPR index (slot * 2) of x[rs1] / x[rs2]; x0, immediate forms, loads, stores and jumps -> the zero slot 32
//   w6 class-specific: VGPR index of x[rd]; stores: of the data register; flw / fld / fsw / fsd: LDS byte offset of the f register
//   w7 class-specific: branches / jal: the target as a text offset; HC_PURE_*: Rec.w0
// What does not depend on the hart is folded at pre-decode: auipc is an add of the constant pc + imm to the zero slot, jal's link
// value is its immediate, branch targets are checked against the text once (a target outside it - or any taken transfer without
// the C extension, which can trap - gets a class that is fine only while nobody takes it).
enum : u32 { HC_BAD = 0, HC_ADD, HC_SUB, HC_SLL, HC_SLT, HC_SLTU, HC_XOR, HC_SRL, HC_SRA, HC_OR, HC_AND, HC_ADDW, HC_SUBW, HC_SLLW,
             HC_SRLW, HC_SRAW, HC_MUL, HC_MULW, HC_JAL, HC_JALR,
             HC_BEQ, HC_BNE, HC_BLT, HC_BGE, HC_BLTU, HC_BGEU,                    // forward: falls through when nobody takes it
             HC_BEQ_B, HC_BNE_B, HC_BLT_B, HC_BGE_B, HC_BLTU_B, HC_BGEU_B,        // backward: falls through when everybody takes it
             HC_BEQ_O, HC_BNE_O, HC_BLT_O, HC_BGE_O, HC_BLTU_O, HC_BGEU_O,        // target cannot be inlined: fine while not taken
             HC_LB, HC_LH, HC_LW, HC_LD, HC_LBU, HC_LHU, HC_LWU, HC_SB, HC_SH, HC_SW, HC_SD, HC_NOP, HC_FLW, HC_FLD, HC_FSW, HC_FSD,
             HC_PURE_INT,                                                         // mulh* div* rem* Zb*: x[rd] = f(x[rs1], x[rs2] | imm)
             HC_PURE_FP_F, HC_PURE_FP_X,                                          // F / D computational: result to f[rd] / to x[rd]
             HC_COUNT };
struct HRec { u32 w[8]; };
WSP_HD HRec mk_hrec(const Rec& r, u32 isa, u64 text_lo, u32 text_len, u32 upc)
{
  const u32 op = r.w0 & 0xff, rs1 = rec_rs1(r), rs2 = rec_rs2(r), rd = rec_rd(r);
  const bool rv32 = (isa & ISA_RV32) != 0, zca = (isa & ISA_C) != 0;
  const u32 size = (r.w0 & F_SIZE4) ? 4u : 2u, zero = 64u;
  u32 hc = HC_BAD, i1 = (rs1 ? rs1 : 32u) * 2u, i2 = ((r.w0 & F_IMM) || !rs2) ? zero : rs2 * 2u, i3 = rd * 2u, aux = 0;
  i64 simm = (i64)r.imm;
  const u64 pc = text_lo + upc;
  if (op >= OP_ADD && op <= OP_SRAW) hc = HC_ADD + (op - OP_ADD);
  else if (op == OP_MUL) hc = HC_MUL;
  else if (op == OP_MULW) hc = HC_MULW;
  else if (op == OP_NOP) hc = HC_NOP;
  else if (op == OP_AUIPC) { const u64 v = pc + (u64)(i64)r.imm; simm = rv32 ? (i64)(i32)v : (i64)v; hc = HC_ADD; i1 = zero; i2 = zero; }
  else if (op == OP_JAL) {
    const u32 t = upc + (u32)r.imm;
    if (zca && t < text_len) { const u64 link = pc + size; simm = rv32 ? (i64)(i32)link : (i64)link; hc = HC_JAL; i1 = zero; i2 = zero; aux = t & ~1u; }
  } else if (op == OP_JALR) { if (zca) { hc = HC_JALR; i2 = zero; } }
  else if (op >= OP_BEQ && op <= OP_BGEU) {
    const u32 t = upc + (u32)r.imm;
    i2 = rs2 ? rs2 * 2u : zero;
    if (zca && t < text_len) { hc = (r.imm < 0 ? HC_BEQ_B : HC_BEQ) + (op - OP_BEQ); aux = t & ~1u; }
    else hc = HC_BEQ_O + (op - OP_BEQ);
  } else if (op >= OP_LB && op <= OP_LWU) { hc = HC_LB + (op - OP_LB); i2 = zero; }
  else if (op >= OP_SB && op <= OP_SD) { hc = HC_SB + (op - OP_SB); i2 = zero; i3 = (rs2 ? rs2 : 32u) * 2u; }
  else if (op == OP_FLW || op == OP_FLD) { hc = HC_FLW + (op - OP_FLW); i2 = zero; i3 = rd * 8u; }
  else if (op == OP_FSW || op == OP_FSD) { hc = HC_FLW + (op - OP_FLW); i2 = zero; i3 = rs2 * 8u; }
  else if ((op >= OP_MULH && op <= OP_REMUW) || op == OP_ZB) { hc = HC_PURE_INT; aux = r.w0; }
  else if (op >= OP_FADD && op <= OP_FMV_X2F) {
    const bool to_x = op == OP_FCMP || op == OP_FCVT_F2I || op == OP_FMV_F2X || op == OP_FCLASS;
    hc = to_x ? HC_PURE_FP_X : HC_PURE_FP_F; i2 = zero; aux = r.w0;
  }
  HRec h;
  h.w[0] = hc << 9; h.w[1] = size; h.w[2] = (u32)(u64)simm; h.w[3] = (u32)((u64)simm >> 32); h.w[4] = i1; h.w[5] = i2; h.w[6] = i3; h.w[7] = aux;
  return h;
}

// OP_ZB functions.  Register forms keep a parameter in Rec.imm (ZB_SHADD: shift | 8 for the .uw forms; ZB_CLZ / CTZ / CPOP / REV8:
// 1 = the 32-bit form); immediate forms (F_IMM) carry the shift amount / bit index there like the base shifts.
enum { ZB_SHADD, ZB_SLLIUW, ZB_ANDN, ZB_ORN, ZB_XNOR, ZB_CLZ, ZB_CTZ, ZB_CPOP, ZB_MIN, ZB_MINU, ZB_MAX, ZB_MAXU, ZB_SEXTB, ZB_SEXTH,
       ZB_ZEXTH, ZB_ROL, ZB_ROR, ZB_ROLW, ZB_RORW, ZB_ORCB, ZB_REV8, ZB_BCLR, ZB_BEXT, ZB_BINV, ZB_BSET, ZB_CZEQZ, ZB_CZNEZ,
       ZB_CLMUL, ZB_CLMULH, ZB_CLMULR };

// The encodings of those extensions (Decoder.cpp:2490-2557 OP-IMM, 2596-2627 OP-IMM-32, 2771-2864 OP, and the OP-32 block), gated
// on the extension the reference checks at execute time (bitmanip.cpp: isRvzba / isRvzbb / isRvzbs / isRvzbc, isRvzicond).  An RV32
// hart gets the 32-bit forms of the operations whose result depends on the register width.
WSP_HD Rec decode_zb(u32 inst, u32 isa)
{
  const u32 op = inst & 0x7f, rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31, f7 = inst >> 25;
  const u32 imm12 = inst >> 20, top5 = imm12 >> 7, amt = imm12 & 0x7f;
  const bool rv32 = (isa & ISA_RV32) != 0, zba = (isa & ISA_ZBA) != 0, zbb = (isa & ISA_ZBB) != 0, zbs = (isa & ISA_ZBS) != 0, zbc = (isa & ISA_ZBC) != 0;
  const u32 amax = rv32 ? 31 : 63;                                          // checkShiftImmediate (Hart.cpp:11736-11746)
  auto rr = [&](u32 fn, i32 par = 0) { return mk(OP_ZB, rd, rs1, rs2, par, 0, fn); };
  auto un = [&](u32 fn, i32 par = 0) { return mk(OP_ZB, rd, rs1, 0, par, 0, fn); };
  auto ri = [&](u32 fn, u32 imm) { return mk(OP_ZB, rd, rs1, 0, (i32)imm, F_IMM, fn); };
  const Rec none = mk(OP_ILLEGAL, 0, 0, 0, 0);
  switch (op) {
    case 0x33:
      switch (f7) {
        case 0x04: if (rv32 && zbb && f3 == 4 && rs2 == 0) return un(ZB_ZEXTH); break;                 // zext.h = pack rd, rs1, x0 (bitmanip.cpp:435-451)
        case 0x05:
          if (f3 >= 4) { if (zbb) return rr(f3 == 4 ? ZB_MIN : f3 == 5 ? ZB_MINU : f3 == 6 ? ZB_MAX : ZB_MAXU); }
          else if (f3 != 0 && zbc) return rr(f3 == 1 ? ZB_CLMUL : f3 == 2 ? ZB_CLMULR : ZB_CLMULH);
          break;
        case 0x07: if ((isa & ISA_ZICOND) && (f3 == 5 || f3 == 7)) return rr(f3 == 5 ? ZB_CZEQZ : ZB_CZNEZ); break;
        case 0x10: if (zba && (f3 == 2 || f3 == 4 || f3 == 6)) return rr(ZB_SHADD, (i32)(f3 >> 1)); break;
        case 0x14: if (zbs && f3 == 1) return rr(ZB_BSET); break;
        case 0x20: if (zbb && (f3 == 4 || f3 == 6 || f3 == 7)) return rr(f3 == 4 ? ZB_XNOR : f3 == 6 ? ZB_ORN : ZB_ANDN); break;
        case 0x24: if (zbs && (f3 == 1 || f3 == 5)) return rr(f3 == 1 ? ZB_BCLR : ZB_BEXT); break;
        case 0x30: if (zbb && (f3 == 1 || f3 == 5)) return rr(f3 == 1 ? (rv32 ? ZB_ROLW : ZB_ROL) : (rv32 ? ZB_RORW : ZB_ROR)); break;
        case 0x34: if (zbs && f3 == 1) return rr(ZB_BINV); break;
        default: break;
      }
      break;
    case 0x13:
      if (f3 == 1) {
        if (top5 == 0x0c) {
          if (!zbb) break;
          if (amt <= 2) return un(ZB_CLZ + amt, rv32 ? 1 : 0);
          if (amt == 4 || amt == 5) return un(amt == 4 ? ZB_SEXTB : ZB_SEXTH);
        } else if (top5 == 5 || top5 == 9 || top5 == 0x0d) {
          if (zbs && amt <= amax) return ri(top5 == 5 ? ZB_BSET : top5 == 9 ? ZB_BCLR : ZB_BINV, amt);
        }
      } else if (f3 == 5) {
        if (top5 == 5) { if (zbb && amt == 7) return un(ZB_ORCB); }
        else if (top5 == 9) { if (zbs && amt <= amax) return ri(ZB_BEXT, amt); }
        else if (top5 == 0x0c) { if (zbb && amt <= amax) return ri(rv32 ? ZB_RORW : ZB_ROR, amt); }
        else if (imm12 == (rv32 ? 0x698u : 0x6b8u)) { if (zbb) return un(ZB_REV8, rv32 ? 1 : 0); }
      }
      break;
    case 0x1b:                                                               // (RV64 only: decode32 has turned OP-IMM-32 / OP-32 away on RV32)
      if (f3 == 1 && (imm12 >> 6) == 2) { if (zba) return ri(ZB_SLLIUW, amt & 63); }
      else if (f3 == 1 && top5 == 0x0c && amt <= 2) { if (zbb) return un(ZB_CLZ + amt, 1); }
      else if (f3 == 5 && f7 == 0x30) { if (zbb) return ri(ZB_RORW, rs2); }
      break;
    case 0x3b:
      if (f7 == 0x04 && f3 == 0) { if (zba) return rr(ZB_SHADD, 8); }
      else if (f7 == 0x04 && f3 == 4) { if (zbb && rs2 == 0) return un(ZB_ZEXTH); }                    // zext.h = packw rd, rs1, x0 (bitmanip.cpp:498-524)
      else if (f7 == 0x10 && (f3 == 2 || f3 == 4 || f3 == 6)) { if (zba) return rr(ZB_SHADD, (i32)((f3 >> 1) | 8)); }
      else if (f7 == 0x30 && (f3 == 1 || f3 == 5)) { if (zbb) return rr(f3 == 1 ? ZB_ROLW : ZB_RORW); }
      break;
    default: break;
  }
  return none;
}

// 32-bit encodings (Decoder.cpp:2232). Extension gating that the reference performs at execute
// time against the (read-only) misa/--isa configuration is folded in here because the ISA is
// fixed for the lifetime of a batch; mstatus.FS-dependent legality stays in exec.
WSP_HD Rec decode32(u32 inst, u32 isa)
{
  u32 op = inst & 0x7f, rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31;
  u32 f7 = inst >> 25;
  i32 immi = (i32)inst >> 20;
  Rec ill = mk(OP_ILLEGAL, 0, 0, 0, 0, F_SIZE4);
  Rec r = ill;
  const bool rv32 = (isa & ISA_RV32) != 0;
  if (rv32 && (op == 0x1b || op == 0x3b)) op = 0x7f;       // OP-IMM-32 / OP-32: RV64 only
  switch (op) {
    case 0x37: r = mk(OP_ADD, rd, 0, 0, (i32)(inst & 0xfffff000u), F_IMM); break;
    case 0x17: r = mk(OP_AUIPC, rd, 0, 0, (i32)(inst & 0xfffff000u)); break;
    case 0x6f: {
      i32 off = ((i32)(inst & 0x80000000u) >> 11) | (i32)(inst & 0xff000) | (i32)((inst >> 9) & 0x800) | (i32)((inst >> 20) & 0x7fe);
      r = mk(OP_JAL, rd, 0, 0, off); break; }
    case 0x67: if (f3 == 0) r = mk(OP_JALR, rd, rs1, 0, immi); break;
    case 0x63: {
      i32 off = ((i32)(inst & 0x80000000u) >> 19) | (i32)((inst & 0x80) << 4) | (i32)((inst >> 20) & 0x7e0) | (i32)((inst >> 7) & 0x1e);
      u32 o = f3 == 0 ? OP_BEQ : f3 == 1 ? OP_BNE : f3 == 4 ? OP_BLT : f3 == 5 ? OP_BGE : f3 == 6 ? OP_BLTU : f3 == 7 ? OP_BGEU : (u32)OP_ILLEGAL;
      if (o != OP_ILLEGAL) r = mk(o, 0, rs1, rs2, off);
      break; }
    case 0x03: if (f3 != 7 && !(rv32 && (f3 == 3 || f3 == 6))) r = mk(OP_LB + f3, rd, rs1, 0, immi); break;      // ld / lwu: RV64 only
    case 0x23: if (f3 <= 3 && !(rv32 && f3 == 3)) r = mk(OP_SB + f3, 0, rs1, rs2, ((i32)(inst & 0xfe000000u) >> 20) | (i32)((inst >> 7) & 31)); break;
    case 0x13: {
      u32 sh = (inst >> 20) & 63, top6 = inst >> 26;
      if (rv32 && sh >= 32 && (f3 == 1 || f3 == 5)) break;                 // shamt[5] must be zero on RV32 (Hart.cpp:11736-11746)
      switch (f3) {
        case 0: r = mk(OP_ADD, rd, rs1, 0, immi, F_IMM); break;
        case 1: if (top6 == 0) r = mk(OP_SLL, rd, rs1, 0, (i32)sh, F_IMM); break;
        case 2: r = mk(OP_SLT, rd, rs1, 0, immi, F_IMM); break;
        case 3: r = mk(OP_SLTU, rd, rs1, 0, immi, F_IMM); break;
        case 4: r = mk(OP_XOR, rd, rs1, 0, immi, F_IMM); break;
        case 5: if (top6 == 0) r = mk(OP_SRL, rd, rs1, 0, (i32)sh, F_IMM); else if (top6 == 0x10) r = mk(OP_SRA, rd, rs1, 0, (i32)sh, F_IMM); break;
        case 6: r = mk(OP_OR, rd, rs1, 0, immi, F_IMM); break;
        default: r = mk(OP_AND, rd, rs1, 0, immi, F_IMM); break;
      }
      break; }
    case 0x1b:
      if (f3 == 0) r = mk(OP_ADDW, rd, rs1, 0, immi, F_IMM);
      else if (f3 == 1 && f7 == 0) r = mk(OP_SLLW, rd, rs1, 0, (i32)rs2, F_IMM);
      else if (f3 == 5 && f7 == 0) r = mk(OP_SRLW, rd, rs1, 0, (i32)rs2, F_IMM);
      else if (f3 == 5 && f7 == 0x20) r = mk(OP_SRAW, rd, rs1, 0, (i32)rs2, F_IMM);
      break;
    case 0x33:
      if (f7 == 1) { if (isa & ISA_M) r = mk(OP_MUL + f3, rd, rs1, rs2, 0); }
      else if (f7 == 0) { const u32 t[8] = { OP_ADD, OP_SLL, OP_SLT, OP_SLTU, OP_XOR, OP_SRL, OP_OR, OP_AND }; r = mk(t[f3], rd, rs1, rs2, 0); }
      else if (f7 == 0x20) { if (f3 == 0) r = mk(OP_SUB, rd, rs1, rs2, 0); else if (f3 == 5) r = mk(OP_SRA, rd, rs1, rs2, 0); }
      break;
    case 0x3b:
      if (f7 == 1) {
        if (isa & ISA_M) { u32 o = f3 == 0 ? OP_MULW : f3 == 4 ? OP_DIVW : f3 == 5 ? OP_DIVUW : f3 == 6 ? OP_REMW : f3 == 7 ? OP_REMUW : (u32)OP_ILLEGAL; if (o) r = mk(o, rd, rs1, rs2, 0); }
      } else if (f7 == 0) { if (f3 == 0) r = mk(OP_ADDW, rd, rs1, rs2, 0); else if (f3 == 1) r = mk(OP_SLLW, rd, rs1, rs2, 0); else if (f3 == 5) r = mk(OP_SRLW, rd, rs1, rs2, 0); }
      else if (f7 == 0x20) { if (f3 == 0) r = mk(OP_SUBW, rd, rs1, rs2, 0); else if (f3 == 5) r = mk(OP_SRAW, rd, rs1, rs2, 0); }
      break;
    case 0x0f: if (f3 == 0 || f3 == 1) r = mk(OP_NOP, 0, 0, 0, 0); break;
    case 0x73:
      if (f3 == 0) {
        if (rd != 0) break;
        if (inst == 0x00000073) r = mk(OP_ECALL, 0, 0, 0, 0);
        else if (inst == 0x00100073) r = mk(OP_EBREAK, 0, 0, 0, 0);
        else if (inst == 0x30200073) r = mk(OP_MRET, 0, 0, 0, 0);
        else if (inst == 0x10200073) { if (isa & ISA_S) r = mk(OP_SRET, 0, 0, 0, 0); }
        else if (inst == 0x10500073) r = mk(OP_WFI, 0, 0, 0, 0);
        else if (f7 == 0x09) { if (isa & ISA_S) r = mk(OP_SFENCE_VMA, 0, rs1, rs2, 0); }
      } else if (f3 != 4) {
        // csr number in imm; for the immediate forms the 5-bit uimm travels in the rs1 field
        r = mk(OP_CSRRW + ((f3 & 3) - 1), rd, rs1, 0, (i32)(inst >> 20), (f3 & 4) ? F_IMM : 0);
      }
      break;
    case 0x2f: {
      u32 f5 = inst >> 27;
      if (!(isa & ISA_A) || (f3 != 2 && f3 != 3) || (rv32 && f3 == 3)) break;      // .d forms: RV64 only (amo.cpp:886, 918, 964)
      u32 fl = f3 == 3 ? F_DBL : 0;
      if (f5 == 2) { if (rs2 == 0) r = mk(OP_LR, rd, rs1, 0, 0, fl); }
      else if (f5 == 3) r = mk(OP_SC, rd, rs1, rs2, 0, fl);
      else if (f5 == 0 || f5 == 1 || f5 == 4 || f5 == 8 || f5 == 12 || f5 == 16 || f5 == 20 || f5 == 24 || f5 == 28) r = mk(OP_AMO, rd, rs1, rs2, (i32)f5, fl);
      break; }
    case 0x07:
      if ((f3 == 0 || f3 >= 5) && (isa & ISA_V)) { r = mk(OP_VLD, rd, rs1, rs2, 0); break; }    // vector widths (Decoder.cpp decodeVecLoad)
      if (f3 == 2 && (isa & ISA_F)) r = mk(OP_FLW, rd, rs1, 0, immi);
      else if (f3 == 3 && (isa & ISA_D)) r = mk(OP_FLD, rd, rs1, 0, immi);
      break;
    case 0x27: {
      i32 imms = ((i32)(inst & 0xfe000000u) >> 20) | (i32)((inst >> 7) & 31);
      if ((f3 == 0 || f3 >= 5) && (isa & ISA_V)) { r = mk(OP_VST, 0, rs1, rs2, 0); break; }
      if (f3 == 2 && (isa & ISA_F)) r = mk(OP_FSW, 0, rs1, rs2, imms);
      else if (f3 == 3 && (isa & ISA_D)) r = mk(OP_FSD, 0, rs1, rs2, imms);
      break; }
    case 0x43: case 0x47: case 0x4b: case 0x4f: {
      u32 fmt = (inst >> 25) & 3, k = (op - 0x43) >> 2;
      if (fmt == 0 && (isa & ISA_F)) r = mk(OP_FMA, rd, rs1, rs2, (i32)(f3 | (k << 3)), 0, inst >> 27);
      else if (fmt == 1 && (isa & ISA_D)) r = mk(OP_FMA, rd, rs1, rs2, (i32)(f3 | (k << 3)), F_DBL, inst >> 27);
      break; }
    case 0x57: if (isa & ISA_V) r = mk(OP_VEC, 0, rs1, rs2, 0); break;       // rd stays out of the record: x[rd] is written explicitly
    case 0x53: {
      if (f7 & 2) break;
      u32 dbl = (f7 & 1) ? (u32)F_DBL : 0;
      if (dbl ? !(isa & ISA_D) : !(isa & ISA_F)) break;
      switch (f7 >> 2) {
        case 0: r = mk(OP_FADD, rd, rs1, rs2, (i32)f3, dbl); break;
        case 1: r = mk(OP_FSUB, rd, rs1, rs2, (i32)f3, dbl); break;
        case 2: r = mk(OP_FMUL, rd, rs1, rs2, (i32)f3, dbl); break;
        case 3: r = mk(OP_FDIV, rd, rs1, rs2, (i32)f3, dbl); break;
        case 0x0b: if (rs2 == 0) r = mk(OP_FSQRT, rd, rs1, 0, (i32)f3, dbl); break;
        case 4: if (f3 <= 2) r = mk(OP_FSGNJ, rd, rs1, rs2, (i32)f3, dbl); break;
        case 5: if (f3 <= 1) r = mk(OP_FMINMAX, rd, rs1, rs2, (i32)f3, dbl); break;
        case 8:
          if (!(isa & ISA_D)) break;
          if (!dbl && rs2 == 1) r = mk(OP_FCVT_FF, rd, rs1, 0, (i32)f3, 0);          // fcvt.s.d
          else if (dbl && rs2 == 0) r = mk(OP_FCVT_FF, rd, rs1, 0, (i32)f3, F_DBL);   // fcvt.d.s
          break;
        case 0x14: if (f3 <= 2) r = mk(OP_FCMP, rd, rs1, rs2, (i32)f3, dbl); break;
        case 0x18: if (rs2 <= (rv32 ? 1u : 3u)) r = mk(OP_FCVT_F2I, rd, rs1, rs2, (i32)f3, dbl); break;             // l / lu forms: RV64 only (float.cpp:900-996)
        case 0x1a: if (rs2 <= (rv32 ? 1u : 3u)) r = mk(OP_FCVT_I2F, rd, rs1, rs2, (i32)f3, dbl); break;
        case 0x1c: if (rs2 == 0) { if (f3 == 0) { if (!(rv32 && dbl)) r = mk(OP_FMV_F2X, rd, rs1, 0, 0, dbl); } else if (f3 == 1) r = mk(OP_FCLASS, rd, rs1, 0, 0, dbl); } break;   // fmv.x.d: RV64 only (float.cpp:1705)
        case 0x1e: if (rs2 == 0 && f3 == 0 && !(rv32 && dbl)) r = mk(OP_FMV_X2F, rd, rs1, 0, 0, dbl); break;
        default: break;
      }
      break; }
    default: break;
  }
  if ((isa & ISA_ZB_ANY) && rec_op(r) == OP_ILLEGAL && (op == 0x13 || op == 0x1b || op == 0x33 || op == 0x3b)) r = decode_zb(inst, isa);
  if (rv32) r = lower_rv32(r);
  r.w0 |= F_SIZE4;
  r.raw = inst;
  return r;
}

// 16-bit encodings (Decoder.cpp:1455-1819), RV64: produces the record of the equivalent base
// instruction (the reference executes c.* through the same exec bodies, Hart.cpp:8471-8515).
WSP_HD Rec decode16(u32 c, u32 isa)
{
  c &= 0xffff;
  u32 q = c & 3, f3 = (c >> 13) & 7;
  u32 rdp = 8 + ((c >> 2) & 7), rs1p = 8 + ((c >> 7) & 7);
  u32 rd = (c >> 7) & 31, rs2 = (c >> 2) & 31, b12 = (c >> 12) & 1;
  i32 imm6 = sx((b12 << 5) | ((c >> 2) & 31), 6);
  Rec r = mk(OP_ILLEGAL, 0, 0, 0, 0);
  bool fd = (isa & ISA_D) != 0;                 // Zcd = C + D (Hart.cpp:800-802)
  const bool rv32 = (isa & ISA_RV32) != 0, ff = (isa & ISA_F) != 0;              // Zcf = C + F on RV32
  if (!(isa & ISA_C)) { r.raw = c; return r; }
  if (q == 0) {
    u32 ud = (((c >> 10) & 7) << 3) | (((c >> 5) & 3) << 6);
    u32 uw = (((c >> 10) & 7) << 3) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 6);
    switch (f3) {
      case 0: {
        u32 imm = (((c >> 11) & 3) << 4) | (((c >> 7) & 0xf) << 6) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 3);
        if (c != 0 && imm != 0) r = mk(OP_ADD, rdp, 2, 0, (i32)imm, F_IMM);
        break; }
      case 1: if (fd) r = mk(OP_FLD, rdp, rs1p, 0, (i32)ud); break;
      case 2: r = mk(OP_LW, rdp, rs1p, 0, (i32)uw); break;
      case 3: if (rv32) { if (ff) r = mk(OP_FLW, rdp, rs1p, 0, (i32)uw); }  // c.flw (Decoder.cpp:1490-1503)
              else r = mk(OP_LD, rdp, rs1p, 0, (i32)ud);
              break;
      case 5: if (fd) r = mk(OP_FSD, 0, rs1p, rdp, (i32)ud); break;
      case 6: r = mk(OP_SW, 0, rs1p, rdp, (i32)uw); break;
      case 7: if (rv32) { if (ff) r = mk(OP_FSW, 0, rs1p, rdp, (i32)uw); }  // c.fsw (Decoder.cpp:1548-1556)
              else r = mk(OP_SD, 0, rs1p, rdp, (i32)ud);
              break;
      default: break;
    }
  } else if (q == 1) {
    switch (f3) {
      case 0: r = mk(OP_ADD, rd, rd, 0, imm6, F_IMM); break;
      case 1:
        if (rv32) {                                                        // c.jal (Decoder.cpp:1573-1588)
          i32 off = sx((b12 << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) |
                       (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
          r = mk(OP_JAL, 1, 0, 0, off);
        } else if (rd != 0) r = mk(OP_ADDW, rd, rd, 0, imm6, F_IMM);
        break;
      case 2: r = mk(OP_ADD, rd, 0, 0, imm6, F_IMM); break;
      case 3:
        if (imm6 == 0) break;
        if (rd == 2) {
          i32 imm = sx((b12 << 9) | (((c >> 6) & 1) << 4) | (((c >> 5) & 1) << 6) | (((c >> 3) & 3) << 7) | (((c >> 2) & 1) << 5), 10);
          r = mk(OP_ADD, 2, 2, 0, imm, F_IMM);
        } else r = mk(OP_ADD, rd, 0, 0, (i32)((u32)imm6 << 12), F_IMM);
        break;
      case 4: {
        u32 f2 = (c >> 10) & 3, sh = (b12 << 5) | ((c >> 2) & 31), k = (c >> 5) & 3;
        if (rv32 && b12 && f2 != 2) break;                                 // shamt[5] / c.subw / c.addw: RV64 only (Decoder.cpp:1628, 1635, 1686)
        if (f2 == 0) r = mk(OP_SRL, rs1p, rs1p, 0, (i32)sh, F_IMM);
        else if (f2 == 1) r = mk(OP_SRA, rs1p, rs1p, 0, (i32)sh, F_IMM);
        else if (f2 == 2) r = mk(OP_AND, rs1p, rs1p, 0, imm6, F_IMM);
        else if (!b12) { const u32 t[4] = { OP_SUB, OP_XOR, OP_OR, OP_AND }; r = mk(t[k], rs1p, rs1p, rdp, 0); }
        else if (k == 0) r = mk(OP_SUBW, rs1p, rs1p, rdp, 0);
        else if (k == 1) r = mk(OP_ADDW, rs1p, rs1p, rdp, 0);
        break; }
      case 5: {
        i32 off = sx((b12 << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) |
                     (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
        r = mk(OP_JAL, 0, 0, 0, off); break; }
      default: {
        i32 off = sx((b12 << 8) | (((c >> 10) & 3) << 3) | (((c >> 5) & 3) << 6) | (((c >> 3) & 3) << 1) | (((c >> 2) & 1) << 5), 9);
        r = mk(f3 == 6 ? OP_BEQ : OP_BNE, 0, rs1p, 0, off); break; }
    }
  } else if (q == 2) {
    switch (f3) {
      case 0: if (!(rv32 && b12)) r = mk(OP_SLL, rd, rd, 0, (i32)((b12 << 5) | rs2), F_IMM); break;                  // Decoder.cpp:1719
      case 1: if (fd) r = mk(OP_FLD, rd, 2, 0, (i32)((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6))); break;
      case 2: if (rd != 0) r = mk(OP_LW, rd, 2, 0, (i32)((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6))); break;
      case 3: if (rv32) { if (ff) r = mk(OP_FLW, rd, 2, 0, (i32)((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6))); }   // c.flwsp (Decoder.cpp:1746-1757)
              else if (rd != 0) r = mk(OP_LD, rd, 2, 0, (i32)((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6)));
              break;
      case 4:
        if (!b12) { if (rs2 == 0) { if (rd != 0) r = mk(OP_JALR, 0, rd, 0, 0); } else r = mk(OP_ADD, rd, 0, rs2, 0); }
        else if (rs2 == 0) { if (rd == 0) r = mk(OP_EBREAK, 0, 0, 0, 0); else r = mk(OP_JALR, 1, rd, 0, 0); }
        else r = mk(OP_ADD, rd, rd, rs2, 0);
        break;
      case 5: if (fd) r = mk(OP_FSD, 0, 2, rs2, (i32)((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6))); break;
      case 6: r = mk(OP_SW, 0, 2, rs2, (i32)((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6))); break;
      default: if (rv32) { if (ff) r = mk(OP_FSW, 0, 2, rs2, (i32)((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6))); }             // c.fswsp (Decoder.cpp:1804-1815)
               else r = mk(OP_SD, 0, 2, rs2, (i32)((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6)));
               break;
    }
  }
  if (rv32) r = lower_rv32(r);
  r.raw = c;
  return r;
}

WSP_HD Rec decode(u32 inst, u32 isa) { return (inst & 3) == 3 ? decode32(inst, isa) : decode16(inst, isa); }

// ---------------------------------------------------------------------------------- state
// Cold CSR slots (SoA rows in HBM: cold[slot * stride + hart_slot]).
enum ColdIx {
  CI_MEDELEG, CI_MIDELEG, CI_MIE, CI_MIP, CI_MTVEC, CI_MCOUNTEREN, CI_MENVCFG, CI_MCOUNTINHIBIT,
  CI_MSCRATCH, CI_MEPC, CI_MCAUSE, CI_MTVAL, CI_STVEC, CI_SCOUNTEREN, CI_SENVCFG, CI_SSCRATCH,
  CI_SEPC, CI_SCAUSE, CI_STVAL, CI_SATP,
  CI_PMPCFG0,                        // pmpcfg0,2,..,14 (RV64: even numbered only, CsRegs.cpp:4777-4784)
  CI_PMPADDR0 = CI_PMPCFG0 + 8,      // pmpaddr0..63
  CI_VSTART = CI_PMPADDR0 + 64, CI_VCSR /* vxrm << 1 | vxsat */, CI_VL, CI_VTYPE,
  CI_COUNT
};

// HF_HOT: the hart may use the pre-decoded fetch and the trap-free inline subset: M-mode, MPRV clear,
// code not self-modified, no counter inhibited (so every hot step advances exec, cycle, instret and
// retired by exactly one).  Recomputed by update_hot_flag() after anything that can change it.
enum : u32 { HF_CODE_DIRTY = 1, HF_LR_VALID = 2, HF_INH_CY = 4, HF_INH_IR = 8, HF_HOT = 16,
             HF_FS_DIRTY = 32 /* mstatus.FS == Dirty: an FP write changes no CSR but fflags (kept by update_hot_flag) */,
             HF_PMP_ON = 64 /* some PMP entry is not Off */, HF_PMP_MLOCK = 128 /* some locked entry is on: M mode is checked too */,
             HF_RING = 256 /* a commit ring is recording: every step takes the general path, whose commit() writes it */,
             HF_NOTRACE = 512 /* this run folds no commit records (set per launch): commit() returns at once */,
             HF_VS_DIRTY = 1024 /* mstatus.VS == Dirty (kept by update_hot_flag): a vector write changes no mstatus bit */,
             HF_PMP_E0ALL = 2048 /* PMP entry 0 is an active NAPOT region over the whole address space: it decides every access */ };

// Everything a hart reads or writes on (almost) every step: lives in VGPRs during a run slice.
struct Hot {
  u64 pc, mstatus, cycle, instret, exec, retired, thash;
  u64 lr_addr, exec_last_ill, stop_value;
  u32 fcsr, priv, status, consec_ill, htif_dev, flags;
};

// CSR numbers
enum {
  C_FFLAGS = 0x001, C_FRM = 0x002, C_FCSR = 0x003,
  C_VSTART = 0x008, C_VXSAT = 0x009, C_VXRM = 0x00A, C_VCSR = 0x00F, C_VL = 0xC20, C_VTYPE = 0xC21, C_VLENB = 0xC22,
  C_SSTATUS = 0x100, C_SIE = 0x104, C_STVEC = 0x105, C_SCOUNTEREN = 0x106, C_SENVCFG = 0x10A,
  C_SSCRATCH = 0x140, C_SEPC = 0x141, C_SCAUSE = 0x142, C_STVAL = 0x143, C_SIP = 0x144, C_SATP = 0x180,
  C_MSTATUS = 0x300, C_MISA = 0x301, C_MEDELEG = 0x302, C_MIDELEG = 0x303, C_MIE = 0x304, C_MTVEC = 0x305,
  C_MCOUNTEREN = 0x306, C_MENVCFG = 0x30A, C_MCOUNTINHIBIT = 0x320,
  C_MSCRATCH = 0x340, C_MEPC = 0x341, C_MCAUSE = 0x342, C_MTVAL = 0x343, C_MIP = 0x344,
  C_PMPCFG0 = 0x3A0, C_PMPADDR0 = 0x3B0,
  C_MCYCLE = 0xB00, C_MINSTRET = 0xB02, C_CYCLE = 0xC00, C_TIME = 0xC01, C_INSTRET = 0xC02,
  C_MVENDORID = 0xF11, C_MARCHID = 0xF12, C_MIMPID = 0xF13, C_MHARTID = 0xF14, C_MCONFIGPTR = 0xF15
};

static constexpr u64 MST_SIE = 1ULL << 1, MST_MIE = 1ULL << 3, MST_SPIE = 1ULL << 5, MST_MPIE = 1ULL << 7, MST_SPP = 1ULL << 8;
static constexpr u64 MST_MPRV = 1ULL << 17, MST_SUM = 1ULL << 18, MST_MXR = 1ULL << 19, MST_TVM = 1ULL << 20, MST_TW = 1ULL << 21, MST_TSR = 1ULL << 22, MST_SD = 1ULL << 63;
static constexpr int MST_VS_SH = 9, MST_MPP_SH = 11, MST_FS_SH = 13, MST_XS_SH = 15;
static constexpr u64 MSTATUS_WMASK = 0x7FFFAAULL;             // CsRegs.cpp:4704
// medeleg: M ecall (11), double trap (16), reserved (17) hardwired zero (CsRegs.cpp:4726-4732); VS ecall (10) and the
// guest page fault / virtual instruction causes (23:20) read as zero without H (CsRegs.cpp:1780-1788)
static constexpr u64 MEDELEG_MASK = ~((1ULL << 11) | (1ULL << 16) | (1ULL << 17) | (0xFULL << 20) | (1ULL << 10));
static constexpr u64 SSTATUS_WMASK = 0x000DE722ULL;           // CsRegs.cpp:4979
static constexpr u64 SSTATUS_RMASK = 0x80000003000DE762ULL;   // CsRegs.cpp:4990
static constexpr u64 MISA_RESET = 0x800000000034112dULL;      // CsRegs.cpp:4721-4724 (read-only)
static constexpr u64 MSTATUS_RESET = (3ULL << MST_MPP_SH) | (0xAULL << 32);   // CsRegs.cpp:4705-4710

// Per-step commit bookkeeping (what printDecodedInstTrace would print, printTrace.cpp:388-493).
enum : u32 {
  CH_IREG = 1, CH_FREG = 2, CH_MEM = 4, CH_GEN = 8, CH_SSTATUS = 16, CH_SEPC = 32, CH_SCAUSE = 64, CH_STVAL = 128,
  CH_MSTATUS = 256, CH_MEPC = 512, CH_MCAUSE = 1024, CH_MTVAL = 2048, CH_EXC = 4096,
  CH_VREG = 8192, CH_VL = 16384, CH_VTYPE = 32768,
  CH_BAIL = 65536,                 // Lane<MEM, DRY = true>: the instruction was not completed (and changed nothing)
  CH_LDST = 131072                 // commit ring only: a scalar load / store / atomic ran; Step::lva / lpa hold its addresses
};
struct Step {
  u32 chg, ireg, freg, gen_csr;
  u32 vreg, vgroup;              // vector group written: first register (operand 0), register count (0: none)
  u64 ival, fval, maddr, mval;
  u64 lva, lpa;                  // ldStAddr_ / ldStPhysAddr1_ of the log's "[va:pa]" suffix (printTrace.cpp:338-346); kept only while a ring records
};

// ---------------------------------------------------------------------------------- lane context
// MEM policy supplies: rx/wx/rf/wf (register files), cold(i) (CSR rows), ld8..ld64/st8..st64 at
// a byte offset into this hart's RAM arena, tlb(i) (word i of this hart's TLB, two words per entry).
//
// TLB entry (Tlb.hpp:28-46 without world id / VMID), packed:
//   word 0: vpn[51:0] | level[53:52] | counter[55:54] | valid, global, user, r, w, x, accessed, dirty [63:56]
//   word 1: ppn[43:0] | asid[59:44]
enum : u32 { TE_VALID = 1, TE_GLOBAL = 2, TE_USER = 4, TE_R = 8, TE_W = 16, TE_X = 32, TE_A = 64, TE_D = 128 };
static constexpr u64 TE_VPN_MASK = 0xfffffffffffffULL;
// DRY = true: the instantiation the kernel's vector warm path runs (wsp_vwarm): anything that would trap, stop the hart or change
// mstatus raises CH_BAIL instead and changes nothing - the general path then executes the instruction in full.  A compile-time
// switch, so that the trap / stop bodies and every hart register only they use drop out of the warm function.
template <class MEM, bool DRY = false> struct Lane {
  Hot h;
  Step s;
  MEM m;
  u64 mem_base, mem_size, tohost, text_lo, text_hi;   // tohost == ~0: none
  u64 stop_pc = ~0ULL;                                 // Hart::untilAddress' address (Hart.cpp:6025): ~0 = none (pcs are even)
  u32 isa, hart_id;
  u32 tlb_n;                                           // TLB entries (power of two; 0 when the ISA has no S)
  u32 vlenb;                                           // bytes per vector register (0 when the ISA has no V)
  u64* ring_ = nullptr; u32* ring_pos_ = nullptr; u32 ring_cap = 0;   // commit ring of this hart (null: not recording), see commit()
  // Lanes-per-element mapping of the vector unit (north_star; the element loops of vector.cpp:808-855, vector-float.cpp:667-691,
  // vector-ls.cpp:156-247).  On the GPU a hart of a V batch is stepped on the general path by a sub-group of vgrp lanes at once,
  // all holding the same hart state and executing the same instructions in lock step - every store to hart state is the same
  // value from every lane - except that the element loops marked "lanes" below are split: lane vsub takes the elements
  // vsub, vsub + vgrp, ...  What the lanes compute apart (exception flags, first faulting element) is combined across the
  // sub-group (sub_or / sub_min) before it touches hart state, so the replicas stay identical.  vgrp = 1: one lane, serial loops.
  u32 vsub = 0, vgrp = 1;
  WSP_HD u32 sub_or(u32 v) { return vgrp > 1 ? m.xlane_or(v) : v; }
  WSP_HD u32 sub_min(u32 v) { return vgrp > 1 ? m.xlane_min(v) : v; }
  WSP_HD void sub_fence() { if (vgrp > 1) m.xlane_fence(); }          // element writes of the other lanes are visible after this

  WSP_HD bool ext(u32 e) const { return (isa & e) != 0; }
  WSP_HD bool in_mem(u64 a, u64 n) const { u64 off = a - mem_base; return a >= mem_base && off <= mem_size - n; }

  // ---- integer / fp register writes with trace capture
  // RV32 (ISA_RV32): registers hold the 32-bit value sign-extended, the commit record carries it zero-extended; the pc and every
  // address wrap at 32 bits
  WSP_HD bool rv32() const { return (isa & ISA_RV32) != 0; }
  WSP_HD u64 amask() const { return rv32() ? 0xffffffffULL : ~0ULL; }
  WSP_HD void wx(u32 rd, u64 v)
  { if (rv32()) { if (rd) m.wx(rd, (u64)(i64)(i32)v); s.ival = (u32)v; } else { if (rd) m.wx(rd, v); s.ival = v; } s.chg |= CH_IREG; s.ireg = rd; }
  WSP_HD void wf(u32 rd, u64 v) { m.wf(rd, v); s.chg |= CH_FREG; s.freg = rd; s.fval = v; }
  WSP_HD void wf_s(u32 rd, u32 v) { wf(rd, ext(ISA_D) ? (0xffffffff00000000ULL | v) : (u64)v); }          // FpRegs.hpp:248-269
  WSP_HD u32 rf_s(u32 r) { u64 v = m.rf(r); return (ext(ISA_D) && (v >> 32) != 0xffffffffu) ? 0x7fc00000u : (u32)v; }  // FpRegs.hpp:285-295

  // ---- mstatus helpers
  WSP_HD u64 legalize_mstatus(u64 v) const                                                                  // CsRegs.cpp:2617-2648
  {
    u32 fs = (v >> MST_FS_SH) & 3, xs = (v >> MST_XS_SH) & 3, vs = (v >> MST_VS_SH) & 3;
    v = (fs == 3 || xs == 3 || vs == 3) ? (v | MST_SD) : (v & ~MST_SD);
    u32 mpp = (v >> MST_MPP_SH) & 3, spp = (v >> 8) & 1;
    if (!ext(ISA_S)) spp = 0;
    if (mpp == PRV_S && !ext(ISA_S)) mpp = PRV_U;
    if (mpp == 2) mpp = PRV_U;
    if (mpp == PRV_U && !ext(ISA_U)) mpp = PRV_M;
    v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)mpp << MST_MPP_SH);
    return (v & ~MST_SPP) | ((u64)spp << 8);
  }
  WSP_HD bool fp_on() const { return ((h.mstatus >> MST_FS_SH) & 3) != 0; }                                  // isFpLegal (F folded into decode)
  WSP_HD void mark_fs_dirty()                                                                               // float.cpp:165-191
  { if (((h.mstatus >> MST_FS_SH) & 3) != 3 && !dry_bail()) { h.mstatus |= (3ULL << MST_FS_SH) | MST_SD; s.chg |= CH_MSTATUS; } }
  WSP_HD void or_fflags(u32 f)                                                                              // float.cpp:140-162, Hart.hpp:3821
  {
    u32 prev = h.fcsr & 0x1f, val = prev | (f & 0x1f);
    if (val != prev) { h.fcsr = (h.fcsr & ~0x1fu) | val; s.chg |= CH_GEN; s.gen_csr = C_FCSR; }
    if (f) mark_fs_dirty();
  }

  WSP_HD bool dry_bail() { if (DRY) { s.chg |= CH_BAIL; return true; } return false; }
  WSP_HD void stop(u32 st, u64 v) { if (dry_bail()) return; h.status = st; h.stop_value = v; }

  // ---- traps: initiateException + initiateTrap (Hart.cpp:3324-3397, 3565-3929)
  WSP_HD void take_exception(u32 cause, u64 epc, u64 tval, bool counts)
  {
    if (dry_bail()) return;
    if (counts) {   // di == nullptr or illegal: the consecutive-illegal stop heuristic (3329-3340)
      if (h.exec == h.exec_last_ill + 1) h.consec_ill++; else h.consec_ill = 0;
      if (h.consec_ill > 16) { stop(ST_STOP_ILLEGAL, 3); return; }
      h.exec_last_ill = h.exec;
    }
    s.chg |= CH_EXC;
    if (ext(CFG_CANCEL_LR_TRAP)) h.flags &= ~(u32)HF_LR_VALID;                            // initiateTrap (Hart.cpp:3570-3571)
    u32 next = PRV_M;
    if (ext(ISA_S) && h.priv != PRV_M && ((m.cold(CI_MEDELEG) >> cause) & 1)) next = PRV_S;
    u32 orig = h.priv;
    h.priv = next;
    u64 tvec, v = h.mstatus;
    if (next == PRV_M) {
      m.cold(CI_MEPC) = epc & ~1ULL; m.cold(CI_MCAUSE) = cause; m.cold(CI_MTVAL) = tval;
      v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)orig << MST_MPP_SH);
      v = (v & ~MST_MPIE) | ((v & MST_MIE) ? MST_MPIE : 0);
      v &= ~MST_MIE;
      s.chg |= CH_MSTATUS | CH_MEPC | CH_MCAUSE | CH_MTVAL;
      tvec = m.cold(CI_MTVEC);
    } else {
      m.cold(CI_SEPC) = epc & ~1ULL; m.cold(CI_SCAUSE) = cause; m.cold(CI_STVAL) = tval;
      v = (v & ~MST_SPP) | ((u64)(orig & 1) << 8);
      v = (v & ~MST_SPIE) | ((v & MST_SIE) ? MST_SPIE : 0);
      v &= ~MST_SIE;
      s.chg |= CH_SSTATUS | CH_SEPC | CH_SCAUSE | CH_STVAL;
      tvec = m.cold(CI_STVEC);
    }
    h.mstatus = v;
    h.pc = tvec & ~3ULL;
  }
  WSP_HD void illegal(const Rec& r, u64 cur_pc) { take_exception(2, cur_pc, ext(CFG_CLR_MTVAL_ILL) ? 0 : r.raw, true); }   // illegalInst (Hart.cpp:3247-3258)

  // ---- interrupts: Hart::isInterruptPossible (Hart.cpp:7028-7083, 7135-7157) without H/AIA/NMI.  A bare system has no
  // timer or external source, so pending = mip as software left it; M-destined = mip & mie & ~mideleg, S-destined =
  // mip & mie & mideleg.  Priority order = mInterrupts_ / sInterrupts_ (Hart.cpp:141-149).
  WSP_HD bool interrupt_possible(u32& cause, u32& next)
  {
    const u64 pend = m.cold(CI_MIP) & m.cold(CI_MIE);
    if (!pend) return false;
    const u64 deleg = m.cold(CI_MIDELEG);
    const u64 mdest = pend & ~deleg;
    if (((h.mstatus & MST_MIE) || h.priv != PRV_M) && mdest) {
      const u8 order[8] = { 11, 3, 7, 9, 1, 5, 12, 13 };
      for (u32 i = 0; i < 8; ++i) if ((mdest >> order[i]) & 1) { cause = order[i]; next = PRV_M; return true; }
    }
    if (h.priv == PRV_M) return false;
    const u64 sdest = pend & deleg;
    if (((h.mstatus & MST_SIE) || h.priv == PRV_U) && sdest) {
      const u8 order[11] = { 11, 3, 7, 9, 1, 5, 12, 10, 2, 6, 13 };
      for (u32 i = 0; i < 11; ++i) if ((sdest >> order[i]) & 1) { cause = order[i]; next = PRV_S; return true; }
    }
    return false;
  }
  // Hart::initiateInterrupt -> initiateTrap(interrupt = true) (Hart.cpp:3287-3300, 3565-3905)
  WSP_HD void take_interrupt(u32 cause, u32 next, u64 epc)
  {
    const u32 orig = h.priv;
    if (ext(CFG_CANCEL_LR_TRAP)) h.flags &= ~(u32)HF_LR_VALID;
    h.priv = next;
    const u64 cv = (1ULL << 63) | cause;
    u64 tvec, v = h.mstatus;
    if (next == PRV_M) {
      m.cold(CI_MEPC) = epc & ~1ULL; m.cold(CI_MCAUSE) = cv; m.cold(CI_MTVAL) = 0;
      v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)orig << MST_MPP_SH);
      v = (v & ~MST_MPIE) | ((v & MST_MIE) ? MST_MPIE : 0);
      v &= ~MST_MIE;
      s.chg |= CH_MSTATUS | CH_MEPC | CH_MCAUSE | CH_MTVAL;
      tvec = m.cold(CI_MTVEC);
    } else {
      m.cold(CI_SEPC) = epc & ~1ULL; m.cold(CI_SCAUSE) = cv; m.cold(CI_STVAL) = 0;
      v = (v & ~MST_SPP) | ((u64)(orig & 1) << 8);
      v = (v & ~MST_SPIE) | ((v & MST_SIE) ? MST_SPIE : 0);
      v &= ~MST_SIE;
      s.chg |= CH_SSTATUS | CH_SEPC | CH_SCAUSE | CH_STVAL;
      tvec = m.cold(CI_STVEC);
    }
    h.mstatus = v;
    const u64 base = tvec & ~3ULL;
    h.pc = (tvec & 3) == 1 ? base + 4ULL * cause : base;     // vectored mode: interrupts only (Hart.cpp:3849-3853)
  }
  // processExternalInterrupt (Hart.cpp:7216-7262): returns true when the step was consumed by an interrupt
  template <bool TRACE> WSP_HD bool poll_interrupt(u64 cur_pc, u32 priv0)
  {
    u32 cause, next;
    if (!interrupt_possible(cause, next)) return false;
    if (ext(ISA_S) && h.priv != PRV_M && (m.cold(CI_SATP) >> 60) > 8) { stop(ST_STOP_UNSUPPORTED, UNS_PAGING); return true; }
    // the interrupted instruction is read for the log only: translation without TLB / A-D side effects, PMA check, no PMP,
    // no trap (Hart::readInst, Hart.cpp:2997-3037)
    u32 iw = 0;
    {
      const bool tr = paging_on(h.priv);
      u64 pa = cur_pc;
      if ((!tr || translate_noupdate(cur_pc, h.priv, 4, pa) == 0) && in_mem(pa, 2)) {
        iw = m.ld16(pa - mem_base);
        if ((iw & 3) == 3) {
          u64 pa2 = pa + 2;
          if (tr && (cur_pc >> 12) != ((cur_pc + 2) >> 12) && translate_noupdate(cur_pc + 2, h.priv, 4, pa2) != 0) iw = 0;
          else if (in_mem(pa2, 2)) iw |= (u32)m.ld16(pa2 - mem_base) << 16;
        }
      }
    }
    take_interrupt(cause, next, cur_pc);
    if (!(h.flags & HF_INH_CY)) h.cycle++;                   // Hart.cpp:7255-7256
    commit<TRACE>(cur_pc, iw, (iw & 3) == 3 ? 4 : 2, priv0, 2);
    return true;
  }

  // ---- data access checks (determineLoad/StoreException, Hart.cpp:1988-2154, 13786-13935)
  WSP_HD u32 eff_priv() const { return ((h.mstatus & MST_MPRV) && h.priv == PRV_M) ? (u32)((h.mstatus >> MST_MPP_SH) & 3) : h.priv; }   // Hart.hpp:3444
  // ---- PMP (PmpManager.hpp, Hart::updateMemoryProtection Hart.cpp:842-899), grain G = 0, no Smepmp
  WSP_HD u32 pmp_cfg_byte(u32 ix) { return (u32)(m.cold(CI_PMPCFG0 + (ix >> 3)) >> (8 * (ix & 7))) & 0xff; }
  // PmpManager::legalizePmpcfg (PmpManager.hpp:480-551): a locked byte keeps its value, the reserved w=1 r=0 keeps the old xwr
  WSP_HD static u64 legalize_pmpcfg(u64 prev, u64 next)
  {
    u64 legal = 0;
    for (u32 i = 0; i < 8; ++i) {
      u32 pb = (u32)(prev >> (8 * i)) & 0xff, nb = (u32)(next >> (8 * i)) & 0xff;
      if (pb >> 7) nb = pb;
      else if ((nb & 3) == 2) nb = (pb & 7) | (nb & ~7u);
      legal |= (u64)nb << (8 * i);
    }
    return legal;
  }
  // CsRegs::isPmpaddrLocked (CsRegs.cpp:6773-6793): own entry locked, or the next one is a locked TOR
  WSP_HD bool pmpaddr_locked(u32 ix)
  {
    if (pmp_cfg_byte(ix) & 0x80) return true;
    if (ix >= 63) return false;
    u32 nb = pmp_cfg_byte(ix + 1);
    return (nb & 0x80) && ((nb >> 3) & 3) == 1;
  }
  WSP_HD void update_pmp_flags()
  {
    u32 f = 0;
    for (u32 k = 0; k < 8; ++k) {
      u64 w = m.cold(CI_PMPCFG0 + k), on = (w | (w >> 1)) & 0x0808080808080808ULL;      // A != Off, per byte at bit 3
      if (on) f |= HF_PMP_ON;
      if ((on << 4) & w) f |= HF_PMP_MLOCK;                                               // ... and L (bit 7)
    }
    // the usual "open everything" setup (pmpaddr0 all ones, NAPOT): entry 0 matches every address and is first in priority
    if ((((u32)m.cold(CI_PMPCFG0) >> 3) & 3) == 3 && (m.cold(CI_PMPADDR0) & 0x003fffffffffffffULL) == 0x003fffffffffffffULL) f |= HF_PMP_E0ALL;
    h.flags = (h.flags & ~(u32)(HF_PMP_ON | HF_PMP_MLOCK | HF_PMP_E0ALL)) | f;
  }
  // ---- The address-translation unit (PMP match, Sv39 walk, TLB) reads the hart but never changes it - its side effects are
  // TLB words in HBM - so it works on a small by-value context `Vm` instead of the whole Lane.  On the GPU its large, rarely
  // taken parts (TLB miss + page walk, sfence.vma, translation without update) are out-of-line functions that receive the context
  // by value: the Lane itself is then never passed by address anywhere, and the general path keeps it in registers.
  struct Vm {
    MEM m; u64 mstatus, mem_base, mem_size; u32 flags, tlb_n, isa;
    WSP_HD bool ext(u32 e) const { return (isa & e) != 0; }
    WSP_HD bool in_mem(u64 a, u64 n) const { u64 off = a - mem_base; return a >= mem_base && off <= mem_size - n; }
    // PmpManager::getPmp / accessPmp (PmpManager.hpp:155-177) over regions unpacked as unpackMemoryProtection (:417-468) and
    // word-aligned as defineRegion (:233-236): r=1 w=2 x=4.  First match in index order; M mode ignores entries that are not
    // locked; no match: M everything, S/U nothing.
    WSP_HD u32 pmp_perm(u32 pm, u64 addr)
    {
      const u32 dflt = pm == PRV_M ? 7u : 0u;
      if (!(flags & HF_PMP_ON) || (pm == PRV_M && !(flags & HF_PMP_MLOCK))) return dflt;
      if (flags & HF_PMP_E0ALL) { const u32 cfg = (u32)m.cold(CI_PMPCFG0) & 0xff; return (pm == PRV_M && !(cfg & 0x80)) ? 7u : (cfg & 7u); }
      for (u32 k = 0; k < 8; ++k) {
        const u64 w = m.cold(CI_PMPCFG0 + k);
        if (!(w & 0x1818181818181818ULL)) continue;
        for (u32 j = 0; j < 8; ++j) {
          const u32 cfg = (u32)(w >> (8 * j)) & 0xff, type = (cfg >> 3) & 3, ix = 8 * k + j;
          if (!type) continue;
          const u64 v = m.cold(CI_PMPADDR0 + ix);
          u64 lo, hi;
          if (type == 1) {                                      // TOR
            lo = (ix ? m.cold(CI_PMPADDR0 + ix - 1) : 0) << 2;
            hi = v << 2; if (hi > 0) hi -= 1;
          } else {
            u64 size_m1 = 3, napot = v;
            if (type == 3) {                                    // NAPOT (pmpaddr is 54 bits: never all ones)
              u32 rzi = 0; while (rzi < 64 && ((v >> rzi) & 1)) ++rzi;
              napot = rzi >= 64 ? 0 : ((v >> rzi) << rzi);
              size_m1 = rzi + 3 >= 64 ? ~0ULL : ((1ULL << (rzi + 3)) - 1);
            }
            lo = napot << 2; hi = lo + size_m1;
          }
          const u64 a0 = (lo >> 2) << 2, a1 = ((hi >> 2) << 2) + 3;
          if (addr >= a0 && addr <= a1) return (pm == PRV_M && !(cfg & 0x80)) ? 7u : (cfg & 7u);
        }
      }
      return dflt;
    }
    // ---- Sv39 + TLB (virtual_memory/VirtMem.cpp, Tlb.hpp, Tlb.cpp): single stage, no Svadu (an access that would need to
    // set A or D page-faults, VirtMem.hpp:1159), no Svpbmt / Svnapot / shadow stack.  The reference's PTE cache is coherent
    // with stores and flushed on sfence / satp / PMP changes, so it is not modelled.
    WSP_HD bool paging_on(u32 pm) { return ext(ISA_S) && pm != PRV_M && (m.cold(CI_SATP) >> 60) == 8; }
    WSP_HD static u32 pf_cause(u32 acc) { return acc == 4 ? 12u : acc == 1 ? 13u : 15u; }      // stage1PageFaultType (VirtMem.cpp:31-40)
    WSP_HD static u32 af_cause(u32 acc) { return acc == 4 ? 1u : acc == 1 ? 5u : 7u; }         // accessFaultType (VirtMem.cpp:57-66)
    // VirtMem::pageTableWalk<Pte39, Va39> (VirtMem.cpp:694-892) behind the canonical-address check (:393-397).
    // acc: 1 read 2 write 4 exec.  On success pa and the new TLB entry words.
    WSP_HD u32 page_walk(u64 va, u32 pm, u32 acc, bool ad_check, u64& pa, u64& e0, u64& e1)
    {
      const u32 pf = pf_cause(acc);
      if ((u64)(((i64)va << 25) >> 25) != va) return pf;
      const bool sum = (mstatus & MST_SUM) != 0, mxr = (mstatus & MST_MXR) != 0;
      const u64 satp = m.cold(CI_SATP);
      u64 root = (satp & 0xfffffffffffULL) << 12, pte = 0;
      int ii = 2; u32 global = 0;
      for (;;) {
        const u64 pte_addr = root + ((va >> (12 + 9 * ii)) & 0x1ff) * 8;
        // isAddrReadable: PMP as Supervisor (Hart.cpp:424-441); memRead: PMA read (Hart.cpp:337-339)
        if (!(pmp_perm(PRV_S, pte_addr) & 1) || !in_mem(pte_addr, 8)) return af_cause(acc);
        pte = m.ld64(pte_addr - mem_base);
        if (pte >> 63) return pf;                                                   // N without Svnapot (VirtMem.hpp:1090)
        if (!(pte & 1) || (!(pte & 2) && (pte & 4)) || ((pte >> 54) & 0x7f)) return pf;   // isValidPte (VirtMem.hpp:558)
        global |= (u32)(pte >> 5) & 1;
        if (!(pte & 0xe)) {                                                         // next level (:747-756)
          if ((pte & 0xd0) || ((pte >> 61) & 3)) return pf;
          if (--ii < 0) return pf;
          root = ((pte >> 10) & 0xfffffffffffULL) << 12;
          continue;
        }
        if ((pte >> 61) & 3) return pf;                                             // Svpbmt off (:772-773)
        const bool u = (pte >> 4) & 1, r = (pte >> 1) & 1, w = (pte >> 2) & 1, x = (pte >> 3) & 1;
        if (pm == PRV_U && !u) return pf;
        if (pm == PRV_S && u && (!sum || acc == 4)) return pf;
        if ((acc == 1 && !(r || (mxr && x))) || (acc == 2 && !w) || (acc == 4 && !x)) return pf;
        for (int j = 0; j < ii; ++j) if ((pte >> (10 + 9 * j)) & 0x1ff) return pf;  // misaligned superpage (:804-806)
        if (ad_check && (!((pte >> 6) & 1) || (acc == 2 && !((pte >> 7) & 1)))) return pf;   // faultOnFirstAccess_ (:809-814)
        break;
      }
      u64 a = va & 0xfff;                                                           // computePhysAddr (VirtMem.cpp:665-689)
      for (int j = 0; j < ii; ++j) a |= ((va >> (12 + 9 * j)) & 0x1ff) << (12 + 9 * j);
      for (int j = ii; j < 3; ++j) a |= ((pte >> (10 + 9 * j)) & (j == 2 ? 0x3ffffffULL : 0x1ffULL)) << (12 + 9 * j);
      pa = a;
      const u64 fl = TE_VALID | (global ? TE_GLOBAL : 0) | ((pte & 16) ? TE_USER : 0) | ((pte & 2) ? TE_R : 0) | ((pte & 4) ? TE_W : 0) |
                     ((pte & 8) ? TE_X : 0) | ((pte & 64) ? TE_A : 0) | ((pte & 128) ? TE_D : 0);
      e0 = (va >> 12) | ((u64)(1 + ii) << 52) | (fl << 56);
      e1 = (a >> 12) | (((satp >> 44) & 0xffff) << 44);
      return 0;
    }
    // Tlb::findEntry (Tlb.hpp:63-71) on the packed words
    WSP_HD static bool tlb_match(u64 e0, u64 e1, u64 vpn, u64 satp)
    { return ((e0 >> 56) & TE_VALID) && (e0 & TE_VPN_MASK) == vpn && (((e0 >> 56) & TE_GLOBAL) || ((e1 >> 44) & 0xffff) == ((satp >> 44) & 0xffff)); }
    // permission part of a TLB hit (VirtMem.cpp:283-310)
    WSP_HD bool tlb_hit_ok(u32 fl, u32 pm, u32 acc) const
    {
      const bool sum = (mstatus & MST_SUM) != 0, mxr = (mstatus & MST_MXR) != 0;
      if (pm == PRV_U && !(fl & TE_USER)) return false;
      if (pm == PRV_S && (fl & TE_USER) && (acc == 4 || !sum)) return false;
      if ((acc == 1 && !((fl & TE_R) || (mxr && (fl & TE_X)))) || (acc == 2 && !(fl & TE_W)) || (acc == 4 && !(fl & TE_X))) return false;
      return true;
    }
    // VirtMem::translate (VirtMem.cpp:261-329): 0 or the exception cause
    WSP_HD u32 translate_miss(u64 va, u32 pm, u32 acc, u64& pa)
    {
      pa = va;
      const u64 vpn = va >> 12, satp = m.cold(CI_SATP);
      const u32 ix = 2 * (u32)(vpn & (tlb_n - 1));
      u64 e0 = m.tlb(ix), e1 = m.tlb(ix + 1);
      if (tlb_match(e0, e1, vpn, satp)) {
        e0 = (e0 & ~(3ULL << 54)) | ((((e0 >> 54) + 1) & 3) << 54);                 // findEntryUpdateTime (Tlb.hpp:91-100)
        const u32 fl = (u32)(e0 >> 56);
        if (!tlb_hit_ok(fl, pm, acc)) { m.tlb(ix) = e0; return pf_cause(acc); }
        if ((fl & TE_A) && (acc != 2 || (fl & TE_D))) { m.tlb(ix) = e0; pa = ((e1 & 0xfffffffffffULL) << 12) | (va & 0xfff); return 0; }
        e0 &= ~((u64)TE_VALID << 56);                                               // :311-312 forget the entry, walk again
        m.tlb(ix) = e0;
      }
      u64 n0, n1;
      const u32 cause = page_walk(va, pm, acc, true, pa, n0, n1);
      if (cause) { pa = va; return cause; }
      if (((e0 >> 56) & TE_VALID) && ((e0 >> 54) & 2)) m.tlb(ix) = e0 - (1ULL << 54);   // Tlb::insertEntry (Tlb.cpp:74-92): busy slot, age it
      else { m.tlb(ix) = n0; m.tlb(ix + 1) = n1; }
      return 0;
    }
    // VirtMem::transAddrNoUpdate (VirtMem.cpp:143-205)
    WSP_HD u32 translate_noupdate(u64 va, u32 pm, u32 acc, u64& pa)
    {
      pa = va;
      const u64 vpn = va >> 12, satp = m.cold(CI_SATP);
      const u32 ix = 2 * (u32)(vpn & (tlb_n - 1));
      const u64 e0 = m.tlb(ix), e1 = m.tlb(ix + 1);
      if (tlb_match(e0, e1, vpn, satp)) {
        if (!tlb_hit_ok((u32)(e0 >> 56), pm, acc)) return pf_cause(acc);
        pa = ((e1 & 0xfffffffffffULL) << 12) | (va & 0xfff);
        return 0;
      }
      u64 n0, n1;
      const u32 cause = page_walk(va, pm, acc, false, pa, n0, n1);
      if (cause) pa = va;
      return cause;
    }
    // Tlb::setMode on a satp.MODE change (Tlb.hpp:378-383) / Tlb::invalidate: valid and counter cleared, the rest kept
    WSP_HD void tlb_invalidate_all() { for (u32 i = 0; i < tlb_n; ++i) m.tlb(2 * i) &= ~((3ULL << 54) | ((u64)TE_VALID << 56)); }
    // execSfence_vma's TLB part (Hart.cpp:12139-12173; Tlb.hpp:176-284): entries are matched whether or not they are valid
    WSP_HD void sfence(bool has_addr, u64 addr, bool has_asid, u64 asid_v)
    {
      const u64 KEEP = ~((3ULL << 54) | ((u64)TE_VALID << 56));
      const u64 asid = asid_v & 0xffffffffULL, vpn = addr >> 12;
      const bool sv39 = (m.cold(CI_SATP) >> 60) == 8;                                // Tlb::mode_: sizeIn4kBytes(Bare) = 0
      if (!has_addr) {
        for (u32 i = 0; i < tlb_n; ++i) {
          const u64 e0 = m.tlb(2 * i);
          if (!has_asid || (!((e0 >> 56) & TE_GLOBAL) && ((m.tlb(2 * i + 1) >> 44) & 0xffff) == asid)) m.tlb(2 * i) = e0 & KEEP;
        }
        return;
      }
      u64 max_size = 0, vpn_of_max = 0;
      for (u32 i = 0; i < tlb_n; ++i) {
        const u64 e0 = m.tlb(2 * i), evpn = e0 & TE_VPN_MASK; const u32 level = (u32)(e0 >> 52) & 3;
        const u64 size = !sv39 ? 0 : level <= 1 ? 1 : level == 2 ? 512 : 262144;
        const u64 pn = (!sv39 || level <= 1) ? evpn : level == 2 ? (evpn >> 9) << 9 : (evpn >> 18) << 18;
        if (pn <= vpn && vpn < pn + size && (!has_asid || (((m.tlb(2 * i + 1) >> 44) & 0xffff) == asid && !((e0 >> 56) & TE_GLOBAL)))) {
          if (size > max_size) { max_size = size; vpn_of_max = evpn; }
          m.tlb(2 * i) = e0 & KEEP;
        }
      }
      if (max_size > 1 && vpn_of_max <= vpn && vpn < vpn_of_max + max_size)          // "subpages covered by super-page"
        for (u32 i = 0; i < tlb_n; ++i) {
          const u64 e0 = m.tlb(2 * i);
          if (!has_asid || (((m.tlb(2 * i + 1) >> 44) & 0xffff) == asid && !((e0 >> 56) & TE_GLOBAL))) m.tlb(2 * i) = e0 & KEEP;
        }
    }

  };
  struct TrRes { u64 pa; u32 cause; };
  WSP_HD Vm vm() const { Vm v; v.m = m; v.mstatus = h.mstatus; v.mem_base = mem_base; v.mem_size = mem_size; v.flags = h.flags; v.tlb_n = tlb_n; v.isa = isa; return v; }
  static WSP_HDN TrRes vm_translate_miss(Vm v, u64 va, u32 pm, u32 acc) { TrRes r; r.cause = v.translate_miss(va, pm, acc, r.pa); return r; }
  static WSP_HDN TrRes vm_translate_noupdate(Vm v, u64 va, u32 pm, u32 acc) { TrRes r; r.cause = v.translate_noupdate(va, pm, acc, r.pa); return r; }
  static WSP_HDN void vm_tlb_invalidate_all(Vm v) { v.tlb_invalidate_all(); }
  static WSP_HDN void vm_sfence(Vm v, bool has_addr, u64 addr, bool has_asid, u64 asid_v) { v.sfence(has_addr, addr, has_asid, asid_v); }
  WSP_HD u32 pmp_perm(u32 pm, u64 addr) { return vm().pmp_perm(pm, addr); }
  WSP_HD bool paging_on(u32 pm) { return ext(ISA_S) && pm != PRV_M && (m.cold(CI_SATP) >> 60) == 8; }
  WSP_HD static u32 pf_cause(u32 acc) { return acc == 4 ? 12u : acc == 1 ? 13u : 15u; }      // stage1PageFaultType (VirtMem.cpp:31-40)
  // VirtMem::translate (VirtMem.cpp:261-329): 0 or the exception cause.  The TLB hit is inline (one 16-byte entry, the
  // permission test, the replacement counter's store); everything else goes out of line.
  WSP_HD u32 translate(u64 va, u32 pm, u32 acc, u64& pa)
  {
    const u64 vpn = va >> 12, satp = m.cold(CI_SATP);
    const u32 ix = 2 * (u32)(vpn & (tlb_n - 1));
    const u64 e0 = m.tlb(ix), e1 = m.tlb(ix + 1);
    if (Vm::tlb_match(e0, e1, vpn, satp)) {
      const u32 fl = (u32)(e0 >> 56);
      const bool sum = (h.mstatus & MST_SUM) != 0, mxr = (h.mstatus & MST_MXR) != 0;
      const bool bad = (pm == PRV_U && !(fl & TE_USER)) || (pm == PRV_S && (fl & TE_USER) && (acc == 4 || !sum)) ||
                       (acc == 1 && !((fl & TE_R) || (mxr && (fl & TE_X)))) || (acc == 2 && !(fl & TE_W)) || (acc == 4 && !(fl & TE_X));
      if (!bad && (fl & TE_A) && (acc != 2 || (fl & TE_D))) {
        m.tlb(ix) = (e0 & ~(3ULL << 54)) | ((((e0 >> 54) + 1) & 3) << 54);          // findEntryUpdateTime (Tlb.hpp:91-100)
        pa = ((e1 & 0xfffffffffffULL) << 12) | (va & 0xfff);
        return 0;
      }
    }
    const TrRes r = vm_translate_miss(vm(), va, pm, acc);
    pa = r.pa;
    return r.cause;
  }
  WSP_HD u32 translate_noupdate(u64 va, u32 pm, u32 acc, u64& pa) { const TrRes r = vm_translate_noupdate(vm(), va, pm, acc); pa = r.pa; return r.cause; }
  WSP_HD void tlb_invalidate_all() { vm_tlb_invalidate_all(vm()); }
  WSP_HD void sfence(bool has_addr, u64 addr, bool has_asid, u64 asid_v) { vm_sfence(vm(), has_addr, addr, has_asid, asid_v); }

  // determineLoadException / determineStoreException (Hart.cpp:1988-2154, 13786-13935).  need: 1 load / LR, 2 store / SC / AMO
  // (validateAmoAddr checks write permission only, amo.cpp:28-56).  0 and the physical addresses of the (up to two) parts,
  // or the exception cause with tval.
  WSP_HD u32 ldst_addr(u64 va, u32 size, u32 need, u64& pa1, u64& pa2, u64& tval)
  {
    const u32 pm = eff_priv(), af = need == 1 ? 5u : 7u;
    const bool tr = paging_on(pm);
    const u64 mask = size - 1;
    const bool misal = (va & mask) != 0;
    u32 c;
    pa1 = va;
    if (misal && ext(CFG_NO_MISAL)) { tval = va; return need == 1 ? 4u : 6u; }           // misalHasPriority_ and not misalDataOk_ (Hart.cpp:2013-2016)
    if (tr && (c = translate(va, pm, need, pa1)) != 0) { tval = va; return c; }
    if (!misal) {
      if (!(pmp_perm(pm, pa1) & need) || !in_mem(pa1, size)) { tval = va; return af; }
      pa2 = pa1;
      return 0;
    }
    const u64 va2 = (va + size - 1) & ~mask;
    const bool cross = (va >> 12) != (va2 >> 12);
    pa2 = (pa1 + size - 1) & ~mask;
    if (cross && tr && (c = translate(va2, pm, need, pa2)) != 0) { tval = va2; return c; }
    const u64 n1 = ((pa1 + size - 1) & ~mask) - pa1;                               // lower part stops at the alignment boundary (:2068-2069)
    if (!(pmp_perm(pm, pa1) & need) || !in_mem(pa1, n1)) { tval = va; return af; }
    if (!(pmp_perm(pm, pa2) & need) || !in_mem(pa2, size)) { tval = va2; return af; }
    if (!cross) pa2 = pa1;
    return 0;
  }
  // satp.MODE Sv48 / Sv57 can only get in through a host poke
  WSP_HD bool paging_unsupported()
  {
    if (ext(ISA_S) && eff_priv() != PRV_M && (m.cold(CI_SATP) >> 60) > 8) { stop(ST_STOP_UNSUPPORTED, UNS_PAGING); return true; }
    return false;
  }

  WSP_HD u64 mem_read(u64 off, u32 size)
  {
    if (!(off & (size - 1))) {
      switch (size) { case 1: return m.ld8(off); case 2: return m.ld16(off); case 4: return m.ld32(off); default: return m.ld64(off); }
    }
    u64 v = 0;
    for (u32 i = 0; i < size; ++i) v |= (u64)m.ld8(off + i) << (8 * i);
    return v;
  }
  WSP_HD void mem_write(u64 off, u32 size, u64 v)
  {
    if (!(off & (size - 1))) {
      switch (size) { case 1: m.st8(off, (u8)v); break; case 2: m.st16(off, (u16)v); break; case 4: m.st32(off, (u32)v); break; default: m.st64(off, v); break; }
      return;
    }
    for (u32 i = 0; i < size; ++i) m.st8(off + i, (u8)(v >> (8 * i)));
  }
  // Hart::memRead / memWrite over an item that may span two physical pages (Hart.hpp:3469-3567)
  WSP_HD u64 mem_read2(u64 pa1, u64 pa2, u32 size)
  {
    if (pa1 == pa2) return mem_read(pa1 - mem_base, size);
    const u32 size1 = size - (u32)(pa1 & (size - 1));
    u64 v = 0;
    for (u32 i = 0; i < size; ++i) v |= (u64)m.ld8((i < size1 ? pa1 + i : pa2 + (i - size1)) - mem_base) << (8 * i);
    return v;
  }
  WSP_HD void mem_write2(u64 pa1, u64 pa2, u32 size, u64 v)
  {
    if (pa1 == pa2) { mem_write(pa1 - mem_base, size, v); return; }
    const u32 size1 = size - (u32)(pa1 & (size - 1));
    for (u32 i = 0; i < size; ++i) m.st8((i < size1 ? pa1 + i : pa2 + (i - size1)) - mem_base, (u8)(v >> (8 * i)));
  }

  // Hart::load + readForLoad (Hart.cpp:2302, 2477)
  // Hart::load / store / amoLoad / lr / sc note the access for the log before anything can fault (Hart.cpp:2306-2309, 2678-2681,
  // amo.cpp:68-72, 146-150, 277-280), and the physical address once it is known (Hart.cpp:2333, 2710)
  WSP_HD void note_ldst(u64 va) { if (ring_) { s.lva = va; s.lpa = va; s.chg |= CH_LDST; } }
  WSP_HD void note_ldst_pa(u64 pa) { if (ring_) s.lpa = pa; }
  WSP_HD bool do_load(u64 va, u32 size, u64 cur_pc, u64& out)
  {
    u64 tval, pa1, pa2;
    va &= amask();
    note_ldst(va);
    if (paging_unsupported()) return false;
    const u32 c = ldst_addr(va, size, 1, pa1, pa2, tval);
    if (c) { take_exception(c, cur_pc, tval, false); return false; }
    note_ldst_pa(pa1);
    if (pa1 == tohost) { out = 0; return true; }                           // Hart.cpp:2509-2513
    out = mem_read2(pa1, pa2, size);
    return true;
  }
  // Hart::store + writeForStore + handleStoreToHost (Hart.cpp:2674, 2727, 2615)
  WSP_HD bool do_store(u64 va, u32 size, u64 val, u64 cur_pc)
  {
    u64 tval, pa1, pa2;
    va &= amask();
    note_ldst(va);
    if (paging_unsupported()) return false;
    const u32 c = ldst_addr(va, size, 2, pa1, pa2, tval);
    if (c) { take_exception(c, cur_pc, tval, false); return false; }
    note_ldst_pa(pa1);
    u64 sv = size == 8 ? val : (val & ((1ULL << (8 * size)) - 1));
    s.chg |= CH_MEM; s.maddr = va; s.mval = sv;                            // the log's m line carries the virtual address (printTrace.cpp:457)
    if (pa1 + size > text_lo && pa1 < text_hi) h.flags |= HF_CODE_DIRTY;    // invalidateDecodeCache (Hart.cpp:7375, 2751-2753)
    if (pa2 != pa1 && pa2 + size > text_lo && pa2 < text_hi) h.flags |= HF_CODE_DIRTY;
    if (tohost != ~0ULL && (pa1 == tohost || pa1 == tohost + 4)) {
      mem_write(pa1 - mem_base, size, sv);
      u64 v = sv;
      if (size == 4) { if (pa1 == tohost + 4) { h.htif_dev = (u32)sv; return true; } v |= (u64)h.htif_dev << 32; }
      else if (size == 8) h.htif_dev = 0;
      u32 cmd = (v >> 48) & 0xff, dev = (v >> 56) & 0xff;
      if (dev == 0 && cmd == 0 && (sv & 1)) stop(ST_STOP_TOHOST, v);
      return true;
    }
    mem_write2(pa1, pa2, size, sv);
    return true;
  }

  // ---- CSR file (subset). class: 1 modelled, 0 not implemented, -1 unsupported.
  WSP_HD int csr_class(u32 num) const
  {
    bool sup = ext(ISA_S);
    switch (num) {
      case C_FFLAGS: case C_FRM: case C_FCSR: return ext(ISA_F) ? 1 : 0;
      case C_SSTATUS: case C_SIE: case C_STVEC: case C_SCOUNTEREN: case C_SENVCFG: case C_SSCRATCH: case C_SEPC:
      case C_SCAUSE: case C_STVAL: case C_SIP: case C_SATP: case C_MEDELEG: case C_MIDELEG: return sup ? 1 : 0;
      case C_MSTATUS: case C_MISA: case C_MIE: case C_MTVEC: case C_MCOUNTEREN: case C_MENVCFG: case C_MCOUNTINHIBIT:
      case C_MSCRATCH: case C_MEPC: case C_MCAUSE: case C_MTVAL: case C_MIP: case C_MCYCLE: case C_MINSTRET:
      case C_MVENDORID: case C_MARCHID: case C_MIMPID: case C_MHARTID: case C_MCONFIGPTR: return 1;
      case C_CYCLE: case C_TIME: case C_INSTRET: return ext(ISA_ZICNTR) ? 1 : 0;
      case C_VSTART: case C_VXSAT: case C_VXRM: case C_VCSR: case C_VL: case C_VTYPE: case C_VLENB: return ext(ISA_V) ? 1 : 0;   // CsRegs::enableVector
      default: break;
    }
    if (rv32()) {   // CsRegs<uint32_t>: the high halves exist, and so do the odd pmpcfg registers
      if (num == 0x310 || num == 0x31A || num == 0xB80 || num == 0xB82) return 1;                                  // mstatush menvcfgh mcycleh minstreth
      if (num >= 0xC80 && num <= 0xC82) return ext(ISA_ZICNTR) ? 1 : 0;                                            // cycleh timeh instreth
      if (num >= 0x3A0 && num <= 0x3EF) return 1;
      if ((num >= 0x723 && num <= 0x73F) || num == 0x757 || (num >= 0xB83 && num <= 0xB9F)) return 1;             // mhpmevent<n>h mseccfgh mhpmcounter<n>h: zero (below)
    }
    if (num >= 0x3A0 && num <= 0x3EF) return (num < 0x3B0 && (num & 1)) ? 0 : 1;   // pmpcfg (even only on rv64), pmpaddr0..63
    // mhpmcounter3..31, mhpmevent3..31 and mseccfg exist with all-zero write masks in the default configuration (no performance
    // counters configured, no Smepmp: CsRegs.cpp:4810, 4848-4866): they read zero, a write changes nothing and is still logged
    if ((num >= 0xB03 && num <= 0xB1F) || (num >= 0x323 && num <= 0x33F) || num == 0x747) return 1;
    if (num == 0x7C7) return -1;                                                        // c_matp (a custom register of the reference)
    return 0;                                                                           // (dcsr / dpc: debug mode only, CsRegs.cpp:1252)
  }
  // CY/TM/IR of m/scounteren read as zero while cycle/time/instret are not implemented (CsRegs.cpp:1524-1539, 2117-2130)
  WSP_HD u64 counteren_rmask() const { return ext(ISA_ZICNTR) ? ~0ULL : ~7ULL; }
  // CsRegs::enableRvc (CsRegs.hpp:2119-2127): xEPC bit 0 reads zero with C, bits 1:0 without; the stored value keeps bit 1
  WSP_HD u64 epc_rmask() const { return ext(ISA_C) ? ~1ULL : ~3ULL; }
  // CsRegs::legalizeEnvcfg (CsRegs.cpp:6719): a reserved CBIE (2) keeps the previous field value
  WSP_HD static u64 legalize_envcfg(u64 prev, u64 next) { return ((next >> 4) & 3) == 2 ? (next & ~0x30ULL) | (prev & 0x30) : next; }
  // The CSR file is kept in its RV64 shape; an RV32 hart sees 32-bit views of it (CsRegs<uint32_t>): the low half of each
  // register with mstatus.SD and the interrupt bit of the cause registers at bit 31, the high halves under their own numbers,
  // pmpcfg<2k+1> = the upper half of row k.
  WSP_HD u64 csr_peek(u32 num)
  {
    if (!rv32()) return csr_peek64(num);
    switch (num) {
      case C_MSTATUS: case C_SSTATUS: case C_MCAUSE: case C_SCAUSE: { const u64 v = csr_peek64(num); return (v & 0x7fffffffULL) | ((v >> 63) << 31); }
      case C_MISA: return 0x4034112dULL;
      case 0x310: return 0;                                                    // mstatush: MBE / SBE hardwired zero
      case 0x31A: return csr_peek64(C_MENVCFG) >> 32;
      case 0xB80: case 0xC80: return h.cycle >> 32;
      case 0xB82: case 0xC82: return h.instret >> 32;
      case 0xC81: return h.exec >> 32;
      default:
        if (num >= 0x3A0 && num <= 0x3AF) return (m.cold(CI_PMPCFG0 + ((num - 0x3A0) >> 1)) >> (32 * (num & 1))) & 0xffffffffULL;
        return csr_peek64(num) & 0xffffffffULL;
    }
  }
  WSP_HD u64 csr_peek64(u32 num)
  {
    switch (num) {
      case C_FFLAGS: return h.fcsr & 0x1f;
      case C_FRM: return (h.fcsr >> 5) & 7;
      case C_FCSR: return h.fcsr;
      case C_VSTART: return m.cold(CI_VSTART); case C_VXSAT: return m.cold(CI_VCSR) & 1; case C_VXRM: return (m.cold(CI_VCSR) >> 1) & 3;
      case C_VCSR: return m.cold(CI_VCSR); case C_VL: return m.cold(CI_VL); case C_VTYPE: return m.cold(CI_VTYPE); case C_VLENB: return vlenb;
      case C_SSTATUS: return h.mstatus & SSTATUS_RMASK;
      case C_SIE: return m.cold(CI_MIE) & m.cold(CI_MIDELEG) & ~0x1444ULL;   // CsRegs::readSie (CsRegs.cpp:195-229)
      case C_SIP: return m.cold(CI_MIP) & m.cold(CI_MIDELEG) & ~0x1444ULL;
      case C_STVEC: return m.cold(CI_STVEC); case C_SCOUNTEREN: return m.cold(CI_SCOUNTEREN) & counteren_rmask(); case C_SENVCFG: return m.cold(CI_SENVCFG);
      case C_SSCRATCH: return m.cold(CI_SSCRATCH); case C_SEPC: return m.cold(CI_SEPC) & epc_rmask(); case C_SCAUSE: return m.cold(CI_SCAUSE);
      case C_STVAL: return m.cold(CI_STVAL); case C_SATP: return m.cold(CI_SATP);
      case C_MSTATUS: return (ext(ISA_F) || ext(ISA_S)) ? h.mstatus : (h.mstatus & ~(3ULL << MST_FS_SH));   // CsRegs.cpp:1977-1987
      case C_MISA: return MISA_RESET;
      case C_MEDELEG: return m.cold(CI_MEDELEG); case C_MIDELEG: return m.cold(CI_MIDELEG); case C_MIE: return m.cold(CI_MIE);
      case C_MTVEC: return m.cold(CI_MTVEC); case C_MCOUNTEREN: return m.cold(CI_MCOUNTEREN) & counteren_rmask(); case C_MENVCFG: return m.cold(CI_MENVCFG);
      case C_MCOUNTINHIBIT: return m.cold(CI_MCOUNTINHIBIT);
      case C_MSCRATCH: return m.cold(CI_MSCRATCH); case C_MEPC: return m.cold(CI_MEPC) & epc_rmask(); case C_MCAUSE: return m.cold(CI_MCAUSE);
      case C_MTVAL: return m.cold(CI_MTVAL); case C_MIP: return m.cold(CI_MIP);
      case C_MCYCLE: case C_CYCLE: return h.cycle;
      case C_MINSTRET: case C_INSTRET: return h.instret;
      case C_TIME: return h.exec;               // time_ ticks once per loop iteration (Hart.hpp:3101, 1 hart, downsample 0)
      case C_MHARTID: return hart_id;
      default:
        if (num >= 0x3A0 && num <= 0x3AF) return m.cold(CI_PMPCFG0 + ((num - 0x3A0) >> 1));
        if (num >= 0x3B0 && num <= 0x3EF) return m.cold(CI_PMPADDR0 + (num - 0x3B0));    // grain G = 0: read value = stored value
        return 0;
    }
  }
  // value update of CsRegs::write (CsRegs.cpp:3908-4165) for the modelled subset
  WSP_HD void csr_store(u32 num, u64 v)
  {
    bool sup = ext(ISA_S);
    if (rv32()) {                                          // a 32-bit write, brought into the RV64 shape of the file
      v &= 0xffffffffULL;
      switch (num) {
        case C_MCAUSE: case C_SCAUSE: v = (v & 0x7fffffffULL) | ((v >> 31) << 63); break;
        case C_MENVCFG: v |= m.cold(CI_MENVCFG) & ~0xffffffffULL; break;
        case 0x31A: num = C_MENVCFG; v = (v << 32) | (m.cold(CI_MENVCFG) & 0xffffffffULL); break;
        case 0x310: return;                                                    // mstatush: nothing writable
        case C_MCYCLE: v |= h.cycle & ~0xffffffffULL; break;
        case C_MINSTRET: v |= h.instret & ~0xffffffffULL; break;
        case 0xB80: h.cycle = (h.cycle & 0xffffffffULL) | (v << 32); return;
        case 0xB82: h.instret = (h.instret & 0xffffffffULL) | (v << 32); return;
        default:
          if (num >= 0x3A0 && num <= 0x3AF) {              // pmpcfg<n>: four entries, half of row n / 2
            const u32 k = CI_PMPCFG0 + ((num - 0x3A0) >> 1), sh = 32 * (num & 1);
            const u64 keep = m.cold(k) & ~(0xffffffffULL << sh);
            m.cold(k) = legalize_pmpcfg(m.cold(k), keep | (v << sh)) & 0x9f9f9f9f9f9f9f9fULL;
            update_pmp_flags();
            return;
          }
          break;
      }
    }
    switch (num) {
      case C_FFLAGS: h.fcsr = (h.fcsr & ~0x1fu) | (u32)(v & 0x1f); break;
      case C_FRM: h.fcsr = (h.fcsr & ~0xe0u) | (u32)((v & 7) << 5); break;
      case C_FCSR: h.fcsr = (u32)(v & 0xff); break;
      case C_VSTART: m.cold(CI_VSTART) = v & (vlenb * 8 - 1); break;          // mask = VLEN - 1 (Hart.cpp:1088-1103)
      case C_VXSAT: m.cold(CI_VCSR) = (m.cold(CI_VCSR) & 6) | (v & 1); break;  // CsRegs.cpp:4556-4606
      case C_VXRM: m.cold(CI_VCSR) = (m.cold(CI_VCSR) & 1) | ((v & 3) << 1); break;
      case C_VCSR: m.cold(CI_VCSR) = v & 7; break;
      case C_VL: m.cold(CI_VL) = v; break;                                     // poke only
      case C_VTYPE: m.cold(CI_VTYPE) = v & 0x80000000000000ffULL; break;
      case C_MSTATUS: {   // value &= writeMask & readMask (CsRegs.cpp:3964): FS is read-only zero without F and S
        const u64 k = (ext(ISA_F) || ext(ISA_S)) ? MSTATUS_WMASK : (MSTATUS_WMASK & ~(3ULL << MST_FS_SH));
        h.mstatus = legalize_mstatus((h.mstatus & ~MSTATUS_WMASK) | (v & k)); break; }
      case C_SSTATUS: h.mstatus = legalize_mstatus((h.mstatus & ~SSTATUS_WMASK) | (v & SSTATUS_WMASK)); break;
      case C_MEDELEG: m.cold(CI_MEDELEG) = v & MEDELEG_MASK; break;
      case C_MIDELEG: m.cold(CI_MIDELEG) = v & ~0x1444ULL; break;   // CsRegs.cpp:4734 all bits; VS/SGEI bits read-only zero without H (1766-1771)
      case C_MIE: { u64 k = 0x999 | (sup ? 0x222 : 0);   /* CsRegs.cpp:4739 mask 0xfff less the S (CsRegs.cpp:1509) and VS bits: legacy U bits 0,4,8 stay writable */ m.cold(CI_MIE) = (m.cold(CI_MIE) & ~k) | (v & k); break; }
      case C_MIP: { u64 k = sup ? 0x222 : 0; m.cold(CI_MIP) = (m.cold(CI_MIP) & ~k) | (v & k); break; }
      case C_SIE: { u64 k = m.cold(CI_MIDELEG) & (sup ? 0x222 : 0);   /* CsRegs::writeSie: sie mask & mie mask & mideleg (CsRegs.cpp:2745-2790) */ m.cold(CI_MIE) = (m.cold(CI_MIE) & ~k) | (v & k); break; }
      case C_SIP: { u64 k = m.cold(CI_MIDELEG) & (sup ? 0x002 : 0);   /* CsRegs::writeSip (CsRegs.cpp:2680-2740) */ m.cold(CI_MIP) = (m.cold(CI_MIP) & ~k) | (v & k); break; }
      case C_MTVEC: m.cold(CI_MTVEC) = v & ~2ULL; break;
      case C_STVEC: m.cold(CI_STVEC) = v & ~2ULL; break;
      case C_MCOUNTEREN: m.cold(CI_MCOUNTEREN) = v & 0xffffffffULL; break;
      case C_SCOUNTEREN: m.cold(CI_SCOUNTEREN) = v; break;
      case C_MENVCFG: m.cold(CI_MENVCFG) = legalize_envcfg(m.cold(CI_MENVCFG), v) & 0xe0000003000000f5ULL; break;
      case C_SENVCFG: m.cold(CI_SENVCFG) = legalize_envcfg(m.cold(CI_SENVCFG), v) & 0x00000003000000f5ULL; break;
      case C_MCOUNTINHIBIT: {
        u64 nv = v & 0xfffffffdULL; m.cold(CI_MCOUNTINHIBIT) = nv;
        h.flags = (h.flags & ~(HF_INH_CY | HF_INH_IR)) | ((nv & 1) ? (u32)HF_INH_CY : 0) | ((nv & 4) ? (u32)HF_INH_IR : 0);
        break; }
      case C_MSCRATCH: m.cold(CI_MSCRATCH) = v; break;
      case C_SSCRATCH: m.cold(CI_SSCRATCH) = v; break;
      case C_MEPC: m.cold(CI_MEPC) = v & ~1ULL; break;
      case C_SEPC: m.cold(CI_SEPC) = v & ~1ULL; break;
      case C_MCAUSE: m.cold(CI_MCAUSE) = v; break;
      case C_SCAUSE: m.cold(CI_SCAUSE) = v; break;
      case C_MTVAL: m.cold(CI_MTVAL) = v; break;
      case C_STVAL: m.cold(CI_STVAL) = v; break;
      case C_SATP:                                         // updateAddressTranslation -> configTranslation -> Tlb::setMode (Hart.cpp:904-919, Tlb.hpp:378-383)
        if ((m.cold(CI_SATP) >> 60) != (v >> 60)) tlb_invalidate_all();
        m.cold(CI_SATP) = v; break;
      case C_MCYCLE: h.cycle = v; break;
      case C_MINSTRET: h.instret = v; break;
      default:
        if (num >= 0x3A0 && num <= 0x3AF) {                    // CsRegs.cpp:4777-4784 mask, legalised per byte
          const u32 k = CI_PMPCFG0 + ((num - 0x3A0) >> 1);
          m.cold(k) = legalize_pmpcfg(m.cold(k), v) & 0x9f9f9f9f9f9f9f9fULL;
          update_pmp_flags();
        } else if (num >= 0x3B0 && num <= 0x3EF) {
          if (!pmpaddr_locked(num - 0x3B0)) { m.cold(CI_PMPADDR0 + (num - 0x3B0)) = v & 0x003fffffffffffffULL; if (num == 0x3B0) update_pmp_flags(); }
        }
        break;
    }
  }
  WSP_HD bool csr_readable(u32 num)
  {
    u32 need = (num >> 8) & 3; if (need == 2) need = 3;
    if (h.priv < need) return false;
    if (num == C_CYCLE || num == C_TIME || num == C_INSTRET) {
      u32 bit = num - C_CYCLE;
      if (h.priv < PRV_M && !((m.cold(CI_MCOUNTEREN) >> bit) & 1)) return false;
      if (h.priv == PRV_U && ext(ISA_S) && !((m.cold(CI_SCOUNTEREN) >> bit) & 1)) return false;
    }
    return true;
  }
  // execCsrr* -> doCsrRead / doCsrWrite (Hart.cpp:13401-13748, 12987, 13244)
  WSP_HD void exec_csr(const Rec& r, u32 op /*0 rw 1 rs 2 rc*/, u64 src, u64 cur_pc)
  {
    u32 num = (u32)r.imm & 0xfff, rd = rec_rd(r);
    bool src_nz = rec_rs1(r) != 0;
    int cls = csr_class(num);
    bool do_read = op != 0 || rd != 0, do_write = op == 0 || src_nz;
    if (cls < 0) { stop(ST_STOP_UNSUPPORTED, UNS_CSR_BASE | num); return; }
    if (cls == 0) { illegal(r, cur_pc); return; }
    if (num <= C_FCSR && !fp_on()) { illegal(r, cur_pc); return; }                       // Hart.cpp:12945-12950
    if (((num >= C_VSTART && num <= C_VCSR) || (num >= C_VL && num <= C_VLENB)) && !vec_on()) { illegal(r, cur_pc); return; }   // Hart.cpp:12952-12958
    if (!csr_readable(num) || (do_write && ((num >> 10) & 3) == 3)) { illegal(r, cur_pc); return; }
    if (num == C_SATP && h.priv == PRV_S && (h.mstatus & MST_TVM)) { illegal(r, cur_pc); return; }
    u64 prev = do_read ? csr_peek(num) : 0;
    if (!do_write) { wx(rd, prev); return; }
    u64 next = op == 0 ? src : op == 1 ? (prev | src) : (prev & ~src);
    // the reference pre-increments then post-decrements the counter being written so that the run
    // loop's own increment is cancelled (Hart.cpp:13251-13255, 13354-13360)
    bool w_ir = (num == C_MINSTRET || (rv32() && num == 0xB82)) && !(h.flags & HF_INH_IR), w_cy = num == C_MCYCLE || (rv32() && num == 0xB80);
    if (w_ir) h.instret++;
    if (w_cy) h.cycle++;
    wx(rd, prev);
    if (num == C_SATP) {
      u32 mode = (u32)(next >> 60);
      if ((mode != 0 && mode != 8 && mode != 9 && mode != 10) || (mode == 8 && ext(CFG_NO_SV39))) return;                      // unsupported MODE: write dropped (13284-13294)
      if (mode > 8) { stop(ST_STOP_UNSUPPORTED, UNS_PAGING); return; }                   // Sv48 / Sv57: not modelled
    }
    if (num >= 0x3B0 && num < 0x3EF && !pmpaddr_locked(num - 0x3B0)) {
      // postCsrUpdate (Hart.cpp:4672-4679) refreshes the region table only when the written entry itself is on: a pmpaddr
      // write under an Off entry whose successor is an active TOR entry leaves the reference with a stale lower bound.
      // Not modelled: stop instead of diverging.
      const u32 ix = num - 0x3B0;
      if (((pmp_cfg_byte(ix) >> 3) & 3) == 0 && ((pmp_cfg_byte(ix + 1) >> 3) & 3) == 1 &&
          (next & 0x003fffffffffffffULL) != m.cold(CI_PMPADDR0 + ix)) { stop(ST_STOP_UNSUPPORTED, UNS_PMP_STALE); return; }
    }
    csr_store(num, next);
    if (num == C_MSTATUS) s.chg |= CH_MSTATUS; else if (num == C_SSTATUS) s.chg |= CH_SSTATUS;
    else { s.chg |= CH_GEN; s.gen_csr = num; }
    if (num <= C_FCSR) mark_fs_dirty();
    else if (num <= C_VCSR) mark_vs_dirty();                                             // postCsrUpdate (Hart.cpp:4731-4734)
    if (w_ir) h.instret--;
    if (w_cy) h.cycle--;
  }

  // ---- floating point (float.cpp). Returns false when the instruction is illegal.
  WSP_HD bool fp_rm(u32 rmf, int& rm) const { u32 k = rmf == 7 ? ((h.fcsr >> 5) & 7) : rmf; rm = (int)k; return k < 5; }   // float.cpp:200-216

  template <class F> WSP_HD u64 fneg(u64 a, unsigned& fl)                                 // doNegate (float-util.hpp:743-758)
  { if (!(a & (F::SIGN - 1))) return a ^ F::SIGN; return wsf::f_sub<F>(0, a, 0 /*exact unless NaN*/, fl); }

  WSP_HD void exec_fp(const Rec& r, u64 a /*x[rs1]*/, u64 cur_pc)
  {
    using namespace wsf;
    u32 op = rec_op(r), rd = rec_rd(r), rs1 = rec_rs1(r), rs2 = rec_rs2(r);
    bool dbl = (r.w0 & F_DBL) != 0;
    if (!fp_on()) { illegal(r, cur_pc); return; }
    unsigned fl = 0; int rm = 0;
    switch (op) {
      case OP_FADD: case OP_FSUB: case OP_FMUL: case OP_FDIV: case OP_FSQRT: {
        if (!fp_rm((u32)r.imm & 7, rm)) { illegal(r, cur_pc); return; }
        if (dbl) {
          u64 x = m.rf(rs1), y = m.rf(rs2), z = 0;
          // round to nearest even, ordinary operands: the FP unit's own result with the exact inexact flag (wsp_softfloat.h)
          const bool hw = rm == RM_RNE && (op == OP_FADD ? hw_add64(x, y, z, fl) : op == OP_FSUB ? hw_add64(x, y ^ F64::SIGN, z, fl)
                                           : op == OP_FMUL ? hw_mul64(x, y, z, fl) : op == OP_FDIV ? hw_div64(x, y, z, fl) : false);
          if (!hw)
            z = op == OP_FADD ? f_add<F64>(x, y, rm, fl) : op == OP_FSUB ? f_sub<F64>(x, y, rm, fl) : op == OP_FMUL ? f_mul<F64>(x, y, rm, fl)
              : op == OP_FDIV ? f_div<F64>(x, y, rm, fl) : f_sqrt<F64>(x, rm, fl);
          wf(rd, z);
        } else {
          u64 x = rf_s(rs1), y = rf_s(rs2), z = 0;
          const bool hw = rm == RM_RNE && (op == OP_FADD ? hw_add32(x, y, z, fl) : op == OP_FSUB ? hw_add32(x, y ^ F32::SIGN, z, fl)
                                           : op == OP_FMUL ? hw_mul32(x, y, z, fl) : op == OP_FDIV ? hw_div32(x, y, z, fl) : false);
          if (!hw)
            z = op == OP_FADD ? f_add<F32>(x, y, rm, fl) : op == OP_FSUB ? f_sub<F32>(x, y, rm, fl) : op == OP_FMUL ? f_mul<F32>(x, y, rm, fl)
              : op == OP_FDIV ? f_div<F32>(x, y, rm, fl) : f_sqrt<F32>(x, rm, fl);
          wf_s(rd, (u32)z);
        }
        or_fflags(fl); mark_fs_dirty(); return; }
      case OP_FMA: {
        if (!fp_rm((u32)r.imm & 7, rm)) { illegal(r, cur_pc); return; }
        u32 k = ((u32)r.imm >> 3) & 3, rs3 = rec_rs3(r);
        if (dbl) {
          u64 x = m.rf(rs1), y = m.rf(rs2), z = m.rf(rs3);
          u64 hz = 0;
          if (rm == RM_RNE && hw_fma64((k == 2 || k == 3) ? x ^ F64::SIGN : x, y, (k == 1 || k == 3) ? z ^ F64::SIGN : z, hz, fl)) { wf(rd, hz); or_fflags(fl); mark_fs_dirty(); return; }
          if (k == 1 || k == 3) z = fneg<F64>(z, fl);
          if (k == 2 || k == 3) x = fneg<F64>(x, fl);
          wf(rd, f_fma<F64>(x, y, z, rm, fl));
        } else {
          u64 x = rf_s(rs1), y = rf_s(rs2), z = rf_s(rs3);
          if (k == 1 || k == 3) z = fneg<F32>(z, fl);
          if (k == 2 || k == 3) x = fneg<F32>(x, fl);
          wf_s(rd, (u32)f_fma<F32>(x, y, z, rm, fl));
        }
        or_fflags(fl); mark_fs_dirty(); return; }
      case OP_FSGNJ: {                                                                     // float.cpp:499-564
        u32 k = (u32)r.imm;
        if (dbl) { u64 x = m.rf(rs1), y = m.rf(rs2), sb = F64::SIGN; u64 sg = k == 0 ? (y & sb) : k == 1 ? (~y & sb) : ((x ^ y) & sb); wf(rd, (x & ~sb) | sg); }
        else { u32 x = rf_s(rs1), y = rf_s(rs2), sb = 0x80000000u; u32 sg = k == 0 ? (y & sb) : k == 1 ? (~y & sb) : ((x ^ y) & sb); wf_s(rd, (x & ~sb) | sg); }
        mark_fs_dirty(); return; }
      case OP_FMINMAX: {                                                                   // float.cpp:569-634
        bool mx = r.imm != 0;
        if (dbl) {
          u64 x = m.rf(rs1), y = m.rf(rs2), z; bool n1 = is_nan<F64>(x), n2 = is_nan<F64>(y);
          if (n1 && n2) z = F64::QNAN; else if (n1) z = y; else if (n2) z = x;
          else if (!mx) z = (lt_ord<F64>(x, y) || (eq_ord<F64>(x, y) && (x >> 63))) ? x : y;
          else z = (lt_ord<F64>(y, x) || (eq_ord<F64>(x, y) && !(x >> 63))) ? x : y;
          if (is_snan<F64>(x) || is_snan<F64>(y)) or_fflags(FL_NV);
          wf(rd, z);
        } else {
          u64 x = rf_s(rs1), y = rf_s(rs2), z; bool n1 = is_nan<F32>(x), n2 = is_nan<F32>(y);
          if (n1 && n2) z = F32::QNAN; else if (n1) z = y; else if (n2) z = x;
          else if (!mx) z = (lt_ord<F32>(x, y) || (eq_ord<F32>(x, y) && (x >> 31))) ? x : y;
          else z = (lt_ord<F32>(y, x) || (eq_ord<F32>(x, y) && !(x >> 31))) ? x : y;
          if (is_snan<F32>(x) || is_snan<F32>(y)) or_fflags(FL_NV);
          wf_s(rd, (u32)z);
        }
        mark_fs_dirty(); return; }
      case OP_FCMP: {                                                                      // float.cpp:694-764
        u32 k = (u32)r.imm; u64 res = 0;
        if (dbl) {
          u64 x = m.rf(rs1), y = m.rf(rs2);
          if (is_nan<F64>(x) || is_nan<F64>(y)) { if (k != 2 || is_snan<F64>(x) || is_snan<F64>(y)) or_fflags(FL_NV); }
          else res = k == 2 ? eq_ord<F64>(x, y) : k == 1 ? lt_ord<F64>(x, y) : (lt_ord<F64>(x, y) || eq_ord<F64>(x, y));
        } else {
          u64 x = rf_s(rs1), y = rf_s(rs2);
          if (is_nan<F32>(x) || is_nan<F32>(y)) { if (k != 2 || is_snan<F32>(x) || is_snan<F32>(y)) or_fflags(FL_NV); }
          else res = k == 2 ? eq_ord<F32>(x, y) : k == 1 ? lt_ord<F32>(x, y) : (lt_ord<F32>(x, y) || eq_ord<F32>(x, y));
        }
        wx(rd, res); return; }
      case OP_FCVT_F2I: {                                                                  // float.cpp:639-669, 898-948
        if (!fp_rm((u32)r.imm & 7, rm)) { illegal(r, cur_pc); return; }
        bool sg = !(rs2 & 1), w = (rs2 & 2) != 0;
        u64 z = dbl ? f_to_int<F64>(m.rf(rs1), rm, sg, w, fl) : f_to_int<F32>(rf_s(rs1), rm, sg, w, fl);
        if (!w) z = (u64)(i64)(i32)(u32)z;
        wx(rd, z); or_fflags(fl); return; }
      case OP_FCVT_I2F: {
        if (!fp_rm((u32)r.imm & 7, rm)) { illegal(r, cur_pc); return; }
        bool sg = !(rs2 & 1), w = (rs2 & 2) != 0;
        if (dbl) wf(rd, rm == RM_RNE ? hw_int_to_f64(a, sg, w, fl) : int_to_f<F64>(a, sg, w, rm, fl)); else wf_s(rd, (u32)int_to_f<F32>(a, sg, w, rm, fl));
        or_fflags(fl); mark_fs_dirty(); return; }
      case OP_FCVT_FF: {
        if (!fp_rm((u32)r.imm & 7, rm)) { illegal(r, cur_pc); return; }
        if (dbl) wf(rd, f32_to_f64(rf_s(rs1), fl)); else wf_s(rd, (u32)f64_to_f32(m.rf(rs1), rm, fl));
        or_fflags(fl); mark_fs_dirty(); return; }
      case OP_FMV_F2X: wx(rd, dbl ? m.rf(rs1) : (u64)(i64)(i32)(u32)m.rf(rs1)); return;   // raw bits: float.cpp:682-688
      case OP_FCLASS: wx(rd, dbl ? fclass<F64>(m.rf(rs1)) : fclass<F32>(rf_s(rs1))); return;
      case OP_FMV_X2F: if (dbl) wf(rd, a); else wf_s(rd, (u32)a); mark_fs_dirty(); return;
      default: illegal(r, cur_pc); return;
    }
  }

  // ---- A extension (amo.cpp)
  WSP_HD void exec_atomic(const Rec& r, u64 va, u64 b, u64 cur_pc)
  {
    u32 op = rec_op(r), rd = rec_rd(r), size = (r.w0 & F_DBL) ? 8 : 4;
    u64 tval, pa1 = va, pa2;
    note_ldst(va);
    if (paging_unsupported()) return;
    if (op == OP_LR) {                                                                     // amo.cpp:142-266
      // determineLoadException first (a translation fault wins), then alignment; the trap value is the LR's own address
      u32 c = ldst_addr(va, size, 1, pa1, pa2, tval);
      if (c == 4 || (c == 0 && (va & (size - 1)))) c = 5;                                  // misalAtomicCauseAccessFault_ (amo.cpp:180-181, 188-201)
      if (c) { take_exception(c, cur_pc, va, false); return; }
      note_ldst_pa(pa1);
      u64 v = mem_read(pa1 - mem_base, size);
      if (size == 4) v = (u64)(i64)(i32)v;
      wx(rd, v);
      h.flags |= HF_LR_VALID; h.lr_addr = pa1 & ~7ULL;                                     // lrResSize_ = 8 (Hart.hpp:6733); physical (amo.cpp:254)
      return;
    }
    if (op == OP_SC) {                                                                     // amo.cpp:273-420
      u32 c = 0;
      if (va & (size - 1)) c = 7;                                                          // misalHasPriority_: before any translation (amo.cpp:304-309)
      else c = ldst_addr(va, size, 2, pa1, pa2, tval);
      bool had = (h.flags & HF_LR_VALID) && h.lr_addr <= pa1 && pa1 + size <= h.lr_addr + 8;
      h.flags &= ~HF_LR_VALID;                                                             // cancelLr(SC): amo.cpp:403-405
      if (c) { take_exception(c, cur_pc, va, false); return; }
      note_ldst_pa(pa1);
      if (!had) { wx(rd, 1); return; }
      u64 sv = size == 8 ? b : (b & 0xffffffffULL);
      s.chg |= CH_MEM; s.maddr = va; s.mval = sv;
      mem_write(pa1 - mem_base, size, sv);
      if (pa1 + size > text_lo && pa1 < text_hi) h.flags |= HF_CODE_DIRTY;
      wx(rd, 0);
      return;
    }
    {                                                                                      // validateAmoAddr (amo.cpp:28-56): misaligned first
      u32 c = (va & (size - 1)) ? 7u : ldst_addr(va, size, 2, pa1, pa2, tval);
      if (c) { take_exception(c, cur_pc, va, false); return; }
      note_ldst_pa(pa1);
    }
    u64 ld = mem_read(pa1 - mem_base, size), res;
    if (size == 4) ld = (u64)(i64)(i32)ld;
    switch ((u32)r.imm) {
      case 0: res = b + ld; break; case 1: res = b; break; case 4: res = b ^ ld; break; case 8: res = b | ld; break; case 12: res = b & ld; break;
      case 16: res = size == 4 ? (u64)(i64)((i32)b < (i32)ld ? (i32)b : (i32)ld) : (u64)((i64)b < (i64)ld ? (i64)b : (i64)ld); break;
      case 20: res = size == 4 ? (u64)(i64)((i32)b > (i32)ld ? (i32)b : (i32)ld) : (u64)((i64)b > (i64)ld ? (i64)b : (i64)ld); break;
      case 24: res = size == 4 ? (u64)((u32)b < (u32)ld ? (u32)b : (u32)ld) : (b < ld ? b : ld); break;
      default: res = size == 4 ? (u64)((u32)b > (u32)ld ? (u32)b : (u32)ld) : (b > ld ? b : ld); break;
    }
    if (!do_store(va, size, res, cur_pc)) return;
    wx(rd, ld);
  }

  // ---- RVV (vector.cpp, vector-ls.cpp, vector-fixed.cpp, vector-float.cpp, vector-mask.cpp, VecRegs.hpp): configuration, every load /
  // store form, the integer / fixed-point / mask / permutation chapters of V 1.0, and F/D arithmetic, fused forms, compares, sqrt,
  // estimates, conversions, widening arithmetic, reductions and slides (DESIGN.md rows a14-a16 list them), at the reference's
  // defaults (VLEN = 8 * vlenb, ELEN 64, every LMUL legal, agnostic = all ones, vill instead of a trap, non-zero vstart traps
  // on arithmetic).  One lane runs its hart's element loop serially (lanes-per-element mapping is the next step).
  // Anything else in the vector encodings stops the hart UNSUPPORTED (UNS_VECTOR | instruction << 20).
  WSP_HD u64 vtype() { return m.cold(CI_VTYPE); }
  WSP_HD static u32 gm_x8(u32 gm) { return gm < 4 ? 8u << gm : gm > 4 ? 8u >> (8 - gm) : 0; }                 // VecRegs.hpp:354-361
  WSP_HD static u64 ones_of(u32 esz) { return esz == 3 ? ~0ULL : (1ULL << (8u << esz)) - 1; }
  WSP_HD bool vec_on() const { return ext(ISA_V) && ((h.mstatus >> MST_VS_SH) & 3) != 0; }                    // isVecLegal
  WSP_HD void mark_vs_dirty() { if (((h.mstatus >> MST_VS_SH) & 3) != 3 && !dry_bail()) { h.mstatus |= (3ULL << MST_VS_SH) | MST_SD; s.chg |= CH_MSTATUS; } }
  WSP_HD u64 vrd(u32 reg, u32 ix, u32 esz) { return m.vld(reg * vlenb + (ix << esz), esz); }
  WSP_HD i64 vrd_s(u32 reg, u32 ix, u32 esz) { const u32 sh = 64 - (8u << esz); return (i64)(vrd(reg, ix, esz) << sh) >> sh; }
  // VecRegs::write (VecRegs.hpp:190-211): the first write fixes the group the commit record reports
  WSP_HD void vwr(u32 reg, u32 ix, u32 esz, u32 groupx8, u64 v)
  {
    if (s.vgroup == 0) { s.vgroup = groupx8 >= 8 ? groupx8 / 8 : 1; s.chg |= CH_VREG; }
    m.vst(reg * vlenb + (ix << esz), esz, v);
  }
  // a split loop may leave this lane without an element: the bookkeeping of "a register group was written" is done up front
  WSP_HD void v_note_write(u32 groupx8) { if (s.vgroup == 0) { s.vgroup = groupx8 >= 8 ? groupx8 / 8 : 1; s.chg |= CH_VREG; } }
  WSP_HD bool v_mask_bit(u32 ix) { return (m.vld(ix >> 3, 0) >> (ix & 7)) & 1; }                               // isActive(0, ix)
  struct VCfg { u32 esz, gx8, vl, elem_max; bool ta, ma, vill; };
  WSP_HD VCfg vcfg()
  {
    const u64 vt = vtype(); VCfg c;
    c.esz = (u32)(vt >> 3) & 7; c.gx8 = gm_x8((u32)vt & 7); c.ta = (vt >> 6) & 1; c.ma = (vt >> 7) & 1; c.vill = (vt >> 63) != 0;
    c.vl = (u32)m.cold(CI_VL);
    const u32 vlmax = c.gx8 * vlenb / (8u << c.esz), single = vlenb >> c.esz;
    c.elem_max = vlmax > single ? vlmax : single;                                                               // VecRegs.hpp:388
    return c;
  }
  // VecRegs::isDestActive (VecRegs.hpp:738-757)
  WSP_HD bool v_dest_active(const VCfg& c, u32 vd, u32 ix, u32 esz, bool masked, u64& val)
  {
    val = vrd(vd, ix, esz);
    if (ix >= c.vl) { if (c.ta) val = ones_of(esz); return false; }
    if (masked && !v_mask_bit(ix)) { if (c.ma) val = ones_of(esz); return false; }
    return true;
  }
  // Element loops with the element width (and where it pays the operation) as compile-time constants: the loop body is a
  // handful of instructions instead of a switch per element (tools/vbench.py: a vadd.vv at e32 / LMUL 8 took 0.5 us per
  // element of a sub-group lane).  esz_switch calls f with K<esz>; v_each is one pass over the destination elements of an
  // elementwise instruction, split over the sub-group (lane vsub takes elements vsub, vsub + vgrp, ...): body(ix) is the value
  // of an active element; tail and masked-off elements become all ones under the agnostic policies and otherwise stay as they
  // are - isDestActive (VecRegs.hpp:738-757) followed by a write of the value it returned.
#ifndef WSP_V_BATCH_ALL
#define WSP_V_BATCH_ALL 0                                // 1: the general path's copy batches too (host test builds of the batch loop)
#endif
  template <u32 V> struct K { static constexpr u32 v = V; };
  template <class F> WSP_HD static void esz_switch(u32 esz, F&& f)
  { switch (esz) { case 0: f(K<0>()); break; case 1: f(K<1>()); break; case 2: f(K<2>()); break; default: f(K<3>()); break; } }
  template <u32 E> WSP_HD u64 vrd_c(u32 reg, u32 ix) { return m.vld(reg * vlenb + (ix << E), E); }
  template <u32 E> WSP_HD void vst_c(u32 reg, u32 ix, u64 v) { m.vst(reg * vlenb + (ix << E), E, v); }
  template <u32 E, bool BATCH = false, class F> WSP_HD void v_each(const VCfg& c, u32 vd, bool masked, F&& body)
  {
    const u64 on = ones_of(E);
    const u32 vl = c.vl < c.elem_max ? c.vl : c.elem_max;
    if (!masked) {
      u32 ix = vsub;
      if (BATCH) {
        // small bodies, four elements at a time: all four are read and computed before any is written, so their LDS round trips
        // overlap (with one wave on the SIMD nothing else hides them; a plain unroll does not help - the compiler keeps every
        // read behind the write before it, which may alias).  Elementwise instructions: an element depends on its own index only.
        for (; ix + 3 * vgrp < vl; ix += 4 * vgrp) {
          const u64 r0 = body(ix), r1 = body(ix + vgrp), r2 = body(ix + 2 * vgrp), r3 = body(ix + 3 * vgrp);
          vst_c<E>(vd, ix, r0 & on); vst_c<E>(vd, ix + vgrp, r1 & on); vst_c<E>(vd, ix + 2 * vgrp, r2 & on); vst_c<E>(vd, ix + 3 * vgrp, r3 & on);
        }
      }
      for (; ix < vl; ix += vgrp) vst_c<E>(vd, ix, body(ix) & on);
    } else
      for (u32 ix = vsub; ix < vl; ix += vgrp) {
        if (v_mask_bit(ix)) vst_c<E>(vd, ix, body(ix) & on);
        else if (c.ma) vst_c<E>(vd, ix, on);
      }
    if (c.ta) for (u32 ix = vl + (vsub + vgrp - vl % vgrp) % vgrp; ix < c.elem_max; ix += vgrp) vst_c<E>(vd, ix, on);
  }
  // The split pass for loops whose destination elements are not SEW wide (widening / narrowing forms): dsz = log2 of the destination
  // element's bytes, elems = the destination's element count.  A destination group that overlaps a source group of ANOTHER element
  // width is legal in places (the serial loop reads an element before the writes so far can reach it; lanes writing apart would
  // not): `apart` false runs the pass unsplit, every lane of the sub-group doing all of it alike.
  WSP_HD static bool v_apart(u32 a, u32 ga8, u32 b, u32 gb8)
  { const u32 ga = ga8 >= 8 ? ga8 / 8 : 1, gb = gb8 >= 8 ? gb8 / 8 : 1; return a + ga <= b || b + gb <= a; }
  template <class F> WSP_HD void v_each_w(const VCfg& c, u32 vd, u32 dsz, u32 elems, bool masked, bool apart, F&& body)
  {
    VCfg c2 = c; c2.elem_max = elems;
    const u32 vsub0 = vsub, vgrp0 = vgrp;
    if (!apart) { vsub = 0; vgrp = 1; }
    esz_switch(dsz, [&](auto e) { constexpr u32 E = decltype(e)::v; v_each<E>(c2, vd, masked, body); });
    vsub = vsub0; vgrp = vgrp0;
    sub_fence();
  }
  WSP_HD void vec_fail(const Rec& r, u64 cur_pc) { illegal(r, cur_pc); if (s.vgroup) mark_vs_dirty(); }       // postVecFail (vector.cpp:672-680)
  WSP_HD void vec_success()                                                                                   // postVecSuccess (vector.cpp:637-669)
  {
    bool dirty = s.vgroup != 0;
    if (m.cold(CI_VSTART) != 0) { m.cold(CI_VSTART) = 0; s.chg |= CH_GEN; s.gen_csr = C_VSTART; dirty = true; }
    if (dirty) mark_vs_dirty();
  }
  WSP_HD void vec_unsupported(u32 inst) { stop(ST_STOP_UNSUPPORTED, (u64)UNS_VECTOR | ((u64)inst << 20)); }
  WSP_HD bool v_check_sew_lmul_vstart(const VCfg& c, const Rec& r, u64 cur_pc)                                // vector.cpp:170-194
  { if (!vec_on() || c.vill || m.cold(CI_VSTART) > 0) { vec_fail(r, cur_pc); return false; } return true; }
  WSP_HD bool v_check_int(const VCfg& c, const Rec& r, u64 cur_pc, bool vs1_is_vec)                           // checkVecIntInst (vector.cpp:133-167)
  {
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return false;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31;
    if (!((inst >> 25) & 1) && (vd == 0 || vs2 == 0 || (vs1_is_vec && vs1 == 0))) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  WSP_HD bool v_check_emul(const Rec& r, u64 cur_pc, u32 groupx8, u32 a, u32 b, u32 c3)                       // checkVecOpsVsEmul (vector.cpp:197-219)
  {
    const u32 mk_ = (groupx8 >= 8 ? groupx8 / 8 : 1) - 1;
    if ((a & mk_) || (b & mk_) || (c3 & mk_)) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  WSP_HD void v_set_vl_vtype(u32 rd, u64 elems, bool write_vl, u32 gm, u32 ew, bool ta, bool ma, bool vill)
  {
    if (write_vl) { m.cold(CI_VL) = elems; s.chg |= CH_VL; }
    wx(rd, m.cold(CI_VL));
    m.cold(CI_VTYPE) = (u64)gm | ((u64)ew << 3) | ((u64)ta << 6) | ((u64)ma << 7) | ((u64)vill << 63);
    s.chg |= CH_VTYPE;
  }
  // Hart::vsetvl (vector.cpp:530-634)
  WSP_HD void v_vsetvl(u32 rd, u32 rs1, u64 avl_reg, u64 vtv)
  {
    bool ma = (vtv >> 7) & 1, ta = (vtv >> 6) & 1;
    u32 gm = vtv & 7, ew = (vtv >> 3) & 7;
    bool vill = (vtv >> 63) != 0 || ew > 3 || (vtv >> 8) != 0;
    u64 elems = 0;
    if (gm == 4) vill = true;
    else {
      const u32 vlmax = (gm_x8(gm) * (vlenb * 8) / (8u << ew)) / 8;
      if (vlmax == 0) vill = true;
      else {
        if (rd != 0 && rs1 == 0) elems = vlmax;
        else if (rd == 0 && rs1 == 0) {
          const u64 vt = vtype(); const u32 prev = gm_x8((u32)vt & 7) * vlenb / (8u << ((u32)(vt >> 3) & 7));
          if (vlmax != prev && !vill) vill = true;                              // cannot change VLMAX: no legalisation at the defaults
          elems = m.cold(CI_VL);
        } else elems = avl_reg <= vlmax ? avl_reg : vlmax;
      }
      if (elems > vlmax) vill = true;
    }
    if (vill) { ma = ta = false; gm = ew = 0; elems = 0; }
    v_set_vl_vtype(rd, elems, vill || rd != 0 || rs1 != 0, gm, ew, ta, ma, vill);
    mark_vs_dirty();
  }
  WSP_HD void v_vsetivli(u32 rd, u32 avl, u32 imm)                                                            // vector.cpp:705-781
  {
    bool ma = (imm >> 7) & 1, ta = (imm >> 6) & 1;
    u32 gm = imm & 7, ew = (imm >> 3) & 7;
    bool vill = (imm >> 8) != 0 || ew > 3;
    u64 elems = avl;
    if (gm == 4) vill = true;
    else {
      const u32 vlmax = (gm_x8(gm) * (vlenb * 8) / (8u << ew)) / 8;
      if (vlmax == 0) vill = true; else if (elems > vlmax) elems = vlmax;
    }
    if (vill) { ma = ta = false; gm = ew = 0; elems = 0; }
    v_set_vl_vtype(rd, elems, true, gm, ew, ta, ma, vill);
    vec_success();
    mark_vs_dirty();
  }
  // unit-stride loads / stores (vector-ls.cpp:112-407)
  WSP_HD bool v_ldst_legal(const VCfg& c, u32 inst, u32 eew, u32& gx8)
  {
    gx8 = c.gx8 * (8u << eew) / (8u << c.esz);                                  // emul = lmul * eew / sew
    const bool sym = gx8 == 1 || gx8 == 2 || gx8 == 4 || gx8 == 8 || gx8 == 16 || gx8 == 32 || gx8 == 64;
    const u32 vd = (inst >> 7) & 31; const bool masked = !((inst >> 25) & 1);
    return sym && vec_on() && eew <= 3 && !c.vill && !(masked && vd == 0);      // isLegalVecLdSt (:32-62)
  }
  WSP_HD void v_ldst_fault(u32 cause, u32 ix, u64 tval, u64 cur_pc)             // :213-243, :395-403
  {
    if (dry_bail()) return;
    mark_vs_dirty();
    m.cold(CI_VSTART) = ix & (vlenb * 8 - 1); s.chg |= CH_GEN; s.gen_csr = C_VSTART;
    take_exception(cause, cur_pc, tval, false);
  }
  // vlm.v / vsm.v (vector-ls.cpp:484-559): the unit-stride byte loop over ceil(vl / 8) bytes at SEW 8, LMUL 1, tail agnostic
  WSP_HD bool v_mask_cfg(VCfg& c, const Rec& r, u64 cur_pc)
  {
    if (!vec_on() || c.vill || !((r.raw >> 25) & 1)) { vec_fail(r, cur_pc); return false; }
    c.vl = (c.vl + 7) / 8; c.esz = 0; c.gx8 = 8; c.ta = true; c.ma = false;
    return true;
  }
  // the address checks of a vector memory instruction have no side effects: bare translation (the TLB model is order dependent)
  WSP_HD bool v_ldst_pure() { return !(ext(ISA_S) && eff_priv() != PRV_M && (m.cold(CI_SATP) >> 60) != 0); }
  // [lo, lo + n) is ordinary memory for a vector load (need 1) / store (need 2) of esz-byte elements: inside the arena, element
  // aligned, no translation (the caller checked v_ldst_pure), one PMP verdict for every byte that allows the access, and - what
  // the per-element loops treat apart - clear of tohost (a load of it reads 0; a store near it is a host call) and, for stores,
  // of the code image (HF_CODE_DIRTY).  Then no element can fault and every element is a plain arena access.
  WSP_HD bool v_plain_range(u64 lo, u64 n, u32 esz, u32 need, bool store)
  {
    if ((lo & (esz - 1)) || n == 0 || !in_mem(lo, n)) return false;
    const u32 pm = eff_priv();
    u32 perm;
    if (!(h.flags & HF_PMP_ON) || (pm == PRV_M && !(h.flags & HF_PMP_MLOCK))) perm = pm == PRV_M ? 7u : 0u;
    else if (h.flags & HF_PMP_E0ALL) { const u32 cfg = (u32)m.cold(CI_PMPCFG0) & 0xff; perm = (pm == PRV_M && !(cfg & 0x80)) ? 7u : (cfg & 7u); }
    else return false;
    if (!(perm & need)) return false;
    const u64 hi = lo + n;                               // exclusive; no wrap: in_mem
    if (tohost != ~0ULL && lo < tohost + 16 && tohost < hi + 8) return false;
    if (store && lo < text_hi && text_lo < hi) return false;
    return true;
  }
  WSP_HD void v_load_unit(const Rec& r, u64 base, u64 stride_reg, u64 cur_pc, u32 eew, bool strided, bool mask_ls = false)
  {
    VCfg c = vcfg();
    if (mask_ls && !v_mask_cfg(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31; const bool masked = !((inst >> 25) & 1);
    u32 gx8;
    if (!v_ldst_legal(c, inst, eew, gx8)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, gx8, vd, 0, 0)) return;
    const u32 esz = 1u << eew, start = (u32)m.cold(CI_VSTART);
    const u32 single = vlenb >> eew, vlmax = c.gx8 * vlenb / (8u << c.esz), elem_max = vlmax > single ? vlmax : single;   // elemMax(eew)
    const u64 stride = strided ? stride_reg : esz;                              // vector-ls.cpp:939-1040: the same loop stepping by x[rs2]
    u64 addr = base + (u64)start * stride;
    if (gx8 < 8) gx8 = 8;
    const bool dry = DRY;
    if ((vgrp > 1 || dry) && v_ldst_pure()) {
      // lanes.  Pass 1 finds the first element that faults (the checks have no side effect without paging), pass 2 loads the
      // elements before it, then every lane takes that element's trap: what the serial loop leaves behind.
      if (start < c.vl) {
        if (!strided && v_plain_range(base + (u64)start * esz, (u64)(c.vl - start) * esz, esz, 1, false)) {
          // every element lies in ordinary memory the hart may read, aligned: no element can fault, none is special
          v_note_write(gx8);
          const u64 off0 = base - mem_base;
          // the whole 8-byte words of an unmasked load from a word-aligned base are copied as words (the register file is
          // byte-contiguous, the arena word-interleaved); the elements behind them one by one
          u32 s0 = start;
          if (!masked && start == 0 && !(off0 & 7)) {
            const u32 nw = (u32)(((u64)c.vl << eew) >> 3);
#pragma unroll 4
            for (u32 wd = vsub; wd < nw; wd += vgrp) m.vst(vd * vlenb + wd * 8, 3, m.ld64(off0 + (u64)wd * 8));
            s0 = (nw * 8) >> eew;
          }
          auto run = [&](auto e) { constexpr u32 E = decltype(e)::v;
            const u64 on = ones_of(E);
            for (u32 ix = s0 + vsub; ix < elem_max; ix += vgrp) {
              if (ix >= c.vl) { if (c.ta) vst_c<E>(vd, ix, on); }
              else if (masked && !v_mask_bit(ix)) { if (c.ma) vst_c<E>(vd, ix, on); }
              else vst_c<E>(vd, ix, mem_read(off0 + ((u64)ix << E), 1u << E));
            } };
          esz_switch(eew, run);
          sub_fence();
          vec_success();
          return;
        }
        u32 first = ~0u;
        for (u32 ix = start + vsub; ix < c.vl; ix += vgrp) {
          if (masked && !v_mask_bit(ix)) continue;
          u64 pa1, pa2, tval;
          if (ldst_addr(base + (u64)ix * stride, esz, 1, pa1, pa2, tval)) { first = ix; break; }
        }
        first = sub_min(first);
        if (dry && first != ~0u) { dry_bail(); return; }                          // a faulting access: the general path's business
        const u32 lim = first < elem_max ? first : elem_max;
        if (lim > start) v_note_write(gx8);
        for (u32 ix = start + vsub; ix < lim; ix += vgrp) {
          u64 elem;
          if (v_dest_active(c, vd, ix, eew, masked, elem)) {
            u64 pa1, pa2, tval;
            ldst_addr(base + (u64)ix * stride, esz, 1, pa1, pa2, tval);
            elem = pa1 == tohost ? 0 : mem_read2(pa1, pa2, esz);
          }
          vwr(vd, ix, eew, gx8, elem);
        }
        sub_fence();
        if (first != ~0u) {
          u64 pa1, pa2, tval = 0;
          const u32 cause = ldst_addr(base + (u64)first * stride, esz, 1, pa1, pa2, tval);
          v_ldst_fault(cause, first, tval, cur_pc); return;
        }
      }
      vec_success();
      return;
    }
    if (start < c.vl)
      for (u32 ix = start; ix < elem_max; ++ix, addr += stride) {
        u64 elem;
        if (!v_dest_active(c, vd, ix, eew, masked, elem)) { vwr(vd, ix, eew, gx8, elem); continue; }
        u64 pa1, pa2, tval;
        if (paging_unsupported()) return;
        const u32 cause = ldst_addr(addr, esz, 1, pa1, pa2, tval);
        if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); return; }
        elem = pa1 == tohost ? 0 : mem_read2(pa1, pa2, esz);
        vwr(vd, ix, eew, gx8, elem);
      }
    vec_success();
  }
  WSP_HD void v_store_unit(const Rec& r, u64 base, u64 stride_reg, u64 cur_pc, u32 eew, bool strided, bool mask_ls = false)
  {
    VCfg c = vcfg();
    if (mask_ls && !v_mask_cfg(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31; const bool masked = !((inst >> 25) & 1);
    u32 gx8;
    if (!v_ldst_legal(c, inst, eew, gx8)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, gx8, vd, 0, 0)) return;
    const u32 esz = 1u << eew, start = (u32)m.cold(CI_VSTART);
    const u64 stride = strided ? stride_reg : esz;
    u64 addr = base + (u64)start * stride;
    const u64 astride = (i64)stride < 0 ? 0 - stride : stride;                     // elements that overlap in memory are stored in order
    const bool dry = DRY;
    if (dry && !(v_ldst_pure() && astride >= esz)) { dry_bail(); return; }
    if ((vgrp > 1 || dry) && v_ldst_pure() && astride >= esz) {
      // lanes, as in v_load_unit; an element that lands in the code image or on tohost has side effects on hart state beyond
      // memory (v_store_elem): any such element sends the instruction through the serial loop below, on every lane alike.
      if (!strided && start < c.vl && v_plain_range(base + (u64)start * esz, (u64)(c.vl - start) * esz, esz, 2, true)) {
        const u64 off0 = base - mem_base;
        u32 s0 = start;
        if (!masked && start == 0 && !(off0 & 7)) {          // whole words as words, as in v_load_unit
          const u32 nw = (u32)(((u64)c.vl << eew) >> 3);
#pragma unroll 4
          for (u32 wd = vsub; wd < nw; wd += vgrp) m.st64(off0 + (u64)wd * 8, m.vld(vd * vlenb + wd * 8, 3));
          s0 = (nw * 8) >> eew;
        }
        auto run = [&](auto e) { constexpr u32 E = decltype(e)::v;
          for (u32 ix = s0 + vsub; ix < c.vl; ix += vgrp) {
            if (masked && !v_mask_bit(ix)) continue;
            mem_write(off0 + ((u64)ix << E), 1u << E, vrd_c<E>(vd, ix));
          } };
        esz_switch(eew, run);
        sub_fence();
        vec_success();
        return;
      }
      u32 first = ~0u, special = 0;
      for (u32 ix = start + vsub; ix < c.vl; ix += vgrp) {
        if (masked && !v_mask_bit(ix)) continue;
        u64 pa1, pa2, tval;
        if (ldst_addr(base + (u64)ix * stride, esz, 2, pa1, pa2, tval)) { first = ix; break; }
        special |= (pa1 + esz > text_lo && pa1 < text_hi) || (pa2 + esz > text_lo && pa2 < text_hi) ||
                   (tohost != ~0ULL && (pa1 - tohost < 8 || pa2 - tohost < 8 || tohost - pa1 < 8 || tohost - pa2 < 8));
      }
      first = sub_min(first);
      special = sub_or(special);
      if (dry && (special || first != ~0u)) { dry_bail(); return; }
      if (!special) {
        const u32 lim = first < c.vl ? first : c.vl;
        for (u32 ix = start + vsub; ix < lim; ix += vgrp) {
          if (masked && !v_mask_bit(ix)) continue;
          u64 pa1, pa2, tval;
          ldst_addr(base + (u64)ix * stride, esz, 2, pa1, pa2, tval);
          mem_write2(pa1, pa2, esz, vrd(vd, ix, eew));
        }
        sub_fence();
        if (first != ~0u) {
          u64 pa1, pa2, tval = 0;
          const u32 cause = ldst_addr(base + (u64)first * stride, esz, 2, pa1, pa2, tval);
          v_ldst_fault(cause, first, tval, cur_pc); return;
        }
        vec_success();
        return;
      }
    }
    for (u32 ix = start; ix < c.vl; ++ix, addr += stride) {
      if (masked && !v_mask_bit(ix)) continue;
      const u64 elem = vrd(vd, ix, eew);
      u64 pa1, pa2, tval;
      if (paging_unsupported()) return;
      const u32 cause = ldst_addr(addr, esz, 2, pa1, pa2, tval);
      if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); return; }
      if (!v_store_elem(pa1, pa2, esz, elem)) return;
    }
    vec_success();
  }
  // writeForStore (Hart.cpp:2727-2800); vector stores leave no m line in the log (printTrace.cpp:452).  False: the hart stopped
  WSP_HD bool v_store_elem(u64 pa1, u64 pa2, u32 esz, u64 elem)
  {
    if (pa1 + esz > text_lo && pa1 < text_hi) h.flags |= HF_CODE_DIRTY;
    if (pa2 != pa1 && pa2 + esz > text_lo && pa2 < text_hi) h.flags |= HF_CODE_DIRTY;
    if (tohost != ~0ULL && (pa1 == tohost || pa1 == tohost + 4)) {
      mem_write(pa1 - mem_base, esz, elem);
      u64 v = elem;
      if (esz == 4) { if (pa1 == tohost + 4) { h.htif_dev = (u32)elem; return true; } v |= (u64)h.htif_dev << 32; }
      else if (esz == 8) h.htif_dev = 0;
      if (((v >> 48) & 0xff) == 0 && ((v >> 56) & 0xff) == 0 && (elem & 1)) { stop(ST_STOP_TOHOST, v); return false; }
      return true;
    }
    mem_write2(pa1, pa2, esz, elem);
    return true;
  }
  // indexed loads / stores vluxei / vloxei / vsuxei / vsoxei (vector-ls.cpp:1268-1382, 1570-1720; checks :32-109, vector.cpp:301-320):
  // data EEW = SEW, index EEW from the encoding, offsets zero-extended
  WSP_HD void v_ldst_indexed(const Rec& r, u64 base, u64 cur_pc, u32 ieew, bool store, u32 fields)
  {
    // fields > 1: vluxseg / vloxseg / vsuxseg / vsoxseg (vector-ls.cpp:2519-2960): all or nothing per element, like the other segment forms
    const VCfg c = vcfg();
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vi = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 ogx8 = (8u << ieew) * c.gx8 / (8u << c.esz);
    const bool sym = ogx8 == 1 || ogx8 == 2 || ogx8 == 4 || ogx8 == 8 || ogx8 == 16 || ogx8 == 32 || ogx8 == 64;
    if (!vec_on() || !sym || c.gx8 * fields > 64 || c.vill) { vec_fail(r, cur_pc); return; }
    const u32 eg0 = c.gx8 >= 8 ? c.gx8 / 8 : 1, eg2 = ogx8 >= 8 ? ogx8 / 8 : 1, seg = eg0 * fields;
    bool ok = !(vd & (eg0 - 1)) && !(vi & (eg2 - 1)) && !(masked && (vd == 0 || vi == 0));
    if (fields > 1) {                                                           // checkVecLdStIndexedInst, segment case (:80-104)
      const bool apart = vi >= vd + seg || vd >= vi + eg2;
      ok = ok && (store ? (c.esz == ieew || apart) : apart) && vd + seg <= 32;
    }
    else if (store) ok = ok && (c.esz == ieew || vd >= vi + eg2 || vi >= vd + eg0);                           // checkSourceOverlap
    else ok = ok && v_overlap_ok(vd, 8u << c.esz, c.gx8, vi, 8u << ieew, ogx8);
    if (!ok) { vec_fail(r, cur_pc); return; }
    const u32 size = 1u << c.esz, start = (u32)m.cold(CI_VSTART), dg = (c.gx8 < 8 ? 8 : c.gx8) * fields;
    u64 pa1[8], pa2[8], val[8], tval;
    if (store) {
      for (u32 ix = start; ix < c.vl; ++ix) {
        if (masked && !v_mask_bit(ix)) continue;
        const u64 vaddr = base + vrd(vi, ix, ieew);
        for (u32 f = 0; f < fields; ++f) {
          if (paging_unsupported()) return;
          const u32 cause = ldst_addr(vaddr + (u64)f * size, size, 2, pa1[f], pa2[f], tval);
          if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); return; }
        }
        for (u32 f = 0; f < fields; ++f) if (!v_store_elem(pa1[f], pa2[f], size, vrd(vd + f * eg0, ix, c.esz))) return;
      }
      vec_success(); return;
    }
    if (start < c.vl)
      for (u32 ix = start; ix < c.elem_max; ++ix) {
        u64 elem;
        if (!v_dest_active(c, vd, ix, c.esz, masked, elem)) {
          for (u32 f = 0; f < fields; ++f) { v_dest_active(c, vd + f * eg0, ix, c.esz, masked, elem); vwr(vd + f * eg0, ix, c.esz, dg, elem); }
          continue;
        }
        const u64 vaddr = base + vrd(vi, ix, ieew);
        for (u32 f = 0; f < fields; ++f) {
          if (paging_unsupported()) return;
          const u32 cause = ldst_addr(vaddr + (u64)f * size, size, 1, pa1[f], pa2[f], tval);
          if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); return; }
          val[f] = pa1[f] == tohost ? 0 : mem_read2(pa1[f], pa2[f], size);
        }
        for (u32 f = 0; f < fields; ++f) vwr(vd + f * eg0, ix, c.esz, dg, val[f]);
      }
    vec_success();
  }
  // segment loads / stores, unit-stride and strided (vector-ls.cpp:1905-2070, 2153-2272), and the fault-only-first forms (:213-243,
  // :2016-2048).  A segment is all or nothing: every field is checked before any is written.  seg = false: vle<eew>ff.v (fields = 1)
  WSP_HD void v_ldst_seg(const Rec& r, u64 base, u64 cur_pc, u32 eew, u32 fields, u64 stride, bool store, bool ff, bool seg)
  {
    const VCfg c = vcfg();
    const u32 inst = r.raw, vd = (inst >> 7) & 31; const bool masked = !((inst >> 25) & 1);
    u32 gx8;
    if (!v_ldst_legal(c, inst, eew, gx8) || gx8 * fields > 64) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, gx8, vd, 0, 0)) return;
    const u32 esz = 1u << eew, start = (u32)m.cold(CI_VSTART);
    const u32 single = vlenb >> eew, vlmax = c.gx8 * vlenb / (8u << c.esz), elem_max = vlmax > single ? vlmax : single;
    if (gx8 < 8) gx8 = 8;
    const u32 group = gx8 / 8, lg = gx8 * fields;                               // the log names group * fields registers (Hart.cpp:5753-5771)
    if (vd + fields * group > 32) { vec_fail(r, cur_pc); return; }
    u64 addr = base + (u64)start * stride;
    u64 pa1[8], pa2[8], val[8];
    if (store) {
      for (u32 ix = start; ix < c.vl; ++ix, addr += stride) {
        if (masked && !v_mask_bit(ix)) continue;
        for (u32 f = 0; f < fields; ++f) {
          u64 tval;
          if (paging_unsupported()) return;
          const u32 cause = ldst_addr(addr + (u64)f * esz, esz, 2, pa1[f], pa2[f], tval);
          if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); return; }
        }
        for (u32 f = 0; f < fields; ++f) if (!v_store_elem(pa1[f], pa2[f], esz, vrd(vd + f * group, ix, eew))) return;
      }
      vec_success(); return;
    }
    if (start < c.vl)
      for (u32 ix = start; ix < elem_max; ++ix, addr += stride) {
        u64 d;
        if (!v_dest_active(c, vd, ix, eew, masked, d)) {
          for (u32 f = 0; f < fields; ++f) { v_dest_active(c, vd + f * group, ix, eew, masked, d); vwr(vd + f * group, ix, eew, lg, d); }
          continue;
        }
        for (u32 f = 0; f < fields; ++f) {
          u64 tval;
          if (paging_unsupported()) return;
          const u32 cause = ldst_addr(addr + (u64)f * esz, esz, 1, pa1[f], pa2[f], tval);
          if (cause) {
            if (ix == 0 || !ff) {
              if (!ff || seg) v_ldst_fault(cause, ix, tval, cur_pc); else { mark_vs_dirty(); take_exception(cause, cur_pc, tval, false); }
              return;
            }
            mark_vs_dirty();
            if (!seg || c.ta) {                                                 // trim vl; the segment form does so only when tail-agnostic
              m.cold(CI_VL) = ix; s.chg |= CH_VL;
              if (c.ta) for (u32 ti = ix; ti < elem_max; ++ti) for (u32 g = 0; g < fields; ++g) vwr(vd + g * group, ti, eew, lg, ones_of(eew));
            }
            vec_success(); return;
          }
          val[f] = pa1[f] == tohost ? 0 : mem_read2(pa1[f], pa2[f], esz);
        }
        for (u32 f = 0; f < fields; ++f) vwr(vd + f * group, ix, eew, lg, val[f]);
      }
    vec_success();
  }
  // vl<nf>re<eew>.v (vectorLoadWholeReg, vector-ls.cpp:565-670) and vs<nf>r.v (vectorStoreWholeReg, :747-821): nf whole registers
  // from vstart, independent of vtype and vl
  WSP_HD void v_ldst_whole(const Rec& r, u64 base, u64 cur_pc, u32 eew, bool store)
  {
    const u32 inst = r.raw, vd = (inst >> 7) & 31, nf = (inst >> 29) + 1;
    if ((nf & (nf - 1)) || !vec_on() || !((inst >> 25) & 1)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, nf * 8, vd, 0, 0)) return;
    if (store) eew = 0;
    const u32 esz = 1u << eew, count = vlenb * nf / esz, start = (u32)m.cold(CI_VSTART);
    u64 addr = base + (u64)start * esz;
    if (!store && start >= count) { vec_success(); return; }
    for (u32 ix = start; ix < count; ++ix, addr += esz) {
      u64 pa1, pa2, tval;
      if (paging_unsupported()) return;
      const u32 cause = ldst_addr(addr, esz, store ? 2 : 1, pa1, pa2, tval);
      if (cause) { v_ldst_fault(cause, ix, tval, cur_pc); if (!store) { s.vgroup = nf; s.chg |= CH_VREG; } return; }
      if (store) { if (!v_store_elem(pa1, pa2, 1, vrd(vd, ix, 0))) return; }
      else vwr(vd, ix, eew, nf * 8, pa1 == tohost ? 0 : mem_read2(pa1, pa2, esz));
    }
    if (!store) { s.vgroup = nf; s.chg |= CH_VREG; }                            // touchReg(vd, 8), then lastVecReg scales by the field count (Hart.cpp:5753-5771)
    vec_success();
  }
  // integer element operation: a = vs2 element, b = vs1 element / scalar / immediate
  enum { VI_ADD, VI_SUB, VI_RSUB, VI_AND, VI_OR, VI_XOR, VI_SLL, VI_SRL, VI_SRA, VI_MINU, VI_MIN, VI_MAXU, VI_MAX, VI_MUL,
         VI_DIVU, VI_DIV, VI_REMU, VI_REM, VI_MULHU, VI_MULHSU, VI_MULH };
  WSP_HD static u64 vi_op(u32 op, u64 a, u64 b, u32 esz)
  {
    const u32 bits = 8u << esz, sh = 64 - bits, shm = bits - 1;
    const i64 sa = (i64)(a << sh) >> sh, sb = (i64)(b << sh) >> sh;
    const u64 on = ones_of(esz), ua = a & on, ub = b & on; u64 z;
    switch (op) {
      case VI_ADD: z = a + b; break; case VI_SUB: z = a - b; break; case VI_RSUB: z = b - a; break;
      case VI_AND: z = a & b; break; case VI_OR: z = a | b; break; case VI_XOR: z = a ^ b; break;
      case VI_SLL: z = ua << (b & shm); break; case VI_SRL: z = ua >> (b & shm); break; case VI_SRA: z = (u64)(sa >> (b & shm)); break;
      case VI_MINU: z = ua < ub ? ua : ub; break; case VI_MIN: z = (u64)(sa < sb ? sa : sb); break;
      case VI_MAXU: z = ua > ub ? ua : ub; break; case VI_MAX: z = (u64)(sa > sb ? sa : sb); break;
      // vdivu / vdiv / vremu / vrem (vector.cpp:6171-6640): x / 0 = all ones, x % 0 = x, min / -1 = min, min % -1 = 0
      case VI_DIVU: z = ub ? ua / ub : ~0ull; break;
      case VI_REMU: z = ub ? ua % ub : ua; break;
      case VI_DIV: z = sb == 0 ? ~0ull : (sa == (i64)(~0ull << shm) && sb == -1) ? (u64)sa : (u64)(sa / sb); break;
      case VI_REM: z = sb == 0 ? (u64)sa : (sa == (i64)(~0ull << shm) && sb == -1) ? 0 : (u64)(sa % sb); break;
      // vmulhu / vmulhsu (signed vs2 x unsigned vs1) / vmulh (vector.cpp:4495-4870): narrow widths multiply in 64 bits
      case VI_MULHU: z = esz == 3 ? wsf::mul64x64(a, b).hi : (ua * ub) >> bits; break;
      case VI_MULHSU: if (esz == 3) { u64 hi = wsf::mul64x64(a, b).hi; if ((i64)a < 0) hi -= b; z = hi; } else z = (u64)((sa * (i64)ub) >> bits); break;
      case VI_MULH: if (esz == 3) { u64 hi = wsf::mul64x64(a, b).hi; if ((i64)a < 0) hi -= b; if ((i64)b < 0) hi -= a; z = hi; } else z = (u64)((sa * sb) >> bits); break;
      default: z = a * b; break;
    }
    return z & on;
  }
  // execVop_vv / vx / vi (vector.cpp:858-1111).  kind: 0 vv, 1 vx, 2 vi
  WSP_HD void v_int_arith(const Rec& r, u64 cur_pc, u32 op, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, kind == 0)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, kind == 0 ? vs1 : 0)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl) {
      v_note_write(dg);
      const bool vv = kind == 0;
      esz_switch(c.esz, [&](auto e) {                                           // lanes
        constexpr u32 E = decltype(e)::v;
#define WSP_VI_CASE(OP) case OP: v_each<E, DRY || WSP_V_BATCH_ALL>(c, vd, masked, [&](u32 ix) { return vi_op(OP, vrd_c<E>(vs2, ix), vv ? vrd_c<E>(vs1, ix) : scalar, E); }); break;
        switch (DRY ? op : ~0u) {                                               // (the general path's copy keeps the one generic loop)
          WSP_VI_CASE(VI_ADD) WSP_VI_CASE(VI_SUB) WSP_VI_CASE(VI_RSUB) WSP_VI_CASE(VI_AND) WSP_VI_CASE(VI_OR) WSP_VI_CASE(VI_XOR)
          WSP_VI_CASE(VI_SLL) WSP_VI_CASE(VI_SRL) WSP_VI_CASE(VI_SRA) WSP_VI_CASE(VI_MINU) WSP_VI_CASE(VI_MIN) WSP_VI_CASE(VI_MAXU)
          WSP_VI_CASE(VI_MAX) WSP_VI_CASE(VI_MUL)
          default: v_each<E>(c, vd, masked, [&](u32 ix) { return vi_op(op, vrd_c<E>(vs2, ix), vv ? vrd_c<E>(vs1, ix) : scalar, E); }); break;
        }
#undef WSP_VI_CASE
      });
      sub_fence();
    }
    vec_success();
  }
  // vmacc / vnmsac / vmadd / vnmsub (vector.cpp:4836-5290): k 0 macc, 1 nmsac, 2 madd, 3 nmsub
  WSP_HD void v_int_muladd(const Rec& r, u64 cur_pc, u32 k, bool vx, u64 xs)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, !vx)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, vx ? 0 : vs1)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl) {
      v_note_write(dg);
      esz_switch(c.esz, [&](auto e) {                                           // lanes
        constexpr u32 E = decltype(e)::v;
        v_each<E, DRY || WSP_V_BATCH_ALL>(c, vd, masked, [&](u32 ix) {
          const u64 e1 = vx ? xs : vrd_c<E>(vs1, ix), e2 = vrd_c<E>(vs2, ix), dd = vrd_c<E>(vd, ix);
          return k == 0 ? e1 * e2 + dd : k == 1 ? dd - e1 * e2 : k == 2 ? e1 * dd + e2 : e2 - e1 * dd;
        });
      });
      sub_fence();
    }
    vec_success();
  }
  // vmerge.vvm / vxm / vim, vmv.v.v / x / i (vector.cpp:7815-7970, 8098-8260)
  WSP_HD void v_merge_mv(const Rec& r, u64 cur_pc, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool vm = (inst >> 25) & 1;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (vm) {
      if (!v_check_emul(r, cur_pc, dg, vd, kind == 0 ? vs1 : 0, 0)) return;
      if (0 < c.vl) {
        v_note_write(dg);
        esz_switch(c.esz, [&](auto e) { constexpr u32 E = decltype(e)::v;       // lanes
          v_each<E, DRY || WSP_V_BATCH_ALL>(c, vd, false, [&](u32 ix) { return kind == 0 ? vrd_c<E>(vs1, ix) : scalar; }); });
        sub_fence();
      }
      if (s.vgroup == 0) { s.vgroup = dg / 8; s.chg |= CH_VREG; }               // touchReg (vector.cpp:8150)
      vec_success();
      return;
    }
    if (vd == 0 || vs2 == 0 || (kind == 0 && vs1 == 0)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, kind == 0 ? vs1 : 0)) return;
    if (0 < c.vl) {
      v_note_write(dg);
      esz_switch(c.esz, [&](auto e) { constexpr u32 E = decltype(e)::v;         // lanes (v0 is neither vd nor a source here)
        v_each<E, DRY || WSP_V_BATCH_ALL>(c, vd, false, [&](u32 ix) { return v_mask_bit(ix) ? (kind == 0 ? vrd_c<E>(vs1, ix) : scalar) : vrd_c<E>(vs2, ix); }); });
      sub_fence();
    }
    vec_success();
  }
  // reduction prologue shared by the integer and FP sums (checkRedOpVsEmul, vector.cpp:222-258)
  WSP_HD bool v_red_check(const VCfg& c, const Rec& r, u64 cur_pc)
  {
    const u32 inst = r.raw, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (masked && (vs2 == 0 || vs1 == 0)) { vec_fail(r, cur_pc); return false; }
    const u32 lm = c.gx8 >= 8 ? c.gx8 / 8 : 1;
    if (vs2 & (lm - 1)) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  WSP_HD void v_red_finish(const VCfg& c, u32 vd, u64 res)
  {
    vwr(vd, 0, c.esz, 8, res);
    if (c.ta) for (u32 ix = 1; ix < (vlenb >> c.esz); ++ix) vwr(vd, ix, c.esz, 8, ones_of(c.esz));
  }
  WSP_HD void v_int_reduce(const Rec& r, u64 cur_pc, u32 op)                    // vector.cpp:3690-3860
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    if (!v_red_check(c, r, cur_pc)) return;
    if (c.vl == 0) return;                                                      // no postVecSuccess (:3733-3734)
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    u64 res = vrd(vs1, 0, c.esz);
    if (vgrp > 1) {
      // lanes: sum / and / or / xor / min / max are associative and commutative on the wrapped element values, so every lane
      // folds the elements it owns and the partial results meet in a butterfly over the sub-group (all lanes end with the total)
      u64 part = 0; u32 have = 0;
      for (u32 ix = vsub; ix < c.vl; ix += vgrp) {
        if (masked && !v_mask_bit(ix)) continue;
        const u64 x = vrd(vs2, ix, c.esz);
        part = have ? vi_op(op, part, x, c.esz) : x; have = 1;
      }
      for (u32 st = 0; (vgrp >> st) > 1; ++st) {
        const u64 op2 = m.xlane_xchg64(part, st); const u32 h2 = (u32)m.xlane_xchg64(have, st);
        if (h2) { part = have ? vi_op(op, part, op2, c.esz) : op2; have = 1; }
      }
      if (have) res = vi_op(op, res, part, c.esz);
      sub_fence();                                                              // every lane has read vs1 / vs2 before vd changes
    } else
      for (u32 ix = 0; ix < c.vl; ++ix) { if (masked && !v_mask_bit(ix)) continue; res = vi_op(op, res, vrd(vs2, ix, c.esz), c.esz); }
    v_red_finish(c, vd, res);
    vec_success();
  }
  WSP_HD void v_mv_s(const VCfg& c, u32 vd, u64 val)                            // vmv.s.x / vfmv.s.f body (vector.cpp:8033-8095)
  {
    if (0 < c.vl) { vwr(vd, 0, c.esz, 8, val & ones_of(c.esz)); if (c.ta) for (u32 i = 1; i < (vlenb >> c.esz); ++i) vwr(vd, i, c.esz, 8, ones_of(c.esz)); }
    vec_success();
  }
  // checkDestSourceOverlap (vector.cpp:323-345)
  WSP_HD static bool v_overlap_ok(u32 dest, u32 dw, u32 dgx8, u32 src, u32 sw, u32 sgx8)
  {
    if (sw == dw) return true;
    const u32 sg = sgx8 >= 8 ? sgx8 / 8 : 1, dg = dgx8 >= 8 ? dgx8 / 8 : 1;
    if (src >= dest + dg || dest >= src + sg) return true;
    if (dw > sw) return sgx8 >= 8 && src == dest + dg - sg;
    return src == dest;
  }
  // checkVecOpsVsEmul with per-operand wide flags (vector.cpp:461-527): op0 is the destination; n = 2 or 3 operands
  WSP_HD bool v_check_ops_wide(const Rec& r, u64 cur_pc, u32 gx8, u32 esz, u32 n, u32 op0, bool w0, u32 op1, bool w1, u32 op2, bool w2, bool dest_src = false)
  {
    const u32 eg = gx8 >= 8 ? gx8 / 8 : 1, wg = 2 * gx8, ewg = wg >= 8 ? wg / 8 : 1, sew = 8u << esz;
    bool ok = !(op0 & ((w0 ? ewg : eg) - 1)) && !(op1 & ((w1 ? ewg : eg) - 1)) && (n < 3 || !(op2 & ((w2 ? ewg : eg) - 1)));
    const u32 g0 = w0 ? wg : gx8, g1 = w1 ? wg : gx8, g2 = w2 ? wg : gx8, s0 = w0 ? 2 * sew : sew, s1 = w1 ? 2 * sew : sew, s2 = w2 ? 2 * sew : sew;
    ok = ok && v_overlap_ok(op0, s0, g0, op1, s1, g1);
    if (n > 2) {
      ok = ok && v_overlap_ok(op0, s0, g0, op2, s2, g2);
      if (s1 != s2) { const u32 c1 = g1 >= 8 ? g1 / 8 : 1, c2 = g2 >= 8 ? g2 / 8 : 1; ok = ok && (op1 >= op2 + c2 || op2 >= op1 + c1); }   // checkSourceOverlap (:351-363)
    }
    if (dest_src) {                                                             // vd is read too: no overlap with a source of another EEW
      const u32 c0 = g0 >= 8 ? g0 / 8 : 1, c1 = g1 >= 8 ? g1 / 8 : 1, c2 = g2 >= 8 ? g2 / 8 : 1;
      if (s0 != s1) ok = ok && (op0 >= op1 + c1 || op1 >= op0 + c0);
      if (n > 2 && s0 != s2) ok = ok && (op0 >= op2 + c2 || op2 >= op0 + c0);
    }
    if (!ok) vec_fail(r, cur_pc);
    return ok;
  }
  // widening integer add / sub / mul (vector.cpp:1170-1560, 5270-5715).  op: 0 add 1 sub 2 mul; sa / sb: vs2 element / second operand signed
  WSP_HD void v_int_widen(const Rec& r, u64 cur_pc, u32 op, bool sa, bool sb, bool vx, u64 xs, bool wide_src)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, !vx)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 wg = c.gx8 * 2;
    if (wg > 64 || c.esz > 2) { vec_fail(r, cur_pc); return; }                 // isDoubleWideLegal (VecRegs.hpp:261-274)
    const u32 dsz = c.esz + 1, vlmax = c.gx8 * vlenb / (8u << c.esz), single = vlenb >> dsz, elems = vlmax > single ? vlmax : single;
    if (!v_check_ops_wide(r, cur_pc, c.gx8, c.esz, vx ? 2 : 3, vd, true, vs2, wide_src, vs1, false, op == 3)) return;
    const u32 dg = wg < 8 ? 8 : wg, sh = 64 - (8u << c.esz);
    if (0 < c.vl) {
      v_note_write(dg);
      const bool apart = (wide_src || v_apart(vd, wg, vs2, c.gx8)) && (vx || v_apart(vd, wg, vs1, c.gx8));
      v_each_w(c, vd, dsz, elems, masked, apart, [&](u32 ix) -> u64 {             // lanes
        const u64 a = wide_src ? vrd(vs2, ix, dsz) : sa ? (u64)vrd_s(vs2, ix, c.esz) : vrd(vs2, ix, c.esz);
        const u64 braw = vx ? (xs & ones_of(c.esz)) : vrd(vs1, ix, c.esz);
        const u64 b = sb ? (u64)((i64)(braw << sh) >> sh) : braw;
        return op == 0 ? a + b : op == 1 ? a - b : op == 2 ? a * b : vrd(vd, ix, dsz) + a * b;   // op 3: vwmacc family (vector.cpp:5686-6170)
      });
    }
    vec_success();
  }
  // narrowing shifts vnsrl / vnsra .wv / .wx / .wi (vector.cpp:3013-3300): vs2 is 2 * SEW wide, shift amount mod 2 * SEW
  WSP_HD void v_narrow_shift(const Rec& r, u64 cur_pc, bool arith, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, kind == 0)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (c.gx8 * 2 > 64 || c.esz > 2) { vec_fail(r, cur_pc); return; }
    if (!v_check_ops_wide(r, cur_pc, c.gx8, c.esz, kind == 0 ? 3 : 2, vd, false, vs2, true, vs1, false)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8, wbits = 16u << c.esz;
    if (0 < c.vl) {
      v_note_write(dg);
      v_each_w(c, vd, c.esz, c.elem_max, masked, v_apart(vd, c.gx8, vs2, c.gx8 * 2), [&](u32 ix) -> u64 {   // lanes
        const u32 amt = (u32)((kind == 0 ? vrd(vs1, ix, c.esz) : scalar) & (wbits - 1));
        return arith ? (u64)(vrd_s(vs2, ix, c.esz + 1) >> amt) : vrd(vs2, ix, c.esz + 1) >> amt;
      });
    }
    vec_success();
  }
  // ---- integer compares into a mask register (vector.cpp:1985-2800) and mask logicals (vector-mask.cpp:14-120)
  WSP_HD void v_wr_mask_bit(u32 reg, u32 ix, bool v)                             // VecRegs::writeMaskRegister (VecRegs.hpp:817-838)
  {
    const u32 o = reg * vlenb + (ix >> 3); const u64 byte = m.vld(o, 0), bit = 1u << (ix & 7);
    m.vst(o, 0, v ? (byte | bit) : (byte & ~bit));
    if (s.vgroup == 0) { s.vgroup = 1; s.chg |= CH_VREG; }
  }
  WSP_HD bool v_rd_mask_bit(u32 reg, u32 ix) { return (m.vld(reg * vlenb + (ix >> 3), 0) >> (ix & 7)) & 1; }
  enum { VC_EQ, VC_NE, VC_LTU, VC_LT, VC_LEU, VC_LE, VC_GTU, VC_GT };
  WSP_HD static bool vc_op(u32 op, u64 a, u64 b, u32 esz)
  {
    const u32 sh = 64 - (8u << esz);
    const i64 sa = (i64)(a << sh) >> sh, sb = (i64)(b << sh) >> sh;
    const u64 ua = a & ones_of(esz), ub = b & ones_of(esz);
    switch (op) {
      case VC_EQ: return ua == ub; case VC_NE: return ua != ub; case VC_LTU: return ua < ub; case VC_LT: return sa < sb;
      case VC_LEU: return ua <= ub; case VC_LE: return sa <= sb; case VC_GTU: return ua > ub; default: return sa > sb;
    }
  }
  // execVmseq_vv ... execVmsgt_vi behind checkVecMaskInst (vector.cpp:366-458).  kind: 0 vv 1 vx 2 vi
  WSP_HD bool v_check_mask_inst(const VCfg& c, const Rec& r, u64 cur_pc, bool two_src)
  {
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return false;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 eg = c.gx8 >= 8 ? c.gx8 / 8 : 1;
    bool ok = !(masked && (vs2 == 0 || (two_src && vs1 == 0)));
    ok = ok && !(vs2 < vd + 1 && vd < vs2 + eg && vd != vs2);                    // mask destination inside a source group
    ok = ok && !(two_src && vs1 < vd + 1 && vd < vs1 + eg && vd != vs1);
    ok = ok && !((vs2 | (two_src ? vs1 : 0)) & (eg - 1));
    if (!ok) vec_fail(r, cur_pc);
    return ok;
  }
  // The destination of a mask-producing compare, one BYTE (eight elements) at a time, the bytes split over the sub-group: lane
  // vsub builds bytes vsub, vsub + vgrp, ... - no two lanes touch one byte.  flag(ix) is the compare of an active element;
  // masked-off bits stay or become 1 (isMaskDestActive, VecRegs.hpp:762-780), bits from vl to the end of the register become 1
  // (finishMaskDest, VecRegs.hpp:639-646).  Only for a destination that is not also a source (the serial loop reads an element
  // before the bits written so far can reach it; lanes writing bytes of a source register would not).  vl > 0.
  template <class F> WSP_HD void v_mask_dest_bytes(const VCfg& c, u32 vd, bool masked, F&& flag)
  {
    for (u32 j = vsub; j < vlenb; j += vgrp) {
      const u32 o = vd * vlenb + j, i0 = 8 * j;
      u32 byte = 0xff;
      if (i0 < c.vl) {
        byte = (u32)m.vld(o, 0);
        const u32 mk = masked ? (u32)m.vld(j, 0) : 0xffu;                        // v0's bits for these elements
        for (u32 b = 0; b < 8; ++b) {
          const u32 ix = i0 + b, bit = 1u << b;
          if (ix >= c.vl) byte |= bit;
          else if (!(mk & bit)) { if (c.ma) byte |= bit; }
          else byte = flag(ix) ? (byte | bit) : (byte & ~bit);
        }
      }
      m.vst(o, 0, byte);
    }
    if (s.vgroup == 0) { s.vgroup = 1; s.chg |= CH_VREG; }
    sub_fence();
  }
  WSP_HD void v_int_compare(const Rec& r, u64 cur_pc, u32 op, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_mask_inst(c, r, cur_pc, kind == 0)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (0 < c.vl) {
      if (vd != vs2 && !(kind == 0 && vd == vs1)) {                             // lanes
        const bool vv = kind == 0;
        esz_switch(c.esz, [&](auto e) { constexpr u32 E = decltype(e)::v;
          v_mask_dest_bytes(c, vd, masked, [&](u32 ix) { return vc_op(op, vrd_c<E>(vs2, ix), vv ? vrd_c<E>(vs1, ix) : scalar, E); }); });
      } else {
        for (u32 ix = 0; ix < c.vl; ++ix) {                                     // isMaskDestActive (VecRegs.hpp:762-780)
          bool flag = v_rd_mask_bit(vd, ix);
          if (masked && !v_mask_bit(ix)) { if (c.ma) flag = true; }
          else flag = vc_op(op, vrd(vs2, ix, c.esz), kind == 0 ? vrd(vs1, ix, c.esz) : scalar, c.esz);
          v_wr_mask_bit(vd, ix, flag);
        }
        for (u32 ix = c.vl; ix < vlenb * 8; ++ix) v_wr_mask_bit(vd, ix, true);  // finishMaskDest (VecRegs.hpp:639-646)
      }
    }
    vec_success();
  }
  WSP_HD void v_mask_logical(const Rec& r, u64 cur_pc, u32 k)                   // vector-mask.cpp:14-120; k = funct6 - 0x18
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31;
    if (!((inst >> 25) & 1)) { vec_fail(r, cur_pc); return; }
    if (0 < c.vl) {
      for (u32 ix = 0; ix < c.vl; ++ix) {
        const bool a = v_rd_mask_bit(vs2, ix), b = v_rd_mask_bit(vs1, ix); bool z;
        switch (k) { case 0: z = a && !b; break; case 1: z = a && b; break; case 2: z = a || b; break; case 3: z = a != b; break;
                     case 4: z = a || !b; break; case 5: z = !(a && b); break; case 6: z = !(a || b); break; default: z = a == b; break; }
        v_wr_mask_bit(vd, ix, z);
      }
      for (u32 ix = c.vl; ix < vlenb * 8; ++ix) v_wr_mask_bit(vd, ix, true);
    }
    if (s.vgroup == 0) { s.vgroup = 1; s.chg |= CH_VREG; }                      // touchMask
    vec_success();
  }
  // vmv<nr>r.v (vmvr_v, vector.cpp:8261-8302)
  WSP_HD void v_mv_whole(const Rec& r, u64 cur_pc, u32 nr)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31;
    if (!((inst >> 25) & 1) || (vd & (nr - 1)) || (vs2 & (nr - 1))) { vec_fail(r, cur_pc); return; }
    if (vd != vs2) for (u32 i = 0; i < nr * vlenb / 8; ++i) m.vst(vd * vlenb + i * 8, 3, m.vld(vs2 * vlenb + i * 8, 3));
    s.vgroup = nr; s.chg |= CH_VREG;                                            // touchReg(vd, nr * 8)
    vec_success();
  }
  // vcpop.m / vfirst.m (vector-mask.cpp:128-187)
  WSP_HD void v_mask_scan_x(const Rec& r, u64 cur_pc, bool first)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, rd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    u64 res = first ? ~0ull : 0;
    for (u32 ix = 0; ix < c.vl; ++ix) {
      if (masked && !v_mask_bit(ix)) continue;
      if (!v_rd_mask_bit(vs2, ix)) continue;
      if (first) { res = ix; break; }
      ++res;
    }
    wx(rd, res);
    vec_success();
  }
  // vmsbf.m / vmsof.m / vmsif.m (vector-mask.cpp:190-338): k 1 before-first, 2 only-first, 3 including-first; bits past vl are kept
  WSP_HD void v_mask_set_first(const Rec& r, u64 cur_pc, u32 k)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (vd == vs2 || (masked && vd == 0)) { vec_fail(r, cur_pc); return; }
    if (0 < c.vl) {
      bool found = false;
      for (u32 ix = 0; ix < c.vl; ++ix) {
        bool flag = v_rd_mask_bit(vd, ix);
        if (masked && !v_mask_bit(ix)) { if (c.ma) flag = true; }
        else {
          const bool input = v_rd_mask_bit(vs2, ix);
          if (k == 1) { found = found || input; flag = !found; }
          else if (k == 3) { flag = !found; found = found || input; }
          else { flag = !found && input; found = found || input; }
        }
        v_wr_mask_bit(vd, ix, flag);
      }
    }
    vec_success();
  }
  // viota.m (vector-mask.cpp:341-430)
  WSP_HD void v_iota(const Rec& r, u64 cur_pc)
  {
    const VCfg c = vcfg();
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 group = c.gx8 <= 8 ? 1 : c.gx8 / 8;
    if (!vec_on() || c.vill || m.cold(CI_VSTART) > 0 || (vs2 >= vd && vs2 < vd + group) || (masked && vd == 0)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, c.gx8, vd, 0, 0)) return;
    u32 sum = 0;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        bool src = v_rd_mask_bit(vs2, ix); u64 d;
        if (v_dest_active(c, vd, ix, c.esz, masked, d)) d = sum & ones_of(c.esz); else src = false;
        vwr(vd, ix, c.esz, 8 * group, d);
        if (src) ++sum;
      }
    vec_success();
  }
  // vzext / vsext .vf2 .vf4 .vf8 (vector.cpp:6673-7200)
  WSP_HD void v_ext(const Rec& r, u64 cur_pc, u32 factor, bool sign)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, false)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 from = c.gx8 / factor, eg = c.gx8 >= 8 ? c.gx8 / 8 : 1, lf = factor == 2 ? 1 : factor == 4 ? 2 : 3;
    if (from == 0 || (vd % eg) || (eg > factor && (vs2 % (eg / factor))) || !v_overlap_ok(vd, 8u << c.esz, c.gx8, vs2, (8u << c.esz) / factor, from) ||
        c.esz < lf) { vec_fail(r, cur_pc); return; }
    const u32 ssz = c.esz - lf, dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl) {
      v_note_write(dg);
      v_each_w(c, vd, c.esz, c.elem_max, masked, v_apart(vd, c.gx8, vs2, from), [&](u32 ix) -> u64 {       // lanes
        return sign ? (u64)vrd_s(vs2, ix, ssz) : vrd(vs2, ix, ssz); });
    }
    vec_success();
  }
  // ---- permutation: vrgather .vv / .vx / .vi (vector.cpp:3302-3507), vslideup / vslidedown .vx / .vi, vslide1up / vslide1down .vx (:3968-4310)
  WSP_HD bool v_valid_index(u32 reg, u64 ix, u32 gx8, u32 esz) const           // VecRegs::isValidIndex (VecRegs.hpp:154-163)
  {
    const u64 file = 32ull * vlenb, size = 1ull << esz;
    if (ix >= file) return false;
    if (ix * size > (((u64)vlenb * gx8) >> 3) - size) return false;
    return (u64)reg * vlenb + ix * size <= file - size;
  }
  WSP_HD static bool v_overlap_any(u32 a, u32 b, u32 gx8) { const u32 g = gx8 >= 8 ? gx8 / 8 : 1; return b < a + g && a < b + g; }   // Hart.hpp:4846-4852
  WSP_HD void v_gather(const Rec& r, u64 cur_pc, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, kind == 0)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (v_overlap_any(vd, vs2, c.gx8) || (kind == 0 && v_overlap_any(vd, vs1, c.gx8))) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, kind == 0 ? vs1 : 0)) return;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        u64 d;
        if (v_dest_active(c, vd, ix, c.esz, masked, d)) {
          const u64 from = kind == 0 ? vrd(vs1, ix, c.esz) : scalar;
          d = 0;
          if ((kind != 0 || from < (u64)vlenb * 8) && v_valid_index(vs2, from, c.gx8, c.esz)) d = vrd(vs2, (u32)from, c.esz);
        }
        vwr(vd, ix, c.esz, dg, d);
      }
    vec_success();
  }
  WSP_HD void v_slide(const Rec& r, u64 cur_pc, bool up, bool one, u64 scalar, bool fp = false)    // fp: vfslide1up / vfslide1down (vector.cpp:4313-4475)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, false)) return;
    if (fp) { int rm; if (!v_check_fp(c, r, cur_pc, rm)) return; scalar = vf_scalar((r.raw >> 15) & 31, c.esz); }
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (up && v_overlap_any(vd, vs2, c.gx8)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, 0)) return;
    const u64 amount = one ? 1 : scalar;
    if (0 < c.vl) {
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        u64 d;
        const bool active = v_dest_active(c, vd, ix, c.esz, masked, d);
        if (up) {
          if (active && ix >= amount) d = vrd(vs2, (u32)(ix - amount), c.esz);
          if (ix >= c.vl || ix >= amount) vwr(vd, ix, c.esz, dg, d);
        } else {
          if (active) { d = 0; if (amount < 32ull * vlenb && v_valid_index(vs2, ix + amount, c.gx8, c.esz)) d = vrd(vs2, (u32)(ix + amount), c.esz); }
          vwr(vd, ix, c.esz, dg, d);
        }
      }
      if (one && up) { u64 d; if (v_dest_active(c, vd, 0, c.esz, masked, d)) d = scalar & ones_of(c.esz); vwr(vd, 0, c.esz, c.gx8, d); }
      if (one && !up) { const u32 slot = c.vl - 1; if (!masked || v_mask_bit(slot)) vwr(vd, slot, c.esz, c.gx8, scalar & ones_of(c.esz)); }
    }
    vec_success();
  }
  // vwredsumu.vs / vwredsum.vs (vector.cpp: vwredsum_vs; checkWideRedOpVsEmul :261-298)
  WSP_HD void v_int_wreduce(const Rec& r, u64 cur_pc, bool sign)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 lm = c.gx8 >= 8 ? c.gx8 / 8 : 1;
    if ((vs2 & (lm - 1)) || (masked && (vs2 == 0 || vs1 == 0)) || (vs1 >= vs2 && vs1 < vs2 + lm) || c.esz > 2) { vec_fail(r, cur_pc); return; }
    if (c.vl != 0) {
      const u32 dsz = c.esz + 1;
      u64 res = vrd(vs1, 0, dsz);
      for (u32 ix = 0; ix < c.vl; ++ix) { if (masked && !v_mask_bit(ix)) continue; res += sign ? (u64)vrd_s(vs2, ix, c.esz) : vrd(vs2, ix, c.esz); }
      vwr(vd, 0, dsz, 8, res & ones_of(dsz));
      if (c.ta) for (u32 ix = 1; ix < (vlenb >> dsz); ++ix) vwr(vd, ix, dsz, 8, ones_of(dsz));
    }
    vec_success();
  }
  // vcompress.vm (vector.cpp: vcompress_vm, execVcompress_vm)
  WSP_HD void v_compress(const Rec& r, u64 cur_pc)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vm_reg = (inst >> 15) & 31, vs2 = (inst >> 20) & 31;
    const u32 eg = c.gx8 >= 8 ? c.gx8 / 8 : 1;
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, 0)) return;
    if (v_overlap_any(vd, vs2, c.gx8) || (vm_reg >= vd && vm_reg < vd + eg) || !((inst >> 25) & 1)) { vec_fail(r, cur_pc); return; }
    if (0 < c.vl) {
      u32 di = 0;
      for (u32 ix = 0; ix < c.vl; ++ix) if (v_rd_mask_bit(vm_reg, ix)) vwr(vd, di++, c.esz, c.gx8, vrd(vs2, ix, c.esz));
      if (c.ta) for (u32 ix = di; ix < c.elem_max; ++ix) vwr(vd, ix, c.esz, c.gx8 < 8 ? 8 : c.gx8, ones_of(c.esz));
      if (s.vgroup == 0) { s.vgroup = eg; s.chg |= CH_VREG; }                   // touchReg(vd, group)
    }
    vec_success();
  }
  // vrgatherei16.vv (vector.cpp:3510-3645): 16-bit indices in vs1 with EMUL = 2 * LMUL / SEW-bytes
  WSP_HD void v_gather_ei16(const Rec& r, u64 cur_pc)
  {
    const VCfg c = vcfg();
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 ig8 = (2 * c.gx8) >> c.esz;
    const bool sym = ig8 == 1 || ig8 == 2 || ig8 == 4 || ig8 == 8 || ig8 == 16 || ig8 == 32 || ig8 == 64;
    if (!sym || (masked && (vd == 0 || vs1 == 0 || vs2 == 0))) { vec_fail(r, cur_pc); return; }
    const u32 eg = c.gx8 >= 8 ? c.gx8 / 8 : 1, ig = ig8 >= 8 ? ig8 / 8 : 1;
    if ((vd % eg) || (vs2 % eg) || (vs1 % ig)) { vec_fail(r, cur_pc); return; }
    const bool ov_src = c.esz != 1 && !(vs2 >= vs1 + ig || vs1 >= vs2 + eg);    // checkSourceOverlap(table, idx)
    if (v_overlap_any(vd, vs2, c.gx8) || (vs1 < vd + eg && vd < vs1 + ig) || ov_src) { vec_fail(r, cur_pc); return; }
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        u64 d;
        if (v_dest_active(c, vd, ix, c.esz, masked, d)) { const u64 from = vrd(vs1, ix, 1); d = v_valid_index(vs2, from, c.gx8, c.esz) ? vrd(vs2, (u32)from, c.esz) : 0; }
        vwr(vd, ix, c.esz, dg, d);
      }
    vec_success();
  }
  // ---- fixed point (vector-fixed.cpp): vsadd[u] vssub[u] (:28-640), roundoff (:645-680), vaadd[u] vasub[u] (:685-1070), vsmul (:1072-1230),
  // vssrl vssra (:1232-1480), vnclip[u] (:1483-1800).  64-bit arithmetic only: the SEW = 64 double-wide cases are split into halves.
  // rounding increment of roundoff(): vd / vd1 = bits d and d-1 of the value, low_nz = any of the bits below d-1 set
  WSP_HD static u32 fx_bit(u32 rm, u32 vd, u32 vd1, bool low_nz)
  { return rm == 0 ? vd1 : rm == 1 ? (vd1 & ((low_nz ? 1u : 0u) | vd)) : rm == 3 ? ((~vd & 1) & (vd1 | (low_nz ? 1u : 0u))) : 0; }
  WSP_HD static u64 fx_round_u64(u32 rm, u64 v, u32 d)                          // roundoff on an unsigned 64-bit value, d in [0, 63]
  {
    if (d == 0) return v;
    return (v >> d) + fx_bit(rm, (u32)(v >> d) & 1, (u32)(v >> (d - 1)) & 1, (v & ((1ull << (d - 1)) - 1)) != 0);
  }
  WSP_HD static i64 fx_round_s64(u32 rm, i64 v, u32 d)
  {
    if (d == 0) return v;
    return (v >> d) + (i64)fx_bit(rm, (u32)(v >> d) & 1, (u32)(v >> (d - 1)) & 1, ((u64)v & ((1ull << (d - 1)) - 1)) != 0);
  }
  enum { FX_SADDU, FX_SADD, FX_SSUBU, FX_SSUB, FX_AADDU, FX_AADD, FX_ASUBU, FX_ASUB, FX_SMUL, FX_SSRL, FX_SSRA, FX_NCLIPU, FX_NCLIP };
  WSP_HD u64 fx_op(u32 op, u32 rm, u64 a, u64 braw, u32 esz, bool& sat)         // a: vs2 element (raw bits), braw: second operand truncated to SEW
  {
    const u32 bits = 8u << esz, sh = 64 - bits;
    const u64 on = ones_of(esz), ua = a & on, ub = braw;
    const i64 sa = (i64)(a << sh) >> sh, sb = (i64)(braw << sh) >> sh;
    const i64 smax = (i64)(on >> 1), smin = -smax - 1;
    switch (op) {
      case FX_SADDU: { const u64 t = (ua + ub) & on; if (t < ua) { sat = true; return on; } return t; }
      case FX_SSUBU: if (ua < ub) { sat = true; return 0; } return ua - ub;
      case FX_SADD: { const u64 t = (u64)sa + (u64)sb; const i64 r = (i64)(t << sh) >> sh;
        if ((sa < 0) == (sb < 0) && (sa < 0) != (r < 0)) { sat = true; return (u64)(sa < 0 ? smin : smax) & on; } return (u64)r & on; }
      case FX_SSUB: { const u64 t = (u64)sa - (u64)sb; const i64 r = (i64)(t << sh) >> sh;
        if ((sa < 0) != (sb < 0) && (sa < 0) != (r < 0)) { sat = true; return (u64)(sa < 0 ? smin : smax) & on; } return (u64)r & on; }
      case FX_AADDU: case FX_ASUBU: case FX_AADD: case FX_ASUB: {
        if (esz < 3) {
          if (op == FX_AADDU) return fx_round_u64(rm, ua + ub, 1) & on;
          if (op == FX_ASUBU) return fx_round_u64(rm, ua - ub, 1) & on;       // wraps mod 2^64; bits 1..SEW are those of the double-wide difference
          return (u64)fx_round_s64(rm, op == FX_AADD ? sa + sb : sa - sb, 1) & on;
        }
        // SEW 64: a 65-bit sum / difference (top, lo)
        const bool sub = op == FX_ASUBU || op == FX_ASUB;
        const u64 lo = sub ? a - braw : a + braw;
        u64 top;
        if (op == FX_AADDU) top = lo < a;
        else if (op == FX_ASUBU) top = a < braw;
        else { const bool ovf = sub ? (((a ^ braw) & (a ^ lo)) >> 63) : ((~(a ^ braw) & (a ^ lo)) >> 63); top = ovf ? (a >> 63) : (lo >> 63); }
        return ((lo >> 1) | (top << 63)) + fx_bit(rm, (u32)(lo >> 1) & 1, (u32)lo & 1, false);
      }
      case FX_SMUL: {
        if (sa == smin && sb == smin) { sat = true; return (u64)smax; }
        if (esz < 3) return (u64)fx_round_s64(rm, sa * sb, bits - 1) & on;
        wsf::U128 p = wsf::mul64x64(a, braw); if ((i64)a < 0) p.hi -= braw; if ((i64)braw < 0) p.hi -= a;      // signed 128-bit product
        return ((p.hi << 1) | (p.lo >> 63)) + fx_bit(rm, (u32)(p.lo >> 63), (u32)(p.lo >> 62) & 1, (p.lo & ((1ull << 62) - 1)) != 0);
      }
      case FX_SSRL: return fx_round_u64(rm, ua, (u32)(braw & (bits - 1))) & on;
      default: return (u64)fx_round_s64(rm, sa, (u32)(braw & (bits - 1))) & on;    // FX_SSRA
    }
  }
  WSP_HD void v_fixed(const Rec& r, u64 cur_pc, u32 op, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    if (!v_check_int(c, r, cur_pc, kind == 0)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const bool narrow = op == FX_NCLIPU || op == FX_NCLIP;
    if (narrow) {
      if (c.gx8 * 2 > 64 || c.esz > 2) { vec_fail(r, cur_pc); return; }
      if (!v_check_ops_wide(r, cur_pc, c.gx8, c.esz, kind == 0 ? 3 : 2, vd, false, vs2, true, vs1, false)) return;
    } else if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, kind == 0 ? vs1 : 0)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8, rm = (u32)(m.cold(CI_VCSR) >> 1) & 3, bits = 8u << c.esz;
    const u64 on = ones_of(c.esz);
    bool sat = false;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        u64 d;
        if (v_dest_active(c, vd, ix, c.esz, masked, d)) {
          const u64 braw = (kind == 0 ? vrd(vs1, ix, c.esz) : scalar) & on;
          if (!narrow) d = fx_op(op, rm, vrd(vs2, ix, c.esz), braw, c.esz, sat);
          else {                                                                // source 2 * SEW <= 64 bits wide
            const u32 amt = (u32)(braw & (2 * bits - 1));
            if (op == FX_NCLIPU) { const u64 t = fx_round_u64(rm, vrd(vs2, ix, c.esz + 1), amt); if (t > on) { sat = true; d = on; } else d = t; }
            else {
              const i64 smax = (i64)(on >> 1), smin = -smax - 1, t = fx_round_s64(rm, vrd_s(vs2, ix, c.esz + 1), amt);
              if (t > smax) { sat = true; d = (u64)smax; } else if (t < smin) { sat = true; d = (u64)smin & on; } else d = (u64)t & on;
            }
          }
        }
        vwr(vd, ix, c.esz, dg, d);
      }
    if (sat) { m.cold(CI_VCSR) |= 1; s.chg |= CH_GEN; s.gen_csr = C_VXSAT; }    // csRegs_.write(VXSAT, 1): only vxsat is logged (CsRegs.cpp:4553-4570)
    vec_success();
  }
  // carry / borrow: vadc vsbc (carry-in from v0; the vm = 1 encodings are reserved) and vmadc vmsbc (carry-out into a mask register;
  // vm = 0 takes the carry-in from v0) (vector.cpp:7196-7815).  op: 0 vadc 1 vmadc 2 vsbc 3 vmsbc
  WSP_HD void v_carry(const Rec& r, u64 cur_pc, u32 op, u32 kind, u64 scalar)
  {
    const VCfg c = vcfg();
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u64 on = ones_of(c.esz);
    if (op == 0 || op == 2) {
      if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
      if (vd == 0 || vs2 == 0 || (kind == 0 && vs1 == 0) || !masked) { vec_fail(r, cur_pc); return; }
      if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, kind == 0 ? vs1 : 0)) return;
      const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
      if (0 < c.vl)
        for (u32 ix = 0; ix < c.elem_max; ++ix) {
          u64 d;
          if (v_dest_active(c, vd, ix, c.esz, false, d)) {
            const u64 a = vrd(vs2, ix, c.esz), b = kind == 0 ? vrd(vs1, ix, c.esz) : scalar, ci = v_mask_bit(ix) ? 1 : 0;
            d = (op == 0 ? a + b + ci : a - b - ci) & on;
          }
          vwr(vd, ix, c.esz, dg, d);
        }
      vec_success(); return;
    }
    if (masked && (vs2 == 0 || (kind == 0 && vs1 == 0))) { vec_fail(r, cur_pc); return; }
    if (!v_check_mask_inst(c, r, cur_pc, kind == 0)) return;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {                                 // up to elemMax: tail bits get ones
        bool flag = true;
        if (ix < c.vl) {
          const u64 a = vrd(vs2, ix, c.esz), b = (kind == 0 ? vrd(vs1, ix, c.esz) : scalar) & on;
          const bool cin = masked && v_mask_bit(ix);
          if (op == 1) { const u64 t = (a + b + (cin ? 1 : 0)) & on; flag = cin ? t <= a : t < a; }
          else flag = cin ? a <= b : a < b;
        }
        v_wr_mask_bit(vd, ix, flag);
      }
    vec_success();
  }
  // ---- FP (vector-float.cpp): checkFpSewLmulVstart (:398-500)
  WSP_HD bool v_fp_legal(u32 esz) const { return fp_on() && (esz == 2 ? ext(ISA_F) : esz == 3 ? ext(ISA_D) : false); }
  WSP_HD bool v_check_fp(const VCfg& c, const Rec& r, u64 cur_pc, int& rm)
  {
    if (!vec_on() || c.vill || m.cold(CI_VSTART) > 0) { vec_fail(r, cur_pc); return false; }
    if (!v_fp_legal(c.esz) || !fp_rm(7, rm)) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  WSP_HD u64 vf_scalar(u32 fr, u32 esz) { return esz == 2 ? (u64)rf_s(fr) : m.rf(fr); }                       // FpRegs::read<T>
  WSP_HD static u64 vf_op(u32 op, u64 a, u64 b, u32 esz, int rm, unsigned& fl)  // 0 add 1 sub 2 mul 3 div 4 rsub 5 rdiv
  {
    using namespace wsf;
    if (op == 4 || op == 5) { const u64 t = a; a = b; b = t; op = op == 4 ? 1 : 3; }
    if (rm == RM_RNE) {                                  // ordinary operands: the FP unit's result with the exact inexact flag (wsp_softfloat.h)
      u64 z = 0; bool hw;
      if (esz == 2) hw = op == 0 ? hw_add32(a, b, z, fl) : op == 1 ? hw_add32(a, b ^ F32::SIGN, z, fl) : op == 2 ? hw_mul32(a, b, z, fl) : hw_div32(a, b, z, fl);
      else hw = op == 0 ? hw_add64(a, b, z, fl) : op == 1 ? hw_add64(a, b ^ F64::SIGN, z, fl) : op == 2 ? hw_mul64(a, b, z, fl) : hw_div64(a, b, z, fl);
      if (hw) return z;
    }
    if (esz == 2) return op == 0 ? f_add<F32>(a, b, rm, fl) : op == 1 ? f_sub<F32>(a, b, rm, fl) : op == 2 ? f_mul<F32>(a, b, rm, fl) : f_div<F32>(a, b, rm, fl);
    return op == 0 ? f_add<F64>(a, b, rm, fl) : op == 1 ? f_sub<F64>(a, b, rm, fl) : op == 2 ? f_mul<F64>(a, b, rm, fl) : f_div<F64>(a, b, rm, fl);
  }
  // doFmin / doFmax (vector-float.cpp:97-172)
  template <class F> WSP_HD static u64 vf_minmax(bool mx, u64 a, u64 b, unsigned& fl)
  {
    using namespace wsf;
    const bool n1 = is_nan<F>(a), n2 = is_nan<F>(b);
    u64 z;
    if (n1 && n2) z = F::QNAN; else if (n1) z = b; else if (n2) z = a;
    else z = mx ? (lt_ord<F>(b, a) ? a : b) : (lt_ord<F>(a, b) ? a : b);
    if (is_snan<F>(a) || is_snan<F>(b)) fl |= FL_NV;
    else if (!n1 && !n2 && ((a ^ b) & F::SIGN) && eq_ord<F>(a, b)) z = mx ? (z & ~(u64)F::SIGN) : (z | F::SIGN);
    return z;
  }
  // vfadd.. (:655-1990) op 0-5; the eight multiply-add forms (:2128-3300) op 8 + (funct6 - 0x28); vfmin / vfmax (:7569-7860) op 16 / 17;
  // vfsgnj / vfsgnjn / vfsgnjx (:7863-8240) op 18-20
  WSP_HD void v_fp_arith(const Rec& r, u64 cur_pc, u32 op, bool vf)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_int(c, r, cur_pc, !vf)) return;
    if (!v_check_fp(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, vf ? 0 : vs1)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    const u64 sc = vf ? vf_scalar(vs1, c.esz) : 0, sign = c.esz == 2 ? 0x80000000ull : 0x8000000000000000ull;
    const bool w = c.esz == 2;
    unsigned fl = 0;
    if (0 < c.vl) {
      v_note_write(dg);
      auto run = [&](auto e) {                                                  // lanes
        constexpr u32 E = decltype(e)::v; constexpr bool w = E == 2;
        // a * b + c3 with one rounding: the FP unit where it applies (round to nearest even, ordinary operands)
        auto fma = [&](u64 a, u64 b, u64 c3) -> u64 {
          u64 z = 0;
          if (rm == RM_RNE && (w ? hw_fma32(a, b, c3, z, fl) : hw_fma64(a, b, c3, z, fl))) return z;
          return w ? f_fma<F32>(a, b, c3, rm, fl) : f_fma<F64>(a, b, c3, rm, fl);
        };
        v_each<E>(c, vd, masked, [&](u32 ix) -> u64 {
          u64 e1 = vf ? sc : vrd_c<E>(vs1, ix), e2 = vrd_c<E>(vs2, ix);
          if (op < 8) return vf_op(op, e2, e1, E, rm, fl);
          if (op < 16) {
            const u32 k = op - 8;
            u64 dd = vrd_c<E>(vd, ix);
            if (k & 1) e1 = w ? fneg<F32>(e1, fl) : fneg<F64>(e1, fl);
            if (k < 4) {                                                         // *madd / *msub: vd is a multiplicand, vs2 the addend
              if (k == 1 || k == 2) e2 = w ? fneg<F32>(e2, fl) : fneg<F64>(e2, fl);
              return fma(e1, dd, e2);
            }
            if (k == 5 || k == 6) dd = w ? fneg<F32>(dd, fl) : fneg<F64>(dd, fl);   // *macc / *msac: vd is the addend
            return fma(e1, e2, dd);
          }
          if (op < 18) return w ? vf_minmax<F32>(op == 17, e2, e1, fl) : vf_minmax<F64>(op == 17, e2, e1, fl);
          return (e2 & ~sign) | ((op == 18 ? e1 : op == 19 ? ~e1 : e2 ^ e1) & sign);
        });
      };
      if (w) run(K<2>()); else run(K<3>());
      sub_fence();
    }
    or_fflags(sub_or(fl));                                                      // updateAccruedFpBits: flags accumulate over all elements
    vec_success();
  }
  // vmfeq / vmfle / vmflt / vmfne .vv / .vf, vmfgt / vmfge .vf (:4219-4925) behind checkVecFpMaskInst (:335-395).  op: 0 eq 1 le 2 lt 3 ne 4 gt 5 ge
  WSP_HD void v_fp_compare(const Rec& r, u64 cur_pc, u32 op, bool vf)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_mask_inst(c, r, cur_pc, !vf)) return;
    if (!v_fp_legal(c.esz) || !fp_rm(7, rm)) { vec_fail(r, cur_pc); return; }
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u64 sc = vf ? vf_scalar(vs1, c.esz) : 0;
    const bool w = c.esz == 2;
    unsigned fl = 0;
    auto cmp = [&](u32 ix) -> bool {
      const u64 a = vrd(vs2, ix, c.esz), b = vf ? sc : vrd(vs1, ix, c.esz);
      const bool nan = w ? (is_nan<F32>(a) || is_nan<F32>(b)) : (is_nan<F64>(a) || is_nan<F64>(b));
      const bool sn = w ? (is_snan<F32>(a) || is_snan<F32>(b)) : (is_snan<F64>(a) || is_snan<F64>(b));
      if (nan) { if (sn || (op != 0 && op != 3)) fl |= FL_NV; return op == 3; }
      const bool lt = w ? lt_ord<F32>(a, b) : lt_ord<F64>(a, b), eq = w ? eq_ord<F32>(a, b) : eq_ord<F64>(a, b);
      return op == 0 ? eq : op == 1 ? (lt || eq) : op == 2 ? lt : op == 3 ? !eq : op == 4 ? (!lt && !eq) : !lt;
    };
    if (0 < c.vl) {
      if (vd != vs2 && !(!vf && vd == vs1)) { v_mask_dest_bytes(c, vd, masked, cmp); fl = sub_or(fl); }   // lanes
      else {
        for (u32 ix = 0; ix < c.vl; ++ix) {
          bool flag = v_rd_mask_bit(vd, ix);
          if (masked && !v_mask_bit(ix)) { if (c.ma) flag = true; }
          else flag = cmp(ix);
          v_wr_mask_bit(vd, ix, flag);
        }
        for (u32 ix = c.vl; ix < vlenb * 8; ++ix) v_wr_mask_bit(vd, ix, true);
      }
    }
    or_fflags(fl);
    vec_success();
  }
  // vfrsqrt7.v / vfrec7.v (vector-float.cpp:175-329): the 7-bit estimates of the V spec (its two 128-entry lookup tables), written over the
  // bit fields.  rec = false: 1 / sqrt(x), true: 1 / x
  WSP_HD static u64 v_est7(bool rec, u64 a, u32 esz, u32 rm, unsigned& fl)
  {
    using namespace wsf;
    const u8 rsq[128] = { 52, 51, 50, 48, 47, 46, 44, 43, 42, 41, 40, 39, 38, 36, 35, 34, 33, 32, 31, 30, 30, 29, 28, 27, 26, 25, 24, 23, 23, 22, 21, 20, 19, 19, 18, 17, 16, 16, 15, 14, 14, 13, 12, 12, 11, 10, 10, 9, 9, 8, 7, 7, 6, 6, 5, 4, 4, 3, 3, 2, 2, 1, 1, 0, 127, 125, 123, 121, 119, 118, 116, 114, 113, 111, 109, 108, 106, 105, 103, 102, 100, 99, 97, 96, 95, 93, 92, 91, 90, 88, 87, 86, 85, 84, 83, 82, 80, 79, 78, 77, 76, 75, 74, 73, 72, 71, 70, 70, 69, 68, 67, 66, 65, 64, 63, 63, 62, 61, 60, 59, 59, 58, 57, 56, 56, 55, 54, 53 };
    const u8 rcp[128] = { 127, 125, 123, 121, 119, 117, 116, 114, 112, 110, 109, 107, 105, 104, 102, 100, 99, 97, 96, 94, 93, 91, 90, 88, 87, 85, 84, 83, 81, 80, 79, 77, 76, 75, 74, 72, 71, 70, 69, 68, 66, 65, 64, 63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42, 41, 40, 40, 39, 38, 37, 36, 35, 35, 34, 33, 32, 31, 31, 30, 29, 28, 28, 27, 26, 25, 25, 24, 23, 23, 22, 21, 21, 20, 19, 19, 18, 17, 17, 16, 15, 15, 14, 14, 13, 12, 12, 11, 11, 10, 9, 9, 8, 8, 7, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 0 };
    const u32 MB = esz == 2 ? 23 : 52, EB = esz == 2 ? 8 : 11; const int B = (1 << (EB - 1)) - 1, maxE = (1 << EB) - 1;
    const u64 sign = 1ull << (MB + EB), mant = (1ull << MB) - 1, inf = (u64)maxE << MB, qnan = inf | (1ull << (MB - 1));
    const u64 sg = a & sign; u64 M = a & mant; const int E = (int)((a >> MB) & (u64)maxE);
    if (E == 0 && M == 0) { fl |= FL_DZ; return sg | inf; }
    if (E == maxE && M == 0 && (rec || !sg)) return rec ? sg : 0;
    if (E == maxE && M != 0) { if (!(M >> (MB - 1))) fl |= FL_NV; return qnan; }
    if (!rec && sg) { fl |= FL_NV; return qnan; }
    int in_exp = E;
    if (E == 0) { int sft = 0; while (!(M & (1ull << MB))) { M <<= 1; ++sft; } M &= mant; in_exp = 1 - sft; }   // frexp normalises subnormals
    if (!rec) {
      const u32 idx = ((u32)(in_exp & 1) << 6) | (u32)(M >> (MB - 6));
      const u64 out_exp = (u64)((3 * B - 1 - in_exp) / 2);
      return ((u64)rsq[idx] << (MB - 7)) | (out_exp << MB);
    }
    if (in_exp < -1 || in_exp > 2 * B) { fl |= FL_NX | FL_OF; return (rm == (sg ? 3u : 2u) || rm == 1) ? (sg | (inf - 1)) : (sg | inf); }
    int out_exp = 2 * B - 1 - in_exp;
    u64 out_sig = (u64)rcp[M >> (MB - 7)] << (MB - 7);
    if (out_exp < 1) { out_sig = ((1ull << MB) | out_sig) >> (1 - out_exp); out_exp = 0; }
    return sg | ((u64)out_exp << MB) | out_sig;
  }
  // vfsqrt.v (vector-float.cpp:3998-4066) and the same-width conversions (:4990-5310).  op: 0 sqrt 1 xu.f 2 x.f 3 f.xu 4 f.x 5 rtz.xu.f 6 rtz.x.f
  WSP_HD void v_fp_unary(const Rec& r, u64 cur_pc, u32 op)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_int(c, r, cur_pc, false)) return;
    if (!v_check_fp(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, 0)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    const bool w = c.esz == 2;
    if (op == 5 || op == 6) rm = RM_RTZ;
    unsigned fl = 0;
    if (0 < c.vl) {
      v_note_write(dg);
      auto run = [&](auto e) {                                                  // lanes
        constexpr u32 E = decltype(e)::v;
        v_each<E>(c, vd, masked, [&](u32 ix) -> u64 {
          const u64 a = vrd_c<E>(vs2, ix);
          if (op >= 7) return v_est7(op == 8, a, E, (u32)rm, fl);                  // 7 vfrsqrt7.v, 8 vfrec7.v
          if (op == 0) return w ? f_sqrt<F32>(a, rm, fl) : f_sqrt<F64>(a, rm, fl);
          if (op == 3 || op == 4) {
            if (rm == RM_RNE) return w ? hw_int_to_f32(a, op == 4, fl) : hw_int_to_f64(a, op == 4, true, fl);
            return w ? int_to_f<F32>(a, op == 4, false, rm, fl) : int_to_f<F64>(a, op == 4, true, rm, fl);
          }
          const bool sg = op == 2 || op == 6;
          return w ? f_to_int<F32>(a, rm, sg, false, fl) : f_to_int<F64>(a, rm, sg, true, fl);
        });
      };
      if (w) run(K<2>()); else run(K<3>());
      sub_fence();
    }
    or_fflags(sub_or(fl));
    vec_success();
  }
  // vfmerge.vfm (vector-float.cpp:4068-4130)
  WSP_HD void v_fp_merge(const Rec& r, u64 cur_pc)
  {
    const VCfg c = vcfg(); int rm;
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, fs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31;
    if (((inst >> 25) & 1) || vd == 0 || vs2 == 0 || !fp_rm(7, rm)) { vec_fail(r, cur_pc); return; }
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, 0)) return;
    if (!v_fp_legal(c.esz)) { vec_fail(r, cur_pc); return; }
    const u64 sc = vf_scalar(fs1, c.esz); const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl) {
      v_note_write(dg);
      esz_switch(c.esz, [&](auto e) { constexpr u32 E = decltype(e)::v; v_each<E>(c, vd, false, [&](u32 ix) { return v_mask_bit(ix) ? sc : vrd_c<E>(vs2, ix); }); });   // lanes
      sub_fence();
    }
    vec_success();
  }
  WSP_HD void v_fp_redsum(const Rec& r, u64 cur_pc, u32 op)                             // vfredusum / vfredosum (:6756-6990): sequential at the defaults
  {
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_fp(c, r, cur_pc, rm)) return;
    if (!v_red_check(c, r, cur_pc)) return;
    if (c.vl == 0) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    u64 res = vrd(vs1, 0, c.esz); unsigned fl = 0;
    for (u32 ix = 0; ix < c.vl; ++ix) { if (masked && !v_mask_bit(ix)) continue; const u64 x = vrd(vs2, ix, c.esz);
      res = op == 0 ? vf_op(0, res, x, c.esz, rm, fl) : c.esz == 2 ? vf_minmax<wsf::F32>(op == 2, res, x, fl) : vf_minmax<wsf::F64>(op == 2, res, x, fl); }   // op 1 vfredmin, 2 vfredmax (:6985-7170)
    v_red_finish(c, vd, res);
    or_fflags(fl);
    vec_success();
  }
  // ---- widening / narrowing FP at SEW 32 <-> 64 (vector-float.cpp: vfwadd / vfwsub .vv .vf .wv .wf :1000-1760, vfwmul :1930-2125, vfwmacc
  // family :3300-3990, vfwcvt.f.f.v :5730-5810, vfncvt.f.f.w :6520-6610, vfwredusum / vfwredosum :7170-7420, vfclass.v :4925-4990).
  // checkFpSewLmulVstart with wide = true (:398-500): SEW 32 needs F and D; SEW 16 would need Zvfh (not modelled -> illegal)
  WSP_HD bool v_check_fp_wide(const VCfg& c, const Rec& r, u64 cur_pc, int& rm)
  {
    if (!vec_on() || c.vill || m.cold(CI_VSTART) > 0) { vec_fail(r, cur_pc); return false; }
    if (!(c.esz == 2 && fp_on() && ext(ISA_F) && ext(ISA_D)) || !fp_rm(7, rm)) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  // op: 0 add 1 sub 2 mul, 8 + k multiply-add (k: 4 macc 5 nmacc 6 msac 7 nmsac)
  WSP_HD void v_fp_widen(const Rec& r, u64 cur_pc, u32 op, bool vf, bool wide_src)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_int(c, r, cur_pc, !vf)) return;
    if (!v_check_fp_wide(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 wg = c.gx8 * 2;
    if (wg > 64) { vec_fail(r, cur_pc); return; }
    if (!v_check_ops_wide(r, cur_pc, c.gx8, 2, vf ? 2 : 3, vd, true, vs2, wide_src, vs1, false, op >= 8)) return;
    const u32 vlmax = c.gx8 * vlenb / 32, single = vlenb >> 3, elems = vlmax > single ? vlmax : single, dg = wg < 8 ? 8 : wg;
    const u64 sc = vf ? vf_scalar(vs1, 2) : 0;
    unsigned fl = 0;
    if (0 < c.vl) {
      v_note_write(dg);
      const bool apart = (wide_src || v_apart(vd, wg, vs2, c.gx8)) && (vf || v_apart(vd, wg, vs1, c.gx8));
      v_each_w(c, vd, 3, elems, masked, apart, [&](u32 ix) -> u64 {             // lanes
        u64 e1 = f32_to_f64(vf ? sc : vrd(vs1, ix, 2), fl);
        const u64 e2 = wide_src ? vrd(vs2, ix, 3) : f32_to_f64(vrd(vs2, ix, 2), fl);
        if (op < 8) return vf_op(op, e2, e1, 3, rm, fl);
        const u32 k = op - 8; u64 dd = vrd(vd, ix, 3), z = 0;
        if (k & 1) e1 = fneg<F64>(e1, fl);
        if (k == 5 || k == 6) dd = fneg<F64>(dd, fl);
        if (rm == RM_RNE && hw_fma64(e1, e2, dd, z, fl)) return z;
        return f_fma<F64>(e1, e2, dd, rm, fl);
      });
      fl = sub_or(fl);
    }
    or_fflags(fl);
    vec_success();
  }
  WSP_HD void v_fp_cvt_ff(const Rec& r, u64 cur_pc, bool narrow, bool rod = false)   // vfwcvt.f.f.v / vfncvt.f.f.w / vfncvt.rod.f.f.w (:6608-6660)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_int(c, r, cur_pc, false)) return;
    if (rod) { if (!v_check_fp(c, r, cur_pc, rm)) return; if (c.esz != 2) { vec_fail(r, cur_pc); return; } rm = RM_ODD; }   // checkVecFpInst without the wide flag: F suffices
    else if (!v_check_fp_wide(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 wg = c.gx8 * 2;
    if (wg > 64) { vec_fail(r, cur_pc); return; }
    if (!v_check_ops_wide(r, cur_pc, c.gx8, 2, 2, vd, !narrow, vs2, narrow, 0, false)) return;
    const u32 dsz = narrow ? 2 : 3, vlmax = c.gx8 * vlenb / 32, single = vlenb >> dsz, elems = vlmax > single ? vlmax : single;
    const u32 dg8 = narrow ? c.gx8 : wg, dg = dg8 < 8 ? 8 : dg8;
    unsigned fl = 0;
    if (0 < c.vl) {
      v_note_write(dg);
      v_each_w(c, vd, dsz, elems, masked, v_apart(vd, dg8, vs2, narrow ? wg : c.gx8), [&](u32 ix) -> u64 {   // lanes
        return narrow ? f64_to_f32(vrd(vs2, ix, 3), rm, fl) : f32_to_f64(vrd(vs2, ix, 2), fl); });
      fl = sub_or(fl);
    }
    or_fflags(fl);
    vec_success();
  }
  WSP_HD void v_fp_wredsum(const Rec& r, u64 cur_pc)                            // vfwredusum / vfwredosum: sequential at the defaults
  {
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_fp_wide(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    const u32 lm = c.gx8 >= 8 ? c.gx8 / 8 : 1;
    if ((vs2 & (lm - 1)) || (masked && (vs2 == 0 || vs1 == 0)) || (vs1 >= vs2 && vs1 < vs2 + lm)) { vec_fail(r, cur_pc); return; }   // checkWideRedOpVsEmul
    unsigned fl = 0;
    if (c.vl != 0) {
      u64 res = vrd(vs1, 0, 3);
      for (u32 ix = 0; ix < c.vl; ++ix) { if (masked && !v_mask_bit(ix)) continue; res = vf_op(0, res, wsf::f32_to_f64(vrd(vs2, ix, 2), fl), 3, rm, fl); }
      vwr(vd, 0, 3, 8, res);
      if (c.ta) for (u32 ix = 1; ix < (vlenb >> 3); ++ix) vwr(vd, ix, 3, 8, ~0ull);
    }
    or_fflags(fl);
    vec_success();
  }
  WSP_HD void v_fp_class(const Rec& r, u64 cur_pc)
  {
    const VCfg c = vcfg(); int rm = 0;
    if (!v_check_int(c, r, cur_pc, false)) return;
    if (!v_check_fp(c, r, cur_pc, rm)) return;
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_emul(r, cur_pc, c.gx8, vd, vs2, 0)) return;
    const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
    if (0 < c.vl)
      for (u32 ix = 0; ix < c.elem_max; ++ix) {
        u64 d;
        if (v_dest_active(c, vd, ix, c.esz, masked, d)) d = c.esz == 2 ? wsf::fclass<wsf::F32>(vrd(vs2, ix, 2)) : wsf::fclass<wsf::F64>(vrd(vs2, ix, 3));
        vwr(vd, ix, c.esz, dg, d);
      }
    vec_success();
  }
  // widening / narrowing int <-> float conversions (vector-float.cpp:5330-5730, 5815-6520; fpConvertTo, float-util.hpp:225-400).
  // code = the vs1 field of VFUNARY0: 8 vfwcvt.xu.f 9 .x.f 0xa .f.xu 0xb .f.x 0xe .rtz.xu.f 0xf .rtz.x.f;
  // 0x10 vfncvt.xu.f.w 0x11 .x.f.w 0x12 .f.xu.w 0x13 .f.x.w 0x16 .rtz.xu.f.w 0x17 .rtz.x.f.w
  WSP_HD void v_fp_cvt_wn(const Rec& r, u64 cur_pc, u32 code)
  {
    using namespace wsf;
    const VCfg c = vcfg(); int rm = 0;
    const bool narrow = code >= 0x10, to_int = !(code == 0xa || code == 0xb || code == 0x12 || code == 0x13), sg = code & 1, rtz = (code & 7) >= 6;
    if (!v_check_int(c, r, cur_pc, false)) return;
    const bool fp_is_sew = (!narrow && to_int) || (narrow && !to_int);          // the float side has SEW bits: checkVecFpInst applies
    if (fp_is_sew) { if (!v_check_fp(c, r, cur_pc, rm)) return; }
    else if (!fp_rm(7, rm)) { vec_fail(r, cur_pc); return; }
    const u32 wg = c.gx8 * 2;
    if (wg > 64 || c.esz > 2) { vec_fail(r, cur_pc); return; }
    const u32 inst = r.raw, vd = (inst >> 7) & 31, vs2 = (inst >> 20) & 31; const bool masked = !((inst >> 25) & 1);
    if (!v_check_ops_wide(r, cur_pc, c.gx8, c.esz, 2, vd, !narrow, vs2, narrow, 0, false)) return;
    if (!fp_is_sew && (c.esz == 0 || !v_fp_legal(c.esz + 1))) { vec_fail(r, cur_pc); return; }   // the float side is 2 * SEW wide
    const u32 dsz = narrow ? c.esz : c.esz + 1, ssz = narrow ? c.esz + 1 : c.esz;
    const u32 vlmax = c.gx8 * vlenb / (8u << c.esz), single = vlenb >> dsz, elems = vlmax > single ? vlmax : single;
    const u32 dg8 = narrow ? c.gx8 : wg, dg = dg8 < 8 ? 8 : dg8;
    if (rtz) rm = RM_RTZ;
    unsigned fl = 0;
    if (0 < c.vl) {
      v_note_write(dg);
      v_each_w(c, vd, dsz, elems, masked, v_apart(vd, dg8, vs2, narrow ? wg : c.gx8), [&](u32 ix) -> u64 {   // lanes
        const u64 a = vrd(vs2, ix, ssz); u64 d;
        if (to_int) {
          if (ssz == 2 && dsz == 3) d = f_to_int<F32>(a, rm, sg, true, fl);
          else if (ssz == 3) d = f_to_int<F64>(a, rm, sg, false, fl) & 0xffffffffull;
          else {                                                                // f32 -> 16 bits through the 32-bit conversion (float-util.hpp:309-327)
            unsigned f2 = 0; const u64 v = f_to_int<F32>(a, rm, sg, false, f2);
            if (sg) { i64 t = (i64)(i32)v; if (t > 32767) { t = 32767; f2 = FL_NV; } else if (t < -32768) { t = -32768; f2 = FL_NV; } d = (u64)t & 0xffff; }
            else { u64 t = v & 0xffffffffull; if (t > 65535) { t = 65535; f2 = FL_NV; } d = t; }
            fl |= f2;
          }
        } else {
          if (ssz == 1) d = int_to_f<F32>(sg ? (u64)(i64)(int16_t)a : (a & 0xffff), sg, false, rm, fl);
          else if (ssz == 2) d = rm == RM_RNE ? hw_int_to_f64(a, sg, false, fl) : int_to_f<F64>(a, sg, false, rm, fl);
          else d = int_to_f<F32>(a, sg, true, rm, fl);
        }
        return d;
      });
      fl = sub_or(fl);
    }
    or_fflags(fl);
    vec_success();
  }
  WSP_HD bool v_fp_mv_check(const VCfg& c, const Rec& r, u64 cur_pc)            // vfmv.* prologue (:503-515, 585-595, 4169-4180)
  {
    int rm;
    if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return false;
    if (!((r.raw >> 25) & 1) || !fp_rm(7, rm)) { vec_fail(r, cur_pc); return false; }
    return true;
  }
  // dispatch: OP-V.  a = x[rs1], b = x[rs2]
  WSP_HD void exec_vector_body(const Rec& r, u64 a, u64 b, u64 cur_pc)
  {
    const u32 inst = r.raw, f3 = (inst >> 12) & 7, f6 = inst >> 26, rd = (inst >> 7) & 31, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31;
    const bool vm = (inst >> 25) & 1;
    s.vreg = rd;
    if (f3 == 7) {                                                              // vsetvli / vsetivli / vsetvl
      const bool is_vli = !(inst >> 31), is_ivli = (inst >> 30) == 3, is_vl = (inst >> 25) == 0x40;
      if (!is_vli && !is_ivli && !is_vl) { illegal(r, cur_pc); return; }
      if (!vec_on()) { vec_fail(r, cur_pc); return; }
      if (is_ivli) { v_vsetivli(rd, rs1, (inst >> 20) & 0x3ff); return; }
      v_vsetvl(rd, rs1, a, is_vli ? (u64)((inst >> 20) & 0x7ff) : b);
      vec_success();
      return;
    }
    const u64 imm = (u64)(i64)((i32)(rs1 << 27) >> 27), uimm = rs1;
    if (f3 == 0 || f3 == 3 || f3 == 4) {                                        // OPIVV / OPIVI / OPIVX
      const u32 kind = f3 == 0 ? 0 : f3 == 4 ? 1 : 2;
      const u64 sc = kind == 1 ? a : imm;
      int op = -1;
      switch (f6) {
        case 0x00: op = VI_ADD; break;
        case 0x02: if (kind != 2) op = VI_SUB; break;
        case 0x03: if (kind != 0) op = VI_RSUB; break;
        case 0x04: if (kind != 2) op = VI_MINU; break;
        case 0x05: if (kind != 2) op = VI_MIN; break;
        case 0x06: if (kind != 2) op = VI_MAXU; break;
        case 0x07: if (kind != 2) op = VI_MAX; break;
        case 0x09: op = VI_AND; break; case 0x0a: op = VI_OR; break; case 0x0b: op = VI_XOR; break;
        case 0x25: v_int_arith(r, cur_pc, VI_SLL, kind, kind == 2 ? uimm : sc); return;
        case 0x27:                                                              // vsmul.vv / .vx; .vi encodes vmv<nr>r.v (Decoder.cpp:633-638)
          if (kind != 2) { v_fixed(r, cur_pc, FX_SMUL, kind, sc); return; }
          if (rs1 != 0 && rs1 != 1 && rs1 != 3 && rs1 != 7) { illegal(r, cur_pc); return; }
          v_mv_whole(r, cur_pc, rs1 + 1); return;
        case 0x28: v_int_arith(r, cur_pc, VI_SRL, kind, kind == 2 ? uimm : sc); return;
        case 0x29: v_int_arith(r, cur_pc, VI_SRA, kind, kind == 2 ? uimm : sc); return;
        case 0x10: case 0x11: v_carry(r, cur_pc, f6 - 0x10, kind, sc); return;                                   // vadc / vmadc
        case 0x12: case 0x13: if (kind == 2) break; v_carry(r, cur_pc, f6 - 0x10, kind, sc); return;             // vsbc / vmsbc
        case 0x20: v_fixed(r, cur_pc, FX_SADDU, kind, sc); return;
        case 0x21: v_fixed(r, cur_pc, FX_SADD, kind, sc); return;
        case 0x22: if (kind == 2) break; v_fixed(r, cur_pc, FX_SSUBU, kind, sc); return;
        case 0x23: if (kind == 2) break; v_fixed(r, cur_pc, FX_SSUB, kind, sc); return;
        case 0x2a: v_fixed(r, cur_pc, FX_SSRL, kind, kind == 2 ? uimm : sc); return;
        case 0x2b: v_fixed(r, cur_pc, FX_SSRA, kind, kind == 2 ? uimm : sc); return;
        case 0x2e: v_fixed(r, cur_pc, FX_NCLIPU, kind, kind == 2 ? uimm : sc); return;
        case 0x2f: v_fixed(r, cur_pc, FX_NCLIP, kind, kind == 2 ? uimm : sc); return;
        case 0x2c: v_narrow_shift(r, cur_pc, false, kind, kind == 2 ? uimm : sc); return;
        case 0x2d: v_narrow_shift(r, cur_pc, true, kind, kind == 2 ? uimm : sc); return;
        case 0x0c: v_gather(r, cur_pc, kind, kind == 2 ? uimm : a); return;
        case 0x0e: if (kind == 0) { v_gather_ei16(r, cur_pc); return; } v_slide(r, cur_pc, true, false, kind == 2 ? uimm : a); return;
        case 0x30: case 0x31: if (kind != 0) break; v_int_wreduce(r, cur_pc, f6 & 1); return;                  // vwredsumu.vs / vwredsum.vs
        case 0x0f: if (kind == 0) break; v_slide(r, cur_pc, false, false, kind == 2 ? uimm : a); return;
        case 0x17:
          if (vm && rs2 != 0) { illegal(r, cur_pc); return; }                   // Decoder.cpp:336-343
          v_merge_mv(r, cur_pc, kind, sc); return;
        case 0x18: case 0x19: case 0x1c: case 0x1d: v_int_compare(r, cur_pc, f6 == 0x18 ? VC_EQ : f6 == 0x19 ? VC_NE : f6 == 0x1c ? VC_LEU : VC_LE, kind, sc); return;
        case 0x1a: case 0x1b: if (kind == 2) break; v_int_compare(r, cur_pc, f6 == 0x1a ? VC_LTU : VC_LT, kind, sc); return;
        case 0x1e: case 0x1f: if (kind == 0) break; v_int_compare(r, cur_pc, f6 == 0x1e ? VC_GTU : VC_GT, kind, sc); return;
        default: break;
      }
      if (op >= 0) { v_int_arith(r, cur_pc, (u32)op, kind, sc); return; }
      vec_unsupported(inst); return;
    }
    if (f3 == 2 || f3 == 6) {                                                   // OPMVV / OPMVX
      const bool vx = f3 == 6;
      if (f6 <= 7 && !vx) { const u8 rop[8] = { VI_ADD, VI_AND, VI_OR, VI_XOR, VI_MINU, VI_MIN, VI_MAXU, VI_MAX }; v_int_reduce(r, cur_pc, rop[f6]); return; }
      if (f6 >= 0x08 && f6 <= 0x0b) { v_fixed(r, cur_pc, FX_AADDU + (f6 - 0x08), vx ? 1 : 0, a); return; }    // vaaddu vaadd vasubu vasub
      if (f6 == 0x17 && !vx) { v_compress(r, cur_pc); return; }
      if ((f6 == 0x0e || f6 == 0x0f) && vx) { v_slide(r, cur_pc, f6 == 0x0e, true, a); return; }             // vslide1up.vx / vslide1down.vx
      if (f6 == 0x10) {
        const VCfg c = vcfg();
        if (!vx) {
          if (rs1 == 0x10 || rs1 == 0x11) { v_mask_scan_x(r, cur_pc, rs1 == 0x11); return; }
          if (rs1 != 0) { illegal(r, cur_pc); return; }                         // Decoder.cpp:512-516
          if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;                   // vmv.x.s (vector.cpp:7972-8030)
          if (!vm) { vec_fail(r, cur_pc); return; }
          wx(rd, (u64)vrd_s(rs2, 0, c.esz)); vec_success(); return;
        }
        if (rs2 != 0) { illegal(r, cur_pc); return; }                           // vmv.s.x (Decoder.cpp:735)
        if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
        if (!vm) { vec_fail(r, cur_pc); return; }
        v_mv_s(c, rd, a); return;
      }
      if (f6 == 0x12 && !vx && rs1 >= 2 && rs1 <= 7) { v_ext(r, cur_pc, rs1 < 4 ? 8 : rs1 < 6 ? 4 : 2, rs1 & 1); return; }   // Decoder.cpp:521-527
      if (f6 == 0x14 && !vx && rs1 != 0x11) {                                   // Decoder.cpp:538-544
        if (rs1 >= 1 && rs1 <= 3) { v_mask_set_first(r, cur_pc, rs1); return; }
        if (rs1 == 0x10) { v_iota(r, cur_pc); return; }
        illegal(r, cur_pc); return;
      }
      if (f6 == 0x14 && !vx && rs1 == 0x11) {                                   // vid.v (vector-mask.cpp:435-510)
        const VCfg c = vcfg();
        if (!v_check_sew_lmul_vstart(c, r, cur_pc)) return;
        if ((!vm && rd == 0) || rs2 != 0) { vec_fail(r, cur_pc); return; }
        if (!v_check_emul(r, cur_pc, c.gx8, rd, 0, 0)) return;
        const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
        if (0 < c.vl) {
          v_note_write(dg);
          esz_switch(c.esz, [&](auto e) { constexpr u32 E = decltype(e)::v; v_each<E>(c, rd, !vm, [&](u32 ix) { return (u64)ix; }); });   // lanes
          sub_fence();
        }
        vec_success(); return;
      }
      if (f6 >= 0x18 && f6 <= 0x1f && !vx) { v_mask_logical(r, cur_pc, f6 - 0x18); return; }
      if (f6 >= 0x20 && f6 <= 0x27) {
        const u8 mop[8] = { VI_DIVU, VI_DIV, VI_REMU, VI_REM, VI_MULHU, VI_MUL, VI_MULHSU, VI_MULH };
        v_int_arith(r, cur_pc, mop[f6 - 0x20], vx ? 1 : 0, a); return;
      }
      if (f6 >= 0x30 && f6 <= 0x37) { v_int_widen(r, cur_pc, f6 >> 1 & 1, f6 & 1, f6 & 1, vx, a, f6 >= 0x34); return; }   // vwaddu vwadd vwsubu vwsub, then the .w forms
      if (f6 == 0x38) { v_int_widen(r, cur_pc, 2, false, false, vx, a, false); return; }                       // vwmulu
      if (f6 == 0x3a) { v_int_widen(r, cur_pc, 2, true, false, vx, a, false); return; }                        // vwmulsu
      if (f6 == 0x3b) { v_int_widen(r, cur_pc, 2, true, true, vx, a, false); return; }                         // vwmul
      if (f6 == 0x3c) { v_int_widen(r, cur_pc, 3, false, false, vx, a, false); return; }                       // vwmaccu
      if (f6 == 0x3d) { v_int_widen(r, cur_pc, 3, true, true, vx, a, false); return; }                         // vwmacc
      if (f6 == 0x3e && vx) { v_int_widen(r, cur_pc, 3, true, false, true, a, false); return; }                // vwmaccus.vx
      if (f6 == 0x3f) { v_int_widen(r, cur_pc, 3, false, true, vx, a, false); return; }                        // vwmaccsu
      if (f6 == 0x2d) { v_int_muladd(r, cur_pc, 0, vx, a); return; }
      if (f6 == 0x2f) { v_int_muladd(r, cur_pc, 1, vx, a); return; }
      if (f6 == 0x29) { v_int_muladd(r, cur_pc, 2, vx, a); return; }
      if (f6 == 0x2b) { v_int_muladd(r, cur_pc, 3, vx, a); return; }
      vec_unsupported(inst); return;
    }
    const bool vf = f3 == 5;                                                    // OPFVV (1) / OPFVF (5)
    switch (f6) {
      case 0x00: v_fp_arith(r, cur_pc, 0, vf); return;
      case 0x02: v_fp_arith(r, cur_pc, 1, vf); return;
      case 0x24: v_fp_arith(r, cur_pc, 2, vf); return;
      case 0x20: v_fp_arith(r, cur_pc, 3, vf); return;
      case 0x27: if (!vf) break; v_fp_arith(r, cur_pc, 4, true); return;
      case 0x21: if (!vf) break; v_fp_arith(r, cur_pc, 5, true); return;
      case 0x28: case 0x29: case 0x2a: case 0x2b: case 0x2c: case 0x2d: case 0x2e: case 0x2f:
        v_fp_arith(r, cur_pc, 8 + (f6 - 0x28), vf); return;
      case 0x04: v_fp_arith(r, cur_pc, 16, vf); return;
      case 0x06: v_fp_arith(r, cur_pc, 17, vf); return;
      case 0x08: case 0x09: case 0x0a: v_fp_arith(r, cur_pc, 18 + (f6 - 0x08), vf); return;
      case 0x18: v_fp_compare(r, cur_pc, 0, vf); return;
      case 0x19: v_fp_compare(r, cur_pc, 1, vf); return;
      case 0x1b: v_fp_compare(r, cur_pc, 2, vf); return;
      case 0x1c: v_fp_compare(r, cur_pc, 3, vf); return;
      case 0x1d: if (!vf) break; v_fp_compare(r, cur_pc, 4, true); return;
      case 0x1f: if (!vf) break; v_fp_compare(r, cur_pc, 5, true); return;
      case 0x01: case 0x03: if (vf) break; v_fp_redsum(r, cur_pc, 0); return;
      case 0x05: case 0x07: if (vf) break; v_fp_redsum(r, cur_pc, f6 == 0x05 ? 1 : 2); return;
      case 0x0e: case 0x0f: if (!vf) break; v_slide(r, cur_pc, f6 == 0x0e, true, 0, true); return;   // vfslide1up.vf / vfslide1down.vf
      case 0x10: {
        const VCfg c = vcfg();
        if (!vf) {                                                              // vfmv.f.s (vector-float.cpp:503-583)
          if (rs1 != 0) { illegal(r, cur_pc); return; }
          if (!v_fp_mv_check(c, r, cur_pc)) return;
          if (!v_fp_legal(c.esz)) { vec_fail(r, cur_pc); return; }
          if (c.esz == 2) wf_s(rd, (u32)vrd(rs2, 0, 2)); else wf(rd, vrd(rs2, 0, 3));
          mark_fs_dirty(); vec_success(); return;
        }
        if (rs2 != 0) { illegal(r, cur_pc); return; }                           // vfmv.s.f (:585-655)
        if (!v_fp_mv_check(c, r, cur_pc)) return;
        if (!v_fp_legal(c.esz)) { vec_fail(r, cur_pc); return; }
        v_mv_s(c, rd, vf_scalar(rs1, c.esz)); return; }
      case 0x12:                                                                // VFUNARY0 (Decoder.cpp:394-425)
        if (vf) break;
        if (rs1 <= 3) { v_fp_unary(r, cur_pc, 1 + rs1); return; }
        if (rs1 == 6 || rs1 == 7) { v_fp_unary(r, cur_pc, rs1 - 1); return; }
        if (rs1 == 0x0c) { v_fp_cvt_ff(r, cur_pc, false); return; }             // vfwcvt.f.f.v
        if (rs1 == 0x14) { v_fp_cvt_ff(r, cur_pc, true); return; }              // vfncvt.f.f.w
        if (rs1 == 0x15) { v_fp_cvt_ff(r, cur_pc, true, true); return; }        // vfncvt.rod.f.f.w
        if ((rs1 >= 0x08 && rs1 <= 0x0b) || rs1 == 0x0e || rs1 == 0x0f || (rs1 >= 0x10 && rs1 <= 0x13) || rs1 == 0x16 || rs1 == 0x17) { v_fp_cvt_wn(r, cur_pc, rs1); return; }
        break;
      case 0x13:                                                                // VFUNARY1 (Decoder.cpp:427-433)
        if (vf) break;
        if (rs1 == 0) { v_fp_unary(r, cur_pc, 0); return; }                     // vfsqrt.v
        if (rs1 == 0x10) { v_fp_class(r, cur_pc); return; }                     // vfclass.v
        if (rs1 == 4 || rs1 == 5) { v_fp_unary(r, cur_pc, rs1 + 3); return; }   // vfrsqrt7.v / vfrec7.v
        break;
      case 0x30: v_fp_widen(r, cur_pc, 0, vf, false); return;                   // vfwadd
      case 0x32: v_fp_widen(r, cur_pc, 1, vf, false); return;                   // vfwsub
      case 0x34: v_fp_widen(r, cur_pc, 0, vf, true); return;                    // vfwadd.w
      case 0x36: v_fp_widen(r, cur_pc, 1, vf, true); return;                    // vfwsub.w
      case 0x38: v_fp_widen(r, cur_pc, 2, vf, false); return;                   // vfwmul
      case 0x3c: case 0x3d: case 0x3e: case 0x3f: v_fp_widen(r, cur_pc, 8 + 4 + (f6 - 0x3c), vf, false); return;   // vfwmacc vfwnmacc vfwmsac vfwnmsac
      case 0x31: case 0x33: if (vf) break; v_fp_wredsum(r, cur_pc); return;     // vfwredusum / vfwredosum
      case 0x17: {
        if (!vf) break;
        if (!vm) { v_fp_merge(r, cur_pc); return; }                             // vfmerge.vfm (Decoder.cpp:811)
        if (rs2 != 0) { illegal(r, cur_pc); return; }                           // vfmv.v.f (:4134-4215)
        const VCfg c = vcfg();
        if (!v_fp_mv_check(c, r, cur_pc)) return;
        if (!v_check_emul(r, cur_pc, c.gx8, rd, 0, 0)) return;
        if (!v_fp_legal(c.esz)) { vec_fail(r, cur_pc); return; }
        const u64 sc = vf_scalar(rs1, c.esz); const u32 dg = c.gx8 < 8 ? 8 : c.gx8;
        if (0 < c.vl) for (u32 ix = 0; ix < c.elem_max; ++ix) { u64 d; if (v_dest_active(c, rd, ix, c.esz, false, d)) d = sc; vwr(rd, ix, c.esz, dg, d); }
        vec_success(); return; }
      default: break;
    }
    vec_unsupported(inst);
  }
  // vector loads / stores: unit-stride and strided, nf = 0, mew = 0 only.  a = x[rs1], b = x[rs2]
  WSP_HD void exec_vector_ldst_body(const Rec& r, u64 a, u64 b, u64 cur_pc, bool store)
  {
    const u32 inst = r.raw, f3 = (inst >> 12) & 7, mop = (inst >> 26) & 3;
    s.vreg = (inst >> 7) & 31;
    const u32 lumop = (inst >> 20) & 31, nf = inst >> 29, mew = (inst >> 28) & 1;
    const u32 eew = f3 == 0 ? 0 : f3 - 4;
    // the warm path takes unit-stride, strided and mask loads / stores (their fault scan comes before the first write)
    if (DRY && !(mew == 0 && nf == 0 && ((mop == 0 && (lumop == 0 || lumop == 0xb)) || mop == 2))) { dry_bail(); return; }
    if (mop == 0 && lumop == 8) {                                               // whole registers (Decoder.cpp:949-967, 1184-1195)
      if (store) { if (mew || f3 != 0 || (nf != 0 && nf != 1 && nf != 3 && nf != 7)) { illegal(r, cur_pc); return; } }
      else if (mew) { vec_fail(r, cur_pc); return; }
      v_ldst_whole(r, a, cur_pc, eew, store); return;
    }
    if (mop == 0 && lumop == 0xb) {                                             // vlm.v / vsm.v (Decoder.cpp:968-972, 1196-1200)
      if (nf != 0 || mew != 0 || f3 != 0) { illegal(r, cur_pc); return; }
      if (store) v_store_unit(r, a, b, cur_pc, 0, false, true); else v_load_unit(r, a, b, cur_pc, 0, false, true);
      return;
    }
    if ((mop == 1 || mop == 3) && mew == 0) { v_ldst_indexed(r, a, cur_pc, eew, store, nf + 1); return; }      // indexed (and indexed segment), unordered and ordered alike
    if (mew == 0 && ((mop == 0 && lumop == 0) || mop == 2) && nf != 0) {        // vlseg / vsseg / vlsseg / vssseg (Decoder.cpp:930-947, 1030-1050)
      v_ldst_seg(r, a, cur_pc, eew, nf + 1, mop == 2 ? b : (u64)(nf + 1) << eew, store, false, true); return;
    }
    if (mew == 0 && mop == 0 && lumop == 0x10 && !store) {                      // vle<eew>ff.v / vlseg<nf>e<eew>ff.v (Decoder.cpp:973-1010)
      v_ldst_seg(r, a, cur_pc, eew, nf + 1, (u64)(nf + 1) << eew, false, true, nf != 0); return;
    }
    if ((!(mop == 0 && lumop == 0) && mop != 2) || nf != 0 || mew != 0) { vec_unsupported(inst); return; }
    if (store) v_store_unit(r, a, b, cur_pc, eew, mop == 2); else v_load_unit(r, a, b, cur_pc, eew, mop == 2);
  }

  // ---- the common, trap-free subset executed inline in the GPU hot loop (caller guarantees HF_HOT).
  // Returns false -- before any side effect -- when the instruction needs the full exec(): anything
  // that can trap, touch tohost / code / CSRs, FP, A, division.  Must behave exactly like exec() for
  // what it accepts.  Register-write / store capture for the commit record only when TRACE.
  WSP_HD void update_hot_flag()
  {
    // an interrupt that is pending and enabled (whatever its delegation / global enable) keeps the hart on the cold path,
    // which polls isInterruptPossible every step
    bool hot = !(h.flags & (HF_CODE_DIRTY | HF_INH_CY | HF_INH_IR | HF_PMP_MLOCK | HF_RING)) && h.priv == PRV_M && !(h.mstatus & MST_MPRV) &&
               (m.cold(CI_MIP) & m.cold(CI_MIE)) == 0;
    h.flags = hot ? (h.flags | HF_HOT) : (h.flags & ~(u32)HF_HOT);
    h.flags = (((h.mstatus >> MST_FS_SH) & 3) == 3) ? (h.flags | HF_FS_DIRTY) : (h.flags & ~(u32)HF_FS_DIRTY);
    h.flags = (((h.mstatus >> MST_VS_SH) & 3) == 3) ? (h.flags | HF_VS_DIRTY) : (h.flags & ~(u32)HF_VS_DIRTY);
  }
  template <bool TRACE> WSP_HD void wx_fast(u32 rd, u64 v)
  {
    if (rv32()) v = (u64)(i64)(i32)v;
    m.wx(rd, rd ? v : 0);                       // x0 slot always holds zero: no branch
    if (TRACE) { s.chg = CH_IREG; s.ireg = rd; s.ival = rv32() ? (u64)(u32)v : v; }
  }
  // pre-decoded fetch: false when the slot is not trustworthy (outside the decoded text, straddling)
  WSP_HD bool hot_fetch(const Rec* recs, Rec& rec) const
  {
    const u64 idx = h.pc - text_lo;
    if (idx >= text_hi - text_lo) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + (idx << 3));
    rec.w0 = q.x; rec.imm = (i32)q.y; rec.pad = q.z; rec.raw = q.w;
#else
    rec = recs[idx >> 1];
#endif
    return (rec.w0 & 0xff) != OP_SLOWFETCH;
  }
  template <bool TRACE> WSP_HD bool exec_fast(u32 op, const Rec& r, u64 a, u64 b, u64 cur_pc)
  {
    const u32 rd = rec_rd(r);
    if (TRACE) s.chg = 0;
    switch (op) {
      case OP_ADD: wx_fast<TRACE>(rd, a + b); return true;
      case OP_SUB: wx_fast<TRACE>(rd, a - b); return true;
      case OP_SLL: wx_fast<TRACE>(rd, a << (b & 63)); return true;
      case OP_SLT: wx_fast<TRACE>(rd, (i64)a < (i64)b); return true;
      case OP_SLTU: wx_fast<TRACE>(rd, a < b); return true;
      case OP_XOR: wx_fast<TRACE>(rd, a ^ b); return true;
      case OP_SRL: wx_fast<TRACE>(rd, a >> (b & 63)); return true;
      case OP_SRA: wx_fast<TRACE>(rd, (u64)((i64)a >> (b & 63))); return true;
      case OP_OR: wx_fast<TRACE>(rd, a | b); return true;
      case OP_AND: wx_fast<TRACE>(rd, a & b); return true;
      case OP_ADDW: wx_fast<TRACE>(rd, (u64)(i64)(i32)((u32)a + (u32)b)); return true;
      case OP_SUBW: wx_fast<TRACE>(rd, (u64)(i64)(i32)((u32)a - (u32)b)); return true;
      case OP_SLLW: wx_fast<TRACE>(rd, (u64)(i64)(i32)((u32)a << (b & 31))); return true;
      case OP_SRLW: wx_fast<TRACE>(rd, (u64)(i64)(i32)((u32)a >> (b & 31))); return true;
      case OP_SRAW: wx_fast<TRACE>(rd, (u64)(i64)((i32)a >> (b & 31))); return true;
      case OP_MUL: wx_fast<TRACE>(rd, a * b); return true;
      case OP_MULW: wx_fast<TRACE>(rd, (u64)(i64)(i32)((u32)a * (u32)b)); return true;
      case OP_AUIPC: wx_fast<TRACE>(rd, cur_pc + (u64)(i64)r.imm); return true;
      case OP_NOP: return true;
      case OP_JAL: {
        if (!ext(ISA_C)) return false;
        wx_fast<TRACE>(rd, cur_pc + ((r.w0 & F_SIZE4) ? 4 : 2)); h.pc = (cur_pc + (u64)(i64)r.imm) & ~1ULL & amask(); return true; }
      case OP_JALR: {
        if (!ext(ISA_C)) return false;
        wx_fast<TRACE>(rd, cur_pc + ((r.w0 & F_SIZE4) ? 4 : 2)); h.pc = (a + (u64)(i64)r.imm) & ~1ULL & amask(); return true; }
      case OP_BEQ: case OP_BNE: case OP_BLT: case OP_BGE: case OP_BLTU: case OP_BGEU: {
        if (!ext(ISA_C)) return false;
        bool take = op == OP_BEQ ? a == b : op == OP_BNE ? a != b : op == OP_BLT ? (i64)a < (i64)b
                  : op == OP_BGE ? (i64)a >= (i64)b : op == OP_BLTU ? a < b : a >= b;
        if (take) h.pc = (cur_pc + (u64)(i64)r.imm) & ~1ULL & amask();
        return true; }
      case OP_LB: case OP_LH: case OP_LW: case OP_LD: case OP_LBU: case OP_LHU: case OP_LWU: {
        const u32 k = op - OP_LB, sz = 1u << (k & 3);
        const u64 va = (a + (u64)(i64)r.imm) & amask();
        if ((va & (sz - 1)) || !in_mem(va, sz) || va == tohost) return false;
        const u64 off = va - mem_base; u64 v;
        switch (k) {
          case 0: v = (u64)(i64)(int8_t)m.ld8(off); break; case 1: v = (u64)(i64)(int16_t)m.ld16(off); break;
          case 2: v = (u64)(i64)(i32)m.ld32(off); break; case 3: v = m.ld64(off); break;
          case 4: v = m.ld8(off); break; case 5: v = m.ld16(off); break; default: v = m.ld32(off); break;
        }
        wx_fast<TRACE>(rd, v); return true; }
      case OP_SB: case OP_SH: case OP_SW: case OP_SD: {
        const u32 k = op - OP_SB, sz = 1u << k;
        const u64 va = (a + (u64)(i64)r.imm) & amask();
        if ((va & (sz - 1)) || !in_mem(va, sz) || va == tohost || va == tohost + 4 || (va + sz > text_lo && va < text_hi)) return false;
        const u64 off = va - mem_base;
        const u64 sv = k == 3 ? b : (b & ((1ULL << (8 * sz)) - 1));
        switch (k) { case 0: m.st8(off, (u8)sv); break; case 1: m.st16(off, (u16)sv); break; case 2: m.st32(off, (u32)sv); break; default: m.st64(off, sv); break; }
        if (TRACE) { s.chg = CH_MEM; s.maddr = va; s.mval = sv; }
        return true; }
      default: return false;
    }
  }
  // trace fold for an instruction that went through exec_fast (no trap, no CSR change, M-mode)
  template <bool TRACE> WSP_HD void trace_fast(const Rec& rec, u64 cur_pc)
  {
    if (!TRACE) return;
    u64 t = h.thash;
    t = fold(t, cur_pc);
    t = fold(t, (u64)rec.raw | ((u64)PRV_M << 32) | ((u64)((rec.w0 & F_SIZE4) ? 4 : 2) << 48));
    u64 k = 1ULL << 60, v = 0;
    if ((s.chg & CH_IREG) && s.ireg) { k = (u64)s.ireg ^ (1ULL << 60); v = s.ival; }
    else if (s.chg & CH_MEM) { k = s.maddr ^ (4ULL << 60); v = s.mval; }
    t = fold(t, k); t = fold(t, v);
    h.thash = t;
  }

  // The vector unit is large: out of line on the GPU.  For the general path the hart goes in by value and what the unit can
  // change (the hot registers and the commit bookkeeping; everything else it changes is in LDS / HBM) comes back by value: no
  // address of the caller's Lane is taken.  kind: 0 OP-V, 1 vector load, 2 vector store.  (The vector warm path has its own
  // copy of the unit, compiled for DRY = true inside wsp_vwarm.)
  struct VecRet { Hot h; Step s; };
  static WSP_HDN VecRet vec_call(Lane L, Rec r, u64 a, u64 b, u64 cur_pc, u32 kind)
  {
    if (kind == 0) L.exec_vector_body(r, a, b, cur_pc); else L.exec_vector_ldst_body(r, a, b, cur_pc, kind == 2);
    VecRet o; o.h = L.h; o.s = L.s; return o;
  }
#if defined(__HIPCC__)
  WSP_HD void exec_vector(const Rec& r, u64 a, u64 b, u64 cur_pc)
  { const unsigned long long t0 = WSP_PROF_T(); const VecRet o = vec_call(*this, r, a, b, cur_pc, 0); h = o.h; s = o.s; WSP_PROF_ADD(5, WSP_PROF_T() - t0); WSP_PROF_ADD(4, 1); }
  WSP_HD void exec_vector_ldst(const Rec& r, u64 a, u64 b, u64 cur_pc, bool store)
  { const unsigned long long t0 = WSP_PROF_T(); const VecRet o = vec_call(*this, r, a, b, cur_pc, store ? 2u : 1u); h = o.h; s = o.s; WSP_PROF_ADD(8, WSP_PROF_T() - t0); WSP_PROF_ADD(7, 1); }
#else
  WSP_HD void exec_vector(const Rec& r, u64 a, u64 b, u64 cur_pc) { exec_vector_body(r, a, b, cur_pc); }
  WSP_HD void exec_vector_ldst(const Rec& r, u64 a, u64 b, u64 cur_pc, bool store) { exec_vector_ldst_body(r, a, b, cur_pc, store); }
#endif

  // ---- Zba / Zbb / Zbs / Zbc / Zicond (bitmanip.cpp:10-1110, Hart.cpp:14728-14760).  a = x[rs1]; b = x[rs2] or the immediate;
  // par = the decoder's parameter (see ZB_*).  An RV32 hart's operands arrive sign-extended and wx() brings the result back into
  // that shape, so only the operations that look at the upper half have a 32-bit form.
  WSP_HD u64 exec_zb(u32 fn, i32 par, u64 a, u64 b) const
  {
    const bool w32 = rv32();
    const u32 aw = (u32)a, sm = w32 ? 31u : 63u, ix = (u32)b & sm, s5 = (u32)b & 31;
    switch (fn) {
      case ZB_SHADD: return (((par & 8) ? (u64)aw : a) << (par & 3)) + b;
      case ZB_SLLIUW: return (u64)aw << (b & 63);
      case ZB_ANDN: return a & ~b;
      case ZB_ORN: return a | ~b;
      case ZB_XNOR: return a ^ ~b;
      case ZB_CLZ: if (par) { u32 n = 0; while (n < 32 && !((aw << n) & 0x80000000u)) ++n; return n; }
                   else { u32 n = 0; while (n < 64 && !((a << n) >> 63)) ++n; return n; }
      case ZB_CTZ: if (par) { u32 n = 0; while (n < 32 && !((aw >> n) & 1)) ++n; return n; }
                   else { u32 n = 0; while (n < 64 && !((a >> n) & 1)) ++n; return n; }
      case ZB_CPOP: { u64 v = par ? (u64)aw : a; v = v - ((v >> 1) & 0x5555555555555555ULL); v = (v & 0x3333333333333333ULL) + ((v >> 2) & 0x3333333333333333ULL);
                      v = (v + (v >> 4)) & 0x0f0f0f0f0f0f0f0fULL; return (v * 0x0101010101010101ULL) >> 56; }
      case ZB_MIN: return (i64)a < (i64)b ? a : b;
      case ZB_MINU: return a < b ? a : b;
      case ZB_MAX: return (i64)a > (i64)b ? a : b;
      case ZB_MAXU: return a > b ? a : b;
      case ZB_SEXTB: return (u64)(i64)(int8_t)a;
      case ZB_SEXTH: return (u64)(i64)(int16_t)a;
      case ZB_ZEXTH: return a & 0xffff;
      case ZB_ROL: return (a << ix) | (a >> ((64 - ix) & 63));
      case ZB_ROR: return (a >> ix) | (a << ((64 - ix) & 63));
      case ZB_ROLW: return (u64)(i64)(i32)((aw << s5) | (aw >> ((32 - s5) & 31)));
      case ZB_RORW: return (u64)(i64)(i32)((aw >> s5) | (aw << ((32 - s5) & 31)));
      case ZB_ORCB: { u64 r = 0; for (u32 i = 0; i < 64; i += 8) if ((a >> i) & 0xff) r |= 0xffULL << i; return r; }
      case ZB_REV8: { u64 r = 0; const u32 n = par ? 4 : 8; for (u32 i = 0; i < n; ++i) r |= ((a >> (8 * i)) & 0xff) << (8 * (n - 1 - i)); return r; }
      case ZB_BCLR: return a & ~(1ULL << ix);
      case ZB_BEXT: return ((w32 ? (u64)aw : a) >> ix) & 1;
      case ZB_BINV: return a ^ (1ULL << ix);
      case ZB_BSET: return a | (1ULL << ix);
      case ZB_CZEQZ: return b == 0 ? 0 : a;
      case ZB_CZNEZ: return b != 0 ? 0 : a;
      default: {                                                             // carry-less multiply: low half, high half, reversed
        const u32 n = w32 ? 32 : 64; const u64 ua = w32 ? (u64)aw : a, ub = w32 ? (u64)(u32)b : b; u64 r = 0;
        for (u32 i = 0; i < n; ++i) {
          if (!((ub >> i) & 1)) continue;
          if (fn == ZB_CLMUL) r ^= ua << i; else if (fn == ZB_CLMULH) { if (i) r ^= ua >> (n - i); } else r ^= ua >> (n - i - 1);
        }
        return r; }
    }
  }

  // ---- M extension (Hart.cpp:13964-14170): registers only, never traps
  WSP_HD u64 muldiv(u32 op, u64 a, u64 b) const
  {
    switch (op) {
      case OP_MUL: return a * b;
      case OP_MULH: case OP_MULHSU: case OP_MULHU:
        if (rv32()) {                                                                        // Hart<uint32_t>: 32-bit operands
          const u32 aw = (u32)a, bw = (u32)b;
          return op == OP_MULH ? (u64)(((i64)(i32)aw * (i64)(i32)bw) >> 32) : op == OP_MULHSU ? (u64)(((i64)(i32)aw * (i64)(u64)bw) >> 32) : ((u64)aw * (u64)bw) >> 32;
        }
        if (op == OP_MULH) { wsf::U128 p = wsf::mul64x64(a, b); u64 hi = p.hi; if ((i64)a < 0) hi -= b; if ((i64)b < 0) hi -= a; return hi; }
        if (op == OP_MULHSU) { wsf::U128 p = wsf::mul64x64(a, b); u64 hi = p.hi; if ((i64)a < 0) hi -= b; return hi; }
        return wsf::mul64x64(a, b).hi;
      case OP_DIV: return b == 0 ? ~0ULL : ((i64)a == INT64_MIN && (i64)b == -1) ? a : (u64)((i64)a / (i64)b);
      case OP_DIVU: return b == 0 ? ~0ULL : a / b;
      case OP_REM: return b == 0 ? a : ((i64)a == INT64_MIN && (i64)b == -1) ? 0 : (u64)((i64)a % (i64)b);
      case OP_REMU: return b == 0 ? a : a % b;
      case OP_MULW: return (u64)(i64)(i32)((u32)a * (u32)b);
      case OP_DIVW: { i32 x = (i32)a, y = (i32)b; return (u64)(i64)(y == 0 ? -1 : (x == INT32_MIN && y == -1) ? x : x / y); }
      case OP_DIVUW: { u32 x = (u32)a, y = (u32)b; return (u64)(i64)(i32)(y == 0 ? ~0u : x / y); }
      case OP_REMW: { i32 x = (i32)a, y = (i32)b; return (u64)(i64)(y == 0 ? x : (x == INT32_MIN && y == -1) ? 0 : x % y); }
      default: { u32 x = (u32)a, y = (u32)b; return (u64)(i64)(i32)(y == 0 ? x : x % y); }   // OP_REMUW
    }
  }

  // ---- one decoded instruction. `op` is passed separately so the GPU kernel can hand in a
  // wave-uniform copy (scalar branch over the switch); a = x[rs1], b = second operand.
  WSP_HD void exec(u32 op, const Rec& r, u64 a, u64 b, u64 cur_pc)
  {
    u32 rd = rec_rd(r);
    u32 size = (r.w0 & F_SIZE4) ? 4 : 2;
    bool zca = ext(ISA_C);
    switch (op) {
      case OP_ADD: wx(rd, a + b); break;
      case OP_SUB: wx(rd, a - b); break;
      case OP_SLL: wx(rd, a << (b & 63)); break;
      case OP_SLT: wx(rd, (i64)a < (i64)b); break;
      case OP_SLTU: wx(rd, a < b); break;
      case OP_XOR: wx(rd, a ^ b); break;
      case OP_SRL: wx(rd, a >> (b & 63)); break;
      case OP_SRA: wx(rd, (u64)((i64)a >> (b & 63))); break;
      case OP_OR: wx(rd, a | b); break;
      case OP_AND: wx(rd, a & b); break;
      case OP_ADDW: wx(rd, (u64)(i64)(i32)((u32)a + (u32)b)); break;
      case OP_SUBW: wx(rd, (u64)(i64)(i32)((u32)a - (u32)b)); break;
      case OP_SLLW: wx(rd, (u64)(i64)(i32)((u32)a << (b & 31))); break;
      case OP_SRLW: wx(rd, (u64)(i64)(i32)((u32)a >> (b & 31))); break;
      case OP_SRAW: wx(rd, (u64)(i64)((i32)a >> (b & 31))); break;
      case OP_MUL: case OP_MULH: case OP_MULHSU: case OP_MULHU: case OP_DIV: case OP_DIVU: case OP_REM: case OP_REMU:
      case OP_MULW: case OP_DIVW: case OP_DIVUW: case OP_REMW: case OP_REMUW: wx(rd, muldiv(op, a, b)); break;
      case OP_ZB: wx(rd, exec_zb(rec_rs3(r), (r.w0 & F_IMM) ? 0 : r.imm, a, b)); break;
      case OP_AUIPC: wx(rd, cur_pc + (u64)(i64)r.imm); break;
      case OP_JAL: {
        u64 t = (cur_pc + (u64)(i64)r.imm) & amask();
        if (!zca && (t & 3)) { take_exception(0, cur_pc, t, true); break; }                  // Hart.cpp:11687-11690
        wx(rd, cur_pc + size); h.pc = t & ~1ULL; break; }
      case OP_JALR: {
        u64 t = (a + (u64)(i64)r.imm) & ~1ULL & amask();
        if (!zca && (t & 3)) { take_exception(0, cur_pc, t, true); break; }                  // Hart.cpp:11711-11714
        wx(rd, cur_pc + size); h.pc = t; break; }
      case OP_BEQ: case OP_BNE: case OP_BLT: case OP_BGE: case OP_BLTU: case OP_BGEU: {
        bool take = op == OP_BEQ ? a == b : op == OP_BNE ? a != b : op == OP_BLT ? (i64)a < (i64)b
                  : op == OP_BGE ? (i64)a >= (i64)b : op == OP_BLTU ? a < b : a >= b;
        if (take) {
          u64 t = (cur_pc + (u64)(i64)r.imm) & amask();
          if (!zca && (t & 3)) { take_exception(0, cur_pc, t, true); break; }                // Hart.cpp:1652-1656
          h.pc = t & ~1ULL;
        }
        break; }
      case OP_LB: case OP_LH: case OP_LW: case OP_LD: case OP_LBU: case OP_LHU: case OP_LWU: {
        u32 k = op - OP_LB; u32 sz = 1u << (k & 3); u64 v;
        if (!do_load(a + (u64)(i64)r.imm, sz, cur_pc, v)) break;
        if (k == 0) v = (u64)(i64)(int8_t)v; else if (k == 1) v = (u64)(i64)(int16_t)v; else if (k == 2) v = (u64)(i64)(i32)v;
        wx(rd, v); break; }
      case OP_SB: case OP_SH: case OP_SW: case OP_SD: do_store(a + (u64)(i64)r.imm, 1u << (op - OP_SB), b, cur_pc); break;
      case OP_NOP: break;
      case OP_ECALL: take_exception(h.priv == PRV_M ? 11 : h.priv == PRV_S ? 9 : 8, cur_pc, 0, true); break;   // Hart.cpp:12006-12014
      case OP_EBREAK: take_exception(3, cur_pc, ext(CFG_CLR_MTVAL_EBREAK) ? 0 : cur_pc, true); break;                        // Hart.cpp:12094-12101
      case OP_MRET: {                                                                        // Hart.cpp:12247-12336
        if (h.priv < PRV_M) { illegal(r, cur_pc); break; }
        u64 v = h.mstatus; u32 saved = (v >> MST_MPP_SH) & 3;
        v = (v & ~MST_MIE) | ((v & MST_MPIE) ? MST_MIE : 0);
        u32 least = ext(ISA_U) ? PRV_U : ext(ISA_S) ? PRV_S : PRV_M;
        v = (v & ~(3ULL << MST_MPP_SH)) | ((u64)least << MST_MPP_SH);
        v |= MST_MPIE;
        if (saved != PRV_M && !ext(CFG_KEEP_MPRV)) v &= ~MST_MPRV;
        csr_store(C_MSTATUS, v); s.chg |= CH_MSTATUS;
        h.pc = m.cold(CI_MEPC) & epc_rmask(); h.priv = saved; break; }
      case OP_SRET: {                                                                        // Hart.cpp:12428-12560
        if (h.priv < PRV_S || ((h.mstatus & MST_TSR) && h.priv == PRV_S)) { illegal(r, cur_pc); break; }
        u64 v = h.mstatus; u32 saved = (v & MST_SPP) ? PRV_S : PRV_U;
        v = (v & ~MST_SIE) | ((v & MST_SPIE) ? MST_SIE : 0);
        v = ext(ISA_U) ? (v & ~MST_SPP) : (v | MST_SPP);
        v |= MST_SPIE; if (!ext(CFG_KEEP_MPRV)) v &= ~MST_MPRV;                              // Hart.cpp:12499 (saved mode is never M here)
        h.mstatus = legalize_mstatus(v); s.chg |= CH_SSTATUS;
        h.pc = m.cold(CI_SEPC) & epc_rmask(); h.priv = saved; break; }
      case OP_WFI:                                                                           // Hart.cpp:12618
        if ((h.priv == PRV_U && ext(ISA_S)) || (h.priv != PRV_M && (h.mstatus & MST_TW))) illegal(r, cur_pc);
        break;
      case OP_SFENCE_VMA:                                                                    // Hart.cpp:12108
        if (h.priv == PRV_U || (h.priv == PRV_S && (h.mstatus & MST_TVM))) { illegal(r, cur_pc); break; }
        sfence(rec_rs1(r) != 0, a, rec_rs2(r) != 0, b);                                      // by register number (Hart.cpp:12139-12173)
        break;
      case OP_CSRRW: case OP_CSRRS: case OP_CSRRC: exec_csr(r, op - OP_CSRRW, (r.w0 & F_IMM) ? (u64)rec_rs1(r) : (a & amask()), cur_pc); break;
      case OP_LR: case OP_SC: case OP_AMO: exec_atomic(r, a & amask(), b, cur_pc); break;
      case OP_FLW: case OP_FLD: {                                                            // float.cpp:281-298, 1019-1036
        if (!fp_on()) { illegal(r, cur_pc); break; }
        u64 v;
        if (!do_load(a + (u64)(i64)r.imm, op == OP_FLW ? 4 : 8, cur_pc, v)) break;
        if (op == OP_FLW) wf_s(rd, (u32)v); else wf(rd, v);
        mark_fs_dirty(); break; }
      case OP_FSW: case OP_FSD: {                                                            // float.cpp:303-321, 1041-1057
        if (!fp_on()) { illegal(r, cur_pc); break; }
        u64 v = m.rf(rec_rs2(r));
        do_store(a + (u64)(i64)r.imm, op == OP_FSW ? 4 : 8, op == OP_FSW ? (v & 0xffffffffULL) : v, cur_pc);
        break; }
      case OP_FADD: case OP_FSUB: case OP_FMUL: case OP_FDIV: case OP_FSQRT: case OP_FMA: case OP_FSGNJ: case OP_FMINMAX:
      case OP_FCMP: case OP_FCVT_F2I: case OP_FCVT_I2F: case OP_FCVT_FF: case OP_FMV_F2X: case OP_FCLASS: case OP_FMV_X2F:
        exec_fp(r, a, cur_pc); break;
      case OP_VEC: exec_vector(r, a, b, cur_pc); break;
      case OP_VLD: exec_vector_ldst(r, a, b, cur_pc, false); break;
      case OP_VST: exec_vector_ldst(r, a, b, cur_pc, true); break;
      default: illegal(r, cur_pc); break;
    }
  }

  // ---- fetch from RAM + decode (slow path; also the only path for self-modified code)
  // fetchInstNoTrap (Hart.cpp:3052-3206). Returns 0 ok, 1 fault (cause, tval), 2 stop.
  WSP_HD int fetch_slow(u64 pc, Rec& r, u64& tval, u32& cause)
  {
    if (ext(ISA_S) && h.priv != PRV_M && (m.cold(CI_SATP) >> 60) > 8) { stop(ST_STOP_UNSUPPORTED, UNS_PAGING); return 2; }
    const bool tr = paging_on(h.priv);
    u64 pa = pc; u32 c;
    cause = 1;
    if (tr && (c = translate(pc, h.priv, 4, pa)) != 0) { cause = c; tval = pc; return 1; }
    if (!(pmp_perm(h.priv, pa) & 4)) { tval = pc; return 1; }                      // Hart.cpp:3079-3084
    u32 inst;
    if (!(pa & 3)) {
      if (!in_mem(pa, 4)) { tval = pc; return 1; }
      inst = m.ld32(pa - mem_base);
      if ((inst & 3) != 3) inst &= 0xffff;
    } else {
      if (!in_mem(pa, 2)) { tval = pc; return 1; }
      inst = m.ld16(pa - mem_base);
      if ((inst & 3) == 3) {
        // second half: starts out as the *virtual* address + 2 and is translated when its page index differs from the first
        // half's physical page index (Hart.cpp:3144-3156)
        u64 pa2 = pc + 2;
        if ((pa >> 12) != (pa2 >> 12) && tr && (c = translate(pc + 2, h.priv, 4, pa2)) != 0) { cause = c; tval = pc + 2; return 1; }
        if (!(pmp_perm(h.priv, pa2) & 4)) { tval = pc + 2; return 1; }              // Hart.cpp:3158-3162
        if (!in_mem(pa2, 2)) { tval = pc + 2; return 1; }
        inst |= (u32)m.ld16(pa2 - mem_base) << 16;
      }
    }
    r = decode(inst, isa);
    return 0;
  }

  // ---- one loop iteration of Hart::untilAddress (Hart.cpp:6025-6201) on the general (cold) path:
  // bookkeeping, fetch from RAM incl. fetch faults, full exec(), retire + commit record.  The hot path
  // (HF_HOT harts at a pre-decoded pc executing the exec_fast subset) is the same step with the
  // bookkeeping reduced to "all four counters advance by one"; see the kernel / host shim drivers.
  WSP_HD void cold_begin(u64& cur_pc, u32& priv0)
  {
    cur_pc = h.pc; priv0 = h.priv;
    s.chg = 0; s.vgroup = 0;
    h.exec++;
    if (!(h.flags & HF_INH_CY)) h.cycle++;
  }
  template <bool TRACE> WSP_HD bool fetch_slow_path(Rec& rec, u64& a, u64& b, u64 cur_pc, u32 priv0)
  {
    u64 tval = 0; u32 fcause = 1;
    int fr = fetch_slow(cur_pc, rec, tval, fcause);
    if (fr == 1) {
      take_exception(fcause, cur_pc, tval, true);
      if (h.status != ST_RUNNING) {                 // consecutive-illegal stop inside the fetch fault
        h.retired++; if (!(h.flags & HF_INH_IR)) h.instret++;
        commit<TRACE>(cur_pc, 0, 2, priv0, 0);
      } else {
        if (!(h.flags & HF_INH_CY)) h.cycle++;      // fetchInstWithTrigger (Hart.cpp:5966-5967)
        commit<TRACE>(cur_pc, 0, 2, priv0, 1);
      }
      return false;
    }
    if (fr != 0) return false;
    h.pc = (cur_pc + ((rec.w0 & F_SIZE4) ? 4 : 2)) & amask();   // Hart.cpp:6120
    a = m.rx(rec_rs1(rec));
    b = (rec.w0 & F_IMM) ? (u64)(i64)rec.imm : m.rx(rec_rs2(rec));
    return true;
  }
  // retire + commit record (Hart.cpp:6128-6179; logStop printTrace.cpp:965-976)
  template <bool TRACE> WSP_HD void step_end(const Rec& rec, u64 cur_pc, u32 priv0)
  {
    if (h.status == ST_STOP_UNSUPPORTED) return;
    const u32 size = (rec.w0 & F_SIZE4) ? 4 : 2;
    if (h.status == ST_RUNNING && (s.chg & CH_EXC)) { commit<TRACE>(cur_pc, rec.raw, size, priv0, 1); return; }
    if (!(h.flags & HF_INH_IR)) h.instret++;
    h.retired++;
    commit<TRACE>(cur_pc, rec.raw, size, priv0, 0);
  }
  // a complete cold step.  `recs` (may be null) is the hart's pre-decoded program image: an M-mode hart whose code is
  // not dirty takes the record from it instead of decoding from RAM (same result: the image is decode() of the same
  // bytes, and M-mode fetch inside the arena cannot fault while no locked PMP entry is on).
  template <bool TRACE> WSP_HD void cold_step(const Rec* recs)
  {
    u64 cur_pc, a = 0, b = 0; u32 priv0; Rec rec; rec.w0 = 0; rec.imm = 0; rec.raw = 0; rec.pad = 0;
    if (h.pc == stop_pc) { h.status = ST_STOP_ADDR; return; }                  // `while (pc_ != address ...` (Hart.cpp:6025): nothing executes
    cold_begin(cur_pc, priv0);
    if (poll_interrupt<TRACE>(cur_pc, priv0)) { update_hot_flag(); return; }
    bool have = false;
    if (recs && h.priv == PRV_M && !(h.flags & (HF_CODE_DIRTY | HF_PMP_MLOCK)) && hot_fetch(recs, rec)) {
      if ((rec.w0 & 0xff) == OP_STOPCHK) { const Rec r2 = decode(rec.raw, isa); rec.w0 = r2.w0; rec.imm = r2.imm; }   // another hart's stop address
      h.pc = (cur_pc + ((rec.w0 & F_SIZE4) ? 4 : 2)) & amask();
      a = m.rx(rec_rs1(rec));
      b = (rec.w0 & F_IMM) ? (u64)(i64)rec.imm : m.rx(rec_rs2(rec));
      have = true;
    } else
      have = fetch_slow_path<TRACE>(rec, a, b, cur_pc, priv0);
    if (have) { exec(rec.w0 & 0xff, rec, a, b, cur_pc); step_end<TRACE>(rec, cur_pc, priv0); }
    update_hot_flag();
  }

  // ---- The TLB-hit fast step: S / U-mode code under Sv39 without the general path (VirtMem::translate's hit case,
  // VirtMem.cpp:261-310; Tlb::findEntryUpdateTime, Tlb.hpp:91-100; the fetch and load / store glue of Hart.cpp:3052-3206, 1988-2154).
  // One instruction of the inline subset (exec_fast's register / branch / jump forms, aligned loads and stores) whose fetch and
  // data translations both HIT the TLB with permission, whose pages lie whole inside one PMP region that allows the access,
  // inside the arena, away from tohost and (stores) from the code image; the record comes from the pre-decoded image by PHYSICAL
  // address, or is decoded from RAM when the code lies outside the image.  Everything is checked before anything is changed: on `false` nothing has happened and the general
  // step executes the instruction (misses and walks, faults, CSRs, traps, misaligned or page-crossing accesses, a straddling
  // fetch ...).  On `true` the side effects are exactly the general path's: the replacement counters of the TLB entries used
  // (fetch first, then data), the register / memory write, the pc.  The caller advances the four counters by one, as for a hot
  // step, and has established: priv != M, satp.MODE = Sv39, code not dirty, no ring, no inhibited counter, no interrupt pending
  // and enabled.  `FastTr` carries what stays valid from one fast step to the next (anything that could change it - a CSR
  // write, sfence.vma, a trap, a TLB refill - is not a fast step and ends the run).
  struct FastTr {
    u64 satp;
    u64 fvpn, fpa, fe0, fe1; u32 fix, fok;            // the fetch page: its TLB entry (word 0 as last stored), physical page, TLB index
    u64 dvpn, de0, de1, dpage; u32 dix, dok, dperm;    // the last data page: its TLB entry, physical page, PMP verdict
    Rec nrec; u64 nidx; u32 nok;                       // the record that follows the current one in its page, fetched ahead
  };
  WSP_HD void fast_tr_begin(FastTr& t)
  {
    t.satp = m.cold(CI_SATP); t.fok = 0; t.fvpn = 0; t.fpa = 0; t.fe0 = 0; t.fe1 = 0; t.fix = 0;
    t.dvpn = 0; t.de0 = 0; t.de1 = 0; t.dpage = ~0ULL; t.dix = 0; t.dok = 0; t.dperm = 0x80; t.nidx = 0; t.nok = 0;
    t.nrec.w0 = 0; t.nrec.imm = 0; t.nrec.pad = 0; t.nrec.raw = 0;
  }
  WSP_HD static Rec fast_rec_load(const Rec* recs, u64 idx)
  {
    Rec rec;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + (idx << 3)); rec.w0 = q.x; rec.imm = (i32)q.y; rec.pad = q.z; rec.raw = q.w;
#else
    rec = recs[idx >> 1];
#endif
    return rec;
  }
  WSP_HD bool fast_tr_allowed()
  {
    return h.priv != PRV_M && tlb_n != 0 && !(h.flags & (HF_CODE_DIRTY | HF_INH_CY | HF_INH_IR | HF_RING)) && (m.cold(CI_SATP) >> 60) == 8 &&
           (m.cold(CI_MIP) & m.cold(CI_MIE)) == 0;
  }
  // pmp_perm for a whole 4 KiB page: the permission bits when one region decides for every byte of it (or none matches any),
  // 0x80 when the first region that touches the page covers only part of it
  WSP_HD u32 pmp_perm_page(u32 pm, u64 page)
  {
    const u32 dflt = pm == PRV_M ? 7u : 0u;
    if (!(h.flags & HF_PMP_ON) || (pm == PRV_M && !(h.flags & HF_PMP_MLOCK))) return dflt;
    if (h.flags & HF_PMP_E0ALL) { const u32 cfg = (u32)m.cold(CI_PMPCFG0) & 0xff; return (pm == PRV_M && !(cfg & 0x80)) ? 7u : (cfg & 7u); }
    const u64 last = page + 4095;
    for (u32 k = 0; k < 8; ++k) {
      const u64 w = m.cold(CI_PMPCFG0 + k);
      if (!(w & 0x1818181818181818ULL)) continue;
      for (u32 j = 0; j < 8; ++j) {
        const u32 cfg = (u32)(w >> (8 * j)) & 0xff, type = (cfg >> 3) & 3, ix = 8 * k + j;
        if (!type) continue;
        const u64 v = m.cold(CI_PMPADDR0 + ix);
        u64 lo, hi;
        if (type == 1) { lo = (ix ? m.cold(CI_PMPADDR0 + ix - 1) : 0) << 2; hi = v << 2; if (hi > 0) hi -= 1; }
        else {
          u64 size_m1 = 3, napot = v;
          if (type == 3) { u32 rzi = 0; while (rzi < 64 && ((v >> rzi) & 1)) ++rzi; napot = rzi >= 64 ? 0 : ((v >> rzi) << rzi); size_m1 = rzi + 3 >= 64 ? ~0ULL : ((1ULL << (rzi + 3)) - 1); }
          lo = napot << 2; hi = lo + size_m1;
        }
        const u64 a0 = (lo >> 2) << 2, a1 = ((hi >> 2) << 2) + 3;
        if (a0 > a1 || last < a0 || page > a1) continue;                              // (an empty TOR range matches nothing)
        if (page >= a0 && last <= a1) return (pm == PRV_M && !(cfg & 0x80)) ? 7u : (cfg & 7u);
        return 0x80;
      }
    }
    return dflt;
  }
  template <bool TRACE> WSP_HD bool fast_step_translated(const Rec* recs, FastTr& t)
  {
    const u32 pm = h.priv;
    const u64 pc = h.pc, vpn = pc >> 12;
    if (!t.fok || t.fvpn != vpn) {
      const u32 ix = 2 * (u32)(vpn & (tlb_n - 1));
      const u64 e0 = m.tlb(ix), e1 = m.tlb(ix + 1);
      if (!Vm::tlb_match(e0, e1, vpn, t.satp)) return false;
      const u32 fl = (u32)(e0 >> 56);
      if ((pm == PRV_U && !(fl & TE_USER)) || (pm == PRV_S && (fl & TE_USER)) || (fl & (TE_X | TE_A)) != (TE_X | TE_A)) return false;
      const u64 pap = (e1 & 0xfffffffffffULL) << 12;
      if (!in_mem(pap, 4096)) return false;
      const u32 perm = pmp_perm_page(pm, pap);
      if ((perm & 0x80) || !(perm & 4)) return false;
      t.fok = 1; t.fvpn = vpn; t.fpa = pap; t.fix = ix; t.fe0 = e0; t.fe1 = e1;
      if (t.dok && t.dix == ix) t.dok = 0;             // (one copy per TLB entry: the fetch side owns this one now)
    }
    const u64 fpa = t.fpa | (pc & 0xfff), idx = fpa - text_lo, text_len = text_hi - text_lo;
    Rec rec;
    if (idx < text_len) {
      rec = (t.nok && t.nidx == idx) ? t.nrec : fast_rec_load(recs, idx);
      const u32 sz0 = (rec.w0 & F_SIZE4) ? 4 : 2;
      // the sequential successor's record, requested now so that it arrives while this instruction's data access is under way
      if ((pc & 0xfff) + sz0 < 0x1000 && idx + sz0 < text_len) { t.nidx = idx + sz0; t.nrec = fast_rec_load(recs, t.nidx); t.nok = 1; } else t.nok = 0;
    } else {
      // code outside the pre-decoded image (a segment the ELF does not mark executable, code the program wrote): fetched from RAM
      // and decoded at every step, as fetch_slow does (the page lies inside the arena: checked with the TLB entry above)
      const u64 o = fpa - mem_base;
      u32 inst = m.ld16(o);
      if ((inst & 3) == 3) {
        if ((pc & 0xfff) == 0xffe) return false;
        inst |= (u32)m.ld16(o + 2) << 16;
      }
      rec = decode(inst, isa);
      t.nok = 0;
    }
    const u32 op = rec.w0 & 0xff, size = (rec.w0 & F_SIZE4) ? 4 : 2;
    if (size == 4 && (pc & 0xfff) == 0xffe) return false;                              // second half on the next page (Hart.cpp:3144-3156)
    // ... and the reference translates the second half of a 4-byte instruction at a 2-byte-aligned pc AGAIN whenever the page
    // number of the virtual pc + 2 differs from the first half's PHYSICAL page number (i.e. whenever VA != PA): one more hit
    const bool second_lookup = size == 4 && (pc & 2) && ((t.fpa >> 12) != ((pc + 2) >> 12));
    const u64 a = m.rx(rec_rs1(rec)), b = (rec.w0 & F_IMM) ? (u64)(i64)rec.imm : m.rx(rec_rs2(rec));
    // F / D with mstatus.FS already Dirty (nothing of mstatus changes then): flw / fld / fsw / fsd like the integer accesses, the
    // computational instructions through exec_fp (registers and fflags only); a reserved rounding mode is the general path's
    const bool fs_dirty = ((h.mstatus >> MST_FS_SH) & 3) == 3;
    const bool is_fls = fs_dirty && op >= OP_FLW && op <= OP_FSD, is_fp = fs_dirty && op >= OP_FADD && op <= OP_FMV_X2F;
    const bool is_ld = (op >= OP_LB && op <= OP_LWU) || (is_fls && op <= OP_FLD), is_st = (op >= OP_SB && op <= OP_SD) || (is_fls && op >= OP_FSW);
    if (is_fp) {
      const bool uses_rm = (op >= OP_FADD && op <= OP_FMA) || op == OP_FCVT_F2I || op == OP_FCVT_I2F || op == OP_FCVT_FF;
      int rm = 0;
      if (uses_rm && !fp_rm((u32)rec.imm & 7, rm)) return false;
    }
    u64 de0 = 0, off = 0, va = 0; u32 dix = 0, sz = 0;
    if (is_ld || is_st) {
      sz = is_fls ? ((op == OP_FLW || op == OP_FSW) ? 4u : 8u) : is_ld ? 1u << ((op - OP_LB) & 3) : 1u << (op - OP_SB);
      va = (a + (u64)(i64)rec.imm) & amask();
      if (va & (sz - 1)) return false;
      const u64 dvpn = va >> 12;
      dix = 2 * (u32)(dvpn & (tlb_n - 1));
      u64 de1;
      if (dix == t.fix) { de0 = t.fe0; de1 = t.fe1; }
      else if (t.dok && t.dvpn == dvpn) { de0 = t.de0; de1 = t.de1; }
      else { de0 = m.tlb(dix); de1 = m.tlb(dix + 1); }
      if (!Vm::tlb_match(de0, de1, dvpn, t.satp)) return false;
      const u32 fl = (u32)(de0 >> 56);
      const bool sum = (h.mstatus & MST_SUM) != 0, mxr = (h.mstatus & MST_MXR) != 0;
      if ((pm == PRV_U && !(fl & TE_USER)) || (pm == PRV_S && (fl & TE_USER) && !sum) || !(fl & TE_A)) return false;
      if (is_ld ? !((fl & TE_R) || (mxr && (fl & TE_X))) : (fl & (TE_W | TE_D)) != (TE_W | TE_D)) return false;
      const u64 dpage = (de1 & 0xfffffffffffULL) << 12, dpa = dpage | (va & 0xfff);
      if (t.dpage != dpage) { t.dperm = pmp_perm_page(pm, dpage); t.dpage = dpage; }
      if ((t.dperm & 0x80) || !(t.dperm & (is_ld ? 1u : 2u))) return false;
      if (!in_mem(dpa, sz)) return false;
      if (tohost != ~0ULL && dpa + sz > tohost && dpa < tohost + 8) return false;
      if (is_st && dpa + sz > text_lo && dpa < text_hi) return false;
      off = dpa - mem_base;
      if (dix != t.fix) { t.dok = 1; t.dvpn = dvpn; t.dix = dix; t.de0 = de0; t.de1 = de1; }      // (nothing can fail from here on)
    }
    h.pc = (pc + size) & amask();
    if (is_fp) {
      s.chg = 0; s.ireg = 0; s.freg = 0; s.gen_csr = 0; s.vgroup = 0;
      exec_fp(rec, a, pc);
    } else if (is_ld && is_fls) {
      const u64 v = sz == 4 ? (u64)m.ld32(off) : m.ld64(off);
      s.chg = 0;
      if (op == OP_FLW) wf_s(rec_rd(rec), (u32)v); else wf(rec_rd(rec), v);
    } else if (is_ld) {
      u64 v;
      switch (op - OP_LB) {
        case 0: v = (u64)(i64)(int8_t)m.ld8(off); break; case 1: v = (u64)(i64)(int16_t)m.ld16(off); break;
        case 2: v = (u64)(i64)(i32)m.ld32(off); break; case 3: v = m.ld64(off); break;
        case 4: v = m.ld8(off); break; case 5: v = m.ld16(off); break; default: v = m.ld32(off); break;
      }
      if (TRACE) s.chg = 0;
      wx_fast<TRACE>(rec_rd(rec), v);
    } else if (is_st) {
      const u64 sb = is_fls ? m.rf(rec_rs2(rec)) : b;
      const u64 sv = sz == 8 ? sb : (sb & ((1ULL << (8 * sz)) - 1));
      switch (sz) { case 1: m.st8(off, (u8)sv); break; case 2: m.st16(off, (u16)sv); break; case 4: m.st32(off, (u32)sv); break; default: m.st64(off, sv); break; }
      if (TRACE) { s.chg = CH_MEM; s.maddr = va; s.mval = sv; }                        // the log's m line carries the virtual address
    } else if (!exec_fast<TRACE>(op, rec, a, b, pc)) { h.pc = pc; return false; }
    // the replacement counters: the fetch's entry, then the data access's (the same entry when code and data share a page)
    t.fe0 = (t.fe0 & ~(3ULL << 54)) | ((((t.fe0 >> 54) + 1) & 3) << 54);
    if (second_lookup) t.fe0 = (t.fe0 & ~(3ULL << 54)) | ((((t.fe0 >> 54) + 1) & 3) << 54);
    if ((is_ld || is_st) && dix == t.fix) t.fe0 = (t.fe0 & ~(3ULL << 54)) | ((((t.fe0 >> 54) + 1) & 3) << 54);
    m.tlb(t.fix) = t.fe0;
    if ((is_ld || is_st) && dix != t.fix) { t.de0 = (de0 & ~(3ULL << 54)) | ((((de0 >> 54) + 1) & 3) << 54); m.tlb(dix) = t.de0; }
    if (TRACE && is_fp) commit<true>(pc, rec.raw, size, pm, 0);        // (register and / or fcsr: the general record builder)
    else if (TRACE) {
      u64 th = h.thash;
      th = fold(th, pc);
      th = fold(th, (u64)rec.raw | ((u64)pm << 32) | ((u64)size << 48));
      u64 k = 1ULL << 60, v = 0;
      if ((s.chg & CH_IREG) && s.ireg) { k = (u64)s.ireg ^ (1ULL << 60); v = s.ival; }
      else if (s.chg & CH_FREG) { k = (u64)s.freg ^ (2ULL << 60); v = s.fval; }
      else if (s.chg & CH_MEM) { k = s.maddr ^ (4ULL << 60); v = s.mval; }
      th = fold(th, k); th = fold(th, v);
      h.thash = th;
    }
    return true;
  }

  // ---- fold one commit record into the running trace hash
  // Commit record of one step.  The digest folds a word stream: pc, meta, then one (key, value) pair per change in the order the
  // reference's log prints them (printTrace.cpp:380-495), key = id ^ kind << 60.  With a commit ring configured
  // (wsp_batch_trace_ring) the record is also appended to this hart's ring as pc, meta, kinds, then (id, value) per pair: the
  // ids are stored whole (a virtual address may use bits 60..63) and the 4-bit kinds travel in their own word, pair k in bits
  // 4k..4k+3; the pair count is patched into bits 56.. of the stored meta word.  Recording stops when the ring is full, the
  // digest goes on.
  template <bool TRACE> WSP_HD void commit(u64 cur_pc, u32 raw, u32 size, u32 priv0, u32 trap)
  {
    if (!TRACE || (h.flags & HF_NOTRACE)) return;
    u64 t = h.thash;
    u64* const rg = ring_; u32 pos = rg ? *ring_pos_ : 0; const u32 cap = ring_cap, meta_at = pos + 1, kinds_at = pos + 2; u32 np = 0; u64 kinds = 0;
#define WSP_RING(w) do { if (rg && pos < cap) rg[pos++] = (w); } while (0)
#define WSP_RING2(w) do { if (pos < cap) rg[pos++] = (w); } while (0)
#define WSP_PAIR(kind, id, v) do { const u64 id_ = (id), v_ = (v); t = fold(t, id_ ^ ((u64)(kind) << 60)); t = fold(t, v_); \
                                   WSP_RING(id_); WSP_RING(v_); if (np < 16) kinds |= (u64)(kind) << (4 * np); ++np; } while (0)
    const u64 meta = (u64)raw | ((u64)priv0 << 32) | ((u64)trap << 40) | ((u64)size << 48);
    t = fold(t, cur_pc); t = fold(t, meta);
    WSP_RING(cur_pc); WSP_RING(meta); WSP_RING(0);
    const u32 c = s.chg;
    if ((c & CH_IREG) && s.ireg) WSP_PAIR(1, s.ireg, s.ival);
    if (c & CH_FREG) WSP_PAIR(2, s.freg, s.fval);
    if (s.vgroup) {          // one v line per register of the group (printTrace.cpp:424-441): the record carries a digest of its bytes
      for (u32 g = 0; g < s.vgroup; ++g) {
        const u32 reg = (s.vreg + g) & 31; u64 d = 0;
        for (u32 w = 0; w < vlenb / 8; ++w) d = fold(d, m.vld(reg * vlenb + 8 * w, 3));
        WSP_PAIR(3, reg, d);
      }
    }
    if (c & CH_MEM) WSP_PAIR(4, s.maddr, s.mval);
    const bool gen_lo = (c & CH_GEN) && s.gen_csr < 0x300;
    if (gen_lo) WSP_PAIR(5, s.gen_csr, csr_peek(s.gen_csr));
    if (c & CH_SSTATUS) WSP_PAIR(5, C_SSTATUS, csr_peek(C_SSTATUS));
    if (c & CH_SEPC) WSP_PAIR(5, C_SEPC, m.cold(CI_SEPC) & epc_rmask());
    if (c & CH_SCAUSE) WSP_PAIR(5, C_SCAUSE, m.cold(CI_SCAUSE));
    if (c & CH_STVAL) WSP_PAIR(5, C_STVAL, m.cold(CI_STVAL));
    if (c & CH_MSTATUS) WSP_PAIR(5, C_MSTATUS, csr_peek(C_MSTATUS));
    if ((c & CH_GEN) && !gen_lo) WSP_PAIR(5, s.gen_csr, csr_peek(s.gen_csr));
    if (c & CH_MEPC) WSP_PAIR(5, C_MEPC, m.cold(CI_MEPC) & epc_rmask());
    if (c & CH_MCAUSE) WSP_PAIR(5, C_MCAUSE, m.cold(CI_MCAUSE));
    if (c & CH_MTVAL) WSP_PAIR(5, C_MTVAL, m.cold(CI_MTVAL));
    if (c & CH_VL) WSP_PAIR(5, C_VL, m.cold(CI_VL));
    if (c & CH_VTYPE) WSP_PAIR(5, C_VTYPE, m.cold(CI_VTYPE));
    if (np == 0) WSP_PAIR(1, 0, 0);                                                               // printTrace.cpp:491-493: an "r x0 0" line
#undef WSP_PAIR
    if (rg) {
      // ring only (not part of the digest): the load / store address pair, then the bytes of every vector register reported
      // above (vlenb / 8 words each, in pair order) - what the text log prints beside the record
      const bool ldst = (c & CH_LDST) != 0;
      if (ldst) { WSP_RING2(s.lva); WSP_RING2(s.lpa); }
      for (u32 g = 0; g < s.vgroup; ++g) for (u32 w = 0; w < vlenb / 8; ++w) WSP_RING2(m.vld(((s.vreg + g) & 31) * vlenb + 8 * w, 3));
      if (kinds_at < pos) { rg[meta_at] |= ((u64)np << 56) | ((u64)ldst << 55); rg[kinds_at] = kinds; }
      *ring_pos_ = pos;
    }
#undef WSP_RING
#undef WSP_RING2
    h.thash = t;
  }
};

}  // namespace wsp
