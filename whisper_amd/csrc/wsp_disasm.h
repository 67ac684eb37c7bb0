// whisper_amd: disassembler for the text column of the commit log (printTrace.cpp:302-517 prints what Disassembler.cpp
// produces).  Own implementation over the instruction word: covers the modelled ISA (RV32 / RV64 I M A F D C Zicsr + privileged,
// the RVV slice of wsp_core.h) and prints "illegal" for everything else.  The column conventions are the reference's
// (Disassembler.cpp:51-325: mnemonic field widths, hex / decimal immediates, ". + 0x.." branch targets, "(x5)" atomics, rounding
// mode suffixes); tests/test_disasm.py checks it word for word against the reference's own Disassembler class.
// Host code only (plain C++): included by wsp_lib.hip, and compiled on its own by the test.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

namespace wspdis {

#include "wsp_disasm_tables.inc"      // generated data: CSR names, OP-V mnemonic / operand patterns (tools/gen_disasm_tables.py)

inline const char* csr_name(unsigned n, char* tmp)
{
  for (const auto& e : WSP_CSR_NAMES) if (e.num == n) return e.name;
  snprintf(tmp, 24, "c%u", n);                       // Disassembler.hpp:69-77: no name known
  return tmp;
}

struct Out {
  std::string s;
  void pad(const char* name, unsigned w) { s += name; for (size_t k = strlen(name); k < w; ++k) s += ' '; }
  void x(unsigned r) { s += 'x'; s += std::to_string(r); }
  void f(unsigned r) { s += 'f'; s += std::to_string(r); }
  void v(unsigned r) { s += 'v'; s += std::to_string(r); }
  void hex(uint32_t v) { char b[16]; snprintf(b, sizeof b, "0x%x", v); s += b; }
  void shex(int32_t v) { if (v < 0) { s += '-'; hex((uint32_t)(-(int64_t)v)); } else hex((uint32_t)v); }
  void dec(int64_t v) { s += std::to_string(v); }
  void rel(int32_t off, uint32_t mask = 0xffffffffu) { s += ". "; s += off < 0 ? '-' : '+'; s += ' '; hex((uint32_t)(off < 0 ? -(int64_t)off : off) & mask); }
};

// generic "name op, op, op" line (Disassembler.cpp:75-122): the mnemonic in a field of max(9, len + 1)
inline void gen(Out& o, const char* name) { const unsigned w = (unsigned)strlen(name) + 1; o.pad(name, w < 9 ? 9 : w); }
// "inst reg, imm(reg)" (Disassembler.cpp:51-71)
inline void ldst(Out& o, const char* name, bool fp, unsigned rd, unsigned rs1, int32_t imm)
{
  o.pad(name, 8); o.s += ' ';
  if (fp) o.f(rd); else o.x(rd);
  o.s += ", ";
  if (imm < 0) { o.s += '-'; imm = -imm; }
  o.hex((uint32_t)imm & 0xfff); o.s += '('; o.x(rs1); o.s += ')';
}
inline void regimm(Out& o, const char* name, unsigned r, int32_t imm) { o.pad(name, 8); o.s += ' '; o.x(r); o.s += ", "; o.shex(imm); }
inline const char* rm_name(unsigned rm) { static const char* t[8] = { "rne", "rtz", "rdn", "rup", "rmm", "inv1", "inv2", "dyn" }; return t[rm & 7]; }
inline int32_t sx(uint32_t v, unsigned bits) { return (int32_t)(v << (32 - bits)) >> (32 - bits); }

inline bool dis16(Out& o, uint32_t c, bool rv64)
{
  const uint32_t q = c & 3, f3 = (c >> 13) & 7, rdp = 8 + ((c >> 2) & 7), rs1p = 8 + ((c >> 7) & 7), rd = (c >> 7) & 31, rs2 = (c >> 2) & 31, b12 = (c >> 12) & 1;
  const int32_t imm6 = sx((b12 << 5) | ((c >> 2) & 31), 6);
  const uint32_t ud = (((c >> 10) & 7) << 3) | (((c >> 5) & 3) << 6), uw = (((c >> 10) & 7) << 3) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 6);
  if (q == 0) {
    switch (f3) {
      case 0: {
        const uint32_t imm = (((c >> 11) & 3) << 4) | (((c >> 7) & 0xf) << 6) | (((c >> 6) & 1) << 2) | (((c >> 5) & 1) << 3);
        if (c == 0 || imm == 0) return false;
        regimm(o, "c.addi4spn ", rdp, (int32_t)imm >> 2); return true; }
      case 1: ldst(o, "c.fld", true, rdp, rs1p, (int32_t)ud); return true;
      case 2: ldst(o, "c.lw", false, rdp, rs1p, (int32_t)uw); return true;
      case 3: ldst(o, rv64 ? "c.ld" : "c.flw", false, rdp, rs1p, (int32_t)(rv64 ? ud : uw)); return true;   // (c.flw / c.fsw print integer register names there)
      case 5: ldst(o, "c.fsd", true, rdp, rs1p, (int32_t)ud); return true;
      case 6: ldst(o, "c.sw", false, rdp, rs1p, (int32_t)uw); return true;
      case 7: if (rv64) ldst(o, "c.sd", false, rdp, rs1p, (int32_t)ud); else ldst(o, "c.fsw", false, rdp, rs1p, (int32_t)uw); return true;
      default: return false;
    }
  }
  if (q == 1) {
    switch (f3) {
      case 0: if (rd == 0) o.s += "c.nop"; else regimm(o, "c.addi", rd, imm6); return true;
      case 1:
        if (rv64) { if (rd == 0) return false; regimm(o, "c.addiw", rd, imm6); return true; }
        { const int32_t off = sx((b12 << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) | (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
          o.s += "c.jal    "; o.rel(off); return true; }
      case 2: regimm(o, "c.li", rd, imm6); return true;
      case 3:
        if (imm6 == 0) return false;
        if (rd == 2) {
          int32_t imm = sx((b12 << 9) | (((c >> 6) & 1) << 4) | (((c >> 5) & 1) << 6) | (((c >> 3) & 3) << 7) | (((c >> 2) & 1) << 5), 10);
          o.s += "c.addi16sp "; if (imm < 0) { o.s += '-'; imm = -imm; } o.hex((uint32_t)imm >> 4); return true;
        }
        regimm(o, "c.lui", rd, (int32_t)(((uint32_t)imm6 << 12) >> 12)); return true;
      case 4: {
        const uint32_t f2 = (c >> 10) & 3, sh = (b12 << 5) | ((c >> 2) & 31), k = (c >> 5) & 3;
        if (!rv64 && b12 && f2 < 2) return false;
        if (f2 == 0) { regimm(o, "c.srli", rs1p, (int32_t)sh); return true; }
        if (f2 == 1) { regimm(o, "c.srai", rs1p, (int32_t)sh); return true; }
        if (f2 == 2) { regimm(o, "c.andi", rs1p, imm6); return true; }
        static const char* n0[4] = { "c.sub    ", "c.xor    ", "c.or     ", "c.and    " };
        if (!b12) { o.s += n0[k]; o.x(rs1p); o.s += ", "; o.x(rdp); return true; }
        if (!rv64 || k > 1) return false;
        o.s += k == 0 ? "c.subw   " : "c.addw   "; o.x(rs1p); o.s += ", "; o.x(rdp); return true; }
      case 5: {
        const int32_t off = sx((b12 << 11) | (((c >> 11) & 1) << 4) | (((c >> 9) & 3) << 8) | (((c >> 8) & 1) << 10) | (((c >> 7) & 1) << 6) | (((c >> 6) & 1) << 7) | (((c >> 3) & 7) << 1) | (((c >> 2) & 1) << 5), 12);
        o.s += "c.j      "; o.rel(off); return true; }
      default: {
        const int32_t off = sx((b12 << 8) | (((c >> 10) & 3) << 3) | (((c >> 5) & 3) << 6) | (((c >> 3) & 3) << 1) | (((c >> 2) & 1) << 5), 9);
        o.pad(f3 == 6 ? "c.beqz" : "c.bnez", 8); o.s += ' '; o.x(rs1p); o.s += ", "; o.rel(off); return true; }
    }
  }
  if (q == 2) {
    switch (f3) {
      case 0: if (!rv64 && b12) return false; o.s += "c.slli   "; o.x(rd); o.s += ", "; o.dec((b12 << 5) | rs2); return true;
      case 1: o.s += "c.fldsp   "; o.f(rd); o.s += ", "; o.hex((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6)); return true;
      case 2: if (rd == 0) return false; o.s += "c.lwsp   "; o.x(rd); o.s += ", "; o.hex((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6)); return true;
      case 3:
        if (!rv64) { o.s += "c.flwsp   "; o.f(rd); o.s += ", "; o.hex((b12 << 5) | (((c >> 4) & 7) << 2) | (((c >> 2) & 3) << 6)); return true; }
        if (rd == 0) return false;
        o.s += "c.ldsp   "; o.x(rd); o.s += ", "; o.hex((b12 << 5) | (((c >> 5) & 3) << 3) | (((c >> 2) & 7) << 6)); return true;
      case 4:
        if (!b12) {
          if (rs2 == 0) { if (rd == 0) return false; o.s += "c.jr     "; o.x(rd); return true; }
          o.s += "c.mv     "; o.x(rd); o.s += ", "; o.x(rs2); return true;
        }
        if (rs2 == 0) { if (rd == 0) { o.s += "c.ebreak"; return true; } o.s += "c.jalr   "; o.x(rd); return true; }
        o.s += "c.add    "; o.x(rd); o.s += ", "; o.x(rs2); return true;
      case 5: o.s += "c.fsdsp   "; o.f(rs2); o.s += ", "; o.hex((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6)); return true;
      case 6: o.s += "c.swsp   "; o.x(rs2); o.s += ", "; o.hex((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6)); return true;
      default:
        if (!rv64) { o.s += "c.swsp   "; o.f(rs2); o.s += ", "; o.hex((((c >> 9) & 0xf) << 2) | (((c >> 7) & 3) << 6)); return true; }   // (c.fswsp prints under c.swsp's name, Disassembler.cpp:817-820)
        o.s += "c.sdsp   "; o.x(rs2); o.s += ", "; o.hex((((c >> 10) & 7) << 3) | (((c >> 7) & 7) << 6)); return true;
    }
  }
  return false;
}

inline bool dis_vec(Out& o, uint32_t inst);

inline bool dis32(Out& o, uint32_t inst, bool rv64)
{
  const uint32_t op = inst & 0x7f, rd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, rs1 = (inst >> 15) & 31, rs2 = (inst >> 20) & 31, f7 = inst >> 25;
  const int32_t immi = (int32_t)inst >> 20;
  auto rrr = [&](const char* n) { gen(o, n); o.x(rd); o.s += ", "; o.x(rs1); o.s += ", "; o.x(rs2); return true; };
  auto rri = [&](const char* n, int64_t imm) { gen(o, n); o.x(rd); o.s += ", "; o.x(rs1); o.s += ", "; o.dec(imm); return true; };
  auto rr = [&](const char* n) { gen(o, n); o.x(rd); o.s += ", "; o.x(rs1); return true; };
  const uint32_t imm12 = inst >> 20, top5 = imm12 >> 7, amt = imm12 & 0x7f, amax = rv64 ? 63 : 31;
  switch (op) {
    case 0x37: regimm(o, "lui", rd, (int32_t)inst >> 12); return true;
    case 0x17: o.s += "auipc    "; o.x(rd); o.s += ", "; o.hex(inst >> 12); return true;
    case 0x6f: {
      const int32_t off = ((int32_t)(inst & 0x80000000u) >> 11) | (int32_t)(inst & 0xff000) | (int32_t)((inst >> 9) & 0x800) | (int32_t)((inst >> 20) & 0x7fe);
      if (rd == 0) o.s += "j        "; else { o.s += "jal      "; o.x(rd); o.s += ", "; }
      o.rel(off, 0xfffff); return true; }
    case 0x67: if (f3 != 0) return false; ldst(o, "jalr", false, rd, rs1, immi); return true;
    case 0x63: {
      static const char* n[8] = { "beq", "bne", nullptr, nullptr, "blt", "bge", "bltu", "bgeu" };
      if (!n[f3]) return false;
      const int32_t off = ((int32_t)(inst & 0x80000000u) >> 19) | (int32_t)((inst & 0x80) << 4) | (int32_t)((inst >> 20) & 0x7e0) | (int32_t)((inst >> 7) & 0x1e);
      o.pad(n[f3], 8); o.s += ' '; o.x(rs1); o.s += ", "; o.x(rs2); o.s += ", "; o.rel(off); return true; }
    case 0x03: {
      static const char* n[8] = { "lb", "lh", "lw", "ld", "lbu", "lhu", "lwu", nullptr };
      if (!n[f3]) return false;
      ldst(o, n[f3], false, rd, rs1, immi); return true; }
    case 0x23: {
      static const char* n[4] = { "sb", "sh", "sw", "sd" };
      if (f3 > 3) return false;
      ldst(o, n[f3], false, rs2, rs1, ((int32_t)(inst & 0xfe000000u) >> 20) | (int32_t)((inst >> 7) & 31)); return true; }
    case 0x13: {
      const uint32_t top6 = inst >> 26, sh = (inst >> 20) & 63;
      switch (f3) {
        case 0: return rri("addi", immi);
        case 1:
          if (top6 == 0 && (rv64 || sh < 32)) return rri("slli", sh);
          if (top5 == 0x0c) { static const char* n[6] = { "clz", "ctz", "cpop", nullptr, "sext.b", "sext.h" }; if (amt < 6 && n[amt]) return rr(n[amt]); return false; }
          if ((top5 == 5 || top5 == 9 || top5 == 0x0d) && amt <= amax) return rri(top5 == 5 ? "bseti" : top5 == 9 ? "bclri" : "binvi", amt);
          return false;
        case 2: return rri("slti", immi);
        case 3: return rri("sltiu", immi);
        case 4: return rri("xori", immi);
        case 5:
          if ((top6 == 0 || top6 == 0x10) && (rv64 || sh < 32)) return rri(top6 ? "srai" : "srli", sh);
          if (top5 == 5 && amt == 7) return rr("orc.b");
          if ((top5 == 9 || top5 == 0x0c) && amt <= amax) return rri(top5 == 9 ? "bexti" : "rori", amt);
          if (imm12 == (rv64 ? 0x6b8u : 0x698u)) return rr(rv64 ? "rev8" : "rev8_32");
          return false;
        case 6: return rri("ori", immi);
        default: return rri("andi", immi);
      } }
    case 0x1b:
      if (f3 == 0) return rri("addiw", immi);
      if (f3 == 1 && f7 == 0) return rri("slliw", rs2);
      if (f3 == 5 && f7 == 0) return rri("srliw", rs2);
      if (f3 == 5 && f7 == 0x20) return rri("sraiw", rs2);
      if (f3 == 1 && (imm12 >> 6) == 2) return rri("slli.uw", amt);
      if (f3 == 1 && top5 == 0x0c && amt <= 2) return rr(amt == 0 ? "clzw" : amt == 1 ? "ctzw" : "cpopw");
      if (f3 == 5 && f7 == 0x30) return rri("roriw", rs2);
      return false;
    case 0x33: {
      static const char* n0[8] = { "add", "sll", "slt", "sltu", "xor", "srl", "or", "and" };
      static const char* n1[8] = { "mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu" };
      if (f7 == 0) return rrr(n0[f3]);
      if (f7 == 1) return rrr(n1[f3]);
      if (f7 == 0x20 && f3 == 0) return rrr("sub");
      if (f7 == 0x20 && f3 == 5) return rrr("sra");
      if (f7 == 0x20 && (f3 == 4 || f3 >= 6)) return rrr(f3 == 4 ? "xnor" : f3 == 6 ? "orn" : "andn");
      if (f7 == 0x04 && f3 == 4) return !rv64 && rs2 == 0 ? rr("zext.h") : rrr("pack");
      if (f7 == 0x05 && f3 != 0) { static const char* n[8] = { nullptr, "clmul", "clmulr", "clmulh", "min", "minu", "max", "maxu" }; return rrr(n[f3]); }
      if (f7 == 0x07 && (f3 == 5 || f3 == 7)) return rrr(f3 == 5 ? "czero.eqz" : "czero.nez");
      if (f7 == 0x10 && (f3 == 2 || f3 == 4 || f3 == 6)) return rrr(f3 == 2 ? "sh1add" : f3 == 4 ? "sh2add" : "sh3add");
      if (f7 == 0x14 && f3 == 1) return rrr("bset");
      if (f7 == 0x24 && (f3 == 1 || f3 == 5)) return rrr(f3 == 1 ? "bclr" : "bext");
      if (f7 == 0x30 && (f3 == 1 || f3 == 5)) return rrr(f3 == 1 ? "rol" : "ror");
      if (f7 == 0x34 && f3 == 1) return rrr("binv");
      return false; }
    case 0x3b: {
      if (f7 == 0) { if (f3 == 0) return rrr("addw"); if (f3 == 1) return rrr("sllw"); if (f3 == 5) return rrr("srlw"); return false; }
      if (f7 == 0x20) { if (f3 == 0) return rrr("subw"); if (f3 == 5) return rrr("sraw"); return false; }
      if (f7 == 1) { static const char* n[8] = { "mulw", nullptr, nullptr, nullptr, "divw", "divuw", "remw", "remuw" }; if (!n[f3]) return false; return rrr(n[f3]); }
      if (f7 == 0x04 && f3 == 0) return rrr("add.uw");
      if (f7 == 0x04 && f3 == 4) return rs2 == 0 ? rr("zext.h") : rrr("packw");
      if (f7 == 0x10 && (f3 == 2 || f3 == 4 || f3 == 6)) return rrr(f3 == 2 ? "sh1add.uw" : f3 == 4 ? "sh2add.uw" : "sh3add.uw");
      if (f7 == 0x30 && (f3 == 1 || f3 == 5)) return rrr(f3 == 1 ? "rolw" : "rorw");
      return false; }
    case 0x0f: {
      if (f3 == 1) { gen(o, "fence.i"); return true; }
      if (f3 != 0) return false;
      const bool tso = (inst >> 28) == 8;
      o.pad(tso ? "fence.tso" : "fence", 8); o.s += ' ';
      std::string p, s; const uint32_t pr = (inst >> 24) & 15, su = (inst >> 20) & 15;
      if (pr & 2) p += 'r'; if (pr & 1) p += 'w'; if (pr & 8) p += 'i'; if (pr & 4) p += 'o';
      if (su & 2) s += 'r'; if (su & 1) s += 'w'; if (su & 8) s += 'i'; if (su & 4) s += 'o';
      if (p.empty() && s.empty()) return true;
      if (p.empty()) p = "0"; if (s.empty()) s = "0";
      o.s += p + ", " + s; return true; }
    case 0x73: {
      if (f3 == 0) {
        if (rd != 0) return false;
        if (inst == 0x00000073) { gen(o, "ecall"); return true; }
        if (inst == 0x00100073) { gen(o, "ebreak"); return true; }
        if (inst == 0x30200073) { gen(o, "mret"); return true; }
        if (inst == 0x10200073) { gen(o, "sret"); return true; }
        if (inst == 0x10500073) { gen(o, "wfi"); return true; }
        if (f7 == 0x09) { gen(o, "sfence.vma"); o.x(rs1); o.s += ", "; o.x(rs2); return true; }
        return false;
      }
      if (f3 == 4) return false;
      static const char* n[8] = { nullptr, "csrrw", "csrrs", "csrrc", nullptr, "csrrwi", "csrrsi", "csrrci" };
      char tmp[24];
      o.pad(n[f3], 9); o.x(rd); o.s += ", "; o.s += csr_name(inst >> 20, tmp); o.s += ", ";
      if (f3 & 4) o.hex(rs1); else o.x(rs1);
      return true; }
    case 0x2f: {
      const uint32_t f5 = inst >> 27; const bool aq = (inst >> 26) & 1, rl = (inst >> 25) & 1;
      if (f3 != 2 && f3 != 3) return false;
      const char* sfx = f3 == 2 ? ".w" : ".d";
      const char* nm = nullptr;
      switch (f5) { case 0: nm = "amoadd"; break; case 1: nm = "amoswap"; break; case 2: nm = "lr"; break; case 3: nm = "sc"; break; case 4: nm = "amoxor"; break;
        case 8: nm = "amoor"; break; case 12: nm = "amoand"; break; case 16: nm = "amomin"; break; case 20: nm = "amomax"; break; case 24: nm = "amominu"; break; case 28: nm = "amomaxu"; break; default: return false; }
      if (f5 == 2 && rs2 != 0) return false;
      o.s += nm; o.s += sfx; if (aq) o.s += ".aq"; if (rl) o.s += ".rl";
      o.s += ' '; o.x(rd); o.s += ", ";
      if (f5 != 2) { o.x(rs2); o.s += ", "; }
      o.s += '('; o.x(rs1); o.s += ')';
      return true; }
    case 0x07:
      if (f3 == 2) { ldst(o, "flw", true, rd, rs1, immi); return true; }
      if (f3 == 3) { ldst(o, "fld", true, rd, rs1, immi); return true; }
      if (f3 == 0 || f3 >= 5) return dis_vec(o, inst);
      return false;
    case 0x27: {
      const int32_t imms = ((int32_t)(inst & 0xfe000000u) >> 20) | (int32_t)((inst >> 7) & 31);
      if (f3 == 2) { ldst(o, "fsw", true, rs2, rs1, imms); return true; }
      if (f3 == 3) { ldst(o, "fsd", true, rs2, rs1, imms); return true; }
      if (f3 == 0 || f3 >= 5) return dis_vec(o, inst);
      return false; }
    case 0x43: case 0x47: case 0x4b: case 0x4f: {
      const uint32_t fmt = (inst >> 25) & 3;
      if (fmt > 1) return false;
      static const char* n[4] = { "fmadd", "fmsub", "fnmsub", "fnmadd" };
      std::string nm = std::string(n[(op - 0x43) >> 2]) + (fmt ? ".d" : ".s");
      gen(o, nm.c_str()); o.f(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.f(rs2); o.s += ", "; o.f(inst >> 27); o.s += ", "; o.s += rm_name(f3);
      return true; }
    case 0x53: {
      if (f7 & 2) return false;
      const bool d = f7 & 1; const char* sf = d ? ".d" : ".s";
      auto fff = [&](const char* n, bool rm) { std::string nm = std::string(n) + sf; gen(o, nm.c_str()); o.f(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.f(rs2); if (rm) { o.s += ", "; o.s += rm_name(f3); } return true; };
      switch (f7 >> 2) {
        case 0: return fff("fadd", true);
        case 1: return fff("fsub", true);
        case 2: return fff("fmul", true);
        case 3: return fff("fdiv", true);
        case 0x0b: { if (rs2 != 0) return false; std::string nm = std::string("fsqrt") + sf; gen(o, nm.c_str()); o.f(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.s += rm_name(f3); return true; }
        case 4: { static const char* n[3] = { "fsgnj", "fsgnjn", "fsgnjx" }; if (f3 > 2) return false; return fff(n[f3], false); }
        case 5: { if (f3 > 1) return false; return fff(f3 ? "fmax" : "fmin", false); }
        case 8: {
          if (!d && rs2 == 1) { gen(o, "fcvt.s.d"); o.f(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.s += rm_name(f3); return true; }
          if (d && rs2 == 0) { gen(o, "fcvt.d.s"); o.f(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.s += rm_name(f3); return true; }
          return false; }
        case 0x14: { static const char* n[3] = { "fle", "flt", "feq" }; if (f3 > 2) return false; std::string nm = std::string(n[f3]) + sf; gen(o, nm.c_str()); o.x(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.f(rs2); return true; }
        case 0x18: { static const char* n[4] = { "fcvt.w", "fcvt.wu", "fcvt.l", "fcvt.lu" }; if (rs2 > 3) return false; std::string nm = std::string(n[rs2]) + sf; gen(o, nm.c_str()); o.x(rd); o.s += ", "; o.f(rs1); o.s += ", "; o.s += rm_name(f3); return true; }
        case 0x1a: { static const char* n[4] = { ".w", ".wu", ".l", ".lu" }; if (rs2 > 3) return false; std::string nm = std::string("fcvt") + sf + n[rs2]; gen(o, nm.c_str()); o.f(rd); o.s += ", "; o.x(rs1); o.s += ", "; o.s += rm_name(f3); return true; }
        case 0x1c:
          if (rs2 != 0) return false;
          if (f3 == 0) { gen(o, d ? "fmv.x.d" : "fmv.x.w"); o.x(rd); o.s += ", "; o.f(rs1); return true; }
          if (f3 == 1) { gen(o, d ? "fclass.d" : "fclass.s"); o.x(rd); o.s += ", "; o.f(rs1); return true; }
          return false;
        case 0x1e: if (rs2 != 0 || f3 != 0) return false; gen(o, d ? "fmv.d.x" : "fmv.w.x"); o.f(rd); o.s += ", "; o.x(rs1); return true;
        default: return false;
      } }
    case 0x57: return dis_vec(o, inst);
    default: return false;
  }
}

// RVV (Disassembler.cpp:327-445): "name vd, operands[, v0.t]"; loads / stores "name vd, (rs1)[, rs2 | vs2][, v0.t]"
inline bool dis_vec(Out& o, uint32_t inst)
{
  const uint32_t op = inst & 0x7f, vd = (inst >> 7) & 31, f3 = (inst >> 12) & 7, s1 = (inst >> 15) & 31, vs2 = (inst >> 20) & 31, vm = (inst >> 25) & 1, f6 = inst >> 26;
  auto mask = [&]() { if (!vm) o.s += ", v0.t"; return true; };
  if (op == 0x07 || op == 0x27) {
    const bool st = op == 0x27;
    const uint32_t mop = (inst >> 26) & 3, lumop = vs2, nf = inst >> 29, mew = (inst >> 28) & 1;
    static const char* ew0[8] = { "8", nullptr, nullptr, nullptr, nullptr, "16", "32", "64" };
    static const char* ew1[8] = { "128", nullptr, nullptr, nullptr, nullptr, "256", "512", "1024" };
    const char* const* ew = mew ? ew1 : ew0;
    if (!ew[f3]) return false;
    std::string nm;
    const std::string seg = nf ? "seg" + std::to_string(nf + 1) : "";
    if (mop == 0) {
      if (lumop == 0) nm = std::string(st ? "vs" : "vl") + seg + "e" + ew[f3] + ".v";
      else if (((lumop == 8 && st) || lumop == 0xb) && mew) return false;         // whole-register stores and the mask forms have no wide (mew) encodings
      else if (lumop == 8) { if (st) { if (f3 != 0 || (nf != 0 && nf != 1 && nf != 3 && nf != 7)) return false; nm = "vs" + std::to_string(nf + 1) + "r.v"; } else nm = "vl" + std::to_string(nf + 1) + "re" + ew[f3] + ".v"; }
      else if (lumop == 0xb) { if (nf || f3) return false; nm = st ? "vsm.v" : "vlm.v"; }
      else if (lumop == 0x10 && !st) nm = std::string("vl") + seg + "e" + ew[f3] + "ff.v";
      else return false;
      o.s += nm; o.s += ' '; o.v(vd); o.s += ", ("; o.x(s1); o.s += ')';
      return mask();
    }
    if (mop == 2) { nm = std::string(st ? "vss" : "vls") + seg + "e" + ew[f3] + ".v"; o.s += nm; o.s += ' '; o.v(vd); o.s += ", ("; o.x(s1); o.s += "), "; o.x(vs2); return mask(); }
    nm = std::string(st ? "vs" : "vl") + (mop == 1 ? "ux" : "ox") + seg + "ei" + ew[f3] + ".v";
    o.s += nm; o.s += ' '; o.v(vd); o.s += ", ("; o.x(s1); o.s += "), "; o.v(vs2);
    return mask();
  }
  if (f3 == 7) {
    if (!(inst >> 31)) {
      o.s += "vsetvli "; o.x(vd); o.s += ", "; o.x(s1); o.s += ", ";
      const uint32_t vt = (inst >> 20) & 0x7ff;
      static const char* ews[8] = { "e8", "e16", "e32", "e64", "e128", "e256", "e512", "e1024" };
      static const char* gms[8] = { "m1", "m2", "m4", "m8", "m?", "mf8", "mf4", "mf2" };
      o.s += ews[(vt >> 3) & 7]; o.s += ','; o.s += gms[vt & 7]; o.s += ','; o.s += ((vt >> 6) & 1) ? "ta" : "tu"; o.s += ','; o.s += ((vt >> 7) & 1) ? "ma" : "mu";
      return true;
    }
    if ((inst >> 30) == 3) {
      o.s += "vsetivli "; o.x(vd); o.s += ", "; o.dec(s1); o.s += ", ";
      const uint32_t vt = (inst >> 20) & 0x3ff;
      static const char* ews[8] = { "e8", "e16", "e32", "e64", "e128", "e256", "e512", "e1024" };
      static const char* gms[8] = { "m1", "m2", "m4", "m8", "m?", "mf8", "mf4", "mf2" };
      o.s += ews[(vt >> 3) & 7]; o.s += ','; o.s += gms[vt & 7]; o.s += ','; o.s += ((vt >> 6) & 1) ? "ta" : "tu"; o.s += ','; o.s += ((vt >> 7) & 1) ? "ma" : "mu";
      return true;
    }
    if ((inst >> 25) == 0x40) { o.s += "vsetvl "; o.x(vd); o.s += ", "; o.x(s1); o.s += ", "; o.x(vs2); return true; }
    return false;
  }
  // arithmetic: the pattern of this encoding point - keyed on the vs1 field, on the vs2 field, or on neither
  const unsigned base = f3 | (f6 << 3) | (vm << 9);
  const char* pat = nullptr;
  for (unsigned sel : { (1u + s1) << 10, (33u + vs2) << 10, 0u }) {
    for (const auto& e : WSP_VEC_PATS) if (e.key == (base | sel)) { pat = e.pat; break; }
    if (pat) break;
  }
  if (!pat) return false;
  for (const char* p = pat; *p; ++p) {
    if (*p != '%') { o.s += *p; continue; }
    switch (*++p) {
      case 'd': o.v(vd); break; case 'D': o.x(vd); break; case 'F': o.f(vd); break;
      case '2': o.v(vs2); break; case 'X': o.x(vs2); break; case 'G': o.f(vs2); break;
      case 'v': o.v(s1); break; case 'x': o.x(s1); break; case 'f': o.f(s1); break;
      case 'i': o.dec(sx(s1, 5)); break;
      case 'u': o.dec(f3 == 3 && (f6 == 0x14 || f6 == 0x15) ? (s1 | ((f6 & 1) << 5)) : s1); break;      // vror.vi: a 6-bit amount, its top bit in funct6
      case 'm': o.s += "v0.t"; break; case 'M': o.s += "v0"; break;
      default: break;
    }
  }
  return true;
}

// The disassembly column of one instruction word as fetched (16-bit words are the low half).
inline std::string disassemble(uint32_t inst, bool rv64)
{
  Out o;
  const bool ok = (inst & 3) == 3 ? dis32(o, inst, rv64) : dis16(o, inst & 0xffff, rv64);
  return ok ? o.s : std::string("illegal");
}

}  // namespace wspdis
