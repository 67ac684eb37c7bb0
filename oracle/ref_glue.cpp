// TEST INFRASTRUCTURE ONLY.  Thin extern "C" shim over the reference's own Decoder and
// Disassembler classes (compiled from /root/reference where they lie, see oracle/Makefile) so
// that Python tests can ask "what does the reference decode / print for this word?".
// Nothing here is product code and nothing here is reference code: it only calls the public
// API declared in /root/reference/Decoder.hpp:45 and /root/reference/Disassembler.hpp:57.
#include <cstdint>
#include <cstring>
#include <string>
#include "Decoder.hpp"
#include "DecodedInst.hpp"
#include "Disassembler.hpp"
extern "C" {
#include "softfloat.h"
}

using namespace WdRiscv;
extern "C" int ref_csr_name(unsigned num, char* buf, int n);

static Decoder& dec(int rv64)
{
  static Decoder d64, d32;
  static bool init = false;
  if (!init) { d64.enableRv64(true); d32.enableRv64(false); init = true; }
  return rv64 ? d64 : d32;
}

extern "C" {

// Reference decode (Decoder.cpp:2232). Returns InstId as unsigned; out[0..3]=op0..op3,
// out[4]=instruction size in bytes; name (optional) receives the mnemonic.
unsigned ref_decode(int rv64, uint32_t inst, uint32_t* out, char* name, int nameLen)
{
  uint32_t op0 = 0, op1 = 0, op2 = 0, op3 = 0;
  const InstEntry& e = dec(rv64).decode(inst, op0, op1, op2, op3);
  out[0] = op0; out[1] = op1; out[2] = op2; out[3] = op3;
  out[4] = ((inst & 3) == 3) ? 4 : 2;
  if (name && nameLen > 0) {
    std::string n(e.name());
    strncpy(name, n.c_str(), nameLen - 1);
    name[nameLen - 1] = 0;
  }
  return unsigned(e.instId());
}

// Reference disassembly (Disassembler.cpp) of one instruction word.
int ref_disassemble(int rv64, uint32_t inst, char* buf, int n)
{
  static Disassembler dis64, dis32;
  static bool init = false;
  if (!init) {
    dis64.enableRv64(true); dis32.enableRv64(false); init = true;
    // Hart.cpp wires the disassembler to its CSR file the same way (CSR names instead of cNNN).
    auto cb = [](unsigned ix) -> std::string_view {
      static thread_local char nm[64];
      return ref_csr_name(ix, nm, sizeof nm) ? std::string_view(nm) : std::string_view();
    };
    dis64.setCsrNameCallback(cb); dis32.setCsrNameCallback(cb);
  }
  std::string s;
  (rv64 ? dis64 : dis32).disassembleInst(inst, dec(rv64), s);
  strncpy(buf, s.c_str(), n - 1);
  buf[n - 1] = 0;
  return int(s.size());
}

// Reference 16->32 bit expansion (Decoder.cpp expandCompressedInst).
uint32_t ref_expand16(int rv64, uint16_t inst) { return dec(rv64).expandCompressedInst(inst); }

// Access to softfloat's thread-local globals from ctypes.
void     ref_sf_set_rm(int rm)   { softfloat_roundingMode = uint_fast8_t(rm); }
void     ref_sf_clear_flags()    { softfloat_exceptionFlags = 0; }
unsigned ref_sf_flags()          { return softfloat_exceptionFlags; }

}

// ---------------------------------------------------------------------------------------------
// The reference's CSR file (CsRegs.hpp:935, CsRegs.cpp) standing alone: names for the
// disassembler (what Hart does through setCsrNameCallback) and the architectural read/write/poke
// masks, so the oracle's CSR restatement can be checked field by field.
#include "CsRegs.hpp"

namespace {
// Most of CsRegs' configuration surface is protected (Hart is its friend); a derived class may
// re-export it without touching the reference.
struct OpenCsRegs : CsRegs<uint64_t> {
  using Base = CsRegs<uint64_t>;
  using Base::Base;
  using Base::peek; using Base::poke; using Base::reset; using Base::enableSupervisorMode; using Base::enableUserMode;
  using Base::enableRvc; using Base::enableRvf; using Base::enableZicntr; using Base::enableHypervisorMode; using Base::enableVector;
};
struct CsrBox {
  PmpManager pmp;
  PmaManager pma;
  OpenCsRegs regs;
  explicit CsrBox(uint64_t mem) : pma(mem), regs(pmp, pma) {}
};
CsrBox* g_names = nullptr;
}

extern "C" {

// Hart::processExtensions order for the CSR-relevant switches (Hart.cpp:553-621), then reset (Hart.cpp:948).
void* ref_csr_new(int rvs, int rvu, int rvf, int rvc, int zicntr)
{
  auto* b = new CsrBox(uint64_t(1) << 32);
  b->regs.enableSupervisorMode(rvs);
  b->regs.enableUserMode(rvu);
  b->regs.enableHypervisorMode(false);
  b->regs.enableRvc(rvc);
  b->regs.enableRvf(rvf);
  b->regs.enableVector(false);
  b->regs.enableZicntr(zicntr);
  b->regs.reset();
  return b;
}
void ref_csr_free(void* p) { delete static_cast<CsrBox*>(p); }
int ref_csr_read(void* p, unsigned num, unsigned priv, uint64_t* v)
{ return static_cast<CsrBox*>(p)->regs.read(CsrNumber(num), PrivilegeMode(priv), *v); }
int ref_csr_write(void* p, unsigned num, unsigned priv, uint64_t v)
{ return static_cast<CsrBox*>(p)->regs.write(CsrNumber(num), PrivilegeMode(priv), v); }
int ref_csr_peek(void* p, unsigned num, uint64_t* v)
{ return static_cast<CsrBox*>(p)->regs.peek(CsrNumber(num), *v); }
int ref_csr_poke(void* p, unsigned num, uint64_t v)
{ return static_cast<CsrBox*>(p)->regs.poke(CsrNumber(num), v); }
int ref_csr_implemented(void* p, unsigned num)
{ auto* c = static_cast<CsrBox*>(p)->regs.findCsr(CsrNumber(num)); return c && c->isImplemented(); }
int ref_csr_name(unsigned num, char* buf, int n)
{
  if (!g_names) g_names = new CsrBox(uint64_t(1) << 32);
  auto* c = g_names->regs.findCsr(CsrNumber(num));
  if (!c) return 0;
  std::string s(c->getName());
  strncpy(buf, s.c_str(), n - 1); buf[n - 1] = 0;
  return int(s.size());
}

}

// The same for System<uint32_t>: CsRegs<uint32_t> (RV32 presence, reset values and masks).
namespace {
struct OpenCsRegs32 : CsRegs<uint32_t> {
  using Base = CsRegs<uint32_t>;
  using Base::Base;
  using Base::peek; using Base::poke; using Base::reset; using Base::enableSupervisorMode; using Base::enableUserMode;
  using Base::enableRvc; using Base::enableRvf; using Base::enableZicntr; using Base::enableHypervisorMode; using Base::enableVector;
};
struct CsrBox32 {
  PmpManager pmp;
  PmaManager pma;
  OpenCsRegs32 regs;
  explicit CsrBox32(uint64_t mem) : pma(mem), regs(pmp, pma) {}
};
}
extern "C" {
void* ref_csr32_new(int rvs, int rvu, int rvf, int rvc, int zicntr)
{
  auto* b = new CsrBox32(uint64_t(1) << 32);
  b->regs.enableSupervisorMode(rvs);
  b->regs.enableUserMode(rvu);
  b->regs.enableHypervisorMode(false);
  b->regs.enableRvc(rvc);
  b->regs.enableRvf(rvf);
  b->regs.enableVector(false);
  b->regs.enableZicntr(zicntr);
  b->regs.reset();
  return b;
}
void ref_csr32_free(void* p) { delete static_cast<CsrBox32*>(p); }
int ref_csr32_read(void* p, unsigned num, unsigned priv, uint32_t* v)
{ return static_cast<CsrBox32*>(p)->regs.read(CsrNumber(num), PrivilegeMode(priv), *v); }
int ref_csr32_write(void* p, unsigned num, unsigned priv, uint32_t v)
{ return static_cast<CsrBox32*>(p)->regs.write(CsrNumber(num), PrivilegeMode(priv), v); }
int ref_csr32_peek(void* p, unsigned num, uint32_t* v)
{ return static_cast<CsrBox32*>(p)->regs.peek(CsrNumber(num), *v); }
int ref_csr32_implemented(void* p, unsigned num)
{ auto* c = static_cast<CsrBox32*>(p)->regs.findCsr(CsrNumber(num)); return c && c->isImplemented(); }
}

// ---------------------------------------------------------------------------------------------
// The reference's PmpManager (PmpManager.hpp, header only) driven the way Hart::updateMemoryProtection does
// (Hart.cpp:842-899): 8 pmpcfg values (pmpcfg0,2,..14) and 64 pmpaddr values in, permission bits out.
extern "C" {
void* ref_pmp_new(const uint64_t* cfg8, const uint64_t* addr64)
{
  auto* m = new PmpManager();
  for (unsigned ix = 0; ix < 64; ++ix) {
    unsigned config = unsigned(cfg8[ix >> 3] >> (8 * (ix & 7))) & 0xff;
    Pmp pmp; uint64_t low = 0, high = 0;
    if (m->unpackMemoryProtection(config, addr64[ix], ix ? addr64[ix - 1] : 0, false, pmp, low, high) and pmp.type() != Pmp::Type::Off)
      m->defineRegion(low, high, pmp, ix);
  }
  m->enable(true);
  return m;
}
void ref_pmp_free(void* p) { delete static_cast<PmpManager*>(p); }
unsigned ref_pmp_perm(void* p, unsigned priv, uint64_t addr)
{
  auto pmp = static_cast<PmpManager*>(p)->accessPmp(PrivilegeMode(priv), addr);
  return unsigned(pmp.isRead()) | (unsigned(pmp.isWrite()) << 1) | (unsigned(pmp.isExec()) << 2);
}
uint64_t ref_pmp_legalize_cfg(uint64_t prev, uint64_t next) { PmpManager m; return m.legalizePmpcfg<uint64_t>(prev, next); }
}

// ---------------------------------------------------------------------------------------------
// The reference's VirtMem + Tlb (virtual_memory/VirtMem.cpp, Tlb.cpp) standing alone over a flat RAM buffer, configured the
// way Hart does it (Hart.cpp:326-468 callbacks, 904-919 satp, 1167-1174 SUM/MXR, 12135-12173 sfence.vma).  Hart reaches the
// TLB and the setters as a friend; this test TU opens them the blunt way, for itself only.
#define private public
#define protected public
#include "virtual_memory/VirtMem.hpp"
#undef private
#undef protected

namespace {
struct VmBox {
  VirtMem vm;
  uint8_t* mem; uint64_t base, size;
  VmBox(unsigned tlbSize, uint8_t* m, uint64_t b, uint64_t s) : vm(0, 4096, tlbSize), mem(m), base(b), size(s)
  {
    auto inside = [this](uint64_t a, unsigned n) { return a >= base && a - base <= size - n; };
    vm.setMemReadCallback([this, inside](uint64_t a, bool, unsigned n, uint64_t& d) { if (!inside(a, n)) return false; d = 0; memcpy(&d, mem + (a - base), n); return true; });
    vm.setMemWriteCallback([this, inside](uint64_t a, bool, unsigned n, uint64_t d) { if (!inside(a, n)) return false; memcpy(mem + (a - base), &d, n); return true; });
    vm.setIsReadableCallback([](uint64_t) { return true; });
    vm.setIsWritableCallback([](uint64_t) { return true; });
    vm.setFaultOnFirstAccess(true);      // no Svadu (Hart.hpp:539-549)
  }
};
}

extern "C" {
void* ref_vm_new(unsigned tlbSize, uint8_t* mem, uint64_t base, uint64_t size) { return new VmBox(tlbSize, mem, base, size); }
void ref_vm_free(void* p) { delete static_cast<VmBox*>(p); }
void ref_vm_config(void* p, uint64_t satp, int sum, int mxr)
{
  auto& vm = static_cast<VmBox*>(p)->vm;
  vm.configTranslation(VirtMem::Mode(satp >> 60), uint32_t((satp >> 44) & 0xffff), satp & 0xfffffffffffULL);
  vm.setSum(sum); vm.setExecReadable(mxr);
}
// acc: 1 read, 2 write, 4 exec.  Returns the ExceptionCause number (0 = none).
unsigned ref_vm_translate(void* p, uint64_t va, unsigned priv, unsigned acc, uint64_t* pa)
{
  auto& vm = static_cast<VmBox*>(p)->vm;
  uint64_t gpa = va;
  auto c = vm.translate(va, PrivilegeMode(priv), false, acc == 1, acc == 2, acc == 4, gpa, *pa);
  return c == ExceptionCause::NONE ? 0 : unsigned(c);
}
unsigned ref_vm_translate_noupdate(void* p, uint64_t va, unsigned priv, unsigned acc, uint64_t* pa)
{
  auto& vm = static_cast<VmBox*>(p)->vm;
  auto c = vm.transAddrNoUpdate(va, PrivilegeMode(priv), false, acc == 1, acc == 2, acc == 4, *pa);
  return c == ExceptionCause::NONE ? 0 : unsigned(c);
}
// Hart installs a Memory write observer that drops cached PTEs overlapped by a store (Memory.hpp:148,395-398)
void ref_vm_ram_written(void* p, uint64_t addr, unsigned size) { static_cast<VmBox*>(p)->vm.invalidatePteCache(addr, size); }
void ref_vm_sfence(void* p, int hasAddr, uint64_t addr, int hasAsid, uint64_t asid)
{
  auto& vm = static_cast<VmBox*>(p)->vm;
  auto& tlb = vm.tlb_;
  uint64_t vpn = vm.pageNumber(addr);
  if (!hasAddr && !hasAsid) tlb.invalidate();
  else if (!hasAddr) tlb.invalidateAsid(uint32_t(asid), 0);
  else if (!hasAsid) tlb.invalidateVirtualPage(vpn, 0);
  else tlb.invalidateVirtualPageAsid(vpn, uint32_t(asid), 0);
  vm.flushPteCache();
}
}
