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
  if (!init) { dis64.enableRv64(true); dis32.enableRv64(false); init = true; }
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
