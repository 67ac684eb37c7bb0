// TEST INFRASTRUCTURE ONLY.  extern "C" shim over the reference's VecRegs class (VecRegs.hpp / VecRegs.cpp, compiled from
// /root/reference where they lie) and the floating-point helpers of float-util.hpp, so that Python tests can pin the vector half
// of the oracle (vtype legality, element indexing, tail / mask agnostic fill, mask bits; negate / arithmetic / fused multiply-add
// / float <-> integer conversion helpers) against the reference's own code.  VecRegs keeps these members protected (Hart is a
// friend): VecBox derives from it and re-exports them.  Nothing here is product code.
#include <cstdint>
#include <cstring>
#include <bit>
#include <stdexcept>
#include "VecRegs.hpp"
#include "float-util.hpp"

using namespace WdRiscv;

struct VecBox : public VecRegs {
  using VecRegs::config; using VecRegs::updateConfig; using VecRegs::elemCount; using VecRegs::isDestActive;
  using VecRegs::isMaskDestActive; using VecRegs::writeMaskRegister; using VecRegs::readMaskRegister; using VecRegs::isActive;
  using VecRegs::getVecData; using VecRegs::read; using VecRegs::write;
};

template <class T> static int dest_active(VecBox* v, unsigned vd, unsigned ix, unsigned emulx8, int masked, uint64_t* val)
{
  T t{};
  try { bool a = v->isDestActive(vd, ix, emulx8, masked != 0, t); *val = (uint64_t)(std::make_unsigned_t<T>)t; return a ? 1 : 0; }
  catch (const std::exception&) { return -1; }
}

// ---- float-util.hpp helpers under SOFT_FLOAT.  rm: RISC-V rounding mode 0..4; *flags receives the RISC-V fflags bits raised.
static unsigned sf_to_riscv_flags(unsigned f)
{ return ((f & softfloat_flag_inexact) ? 1u : 0) | ((f & softfloat_flag_underflow) ? 2u : 0) | ((f & softfloat_flag_overflow) ? 4u : 0) |
         ((f & softfloat_flag_infinite) ? 8u : 0) | ((f & softfloat_flag_invalid) ? 16u : 0); }
template <class F, class U> static uint64_t fp_arith(int op, uint64_t a, uint64_t b, uint64_t c)
{
  F x = std::bit_cast<F>((U)a), y = std::bit_cast<F>((U)b), z = std::bit_cast<F>((U)c), r{};
  switch (op) { case 0: r = doFadd(x, y); break; case 1: r = doFsub(x, y); break; case 2: r = doFmul(x, y); break; case 3: r = doFdiv(x, y); break;
                case 4: r = doFsqrt(x); break; case 5: r = fusedMultiplyAdd(x, y, z); break; default: r = doNegate(x); break; }
  return (uint64_t)std::bit_cast<U>(r);
}

extern "C" {

void* ref_vec_new(unsigned bytesPerReg, unsigned minBytesPerElem, unsigned maxBytesPerElem)
{ auto* v = new VecBox(); v->config(bytesPerReg, minBytesPerElem, maxBytesPerElem, nullptr, nullptr); return v; }
void ref_vec_free(void* p) { delete static_cast<VecBox*>(p); }
// the cached vtype fields + vl, as execVsetvl* leave them (vector.cpp:596-634)
void ref_vec_update(void* p, unsigned sew, unsigned gm, int ma, int ta, int vill, unsigned vl)
{ auto* v = static_cast<VecBox*>(p); v->updateConfig(ElementWidth(sew), GroupMultiplier(gm), ma != 0, ta != 0, vill != 0); v->elemCount(vl); }
int ref_vec_legal(void* p, unsigned sew, unsigned gm) { return static_cast<VecBox*>(p)->legalConfig(ElementWidth(sew), GroupMultiplier(gm)) ? 1 : 0; }
int ref_vec_valid_index(void* p, unsigned reg, uint64_t ix, unsigned gx8, unsigned elemSize) { return static_cast<VecBox*>(p)->isValidIndex(reg, ix, gx8, elemSize) ? 1 : 0; }
unsigned ref_vec_vlmax(void* p) { return static_cast<VecBox*>(p)->vlmax(); }
unsigned ref_vec_vlmax_for(void* p, unsigned sew, unsigned gm) { return static_cast<VecBox*>(p)->vlmax(ElementWidth(sew), GroupMultiplier(gm)); }
unsigned ref_vec_elem_max(void* p, unsigned eew) { return static_cast<VecBox*>(p)->elemMax(ElementWidth(eew)); }
unsigned ref_vec_groupx8(unsigned gm) { return VecRegs::groupMultiplierX8(GroupMultiplier(gm)); }
void ref_vec_set_reg(void* p, unsigned reg, const uint8_t* bytes) { auto s = static_cast<VecBox*>(p)->getVecData(reg); memcpy(s.data(), bytes, s.size()); }
void ref_vec_get_reg(void* p, unsigned reg, uint8_t* bytes) { auto s = static_cast<VecBox*>(p)->getVecData(reg); memcpy(bytes, s.data(), s.size()); }
int ref_vec_dest_active(void* p, unsigned vd, unsigned ix, unsigned esz, unsigned emulx8, int masked, uint64_t* val)
{
  auto* v = static_cast<VecBox*>(p);
  switch (esz) { case 0: return dest_active<int8_t>(v, vd, ix, emulx8, masked, val); case 1: return dest_active<int16_t>(v, vd, ix, emulx8, masked, val);
                 case 2: return dest_active<int32_t>(v, vd, ix, emulx8, masked, val); default: return dest_active<int64_t>(v, vd, ix, emulx8, masked, val); }
}
int ref_vec_mask_dest_active(void* p, unsigned vd, unsigned ix, int masked, unsigned nelems, int* val)
{ bool b = false; try { bool a = static_cast<VecBox*>(p)->isMaskDestActive(vd, ix, masked != 0, nelems, b); *val = b; return a; } catch (const std::exception&) { return -1; } }
int ref_vec_write_mask(void* p, unsigned reg, unsigned i, int value)
{ try { static_cast<VecBox*>(p)->writeMaskRegister(reg, i, value != 0); return 0; } catch (const std::exception&) { return -1; } }
int ref_vec_write_elem(void* p, unsigned reg, uint64_t ix, unsigned esz, unsigned gx8, uint64_t val)
{
  auto* v = static_cast<VecBox*>(p);
  try {
    switch (esz) { case 0: v->write(reg, ix, gx8, (uint8_t)val); break; case 1: v->write(reg, ix, gx8, (uint16_t)val); break;
                   case 2: v->write(reg, ix, gx8, (uint32_t)val); break; default: v->write(reg, ix, gx8, (uint64_t)val); break; }
    return 0;
  } catch (const std::exception&) { return -1; }
}

// op: 0 add 1 sub 2 mul 3 div 4 sqrt 5 fused multiply-add 6 negate (doFadd .. doNegate, float-util.hpp:743-931)
uint64_t ref_fp_arith(int op, int dbl, uint64_t a, uint64_t b, uint64_t c, unsigned rm, unsigned* flags)
{
  softfloat_exceptionFlags = 0;
  int prev = setSimulatorRoundingMode(RoundingMode(rm));
  uint64_t r = dbl ? fp_arith<double, uint64_t>(op, a, b, c) : fp_arith<float, uint32_t>(op, a, b, c);
  undoSetSimulatorRoundingMode(prev);
  *flags = sf_to_riscv_flags(softfloat_exceptionFlags);
  return r;
}
// fpConvertTo (float-util.hpp:229-520).  code: 0 f32->i16 1 f32->u16 2 f64->i32 3 f64->u32 4 f32->i64 5 f32->u64
//   6 i16->f32 7 u16->f32 8 i32->f64 9 u32->f64 10 i64->f32 11 u64->f32 12 f32->i32 13 f32->u32 14 f64->i64 15 f64->u64
uint64_t ref_fp_convert(int code, uint64_t a, unsigned rm, unsigned* flags)
{
  softfloat_exceptionFlags = 0;
  int prev = setSimulatorRoundingMode(RoundingMode(rm));
  float f = std::bit_cast<float>((uint32_t)a); double d = std::bit_cast<double>(a);
  uint64_t r = 0;
  switch (code) {
    case 0: r = (uint16_t)fpConvertTo<int16_t>(f); break; case 1: r = fpConvertTo<uint16_t>(f); break;
    case 2: r = (uint32_t)fpConvertTo<int32_t>(d); break; case 3: r = fpConvertTo<uint32_t>(d); break;
    case 4: r = (uint64_t)fpConvertTo<int64_t>(f); break; case 5: r = fpConvertTo<uint64_t>(f); break;
    case 6: r = std::bit_cast<uint32_t>(fpConvertTo<float>((int16_t)a)); break; case 7: r = std::bit_cast<uint32_t>(fpConvertTo<float>((uint16_t)a)); break;
    case 8: r = std::bit_cast<uint64_t>(fpConvertTo<double>((int32_t)a)); break; case 9: r = std::bit_cast<uint64_t>(fpConvertTo<double>((uint32_t)a)); break;
    case 10: r = std::bit_cast<uint32_t>(fpConvertTo<float>((int64_t)a)); break; case 11: r = std::bit_cast<uint32_t>(fpConvertTo<float>((uint64_t)a)); break;
    case 12: r = (uint32_t)fpConvertTo<int32_t>(f); break; case 13: r = fpConvertTo<uint32_t>(f); break;
    case 14: r = (uint64_t)fpConvertTo<int64_t>(d); break; default: r = fpConvertTo<uint64_t>(d); break;
  }
  undoSetSimulatorRoundingMode(prev);
  *flags = sf_to_riscv_flags(softfloat_exceptionFlags);
  return r;
}

}
