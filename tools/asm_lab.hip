// Timing bench for the hand-written converged loop alone (whisper_amd/csrc/wsp_conv_core.h): builds in seconds, so the loop's
// structure can be tuned without the library's long device compile.  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value [-DLAB_CORE='"path/of/variant.h"'] tools/asm_lab.hip -o /tmp/asm_lab
//   /tmp/asm_lab [waves=1024] [harts_per_wave=4] [iterations=2000]
//
// One program per class: a loop of 64 instructions of the class + `addi x9, x9, -1; bnez x9, loop`; every wave runs it from LDS
// register files the kernel fills; prints ns per simulated step per wave and checks the retired count and one result register.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#define WSP_PROF_ADD(k, v) do { } while (0)
#define WSP_PROF_T() 0ULL
#include "../whisper_amd/csrc/wsp_core.h"
#ifndef LAB_CORE
#define LAB_CORE "../whisper_amd/csrc/wsp_conv_core.h"
#endif
#include LAB_CORE
using namespace wsp;

struct WspPureRet { u64 val; u32 fcsr, ok; };
extern "C" __device__ __noinline__ __attribute__((used)) WspPureRet wsp_pure_fp(u32 w0, i32 imm, u64 xa, u64 f1, u64 f2, u64 f3, u32 fcsr, u32 isa)
{ WspPureRet r; r.val = f1 + f2 + f3 + xa + w0 + (u32)imm + isa; r.fcsr = fcsr; r.ok = 1; return r; }
extern "C" __device__ __noinline__ __attribute__((used)) u64 wsp_pure_int(u32 w0, i32 imm, u64 a, u64 b, u32 isa)
{ return a * b + w0 + (u32)imm + isa; }

extern __shared__ u64 lab_lds[];

__global__ void __launch_bounds__(64, 1) lab_kernel(const HRec* hrecs, u64 mbase, u64 msize, u64 tlo, u32 tlen, u32 lpw_shift, u32 n, u32 isa,
                                                    u8* arena_all, u64 iters, u32* out_vit, u64* out_x, u32 cfg)
{
  const u32 lane = threadIdx.x, lpw = 1u << lpw_shift;
  const bool active = lane < lpw;
  const u32 own = active ? lane : 0;
  if (active) {
    u64* xs = lab_lds + own * 33; u64* fs = lab_lds + (lpw + own) * 33;
    for (u32 i = 0; i < 33; ++i) { xs[i] = 0; fs[i] = 0; }
    xs[8] = mbase + 0x4000; xs[9] = iters; xs[12] = 3; xs[13] = 5; xs[14] = 7;
  }
  __syncthreads();
  const u32 xb = own * 33 * 8, foff = lpw * 33 * 8;
  u8* arena = arena_all + (size_t)blockIdx.x * lpw * msize + (size_t)own * 8;
  u32 vpc = active ? 0u : 0xffffffffu, fcsr = 0, vit = 0, ncv = 0;
  u64 th = 0;
  conv_core(hrecs, hrecs, mbase, msize, tlo, tlo + tlen, tlen, cfg, lpw_shift, foff, n, isa, xb, arena, ~0ULL, vpc, fcsr, th, vit, ncv);
  __syncthreads();
  if (active) {
    out_vit[(blockIdx.x * lpw + lane) * 3 + 0] = vit; out_vit[(blockIdx.x * lpw + lane) * 3 + 1] = ncv; out_vit[(blockIdx.x * lpw + lane) * 3 + 2] = vpc;
    out_x[blockIdx.x * lpw + lane] = lab_lds[own * 33 + 16] + lab_lds[own * 33 + 9];
  }
}

// ---- a few encoders
static u32 R(u32 f7, u32 rs2, u32 rs1, u32 f3, u32 rd, u32 op) { return f7 << 25 | rs2 << 20 | rs1 << 15 | f3 << 12 | rd << 7 | op; }
static u32 I(i32 imm, u32 rs1, u32 f3, u32 rd, u32 op) { return ((u32)imm & 0xfff) << 20 | rs1 << 15 | f3 << 12 | rd << 7 | op; }
static u32 S(i32 imm, u32 rs2, u32 rs1, u32 f3) { return (((u32)imm >> 5) & 0x7f) << 25 | rs2 << 20 | rs1 << 15 | f3 << 12 | ((u32)imm & 0x1f) << 7 | 0x23; }
static u32 B(i32 off, u32 rs2, u32 rs1, u32 f3)
{ u32 o = (u32)off; return ((o >> 12) & 1) << 31 | ((o >> 5) & 0x3f) << 25 | rs2 << 20 | rs1 << 15 | f3 << 12 | ((o >> 1) & 0xf) << 8 | ((o >> 11) & 1) << 7 | 0x63; }
static u32 J(i32 off, u32 rd) { u32 o = (u32)off; return ((o >> 20) & 1) << 31 | ((o >> 1) & 0x3ff) << 21 | ((o >> 11) & 1) << 20 | ((o >> 12) & 0xff) << 12 | rd << 7 | 0x6f; }

static std::vector<u32> program(const std::string& kind)
{
  std::vector<u32> w;
  for (u32 k = 0; k < 64; ++k) {
    const u32 r = 16 + (k & 7);
    if (kind == "add") w.push_back(R(0, 13, 12, 0, r, 0x33));
    else if (kind == "addi") w.push_back(I(1, r, 0, r, 0x13));
    else if (kind == "xor") w.push_back(R(0, 13, 12, 4, r, 0x33));
    else if (kind == "addw") w.push_back(R(0, 13, 12, 0, r, 0x3b));
    else if (kind == "mul") w.push_back(R(1, 13, 12, 0, r, 0x33));
    else if (kind == "lw") w.push_back(I(4 * (k & 15), 8, 2, r, 0x03));
    else if (kind == "ld") w.push_back(I(8 * (k & 15), 8, 3, r, 0x03));
    else if (kind == "sw") w.push_back(S(4 * (k & 15), 12, 8, 2));
    else if (kind == "bne_nt") w.push_back(B(-4 * (i32)k, 12, 12, 1));
    else if (kind == "beq_t") w.push_back(B(4, 0, 0, 0));
    else if (kind == "jal") w.push_back(J(4, 0));
    else if (kind == "mulh") w.push_back(R(1, 13, 12, 1, r, 0x33));
    else if (kind == "fadd") w.push_back(R(1, 2, 1, 7, 3, 0x53));
    else if (kind == "mix") {
      const u32 m[8] = { I(0, 8, 2, 16, 0x03), I(64, 8, 2, 17, 0x03), R(1, 17, 16, 0, 16, 0x3b), R(0, 16, 18, 0, 18, 0x3b),
                         I(4, 19, 0, 19, 0x13), I(64, 20, 0, 20, 0x13), I(1, 21, 0, 21, 0x13), I(16, 0, 0, 22, 0x13) };
      w.push_back(m[k & 7]);
    } else { fprintf(stderr, "unknown class %s\n", kind.c_str()); exit(2); }
  }
  w.push_back(I(-1, 9, 0, 9, 0x13));
  w.push_back(B(-4 * 65, 0, 9, 1));
  w.push_back(0x00000073);                               // ecall: ends the run (outside the inline subset)
  return w;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
  const u32 waves = argc > 1 ? atoi(argv[1]) : 1024, lpw = argc > 2 ? atoi(argv[2]) : 4; const u64 iters = argc > 3 ? atoll(argv[3]) : 2000;
  const char* classes = argc > 4 ? argv[4] : "add,addi,xor,addw,mul,lw,ld,sw,bne_nt,beq_t,jal,mulh,fadd,mix";
  u32 lpw_shift = 0; while ((1u << lpw_shift) < lpw) ++lpw_shift;
  const u64 mbase = 0x80000000ULL, msize = 1 << 16; const u32 isa = ISA_M | ISA_A | ISA_F | ISA_D | ISA_C;
  u8* arena; CK(hipMalloc(&arena, (size_t)waves * lpw * msize)); CK(hipMemset(arena, 0, (size_t)waves * lpw * msize));
  u32* d_vit; u64* d_x; CK(hipMalloc(&d_vit, waves * lpw * 12)); CK(hipMalloc(&d_x, waves * lpw * 8));
  std::string cl = classes; size_t pos = 0;
  while (pos < cl.size()) {
    size_t e = cl.find(',', pos); if (e == std::string::npos) e = cl.size();
    const std::string kind = cl.substr(pos, e - pos); pos = e + 1;
    std::vector<u32> w = program(kind);
    const u32 tlen = (u32)w.size() * 4; const u32 nslots = tlen / 2;
    std::vector<HRec> h(nslots + 2);
    for (u32 i = 0; i < nslots; ++i) {
      Rec r; if (i & 1) { r = mk(OP_ILLEGAL, 0, 0, 0, 0); } else r = decode(w[i / 2], isa);
      h[i] = mk_hrec(r, isa, mbase, tlen, i * 2);
      if (i & 1) h[i].w[0] = 0;
    }
    for (u32 i = nslots; i < nslots + 2; ++i) { memset(&h[i], 0, sizeof(HRec)); h[i].w[1] = 2; h[i].w[4] = 64; h[i].w[5] = 64; }
    HRec* dh; CK(hipMalloc(&dh, h.size() * sizeof(HRec))); CK(hipMemcpy(dh, h.data(), h.size() * sizeof(HRec), hipMemcpyHostToDevice));
    const u64 steps = iters * 66;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(t0));
      hipLaunchKernelGGL(lab_kernel, dim3(waves), dim3(64), 2 * lpw * 33 * 8, 0, dh, mbase, msize, mbase, tlen, lpw_shift, (u32)(steps + 100), isa, arena, iters, d_vit, d_x, 4u | 8u | (getenv("LAB_DIV") ? 0x10000u : 0u) | (getenv("LAB_TR") ? 0x20000u : 0u));
      CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms;
    }
    std::vector<u32> vit(waves * lpw * 3); std::vector<u64> xs(waves * lpw);
    CK(hipMemcpy(vit.data(), d_vit, vit.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(xs.data(), d_x, xs.size() * 8, hipMemcpyDeviceToHost));
    bool ok = true; for (u32 i = 0; i < waves * lpw; ++i) ok = ok && vit[3 * i] == steps && vit[3 * i + 1] == 1 && vit[3 * i + 2] == tlen - 4 && xs[i] == xs[0];
    printf("{\"kind\": \"%s\", \"ok\": %s, \"steps\": %llu, \"ms\": %.3f, \"ns_per_step\": %.1f, \"x16\": %llu, \"vit\": %u, \"ncv\": %u, \"pc\": %u}\n", kind.c_str(), ok ? "true" : "false",
           (unsigned long long)steps, best, best * 1e6 / steps, (unsigned long long)xs[0], vit[0], vit[1], vit[2]);
    CK(hipFree(dh));
  }
  return 0;
}
