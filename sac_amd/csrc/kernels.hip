// gfx950 (MI355X) kernels for sac's predictor cascade + optimizer cost.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contraction OFF: the reference is
// plain SSE2 mul/add). FP64 everywhere, reference operation order.
//
// Decomposition: the cascade of src/profile.cpp:113-186 is feed-forward between
// stages (no stage reads a later stage's state), so every stage is its own time-serial kernel
// over a window, one workgroup per (frame, candidate) chain:
//     LPC (k_cov,k_chol,k_ssum,k_serr,k_pm) -> k_lms<1280,640> -> k_lms<256,128> -> k_lms<32,16> -> k_lms<8,4> -> k_mix -> k_cost
//
// Exact serial sums: the reference's dot products are strict left-to-right FP64 chains
// (src/pred/nlms.h:110-111). v_mfma_f64_4x4x4 with B == 1.0 computes, per output row,
//     D = (((C + a0) + a1) + a2) + a3      with IEEE rounding after every add
// (an FMA chain whose multiplier is exactly 1; verified on hardware, tools/ubench.hip, and by
// the parity tests), at 5.8 cycles/term against 14 for a dependent v_add_f64 -- so the long
// chains run on the matrix pipe as *adders*, not as a GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>
#include "jobs.h"

// ------------------------------------------------------------------------------------------
// x86 rsqrtss emulation (src/pred/nlms.h:34-57): f64 -> f32 (RNE), table on exponent parity x
// top `bits` mantissa bits, exponent rescale, f32 -> f64. Table entries are floats in [0.5, 1) with at
// most 15 significant mantissa bits (Intel and AMD: 12), kept as 16-bit words (jobs.h: RSQ_PACK) so that
// the table costs 16 KB of LDS per workgroup instead of 32.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rsq_emul(double a, const uint16_t *tab, int bits) {
  const float f = (float)a;
  const uint32_t u = __float_as_uint(f);
  const uint32_t e = (u >> 23) & 0xffu, mant = u & 0x7fffffu;
  const uint32_t par = e & 1u;
  const uint32_t idx = (par << bits) | (mant >> (23 - bits));
  uint32_t tb = RSQ_UNPACK(tab[idx]);
  const int q = ((int)e - (par ? 127 : 128)) >> 1;  // e - ec is even: exact
  tb -= (uint32_t)(q << 23);
  if (e == 255u) tb = mant ? (u | 0x400000u) : 0u;  // NaN stays NaN, +inf -> +0
  return (double)__uint_as_float(tb);
}

// rsq_emul with the integer part folded (argument is eg + 0.001 > 0 or NaN):
//   index  = bits [23-bits, 23] of the float (top mantissa bits + exponent parity)
//   rescale: table value * 2^-q, q = ((e+1)>>1) - 64
__device__ __forceinline__ double rsq_emul2(double a, const uint16_t *tab, int bits) {
  const uint32_t u = __float_as_uint((float)a);
  const uint32_t off = (u >> (22 - bits)) & (((2u << bits) - 1u) << 1);  // byte offset of the 16-bit table entry
  const uint32_t x = ((u + 0x800000u) >> 1) & 0x7f800000u;
  uint32_t tb = RSQ_UNPACK(*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tab) + off)) + (0x20000000u - x);
  // NaN stays NaN, +inf -> +0: rare, so one class test and a wave-uniform branch instead of seven selects per tap
  if (__builtin_amdgcn_ballot_w64(__builtin_amdgcn_classf(__uint_as_float(u), 0x207)) != 0ull) {
    asm volatile("" ::: "memory");  // keeps this a branch (the compiler turns it back into selects otherwise)
    if ((u & 0x7f800000u) == 0x7f800000u) tb = (u & 0x7fffffu) ? (u | 0x400000u) : 0u;
  }
  return (double)__uint_as_float(tb);
}


// rsq_emul2 for R arguments at once: the table reads are in flight together and one class test covers them all (a wave
// that runs alone on its SIMD -- the (256,128) stage -- is bound by the latency of each tap's chain, not by issue).
template <int R>
__device__ __forceinline__ void rsq_emul2_n(const double (&a)[R], double (&out)[R], const uint16_t *tab, int bits) {
  uint32_t u[R], tb[R];
  bool special = false;
#pragma unroll
  for (int r = 0; r < R; r++) {
    u[r] = __float_as_uint((float)a[r]);
    const uint32_t off = (u[r] >> (22 - bits)) & (((2u << bits) - 1u) << 1);
    tb[r] = RSQ_UNPACK(*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tab) + off));
    special |= (u[r] & 0x7f800000u) == 0x7f800000u;
  }
#pragma unroll
  for (int r = 0; r < R; r++) tb[r] += 0x20000000u - (((u[r] + 0x800000u) >> 1) & 0x7f800000u);
  if (__builtin_amdgcn_ballot_w64(special) != 0ull) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; r++)
      if ((u[r] & 0x7f800000u) == 0x7f800000u) tb[r] = (u[r] & 0x7fffffu) ? (u[r] | 0x400000u) : 0u;  // NaN stays NaN, +inf -> +0
  }
#pragma unroll
  for (int r = 0; r < R; r++) out[r] = (double)__uint_as_float(tb[r]);
}

// (int32_t)x as the reference's x86-64 build converts (cvttsd2si): anything outside int32, NaN included, becomes
// INT32_MIN (the GPU's own conversion saturates). Only garbage residuals on the decoder side get there.
__device__ __forceinline__ int32_t cvt_i32_x86(double x) {
  return (x < 2147483648.0 && x > -2147483649.0) ? (int32_t)x : (int32_t)0x80000000;
}

__device__ __forceinline__ double bcast(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}


// Exact serial sum of T4*4 LDS-resident products on the matrix pipe, hand-scheduled: the compiler
// (ROCm 7.2) sinks LDS reads next to their use and drains lgkmcnt(0) per group, which doubled the
// per-MFMA time (48 vs 23 cycles). Here the 8 operands of the next group are always in flight
// under the 8 dependent MFMAs of the current one. `addr` = LDS byte address of this lane's first
// operand; consecutive steps are 32 bytes apart. ITER = T4/16. Reads past the end return
// unused data (still inside the workgroup's LDS allocation: the ring follows the products).
#define MF(areg) "v_mfma_f64_4x4x4_4b_f64 %0, " areg ", %2, %0\n s_nop 3\n"
template <int ITER>
__device__ __forceinline__ double mfma_chain_lds(uint32_t addr, double acc = 0.) {  // acc: this D-layout lane's start value
  double one = 1.0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  uint32_t cnt;
  asm volatile(
      "ds_read_b64 %3, %1\n ds_read_b64 %4, %1 offset:32\n ds_read_b64 %5, %1 offset:64\n ds_read_b64 %6, %1 offset:96\n"
      "ds_read_b64 %7, %1 offset:128\n ds_read_b64 %8, %1 offset:160\n ds_read_b64 %9, %1 offset:192\n ds_read_b64 %10, %1 offset:224\n"
      "s_mov_b32 %19, %20\n"
      "1:\n"
      "ds_read_b64 %11, %1 offset:256\n ds_read_b64 %12, %1 offset:288\n ds_read_b64 %13, %1 offset:320\n ds_read_b64 %14, %1 offset:352\n"
      "ds_read_b64 %15, %1 offset:384\n ds_read_b64 %16, %1 offset:416\n ds_read_b64 %17, %1 offset:448\n ds_read_b64 %18, %1 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "v_add_u32 %1, 512, %1\n"
      "ds_read_b64 %3, %1\n ds_read_b64 %4, %1 offset:32\n ds_read_b64 %5, %1 offset:64\n ds_read_b64 %6, %1 offset:96\n"
      "ds_read_b64 %7, %1 offset:128\n ds_read_b64 %8, %1 offset:160\n ds_read_b64 %9, %1 offset:192\n ds_read_b64 %10, %1 offset:224\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "s_sub_u32 %19, %19, 1\n s_cmp_lg_u32 %19, 0\n s_cbranch_scc1 1b\n"
      "s_waitcnt lgkmcnt(0)\n s_nop 7\n"
      : "+v"(acc), "+v"(addr), "+v"(one), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6),
        "=&v"(a7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7), "=&s"(cnt)
      : "n"(ITER)
      : "memory", "scc");
  return acc;
}
#undef MF


// One MFMA adder pass over 16 segments x 128 products: chain j (segment j) starts from the value
// its D-layout lanes hold in `acc`, lane (k = lane>>4, j = lane&15) feeds term 4s+k of segment j
// at step s (32 steps, 32 bytes apart from `addr`). Returns D: lanes 16*i+4*blk+{0..3} hold chain
// 4*blk+i. Straight-line, operands of the next group of 8 always in flight.
#define MF(areg) "v_mfma_f64_4x4x4_4b_f64 %0, " areg ", %2, %0\n s_nop 3\n"
__device__ __forceinline__ double mfma_pass32(uint32_t addr, double acc) {
  double one = 1.0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  asm volatile(
      "ds_read_b64 %3, %1\n ds_read_b64 %4, %1 offset:32\n ds_read_b64 %5, %1 offset:64\n ds_read_b64 %6, %1 offset:96\n"
      "ds_read_b64 %7, %1 offset:128\n ds_read_b64 %8, %1 offset:160\n ds_read_b64 %9, %1 offset:192\n ds_read_b64 %10, %1 offset:224\n"
      "ds_read_b64 %11, %1 offset:256\n ds_read_b64 %12, %1 offset:288\n ds_read_b64 %13, %1 offset:320\n ds_read_b64 %14, %1 offset:352\n"
      "ds_read_b64 %15, %1 offset:384\n ds_read_b64 %16, %1 offset:416\n ds_read_b64 %17, %1 offset:448\n ds_read_b64 %18, %1 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:512\n ds_read_b64 %4, %1 offset:544\n ds_read_b64 %5, %1 offset:576\n ds_read_b64 %6, %1 offset:608\n"
      "ds_read_b64 %7, %1 offset:640\n ds_read_b64 %8, %1 offset:672\n ds_read_b64 %9, %1 offset:704\n ds_read_b64 %10, %1 offset:736\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "ds_read_b64 %11, %1 offset:768\n ds_read_b64 %12, %1 offset:800\n ds_read_b64 %13, %1 offset:832\n ds_read_b64 %14, %1 offset:864\n"
      "ds_read_b64 %15, %1 offset:896\n ds_read_b64 %16, %1 offset:928\n ds_read_b64 %17, %1 offset:960\n ds_read_b64 %18, %1 offset:992\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "s_nop 7\n"
      : "+v"(acc), "+v"(addr), "+v"(one), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6),
        "=&v"(a7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
      :
      : "memory");
  return acc;
}
#undef MF

#define MF(areg) "v_mfma_f64_4x4x4_4b_f64 %0, " areg ", %2, %0\n s_nop 3\n"
// 56 steps (224 terms), straight-line (generated: 8 steps per group, the next group's operands always in flight)
__device__ __forceinline__ double mfma_chain56(uint32_t addr, double acc = 0.) {
  double one = 1.0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  asm volatile(
      "ds_read_b64 %3, %1 offset:0\n ds_read_b64 %4, %1 offset:32\n ds_read_b64 %5, %1 offset:64\n ds_read_b64 %6, %1 offset:96\n ds_read_b64 %7, %1 offset:128\n ds_read_b64 %8, %1 offset:160\n ds_read_b64 %9, %1 offset:192\n ds_read_b64 %10, %1 offset:224\n"
      "ds_read_b64 %11, %1 offset:256\n ds_read_b64 %12, %1 offset:288\n ds_read_b64 %13, %1 offset:320\n ds_read_b64 %14, %1 offset:352\n ds_read_b64 %15, %1 offset:384\n ds_read_b64 %16, %1 offset:416\n ds_read_b64 %17, %1 offset:448\n ds_read_b64 %18, %1 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:512\n ds_read_b64 %4, %1 offset:544\n ds_read_b64 %5, %1 offset:576\n ds_read_b64 %6, %1 offset:608\n ds_read_b64 %7, %1 offset:640\n ds_read_b64 %8, %1 offset:672\n ds_read_b64 %9, %1 offset:704\n ds_read_b64 %10, %1 offset:736\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "ds_read_b64 %11, %1 offset:768\n ds_read_b64 %12, %1 offset:800\n ds_read_b64 %13, %1 offset:832\n ds_read_b64 %14, %1 offset:864\n ds_read_b64 %15, %1 offset:896\n ds_read_b64 %16, %1 offset:928\n ds_read_b64 %17, %1 offset:960\n ds_read_b64 %18, %1 offset:992\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:1024\n ds_read_b64 %4, %1 offset:1056\n ds_read_b64 %5, %1 offset:1088\n ds_read_b64 %6, %1 offset:1120\n ds_read_b64 %7, %1 offset:1152\n ds_read_b64 %8, %1 offset:1184\n ds_read_b64 %9, %1 offset:1216\n ds_read_b64 %10, %1 offset:1248\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "ds_read_b64 %11, %1 offset:1280\n ds_read_b64 %12, %1 offset:1312\n ds_read_b64 %13, %1 offset:1344\n ds_read_b64 %14, %1 offset:1376\n ds_read_b64 %15, %1 offset:1408\n ds_read_b64 %16, %1 offset:1440\n ds_read_b64 %17, %1 offset:1472\n ds_read_b64 %18, %1 offset:1504\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:1536\n ds_read_b64 %4, %1 offset:1568\n ds_read_b64 %5, %1 offset:1600\n ds_read_b64 %6, %1 offset:1632\n ds_read_b64 %7, %1 offset:1664\n ds_read_b64 %8, %1 offset:1696\n ds_read_b64 %9, %1 offset:1728\n ds_read_b64 %10, %1 offset:1760\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "s_nop 7\n"
      : "+v"(acc), "+v"(addr), "+v"(one), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6),
        "=&v"(a7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
      :
      : "memory");
  return acc;
}
// 40 steps (160 terms), straight-line (generated: 8 steps per group, the next group's operands always in flight)
__device__ __forceinline__ double mfma_chain40(uint32_t addr, double acc = 0.) {
  double one = 1.0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  asm volatile(
      "ds_read_b64 %3, %1 offset:0\n ds_read_b64 %4, %1 offset:32\n ds_read_b64 %5, %1 offset:64\n ds_read_b64 %6, %1 offset:96\n ds_read_b64 %7, %1 offset:128\n ds_read_b64 %8, %1 offset:160\n ds_read_b64 %9, %1 offset:192\n ds_read_b64 %10, %1 offset:224\n"
      "ds_read_b64 %11, %1 offset:256\n ds_read_b64 %12, %1 offset:288\n ds_read_b64 %13, %1 offset:320\n ds_read_b64 %14, %1 offset:352\n ds_read_b64 %15, %1 offset:384\n ds_read_b64 %16, %1 offset:416\n ds_read_b64 %17, %1 offset:448\n ds_read_b64 %18, %1 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:512\n ds_read_b64 %4, %1 offset:544\n ds_read_b64 %5, %1 offset:576\n ds_read_b64 %6, %1 offset:608\n ds_read_b64 %7, %1 offset:640\n ds_read_b64 %8, %1 offset:672\n ds_read_b64 %9, %1 offset:704\n ds_read_b64 %10, %1 offset:736\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "ds_read_b64 %11, %1 offset:768\n ds_read_b64 %12, %1 offset:800\n ds_read_b64 %13, %1 offset:832\n ds_read_b64 %14, %1 offset:864\n ds_read_b64 %15, %1 offset:896\n ds_read_b64 %16, %1 offset:928\n ds_read_b64 %17, %1 offset:960\n ds_read_b64 %18, %1 offset:992\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "ds_read_b64 %3, %1 offset:1024\n ds_read_b64 %4, %1 offset:1056\n ds_read_b64 %5, %1 offset:1088\n ds_read_b64 %6, %1 offset:1120\n ds_read_b64 %7, %1 offset:1152\n ds_read_b64 %8, %1 offset:1184\n ds_read_b64 %9, %1 offset:1216\n ds_read_b64 %10, %1 offset:1248\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF("%11") MF("%12") MF("%13") MF("%14") MF("%15") MF("%16") MF("%17") MF("%18")
      "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
      MF("%3") MF("%4") MF("%5") MF("%6") MF("%7") MF("%8") MF("%9") MF("%10")
      "s_nop 7\n"
      : "+v"(acc), "+v"(addr), "+v"(one), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6),
        "=&v"(a7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
      :
      : "memory");
  return acc;
}
#undef MF

// mfma_pass32 that KEEPS the 32 operands of every lane (64 VGPRs), and the same pass from kept operands: no LDS access at all.
// For a wave that sums the same products twice (k_lms1c's walker: candidate pass, then verification while the other waves
// are busy in the LDS).
struct Op32 { double v[32]; };
__device__ __forceinline__ double mfma_pass32_keep(uint32_t addr, double acc, Op32 &p) {
  double one = 1.0;
  asm volatile(
      "ds_read_b64 %0, %33 offset:0\n" "ds_read_b64 %1, %33 offset:32\n" "ds_read_b64 %2, %33 offset:64\n" "ds_read_b64 %3, %33 offset:96\n" "ds_read_b64 %4, %33 offset:128\n" "ds_read_b64 %5, %33 offset:160\n" "ds_read_b64 %6, %33 offset:192\n" "ds_read_b64 %7, %33 offset:224\n"
      "ds_read_b64 %8, %33 offset:256\n" "ds_read_b64 %9, %33 offset:288\n" "ds_read_b64 %10, %33 offset:320\n" "ds_read_b64 %11, %33 offset:352\n" "ds_read_b64 %12, %33 offset:384\n" "ds_read_b64 %13, %33 offset:416\n" "ds_read_b64 %14, %33 offset:448\n" "ds_read_b64 %15, %33 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      "v_mfma_f64_4x4x4_4b_f64 %32, %0, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %1, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %2, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %3, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %4, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %5, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %6, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %7, %34, %32\n s_nop 3\n"
      "ds_read_b64 %16, %33 offset:512\n" "ds_read_b64 %17, %33 offset:544\n" "ds_read_b64 %18, %33 offset:576\n" "ds_read_b64 %19, %33 offset:608\n" "ds_read_b64 %20, %33 offset:640\n" "ds_read_b64 %21, %33 offset:672\n" "ds_read_b64 %22, %33 offset:704\n" "ds_read_b64 %23, %33 offset:736\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      "v_mfma_f64_4x4x4_4b_f64 %32, %8, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %9, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %10, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %11, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %12, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %13, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %14, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %15, %34, %32\n s_nop 3\n"
      "ds_read_b64 %24, %33 offset:768\n" "ds_read_b64 %25, %33 offset:800\n" "ds_read_b64 %26, %33 offset:832\n" "ds_read_b64 %27, %33 offset:864\n" "ds_read_b64 %28, %33 offset:896\n" "ds_read_b64 %29, %33 offset:928\n" "ds_read_b64 %30, %33 offset:960\n" "ds_read_b64 %31, %33 offset:992\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      "v_mfma_f64_4x4x4_4b_f64 %32, %16, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %17, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %18, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %19, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %20, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %21, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %22, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %23, %34, %32\n s_nop 3\n"
      "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
      "v_mfma_f64_4x4x4_4b_f64 %32, %24, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %25, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %26, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %27, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %28, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %29, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %30, %34, %32\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %32, %31, %34, %32\n s_nop 3\n"
      "s_nop 7\n"
      : "=&v"(p.v[0]), "=&v"(p.v[1]), "=&v"(p.v[2]), "=&v"(p.v[3]), "=&v"(p.v[4]), "=&v"(p.v[5]), "=&v"(p.v[6]), "=&v"(p.v[7]), "=&v"(p.v[8]), "=&v"(p.v[9]), "=&v"(p.v[10]), "=&v"(p.v[11]), "=&v"(p.v[12]), "=&v"(p.v[13]), "=&v"(p.v[14]), "=&v"(p.v[15]), "=&v"(p.v[16]), "=&v"(p.v[17]), "=&v"(p.v[18]), "=&v"(p.v[19]), "=&v"(p.v[20]), "=&v"(p.v[21]), "=&v"(p.v[22]), "=&v"(p.v[23]), "=&v"(p.v[24]), "=&v"(p.v[25]), "=&v"(p.v[26]), "=&v"(p.v[27]), "=&v"(p.v[28]), "=&v"(p.v[29]), "=&v"(p.v[30]), "=&v"(p.v[31]), "+v"(acc)
      : "v"(addr), "v"(one)
      : "memory");
  return acc;
}
__device__ __forceinline__ double mfma_pass32_regs(double acc, const Op32 &p) {
  double one = 1.0;
  asm volatile(
      "s_nop 7\n"  // the start value may come straight from a vector instruction
      "v_mfma_f64_4x4x4_4b_f64 %0, %1, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %2, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %3, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %4, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %5, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %6, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %7, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %8, %33, %0\n s_nop 3\n"
      "v_mfma_f64_4x4x4_4b_f64 %0, %9, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %10, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %11, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %12, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %13, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %14, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %15, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %16, %33, %0\n s_nop 3\n"
      "v_mfma_f64_4x4x4_4b_f64 %0, %17, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %18, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %19, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %20, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %21, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %22, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %23, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %24, %33, %0\n s_nop 3\n"
      "v_mfma_f64_4x4x4_4b_f64 %0, %25, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %26, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %27, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %28, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %29, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %30, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %31, %33, %0\n s_nop 3\n" "v_mfma_f64_4x4x4_4b_f64 %0, %32, %33, %0\n s_nop 3\n"
      "s_nop 7\n"
      : "+v"(acc)
      : "v"(p.v[0]), "v"(p.v[1]), "v"(p.v[2]), "v"(p.v[3]), "v"(p.v[4]), "v"(p.v[5]), "v"(p.v[6]), "v"(p.v[7]), "v"(p.v[8]), "v"(p.v[9]), "v"(p.v[10]), "v"(p.v[11]), "v"(p.v[12]), "v"(p.v[13]), "v"(p.v[14]), "v"(p.v[15]), "v"(p.v[16]), "v"(p.v[17]), "v"(p.v[18]), "v"(p.v[19]), "v"(p.v[20]), "v"(p.v[21]), "v"(p.v[22]), "v"(p.v[23]), "v"(p.v[24]), "v"(p.v[25]), "v"(p.v[26]), "v"(p.v[27]), "v"(p.v[28]), "v"(p.v[29]), "v"(p.v[30]), "v"(p.v[31]), "v"(one));
  return acc;
}

// mfma_pass32 with TWO accumulators per lane on the same operands: a candidate wave then carries 8 candidates per segment, two
// waves the 16 of a chain -- half the LDS operand traffic of four waves, and the two MFMAs of a step do not wait for each other
// (the two-channel kernel, where eight candidate waves shared four matrix pipes and the LDS).
#define MF2(areg) "v_mfma_f64_4x4x4_4b_f64 %0, " areg ", %3, %0\n s_nop 1\n v_mfma_f64_4x4x4_4b_f64 %1, " areg ", %3, %1\n s_nop 1\n"
__device__ __forceinline__ void mfma_pass32x2(uint32_t addr, double &acc0, double &acc1) {
  double one = 1.0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  asm volatile(
      "ds_read_b64 %4, %2 offset:0\n ds_read_b64 %5, %2 offset:32\n ds_read_b64 %6, %2 offset:64\n ds_read_b64 %7, %2 offset:96\n ds_read_b64 %8, %2 offset:128\n ds_read_b64 %9, %2 offset:160\n ds_read_b64 %10, %2 offset:192\n ds_read_b64 %11, %2 offset:224\n"
      "ds_read_b64 %12, %2 offset:256\n ds_read_b64 %13, %2 offset:288\n ds_read_b64 %14, %2 offset:320\n ds_read_b64 %15, %2 offset:352\n ds_read_b64 %16, %2 offset:384\n ds_read_b64 %17, %2 offset:416\n ds_read_b64 %18, %2 offset:448\n ds_read_b64 %19, %2 offset:480\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF2("%4") MF2("%5") MF2("%6") MF2("%7") MF2("%8") MF2("%9") MF2("%10") MF2("%11")
      "ds_read_b64 %4, %2 offset:512\n ds_read_b64 %5, %2 offset:544\n ds_read_b64 %6, %2 offset:576\n ds_read_b64 %7, %2 offset:608\n ds_read_b64 %8, %2 offset:640\n ds_read_b64 %9, %2 offset:672\n ds_read_b64 %10, %2 offset:704\n ds_read_b64 %11, %2 offset:736\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF2("%12") MF2("%13") MF2("%14") MF2("%15") MF2("%16") MF2("%17") MF2("%18") MF2("%19")
      "ds_read_b64 %12, %2 offset:768\n ds_read_b64 %13, %2 offset:800\n ds_read_b64 %14, %2 offset:832\n ds_read_b64 %15, %2 offset:864\n ds_read_b64 %16, %2 offset:896\n ds_read_b64 %17, %2 offset:928\n ds_read_b64 %18, %2 offset:960\n ds_read_b64 %19, %2 offset:992\n"
      "s_waitcnt lgkmcnt(8)\n s_nop 1\n"
      MF2("%4") MF2("%5") MF2("%6") MF2("%7") MF2("%8") MF2("%9") MF2("%10") MF2("%11")
      "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
      MF2("%12") MF2("%13") MF2("%14") MF2("%15") MF2("%16") MF2("%17") MF2("%18") MF2("%19")
      "s_nop 7\n"
      : "+v"(acc0), "+v"(acc1), "+v"(addr), "+v"(one), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6),
        "=&v"(a7), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
      :
      : "memory");
}
#undef MF2

#ifndef DPP_BC
#define DPP_BC true
#endif
// Lane i of every 16-lane row receives the value of lane i-O of the same row (0. for i < O): DPP row_shr.
template <int O>
__device__ __forceinline__ double row_shr(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + O, 0xf, 0xf, DPP_BC);  // (bound_ctrl: the zero comes from the DPP unit, no register to clear first)
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + O, 0xf, 0xf, DPP_BC);
  return __hiloint2double(hi, lo);
}
template <int O>
__device__ __forceinline__ int row_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + O, 0xf, 0xf, DPP_BC); }
// Inclusive prefix sum over each 16-lane row (tree order: only ever used as a guess)
__device__ __forceinline__ double row_prefix16(double v) {
  v += row_shr<1>(v);
  v += row_shr<2>(v);
  v += row_shr<4>(v);
  v += row_shr<8>(v);
  return v;
}

// ulp of a normal double as a power of two (0 for zero / tiny values: no candidates then)
__device__ __forceinline__ double ulp_pow2(double x, double *inv) {
  const int e = (__double2hiint(x) >> 20) & 0x7ff;
  if (e < 64 || e > 1900) { *inv = 0.; return 0.; }
  *inv = __hiloint2double((1023 + 1075 - e) << 20, 0);  // 2^(52 - (e-1023))
  return __hiloint2double((e - 52) << 20, 0);           // 2^(e-1023-52)
}

// ------------------------------------------------------------------------------------------
// Speculative exact summation of one chain of NSEG segments x 4*STEPS products in LDS (segment stride SEGP
// doubles starting at LDS byte address prods_addr; chains NSEG..15 of the MFMA passes idle). Called by every
// wave of the workgroup (it contains a workgroup barrier); `group` in 0..3 marks the four waves that sum
// candidates for this chain, -1 the bystanders; `walker` is the one candidate wave (group 0) that follows the
// chain and verifies it. sA[j]: approximate sum of segment j (any order -- the caller's tree reduction).
// Returns, in every lane of the walker, the strict left-to-right FP64 sum (nlms.h:110-111).
//   P1    the prefix of sA guesses the segment inputs g_j (boundary j = input of segment j). Each candidate
//         wave sums every segment EXACTLY from 16 candidate inputs g_j + (m-8) ulp: the four columns of the
//         4x4 MFMA output share their A terms (B == 1) but keep their own C, so wave w, column c carries
//         candidate 4w + c. Each output is stored as its offset on the next boundary's candidate grid:
//         sJ[j][m] = offset at boundary j+1 given offset m at boundary j.
//   walk  d_{l+1} = step_l(d_l) from the exact x_1, step_l(d) = sJ[l][d & 15] + (d - (d & 15)) converted to the
//         next boundary's binade (to nearest): offsets outside 0..15 are extended periodically (a shift by 16q ulps) -- a
//         guess, but a good one: shifts by high powers of two survive the coarser binades a segment's partial
//         sums pass through (4 candidates / shifts by 4q failed on 30-45 % of the samples, 16 fail on 5-9 %).
//         The walker holds the whole table in four registers and walks on the scalar unit: the offset is an SGPR,
//         a look-up a v_readlane with a computed lane, the conversion three per-boundary numbers (shift left, rounding
//         term, shift right) computed for all boundaries at once before the pass. (Round 1 solved the recursion for all
//         boundaries at once by a fixed-point iteration on DPP prefix sums with LDS look-ups, 6.5 iterations of ~35
//         vector instructions: 1.2k cycles more per sample.)
//   P2    every segment re-summed exactly from its walked input; each output must EQUAL the next walked
//         input bit for bit. By induction from x_0 = +0 the last output then IS the sequential sum. A
//         mismatch pins the first wrong boundary to the exact value just computed and re-walks the later
//         ones (each round fixes at least one boundary).
// Everything before P2 only affects speed; exactness rests on P2 alone.
// ------------------------------------------------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
// Layout: MFMA chain 4*blk + i carries segment 4*i + blk, so that the D-layout lane 16*i + 4*blk + col -- the lane that holds the
// accumulator of that chain's candidate column col -- is lane 4*segment + col: the segments lie along the wave in order. The
// guess (a prefix sum over the segments) is then a DPP scan in the very lanes that need it as MFMA start values, the "next
// boundary" of a lane is the scan's inclusive value in the same lane, and the walker addresses boundary l as lane 4*l: no
// cross-lane shuffle (ds_bpermute) anywhere.
//
// The three phases are separate members so that a kernel can put its own work between them (k_lms1c updates with the walk's
// prediction while the walker verifies it); spec_exact_sum() below is the plain sequence: candidates | barrier | first_walk, verify.
template <int NSEG, int SEGP>
struct SpecSum {
  static constexpr int NB = NSEG - 1;  // last boundary = input of the last segment
  uint32_t paddr;                      // LDS byte address of this A-operand lane's first term
  int sD, col;                         // D-layout lane: segment sD, candidate column col
  // boundary sD (input of this lane's segment): candidate grid gBase + m * gU; boundary sD + 1 (its output): nB + m * nU
  double gBase, gU, gIU, nB, nU, nIU;
  int ksp, ksn, krn;                   // the walk's off-grid conversion (lane 4*l: step from boundary l)
  int tb0, tb1, tb2, tb3;              // walker: the 15 x 16 table, boundary l's row in lanes 16 * (l & 3) + m of tb[l >> 2]
  double xs, xn;                       // walker, lane 4*s: x_s, the input of segment s, and x_{s+1}, the output it must produce
  int dS, dE;                          // walker, lane 4*s: their offsets on the candidate grids
  int sp, sd;                          // boundaries <= sp are exact; sd = offset of x_sp on its grid
  int dw[NSEG + 1];                    // walker: the walked offsets d_2 .. d_NB, scalars until verify() needs them in lanes
  int walk_its;

  // P1, the four candidate waves (group 0..3) of the chain. sOut (optional): the last segment's outputs per candidate, as doubles --
  // what the sum will be if the walk ends on candidate m.
  // WIDE: the idle sixteenth chain sums the LAST segment a second time, from 16 more candidates on either side of the grid
  // (grid indices -8..-1 and 16..23): sOut then has 32 entries, index = grid index + 8.
  // DUAL: two candidate waves (group 0, 1) with two accumulators each (mfma_pass32x2): candidates 8*group + 4*a + column.
  template <bool KEEP = false, bool WIDE = false, bool DUAL = false>
  __device__ __forceinline__ void candidates(uint32_t prods_addr, int (*sJ)[16], double *sX1, const double *sA, double *sOut, int group,
                                             bool walker, int lane, Op32 *ops = nullptr) {
    const int jA = lane & 15, kA = lane >> 4;          // A-operand lane: chain jA, term 4s+kA of its segment at step s
    const int segA = 4 * (jA & 3) + (jA >> 2);         // ... and the segment that chain carries
    sD = lane >> 2; col = lane & 3;
    paddr = prods_addr + (uint32_t)((WIDE && segA == NSEG ? NB : segA) * SEGP + kA) * 8u;
    walk_its = 0;
    const double A = sD < NSEG ? sA[sD] : 0.;
    // inclusive prefix over the segments, any order (only a guess): quads of a row, then rows
    double incl = A;
    incl += row_shr<4>(incl);
    incl += row_shr<8>(incl);
    {
      int lo = __double2loint(incl), hi = __double2hiint(incl);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
      hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
      incl += __hiloint2double(hi, lo);
      lo = __double2loint(incl); hi = __double2hiint(incl);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
      hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
      incl += __hiloint2double(hi, lo);
    }
    // the previous segment's inclusive value, MOVED (not recomputed: both neighbours must use the same bits for a boundary):
    // quads 1..3 from the quad before (row_shr:4), quad 0 of rows 1..3 from the last lane of the row before (row_bcast:15)
    double prev;
    {
      int lo = __double2loint(incl), hi = __double2hiint(incl);
      int plo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xE, false);
      int phi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xE, false);
      plo = __builtin_amdgcn_update_dpp(plo, lo, 0x142, 0xE, 0x1, false);
      phi = __builtin_amdgcn_update_dpp(phi, hi, 0x142, 0xE, 0x1, false);
      prev = __hiloint2double(phi, plo);
    }
    gU = ulp_pow2(prev, &gIU);
    gBase = prev - 8. * gU;
    nU = ulp_pow2(incl, &nIU);
    nB = incl - 8. * nU;
    const int m16 = (DUAL ? 8 : 4) * group + col;
    int gi = m16;  // grid index of this lane's (first) candidate
    if constexpr (WIDE) {
      const double gBq = row_shr<4>(gBase), gUq = row_shr<4>(gU);  // the quad before (all lanes shift: a DPP source lane must be active)
      if (sD == NSEG) { gBase = gBq; gU = gUq; gi = m16 < 8 ? m16 - 8 : m16 + 8; }  // the last boundary's grid
    }
    const double stv = gBase + (double)gi * gU;
    if (walker) {
      // the walk's conversion of an off-grid offset from boundary l's ulps to boundary l + 1's, as three numbers per boundary:
      // ((hi << ksp) + krn) >> ksn, to nearest (a guess only)
      const int ex = (__double2hiint(gU) >> 20) & 0x7ff, exn = (__double2hiint(nU) >> 20) & 0x7ff;  // exponent fields of the ulps (0: no grid)
      int ks = (ex && exn) ? ex - exn : 0;
      ks = ks > 31 ? 31 : (ks < -31 ? -31 : ks);
      ksp = max(ks, 0); ksn = max(-ks, 0); krn = ksn ? 1 << (ksn - 1) : 0;
    }
    double o;
    if constexpr (DUAL) {
      static_assert(!KEEP && !WIDE, "DUAL is the plain candidate pass of the two-channel kernel");
      double o2 = gBase + (double)(m16 + 4) * gU;
      o = stv;
      mfma_pass32x2(paddr, o, o2);
      if (sD < NSEG) sJ[sD][m16 + 4] = (sD < NB) ? (int)((o2 - nB) * nIU) : 0;
    } else if constexpr (KEEP) o = mfma_pass32_keep(paddr, stv, *ops);
    else o = mfma_pass32(paddr, stv);
    if (sD < NSEG) sJ[sD][m16] = (sD < NB) ? (int)((o - nB) * nIU) : 0;
    if (sD == 0 && m16 == 0) *sX1 = o;  // segment 0 summed from +0: x_1, exact
    if (sOut && (sD == NB || (WIDE && sD == NSEG))) sOut[WIDE ? gi + 8 : m16] = o;
  }

  // walk d_{l+1} = step_l(d_l) for l >= sp, one boundary after the other on the scalar unit: the offset lives in an SGPR, a table
  // look-up is a v_readlane with a computed lane (no LDS round trip, no cross-lane work); the offsets stay scalars (dw) until
  // verify() moves them to their lanes. Branches are what costs here (~25 cycles
  // each, taken or not), so the first walk (FIRST: sp == 1) has no start tests. Returns the offset at the last boundary.
  template <bool FIRST>
  __device__ __forceinline__ int walk() {
    int ds = sd;
    auto step = [&](auto lc) {
      constexpr int l = decltype(lc)::value;
      int t = __builtin_amdgcn_readlane(l < 4 ? tb0 : l < 8 ? tb1 : l < 12 ? tb2 : tb3, 16 * (l & 3) + (ds & 15));
      if ((unsigned)ds > 15u)  // off the grid (one compare on the way that stays on it; an index without the mask -- an add --
                               // measured slower: a lane select fresh from a v_readlane costs wait states)
        t += (((ds & ~15) << __builtin_amdgcn_readlane(ksp, 4 * l)) + __builtin_amdgcn_readlane(krn, 4 * l)) >> __builtin_amdgcn_readlane(ksn, 4 * l);
      ds = __builtin_amdgcn_readfirstlane(t);
      dw[l + 1] = ds;
      walk_its++;
    };
    if (FIRST) static_for<1, NB>(step);
    else static_for<1, NB>([&](auto lc) { if (decltype(lc)::value >= sp) step(lc); });
    return ds;
  }

  // the walker, after the barrier behind candidates(): table into registers, first walk
  __device__ __forceinline__ int first_walk(int (*sJ)[16], const double *sX1, int lane) {
    const int *sJf = &sJ[0][0];
    tb0 = sJf[lane]; tb1 = sJf[64 + lane]; tb2 = sJf[128 + lane]; tb3 = sJf[192 + lane];
    const double x1 = *sX1;
    xs = sD == 1 ? x1 : 0.; xn = sD == 0 ? x1 : 0.;  // x_0 = +0 and x_1 are exact
    dS = 8; dE = 8;
#pragma unroll
    for (int l = 0; l <= NSEG; l++) dw[l] = 8;
    sp = 1;
    sd = __builtin_amdgcn_readlane(tb0, 0);
    return walk<true>();
  }

  // P2 until it holds: the exact sum, in every lane of the walker
  template <bool KEEP = false>
  __device__ __forceinline__ double verify(int dbg, long long &rounds, int lane, int *hist = nullptr, const Op32 *ops = nullptr) {
    double res = 0.;
    for (int guard = 0; guard < 17; guard++) {
      if (guard > 0) walk<false>();
      // the walked offsets (scalars) to their lanes: 4*(l+1) as that segment's input, 4*l as this segment's expected output
      static_for<1, NB>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dS) : "s"(dw[l + 1]), "i"(4 * (l + 1)));
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dE) : "s"(dw[l + 1]), "i"(4 * l));
      });
      if (sD > sp) xs = gBase + (double)dS * gU;
      if (sD >= sp) xn = nB + (double)dE * nU;
      // every boundary at once (column 0 of every segment; the other columns sum from whatever they hold)
      if constexpr (KEEP) res = mfma_pass32_regs(xs, *ops); else res = mfma_pass32(paddr, xs);
      const bool bad = (col == 0) && (sD < NB) && (__double_as_longlong(res) != __double_as_longlong(xn));
      const unsigned long long mask = __ballot(bad);
      rounds++;
#ifdef SACOPT_ABLATE
      if (hist && mask == 0ull) {  // statistics (tools/ongrid.py): how far from the guess the exact boundary values lie
        const bool inb = col == 0 && sD >= 1 && sD <= NB;
        const double q = (xs - gBase) * gIU;
        const bool ongrid = !inb || (gU != 0. && q == (double)(long long)q && gBase + q * gU == xs);
        int dev = (inb && ongrid) ? abs((int)q - 8) : 0;
        for (int o = 1; o < 64; o <<= 1) dev = max(dev, __shfl_xor(dev, o));
        const bool allgrid = __ballot(!ongrid) == 0ull;
        if (lane == 0) {
          const int bin = !allgrid ? 7 : dev <= 7 ? 0 : dev <= 15 ? 1 : dev <= 23 ? 2 : dev <= 31 ? 3 : dev <= 63 ? 4 : dev <= 127 ? 5 : 6;
          hist[bin]++;
          if (guard > 0) hist[8]++;
        }
      }
#endif
      if (mask == 0ull || (dbg & 16)) break;
      const int f = __builtin_amdgcn_readfirstlane((int)(__builtin_ctzll(mask) >> 2));  // first failing segment
      const double xt = bcast(res, 4 * f);  // exact x_{f+1}
      if (sD == f + 1) xs = xt;
      if (sD == f) xn = xt;
      sp = f + 1;
      const double bq = bcast(nB, 4 * f), iq = bcast(nIU, 4 * f);
      sd = __builtin_amdgcn_readfirstlane((int)((xt - bq) * iq));
    }
    if (dbg & 1024) rounds += (long long)walk_its << 32;
    return bcast(res, 4 * NB);  // the last segment's output: the sum
  }
};

template <int NSEG, int STEPS, int SEGP, bool DUAL = false>
__device__ __forceinline__ double spec_exact_sum(uint32_t prods_addr, int (*sJ)[16], double *sX1, const double *sA, int group,
                                                 bool walker, int dbg, long long &rounds, const int lane, int *hist = nullptr) {
  static_assert(STEPS == 32, "one pass = 32 MFMA steps");
  SpecSum<NSEG, SEGP> ss;
  if (group >= 0) ss.template candidates<false, false, DUAL>(prods_addr, sJ, sX1, sA, nullptr, group, walker, lane);
  __syncthreads();
  double res = 0.;
  if (walker) {
    ss.first_walk(sJ, sX1, lane);
    res = ss.verify(dbg, rounds, lane, hist);
  }
  return res;
}
#define SPEC_RESULT_LANE(NSEG) 0  // the sum is returned to every lane of the walker
// ------------------------------------------------------------------------------------------
// LMSADA2 stereo pair (src/pred/nlms.h:96-139), driven as StereoHigh/StereoNormal drive it
// (src/profile.cpp:126-148,176-186):
//   pred0(t) = lms[0].Predict(x1(t-1));  pred1(t) = lms[1].Predict(x0(t));  lms[c].Update(xc(t))
// One workgroup per chain. Taps are spread over the threads (w, eg in registers), the history
// is a ring of stage inputs in LDS (shifts of nlms.h:109,130 become index arithmetic), products
// go to LDS in the operand layout of the f64 MFMA, wave 0 sums them on the matrix pipe.
// ------------------------------------------------------------------------------------------
#define L1_SEG 128   // products per segment (32 MFMA steps)
#define L1_SEGP 130  // segment stride in LDS: the 16 chain lanes of a ds_read_b64 half on 16 different 4-bank groups
// The samples a launch covers: [job.t0, job.t0 + job.n) as the job says (clen == 0), or -- chunk launches of one shared job table,
// whose jobs describe the WHOLE window [0, job.n) -- the chunk [ct0, ct0 + clen) clipped to the window; a chain whose window
// ends before the chunk has nothing to do (its workgroup leaves at once: nothing loaded, nothing stored).
#define LMS_SPAN(job, ct0, clen)                                                                        \
  const int t0 = (clen) > 0 ? (ct0) : (job).t0;                                                         \
  const int tend = (clen) > 0 ? ((job).n < (ct0) + (clen) ? (job).n : (ct0) + (clen)) : (job).t0 + (job).n; \
  if ((clen) > 0 && (ct0) > 0 && tend <= t0) return;
#ifndef S2_CHP
#define S2_CHP LMS_CHUNK  // two-channel (1280,640) kernel: samples staged per refill
#endif
#ifndef S2_DUAL
#define S2_DUAL 1  // two-channel (1280,640) kernel: two candidate waves per channel with two accumulators each
#endif
#ifndef UNCOND_UPDATE
#define UNCOND_UPDATE 1
#endif
#ifndef H_BATCH_RSQ
#define H_BATCH_RSQ 1
#endif
#ifndef H_SPLIT
#define H_SPLIT 224  // (256,128) stage: terms summed from +0 by wave 0 (56 MFMA steps); the candidate waves guess that sum first, so they get the shorter rest (40 steps)
#endif
static_assert(H_SPLIT == 192 || H_SPLIT == 224, "chains exist for 192 + 192 and 224 + 160 terms");
#ifndef H_HALVES
#define H_HALVES 1  // (256,128): the two-halves exact-hit sum (0: the undivided 96-step chain; A/B switch)
#endif
#ifndef H_PLAIN_MIN
#define H_PLAIN_MIN 512  // (30 / 40 frames = 330 / 440 chains are still 1-2 % faster with the halves)
#endif
#ifndef LMS_HALVES_WAVES
#define LMS_HALVES_WAVES 2  // candidate waves of the (256,128) stage's second half: 32 candidates per channel each
#endif
#ifndef LMS_TABG
#define LMS_TABG 1  // 75-frame encode call, same run: 1276 -> 1251 ms; 2 (also the (256,128) stage: 768 look-ups per sample through the L1) 1284 ms
#endif
template <int S, int D, int NT, int R, int RL, bool SPEC, int CHP = LMS_CHUNK, bool HALV = true>
__global__ __launch_bounds__(NT) void k_lms(const LmsJob *__restrict__ jobs, const uint16_t *__restrict__ gtab,
                                            int bits, int ct0, int clen) {
  constexpr int T = S + D, T4 = T / 4, M = RL - 1, CH = CHP;  // CH: samples staged per refill (divides LMS_CHUNK, the unit of the host's chunk edges)
  static_assert(LMS_CHUNK % CHP == 0, "refill size must divide LMS_CHUNK");
  // SPEC: 15 segments of 128 products, padded to a stride of 130 doubles so that the 16 segment
  // lanes of a ds_read_b64 half fall on 16 different 4-bank groups
  constexpr int SEG = L1_SEG, SEGP = L1_SEGP, NSEG = 15, TP = SPEC ? 16 * SEGP : T + 4;  // T + 4: the two channels' MFMA operands on different banks
  static_assert(T % 4 == 0, "tap count must be a multiple of 4");
  static_assert(RL >= CH + (S > D ? S : D) + 1, "ring too short");
  static_assert(NT * R >= 2 * T, "not enough tap slots");
  static_assert(T4 % 16 == 0 || T4 <= 16, "unsupported tap count");
  __shared__ double prods[2][TP];
  __shared__ double ring[2][RL];
  __shared__ double predv[2];
  // predictions of the current chunk; written to HBM once per chunk by all threads (a store per
  // sample would put an HBM write round trip in front of every barrier)
  __shared__ double pbuf[2][CH];
  // speculative exact summation (SPEC): boundary tables of spec_exact_sum, per channel
  __shared__ int sJ[2][SPEC ? 16 : 1][16];
  __shared__ double sX1[2];
  __shared__ double sA[2][16];  // SPEC: approximate segment sums (the guess of spec_exact_sum)
  // TABG: the single-wave stages (32,16) and (8,4) read the rsqrt table from global memory (16 KB: it stays in the CU's vector L1)
  // instead of keeping a copy of their own in LDS -- 7 KB of LDS per workgroup instead of 23, so that they fit beside TWO lead
  // workgroups (2 x 76.5 KB of the CU's 160)
  constexpr bool TABG = (LMS_TABG >= 1 && NT == 64) || (LMS_TABG >= 2 && NT == 256);
  __shared__ __attribute__((aligned(256))) uint16_t tab_l[TABG ? 1 : RSQ_LDS_MAX];  // (most aligned = first in the LDS layout: its reads then take the base as an immediate offset)
  const uint16_t *const tab = TABG ? gtab : tab_l;
  // HALVES (the (256,128) stage): the 384-term chain of a channel is cut in two. Wave 0 sums the first half exactly from
  // +0 while waves 1 and 2 sum the second half from 2 x 32 candidate inputs per channel around a guess of the first half's sum
  // (rows of the MFMA = channel x 8 candidate groups, its 4 output columns keep their own C: 64 accumulators = 2 x 32
  // candidates in ONE chain of 48 steps per wave). If the exact first half IS one of the candidates, bit for bit, that candidate's
  // output is the exact left-to-right sum of all 384 terms; if not, wave 0 sums the second half from the exact value
  // (48 more steps, as long as the undivided chain). Exactness never rests on the guess.
  constexpr bool HALVES = H_HALVES && HALV && !SPEC && T == 384;
  constexpr bool ALLVALID = !SPEC && NT * R == 2 * T;  // every tap slot is used: no per-slot validity tests (H_FAST)
  __shared__ double hX1[2], hOut[2][64], hGb[2], hIu[2], hU[2], hP2[2];

  const LmsJob job = jobs[blockIdx.x];
  LMS_SPAN(job, ct0, clen)
  const int tid = threadIdx.x;
#ifdef SACOPT_ABLATE
  const int dbg = bits >> 8;  // A/B timing switches of tools/ builds (tools/ablate.py): 2 = skip the update, 16 = single P2 pass
#else
  constexpr int dbg = 0;      // the product library has no switch that changes a result
#endif
  bits &= 0xff;
  if (!TABG) for (int i = tid; i < (2 << bits); i += NT) tab_l[i] = gtab[i];
  for (int i = tid; i < 2 * RL; i += NT) (&ring[0][0])[i] = 0.;
  if (SPEC)
    for (int i = tid; i < 2 * TP; i += NT) (&prods[0][0])[i] = 0.;  // chain 15 of the MFMA passes is idle: keep its operands finite
  __syncthreads();

  // per-slot constants. Carried state: w at [tap], eg at [R*NT + tap], tap = channel*T + i.
  double w[R], eg[R], mu[R];
  int hoff[R], hsel[R], paddr[R], cs[R], tap[R];
  bool valid[R];
  // SPEC: a half-wave owns one segment (128 taps, 4 per lane). NT = 512: of both channels (slots 0..3 channel 0, 4..7
  // channel 1); NT = 1024: half-waves 0..15 channel 0, 16..31 channel 1.
  constexpr bool BOTH = (NT == 512);
  const int hseg = (tid >> 5) & 15, hl = tid & 31, hch = tid >> 9;
#pragma unroll
  for (int r = 0; r < R; r++) {
    int c, i;
    if constexpr (SPEC) {
      // lane l: taps l, 32+l, 64+l, 96+l of the segment: its approximate sum is a half-wave reduction
      c = BOTH ? (r >> 2) : hch;
      i = SEG * hseg + 32 * (r & 3) + hl;
      valid[r] = hseg < NSEG;
    } else {
      const int g = r * NT + tid;
      valid[r] = g < 2 * T;
      c = (g >= T) ? 1 : 0;
      i = g - c * T;
    }
    cs[r] = c;
    tap[r] = c * T + i;
    int src, lag;
    if (i < S) { src = c; lag = 1 + i; mu[r] = job.mu1; }
    else { const int j = i - S; src = 1 - c; lag = (c == 0 ? 1 : 0) + j; mu[r] = job.mu2; }
    hsel[r] = src;
    hoff[r] = (-lag) & M;
    paddr[r] = SPEC ? c * TP + (i / SEG) * SEGP + (i % SEG) : c * TP + i;
    w[r] = 0.; eg[r] = 0.;
    if (!valid[r]) { hsel[r] = 0; hoff[r] = 0; if (!SPEC) paddr[r] = 0; }  // (SPEC: the idle sixteenth segment keeps its own product slots)
    if (t0 > 0 && job.state && valid[r]) {  // resume: weights / gradient energies of the previous chunk
      w[r] = job.state[tap[r]];
      eg[r] = job.state[R * NT + tap[r]];
    }
  }
  if (t0 > 0) {  // history before the chunk comes straight from the stage's input arrays
    for (int idx = tid; idx < 2 * (RL - CH); idx += NT) {
      const int c = idx / (RL - CH), tau = t0 - (RL - CH) + (idx - c * (RL - CH));
      const double *xs = c ? job.x[1] : job.x[0];
      ring[c][tau & M] = (tau >= 0) ? xs[tau] : 0.;
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (SPEC && wave >= 4) __builtin_amdgcn_s_setprio(1);  // the younger wave of a SIMD first (see k_lms1c): optimize 471 -> 465 ms on the 60 s clip
  const int mk = lane >> 4, mc = lane & 15;
  // MFMA operand map: lane (k = lane>>4, chain = lane&15) feeds term 4s+k of its chain at step s.
  // Chains 0/1 are the two channels; chains 2..15 just repeat them (harmless, keeps loads unconditional).
  const double *mp = &prods[mc & 1][mk];

  auto flush = [&](int tb, int te) {  // p[k] and val[k+1] = x - p (CalcError, profile.cpp:141-148) for times [tb, te)
    for (int idx = tid; idx < 2 * CH; idx += NT) {
      const int c = idx / CH, k = idx - c * CH, tau = tb + k;
      if (tau < te) {
        const double pv = pbuf[c][k];
        double *po = c ? job.p[1] : job.p[0], *xo = c ? job.xn[1] : job.xn[0];
        po[tau] = pv;
        if (xo) xo[tau] = ring[c][tau & M] - pv;
      }
    }
  };
  long long hmiss = 0;
#ifdef H_PROF  // phase clocks of every wave of the (256,128) stage (tools/h_phases.py): products | sums until the barrier | look-up (+ second pass) | update
  long long hp[4] = {0, 0, 0, 0}, hpT = clock64();
#define H_TICK(i) { if (HALVES || SPEC) { const long long now_ = clock64(); hp[i] += now_ - hpT; hpT = now_; } }
#else
#define H_TICK(i)
#endif
  for (int t = t0; t < tend; t++) {
    if ((t & (CH - 1)) == 0) {
      if (HALVES) __syncthreads();  // the last prediction went to pbuf after the sample's last barrier
      if (t > t0) flush(t - CH, t);
      for (int idx = tid; idx < 2 * CH; idx += NT) {
        const int c = idx / CH, k = idx - c * CH, tau = t + k;
        const double *xs = c ? job.x[1] : job.x[0];
        ring[c][tau & M] = (tau < tend) ? xs[tau] : 0.;
      }
      __syncthreads();
    }
    // ---- products w[i]*hist[i] (nlms.h:111)
    double h[R], part[2] = {0., 0.};
#pragma unroll
    for (int r = 0; r < R; r++) h[r] = ring[hsel[r]][(t + hoff[r]) & M];  // all reads first (not moved across the product stores otherwise)
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double pr = w[r] * h[r];
      if (ALLVALID || (SPEC && UNCOND_UPDATE) || valid[r]) (&prods[0][0])[paddr[r]] = pr;  // (SPEC: the idle segment's slots take whatever its lanes compute)
      if (SPEC) part[BOTH ? (r >> 2) : 0] += pr;
    }
    if constexpr (SPEC) {
      // approximate segment sums (tree order: a guess only): 16-lane rows by DPP, then row 0 -> 1 and row 2 -> 3
#pragma unroll
      for (int c = 0; c < (BOTH ? 2 : 1); c++) {
        double v = row_prefix16(part[c]);
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
        v += __hiloint2double(hi, lo);
        if (hl == 31 && valid[0]) sA[BOTH ? c : hch][hseg] = v;
      }
    }
    H_TICK(0)
    __syncthreads();
    // ---- pred = 0.; for i: pred += w[i]*hist[i]  -- strict order, on the matrix pipe
    if constexpr (SPEC) {
      static_assert(T == NSEG * SEG && NT * R == 4096 && (NT == 512 || NT == 1024), "SPEC is laid out for the (1280,640) stage, phases on waves 0..7");
      // waves 0..7: channel = wave & 1, candidate group = wave >> 1, so that the two channels' walkers (waves 0, 1)
      // and each channel's four candidate waves sit on different SIMDs
      const int sc = wave & 1, grp = wave < (S2_DUAL ? 4 : 8) ? (wave >> 1) : -1;
      long long rounds_ = 0;
      const double res = spec_exact_sum<NSEG, SEG / 4, SEGP, S2_DUAL>((uint32_t)(uintptr_t)&prods[sc][0], sJ[sc], &sX1[sc], sA[sc], grp, grp == 0, dbg, rounds_, lane);
      if (grp == 0 && lane == SPEC_RESULT_LANE(NSEG)) {
        predv[sc] = res;
        pbuf[sc][t & (CH - 1)] = res;
      }
      H_TICK(1)
    } else if constexpr (HALVES) {
      constexpr int H = H_SPLIT;  // terms of the first part (wave 0, from +0); the candidate waves sum the rest after guessing its sum
      if (wave == 0) {
        const double acc = H == 192 ? mfma_chain_lds<3>((uint32_t)(uintptr_t)mp) : mfma_chain56((uint32_t)(uintptr_t)mp);
        if (lane == 0 || lane == 16) hX1[lane >> 4] = acc;
      } else if (wave >= 1 && wave <= LMS_HALVES_WAVES) {  // 32 candidates per channel each: offsets -32..-1 and 0..31 ulps from the guess
        double g0 = (prods[0][lane] + prods[0][64 + lane]) + prods[0][128 + lane];  // any order: only a guess
        double g1 = (prods[1][lane] + prods[1][64 + lane]) + prods[1][128 + lane];
        if (H > 192 && lane < H - 192) { g0 += prods[0][192 + lane]; g1 += prods[1][192 + lane]; }
        {  // wave sums by DPP: 16-lane rows, then row 0 -> 1 and 2 -> 3, then rows 0..1 -> 2..3: lane 63 has the total
          double gg[2] = {g0, g1};
#pragma unroll
          for (int q = 0; q < 2; q++) {
            double v = row_prefix16(gg[q]);
            int lo = __double2loint(v), hi = __double2hiint(v);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
            v += __hiloint2double(hi, lo);
            lo = __double2loint(v); hi = __double2hiint(v);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
            v += __hiloint2double(hi, lo);
            gg[q] = bcast(v, 63);
          }
          g0 = gg[0]; g1 = gg[1];
        }
        // D-layout lane: chain 4*blk + i (i = lane >> 4, blk = (lane >> 2) & 3), column lane & 3; chain & 1 = channel
        const int i = lane >> 4, blk = (lane >> 2) & 3, chD = i & 1, m = 32 * (wave - 1) + 8 * blk + 4 * (i >> 1) + (lane & 3);
        const double g = chD ? g1 : g0;
        double iu;
        const double u = ulp_pow2(g, &iu);
        const double stv = (g - 16. * LMS_HALVES_WAVES * u) + (double)m * u;
        const double out = H == 192 ? mfma_chain_lds<3>((uint32_t)(uintptr_t)(mp + H), stv) : mfma_chain40((uint32_t)(uintptr_t)(mp + H), stv);
        hOut[chD][m] = out;
        if (m == 0) { hGb[chD] = stv; hIu[chD] = iu; hU[chD] = u; }
      }
      H_TICK(1)
      __syncthreads();
      double pr[2];
      bool miss = false;
      const double xt0 = ring[0][t & M], xt1 = ring[1][t & M];  // (for the update: fetched with the look-up's first round trip)
#pragma unroll
      for (int c = 0; c < 2; c++) {  // which candidate is the exact first part? (its start value recomputed as its wave computed it)
        const double x1 = hX1[c], gb = hGb[c];
        int mi = (int)((x1 - gb) * hIu[c]);
        mi = mi < 0 ? 0 : (mi > 32 * LMS_HALVES_WAVES - 1 ? 32 * LMS_HALVES_WAVES - 1 : mi);
        if (__double_as_longlong(gb + (double)mi * hU[c]) == __double_as_longlong(x1)) pr[c] = hOut[c][mi];
        else miss = true;
      }
      if (miss) {  // uniform: every thread read the same values
        hmiss++;
        if (wave == 0) {
          const double acc = H == 192 ? mfma_chain_lds<3>((uint32_t)(uintptr_t)(mp + H), hX1[(lane >> 4) & 1]) : mfma_chain40((uint32_t)(uintptr_t)(mp + H), hX1[(lane >> 4) & 1]);
          if (lane == 0 || lane == 16) hP2[lane >> 4] = acc;
        }
        __syncthreads();
        pr[0] = hP2[0]; pr[1] = hP2[1];
      }
      if (tid < 2) pbuf[tid][t & (CH - 1)] = tid ? pr[1] : pr[0];
      H_TICK(2)
      // ---- Update (nlms.h:114-129)
      const double e0 = xt0 - pr[0];
      const double e1 = xt1 - pr[1];
      if constexpr (ALLVALID && H_BATCH_RSQ) {  // the taps' chains side by side
        if (!(dbg & 2)) {
          double grad[R], arg[R], rs[R];
#pragma unroll
          for (int r = 0; r < R; r++) {
            grad[r] = (cs[r] ? e1 : e0) * h[r];
            eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad[r] * grad[r]);
            arg[r] = eg[r] + 0.001;
          }
          rsq_emul2_n<R>(arg, rs, tab, bits);
#pragma unroll
          for (int r = 0; r < R; r++) w[r] += (mu[r] * grad[r] * rs[r]);
        }
      } else {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if ((ALLVALID || valid[r]) && !(dbg & 2)) {
          const double err = cs[r] ? e1 : e0;
          const double grad = err * h[r];
          eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad * grad);
          w[r] += (mu[r] * grad * rsq_emul2(eg[r] + 0.001, tab, bits));
        }
      }
      }
      H_TICK(3)
      continue;
    } else
    if (wave == 0) {
      double acc = 0.;
      if constexpr (T4 % 16 == 0) {
        acc = mfma_chain_lds<T4 / 16>((uint32_t)(uintptr_t)mp);
      } else {
        double a[T4];
#pragma unroll
        for (int u = 0; u < T4; u++) a[u] = mp[4 * u];
#pragma unroll
        for (int u = 0; u < T4; u++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], 1.0, acc, 0, 0, 0);
      }
      // chain c = 4*blk + i lives in D lanes 16*i + 4*blk + j: chain 0 -> lane 0, chain 1 -> lane 16
      if (lane == 0 || lane == 16) {
        const int c = lane >> 4;
        predv[c] = acc;
        pbuf[c][t & (CH - 1)] = acc;
      }
    }
    __syncthreads();
    if (SPEC) { H_TICK(2) }
    // ---- Update (nlms.h:114-129)
    const double e0 = ring[0][t & M] - predv[0];
    const double e1 = ring[1][t & M] - predv[1];
#pragma unroll
    for (int r = 0; r < R; r++) {
      if ((UNCOND_UPDATE || valid[r]) && !(dbg & 2)) {  // (slots without a tap update scratch values of their own -- no exec masks)
        const double err = cs[r] ? e1 : e0;
        const double grad = err * h[r];
        eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad * grad);
        w[r] += (mu[r] * grad * rsq_emul2(eg[r] + 0.001, tab, bits));
      }
    }
    if (SPEC) { H_TICK(3) }
  }
  if (HALVES) __syncthreads();
  if (tend > t0) flush((tend - 1) & ~(CH - 1), tend);
#ifdef H_PROF
  if (HALVES || SPEC) { __syncthreads(); if (lane == 0 && tend - t0 >= 64) for (int i = 0; i < 4; i++) job.p[0][t0 + 4 * wave + i] = (double)hp[i]; }
#endif
  if (HALVES && (dbg & 128)) {  // DEBUG: samples whose first half missed the candidates
    __syncthreads();
    if (tid == 0) job.p[0][t0] = (double)hmiss;
  }
  if (job.state) {
#pragma unroll
    for (int r = 0; r < R; r++)
      if (valid[r]) { job.state[tap[r]] = w[r]; job.state[R * NT + tap[r]] = eg[r]; }
  }
}

// ------------------------------------------------------------------------------------------
// LMSADA2(1280,640), both channels of a chain in one workgroup of 512 threads, built so that TWO workgroups share a CU (the
// kernel of launches with more chains than CUs to give each two, where CU-time is what counts). A sample is a VALU-bound
// update of all taps followed by serial sections of one or two waves per channel (candidate pass, walk, verification), so a
// CU that holds one chain idles most of its issue slots; with two resident chains one chain's update runs under the other's
// serial sections. What it takes: at most 128 VGPRs (four waves per SIMD; the generic k_lms<1280,640,512,8,...> instance
// has 169: its per-slot index arrays) and at most 80 KB of LDS (it has 93 KB). So, against that instance:
//   * the history is not a ring: hist[c][k] = x_c(tb - 1280 + k), tb = start of the current block of S2_CH samples; at a block
//     edge the last 1280 values move down and the next block is loaded. A tap's history value is then at a per-thread base
//     address that advances by 8 bytes per sample plus a compile-time offset: no index arithmetic per tap;
//   * slot -> (channel, tap), product address, step size are compile-time functions of the slot number and three per-thread
//     values; 128-sample blocks (pbuf 2 KB). LDS: table 16 KB + products 33 KB + history 22 KB + 5 KB = 76 KB.
// Layout as in k_lms<.., SPEC>: half-wave hseg owns segment hseg of BOTH channels (slots 0..3 channel 0, 4..7 channel 1, taps
// 128 hseg + 32 q + hl), the sixteenth half-wave idles; carried state as k_lms1c (w at [tap], eg at [4096 + tap], tap =
// channel * 1920 + i), so a chain may change kernels from chunk to chunk. Same arithmetic, same sums: bit-identical results.
// ------------------------------------------------------------------------------------------
#define S2_NT 512
#define S2_CH 128
#define S2_HL (1280 + S2_CH)
#ifdef S2_OCC
__device__ double s2occ[4 * 1024];
#endif
__global__ __launch_bounds__(S2_NT, 4) void k_lms2(const LmsJob *__restrict__ jobs, const uint16_t *__restrict__ gtab, int bits, int ct0, int clen) {
  constexpr int S = 1280, T = 1920, NT = S2_NT, CH = S2_CH, HL = S2_HL, R = 8;
  constexpr int SEG = L1_SEG, SEGP = L1_SEGP, NSEG = 15, TP = 16 * SEGP;
  // (the table through the vector L1 instead -- 60 KB of LDS, so that a pair and a (256,128) workgroup fit a CU -- was measured: 3840
  // look-ups per sample are too many for the L1, 75 frames 1215 -> 1285 ms)
  __shared__ __attribute__((aligned(256))) uint16_t tab[RSQ_LDS_MAX];
#ifdef S2_PAD  // (timing builds: so much more LDS that only one workgroup fits a CU)
  __shared__ double s2pad[S2_PAD];
  if (ct0 < 0) s2pad[threadIdx.x] = 1.;
#endif
#ifdef S2_OCC
  const unsigned long long occ_t0 = wall_clock64();
#endif
  __shared__ double prods[2][TP];
  __shared__ double hist[2][HL];
  __shared__ double pbuf[2][CH];
  __shared__ int sJ[2][16][16];
  __shared__ double sX1[2], predv[2];
  __shared__ double sA[2][16];

  const LmsJob job = jobs[blockIdx.x];
  LMS_SPAN(job, ct0, clen)
  const int tid = threadIdx.x;
#ifdef SACOPT_ABLATE
  const int dbg = bits >> 8;
#else
  constexpr int dbg = 0;
#endif
  bits &= 0xff;
  for (int i = tid; i < (2 << bits); i += NT) tab[i] = gtab[i];
  for (int i = tid; i < 2 * TP; i += NT) (&prods[0][0])[i] = 0.;  // chain 15 of the MFMA passes is idle: keep its operands finite

  const int hseg = tid >> 5, hl = tid & 31, lane = tid & 63, wave = tid >> 6;
  const bool own = hseg < S / SEG, valid = hseg < NSEG;
  const int hs = valid ? hseg : NSEG - 1;  // (the idle half-wave reads where segment 14 reads)
  // history: slot (c, q) reads hist_flat[a_c + tl + 96 - 32 q], tl = t - tb (see above)
  //   own taps   (i < 1280):  x_c(t - 1 - i)                       = hist[c][tl + 1279 - i]
  //   other taps (j = i - 1280): c = 0: x_1(t - 1 - j) = hist[1][tl + 1279 - j];  c = 1: x_0(t - j) = hist[0][tl + 1280 - j]
  const int bi = (own ? 1183 : 2464) - SEG * hs - hl;
  uint32_t a0 = (uint32_t)(uintptr_t)&hist[0][0] + 8u * (uint32_t)(own ? bi : bi + HL - 1);
  uint32_t a1 = (uint32_t)(uintptr_t)&hist[0][0] + 8u * (uint32_t)(own ? bi + HL : bi);
  const uint32_t pa = (uint32_t)(uintptr_t)&prods[0][0] + 8u * (uint32_t)(hseg * SEGP + hl);
  const double mu = own ? job.mu1 : job.mu2;
  const int tap0 = SEG * hseg + hl;  // tap of slot (c, q): c * T + tap0 + 32 q
  double w[R], eg[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    w[r] = 0.; eg[r] = 0.;
    if (t0 > 0 && job.state && valid) {  // resume: weights / gradient energies of the previous chunk
      const int tp = (r >> 2) * T + tap0 + 32 * (r & 3);
      w[r] = job.state[tp];
      eg[r] = job.state[4096 + tp];
    }
  }
  // history before the chunk comes straight from the stage's input arrays
  for (int idx = tid; idx < 2 * S; idx += NT) {
    const int c = idx / S, k = idx - c * S, tau = t0 - S + k;
    hist[c][k] = (tau >= 0) ? (c ? job.x[1] : job.x[0])[tau] : 0.;
  }
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the younger wave of a SIMD first (see k_lms1c)

  auto lds_f64 = [](uint32_t addr) { return *(const __attribute__((address_space(3))) double *)(uintptr_t)addr; };
#ifdef H_PROF  // phase clocks of every wave (tools/s_phases.py): products until the barrier | sum until its barrier | wait there | update
  long long hp[4] = {0, 0, 0, 0}, hpT = clock64();
#define S2_TICK(i) { const long long now_ = clock64(); hp[i] += now_ - hpT; hpT = now_; }
#else
#define S2_TICK(i)
#endif
  for (int t = t0; t < tend; t++) {
    const int tl = t & (CH - 1);
    if (tl == 0) {
      if (t > t0) {
        // p[k] and val[k+1] = x - p (CalcError, profile.cpp:141-148) of the block that ended, then the history moves down
        for (int idx = tid; idx < 2 * CH; idx += NT) {
          const int c = idx / CH, k = idx - c * CH, tau = t - CH + k;
          const double pv = pbuf[c][k];
          (c ? job.p[1] : job.p[0])[tau] = pv;
          double *xo = c ? job.xn[1] : job.xn[0];
          if (xo) xo[tau] = hist[c][S + k] - pv;
        }
        double mv[(2 * S + NT - 1) / NT];
#pragma unroll
        for (int u = 0; u < (2 * S + NT - 1) / NT; u++) {
          const int idx = tid + NT * u;
          mv[u] = idx < 2 * S ? hist[idx / S][idx % S + CH] : 0.;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (2 * S + NT - 1) / NT; u++) {
          const int idx = tid + NT * u;
          if (idx < 2 * S) hist[idx / S][idx % S] = mv[u];
        }
        a0 -= 8u * CH; a1 -= 8u * CH;
      }
      for (int idx = tid; idx < 2 * CH; idx += NT) {
        const int c = idx / CH, k = idx - c * CH, tau = t + k;
        hist[c][S + k] = (tau < tend) ? (c ? job.x[1] : job.x[0])[tau] : 0.;
      }
      __syncthreads();
    }
    // ---- products w[i]*hist[i] (nlms.h:111) and the approximate segment sums (tree order: a guess only)
    double h[R];
#pragma unroll
    for (int r = 0; r < R; r++) h[r] = lds_f64((r < 4 ? a0 : a1) + 8u * (uint32_t)(96 - 32 * (r & 3)));  // all reads first
    double part[2];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double pr = w[r] * h[r];
      *(__attribute__((address_space(3))) double *)(uintptr_t)(pa + 8u * (uint32_t)((r >> 2) * TP + 32 * (r & 3))) = pr;  // (the idle half-wave has slots of its own)
      if (r & 3) part[r >> 2] += pr; else part[r >> 2] = pr;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double v = row_prefix16(part[c]);
      int lo = __double2loint(v), hi = __double2hiint(v);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
      hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
      v += __hiloint2double(hi, lo);
      if (hl == 31 && valid) sA[c][hseg] = v;
    }
    a0 += 8u; a1 += 8u;
    S2_TICK(0)
    __syncthreads();
    // ---- pred = 0.; for i: pred += w[i]*hist[i]  -- strict order: waves 0..3 sum candidates (channel = wave & 1, two groups of
    // eight per channel), waves 0 and 1 walk and verify
    {
      const int sc = wave & 1, grp = wave < 4 ? (wave >> 1) : -1;
      long long rounds_ = 0;
      // every lane-derived value of the sum (segment, column, candidate index as a double, operand address) is recomputed per
      // sample from a lane number the compiler cannot see through: hoisted out of the loop they were spilled (128 VGPRs)
      int lane_ = lane;
      asm volatile("" : "+v"(lane_));
      const double res = spec_exact_sum<NSEG, SEG / 4, SEGP, true>((uint32_t)(uintptr_t)&prods[sc][0], sJ[sc], &sX1[sc], sA[sc], grp, grp == 0, dbg, rounds_, lane_);
      if (grp == 0 && lane == SPEC_RESULT_LANE(NSEG)) {
        predv[sc] = res;
        pbuf[sc][tl] = res;
      }
    }
    S2_TICK(1)
    __syncthreads();
    S2_TICK(2)
    // ---- Update (nlms.h:114-129)
    const double e0 = hist[0][S + tl] - predv[0];
    const double e1 = hist[1][S + tl] - predv[1];
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!(dbg & 2)) {  // (the idle half-wave updates scratch values of its own -- no exec masks)
        const double grad = (r < 4 ? e0 : e1) * h[r];
        eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad * grad);
        w[r] += (mu * grad * rsq_emul2(eg[r] + 0.001, tab, bits));
      }
    }
    S2_TICK(3)
  }
  __syncthreads();
#ifdef H_PROF
  if (lane == 0 && tend - t0 >= 64) for (int i = 0; i < 4; i++) pbuf[0][4 * wave + i] = (double)hp[i];  // (flushed below as p[0][tb + ...]: timing builds only)
  __syncthreads();
#endif
  if (tend > t0) {  // the last (possibly short) block
    const int tb = (tend - 1) & ~(CH - 1);
    for (int idx = tid; idx < 2 * CH; idx += NT) {
      const int c = idx / CH, k = idx - c * CH, tau = tb + k;
      if (tau < tend) {
        const double pv = pbuf[c][k];
        (c ? job.p[1] : job.p[0])[tau] = pv;
        double *xo = c ? job.xn[1] : job.xn[0];
        if (xo) xo[tau] = hist[c][S + k] - pv;
      }
    }
  }
  if (job.state && valid) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int tp = (r >> 2) * T + tap0 + 32 * (r & 3);
      job.state[tp] = w[r];
      job.state[4096 + tp] = eg[r];
    }
  }
#ifdef S2_OCC  // timing builds (tools/s2_occ.py): where and when this workgroup was resident
  if (tid == 0 && blockIdx.x < 1024) {
    s2occ[4 * blockIdx.x + 0] = (double)occ_t0;
    s2occ[4 * blockIdx.x + 1] = (double)wall_clock64();
    s2occ[4 * blockIdx.x + 2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    s2occ[4 * blockIdx.x + 3] = (double)__builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
  }
#endif
}
#ifdef S2_OCC
extern "C" int sacopt_s2occ_read(double *host, int n) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(s2occ), sizeof(double) * (size_t)n) != hipSuccess; }
#endif

// ------------------------------------------------------------------------------------------
// LMSADA2(1280,640), one workgroup per (chain, CHANNEL). The two filters of a stereo pair only
// read each other's *inputs* (profile.cpp:128-132), never each other's state, so they are
// independent chains: blockIdx.x = 2*chain + channel. Compared with k_lms<..., SPEC> (both
// channels in one workgroup) the tap update -- VALU-throughput bound -- has twice the SIMDs, and
// the serial phases of one channel no longer wait for the other's verification retries.
//
// Per sample (3 barriers): products w[i]*hist[i] -> LDS (15 segments of 128, stride 130: bank-conflict-free
// MFMA operand reads) | candidate pass (SpecSum: wave 15 -- no taps -- and waves 0..2) | wave 15 walks and publishes the sum the
// walk predicts | update (nlms.h:114-129) with it by the tap waves while wave 15 verifies; redone from the kept weights in the
// rare case that the prediction was not the exact sum. (The 512-thread variant of round 2 -- no spare wave, round 1's order
// candidate pass | walk + verification | update -- is in the history of this file.) Predictions are buffered in LDS and
// written out once per 256 samples.
// ------------------------------------------------------------------------------------------
#define L1_NT 1024
#define L1_R (2048 / L1_NT)
#define L1_CH 256
#define L1_HO (1281 + L1_CH)  // own channel's stage inputs:   ho[k] = x_own(tb - 1281 + k), tb = start of the current block of L1_CH samples
#define L1_HX (641 + L1_CH)   // other channel's stage inputs: hx[k] = x_other(tb - 641 + k)

__global__ __launch_bounds__(L1_NT) void k_lms1c(const LmsJob *__restrict__ jobs, const uint16_t *__restrict__ gtab, int bits, int ct0, int clen) {
  constexpr int S = 1280, T = 1920, NT = L1_NT, R = L1_R, CH = L1_CH;
  constexpr int SEG = L1_SEG, SEGP = L1_SEGP, NSEG = T / SEG;  // 15 segments of 128 products
  __shared__ double prods[16 * SEGP];  // (one buffer also with early prediction: the verifier works from registers)
  __shared__ double sOut[32], vexact;
  __shared__ int specf, vok;
  __shared__ double sA[16];  // approximate segment sums (the guess of spec_exact_sum)
  // The history is not a ring (as in k_lms2): at a block edge the last 1281 / 641 values move down and the next block is loaded, a
  // tap's value is at a per-thread base address that advances by 8 bytes per sample -- no index arithmetic per tap. (One value
  // more than the filter is long: the redo of sample t - 1's update at a block edge reads one sample further back.)
  __shared__ double ho[L1_HO];  // taps 0..1279: lags 1..1280
  __shared__ double hx[L1_HX];  // taps 1280..1919
  __shared__ double pbuf[CH];
  __shared__ int sJ[16][16];
  __shared__ double sX1;
  __shared__ double predv;
  __shared__ uint16_t tab[RSQ_LDS_MAX];
#ifdef SACOPT_ABLATE
  __shared__ int hist[16];
  if (threadIdx.x < 16) hist[threadIdx.x] = 0;
#else
  int *const hist = nullptr;
#endif

  const LmsJob job = jobs[blockIdx.x >> 1];
  const int c = blockIdx.x & 1;
  const double *xo_in = c ? job.x[1] : job.x[0];
  const double *xx_in = c ? job.x[0] : job.x[1];
  double *pout = c ? job.p[1] : job.p[0];
  double *xnout = c ? job.xn[1] : job.xn[0];
  const int lagx = c ? 0 : 1;  // ch0: Predict(x1(t-1)); ch1: Predict(x0(t))  (profile.cpp:128-132)
  LMS_SPAN(job, ct0, clen)
  const int tid = threadIdx.x;
#ifdef SACOPT_ABLATE
  const int dbg = bits >> 8;
#else
  constexpr int dbg = 0;
#endif
  bits &= 0xff;
  for (int i = tid; i < (2 << bits); i += NT) tab[i] = gtab[i];
  for (int i = tid; i < 16 * SEGP; i += NT) prods[i] = 0.;  // chain 15 of the MFMA passes is idle: keep its operands finite
  // history before the chunk comes straight from the stage's input arrays
  for (int k = tid; k < L1_HO - CH; k += NT) { const int tau = t0 - (L1_HO - CH) + k; ho[k] = tau >= 0 ? xo_in[tau] : 0.; }
  for (int k = tid; k < L1_HX - CH; k += NT) { const int tau = t0 - (L1_HX - CH) + k; hx[k] = tau >= 0 ? xx_in[tau] : 0.; }

  double w[R], eg[R];
  // carried state: same layout as k_lms2 (w at [tap], eg at [4096 + tap], tap = channel*1920 + i),
  // so the two kernels may alternate between the chunks of one chain
  double *st = job.state ? job.state + c * T : nullptr;
  // Wave `seg` owns the 128 taps of segment `seg`, lane l the taps l and 64 + l of it: consecutive lanes read consecutive
  // history values, and a segment's approximate sum is a wave reduction. Wave 15 owns no taps.
  constexpr int HW = NT / 16;
  static_assert(HW * R == SEG && R == 2, "slots");
  const int seg = tid / HW, hl = tid % HW;
  const bool valid = seg < NSEG, own = seg < S / SEG;
  const int i0 = SEG * (valid ? seg : NSEG - 1) + hl;  // tap of slot 0; slot 1: i0 + 64  (wave 15 addresses what wave 14 addresses)
  // history value of slot r at time t: own x_own(t - 1 - i) = ho[tl + 1280 - i]; other (j = i - 1280) x_other(t - lagx - j) = hx[tl + 641 - lagx - j]
  const uint32_t a1 = own ? (uint32_t)(uintptr_t)&ho[0] + 8u * (uint32_t)(1280 - i0 - 64) : (uint32_t)(uintptr_t)&hx[0] + 8u * (uint32_t)(641 - lagx - (i0 - S) - 64);
  const double mu = own ? job.mu1 : job.mu2;
  const int pa = seg * SEGP + hl;  // product slot of slot 0; slot 1: + 64
#pragma unroll
  for (int r = 0; r < R; r++) {
    w[r] = 0.; eg[r] = 0.;
    if (t0 > 0 && st && valid) { w[r] = st[i0 + 64 * r]; eg[r] = st[4096 + i0 + 64 * r]; }
  }
  const int lane = tid & 63, wave = tid >> 6;
  long long rounds = 0;
  // Issue priority by age, youngest first: a SIMD serves its four waves oldest first, so the waves 12..14 used to start their
  // latency-bound tail (products + segment sum) only after the other three had finished updating; with the order reversed
  // that tail runs under the older waves' vector work (same-run A/B: 6.70k -> 6.62k cycles per sample)
  if (wave >= 12) __builtin_amdgcn_s_setprio(3); else if (wave >= 8) __builtin_amdgcn_s_setprio(2); else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#ifdef L1_PROF  // phase clocks of every wave (tools/l1_phases.py): products | P1 | walk + verification | update, until its barrier
  long long pf[4] = {0, 0, 0, 0}, pfT = clock64();
#define L1_TICK(i) { const long long now_ = clock64(); pf[i] += now_ - pfT; pfT = now_; }
#else
#define L1_TICK(i)
#endif
  auto lds_f64 = [](uint32_t addr) { return *(const __attribute__((address_space(3))) double *)(uintptr_t)addr; };

  auto flush = [&](int tb, int te, int tcur) {  // p and val[k+1] = x - p (CalcError, profile.cpp:141-148) for times [tb, te); tcur: start of the block the history is based on
    for (int k = tid; k < CH; k += NT) {
      const int tau = tb + k;
      if (tau < te) {
        const double pv = pbuf[k];
        pout[tau] = pv;
        if (xnout) xnout[tau] = ho[tau - tcur + 1281] - pv;
      }
    }
  };

  // ---- Early prediction. Wave 15 owns no taps: it is candidate group 0, the walker and the verifier (waves 0..2 are the other
  // candidate groups). When the walk ends on a candidate of the last boundary, the sum it predicts (sOut) is published at once
  // and the tap waves update -- and multiply for the next sample -- with it while wave 15 runs the verification pass(es). The
  // weights of before are kept for one sample; when the exact sum differs from the prediction (the walk was a guess: a few
  // per cent of the samples) the update and the products are done again. A sample's result is only ever the verified sum.
  const int grp = wave == 15 ? 0 : (wave < 3 ? wave + 1 : -1);
  const bool walker = wave == 15;
  SpecSum<NSEG, SEGP> ss;
  double h[R], wS[R], egS[R];
  long long n_spec = 0, n_redo = 0;  // (L1_PROF statistics)
  int pend = 0;  // the update of sample t - 1 used a prediction: its verification is due behind the next barrier
  auto products = [&](int tl) {  // w[i]*hist[i] (nlms.h:111) -> prods, approximate segment sum -> sA; tl = t - block start
    double part = 0.;
    const uint32_t a = a1 + 8u * (uint32_t)tl;
#pragma unroll
    for (int r = 0; r < R; r++) h[r] = lds_f64(a + 512u * (uint32_t)(1 - r));  // all history reads first: behind a product store the compiler would not move them
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double pr = w[r] * h[r];
      prods[pa + 64 * r] = pr;  // (every wave of this loop carries taps)
      part = r ? part + pr : pr;
    }
    part = row_prefix16(part);  // tree order: a guess only
    int lo = __double2loint(part), hi = __double2hiint(part);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
    part += __hiloint2double(hi, lo);
    lo = __double2loint(part); hi = __double2hiint(part);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
    part += __hiloint2double(hi, lo);
    if (hl == HW - 1) sA[seg] = part;
  };
  auto update = [&](int tl, double pred, const double (&hh)[R]) {  // nlms.h:114-129; tl may be -1 (the redo of the last sample of the block before)
    const double err = ho[1281 + tl] - pred;
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!(dbg & 2)) {
        const double grad = err * hh[r];
        eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad * grad);
        w[r] += (mu * grad * rsq_emul2(eg[r] + 0.001, tab, bits));
      }
    }
  };
  auto settle = [&](int tl, bool again) {  // behind a barrier: was the prediction used for the sample before (tl - 1) the exact sum?
    if (pend && !vok) {  // no (uniform): update it again from the kept weights, with the exact sum
      n_redo++;
      double hold[R];
      const uint32_t a = a1 + 8u * (uint32_t)(tl - 1);
#pragma unroll
      for (int r = 0; r < R; r++) {
        hold[r] = lds_f64(a + 512u * (uint32_t)(1 - r));
        w[r] = wS[r]; eg[r] = egS[r];
      }
      update(tl - 1, vexact, hold);
      if (again) { products(tl); __syncthreads(); }
    }
    pend = 0;
  };
  // Two loops with the same sequence of barriers: the walker's (no taps; the 32 MFMA operands of each of its lanes stay in
  // registers from the candidate pass, so that its verification needs no LDS access while the tap waves are busy in the LDS)
  // and the tap waves'.
  if (walker) {
    Op32 ops;
    for (int t = t0; t < tend; t++) {
      if (((t - t0) & (CH - 1)) == 0) {  // (block edge: the history moves, the next block is loaded)
        if (t > t0) __syncthreads();
        __syncthreads();
      }
      L1_TICK(0)
      __syncthreads();
      if (pend && !vok) { n_redo++; __syncthreads(); }
      ss.template candidates<true, true>((uint32_t)(uintptr_t)&prods[0], sJ, &sX1, sA, sOut, 0, true, lane, &ops);
      __syncthreads();
      double pv = 0.;
      const double so = sOut[lane & 31];  // (fetched with the table: one LDS round trip less behind the walk)
      const int dl = ss.first_walk(sJ, &sX1, lane);
      int sf = 0;
      if ((unsigned)(dl + 8) < 32u) {  // (32 candidates at the last boundary: SpecSum::candidates<.., WIDE>)
        pv = bcast(so, dl + 8);
        sf = 1;
        if (lane == 0) { predv = pv; specf = 1; }
      } else {  // the walk left the candidates at the last boundary: nothing to predict with
        const double ex = ss.template verify<true>(dbg, rounds, lane, hist, &ops);
        if (lane == 0) { predv = ex; specf = 0; pbuf[(t - t0) & (CH - 1)] = ex; }
      }
      L1_TICK(1)
      __syncthreads();
      L1_TICK(2)
      if (sf) {
        const double ex = ss.template verify<true>(dbg, rounds, lane, hist, &ops);
        if (lane == 0) { vok = __double_as_longlong(ex) == __double_as_longlong(pv); vexact = ex; pbuf[(t - t0) & (CH - 1)] = ex; }
      }
      pend = sf;
      n_spec += sf;
      L1_TICK(3)
    }
    __syncthreads();
  } else {
    for (int t = t0; t < tend; t++) {
      const int tl = (t - t0) & (CH - 1);
      if (tl == 0) {
        if (t > t0) {  // the last L1_HO - CH / L1_HX - CH values move to the front (by the NTAP = 960 threads of this loop)
          constexpr int NTAP = NT - 64;
          static_assert(2 * NTAP >= L1_HO - CH && NTAP >= L1_HX - CH, "move");
          double mo[2], mx;
#pragma unroll
          for (int u = 0; u < 2; u++) { const int k = tid + NTAP * u; mo[u] = k < L1_HO - CH ? ho[k + CH] : 0.; }
          mx = tid < L1_HX - CH ? hx[tid + CH] : 0.;
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 2; u++) { const int k = tid + NTAP * u; if (k < L1_HO - CH) ho[k] = mo[u]; }
          if (tid < L1_HX - CH) hx[tid] = mx;
        }
        for (int k = tid; k < CH; k += NT) {
          const int tau = t + k;
          ho[L1_HO - CH + k] = (tau < tend) ? xo_in[tau] : 0.;
          hx[L1_HX - CH + k] = (tau < tend) ? xx_in[tau] : 0.;
        }
        __syncthreads();
      }
      products(tl);
      L1_TICK(0)
      __syncthreads();
      settle(tl, true);
      if (tl == 0 && t > t0) flush(t - CH, t, t);  // (behind the barrier: the last prediction of the block is the verifier's)
      if (grp > 0) ss.template candidates<false, true>((uint32_t)(uintptr_t)&prods[0], sJ, &sX1, sA, sOut, grp, false, lane);
      __syncthreads();
      L1_TICK(1)
      __syncthreads();
      L1_TICK(2)
      const int sf = specf;
#pragma unroll
      for (int r = 0; r < R; r++) { wS[r] = w[r]; egS[r] = eg[r]; }
      update(tl, predv, h);
      pend = sf;
      L1_TICK(3)
    }
    __syncthreads();
    settle(((tend - 1 - t0) & (CH - 1)) + 1, false);
  }
  if (tend > t0) { const int tb = t0 + ((tend - 1 - t0) & ~(CH - 1)); flush(tb, tend, tb); }
#ifdef L1_PROF
  __syncthreads();
  if (lane == 0 && tend - t0 >= 128) { for (int i = 0; i < 4; i++) pout[t0 + 4 * wave + i] = (double)pf[i]; }
  if (tid == 960 && tend - t0 >= 128) { pout[t0 + 64] = (double)n_spec; pout[t0 + 65] = (double)n_redo; pout[t0 + 67] = (double)rounds; }
#endif
  if (dbg & 128) {  // DEBUG: number of P2 passes
    __syncthreads();
    if (tid == 960) pout[t0] = (double)rounds;
  }
#ifdef SACOPT_ABLATE
  if (dbg & 2048) {  // DEBUG: histogram of the largest distance (in ulps) between guess and exact value over the boundaries of a sample
    __syncthreads();
    if (tid < 9) pout[t0 + tid] = (double)hist[tid];
  }
#endif
  if (st && valid) {
#pragma unroll
    for (int r = 0; r < R; r++) { st[i0 + 64 * r] = w[r]; st[4096 + i0 + 64 * r] = eg[r]; }
  }
}

// ------------------------------------------------------------------------------------------
// LPC3(alpha, S=16, D=8, k=8, regp=500) (src/pred/lpc.h:172-228) incl. AutoCov (math/cov.h),
// Cholesky all-orders solve (math/cholesky.cpp:16-77) and BlendErr (pred/blend.h:7-53);
// LPC2(alpha,16,8,k=32,regp=0) (lpc.h:254-293) for the fast profile.
//
// The OLS stage only ever sees input samples (no feedback from any prediction), and inside it
// the only true time recurrences are elementwise:  c[j][i] <- alpha*c[j][i] + h_i*h_j  (cov.h:25),
// b[i] (cov.h:37) and BlendErr's serr[o] (blend.h:34). Everything else -- the Cholesky
// factorisation + all-orders solve every k samples, the all-orders dot products, the weight
// normalisation and the blended prediction -- depends only on those values at one time instant,
// so it is computed in parallel over time, each scalar in the reference's operation order:
//   k_cov   serial in t, one lane per matrix element       -> covariance snapshots every k samples
//   k_chol  one wave per snapshot                           -> sol (all orders) per snapshot
//   k_ssum  one thread per (t, order)                       -> ssum[t][o]        (lpc.h:195-199)
//   k_serr  serial in t, one lane per order                 -> serr[t][o]        (blend.h:32-35)
//   k_pm    one thread per t                                -> p[0](t), val[ch][1](t)
//   ch0: Predict(v1(t-1)) ; ch1: Predict(v0(t))            (profile.cpp:128, :150-158)
// ------------------------------------------------------------------------------------------
#define LPC_N 24
#ifndef CHOL_PK
#define CHOL_PK 1  // k_chol: two snapshots per wave where 2 x (N + 1) rows fit (A/B switch)
#endif
#define LPC_S 16
#define LPC_TRI (LPC_N * (LPC_N + 1) / 2) /* 300 */
#define LPC_SNAP (LPC_TRI + LPC_N)        /* 324: packed lower triangle + b */
#define LPC_RL 1024
#define LPC_CH 512
// The shipped profiles use LPC3/LPC2(alpha, S = 16, D = 8): N = 24 (profile.cpp:5,49,116). The kernels are templates over
// (N, S) -- the reference's classes take S and D as constructor arguments (lpc.h:181,257) -- and are also instantiated for
// N = 32, S = 24 (SACOPT_PROFILE_HIGH32: BASELINE configs[4]'s "32-tap OLS").
#define LPC_TRI_(N) ((N) * ((N) + 1) / 2)
#define LPC_SNAP_(N) (LPC_TRI_(N) + (N))
#define LPC_COV_NT(N) (((LPC_SNAP_(N) + 63) / 64) * 64) /* threads of k_cov: one per covariance / b element (384, 576) */
// MathUtils::S2U (utils.h:56-61). Beyond |v| = 2^30 (a diverged filter) the reference's `2 * (-val)` / `2 * val - 1` overflow;
// its build wraps (oracle/Makefile: -fwrapv, pinned in tests/test_oracle_vs_ref.py), so the arithmetic here is modulo 2^32 by
// construction -- as signed expressions the compiler could assume the result non-negative.
__device__ __forceinline__ int32_t s2u(int32_t v) {
  const uint32_t x = (uint32_t)v;
  return v < 0 ? (int32_t)(2u * (0u - x)) : (v > 0 ? (int32_t)(2u * x - 1u) : 0);
}
// abs() of <cstdlib> as the reference's build computes it: |INT32_MIN| stays INT32_MIN
__device__ __forceinline__ int32_t abs_wrap(int32_t v) { return v < 0 ? (int32_t)(0u - (uint32_t)v) : v; }

// hist[i] at time t for channel ch (cov.h:44-47 shifts expressed as indices into the inputs)
template <int S>
__device__ __forceinline__ double lpc_hist(const int32_t *own, const int32_t *oth, int lagb, int i, int t) {
  const int idx = (i < S) ? (t - 1 - i) : (t - lagb - (i - S));
  const int32_t *src = (i < S) ? own : oth;
  return idx >= 0 ? (double)src[idx] : 0.;
}

// AutoCov::UpdateCov/UpdateB (cov.h:19-43), snapshot after every KS-th Update (lpc.h:213-219)
template <int KS, int N, int S>
__global__ __launch_bounds__(LPC_COV_NT(N)) void k_cov(const LpcJob *__restrict__ jobs) {
  constexpr int RL = LPC_RL, M = RL - 1, CH = LPC_CH, TRI = LPC_TRI_(N), SNAP = LPC_SNAP_(N), NT = LPC_COV_NT(N);
  __shared__ double ring[2][RL];
  const LpcJob &job = jobs[blockIdx.x];
  const int ch = blockIdx.y, oc = 1 - ch, lagb = (ch == 0) ? 1 : 0;
  const int tid = threadIdx.x, n = job.n, t0 = job.t0;
  const double alpha = job.alpha;
  const int32_t *s0 = job.s[0], *s1 = job.s[1];
  double *snap = ch ? job.snap[1] : job.snap[0];
  double *cst = ch ? job.cstate[1] : job.cstate[0];
  for (int i = tid; i < 2 * RL; i += NT) (&ring[0][0])[i] = 0.;
  if (t0 > 0) {  // history before the chunk
    __syncthreads();
    for (int idx = tid; idx < 2 * (RL - CH); idx += NT) {
      const int c = idx / (RL - CH), tau = t0 - (RL - CH) + (idx - c * (RL - CH));
      const int32_t *ss = c ? s1 : s0;
      if (tau >= 0) ring[c][tau & M] = (double)ss[tau];
    }
  }
  // element e: e < TRI -> c[j][i] (packed j(j+1)/2+i); TRI..SNAP-1 -> b[e-TRI]
  const int e = tid;
  const bool isc = e < TRI, isb = !isc && e < SNAP;
  int j = 0, i = 0;
  if (isc) { while ((j + 1) * (j + 2) / 2 <= e) j++; i = e - j * (j + 1) / 2; }
  else if (isb) i = e - TRI;
  auto haddr = [&](int q) -> int {
    if (q < S) return ((-(1 + q)) & M) | (ch << 16);
    return ((-(lagb + (q - S))) & M) | (oc << 16);
  };
  const int ai = haddr(i), aj = haddr(j);
  double cv = (isc && i == j) ? 1.0 : 0.0;  // cov.h:17
  if (t0 > 0 && cst && (isc || isb)) cv = cst[e];
  for (int t = t0; t < n; t++) {
    if ((t & (CH - 1)) == 0) {
      __syncthreads();
      for (int idx = tid; idx < 2 * CH; idx += NT) {
        const int c = idx / CH, k = idx - c * CH, tau = t + k;
        const int32_t *ss = c ? s1 : s0;
        ring[c][tau & M] = (tau < n) ? (double)ss[tau] : 0.;
      }
      __syncthreads();
    }
    const double hi = ring[ai >> 16][(t + (ai & 0xffff)) & M];
    const double hj = isc ? ring[aj >> 16][(t + (aj & 0xffff)) & M] : ring[ch][t & M];  // b: val*hist[i]
    if (isc) cv = alpha * cv + (hi * hj);
    else if (isb) cv = alpha * cv + (hj * hi);
    if (((t + 1) % KS) == 0 && (isc || isb)) snap[(size_t)(t / KS) * SNAP + e] = cv;
  }
  if (cst && (isc || isb)) cst[e] = cv;
}

// Cholesky::Factor + Solve for one snapshot (cholesky.cpp:16-77). Lane r owns matrix row r;
// lane N carries CalcY (:37-46) as an extra row. Each scalar's inner sum runs in the
// reference's k order. FAST: full-order Solve(b,x) only, regp = 0; else all orders, regp = 500.
// PK (N + 1 <= 32): TWO snapshots per wave, one per 32-lane half (snap / solo are then per-half pointers, `live` says whether
// the half has a snapshot at all) -- the same instructions serve both, a broadcast is a ds_bpermute within the half instead of
// a v_readlane. 25 of 64 lanes carried a row before, 50 do now: per snapshot half the instructions (k_chol is issue-bound:
// profiles/r03b_pmc_75frames.json, 12 % of the vector-ALU cycles of a 75-frame call).
template <bool FAST, int N = LPC_N, bool PK = false>
__device__ __forceinline__ bool chol_wave(const double *snap, double *solo, int lane_, bool live = true) {  // true = "matrix indefinit", solo untouched
  constexpr int TRI = LPC_TRI_(N);
  constexpr double REGP = FAST ? 0. : 500.;
  static_assert(N < 64, "one lane per matrix row + one for CalcY");
  static_assert(!PK || N < 32, "a half-wave per snapshot");
  const int lane = PK ? (lane_ & 31) : lane_;
  const int half4 = (lane_ & 32) << 2;  // byte address of the half's lane 0 for ds_bpermute
  auto bc = [&](double v, int src) -> double {
    if constexpr (PK) {
      const int lo = __builtin_amdgcn_ds_bpermute(half4 + 4 * src, __double2loint(v));
      const int hi = __builtin_amdgcn_ds_bpermute(half4 + 4 * src, __double2hiint(v));
      return __hiloint2double(hi, lo);
    } else return bcast(v, src);
  };
  double m[N];
  const int row = lane <= N ? lane : 0;
#pragma unroll
  for (int q = 0; q < N; q++) {
    double v = 0.;
    if (live) {
      if (row < N) { if (q <= row) v = snap[row * (row + 1) / 2 + q]; }
      else v = snap[TRI + q];
    }
    if (q == lane) v += REGP;  // cholesky.cpp:20
    m[q] = v;
  }
  bool fail = !live;
#pragma unroll
  for (int j = 0; j < N; j++) {
    if (!fail) {
      double sum = m[j];
#pragma unroll
      for (int q = 0; q < j; q++) sum -= (m[q] * bc(m[q], j));
      const double dj = bc(sum, j);
      if (dj > 0.) {
        const double diag = sqrt(dj);
        m[j] = (lane == j) ? diag : sum / diag;
      } else fail = true;  // "matrix indefinit": the previous solution stays in force (lpc.h:216-218)
    }
  }
  if (fail) return true;
  double sol[N];
#pragma unroll
  for (int i = 0; i < N; i++) sol[i] = 0.;
  if constexpr (!FAST) {  // Solve(b) for every leading order (cholesky.cpp:63-77): lane o = order o
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
      double sum = bc(m[i], N);
#pragma unroll
      for (int j = i + 1; j < N; j++) {
        const double mji = bc(m[i], j);
        if (j <= lane) sum -= (mji * sol[j]);
      }
      const double d = bc(m[i], i);
      if (i <= lane) sol[i] = sum / d;
    }
    if (lane < N) {
#pragma unroll
      for (int i = 0; i < N; i++)
        if (i <= lane) solo[lane * (lane + 1) / 2 + i] = sol[i];
    }
  } else {  // Solve(b, x) (cholesky.cpp:48-60), uniform across lanes
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
      double sum = bc(m[i], N);
#pragma unroll
      for (int j = i + 1; j < N; j++) sum -= (bc(m[i], j) * sol[j]);
      sol[i] = sum / bc(m[i], i);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < N; i++) solo[i] = sol[i];
    }
  }
  return false;
}
template <bool FAST, int N>
__global__ __launch_bounds__(64) void k_chol(const LpcJob *__restrict__ jobs) {
  constexpr bool PK = CHOL_PK && N < 32;  // two snapshots per wave
  const LpcJob &job = jobs[blockIdx.y >> 1];
  const int lane = threadIdx.x;
  const int ch = blockIdx.y & 1, k = job.t0 / (FAST ? 32 : 8) + (PK ? 2 * blockIdx.x + (lane >> 5) : blockIdx.x);
  const bool live = k < job.nsnap;
  if (!PK && !live) return;
  if (PK && job.t0 / (FAST ? 32 : 8) + 2 * (int)blockIdx.x >= job.nsnap) return;  // (uniform: neither half has a snapshot)
  const double *snap = (ch ? job.snap[1] : job.snap[0]) + (size_t)k * LPC_SNAP_(N);
  double *solo = (ch ? job.sol[1] : job.sol[0]) + (size_t)k * (FAST ? N : LPC_TRI_(N));
  int32_t *failo = (ch ? job.fail[1] : job.fail[0]) + k;
  const bool fail = chol_wave<FAST, N, PK>(snap, solo, lane, live);
  if ((PK ? (lane & 31) : lane) == 0 && live) *failo = fail ? 1 : 0;
}

// index of the snapshot whose solution is in force when predicting sample t (-1: none yet)
template <int KS>
__device__ __forceinline__ int lpc_sol_index(const int32_t *fail, int t) {
  int k = t / KS - 1;  // snapshot k is taken after Update(t = KS*(k+1)-1)
  while (k >= 0 && fail[k]) k--;
  return k;
}

// LPC3::Predict all-orders sums (lpc.h:195-199): thread per (t, order)
template <int N, int S>
__global__ __launch_bounds__(256) void k_ssum(const LpcJob *__restrict__ jobs) {
  const LpcJob &job = jobs[blockIdx.y >> 1];
  const int ch = blockIdx.y & 1, n = job.n;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int t = job.t0 + (int)(gid / N), o = (int)(gid % N);
  if (t >= n) return;
  const int32_t *own = ch ? job.s[1] : job.s[0], *oth = ch ? job.s[0] : job.s[1];
  const int lagb = (ch == 0) ? 1 : 0;
  const int k = lpc_sol_index<8>(ch ? job.fail[1] : job.fail[0], t);
  double sum = 0.;
  if (k >= 0) {
    const double *sol = (ch ? job.sol[1] : job.sol[0]) + (size_t)k * LPC_TRI_(N) + o * (o + 1) / 2;
    for (int i = 0; i <= o; i++) sum += sol[i] * lpc_hist<S>(own, oth, lagb, i, t);
  }
  (ch ? job.ss[1] : job.ss[0])[(size_t)t * N + o] = sum;
}

// BlendErr running squared errors (blend.h:32-35): serial in t, lane per order
template <int N>
__global__ __launch_bounds__(64) void k_serr(const LpcJob *__restrict__ jobs) {
  const LpcJob &job = jobs[blockIdx.x];
  const int ch = blockIdx.y, lane = threadIdx.x, n = job.n;
  if (lane >= N) return;
  const int32_t *own = ch ? job.s[1] : job.s[0];
  const double *ss = ch ? job.ss[1] : job.ss[0];
  double *so = ch ? job.serr[1] : job.serr[0];
  double *est = ch ? job.estate[1] : job.estate[0];
  double serr = (job.t0 > 0 && est) ? est[lane] : 0.;
  int t = job.t0;
  for (; t + 8 <= n; t += 8) {
    double q[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double e = ((double)own[t + u] - ss[(size_t)(t + u) * N + lane]);
      q[u] = (1. - 0.98) * (e * e);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      serr = 0.98 * serr + q[u];
      so[(size_t)(t + u) * N + lane] = serr;
    }
  }
  for (; t < n; t++) {
    const double e = ((double)own[t] - ss[(size_t)t * N + lane]);
    serr = 0.98 * serr + (1. - 0.98) * (e * e);
    so[(size_t)t * N + lane] = serr;
  }
  if (est) est[lane] = serr;
}

// BlendErr weights (blend.h:38-45) + blended prediction (blend.h:15-17): thread per t
template <int N>
__global__ __launch_bounds__(256) void k_pm(const LpcJob *__restrict__ jobs) {
  const LpcJob &job = jobs[blockIdx.y >> 1];
  const int ch = blockIdx.y & 1, n = job.n;
  const int t = job.t0 + blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int32_t *own = ch ? job.s[1] : job.s[0];
  const double *ss = (ch ? job.ss[1] : job.ss[0]) + (size_t)t * N;
  double pm = 0.;
  if (t > 0) {  // weights as left by Update(t-1); before the first update they are all zero
    const double *se = (ch ? job.serr[1] : job.serr[0]) + (size_t)(t - 1) * N;
    double w[N];
    double wsum = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const double tt = 10000. / (se[i] + 0.0001);
      w[i] = tt * tt;
      wsum += w[i];
    }
    if (wsum > 0) {
#pragma unroll
      for (int i = 0; i < N; i++) w[i] /= wsum;
    }
#pragma unroll
    for (int i = 0; i < N; i++) pm += w[i] * ss[i];
  }
  const double v = (double)own[t];
  (ch ? job.p[1] : job.p[0])[t] = pm;
  (ch ? job.x1[1] : job.x1[0])[t] = v - pm;  // val[ch][1] = val[ch][0] - p[0] (profile.cpp:144)
}

// Fast profile: LPC2::Predict (lpc.h:261-267) + residual (profile.cpp:30-31), thread per t
__global__ __launch_bounds__(256) void k_pred2(const LpcJob *__restrict__ jobs) {
  const LpcJob &job = jobs[blockIdx.y >> 1];
  const int ch = blockIdx.y & 1, n = job.n;
  const int t = job.t0 + blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int32_t *own = ch ? job.s[1] : job.s[0], *oth = ch ? job.s[0] : job.s[1];
  const int lagb = (ch == 0) ? 1 : 0;
  const int k = lpc_sol_index<32>(ch ? job.fail[1] : job.fail[0], t);
  double p = 0.;
  if (k >= 0) {
    const double *sol = (ch ? job.sol[1] : job.sol[0]) + (size_t)k * LPC_N;
    for (int i = 0; i < LPC_N; i++) p += sol[i] * lpc_hist<LPC_S>(own, oth, lagb, i, t);
  }
  const int32_t e = cvt_i32_x86((double)own[t] - round(p));
  const int32_t ue = s2u(e);
  int32_t *eo = ch ? job.e[1] : job.e[0], *uo = ch ? job.u[1] : job.u[0];
  if (eo) eo[t] = e;
  if (uo) uo[t] = ue;
  if (job.umax) atomicMax(&job.umax[ch], ue);
}

// ------------------------------------------------------------------------------------------
// Stage mixer + residual: RLS(NS, alpha) (src/pred/rls.cpp:61-108) over the cumulative stage sums
// (profile.cpp:134), prediction rounding and residual (profile.cpp:152-157), S2U and max
// (codec.cpp:188-211). The two channels' mixers are independent: one UNIT per (chain, channel).
//
// A wave issues one FP64 instruction every ~14 cycles whether or not it depends on the previous
// one, so the ~260 scalar operations of one RLS step are spread over 16 lanes: lane (i, j) of a
// unit owns P[i][j]; row sums come from quad broadcasts (DPP), the column / vector gathers from
// cross-lane reads. Every scalar is still produced by the reference's operation sequence (each
// sum starts at 0. and adds its terms in index order). 4 units per wave.
// ------------------------------------------------------------------------------------------
template <int JJ>
__device__ __forceinline__ double quad_bcast(double v) {  // value of lane JJ of this lane's quad
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, JJ * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, JJ * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int NS>
__device__ __forceinline__ double pick(const double (&a)[NS], int k) {  // a[k] for a register array (0. out of range)
  double r = 0.;
#pragma unroll
  for (int q = 0; q < NS; q++) r = (k == q) ? a[q] : r;
  return r;
}

template <int NS>
__global__ __launch_bounds__(64) void k_mix(const MixJob *__restrict__ jobs, int njobs, int ct0, int clen) {
  const int lane = threadIdx.x, i = (lane >> 2) & 3, j = lane & 3, base16 = lane & ~15;
  const int unit = blockIdx.x * 4 + (lane >> 4);
  bool live = unit < 2 * njobs;
  const MixJob &job = jobs[live ? (unit >> 1) : 0];
  // (clen > 0: chunk [ct0, ct0 + clen) of a job that describes the whole window, see LMS_SPAN; a unit whose window ends before the
  // chunk is not live: it stores nothing)
  const int ch = unit & 1, t0 = clen > 0 ? ct0 : job.t0;
  const int nspan = clen > 0 ? (job.n - ct0 < clen ? job.n - ct0 : clen) : job.n;
  if (clen > 0 && ct0 > 0 && nspan <= 0) live = false;
  const int n = live ? (nspan > 0 ? nspan : 0) : 0;
  const double alpha = job.alpha;
  const int32_t *s = job.s[ch];
  const double *p0 = job.p0[ch], *x1 = job.x1[ch];
  const double *pk[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) pk[k] = job.pk[k][ch];
  int32_t *eo = job.e[ch], *uo = job.u[ch];
  double *po = job.pout[ch];
  const bool direct = job.direct != 0;
  const bool inm = i < NS && j < NS, writer = (lane & 15) == 0;
  double w[NS], P = (inm && i == j) ? 1. : 0.;
#pragma unroll
  for (int k = 0; k < NS; k++) w[k] = 0.;
  int32_t umax = 0;
  double *stp = job.state ? job.state + ch * MIX_STATE : nullptr;
  if (live && t0 > 0 && stp) {
#pragma unroll
    for (int k = 0; k < NS; k++) w[k] = stp[k];
    if (inm) P = stp[4 + i * NS + j];
    umax = (int32_t)stp[20];
  }
  int nmax = n;  // wave-uniform trip count: the longest of the 4 units
  nmax = max(nmax, __shfl_xor(nmax, 16));
  nmax = max(nmax, __shfl_xor(nmax, 32));
  const double ia = 1. / alpha;
  // inputs of the next step are fetched one step ahead
  double npk[NS], np0 = 0., nx1 = 0.;
  int32_t ns = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) npk[k] = 0.;
  if (n > 0) {
#pragma unroll
    for (int k = 0; k < NS; k++) npk[k] = pk[k][t0];
    np0 = p0[t0]; nx1 = x1[t0]; ns = s[t0];
  }
  for (int it = 0; it < nmax; it++) {
    if (it < n) {
      const int t = t0 + it;
      double h[NS];
      h[0] = npk[0];
#pragma unroll
      for (int k = 1; k < NS; k++) h[k] = direct ? npk[k] : h[k - 1] + npk[k];  // pv (profile.cpp:134)
      const double cp0 = np0, cx1 = nx1;
      const int32_t v = ns;
      if (it + 1 < n) {
#pragma unroll
        for (int k = 0; k < NS; k++) npk[k] = pk[k][t + 1];
        np0 = p0[t + 1]; nx1 = x1[t + 1]; ns = s[t + 1];
      }
      double pc[NS];  // column j of P (before the update): P[k][j]
#pragma unroll
      for (int k = 0; k < NS; k++) pc[k] = __shfl(P, base16 + 4 * k + j);
      double p = 0.;
#pragma unroll
      for (int k = 0; k < NS; k++) p += h[k] * w[k];  // RLS::Predict (rls.cpp:67-72)
      const int32_t q = cvt_i32_x86(round(cp0 + p));
      const int32_t e = (int32_t)((uint32_t)v - (uint32_t)q);  // (q = INT32_MIN from a diverged filter: wraps like the reference's build)
      const int32_t ue = s2u(e);
      if (writer) {
        if (eo) eo[t] = e;
        if (uo) uo[t] = ue;
        if (po) po[t] = p;
      }
      umax = ue > umax ? ue : umax;
      // RLS::Update(val[ch][1]) (rls.cpp:80-108)
      const double pr = P * pick<NS>(h, j);  // P[i][j]*hist[j]
      double vt = 0.;                        // vt1[i] = sum_j P[i][j]*hist[j]
      vt += quad_bcast<0>(pr);
      vt += quad_bcast<1>(pr);
      vt += quad_bcast<2>(pr);
      if (NS > 3) vt += quad_bcast<3>(pr);
      double v1[NS];
#pragma unroll
      for (int k = 0; k < NS; k++) v1[k] = __shfl(vt, base16 + 4 * k);
      double inner = 0.;
#pragma unroll
      for (int k = 0; k < NS; k++) inner += h[k] * v1[k];
      const double sc = 1. / (alpha + inner);
      double kk[NS];
#pragma unroll
      for (int k = 0; k < NS; k++) kk[k] = sc * v1[k];
      const double ki = pick<NS>(kk, i);
      double m2 = 0.;  // m2[i][j] = sum_k (k[i]*hist[k])*P[k][j]
#pragma unroll
      for (int k = 0; k < NS; k++) m2 += (ki * h[k]) * pc[k];
      P = ia * (P - m2);
      const double dv = cx1 - p;
#pragma unroll
      for (int k = 0; k < NS; k++) w[k] = w[k] + dv * kk[k];
    }
  }
  if (!live) return;
  if (writer && job.umax) job.umax[ch] = umax;
  if (stp) {
    if (writer) {
#pragma unroll
      for (int k = 0; k < NS; k++) stp[k] = w[k];
      stp[20] = (double)umax;
    }
    if (inm) stp[4 + i * NS + j] = P;
  }
}

// ------------------------------------------------------------------------------------------
// CostGolomb::Calc (src/codec.cpp:315-339) over a list of residual buffers in call order, the
// running mean carried from buffer to buffer exactly as one CostGolomb instance does (:364).
// One wave per frame: lane 0 walks the RunExp recurrence (the only true serial dependency),
// all lanes do the divisions / bit counts. nbits is a sum of small integers, exact in FP64 in
// any order, so it is accumulated in int64.
// ------------------------------------------------------------------------------------------
#define COST_BLK 512
struct CostLds {
  double cval[COST_BLK];
  int32_t uval[COST_BLK];
  double smean[COST_BLK];
  long long red[64];
};
// One buffer: returns sum of bits (valid in lane 0), advances `mean` (lane 0). count_bits = false: recurrence only.
__device__ __forceinline__ long long cost_buffer(const int32_t *e, int from, int n, double &mean, bool count_bits, int lane,
                                                 const int *__restrict__ pow2_bits, CostLds &L) {
  long long nb = 0;
  for (int b0 = from; b0 < n; b0 += COST_BLK) {
    const int cnt = min(COST_BLK, n - b0);
    for (int k = lane; k < cnt; k += 64) {
      const int32_t val = s2u(e[b0 + k]);
      L.uval[k] = val;
      L.cval[k] = (1. - 0.98) * val;  // RunExp::Update (utils.h:13-15)
    }
    __syncthreads();
    if (lane == 0) {  // the RunExp recurrence alone: 2 dependent FP64 ops per sample
      int k = 0;
      for (; k + 8 <= cnt; k += 8) {
        double cv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) cv[u] = L.cval[k + u];
#pragma unroll
        for (int u = 0; u < 8; u++) { L.smean[k + u] = mean; mean = 0.98 * mean + cv[u]; }
      }
      for (; k < cnt; k++) { L.smean[k] = mean; mean = 0.98 * mean + L.cval[k]; }
    }
    __syncthreads();
    if (count_bits)
      for (int k = lane; k < cnt; k += 64) {
        const int mi = (int)L.smean[k];
        const int m = mi > 1 ? mi : 1;
        nb += L.uval[k] / m + 1;
        if (m > 1) {  // ceil(log(m)/log(2)) with the host libm's value at exact powers of two
          const int fl = 31 - __clz(m);
          nb += ((m & (m - 1)) == 0) ? pow2_bits[fl] : fl + 1;
        }
      }
    __syncthreads();
  }
  if (!count_bits) return 0;
  L.red[lane] = nb;
  __syncthreads();
  long long tot = 0;
  if (lane == 0) for (int i = 0; i < 64; i++) tot += L.red[i];
  __syncthreads();
  return tot;
}

// Sequential form: one wave per frame walks all buffers in call order. Also the fallback of the
// buffer-parallel form below (returns at once when that one verified: *job.ok != 0).
__global__ __launch_bounds__(64) void k_cost(const CostJob *__restrict__ jobs, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  const CostJob job = jobs[blockIdx.x];
  if (job.ok && *job.ok) return;
  const int lane = threadIdx.x, n = job.n;
  double mean = job.state ? *job.state : 0.;
  const bool single = job.nbuf < 0;
  const int nbuf = single ? 1 : job.nbuf;
  for (int b = 0; b < nbuf; b += 2) {
    double c[2] = {0., 0.};
    for (int h = 0; h < (single ? 1 : 2); h++) {
      const long long tot = cost_buffer(job.bufs[b + h], 0, n, mean, true, lane, pow2_bits, L);
      c[h] = n ? (double)tot / (double)n : 0.;
    }
    if (lane == 0) job.costs[b >> 1] = single ? c[0] : (c[0] + c[1]) / 2.;  // GetCost (codec.cpp:349-351)
  }
  if (lane == 0 && job.state) *job.state = mean;
}

// Buffer-parallel form. The running mean forgets its start: after W samples the influence of the
// initial value is 0.98^W (< 1e-35 for W = 4096) and, both trajectories being rounded onto the same
// grid with the same inputs, they coalesce bit for bit. So: (tail) the mean after buffer b is
// GUESSED by running the recurrence over the buffer's last W samples from 0; (par) every buffer is
// costed in parallel from the guessed end of its predecessor; (fin) each buffer's actual end value
// must EQUAL the guess its successor started from, bit for bit -- then, by induction from the
// caller's initial state, every buffer saw exactly the state the sequential instance would have
// given it. Any mismatch leaves *ok = 0 and the sequential kernel recomputes the frame.
#define COST_WARM 4096
__global__ __launch_bounds__(64) void k_cost_tail(const CostJob *__restrict__ jobs, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  const CostJob job = jobs[blockIdx.y];
  const int b = blockIdx.x, n = job.n;
  if (b >= job.nbuf - 1) return;
  double mean = 0.;
  const int from = n > COST_WARM ? ((n - COST_WARM) & ~7) : 0;
  cost_buffer(job.bufs[b], from, n, mean, false, threadIdx.x, pow2_bits, L);
  if (threadIdx.x == 0) job.scr[b] = mean;  // F[b]
}
__global__ __launch_bounds__(64) void k_cost_par(const CostJob *__restrict__ jobs, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  const CostJob job = jobs[blockIdx.y];
  const int b = blockIdx.x, n = job.n, nbuf = job.nbuf;
  if (b >= nbuf) return;
  double mean = b ? job.scr[b - 1] : (job.state ? *job.state : 0.);
  const long long tot = cost_buffer(job.bufs[b], 0, n, mean, true, threadIdx.x, pow2_bits, L);
  if (threadIdx.x == 0) {
    job.scr[nbuf + b] = mean;                                   // E[b]
    job.scr[2 * nbuf + b] = n ? (double)tot / (double)n : 0.;   // Calc(buffer b)
  }
}
__global__ void k_cost_fin(const CostJob *__restrict__ jobs, int njobs) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= njobs) return;
  const CostJob job = jobs[f];
  const int nbuf = job.nbuf;
  bool ok = true;
  for (int b = 0; b + 1 < nbuf; b++) ok = ok && (__double_as_longlong(job.scr[b]) == __double_as_longlong(job.scr[nbuf + b]));
  if (ok) {
    for (int b = 0; b < nbuf; b += 2) job.costs[b >> 1] = (job.scr[2 * nbuf + b] + job.scr[2 * nbuf + b + 1]) / 2.;  // GetCost (codec.cpp:349-351)
    if (job.state) *job.state = job.scr[2 * nbuf - 1];
  }
  *job.ok = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// CostGolomb::Calc of ONE buffer, segment-parallel, for sacopt_cost_golomb_noalloc (include/sacopt.h): the buffer is cut into
// segments of COST_SEG samples and the scheme of k_cost_tail/par/fin above is applied to the segments: the mean at the end
// of segment b is GUESSED from its last COST_WARM samples, every segment is costed from the guessed end of its predecessor
// (segment 0: the caller's state), and the result stands only if every actual end EQUALS the guess its successor started
// from (induction from the caller's state); otherwise one wave walks the whole buffer. The bit counts are integers (exact in
// FP64 in any order, < 2^53), so nbits = sum of the segments' sums.
// ctl: [0] state (in/out), [1] cost (out), [2] verified flag (as double), [8 + b] F, [8 + NS + b] E, [8 + 2 NS + b] bits
// ------------------------------------------------------------------------------------------
#define COST_SEG 16384
#define NOALLOC_MAX_N (1 << 20)                     /* samples per call */
#define NOALLOC_MAX_SEG (NOALLOC_MAX_N / COST_SEG)  /* 64 */
#define NOALLOC_CTL_DOUBLES (8 + 3 * NOALLOC_MAX_SEG)
#define NOALLOC_STAGE_BYTES ((size_t)NOALLOC_MAX_N * 4)
#define NOALLOC_BYTES (NOALLOC_STAGE_BYTES + NOALLOC_CTL_DOUBLES * 8)
// The scratch of the entry point that allocates nothing: staged input (host callers) + the control block above. One copy
// per GPU, for the life of the process. In host code this NAME is only a handle: its device address is taken once per
// device with hipGetSymbolAddress (sacopt_noalloc_scratch below) and only that pointer is ever handed to a kernel or a copy.
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

__global__ __launch_bounds__(64) void k_cseg_tail(const int32_t *__restrict__ e, int n, double *__restrict__ ctl, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  const int b = blockIdx.x, end = min(n, (b + 1) * COST_SEG);  // only segments that have a successor: end = (b+1)*COST_SEG < n
  double mean = 0.;
  cost_buffer(e, end - COST_WARM, end, mean, false, threadIdx.x, pow2_bits, L);
  if (threadIdx.x == 0) ctl[8 + b] = mean;
}
__global__ __launch_bounds__(64) void k_cseg_par(const int32_t *__restrict__ e, int n, int nseg, double *__restrict__ ctl, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  const int b = blockIdx.x, from = b * COST_SEG, end = min(n, from + COST_SEG);
  double mean = b ? ctl[8 + b - 1] : ctl[0];
  const long long tot = cost_buffer(e, from, end, mean, true, threadIdx.x, pow2_bits, L);
  if (threadIdx.x == 0) {
    ctl[8 + NOALLOC_MAX_SEG + b] = mean;
    ctl[8 + 2 * NOALLOC_MAX_SEG + b] = (double)tot;
  }
}
// one wave: verification and, when it fails, the sequential walk
__global__ __launch_bounds__(64) void k_cseg_fin(const int32_t *__restrict__ e, int n, int nseg, double *__restrict__ ctl, const int *__restrict__ pow2_bits) {
  __shared__ CostLds L;
  bool ok = true;
  for (int b = 0; b + 1 < nseg; b++) ok = ok && (__double_as_longlong(ctl[8 + b]) == __double_as_longlong(ctl[8 + NOALLOC_MAX_SEG + b]));
  double mean, nbits = 0.;
  if (ok) {
    for (int b = 0; b < nseg; b++) nbits += ctl[8 + 2 * NOALLOC_MAX_SEG + b];
    mean = ctl[8 + NOALLOC_MAX_SEG + nseg - 1];
  } else {
    mean = ctl[0];
    nbits = (double)cost_buffer(e, 0, n, mean, true, threadIdx.x, pow2_bits, L);  // valid in lane 0
  }
  if (threadIdx.x == 0) {
    ctl[0] = mean;
    ctl[1] = nbits / (double)n;  // codec.cpp:335
    ctl[2] = ok ? 1. : 0.;
  }
}

// CostMeanRMS::Calc (codec.cpp:302-312) for a list of buffers: blockIdx.x = buffer; sum of |e| in integers (exact), / n
__global__ __launch_bounds__(256) void k_meanrms_batch(const int32_t *const *__restrict__ bufs, const int32_t *__restrict__ ns, double *__restrict__ out) {
  __shared__ long long red[256];
  const int32_t *e = bufs[blockIdx.x];
  const int n = ns[blockIdx.x];
  long long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += abs_wrap(e[i]);  // (codec.cpp:307: abs of an int)
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long tot = 0;
    for (int i = 0; i < 256; i++) tot += red[i];
    out[blockIdx.x] = n ? (double)tot / (double)n : 0.;
  }
}
// out[i] = *ptrs[i] (the per-chain maxima of the s2u residuals, scattered over the arena)
__global__ void k_gather_i32(const int32_t *const *__restrict__ ptrs, int n, int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *ptrs[i];
}

// single buffer variant used by sacopt_cost_golomb / meanrms
__global__ void k_meanrms(const int32_t *e, int n, double *out) {
  // CostMeanRMS::Calc (codec.cpp:302-312): sum of |e| as integers (exact in FP64), / n
  __shared__ long long red[256];
  long long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += abs_wrap(e[i]);  // (codec.cpp:307: abs of an int)
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long tot = 0;
    for (int i = 0; i < 256; i++) tot += red[i];
    *out = n ? (double)tot / (double)n : 0.;
  }
}

// ------------------------------------------------------------------------------------------
// Remap (src/coder/map.cpp:89-152): Analyse marks the sample values a frame uses; Map(pred,err)
// counts the used values in (pred, pred+err] (err > 0) or [pred+err, pred) (err < 0). The count
// is a rank difference on the used-value table, so: mark -> prefix sum -> O(1) per sample.
// blockIdx.y = job (one per frame x channel).
// ------------------------------------------------------------------------------------------
__global__ void k_map_mark(const MapJob *jobs) {  // Remap::Analyse (map.cpp:111-130); tab zeroed beforehand
  const MapJob job = jobs[blockIdx.y];
  if (blockIdx.x == 0 && threadIdx.x == 0) job.tab[MAP_SCALE] = 1;  // isUsed(0) is always true (map.cpp:138)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < job.numsamples; i += gridDim.x * blockDim.x) {
    const int v = job.s[i];
    if (v != 0 && v >= -MAP_SCALE && v <= MAP_SCALE) job.tab[v + MAP_SCALE] = 1;  // idempotent store; |v| > scale: "val too large", not marked
  }
}

#define MAP_SCAN_NT 1024
#define MAP_SCAN_PER ((MAP_TAB + MAP_SCAN_NT - 1) / MAP_SCAN_NT)
__global__ void __launch_bounds__(MAP_SCAN_NT) k_map_scan(const MapJob *jobs) {  // in-place inclusive prefix sum of tab
  __shared__ int part[MAP_SCAN_NT];
  int32_t *tab = jobs[blockIdx.x].tab;
  const int lo = threadIdx.x * MAP_SCAN_PER;
  const int hi = lo + MAP_SCAN_PER < MAP_TAB ? lo + MAP_SCAN_PER : MAP_TAB;
  int s = 0;
  for (int i = lo; i < hi; i++) s += tab[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < MAP_SCAN_NT; d <<= 1) {  // Hillis-Steele over the per-thread totals
    const int v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int i = lo; i < hi; i++) { run += tab[i]; tab[i] = run; }
}

__device__ __forceinline__ int map_rank(const int32_t *tab, long long v) {  // #used values <= v
  if (v < -MAP_SCALE) return 0;
  return tab[(int)(v > MAP_SCALE ? MAP_SCALE : v) + MAP_SCALE];
}
__device__ __forceinline__ int map_used(const int32_t *tab, long long a, long long b) { return a <= b ? map_rank(tab, b) - map_rank(tab, a - 1) : 0; }  // #used values in [a, b]

__global__ void k_map_apply(const MapJob *jobs) {  // Remap::Map + S2U + running max (codec.cpp:175-204)
  const MapJob job = jobs[blockIdx.y];
  int mx = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += gridDim.x * blockDim.x) {
    const int err = job.e[i];
    const int pred = (int)((uint32_t)job.sw[i] - (uint32_t)err);  // codec.cpp:175 (wraps when err did)
    // Map's loop visits pred + sgn * i for i = 1..|err| (map.cpp:141-152). A diverged filter's prediction can be anything in int32
    // (cvt_i32_x86 gives INT32_MIN out of range) and err = val - pred wraps with it, so the walk may run over the end of int32
    // and come back at the other: the reference's `pred + sgn * i` is signed overflow there (undefined; its build wraps, and so
    // does the oracle's, compiled with -fwrapv). Counted as that walk: the part up to the end of int32, then the part from the
    // other end. err == INT32_MIN: `err = -err` stays negative and the loop does not run.
    constexpr long long IMAX = 2147483647ll, IMIN = -2147483648ll;
    int merr = 0;
    if (err > 0) {
      const long long hi = (long long)pred + err;
      merr = map_used(job.tab, (long long)pred + 1, hi < IMAX ? hi : IMAX) + (hi > IMAX ? map_used(job.tab, IMIN, hi - 4294967296ll) : 0);
    } else if (err < 0 && err != (int)IMIN) {
      const long long lo = (long long)pred + err;
      merr = -(map_used(job.tab, lo > IMIN ? lo : IMIN, (long long)pred - 1) + (lo < IMIN ? map_used(job.tab, lo + 4294967296ll, IMAX) : 0));
    }
    const int u = merr < 0 ? 2 * (-merr) : (merr > 0 ? 2 * merr - 1 : 0);  // MathUtils::S2U (utils.h:56-61)
    if (job.m) job.m[i] = u;
    mx = u > mx ? u : mx;
  }
  for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(job.mmax, mx);
}

// ------------------------------------------------------------------------------------------
// Known-answer driver of the exact serial adders (sacopt_serial_sum, tests only -- but it runs the very device helpers the
// LMS kernels call): out[v] = ((0. + x[0]) + x[1]) + ... + x[N-1] for vector v, by
//   N = 1920  spec_exact_sum (15 segments of 128 on mfma_pass32, speculation + bit-exact verification), as k_lms1c
//   N = 384   mfma_chain_lds<6>, as the (256,128) stage's undivided chain
//   N = 48,12 the v_mfma_f64_4x4x4 builtin chain, as the (32,16) and (8,4) stages
// One workgroup of 4 waves per vector.
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_sumkat(const double *__restrict__ x, double *__restrict__ out) {
  constexpr int SEG = L1_SEG, SEGP = L1_SEGP, NSEG = 15;
  __shared__ double prods[N == 1920 ? 16 * SEGP : 2 * (N + 4) + 64];
  __shared__ double sA[16], sX1;
  __shared__ int sJ[16][16];
  const double *xv = x + (size_t)blockIdx.x * N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (int)(sizeof(prods) / 8); i += 256) prods[i] = 0.;
  __syncthreads();
  if constexpr (N == 1920) {
    for (int i = tid; i < N; i += 256) prods[(i / SEG) * SEGP + (i % SEG)] = xv[i];
    __syncthreads();
    if (tid < NSEG) {  // the guess: any order (here pairwise) -- exactness must not depend on it
      double a = 0., b = 0.;
      for (int k = 0; k < SEG; k += 2) { a += prods[tid * SEGP + k]; b += prods[tid * SEGP + k + 1]; }
      sA[tid] = a + b;
    }
    __syncthreads();
    long long rounds = 0;
    const double res = spec_exact_sum<NSEG, SEG / 4, SEGP>((uint32_t)(uintptr_t)&prods[0], sJ, &sX1, sA, wave, wave == 0, 0, rounds, lane);
    if (wave == 0 && lane == SPEC_RESULT_LANE(NSEG)) out[blockIdx.x] = res;
  } else {
    constexpr int T4 = N / 4, TP = N + 4;
    for (int i = tid; i < N; i += 256) { prods[i] = xv[i]; prods[TP + i] = xv[i]; }  // the stages' two-channel operand layout
    __syncthreads();
    if (wave == 0) {
      const double *mp = &prods[((lane & 15) & 1) * TP + (lane >> 4)];
      double acc = 0.;
      if constexpr (T4 % 16 == 0) acc = mfma_chain_lds<T4 / 16>((uint32_t)(uintptr_t)mp);
      else {
        double a[T4];
#pragma unroll
        for (int u = 0; u < T4; u++) a[u] = mp[4 * u];
#pragma unroll
        for (int u = 0; u < T4; u++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], 1.0, acc, 0, 0, 0);
      }
      if (lane == 0) out[blockIdx.x] = acc;
    }
  }
}

#include "unpredict.h"
#include "bitplane.h"

// ------------------------------------------------------------------------------------------
// host-callable launchers (sacopt.cpp is compiled by the host compiler pass of the same TU set)
// ------------------------------------------------------------------------------------------
// the (256,128) stage: threads x taps per thread (768 tap slots of both channels)
#ifndef H2_NT
#define H2_NT 256
#define H2_R 3
#endif
extern "C" int sacopt_lms_state_doubles(int S, int D) {  // 2 * R * NT of the matching k_lms instance
  if (S == 1280 && D == 640) return 2 * 4 * 1024;
  if (S == 256 && D == 128) return 2 * H2_R * H2_NT;
  if (S == 32 && D == 16) return 2 * 2 * 64;
  if (S == 8 && D == 4) return 2 * 1 * 64;
  return 0;
}
// Which kernel a (1280,640) launch of `njobs` chains uses: 1 = k_lms1c (one workgroup per channel), 0 = k_lms<1280,640,...>
// (one workgroup per chain). k_lms1c is faster per chain (7.0k against 10k cycles per sample) but takes two CUs for it
// (14k against 10k CU-cycles): right while the call is bound by the latency of a few chains (the 60 s clip: 15), wrong once
// the optimizer's rounds fill the chip (measured, frames per call: 24: 888 against 860 ms, 30: 1036 against 922, 75: 2130
// against 2020 with the two-channel kernel everywhere).
#ifdef SACOPT_ABLATE
#define ABL_SW(bits) ((bits) >> 8)  // kernel-variant switches of tools/ builds
#else
#define ABL_SW(bits) 0
#endif
extern "C" int sacopt_lms_variant(int S, int D, int njobs, int bits, int limit) {  // limit: most chains that still get two CUs each
  if (S != 1280 || D != 640) return 0;
  return (!(ABL_SW(bits) & (8 | 256)) && (njobs <= limit || (ABL_SW(bits) & 512))) ? 1 : 0;
}
// ct0 / clen: see LMS_SPAN (0, 0: every job carries its own span)
extern "C" int sacopt_launch_lms(int S, int D, const LmsJob *jobs, int njobs, const uint16_t *tab, int bits,
                                 hipStream_t st, int limit, int ct0, int clen) {
  if (njobs <= 0) return 0;
  if (S == 1280 && D == 640) {
    if (sacopt_lms_variant(S, D, njobs, bits, limit)) hipLaunchKernelGGL(k_lms1c, dim3(2 * njobs), dim3(L1_NT), 0, st, jobs, tab, bits, ct0, clen);
    else if (ABL_SW(bits) & 8) hipLaunchKernelGGL((k_lms<1280, 640, 1024, 4, 2048, false>), dim3(njobs), dim3(1024), 0, st, jobs, tab, bits, ct0, clen);
    // 1024 threads are faster for a chain alone (9.4k against 10.0k cycles per sample) but slower in the optimizer's
    // rounds, where the follower stages want room on the same CUs (optimize 571 against 526 ms): A/B switch only
    else if (ABL_SW(bits) & 4096) hipLaunchKernelGGL((k_lms<1280, 640, 1024, 4, 2048, true>), dim3(njobs), dim3(1024), 0, st, jobs, tab, bits, ct0, clen);
    else if (ABL_SW(bits) & 16384) hipLaunchKernelGGL((k_lms<1280, 640, 512, 8, 2048, true, S2_CHP>), dim3(njobs), dim3(512), 0, st, jobs, tab, bits, ct0, clen);
    else hipLaunchKernelGGL(k_lms2, dim3(njobs), dim3(S2_NT), 0, st, jobs, tab, bits, ct0, clen);
  }
  // follower stages: short rings (128 samples per refill) keep their LDS small, so that more of them share a CU and
  // fit beside a lead workgroup (A/B switch 8192: the 512-sample instances)
  else if (ABL_SW(bits) & 8192) {
    if (S == 256 && D == 128) hipLaunchKernelGGL((k_lms<256, 128, 256, 3, 1024, false>), dim3(njobs), dim3(256), 0, st, jobs, tab, bits, ct0, clen);
    else if (S == 32 && D == 16) hipLaunchKernelGGL((k_lms<32, 16, 64, 2, 1024, false>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits, ct0, clen);
    else if (S == 8 && D == 4) hipLaunchKernelGGL((k_lms<8, 4, 64, 1, 1024, false>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits, ct0, clen);
    else return 1;
  }
  // (256,128): the two-halves sum is 136-139 MFMAs per stereo sample for a chain of 56 + look-up instead of 96 dependent steps -- right
  // while a launch is bound by the latency of its chains, wrong when the chip is full and four of these workgroups share a CU's
  // matrix pipes (packed four to a CU the stage costs 2.0k CU-cycles per stereo sample): launches of more than H_PLAIN_MIN chains
  // take the undivided 96-step chain (same-run A/B on the encode call: 75 frames 1248 -> 1223 ms; 15 frames 401 -> 432 ms the other way)
  else if (S == 256 && D == 128 && njobs > H_PLAIN_MIN) hipLaunchKernelGGL((k_lms<256, 128, H2_NT, H2_R, 512, false, 128, false>), dim3(njobs), dim3(H2_NT), 0, st, jobs, tab, bits, ct0, clen);
  else if (S == 256 && D == 128) hipLaunchKernelGGL((k_lms<256, 128, H2_NT, H2_R, 512, false, 128>), dim3(njobs), dim3(H2_NT), 0, st, jobs, tab, bits, ct0, clen);
  else if (S == 32 && D == 16) hipLaunchKernelGGL((k_lms<32, 16, 64, 2, 256, false, 128>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits, ct0, clen);
  else if (S == 8 && D == 4) hipLaunchKernelGGL((k_lms<8, 4, 64, 1, 256, false, 128>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits, ct0, clen);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
// max_n: longest span n - t0 of the jobs
template <int N, int S>
static int launch_lpc3_ns(const LpcJob *jobs, int njobs, int max_n, hipStream_t st) {
  const int max_snap = max_n / 8;
  hipLaunchKernelGGL((k_cov<8, N, S>), dim3(njobs, 2), dim3(LPC_COV_NT(N)), 0, st, jobs);
  if (max_snap) hipLaunchKernelGGL((k_chol<false, N>), dim3((CHOL_PK && N < 32) ? (max_snap + 1) / 2 : max_snap, 2 * njobs), dim3(64), 0, st, jobs);
  hipLaunchKernelGGL((k_ssum<N, S>), dim3((unsigned)(((long long)max_n * N + 255) / 256), 2 * njobs), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL((k_serr<N>), dim3(njobs, 2), dim3(64), 0, st, jobs);
  hipLaunchKernelGGL((k_pm<N>), dim3((max_n + 255) / 256, 2 * njobs), dim3(256), 0, st, jobs);
  return hipGetLastError() != hipSuccess;
}
// ols_n: S + D of LPC3 (24 = the shipped profiles' (16,8); 32 = (24,8))
extern "C" int sacopt_launch_lpc3(const LpcJob *jobs, int njobs, int max_n, int ols_n, hipStream_t st) {
  if (njobs <= 0 || max_n <= 0) return 0;
  if (ols_n == 24) return launch_lpc3_ns<24, 16>(jobs, njobs, max_n, st);
  if (ols_n == 32) return launch_lpc3_ns<32, 24>(jobs, njobs, max_n, st);
  return 1;
}
extern "C" int sacopt_launch_lpc2(const LpcJob *jobs, int njobs, int max_n, hipStream_t st) {
  if (njobs <= 0 || max_n <= 0) return 0;
  const int max_snap = max_n / 32;
  hipLaunchKernelGGL((k_cov<32, LPC_N, LPC_S>), dim3(njobs, 2), dim3(LPC_COV_NT(LPC_N)), 0, st, jobs);
  if (max_snap) hipLaunchKernelGGL((k_chol<true, LPC_N>), dim3(CHOL_PK ? (max_snap + 1) / 2 : max_snap, 2 * njobs), dim3(64), 0, st, jobs);
  hipLaunchKernelGGL(k_pred2, dim3((max_n + 255) / 256, 2 * njobs), dim3(256), 0, st, jobs);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_mix(int ns, const MixJob *jobs, int njobs, hipStream_t st, int ct0, int clen) {
  if (njobs <= 0) return 0;
  const int blocks = (2 * njobs + 3) / 4;
  if (ns == 4) hipLaunchKernelGGL((k_mix<4>), dim3(blocks), dim3(64), 0, st, jobs, njobs, ct0, clen);  // 4 units per wave
  else if (ns == 3) hipLaunchKernelGGL((k_mix<3>), dim3(blocks), dim3(64), 0, st, jobs, njobs, ct0, clen);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_cost(const CostJob *jobs, int njobs, const int *pow2_bits, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(k_cost, dim3(njobs), dim3(64), 0, st, jobs, pow2_bits);
  return hipGetLastError() != hipSuccess;
}
// Buffer-parallel CostGolomb over nbuf (> 0, even) buffers per frame; jobs[].scr / ok must be set.
// Always followed by the sequential kernel, which returns immediately for verified frames.
extern "C" int sacopt_launch_cost_par(const CostJob *jobs, int njobs, int nbuf, const int *pow2_bits, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(k_cost_tail, dim3(nbuf, njobs), dim3(64), 0, st, jobs, pow2_bits);
  hipLaunchKernelGGL(k_cost_par, dim3(nbuf, njobs), dim3(64), 0, st, jobs, pow2_bits);
  hipLaunchKernelGGL(k_cost_fin, dim3((njobs + 63) / 64), dim3(64), 0, st, jobs, njobs);
  hipLaunchKernelGGL(k_cost, dim3(njobs), dim3(64), 0, st, jobs, pow2_bits);
  return hipGetLastError() != hipSuccess;
}
// ---- the entry point that allocates nothing (sacopt_cost_golomb_noalloc) -------------------------------------------------
// Device address of `scratch` on the CURRENT device, looked up once per device (every GPU has its own copy at its own address).
extern "C" int sacopt_noalloc_scratch(int device, void **ptr, size_t *stage_bytes, size_t *ctl_doubles, int *max_n) {
  // per device; contexts may be created from several threads at once: two threads that both find it empty look the same
  // address up and store the same value
  static std::atomic<void *> dev_ptr[64];
  if (device < 0 || device >= 64) return 1;
  void *p = dev_ptr[device].load(std::memory_order_acquire);
  if (!p) {
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) { (void)hipGetLastError(); return 1; }
    dev_ptr[device].store(p, std::memory_order_release);
  }
  *ptr = p;
  *stage_bytes = NOALLOC_STAGE_BYTES;
  *ctl_doubles = NOALLOC_CTL_DOUBLES;
  *max_n = NOALLOC_MAX_N;
  return 0;
}
// e: n residuals on the device (1 <= n <= NOALLOC_MAX_N); ctl: the control block inside the scratch (ctl[0] = state in).
extern "C" int sacopt_launch_cost_noalloc(const int32_t *e, int n, double *ctl, const int *pow2_bits, hipStream_t st) {
  if (n <= 0 || n > NOALLOC_MAX_N) return 1;
  const int nseg = (n + COST_SEG - 1) / COST_SEG;
  if (nseg > 1) hipLaunchKernelGGL(k_cseg_tail, dim3(nseg - 1), dim3(64), 0, st, e, n, ctl, pow2_bits);
  hipLaunchKernelGGL(k_cseg_par, dim3(nseg), dim3(64), 0, st, e, n, nseg, ctl, pow2_bits);
  hipLaunchKernelGGL(k_cseg_fin, dim3(1), dim3(64), 0, st, e, n, nseg, ctl, pow2_bits);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_meanrms_batch(const int32_t *const *bufs, const int32_t *ns, int nbuf, double *out, hipStream_t st) {
  if (nbuf <= 0) return 0;
  hipLaunchKernelGGL(k_meanrms_batch, dim3(nbuf), dim3(256), 0, st, bufs, ns, out);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_gather_i32(const int32_t *const *ptrs, int n, int32_t *out, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_gather_i32, dim3((n + 255) / 256), dim3(256), 0, st, ptrs, n, out);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_meanrms(const int32_t *e, int n, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_meanrms, dim3(1), dim3(256), 0, st, e, n, out);
  return hipGetLastError() != hipSuccess;
}
// Stage-level test entries: one Cholesky snapshot through k_chol; serial sums through the MFMA adder helpers.
extern "C" int sacopt_launch_chol(const LpcJob *job, int n, int fast, hipStream_t st) {
  if (fast && n == 24) hipLaunchKernelGGL((k_chol<true, 24>), dim3(1, 1), dim3(64), 0, st, job);
  else if (!fast && n == 24) hipLaunchKernelGGL((k_chol<false, 24>), dim3(1, 1), dim3(64), 0, st, job);
  else if (!fast && n == 32) hipLaunchKernelGGL((k_chol<false, 32>), dim3(1, 1), dim3(64), 0, st, job);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_sumkat(int nterms, const double *x, int count, double *out, hipStream_t st) {
  if (count <= 0) return 0;
  if (nterms == 1920) hipLaunchKernelGGL((k_sumkat<1920>), dim3(count), dim3(256), 0, st, x, out);
  else if (nterms == 384) hipLaunchKernelGGL((k_sumkat<384>), dim3(count), dim3(256), 0, st, x, out);
  else if (nterms == 48) hipLaunchKernelGGL((k_sumkat<48>), dim3(count), dim3(256), 0, st, x, out);
  else if (nterms == 12) hipLaunchKernelGGL((k_sumkat<12>), dim3(count), dim3(256), 0, st, x, out);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
// payload sizes of the host's bit-plane coder for nbuf residual buffers (bitplane.h). T / msb / sse / lb: scratch.
extern "C" int sacopt_launch_bpsize(const BpJob *jobs, int nbuf, int maxn, int maxplanes, int32_t *T, uint8_t *msb, uint32_t *sse, uint8_t *lb,
                                    const int16_t *fwd, const uint16_t *inv, const uint16_t *sse_init, int tscale, int xscale, int32_t *sizes,
                                    hipStream_t st) {
  if (nbuf <= 0) return 0;
  BpTables tbp = {fwd, inv, sse_init, tscale, xscale};
  // the stage-pipeline form (bitplane.h): up to 16 workgroup rounds of two coders per CU; beyond that the lane-per-coder forms
  // move more decisions per CU
  if (BP_PIPE && nbuf <= BP_PIPE_MAX) {
    if (nbuf <= BPP_ONE_MAX) hipLaunchKernelGGL((k_bppipe<false, 1>), dim3(nbuf), dim3(BPP_STAGES * 64), 0, st, jobs, nbuf, tbp, sizes, (const BpEmit *)nullptr);
    else hipLaunchKernelGGL((k_bppipe<false, 2>), dim3((nbuf + 1) / 2), dim3(2 * BPP_STAGES * 64), 0, st, jobs, nbuf, tbp, sizes, (const BpEmit *)nullptr);
    return hipGetLastError() != hipSuccess;
  }
  const int nblk = (nbuf + 63) / 64;
  if (maxn > 0) hipLaunchKernelGGL(k_bp_transpose, dim3((maxn + 255) / 256, nblk), dim3(256), 0, st, jobs, nbuf, maxn, T);
  BpTables tb = {fwd, inv, sse_init, tscale, xscale};
  // few buffers (an optimizer round, a clip's channels): eight coders per wave, everything a decision touches in LDS (bitplane.h);
  // beyond 4096 buffers the chip could not hold all waves of that form at once and the 64-coder form moves more decisions per CU
  if (nbuf <= BP_WPC_MAX) hipLaunchKernelGGL((k_bpsize<8, true>), dim3((nbuf + 7) / 8), dim3(512), 0, st, jobs, nbuf, maxn, maxplanes, T, msb, sse, lb, tb, sizes);
  else if (nbuf <= BP_LAT_MAX) hipLaunchKernelGGL((k_bpsize<8>), dim3((nbuf + 7) / 8), dim3(64), 0, st, jobs, nbuf, maxn, maxplanes, T, msb, sse, lb, tb, sizes);
  else hipLaunchKernelGGL((k_bpsize<64>), dim3(nblk), dim3(64), 0, st, jobs, nbuf, maxn, maxplanes, T, msb, sse, lb, tb, sizes);
  return hipGetLastError() != hipSuccess;
}
// the payload bytes (sacopt_encode_payloads): the map tables of the mapped buffers first (k_mapenc, which[0..nmapped) = their
// indices), then every buffer's bit planes from the state that left (k_bppipe<true>)
extern "C" int sacopt_launch_payloads(const BpJob *jobs, const BpEmit *emit, int nbuf, const int32_t *which, int nmapped,
                                      const int16_t *fwd, const uint16_t *inv, const uint16_t *sse_init, int tscale, int xscale, int32_t *sizes,
                                      hipStream_t st) {
  if (nbuf <= 0) return 0;
  if (nbuf > BP_PIPE_MAX) return 1;
  BpTables tbp = {fwd, inv, sse_init, tscale, xscale};
  if (nmapped > 0) hipLaunchKernelGGL(k_mapenc, dim3(nmapped), dim3(64), 0, st, emit, which, tbp);
  if (nbuf <= BPP_ONE_MAX) hipLaunchKernelGGL((k_bppipe<true, 1>), dim3(nbuf), dim3(BPP_STAGES * 64), 0, st, jobs, nbuf, tbp, sizes, emit);
  else hipLaunchKernelGGL((k_bppipe<true, 2>), dim3((nbuf + 1) / 2), dim3(2 * BPP_STAGES * 64), 0, st, jobs, nbuf, tbp, sizes, emit);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_map(const MapJob *jobs, int njobs, int max_numsamples, int max_n, hipStream_t st) {
  if (njobs <= 0) return 0;
  int gb = (max_numsamples + 255) / 256; gb = gb < 1 ? 1 : (gb > 256 ? 256 : gb);
  hipLaunchKernelGGL(k_map_mark, dim3(gb, njobs), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(k_map_scan, dim3(njobs), dim3(MAP_SCAN_NT), 0, st, jobs);
  gb = (max_n + 255) / 256; gb = gb < 1 ? 1 : (gb > 256 ? 256 : gb);
  hipLaunchKernelGGL(k_map_apply, dim3(gb, njobs), dim3(256), 0, st, jobs);
  return hipGetLastError() != hipSuccess;
}
