// Issue rate of a few vector instructions on one GPU: every SIMD runs 4 waves of independent chains.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHAINS 8
#define ITERS 65536

template <int kOp>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, float a, float b) {
  float x[CHAINS];
  f32x2 p[CHAINS];
  double d[CHAINS];
  for (int k = 0; k < CHAINS; ++k) {
    x[k] = threadIdx.x * 1e-3f + k;
    p[k] = f32x2{x[k], x[k] + 1.0f};
    d[k] = x[k];
  }
  f32x2 pa = {a, a}, pb = {b, b};
  double da = a, db = b;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
      if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pa), "v"(pb));
      if (kOp == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(da), "v"(db));
      if (kOp == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[k]) : "v"(pa), "v"(pb));
      if (kOp == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a));
      if (kOp == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(da));
      if (kOp == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(da));
      if (kOp == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 10) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[k]), "v"(a) : "vcc");
      if (kOp == 12) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[k]), "v"(da) : "vcc");
      if (kOp == 13) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
      if (kOp == 14) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
      if (kOp == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[k]) : "v"(a) : "s20", "s21");
      if (kOp == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 18) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 19) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 21) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 22) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(x[k]), "v"(a) : "s20", "s21");
      if (kOp == 23) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 24) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 25) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "+v"(x[k]) : "v"(a));
      if (kOp == 26) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[k]) : "v"(a), "v"(b) : "vcc");
      if (kOp == 27) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 28) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 29) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[k]));
      if (kOp == 30) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[k]) : "v"(a));
      if (kOp == 31) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[k]) : "v"(da));
      if (kOp == 32) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(a));
      if (kOp == 34) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 35) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 36) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      if (kOp == 37) asm volatile("v_mov_b64 %0, %1" : "=v"(d[k]) : "v"(da));
      if (kOp == 40) { if (k & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); }
      if (kOp == 41) { if (k & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[k]) : "v"(pa), "v"(pb)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); }
      if (kOp == 42) { if (k & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a)); }
      if (kOp == 43) { if (k & 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(da), "v"(db)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)); }
      if (kOp == 44) { if (k & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[k]) : "v"(pa), "v"(pb)); else asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(p[k^1].x), "v"(p[k^1].y), "v"(b)); }
      if (kOp == 45) { if (k & 1) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(p[k].x), "+v"(p[k].y) : "v"(a), "v"(b)); else asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(p[k^1].x), "v"(p[k^1].y), "v"(b)); }
      if (kOp == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pa));
      if (kOp == 50) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 51) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 52) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      if (kOp == 53) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[k]) : "v"(a));
      if (kOp == 54) asm volatile("v_perm_b32 %0, %0, %1, s20" : "+v"(x[k]) : "v"(a) : "s20");
    }
  }
  float s = 0;
  for (int k = 0; k < CHAINS; ++k) s += x[k] + p[k].x + p[k].y + (float)d[k];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int kOp>
void run(const char* name, float* out, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rate_kernel<kOp><<<cus, 1024>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e0);
  rate_kernel<kOp><<<cus, 1024>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // wave instructions per SIMD: 4 waves x ITERS x CHAINS
  double per_simd = 4.0 * ITERS * CHAINS;
  printf("%-28s %8.3f ms  -> %.2f ns per wave instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  float* out;
  hipMalloc(&out, sizeof(float) * 1024 * cus);
  run<0>("v_fma_f32", out, cus);
  run<1>("v_pk_fma_f32", out, cus);
  run<4>("v_pk_fma_f32 op_sel_hi", out, cus);
  run<15>("v_pk_mul_f32", out, cus);
  run<2>("v_fma_f64", out, cus);
  run<6>("v_mul_f64", out, cus);
  run<7>("v_add_f64", out, cus);
  run<3>("v_max3_f32", out, cus);
  run<5>("v_cndmask_b32", out, cus);
  run<16>("v_cndmask_b32_e64 sgpr", out, cus);
  run<24>("v_cndmask_b32 dst!=src", out, cus);
  run<25>("v_cndmask_b32 0,v", out, cus);
  run<17>("v_mov_b32", out, cus);
  run<18>("v_bfi_b32", out, cus);
  run<19>("v_min_f32", out, cus);
  run<20>("v_add_u32", out, cus);
  run<21>("v_cmp+v_cndmask (2 instr)", out, cus);
  run<22>("v_cmp_lt_f32_e64 sgpr", out, cus);
  run<23>("v_lshl_add_u32", out, cus);
  run<26>("v_mad_u64_u32", out, cus);
  run<27>("v_xor_b32", out, cus);
  run<28>("v_and_b32", out, cus);
  run<29>("v_lshlrev_b32", out, cus);
  run<36>("v_sub_u32", out, cus);
  run<30>("v_cvt_f64_u32", out, cus);
  run<31>("v_cvt_f32_f64", out, cus);
  run<32>("v_ldexp_f64", out, cus);
  run<34>("v_add_f32", out, cus);
  run<35>("v_mul_f32", out, cus);
  run<37>("v_mov_b64", out, cus);
  run<50>("v_fma_mix_f32 lo half", out, cus);
  run<51>("v_fma_mix_f32 hi half", out, cus);
  run<52>("v_perm_b32", out, cus);
  run<54>("v_perm_b32 sgpr selector", out, cus);
  run<53>("v_cvt_f32_ubyte1", out, cus);
  run<40>("mix fma_f32 / max3", out, cus);
  run<41>("mix pk_fma / max3", out, cus);
  run<42>("mix fma_f32 / add_u32", out, cus);
  run<43>("mix fma_f64 / max3", out, cus);
  run<44>("pk_fma -> dependent max3", out, cus);
  run<45>("2 fma -> dependent max3 (3 instr per 2 slots)", out, cus);
  run<8>("v_mad_u32_u24", out, cus);
  run<9>("v_mul_lo_u32", out, cus);
  run<10>("v_mul_hi_u32", out, cus);
  run<11>("v_cmp_lt_f32", out, cus);
  run<12>("v_cmp_lt_f64", out, cus);
  run<13>("v_rsq_f64", out, cus);
  run<14>("v_rcp_f32", out, cus);
  hipFree(out);
  return 0;
}
