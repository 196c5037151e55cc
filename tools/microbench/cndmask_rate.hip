// What a select costs by where its lane mask comes from: every SIMD runs 4 waves of 8 independent chains.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/cndmask_rate.hip -o /tmp/cndmask_rate && /tmp/cndmask_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define ITERS 16384

template <int kOp>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, float a, float b) {
  float x[CHAINS], y[CHAINS];
  for (int k = 0; k < CHAINS; ++k) {
    x[k] = threadIdx.x * 1e-3f + k;
    y[k] = x[k] + 0.5f;
  }
  if (kOp == 1) asm volatile("s_mov_b64 vcc, -1" ::: "vcc");
  if (kOp == 2) asm volatile("s_mov_b64 vcc, 0" ::: "vcc");
  if (kOp == 3) asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
  if (kOp == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x[0]), "v"(a) : "vcc");
  if (kOp == 6 || kOp == 7 || kOp == 8 || kOp == 13 || kOp == 16 || kOp == 17) asm volatile("s_mov_b64 s[20:21], 0x55555555\n\ts_mov_b64 s[22:23], 0x33333333" ::: "s20", "s21", "s22", "s23");
  for (int it = 0; it < ITERS; ++it) {
    if (kOp == 6) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]" ::: "vcc", "scc");  // the mask made on the scalar side once per 8 selects
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
      if (kOp <= 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : );
      if (kOp == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a));
      if (kOp == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a));
      if (kOp == 7) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc", "scc");
      if (kOp == 8) asm volatile("s_and_b64 s[24:25], s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(x[k]) : "v"(a) : "s24", "s25", "scc");
      if (kOp == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 10) asm volatile("v_cmp_lt_f32_e64 s[24:25], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(x[k]) : "v"(a) : "s24", "s25");
      if (kOp == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 13) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc", "scc");
      if (kOp == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, s[22:23]\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc", "scc");
      if (kOp == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 s[24:25], vcc, s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(x[k]) : "v"(a) : "vcc", "scc", "s24", "s25");
      if (kOp == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");
      if (kOp == 19) asm volatile("v_cmp_lt_f64 vcc, %2, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a), "v"((double)b) : "vcc");
      if (kOp == 20) asm volatile("v_cmp_lt_f64 vcc, %3, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a), "v"((double)b) : "vcc");
      if (kOp == 21) asm volatile("v_cmp_lt_f64_e64 s[24:25], %3, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, s[24:25]\n\tv_cndmask_b32_e64 %1, %1, %2, s[24:25]" : "+v"(x[k]), "+v"(y[k]) : "v"(a), "v"((double)b) : "s24", "s25");
      if (kOp == 22) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_add_u32 %1, %1, %2\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 23) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, s[22:23]" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 24) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 25) asm volatile("v_cmp_lt_f64 vcc, %3, %3\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a), "v"((double)b) : "vcc");
      if (kOp == 26) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 27) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2\n\tv_fma_f32 %0, %0, %2, %2\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 29) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_fma_f32 %0, %0, %2, %2\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(y[k]) : "v"(a) : "vcc");
      if (kOp == 12) asm volatile("v_cmp_lt_f32_e64 s[24:25], %0, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(x[k]) : "v"(a) : "s24", "s25");
    }
  }
  float s = 0;
  for (int k = 0; k < CHAINS; ++k) s += x[k] + y[k];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int kOp>
void run(const char* name, float* out, int cus, int instr_per_slot) {
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
  double per_simd = 4.0 * ITERS * CHAINS;
  printf("%-72s %8.3f ms -> %6.2f cycles per slot of %d instruction(s) at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4, instr_per_slot);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, sizeof(float) * 1024 * cus);
  run<0>("v_cndmask_b32 (vcc as the kernel found it)", out, cus, 1);
  run<1>("v_cndmask_b32, vcc = -1 by s_mov before the loop", out, cus, 1);
  run<2>("v_cndmask_b32, vcc = 0", out, cus, 1);
  run<3>("v_cndmask_b32, vcc = 0x55555555", out, cus, 1);
  run<4>("v_cndmask_b32, vcc from one v_cmp before the loop", out, cus, 1);
  run<5>("v_cndmask_b32_e64 with vcc", out, cus, 1);
  run<6>("s_and_b64 vcc once per 8 x v_cndmask_b32 vcc", out, cus, 1);
  run<7>("s_and_b64 vcc; v_cndmask_b32 vcc", out, cus, 2);
  run<8>("s_and_b64 s[24:25]; v_cndmask_b32_e64 s[24:25]", out, cus, 2);
  run<9>("v_cmp vcc; v_cndmask_b32 vcc", out, cus, 2);
  run<10>("v_cmp_e64 s[24:25]; v_cndmask_b32_e64 s[24:25]", out, cus, 2);
  run<11>("v_cmp vcc; v_fma; v_cndmask_b32 vcc", out, cus, 3);
  run<12>("v_cmp_e64 s[24:25]; v_fma; v_cndmask_b32_e64 s[24:25]", out, cus, 3);
  run<13>("s_and_b64 vcc; v_cndmask_b32_e64 vcc", out, cus, 2);
  run<14>("v_cmp vcc; 4 x v_cndmask_b32 vcc", out, cus, 5);
  run<15>("v_cmp vcc; 8 x v_fma; v_cndmask_b32 vcc", out, cus, 10);
  run<16>("v_cmp vcc; s_and_b64 vcc, vcc, s; v_cndmask_b32 vcc", out, cus, 3);
  run<17>("v_cmp vcc; s_and_b64 s[24:25], vcc, s; v_cndmask_b32_e64 s[24:25]", out, cus, 3);
  run<18>("v_cmp vcc; v_cndmask vcc; 2 x v_fma; v_cndmask vcc", out, cus, 5);
  run<19>("v_cmp_f64 vcc; s_nop 1; 2 x v_cndmask vcc", out, cus, 3);
  run<20>("v_cmp_f64 vcc; s_nop 1; v_cndmask vcc; v_cndmask vcc (two registers)", out, cus, 3);
  run<21>("v_cmp_f64_e64 s; s_nop 1; 2 x v_cndmask_e64 s", out, cus, 3);
  run<22>("v_cmp vcc; v_cndmask vcc; v_add_u32; v_cndmask vcc", out, cus, 4);
  run<23>("v_cmp vcc; v_cndmask vcc; v_cndmask_e64 s", out, cus, 3);
  run<24>("v_cmp vcc; s_nop 1; v_cndmask vcc", out, cus, 2);
  run<25>("v_cmp_f64 vcc; v_cndmask vcc", out, cus, 2);
  run<26>("v_cmp vcc; 2 x v_cndmask vcc (two registers)", out, cus, 3);
  run<27>("v_cmp vcc; 2 x v_cndmask_e64 vcc (two registers)", out, cus, 3);
  run<28>("v_cmp vcc; 3 x v_fma; v_cndmask vcc", out, cus, 5);
  run<29>("v_cmp vcc; v_cndmask vcc; v_fma; v_cndmask vcc", out, cus, 4);
  return 0;
}
