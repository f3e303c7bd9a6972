// Issue cost of the integer instructions a multi-limb multiply is built from, on every CU
// at 1, 2 and 4 waves per SIMD: cycles of the shader clock per wave-instruction per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o instr_rate instr_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int ID>
__global__ __launch_bounds__(1024) void k_rate(int iters, u64* cycles, u32* sink) {
  u64 acc[16];
  u32 c[16];
  u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = (u64)i * 0x9E3779B97F4A7C15ull + a; c[i] = i * 77u + b; }
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {  // 256 instructions (or pairs) per trip: the loop's own branch is below 2 % of a trip
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (ID == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (ID == 1) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(c[i]) : : "vcc");
        if (ID == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(c[i]) : "v"(c[(i + 1) & 15]));
        if (ID == 4) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 5) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 6) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 7) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        if (ID == 8) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        if (ID == 9) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
        if (ID == 10) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(c[i]) : "v"(a));
        if (ID == 11) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(c[i]) : "v"(a), "v"(b) : "vcc");
        if (ID == 12) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(c[i]), "+v"(a) : "v"(b) : "vcc");
        if (ID == 13) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mov_b32 %1, %2" : "+v"(acc[i]), "=v"(c[i]) : "v"(a), "v"(b) : "vcc");
        if (ID == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, s12, %0" : "+v"(acc[i]) : "v"(a) : "vcc", "s12");
        if (ID == 15) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s10", "s11");
        if (ID == 16) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
        if (ID == 17) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(a) : "vcc");
        if (ID == 19) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_mov_b32 %4, %2" : "+v"(acc[i]), "+v"(c[i]) : "v"(a), "v"(b), "v"(c[(i + 1) & 15]) : "vcc");
        if (ID == 20) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (ID == 21) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_subbrev_co_u32 %2, vcc, 0, %2, vcc" : "+v"(c[i]), "+v"(a) : "v"(b) : "vcc");
        if (ID == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[14:15]" : "+v"(c[i]) : "v"(a) : "s14", "s15");
        if (ID == 25) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        if (ID == 26) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(a) : "vcc");
        if (ID == 27) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 28) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        if (ID == 30) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]), "v"(acc[(i + 2) & 15]));
        if (ID == 31) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
        if (ID == 32) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
        if (ID == 33) asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %3, %1, %2, %3" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(acc[(i + 1) & 15]), "v"(acc[(i + 2) & 15]));
        if (ID == 22) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(c[i]) : "v"(a));
        if (ID == 23) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
      }
    }
  }
  u64 t1 = __builtin_readcyclecounter();
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) x ^= acc[i] ^ c[i];
  if (x == 0x123456789abcdefull) sink[0] = (u32)x;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// iterations per launch (argv[1]; 2000 = launches of 0.3-2 ms, 200000 = launches of 30-200 ms, long enough
// for the clock the GPU holds under sustained load) and whether only the multiply tests run (argv[2])
static int g_iters = 2000;
static bool g_quick = false;

template <int ID>
void run(const char* name, int per_iter_instrs) {
  if (g_quick && ID != 0 && ID != 1 && ID != 11 && ID != 2) return;
  u64* d_cyc; u32* d_sink;
  hipMalloc(&d_cyc, 4096 * 8); hipMalloc(&d_sink, 64);
  const int iters = g_iters;
  for (int waves = 1; waves <= 8; waves *= 2) {
    // one block per CU with `waves` waves on each SIMD; eight waves per SIMD as two blocks per CU
    int threads = 256 * (waves > 4 ? 4 : waves), blocks = 256 * (waves > 4 ? waves / 4 : 1);
    hipLaunchKernelGGL(k_rate<ID>, dim3(blocks), dim3(threads), 0, 0, 10, d_cyc, d_sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<ID>, dim3(blocks), dim3(threads), 0, 0, iters, d_cyc, d_sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 h[512]; hipMemcpy(h, d_cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += (double)h[i]; avg /= 256;
    double n_instr = (double)iters * 256 * per_iter_instrs;  // wave-instructions per wave
    // cycles per wave-instruction per SIMD = block cycles / (instructions per wave * waves per SIMD)
    printf("%-34s waves/SIMD %d: %6.2f cyc/instr/SIMD (counter)  %7.3f ms  -> %.2f Ginstr/s/SIMD-lane-group\n", name, waves,
           avg / (n_instr * (waves > 4 ? 4 : waves)), ms, n_instr * waves * 1024 / (ms * 1e-3) / 1e9);
  }
  hipFree(d_cyc); hipFree(d_sink);
}

int main(int argc, char** argv) {
  if (argc > 1) g_iters = atoi(argv[1]);
  if (argc > 2) g_quick = atoi(argv[2]) != 0;
  if (g_iters < 1) g_iters = 2000;
  run<0>("v_mad_u64_u32", 1);
  run<14>("v_mad_u64_u32 const", 1);
  run<15>("v_mad_u64_u32 sgpr carry", 1);
  run<20>("v_mad_i64_i32", 1);
  run<1>("v_addc_co_u32", 1);
  run<2>("v_add_u32", 1);
  run<3>("v_mov_b32", 1);
  run<4>("v_mul_lo_u32", 1);
  run<5>("v_mul_hi_u32", 1);
  run<6>("v_mul_u32_u24", 1);
  run<7>("v_mad_u32_u24", 1);
  run<23>("v_mad_u32_u16", 1);
  run<22>("v_pk_mul_lo_u16", 1);
  run<8>("v_add3_u32", 1);
  run<9>("v_lshl_add_u64", 1);
  run<16>("v_lshrrev_b64", 1);
  run<10>("v_alignbit_b32", 1);
  run<17>("v_and_b32", 1);
  run<18>("v_cndmask_b32", 1);
  run<24>("v_cndmask_b32_e64 sgpr", 1);
  run<25>("v_bfi_b32", 1);
  run<26>("v_cmp + v_cndmask", 2);
  run<27>("v_xor_b32", 1);
  run<28>("v_and_or_b32", 1);
  run<11>("mad + addc", 2);
  run<19>("mad + addc + mov", 3);
  run<13>("mad + mov", 2);
  run<12>("add_co + addc", 2);
  run<21>("sub_co + subb", 2);
  run<30>("v_fma_f64", 1);
  run<33>("v_fma_f64 x2 same operands", 2);
  run<31>("v_add_f64", 1);
  run<32>("v_mul_f64", 1);
  return 0;
}
