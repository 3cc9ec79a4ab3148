// Micro-benchmark: issue rate of the integer VALU instructions the scan kernel is built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 0x12345, d = 7, e = 9, f = 11, g = 13, h = 17;
  int s = 0;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {  // independent v_and_b32 (8 chains)
      REP16(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_and_b32 %3, %3, %1\n v_and_b32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));)
    } else if (KIND == 1) {  // dependent v_and_b32 chain
      REP16(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1" : "+v"(a), "+v"(b));)
    } else if (KIND == 2) {  // independent v_bitop3
      REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %3, %3, %1, %2 bitop3:0xf1\n v_bitop3_b32 %4, %4, %1, %2 bitop3:0xf1\n v_bitop3_b32 %5, %5, %1, %2 bitop3:0xf1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));)
    } else if (KIND == 3) {  // dependent v_bitop3 chain
      REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1" : "+v"(a), "+v"(b), "+v"(c));)
    } else if (KIND == 4) {  // add_co + addc pairs (VCC dependent)
      REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_add_co_u32 %2, vcc, %2, %2\n v_subb_co_u32 %1, vcc, %1, 0, vcc" : "+v"(a), "+v"(s), "+v"(c) :: "vcc");)
    } else if (KIND == 5) {  // independent v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));)
    } else if (KIND == 6) {  // independent v_min_i32
      REP16(asm volatile("v_min_i32 %0, %0, %1\n v_min_i32 %2, %2, %1\n v_min_i32 %3, %3, %1\n v_min_i32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));)
    } else if (KIND == 7) {  // add_co to SGPR pair (VOP3) + addc
      unsigned long long cb;
      REP16(asm volatile("v_add_co_u32 %0, %3, %0, %0\n v_addc_co_u32 %1, %3, 0, %1, %3\n v_add_co_u32 %2, %3, %2, %2\n v_subb_co_u32 %1, %3, %1, 0, %3" : "+v"(a), "+v"(s), "+v"(c), "=&s"(cb));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + s;
}

template <int KIND>
void run(const char* name, int blocks_per_cu) {
  uint32_t* out;
  int nb = 256 * blocks_per_cu, iters = 2000;
  hipMalloc(&out, nb * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<nb, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<nb, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = (double)iters * 64 * (nb * 4.0) / 1024.0;  // wave-instructions per SIMD
  printf("%-28s blocks/CU=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
         ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
  hipFree(out);
}

int main() {
  for (int bpc : {1, 2, 4}) {
    run<0>("v_and independent", bpc);
    run<1>("v_and dependent", bpc);
    run<2>("v_bitop3 independent", bpc);
    run<3>("v_bitop3 dependent", bpc);
    run<4>("add_co/addc vcc", bpc);
    run<7>("add_co/addc sgpr", bpc);
    run<5>("v_add_u32 independent", bpc);
    run<6>("v_min_i32 independent", bpc);
  }
  return 0;
}
