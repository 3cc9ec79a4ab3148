// Micro-benchmark 2: does the VGPR bank (index mod 4) of the source operands set the issue rate of 3-source VALU ops?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t acc = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {  // bitop3, sources in banks 0,1,2 ; four independent dests
      REP16(asm volatile("v_bitop3_b32 v20, v24, v25, v26 bitop3:0xf1\n v_bitop3_b32 v21, v28, v29, v30 bitop3:0xf1\n v_bitop3_b32 v22, v32, v33, v34 bitop3:0xf1\n v_bitop3_b32 v23, v24, v29, v34 bitop3:0xf1" ::: CLOB);)
    } else if (KIND == 1) {  // bitop3, all three sources in bank 0
      REP16(asm volatile("v_bitop3_b32 v21, v24, v28, v32 bitop3:0xf1\n v_bitop3_b32 v22, v24, v28, v32 bitop3:0xf1\n v_bitop3_b32 v23, v24, v28, v32 bitop3:0xf1\n v_bitop3_b32 v25, v24, v28, v32 bitop3:0xf1" ::: CLOB);)
    } else if (KIND == 2) {  // bitop3, two sources share a bank
      REP16(asm volatile("v_bitop3_b32 v21, v24, v28, v33 bitop3:0xf1\n v_bitop3_b32 v22, v24, v28, v33 bitop3:0xf1\n v_bitop3_b32 v23, v24, v28, v33 bitop3:0xf1\n v_bitop3_b32 v25, v24, v28, v33 bitop3:0xf1" ::: CLOB);)
    } else if (KIND == 3) {  // v_min_i32 sources in different banks
      REP16(asm volatile("v_min_i32 v20, v24, v25\n v_min_i32 v21, v26, v27\n v_min_i32 v22, v28, v29\n v_min_i32 v23, v30, v31" ::: CLOB);)
    } else if (KIND == 4) {  // v_min_i32 sources in the same bank
      REP16(asm volatile("v_min_i32 v20, v24, v28\n v_min_i32 v21, v24, v28\n v_min_i32 v22, v24, v28\n v_min_i32 v23, v24, v28" ::: CLOB);)
    } else if (KIND == 5) {  // v_and sources same bank
      REP16(asm volatile("v_and_b32 v20, v24, v28\n v_and_b32 v21, v24, v28\n v_and_b32 v22, v24, v28\n v_and_b32 v23, v24, v28" ::: CLOB);)
    } else if (KIND == 6) {  // v_max_i32 / v_or3
      REP16(asm volatile("v_or3_b32 v20, v24, v25, v26\n v_or3_b32 v21, v28, v29, v30\n v_or3_b32 v22, v32, v33, v34\n v_or3_b32 v23, v24, v29, v34" ::: CLOB);)
    } else if (KIND == 7) {  // v_add3_u32 banks 0,1,2
      REP16(asm volatile("v_add3_u32 v20, v24, v25, v26\n v_add3_u32 v21, v28, v29, v30\n v_add3_u32 v22, v32, v33, v34\n v_add3_u32 v23, v24, v29, v34" ::: CLOB);)
    } else if (KIND == 8) {  // v_alignbit with constant shift
      REP16(asm volatile("v_alignbit_b32 v20, v24, v25, 31\n v_alignbit_b32 v21, v28, v29, 31\n v_alignbit_b32 v22, v32, v33, 31\n v_alignbit_b32 v23, v24, v29, 31" ::: CLOB);)
    } else if (KIND == 9) {  // v_lshl_add_u32
      REP16(asm volatile("v_lshl_add_u32 v20, v24, 1, v25\n v_lshl_add_u32 v21, v28, 1, v29\n v_lshl_add_u32 v22, v32, 1, v33\n v_lshl_add_u32 v23, v24, 1, v29" ::: CLOB);)
    } else if (KIND == 10) { // v_add_co_u32 VOP2 (vcc) alone
      REP16(asm volatile("v_add_co_u32 v20, vcc, v24, v25\n v_add_co_u32 v21, vcc, v26, v27\n v_add_co_u32 v22, vcc, v28, v29\n v_add_co_u32 v23, vcc, v30, v31" ::: CLOB, "vcc");)
    } else if (KIND == 11) { // v_cmp + nothing
      REP16(asm volatile("v_cmp_lt_i32 vcc, v24, v25\n v_cmp_lt_i32 vcc, v26, v27\n v_cmp_lt_i32 vcc, v28, v29\n v_cmp_lt_i32 vcc, v30, v31" ::: CLOB, "vcc");)
    } else if (KIND == 12) { // v_lshrrev / v_ashrrev
      REP16(asm volatile("v_lshrrev_b32 v20, 31, v24\n v_ashrrev_i32 v21, 31, v25\n v_lshlrev_b32 v22, 1, v26\n v_lshlrev_b32 v23, 1, v27" ::: CLOB);)
    } else if (KIND == 13) { // v_sub_u32 / v_subrev
      REP16(asm volatile("v_sub_u32 v20, v24, v25\n v_sub_u32 v21, v26, v27\n v_sub_u32 v22, v28, v29\n v_sub_u32 v23, v30, v31" ::: CLOB);)
    } else if (KIND == 14) { // v_xad_u32
      REP16(asm volatile("v_xad_u32 v20, v24, v25, v26\n v_xad_u32 v21, v28, v29, v30\n v_xad_u32 v22, v32, v33, v34\n v_xad_u32 v23, v24, v29, v34" ::: CLOB);)
    } else if (KIND == 15) { // v_and_or_b32
      REP16(asm volatile("v_and_or_b32 v20, v24, v25, v26\n v_and_or_b32 v21, v28, v29, v30\n v_and_or_b32 v22, v32, v33, v34\n v_and_or_b32 v23, v24, v29, v34" ::: CLOB);)
    } else if (KIND == 16) { // v_bfe_u32 with constants
      REP16(asm volatile("v_bfe_u32 v20, v24, 4, 4\n v_bfe_u32 v21, v25, 8, 4\n v_bfe_u32 v22, v26, 12, 4\n v_bfe_u32 v23, v27, 16, 4" ::: CLOB);)
    } else if (KIND == 17) { // v_max_i32
      REP16(asm volatile("v_max_i32 v20, v24, v25\n v_max_i32 v21, v26, v27\n v_max_i32 v22, v28, v29\n v_max_i32 v23, v30, v31" ::: CLOB);)
    } else if (KIND == 18) { // v_min_u32
      REP16(asm volatile("v_min_u32 v20, v24, v25\n v_min_u32 v21, v26, v27\n v_min_u32 v22, v28, v29\n v_min_u32 v23, v30, v31" ::: CLOB);)
    } else if (KIND == 19) { // v_min3_i32
      REP16(asm volatile("v_min3_i32 v20, v24, v25, v26\n v_min3_i32 v21, v28, v29, v30\n v_min3_i32 v22, v32, v33, v34\n v_min3_i32 v23, v24, v29, v34" ::: CLOB);)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name) {
  uint32_t* out; int nb = 256 * 4, iters = 2000;
  (void)hipMalloc(&out, nb * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND><<<nb, 256>>>(out, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<KIND><<<nb, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = (double)iters * 64 * (nb * 4.0) / 1024.0;
  printf("%-40s %.3f ms  %.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / insts_per_simd);
  (void)hipFree(out);
}
int main() {
  run<5>("v_and same bank");
  run<0>("bitop3 srcs banks 0,1,2");
  run<1>("bitop3 srcs all bank 0");
  run<2>("bitop3 two srcs share bank");
  run<3>("v_min_i32 different banks");
  run<4>("v_min_i32 same bank");
  run<17>("v_max_i32");
  run<18>("v_min_u32");
  run<19>("v_min3_i32");
  run<6>("v_or3 banks 0,1,2");
  run<7>("v_add3 banks 0,1,2");
  run<8>("v_alignbit const shift");
  run<9>("v_lshl_add_u32");
  run<10>("v_add_co_u32 (vcc) alone");
  run<11>("v_cmp_lt_i32");
  run<12>("shifts");
  run<13>("v_sub_u32");
  run<14>("v_xad_u32");
  run<15>("v_and_or_b32");
  run<16>("v_bfe_u32");
  return 0;
}
