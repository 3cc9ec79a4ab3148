// Micro-benchmark: how fast a kernel writes into page-locked host memory over the bus, by store width (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/host_write_bench.hip -o /tmp/host_write_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void wr(T* dst, size_t n, T v) {   // consecutive lanes write consecutive elements
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

// what bin_rows_kernel does: a wave writes a "row" of len bytes, 64 bytes per store instruction, rows at arbitrary offsets
__global__ void rows_bytes(char* dst, size_t n_rows, int len) {
  const int lane = threadIdx.x & 63;
  for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    char* d = dst + r * (size_t)len;
    for (int x = lane; x < len; x += 64) d[x] = (char)x;
  }
}
// the same rows with 16-byte stores for the aligned middle and byte stores for the ragged ends
__global__ void rows_wide(char* dst, size_t n_rows, int len) {
  const int lane = threadIdx.x & 63;
  for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    char* d = dst + r * (size_t)len;
    const uintptr_t a = (uintptr_t)d;
    const int head = (int)((16 - (a & 15)) & 15), body = (len - head) / 16, tail = len - head - body * 16;
    if (lane < head) d[lane] = (char)lane;
    uint4* w = reinterpret_cast<uint4*>(d + head);
    for (int x = lane; x < body; x += 64) w[x] = make_uint4(x, x, x, x);
    if (lane < tail) d[head + body * 16 + lane] = (char)lane;
  }
}

template <typename F>
void timeit(const char* name, size_t bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.1f us  %6.1f GB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = 3300000;                        // an eighth-of-the-genome lane's text
  char* h; hipHostMalloc((void**)&h, bytes + 4096, hipHostMallocDefault);
  char* d; hipMalloc((void**)&d, bytes + 4096);
  for (int grid : {64, 256, 1024, 4096}) {
    char nm[96];
    snprintf(nm, sizeof nm, "host  u8   grid %d", grid);  timeit(nm, bytes, [&] { wr<uint8_t><<<grid, 256>>>((uint8_t*)h, bytes, 1); });
    snprintf(nm, sizeof nm, "host  u32  grid %d", grid);  timeit(nm, bytes, [&] { wr<uint32_t><<<grid, 256>>>((uint32_t*)h, bytes / 4, 1u); });
    snprintf(nm, sizeof nm, "host  u128 grid %d", grid);  timeit(nm, bytes, [&] { wr<uint4><<<grid, 256>>>((uint4*)h, bytes / 16, make_uint4(1, 2, 3, 4)); });
    snprintf(nm, sizeof nm, "host  rows of 523 bytes, byte stores, grid %d", grid); timeit(nm, bytes, [&] { rows_bytes<<<grid, 64>>>(h, bytes / 523, 523); });
    snprintf(nm, sizeof nm, "host  rows of 523 bytes, 16-byte stores, grid %d", grid); timeit(nm, bytes, [&] { rows_wide<<<grid, 64>>>(h, bytes / 523, 523); });
  }
  timeit("device rows of 523 bytes, byte stores, grid 4096", bytes, [&] { rows_bytes<<<4096, 64>>>(d, bytes / 523, 523); });
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
  hipEventRecord(e0); hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.1f us  %6.1f GB/s\n", "hipMemcpyAsync device -> host", ms * 1e3, bytes / (ms * 1e-3) / 1e9);
  return 0;
}
