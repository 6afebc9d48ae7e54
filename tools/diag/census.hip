// Residency census (diagnostic, not part of the product): how many workgroups of a given shape does a gfx950 CU really
// hold?  Each workgroup spins ~100 us on s_memrealtime and stamps start / end; the host counts the overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int LDS_BYTES, int VGPRS>
__global__ void census(unsigned long long* out) {
  __shared__ char lds[LDS_BYTES];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (char)t0;
  if (VGPRS > 128) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  else if (VGPRS > 64) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 10000ull) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(lds[5] & 0);
  }
}

template <int L, int V>
void run(int threads) {
  const int blocks = 2048;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks);
  hipLaunchKernelGGL((census<L, V>), dim3(blocks), dim3(threads), 0, 0, d);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int i = 0; i < blocks; i++) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); }
  int best = 0;
  for (int s = 0; s < 400; s++) {
    unsigned long long t = lo + (hi - lo) * s / 400;
    int c = 0;
    for (int i = 0; i < blocks; i++) c += (h[2 * i] <= t && h[2 * i + 1] >= t);
    best = std::max(best, c);
  }
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, census<L, V>, threads, 0);
  printf("LDS %6d B  VGPR>%3d  threads %3d : max concurrent workgroups %4d (%.2f per CU), occupancy API %d\n", L, V, threads, best, best / 256.0, occ);
  hipFree(d);
}

int main() {
  run<32768, 0>(320); run<65536, 0>(320); run<66000, 0>(320); run<71432, 0>(320); run<81416, 0>(320); run<81920, 0>(320);
  run<32768, 168>(320); run<71432, 168>(320); run<81416, 168>(320);
  run<71432, 168>(256); run<81416, 168>(256); run<71432, 128>(320); run<81416, 128>(320);
  run<40000, 168>(320); run<50000, 168>(320); run<60000, 168>(320);
  return 0;
}
