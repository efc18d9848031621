// does gfx950 execute scalar atomics?  every wave draws a ticket with s_atomic_add; the tickets must be a permutation of 0..waves-1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void k(int* c, int* o) {
  int r;
  asm volatile("s_mov_b32 %0, 1\n s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(c) : "memory");
  if ((threadIdx.x & 63) == 0) o[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
}
int main() {
  int *c, *o; const int blocks = 1024, waves = blocks * 4;
  hipMalloc(&c, 4); hipMemset(c, 0, 4); hipMalloc(&o, waves * 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, c, o);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<int> h(waves); int hc;
  hipMemcpy(h.data(), o, waves * 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = hc == waves;
  for (int i = 0; i < waves; ++i) ok &= h[i] == i;
  printf("counter %d (expected %d), tickets %s\n", hc, waves, ok ? "a permutation: scalar atomics work" : "NOT a permutation");
  return ok ? 0 : 1;
}
