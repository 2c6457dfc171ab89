// Which physical CUs does a hipExtStreamCreateWithCUMask stream use on MI355X?  For a few masks: histogram of (XCC_ID, HW_ID cu/se) over
// the workgroups of a kernel launched on the masked stream, and the time of a fixed amount of work (does the mask restrict at all?).
//   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin && tools/cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

__global__ void k_probe(unsigned* out, int spin) {
  unsigned xcc = __builtin_amdgcn_s_getreg(6164);        // HW_REG_XCC_ID[3:0]
  unsigned hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);   // HW_REG_HW_ID full
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hwid + (a == 12345.f); }
}

int main() {
  const int G = 2048;
  unsigned* d; hipMalloc(&d, G * 8);
  std::vector<unsigned> h(G * 2);
  struct M { const char* name; uint32_t w[8]; };
  M masks[] = {{"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
               {"first 64 bits", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
               {"first 192 bits", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0, 0}},
               {"low 24 bits of every word", {0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu}},
               {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}}};
  for (auto& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { printf("%s: create failed %s\n", m.name, hipGetErrorString(e)); continue; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_probe<<<G, 256, 0, st>>>(d, 100);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    k_probe<<<G * 4, 256, 0, st>>>(d, 20000);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    k_probe<<<G, 256, 0, st>>>(d, 2000);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc; std::map<unsigned, int> cus;
    for (int i = 0; i < G; ++i) { per_xcc[h[i * 2] & 15]++; cus[(h[i * 2] & 15) << 16 | ((h[i * 2 + 1] >> 8) & 0xf) << 4 | ((h[i * 2 + 1] >> 13) & 0x7) << 8 | ((h[i*2+1] >> 12) & 1)]++; }
    printf("%-28s work %.3f ms; distinct (xcc,se,sh,cu) %zu; WGs per XCC:", m.name, ms, cus.size());
    for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    hipStreamDestroy(st);
  }
  return 0;
}
