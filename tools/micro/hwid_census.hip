// hwid_census.hip -- where do the workgroups of a grid land?  Every workgroup records HW_REG_HW_ID of its four waves, the XCC id
// and a start timestamp; the host prints which blockIdx values share a CU and which wave slots / thread-group ids they got.
// Evidence for the start-offset ("stagger") logic of csrc/conv2d_bx3.h.   hipcc --offload-arch=gfx950 -O2 -o hwid_census hwid_census.hip
//   usage: hwid_census [blocks=512] [lds_kb=70] [work_iters=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) census(unsigned* out, long long* t0, int iters) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wave) * 2 + 0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID, 32 bits
    out[(blockIdx.x * 4 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // XCC_ID
    if (wave == 0) t0[blockIdx.x] = (long long)__builtin_readcyclecounter();
  }
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;      // keep the block resident for a while
  lds[threadIdx.x] = v;
  __syncthreads();
  if (v == 12345.f) out[0] = (unsigned)lds[(threadIdx.x + 1) & 255];
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512, lds_kb = argc > 2 ? atoi(argv[2]) : 70, iters = argc > 3 ? atoi(argv[3]) : 20000;
  unsigned* d; long long* dt;
  hipMalloc(&d, blocks * 8 * sizeof(unsigned)); hipMalloc(&dt, blocks * sizeof(long long));
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census, dim3(blocks), dim3(256), lds_kb * 1024, 0, d, dt, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(blocks * 8); std::vector<long long> ht(blocks);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, ht.size() * 8, hipMemcpyDeviceToHost);
  long long tmin = *std::min_element(ht.begin(), ht.end());
  std::map<unsigned, std::vector<int>> by_cu;
  printf("# block xcc se sh cu | wave_id simd tg_id (per wave) | start cycles\n");
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw0 = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 15;
    const unsigned cu = (hw0 >> 8) & 15, sh = (hw0 >> 12) & 1, se = (hw0 >> 13) & 7;
    by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    if (b < 40 || (b >= 250 && b < 270) || b >= blocks - 8) {
      printf("%4d  %u %u %u %2u |", b, xcc, se, sh, cu);
      for (int w = 0; w < 4; ++w) { const unsigned hw = h[(b * 4 + w) * 2]; printf(" w%u s%u t%u", hw & 15, (hw >> 4) & 3, (hw >> 16) & 15); }
      printf(" | %lld  raw %08x\n", ht[b] - tmin, hw0);
    }
  }
  printf("# distinct (xcc,se,sh,cu): %zu\n", by_cu.size());
  int shown = 0;
  std::map<size_t, int> hist;
  for (auto& kv : by_cu) {
    hist[kv.second.size()]++;
    if (shown++ < 24) {
      printf("cu %06x:", kv.first);
      for (int b : kv.second) printf(" %d(w%u t%u @%lld)", b, h[(b * 4) * 2] & 15, (h[(b * 4) * 2] >> 16) & 15, ht[b] - tmin);
      printf("\n");
    }
  }
  for (auto& kv : hist) printf("# %d CUs hold %zu blocks\n", kv.second, kv.first);
  return 0;
}
