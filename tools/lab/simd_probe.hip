// Where do the waves of a workgroup land?  Every wave of a launch records its HW_ID (wave slot, SIMD, CU, SE), XCC_ID and LDS_ALLOC and
// then stays resident for ~40 us, so that the workgroups pile up on the CUs the way the BiGRU scans' do.  The question behind it
// (round 5): the fused GruBlock forward keeps ONE scanning wave per workgroup alive (wave 0) -- are the scanning waves of the workgroups
// that share a CU on the SAME SIMD (then they share one issue port and the other three SIMDs idle)?
//   hipcc --offload-arch=gfx950 -O2 tools/lab/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <array>
#include <algorithm>

#define GETREG(id) __builtin_amdgcn_s_getreg((id) | (31 << 11))

__global__ void probe_kernel(unsigned* out, int hold_us) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (threadIdx.x == 0) lds[0] = 1.f;
  const unsigned hw = GETREG(4), xcc = GETREG(20), la = GETREG(6);
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + ((size_t)blockIdx.x * nw + wave) * 4;
    o[0] = hw; o[1] = xcc; o[2] = la; o[3] = (unsigned)wall_clock64();
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)hold_us * 100ull) __builtin_amdgcn_s_sleep(8);
}

static void run(const char* name, int grid, int threads, size_t lds, int hold_us) {
  const int nw = threads / 64;
  unsigned* d;
  hipMalloc(&d, (size_t)grid * nw * 16);
  hipMemset(d, 0, (size_t)grid * nw * 16);
  hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(threads), lds, 0, d, hold_us);
  hipDeviceSynchronize();
  std::vector<unsigned> h((size_t)grid * nw * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  hipFree(d);
  // per CU (xcc, se, sh, cu): which SIMD does wave w of each resident workgroup sit on?
  std::map<unsigned, std::vector<std::array<int, 3>>> cus;      // key -> (block, wave, simd)
  int hist[8][4] = {};                                          // wave index in the workgroup -> SIMD histogram
  for (int b = 0; b < grid; ++b)
    for (int w = 0; w < nw; ++w) {
      const unsigned hw = h[((size_t)b * nw + w) * 4], xcc = h[((size_t)b * nw + w) * 4 + 1] & 15;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      if (w < 8) hist[w][simd]++;
      cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({b, w, simd});
    }
  printf("== %s: grid %d x %d threads, %zu B LDS: %zu distinct CUs\n", name, grid, threads, lds, cus.size());
  for (int w = 0; w < nw && w < 8; ++w) printf("   wave %d of a workgroup -> SIMD 0/1/2/3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // wave 0 of the workgroups sharing a CU: how many distinct SIMDs?
  int same = 0, total = 0, shown = 0;
  for (auto& kv : cus) {
    int simds[4] = {}, n0 = 0;
    for (auto& e : kv.second)
      if (e[1] == 0) { simds[e[2]]++; n0++; }
    if (n0 < 2) continue;
    total++;
    const int mx = std::max(std::max(simds[0], simds[1]), std::max(simds[2], simds[3]));
    if (mx == n0) same++;
    if (shown < 6) {
      printf("   CU %05x: wave-0 of %d workgroups on SIMDs [%d %d %d %d]; blocks:", kv.first, n0, simds[0], simds[1], simds[2], simds[3]);
      for (auto& e : kv.second)
        if (e[1] == 0) printf(" %d", e[0]);
      printf("\n");
      shown++;
    }
  }
  printf("   CUs with >= 2 workgroups: %d, of which ALL wave-0s on one SIMD: %d\n", total, same);
}

int main() {
  run("fused GruBlock forward, W axis (T 64)", 768, 256, 50688, 40);
  run("fused GruBlock forward, H axis (T 16)", 3072, 256, 13056, 40);
  run("one-wave workgroups (bigru_bwd), 768", 768, 64, 0, 40);
  run("one-wave workgroups (bigru_bwd), 3072", 3072, 64, 0, 40);
  return 0;
}
