// Do two workgroups of k_lms1c's shape (1024 threads, ~60 KB LDS) share a CU? Each workgroup records its
// HW_ID / XCC_ID and the wall-clock interval it was resident; the host counts overlapping intervals per CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/occ_probe tools/occ_probe.hip ; run: /tmp/occ_probe [ngroups] [lds_kb] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };
// VG: highest VGPR the kernel is made to allocate (a clobber), so that the probe has the register footprint of the kernel in question
template <int VG>
__global__ void probe(Rec *out, int lds_bytes, unsigned long long ticks) {
  extern __shared__ double lds[];
  if (lds_bytes == -1) lds[threadIdx.x] = 0.;
  if (lds_bytes == -2) {  // (never true: forces a private segment -- scratch -- when ticks' top bit asks for it at compile time)
    volatile int spill[16];
    for (int i = 0; i < 16; i++) spill[i] = i * (int)ticks;
    lds[threadIdx.x] = spill[threadIdx.x & 15];
  }
  if (VG == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VG == 168) asm volatile("v_mov_b32 v168, 0" ::: "v168");
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    Rec r;
    r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    r.xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    r.t0 = t0; r.t1 = wall_clock64();
    out[blockIdx.x] = r;
  }
}
int main(int argc, char **argv) {
  const int ng = argc > 1 ? atoi(argv[1]) : 512, kb = argc > 2 ? atoi(argv[2]) : 60, nt = argc > 3 ? atoi(argv[3]) : 1024, vg = argc > 4 ? atoi(argv[4]) : 0;
  auto kern = vg == 127 ? probe<127> : vg == 168 ? probe<168> : probe<0>;
  Rec *d; hipMalloc(&d, ng * sizeof(Rec));
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(ng), dim3(nt), kb * 1024, 0, d, kb * 1024, 20000ull);  // warm-up (scratch set-up, code upload)
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(ng), dim3(nt), kb * 1024, 0, d, kb * 1024, 20000ull);  // 200 us at 100 MHz
  hipEventRecord(b);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<Rec> h(ng); hipMemcpy(h.data(), d, ng * sizeof(Rec), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
  for (auto &r : h) cu[(r.xcc << 16) | (r.hwid & 0xff00)].push_back({r.t0, r.t1});  // cu_id[11:8], sh_id[12], se_id[15:13]
  int overl = 0;
  for (auto &kv : cu) {
    auto &v = kv.second; std::sort(v.begin(), v.end());
    for (size_t i = 1; i < v.size(); i++) if (v[i].first < v[i - 1].second) overl++;
  }
  printf("%d workgroups x %d threads, %d KB LDS, VGPR clobber %d: %.3f ms (one spin = 0.200 ms), %zu distinct CUs, %d overlapping pairs on a CU\n", ng, nt, kb, vg, ms,
         cu.size(), overl);
  return 0;
}
