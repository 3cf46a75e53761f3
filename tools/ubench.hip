// Micro-benchmarks that decide the summation strategy of the LMS kernels (gfx950).
//  1. dependent v_add_f64 chain: cycles per add (1 wave; 1..8 waves per SIMD)
//  2. LDS-fed serial chain: ds_read_b128 + 2 adds per pair
//  3. v_mfma_f64_4x4x4 (4 blocks) with B = ones: is it a k-ordered chain of IEEE adds?
//     (bitwise compare against sequential double adds), lane maps, dependent latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void k_addchain(const double* in, double* out, long long* cyc, int n) {
  double acc = in[threadIdx.x & 63];
  double a = in[64], b = in[65], c = in[66], d = in[67];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 4) {
    acc += a; acc += b; acc += c; acc += d;
    asm volatile("" : "+v"(acc));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// two independent chains interleaved
__global__ void k_addchain2(const double* in, double* out, long long* cyc, int n) {
  double acc = in[threadIdx.x & 63], acc2 = in[(threadIdx.x + 1) & 63];
  double a = in[64], b = in[65];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 2) {
    acc += a; acc2 += b; acc += b; acc2 += a;
    asm volatile("" : "+v"(acc), "+v"(acc2));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// one lane (or `lanes` lanes) walks products in LDS sequentially
__global__ void k_ldschain(const double* in, double* out, long long* cyc, int n, int lanes) {
  __shared__ double buf[4][2048];
  for (int i = threadIdx.x; i < 4 * 2048; i += blockDim.x) (&buf[0][0])[i] = in[i & 1023];
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  double acc = 0.;
  if ((int)threadIdx.x < lanes) {
    const double* p = buf[threadIdx.x & 3];
#pragma unroll 8
    for (int i = 0; i < n; i += 2) {
      double2 v = *(const double2*)(p + i);
      acc += v.x; acc += v.y;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// MFMA f64 4x4x4 (4 blocks): D = A*B + C. Probe lane maps with one-hot operands.
__global__ void k_mfma_probe(const double* a, const double* b, const double* c, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
// chain: nsteps dependent MFMAs, A read from memory per step (lane l, step s: a[s*64+l]); B = b[l]
__global__ void k_mfma_chain(const double* a, const double* b, double* d, long long* cyc, int nsteps) {
  int l = threadIdx.x;
  double acc = 0.;
  double bb = b[l];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int s = 0; s < nsteps; s++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s * 64 + l], bb, acc, 0, 0, 0);
  long long t1 = __builtin_amdgcn_s_memtime();
  d[l] = acc;
  if (l == 0) cyc[0] = t1 - t0;
}
// same but A kept in a register (pure dependent-MFMA latency)
__global__ void k_mfma_lat(const double* a, const double* b, double* d, long long* cyc, int nsteps) {
  int l = threadIdx.x;
  double acc = 0., aa = a[l], bb = b[l];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < nsteps; s += 4) {
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, bb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, bb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, bb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, bb, acc, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  d[l] = acc;
  if (l == 0) cyc[0] = t1 - t0;
}
typedef double double4_ __attribute__((ext_vector_type(4)));
__global__ void k_mfma16_lat(const double* a, const double* b, double* d, long long* cyc, int nsteps) {
  int l = threadIdx.x;
  double4_ acc = {0., 0., 0., 0.};
  double aa = a[l], bb = b[l];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < nsteps; s += 2) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb, acc, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  d[l] = acc[0] + acc[1] + acc[2] + acc[3];
  if (l == 0) cyc[0] = t1 - t0;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  double *din, *dout; long long* dcyc;
  CK(hipMalloc(&din, 1 << 20)); CK(hipMalloc(&dout, 1 << 24)); CK(hipMalloc(&dcyc, 64));
  std::vector<double> h(8192);
  srand(1);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 3.0;
  CK(hipMemcpy(din, h.data(), 8192 * 8, hipMemcpyHostToDevice));
  long long cyc; const int N = 1 << 16;
  auto wall = [&](auto launch) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
  for (int thr : {64, 256, 512, 1024}) {
    float ms = wall([&] { k_addchain<<<1, thr>>>(din, dout, dcyc, N); });
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("addchain threads=%4d: %.2f memtime-ticks/add, wall %.3f ms => %.2f ns/add\n", thr, cyc / (double)N, ms, ms * 1e6 / N);
  }
  {
    float ms = wall([&] { k_addchain2<<<1, 64>>>(din, dout, dcyc, N); });
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("addchain2 (2 interleaved chains, %d adds each): %.2f ticks per pair, wall %.2f ns per pair\n", N, cyc / (double)N, ms * 1e6 / N);
  }
  for (int lanes : {1, 2, 4, 64}) {
    float ms = wall([&] { k_ldschain<<<1, 64>>>(din, dout, dcyc, 2048, lanes); });
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("ldschain lanes=%2d: %.2f ticks/term (2048 terms) wall %.3f ms\n", lanes, cyc / 2048.0, ms);
  }
  // ---- MFMA lane-map probe
  {
    std::vector<double> a(64), b(64), c(64, 0.), d(64);
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dd, 512));
    // A one-hot at lane la (value 2), B all lanes = distinct primes-ish to decode k/j
    printf("mfma_f64_4x4x4 lane map probe: for A one-hot at lane la with B[l]=l+1: nonzero D lanes -> value\n");
    for (int la = 0; la < 64; la += 1) {
      for (int l = 0; l < 64; l++) { a[l] = (l == la) ? 1.0 : 0.0; b[l] = l + 1; }
      CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
      k_mfma_probe<<<1, 64>>>(da, db, dc, dd); CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
      if (la < 20 || la % 16 == 0) { printf(" la=%2d:", la); for (int l = 0; l < 64; l++) if (d[l] != 0.) printf(" D[%d]=%g", l, d[l]); printf("\n"); }
    }
    // ---- numerics: B = all ones. Then D[lane] = C + sum_k A[i][k]. Find for each D lane which A lanes feed it and in which order.
    // Use the probe above generically: with B=1 everywhere, D lane dl gets A lanes set S(dl). Determine order by testing
    // non-associativity: values big, tiny, -big ...
    for (int l = 0; l < 64; l++) b[l] = 1.0;
    CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    std::vector<std::vector<int>> feeders(64);
    for (int la = 0; la < 64; la++) {
      for (int l = 0; l < 64; l++) a[l] = (l == la) ? 1.0 : 0.0;
      CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
      k_mfma_probe<<<1, 64>>>(da, db, dc, dd); CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
      for (int l = 0; l < 64; l++) if (d[l] != 0.) feeders[l].push_back(la);
    }
    printf("with B=ones: feeders of D lane 0:"); for (int f : feeders[0]) printf(" %d", f); printf(" ; D lane 1:"); for (int f : feeders[1]) printf(" %d", f);
    printf(" ; D lane 17:"); for (int f : feeders[17]) printf(" %d", f); printf("\n");
    // random trials: compare D with all 24 orders of sequential adds (C first) and with exact/fma-free variants
    int perm[24][4], np = 0; { int p[4] = {0, 1, 2, 3}; do { memcpy(perm[np++], p, 16); } while (std::next_permutation(p, p + 4)); }
    long cnt[24] = {0}; long trials = 0, cfirst_any = 0;
    for (int tr = 0; tr < 400; tr++) {
      for (int l = 0; l < 64; l++) { a[l] = ldexp(rand() / (double)RAND_MAX - 0.5, rand() % 40 - 20); c[l] = ldexp(rand() / (double)RAND_MAX - 0.5, rand() % 40 - 20); }
      CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
      k_mfma_probe<<<1, 64>>>(da, db, dc, dd); CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
      for (int l = 0; l < 64; l++) {
        if (feeders[l].size() != 4) continue;
        trials++; bool any = false;
        for (int q = 0; q < 24; q++) {
          volatile double s = c[l];
          for (int k = 0; k < 4; k++) s = s + a[feeders[l][perm[q][k]]];
          double sv = s;
          if (!memcmp(&sv, &d[l], 8)) { cnt[q]++; any = true; }
        }
        if (any) cfirst_any++;
      }
    }
    printf("mfma numerics: %ld trials; matched some C-first sequential order: %ld\n", trials, cfirst_any);
    for (int q = 0; q < 24; q++) if (cnt[q] > trials * 0.5) printf("  order %d%d%d%d (indices into ascending feeder lanes) matches %ld/%ld\n", perm[q][0], perm[q][1], perm[q][2], perm[q][3], cnt[q], trials);
    for (int n : {1024}) {
      float ms = wall([&] { k_mfma_lat<<<1, 64>>>(da, db, dd, dcyc, n); });
      CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
      printf("mfma 4x4x4 dependent latency: %.2f ticks per MFMA (%.2f per term), wall %.2f ns per MFMA\n", cyc / (double)n, cyc / (4.0 * n), ms * 1e6 / n);
      ms = wall([&] { k_mfma16_lat<<<1, 64>>>(da, db, dd, dcyc, n); });
      CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
      printf("mfma 16x16x4 dependent latency: %.2f ticks per MFMA, wall %.2f ns per MFMA\n", cyc / (double)n, ms * 1e6 / n);
      ms = wall([&] { k_mfma_chain<<<1, 64>>>(din, db, dd, dcyc, 100); });
      CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
      printf("mfma 4x4x4 chain with global A loads (100 steps): %.2f ticks per MFMA\n", cyc / 100.0);
    }
  }
  // memtime tick rate
  {
    float ms = wall([&] { k_addchain<<<1, 64>>>(din, dout, dcyc, 1 << 20); });
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("memtime: %lld ticks in %.3f ms => %.1f MHz tick\n", cyc, ms, cyc / (ms * 1e3));
  }
  return 0;
}
