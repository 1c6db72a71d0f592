// Prototype: sparse (CSR by pixel) H-step inner product on gfx950 - is the LDS gather fast enough?
// Entry = (channel u16 | bf16 value << 16).  One wave per pixel at a time, 2 entries per lane per
// batch (packed fp32 math), GW table [K][n] fp32 in LDS, H column uniform (scalar loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int K = 5, KP = 8;

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void sparse_h(const uint2* __restrict__ ent, const int* __restrict__ ptr,
                                                const float* __restrict__ gw /* [n][KP] */, const float* __restrict__ h_t /* [p][KP] */,
                                                float* __restrict__ num /* [p][KP] */, int n, int p, int px_per_wg) {
  extern __shared__ float gwl[];  // [K][n]
  for (int i = threadIdx.x; i < n * K; i += 256) {
    const int kk = i / n, c = i - kk * n;
    gwl[i] = gw[c * KP + kk];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j0 = blockIdx.x * px_per_wg;
  // each wave owns a contiguous run of pixels; the (up to NB) entry batches of the NEXT pixel are
  // requested before the current pixel is processed
  constexpr int NB = 4;
  const int per_wave = px_per_wg / 4;
  const int jb = j0 + wave * per_wave, je = min(p, jb + per_wave);
  auto fetch = [&](int j, uint2 (&e)[NB], int& cnt) {
    const int beg = ptr[j], end = ptr[j + 1];
    cnt = end - beg;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int b = beg + u * 64 + lane;
      e[u] = b < end ? ent[b] : uint2{0u, 0u};
    }
  };
  uint2 ecur[NB], enxt[NB];
  int ccur = 0, cnxt = 0;
  if (jb < je) fetch(jb, ecur, ccur);
  for (int j = jb; j < je; ++j) {
    if (j + 1 < je) fetch(j + 1, enxt, cnxt);
    float hk[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) hk[kk] = h_t[(size_t)j * KP + kk];
    f2 acc[K];
    f2 kl = {0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[kk] = f2{0.f, 0.f};
    auto process = [&](const uint2 e) {
      const int c0 = e.x & 0xffff, c1 = e.y & 0xffff;
      const f2 x = {__uint_as_float(e.x & 0xffff0000u), __uint_as_float(e.y & 0xffff0000u)};
      f2 g[K];
#pragma unroll
      for (int kk = 0; kk < K; ++kk) g[kk] = f2{gwl[kk * n + c0], gwl[kk * n + c1]};
      f2 y = hk[0] * g[0];
#pragma unroll
      for (int kk = 1; kk < K; ++kk) y = hk[kk] * g[kk] + y;
      const f2 r = x * f2{__builtin_amdgcn_rcpf(y.x), __builtin_amdgcn_rcpf(y.y)} + f2{1e-37f, 1e-37f};
#pragma unroll
      for (int kk = 0; kk < K; ++kk) acc[kk] = g[kk] * r + acc[kk];
      kl = x * f2{__builtin_amdgcn_logf(r.x), __builtin_amdgcn_logf(r.y)} + kl;
    };
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (u * 64 < ccur) process(ecur[u]);  // wave-uniform condition
    for (int b = ptr[j] + NB * 64 + lane; b < ptr[j + 1]; b += 64) process(ent[b]);  // rare: > NB batches
    float s[K + 1];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) s[kk] = wave_sum(acc[kk].x + acc[kk].y);
    s[K] = wave_sum(kl.x + kl.y);
    if (lane == 0) {
#pragma unroll
      for (int kk = 0; kk <= K; ++kk) num[(size_t)j * KP + kk] = s[kk];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) ecur[u] = enxt[u];
    ccur = cnxt;
  }
}

int main() {
  const int n = 2048, p = 512 * 512;
  std::mt19937 rng(1);
  std::vector<int> ptr(p + 1);
  std::vector<uint2> ent;
  ent.reserve((size_t)p * 240);
  std::vector<float> spec(n);
  for (int c = 0; c < n; ++c) spec[c] = 0.05f + (float)std::exp(-0.5 * std::pow((c % 300 - 150) / 40.0, 2));
  double tot = 0; for (float v : spec) tot += v;
  std::vector<uint32_t> row;
  for (int j = 0; j < p; ++j) {
    ptr[j] = (int)ent.size();
    row.clear();
    for (int c = 0; c < n; ++c) {
      std::poisson_distribution<int> d(500.0 * spec[c] / tot);
      const int x = d(rng);
      if (x > 0) {
        uint32_t fb; float xf = (float)x; std::memcpy(&fb, &xf, 4);
        row.push_back((uint32_t)c | (fb & 0xffff0000u));
      }
    }
    if (row.size() & 1) row.push_back(0u);  // zero entry: channel 0, x = 0
    for (size_t i = 0; i < row.size(); i += 2) ent.push_back(uint2{row[i], row[i + 1]});
    if (j == 0) printf("nnz of pixel 0: %zu\n", row.size());
  }
  ptr[p] = (int)ent.size();
  printf("pairs %zu  (%.1f nnz/pixel, %.2f GB)\n", ent.size(), 2.0 * ent.size() / p, ent.size() * 8.0 / 1e9);
  std::vector<float> gw((size_t)n * KP, 0.f), ht((size_t)p * KP, 0.f);
  for (int c = 0; c < n; ++c) for (int k = 0; k < K; ++k) gw[c * KP + k] = 0.01f + 0.001f * ((c * 7 + k * 13) % 97);
  for (int j = 0; j < p; ++j) for (int k = 0; k < K; ++k) ht[(size_t)j * KP + k] = 0.2f;
  uint2* d_ent; int* d_ptr; float *d_gw, *d_ht, *d_num;
  hipMalloc(&d_ent, ent.size() * 8); hipMalloc(&d_ptr, (p + 1) * 4); hipMalloc(&d_gw, gw.size() * 4);
  hipMalloc(&d_ht, ht.size() * 4); hipMalloc(&d_num, ht.size() * 4);
  hipMemcpy(d_ent, ent.data(), ent.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ptr, ptr.data(), (p + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_gw, gw.data(), gw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_ht, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  for (int ppw : {128, 256, 512}) {
    const int nblk = (p + ppw - 1) / ppw;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(sparse_h, dim3(nblk), dim3(256), n * K * 4, 0, d_ent, d_ptr, d_gw, d_ht, d_num, n, p, ppw);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(sparse_h, dim3(nblk), dim3(256), n * K * 4, 0, d_ent, d_ptr, d_gw, d_ht, d_num, n, p, ppw);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("px/WG %4d (%5d WGs): %.1f us per launch, %.2f TB/s of entries\n", ppw, nblk, ms / 20 * 1e3, ent.size() * 8.0 / (ms / 20 * 1e-3) / 1e12);
  }
  std::vector<float> out(16);
  hipMemcpy(out.data(), d_num, 64, hipMemcpyDeviceToHost);
  printf("num[0] = %g %g %g %g %g kl %g\n", out[0], out[1], out[2], out[3], out[4], out[5]);
  return 0;
}
