// Ceiling of 128-byte random row-piece gathers on MI355X: what the aggregation kernel's access
// pattern can reach with no index chains, no row structure and no epilogue.  8-lane groups fetch
// the 128-byte piece `slice` of a random row; slice = blockIdx % 8 (one XCD per slice, as in
// the kernel) or all slices on every XCD; U loads in flight per lane.
//   usage: gather_ceiling <rows> <pitch_floats> <gathers_per_slice> <pinned 0/1> [file of int32 row ids]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R>
__device__ __forceinline__ int group_bcast(int v) { return __builtin_amdgcn_ds_swizzle(v, (R << 5) | 0x18); }

// The kernel's way of getting indices to the lane groups: lane 8 g + j loads index 8 j + g of a
// 64-index window (one coalesced load per lane), ds_swizzle broadcasts inside the group; a
// second swizzle per round stands for the edge weight.  U = 8 only.
__global__ __launch_bounds__(256, 8) void gather_swizzle_kernel(const float *x, long long pitch, const int *idx,
                                                                long long n_idx, int pinned, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = pinned ? blockIdx.x % 8 : (blockIdx.x / 8) % 8;
  const long long vb = blockIdx.x / 8;
  f4 acc = f4(0.f);
  for (long long base = (vb * 4 + wave) * 256; base < n_idx; base += (long long)(gridDim.x / 8) * 4 * 256) {
#pragma unroll 1
    for (int rep = 0; rep < 4; rep++) {
      const long long b0 = base + rep * 64;
      if (b0 >= n_idx) break;
      const int my_c = idx[min(b0 + 8 * t + g, n_idx - 1)];
      const int my_w = my_c ^ 0x3f800000;
      int c[8];
      float w[8];
      f4 v[8];
#define RND(r) c[r] = group_bcast<r>(my_c); w[r] = __int_as_float(group_bcast<r>(my_w) & 0x3fffffff);
      RND(0) RND(1) RND(2) RND(3) RND(4) RND(5) RND(6) RND(7)
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f4 *>(x + (size_t)c[u] * pitch + slice * 32 + t * 4);
#pragma unroll
      for (int u = 0; u < 8; u++) acc += w[u] * v[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U>
__global__ __launch_bounds__(256, 8) void gather_kernel(const float *x, long long pitch, const int *idx,
                                                        long long n_idx, int pinned, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = pinned ? blockIdx.x % 8 : (blockIdx.x / 8) % 8;
  const long long vb = blockIdx.x / 8;
  // a wave covers 64 * U / 8... indices [base, base + 8 * U): group g takes idx[base + u * 8 + g]
  const long long per_wave = 8 * U;
  f4 acc = f4(0.f);
  for (long long base = (vb * 4 + wave) * per_wave * 4; base < n_idx; base += (long long)(gridDim.x / 8) * 4 * per_wave * 4) {
#pragma unroll 1
    for (int rep = 0; rep < 4; rep++) {
      const long long b0 = base + rep * per_wave;
      if (b0 >= n_idx) break;
      int c[U];
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) c[u] = idx[min(b0 + u * 8 + g, n_idx - 1)];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const f4 *>(x + (size_t)c[u] * pitch + slice * 32 + t * 4);
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char **argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 128064;
  const long long pitch = argc > 2 ? atoll(argv[2]) : 256;
  const long long n_idx = argc > 3 ? atoll(argv[3]) : 894830;
  const int pinned = argc > 4 ? atoi(argv[4]) : 1;
  float *x, *out;
  int *idx;
  CHECK(hipMalloc(&x, rows * pitch * 4));
  CHECK(hipMemset(x, 0, rows * pitch * 4));
  CHECK(hipMalloc(&out, 256));
  std::vector<int> h(n_idx);
  unsigned long long s = 88172645463325252ull;
  for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % (unsigned long long)rows); }
  if (argc > 5) {  // row ids from a file (e.g. the column array of a real batch), clamped to the matrix
    FILE *f = fopen(argv[5], "rb");
    if (!f) { printf("cannot open %s\n", argv[5]); return 1; }
    const size_t got = fread(h.data(), 4, (size_t)n_idx, f);
    fclose(f);
    if ((long long)got != n_idx) { printf("short file: %zu ids\n", got); return 1; }
    for (auto &v : h) v = (int)((unsigned)v % (unsigned long long)rows);
  }
  CHECK(hipMalloc(&idx, n_idx * 4));
  CHECK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](int U, int blocks_per_slice) {
    const dim3 grid(blocks_per_slice * 8), block(256);
    auto launch = [&]() {
      if (U == 0) hipLaunchKernelGGL(gather_swizzle_kernel, grid, block, 0, 0, x, pitch, idx, n_idx, pinned, out);
      else if (U == 8) hipLaunchKernelGGL(gather_kernel<8>, grid, block, 0, 0, x, pitch, idx, n_idx, pinned, out);
      else if (U == 16) hipLaunchKernelGGL(gather_kernel<16>, grid, block, 0, 0, x, pitch, idx, n_idx, pinned, out);
      else hipLaunchKernelGGL(gather_kernel<4>, grid, block, 0, 0, x, pitch, idx, n_idx, pinned, out);
    };
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 20 * 1e3;
    printf("rows %lld pitch %lld gathers/slice %lld pinned %d U %2d blocks/slice %5d: %7.1f us  %6.2f TB/s gathered  %6.1f G lines/s\n",
           rows, pitch, n_idx, pinned, U, blocks_per_slice, us, n_idx * 8 * 128.0 / us / 1e6, n_idx * 8 / us / 1e3);
  };
  for (int U : {0, 8, 16})
    for (int bps : {256, 1024, 4096}) run(U, bps);
  CHECK(hipDeviceSynchronize());
  return 0;
}
