// Do L2-served gathers and HBM-cold gathers overlap on MI355X, and at which level do they get in
// each other's way?  128-byte row pieces, slice = blockIdx % 8 (one XCD per slice) as in the
// aggregation kernel.  Hit stream: random rows of a 16k-row matrix (the slice's 2 MB stay in the
// XCD's L2).  Miss stream: random rows of a 4 GB table, a fresh index set per launch.
// Persistent workgroups pull 256-index windows from per-(slice, stream) counters.
//   mode 0: hit stream only        mode 1: miss stream only
//   mode 2: every workgroup alternates between the streams (4 hit windows, then 1 miss window)
//   mode 3: a workgroup serves one stream, chosen by its index (1 in 5 serves misses)
//   mode 4: a workgroup serves one stream, chosen by the compute unit it runs on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Stream { const float *x; const int *idx; int n; };

__device__ __forceinline__ f4 window(const Stream &s, int w, int slice, int wave, int g, int t) {
  // a window = 256 indices: 64 per wave, 8 loads in flight per lane group
  f4 acc = f4(0.f);
  const int b0 = w * 256 + wave * 64;
  int c[8];
  f4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) c[u] = s.idx[min(b0 + u * 8 + g, s.n - 1)];
#pragma unroll
  for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f4 *>(s.x + (size_t)c[u] * 260 + slice * 32 + t * 4);
#pragma unroll
  for (int u = 0; u < 8; u++) acc += v[u];
  return acc;
}

__global__ __launch_bounds__(256, 8) void probe_kernel(Stream hit, Stream miss, int mode, int *counters, int miss_cu_mod, float *out) {
  __shared__ int next;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = blockIdx.x % 8;
  const int vb = blockIdx.x / 8;
  int *c_hit = counters + slice * 2, *c_miss = counters + slice * 2 + 1;
  const int w_hit = (hit.n + 255) / 256, w_miss = (miss.n + 255) / 256;
  int role;   // 0 hit, 1 miss, 2 both
  if (mode == 0) role = 0;
  else if (mode == 1) role = 1;
  else if (mode == 2) role = 2;
  else if (mode == 3) role = (vb % 5 == 4);
  else {
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);   // HW_ID: CU_ID in bits 11:8
    role = (int)(((hw >> 8) & 15) % miss_cu_mod == 0);
  }
  f4 acc = f4(0.f);
  bool hit_left = role != 1, miss_left = role != 0;
  int turn = 0;
  constexpr int kGrab = 8;   // windows per counter bump (bumps of one address retire ~10 per us)
  while (hit_left || miss_left) {
    const bool take_miss = miss_left && (!hit_left || (role == 2 ? (turn % 5 == 4) : true));
    __syncthreads();
    if (threadIdx.x == 0) next = atomicAdd(take_miss ? c_miss : c_hit, kGrab);
    __syncthreads();
    const int w0 = next;
    const Stream &st = take_miss ? miss : hit;
    const int w_end = take_miss ? w_miss : w_hit;
    if (w0 >= w_end) { if (take_miss) miss_left = false; else hit_left = false; }
    for (int w = w0; w < min(w0 + kGrab, w_end); w++) acc += window(st, w, slice, wave, g, t);
    turn++;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char **argv) {
  const long long hit_rows = 16326, table_rows = 4000000;
  const int n_hit = argc > 1 ? atoi(argv[1]) : 708072, n_miss = argc > 2 ? atoi(argv[2]) : 186758;
  const int sets = 8;
  float *xh, *xt, *out;
  int *ih, *im, *counters;
  CHECK(hipMalloc(&xh, hit_rows * 1040));
  CHECK(hipMemset(xh, 0, hit_rows * 1040));
  CHECK(hipMalloc(&xt, table_rows * 1040));
  CHECK(hipMemset(xt, 0, table_rows * 1040));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMalloc(&counters, 64 * sizeof(int)));
  std::vector<int> h(n_hit), m((size_t)n_miss * sets);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto &v : h) v = (int)(rnd() % (unsigned long long)hit_rows);
  // misses: 111.7k distinct rows per set, each referenced 1.67x on average like the real batch
  for (int k = 0; k < sets; k++) {
    std::vector<int> distinct(111738);
    for (auto &v : distinct) v = (int)(rnd() % (unsigned long long)table_rows);
    for (int i = 0; i < n_miss; i++) m[(size_t)k * n_miss + i] = distinct[rnd() % distinct.size()];
  }
  CHECK(hipMalloc(&ih, h.size() * 4));
  CHECK(hipMemcpy(ih, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&im, m.size() * 4));
  CHECK(hipMemcpy(im, m.data(), m.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char *names[] = {"hits only", "misses only", "mixed in every workgroup", "workgroup roles", "compute-unit roles"};
  for (int mode = 0; mode < 5; mode++) {
    for (int cu_mod : {3, 4, 2}) {
      if (mode != 4 && cu_mod != 3) continue;
      for (int wgs_per_cu : {8, 4}) {
        const dim3 grid(256 * wgs_per_cu), block(256);
        float total = 0;
        const int reps = 16;
        for (int i = -2; i < reps; i++) {
          CHECK(hipMemsetAsync(counters, 0, 64 * sizeof(int)));
          Stream hs{xh, ih, n_hit}, ms{xt, im + (size_t)((i + 2) % sets) * n_miss, n_miss};
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(probe_kernel, grid, block, 0, 0, hs, ms, mode, counters, cu_mod, out);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms_;
          CHECK(hipEventElapsedTime(&ms_, e0, e1));
          if (i >= 0) total += ms_;
        }
        printf("%-26s cu_mod %d  %d workgroups/CU: %7.1f us\n", names[mode], cu_mod, wgs_per_cu, total / reps * 1e3);
      }
    }
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
