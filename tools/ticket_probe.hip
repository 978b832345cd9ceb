// What a grid-wide meeting point costs on MI355X, measured: every workgroup of a launch takes a ticket from ONE
// device-scope counter (returning atomicAdd), or from the counter of "its" XCD (workgroup b -> counter b % 8) and the
// eight last arrivers from a top counter.  Compared with an empty kernel of the same grid and with two dependent
// empty launches.  hipcc --offload-arch=gfx950 -O3 tools/ticket_probe.hip -o tools/ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void Empty(unsigned* sink) { if (threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1; }

__global__ void OneTicket(unsigned* counter, unsigned* last_flag) {
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(counter, 1u);
    if (t == gridDim.x - 1) { *last_flag = t; *counter = 0; }
  }
}

__global__ void ShardedTicket(unsigned* counters /* [8] + top at [16] */, unsigned* last_flag) {
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x % 8;
    const unsigned per = (gridDim.x - x + 7) / 8;
    const unsigned t = atomicAdd(counters + x * 32, 1u);   // one 128-byte line per counter
    if (t == per - 1) {
      counters[x * 32] = 0;
      const unsigned top = atomicAdd(counters + 8 * 32, 1u);
      if (top == 7) { *last_flag = top; counters[8 * 32] = 0; }
    }
  }
}

template <typename F>
float Time(F launch, int iters) {
  hipEvent_t a, z;
  hipEventCreate(&a); hipEventCreate(&z);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(z);
  hipEventSynchronize(z);
  float ms = 0;
  hipEventElapsedTime(&ms, a, z);
  return ms * 1000.f / iters;   // us per launch
}

int main() {
  unsigned* buf;
  OK(hipMalloc(&buf, 4096 * 4));
  OK(hipMemset(buf, 0, 4096 * 4));
  unsigned* flag = buf + 2048;
  const int iters = 2000;
  for (int grid : {256, 1024, 4096, 16384}) {
    const float empty = Time([&] { Empty<<<grid, 256>>>(flag); }, iters);
    const float one = Time([&] { OneTicket<<<grid, 256>>>(buf, flag); }, iters);
    const float sharded = Time([&] { ShardedTicket<<<grid, 256>>>(buf, flag); }, iters);
    std::printf("{\"workgroups\": %d, \"empty_kernel_us\": %.2f, \"one_counter_us\": %.2f, \"per_xcd_counters_us\": %.2f, "
                "\"ns_per_ticket_one_counter\": %.1f}\n", grid, empty, one, sharded, (one - empty) * 1000.f / grid);
  }
  OK(hipDeviceSynchronize());
  return 0;
}
