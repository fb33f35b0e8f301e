// Microbenchmark (tuning aid): what does a kernel launch cost as a function of the kernel's CODE SIZE when the executed
// path is the same few instructions?  Motivation: the launch-bound searches (one or two launches per BFS / SSSP level)
// got slower whenever never-executed code was added to the level kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/launch_cost.hip -o tools/micro/launch_cost && tools/micro/launch_cost
// Prints, per padding size and grid size, the average time of one launch in a stream of back-to-back launches that
// alternate with a small kernel (as head / level kernels do), and the code size of each kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void small_kernel(float* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}

// executed path: one load, one compare, (one store); `flag` is 0 at run time, the padded branch is dead weight
template <int I>
__device__ __forceinline__ float pad_block(float x, const float* p) {  // ~7.7 KB of code each
#pragma unroll
  for (int i = 0; i < 256; ++i) x = x * 1.0001f + p[(threadIdx.x + (I * 256 + i) * 7) & 1023];
  return x;
}
template <int I, int N>
__device__ __forceinline__ float pad_blocks(float x, const float* p, int flag) {
  if constexpr (I < N) {
    if (flag & (1 << I)) x = pad_block<I>(x, p);  // own basic block each: keeps the compile time sane
    return pad_blocks<I + 1, N>(x, p, flag);
  } else {
    return x;
  }
}
template <int N>
__global__ __launch_bounds__(256) void padded_kernel(float* p, int flag) {
  if (flag) {
    float x = p[threadIdx.x];
    x = pad_blocks<0, N>(x, p, flag);
    p[threadIdx.x] = x;
  } else if (threadIdx.x == 0 && blockIdx.x == 0) {
    p[1] += 1.0f;
  }
}

template <int N>
static int run(float* d, hipStream_t s, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 2000;
  for (int i = 0; i < 50; ++i) {
    hipLaunchKernelGGL(small_kernel, dim3(1), dim3(256), 0, s, d);
    hipLaunchKernelGGL(padded_kernel<N>, dim3(grid), dim3(256), 0, s, d, 0);
  }
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(small_kernel, dim3(1), dim3(256), 0, s, d);
    hipLaunchKernelGGL(padded_kernel<N>, dim3(grid), dim3(256), 0, s, d, 0);
  }
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&padded_kernel<N>)));
  printf("pad blocks %3d (x 7.7 KB)  grid %4d : %.2f us per (small + padded) pair  [binary %zu B, %d VGPRs]\n", N, grid, ms * 1e3 / reps,
         (size_t)fa.binaryVersion, fa.numRegs);
  // the same pair through a captured graph of 64 pairs (how the engine replays level groups)
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 64; ++i) {
    hipLaunchKernelGGL(small_kernel, dim3(1), dim3(256), 0, s, d);
    hipLaunchKernelGGL(padded_kernel<N>, dim3(grid), dim3(256), 0, s, d, 0);
  }
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < 32; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("                          %.2f us per pair inside a replayed graph of 64 pairs\n", ms * 1e3 / (32 * 64));
  (void)hipGraphExecDestroy(ge);
  (void)hipGraphDestroy(g);
  return 0;
}

int main() {
  float* d = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void**>(&d), 4096 * sizeof(float)));
  CHECK(hipMemset(d, 0, 4096 * sizeof(float)));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int grid : {256, 2048}) {
    if (run<0>(d, s, grid)) return 1;
    if (run<1>(d, s, grid)) return 1;
    if (run<3>(d, s, grid)) return 1;
    if (run<6>(d, s, grid)) return 1;
    if (run<12>(d, s, grid)) return 1;
  }
  return 0;
}
