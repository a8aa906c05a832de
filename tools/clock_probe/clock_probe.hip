// Average shader clock while the ring GEMM runs: a one-wave kernel on a second stream reads the shader-clock counter
// (clock64) and the constant wall-clock counter (wall_clock64) around a fixed wall-time spin, (a) on an idle chip,
// (b) beside back-to-back launches of mo_gemm_bf16_256 with random / all-zero operands.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/clock_probe/clock_probe.hip -L multimodal_outage_amd -lmo_hip \
//         -Wl,-rpath,$PWD/multimodal_outage_amd -o gpurun_out/clock_probe && gpurun_out/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include "mo_hip.h"

__global__ void clk_probe(unsigned long long* out, long long spin_ticks) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  while ((long long)(wall_clock64() - w0) < spin_ticks) {}
  out[0] = clock64() - c0;
  out[1] = wall_clock64() - w0;
}

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

int main() {
  const int N = 3000, KP = 3008, J = 24576;
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", wall_khz);
  std::vector<uint16_t> hA((size_t)N * KP, 0), hX((size_t)N * J);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) hA[(size_t)r * KP + c] = bf16_of(rnd());
  for (auto& v : hX) v = bf16_of(rnd());
  void *A, *X, *Y, *Z; unsigned long long* out;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&X, hX.size() * 2); hipMalloc(&Y, hX.size() * 2); hipMalloc(&Z, hX.size() * 2);
  hipMalloc(&out, 16);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice);
  hipMemset(Z, 0, hX.size() * 2);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  const long long spin = (long long)wall_khz * 4;          // 4 ms of wall time
  auto probe = [&](const char* what) {
    unsigned long long h[2];
    hipLaunchKernelGGL(clk_probe, dim3(1), dim3(64), 0, s2, out, spin);
    hipStreamSynchronize(s2);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-44s shader clock %.0f MHz (over %.2f ms)\n", what, (double)h[0] / h[1] * wall_khz / 1e3, (double)h[1] / wall_khz);
  };
  probe("idle chip:");
  for (int zero = 0; zero < 2; ++zero) {
    const void* B = zero ? Z : X;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) mo_gemm_bf16_256(A, KP, KP, B, J, 1, nullptr, J, N, J, N, 0, Y, s1);
    hipStreamSynchronize(s1);
    hipEventRecord(e0, s1);
    const int reps = 24;
    for (int i = 0; i < reps; ++i) mo_gemm_bf16_256(A, KP, KP, B, J, 1, nullptr, J, N, J, N, 0, Y, s1);
    hipEventRecord(e1, s1);
    probe(zero ? "beside the ring GEMM, all-zero B operand:" : "beside the ring GEMM, random operands:");
    hipStreamSynchronize(s1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("  GEMM %.0f us per launch = %.0f TFLOP/s\n", ms / reps * 1e3, 2.0 * N * N * J / (ms / reps * 1e-3) / 1e12);
  }
  return 0;
}
