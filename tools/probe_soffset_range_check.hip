// Does the range check of a raw buffer load include the SGPR offset?  (The compiler documentation says soffset is
// "excluded from bounds checking"; the small-graph kernels and the FC data gradient put the wave-uniform part of an
// address there and count on rows past the tensor reading 0.)  Measured on gfx950 / ROCm 7.2, round 3:
//   in-range 100 | voffset beyond 0 | soffset beyond 0 | voffset 60 + soffset 4 -> 0
// i.e. voffset + soffset is what is compared with num_records: an out-of-range SGPR offset reads 0, like a VGPR one.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_soffset_range_check.hip -o /tmp/soff && /tmp/soff
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out, int soff) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 64, 0x00020000);
  // in range via voffset; beyond via voffset; beyond via soffset; voffset in range + soffset pushing beyond
  out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * threadIdx.x, 0, 0));
  out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 64 + 4 * threadIdx.x, 0, 0));
  out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * threadIdx.x, soff, 0));
  out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 60 + 4 * threadIdx.x, soff / 16, 0));
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  float *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 16); (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o, 64);
  float r[4]; (void)hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("in-range %g | voffset beyond %g | soffset beyond %g (116 = NOT range-checked) | voff 60 + soff 4 %g (116 = not checked)\n", r[0], r[1], r[2], r[3]);
  return 0;
}
