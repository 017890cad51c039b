// The synthetic MFMA neighbour of pk_mfma_repro.hip as a tiny shared library, so that tools/graph_bisect3.py can put it where the library's convolution
// stands (same process, same torch-captured graphs).   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libsynth.so synth_aggressor.hip
#include <hip/hip_runtime.h>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit 0: LDS weight panel + ds_read_b128; bit 1: buffer-descriptor loads; bit 2: ds_bpermute + SiLU + packed bf16 epilogue; bit 3: a barrier per iteration
template <int MODE>
__global__ __launch_bounds__(256) void synth_kernel(bf16* out, int iters, const bf16* x, unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  if (MODE & 1) {
    for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) {
      bf16x8 w;
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = (bf16)(0.001f * (float)((i + k) & 15));
      *(bf16x8*)(lds + (size_t)i * 16) = w;
    }
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  bf16x8 a0, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a0[k] = (bf16)(0.001f * (float)((threadIdx.x + k) & 15)); b[k] = (bf16)(0.002f * (float)((threadIdx.x * 3 + k) & 15)); }
  bf16x8 a1 = a0, a2 = a0, a3 = a0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  unsigned off = (blockIdx.x * 256 + threadIdx.x) * 16;
  for (int i = 0; i < iters; ++i) {
    if (MODE & 2) { b = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off % x_bytes, 0, 0)); off += 4096 * 16; }
    if (MODE & 1) {
      a0 = *(const bf16x8*)(lds + ((i * 4 + 0) & 63) * 1024 + lane * 16); a1 = *(const bf16x8*)(lds + ((i * 4 + 1) & 63) * 1024 + lane * 16);
      a2 = *(const bf16x8*)(lds + ((i * 4 + 2) & 63) * 1024 + lane * 16); a3 = *(const bf16x8*)(lds + ((i * 4 + 3) & 63) * 1024 + lane * 16);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b, acc3, 0, 0, 0);
    if (MODE & 8) __syncthreads();
  }
  f32x4 s = acc0 + acc1 + acc2 + acc3;
  bf16x4 o;
  if (MODE & 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane ^ 16) << 2), __builtin_bit_cast(int, s[j])));
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float v = s[j]; o[j] = (bf16)(v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * v))); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)s[j];
  }
  *(bf16x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = o;
}

// out: >= wgs * 256 * 8 bytes; x: any readable buffer of x_bytes (>= 16)
extern "C" int synth_launch(int mode, void* out, int wgs, int iters, const void* x, unsigned x_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (mode & 1) ? 64 * 1024 : 0;
#define GO(M) do { static bool once = false; if (!once) { (void)hipFuncSetAttribute((const void*)synth_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); once = true; } \
                   synth_kernel<M><<<wgs, 256, lds, s>>>((bf16*)out, iters, (const bf16*)x, x_bytes); } while (0)
  switch (mode & 15) {
    case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break;
    case 8: GO(8); break; case 9: GO(9); break; case 10: GO(10); break; case 11: GO(11); break; case 12: GO(12); break; case 13: GO(13); break; case 14: GO(14); break; default: GO(15); break;
  }
  return (int)hipGetLastError();
}
