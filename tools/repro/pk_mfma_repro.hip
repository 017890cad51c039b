// Stand-alone check (no torch, no library): does a kernel doing packed-fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_fma_f32) keep its results when an
// MFMA kernel of ANOTHER stream is resident beside it?   hipcc --offload-arch=gfx950 -O3 -o pk_mfma_repro pk_mfma_repro.hip && ./pk_mfma_repro
// The interpolation kernel mirrors mgdt's bilinear_kernel<bf16, 8> (four 16-byte bf16 loads, an 8-channel lerp, v_cvt_pk_bf16_f32, a 16-byte store);
// its output alone is the reference, then it is launched again and again on stream B while stream A runs MFMA waves, and compared bit for bit.
// Build the same file with -Xclang -target-feature -Xclang -packed-fp32-ops for the control (no packed instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void lerp_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int H, int W, int C, int Ho, int Wo, unsigned total) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int Q = C / 8, q = i % Q, p = i / Q, ox = p % Wo, r = p / Wo, oy = r % Ho, n = r / Ho;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 > H - 1 ? H - 1 : y0; x0 = x0 > W - 1 ? W - 1 : x0;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const bf16* b = x + ((size_t)n * H * W) * C + q * 8;
    const bf16x8 v00 = *(const bf16x8*)(b + ((size_t)y0 * W + x0) * C), v01 = *(const bf16x8*)(b + ((size_t)y0 * W + x1) * C);
    const bf16x8 v10 = *(const bf16x8*)(b + ((size_t)y1 * W + x0) * C), v11 = *(const bf16x8*)(b + ((size_t)y1 * W + x1) * C);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      o[k] = (bf16)(((float)v00[k] * lx0 + (float)v01[k] * lx1) * ly0 + ((float)v10[k] * lx0 + (float)v11[k] * lx1) * ly1);
    *(bf16x8*)(y + ((size_t)(n * Ho + oy) * Wo + ox) * C + q * 8) = o;
  }
}

// the aggressor: MFMA waves that also do what mgdt's conv_igemm does around its MFMAs - a weight panel in (dynamic) LDS read with ds_read_b128, activations
// through a buffer descriptor, a ds_bpermute transpose and packed bf16 conversion in the epilogue
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, const bf16* x, unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) {
    bf16x8 w;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (bf16)(0.001f * (float)((i + k) & 15));
    *(bf16x8*)(lds + (size_t)i * 16) = w;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  unsigned off = (blockIdx.x * 256 + threadIdx.x) * 16;
  for (int i = 0; i < iters; ++i) {
    const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off % x_bytes, 0, 0));
    off += 4096 * 16;
    const bf16x8 a0 = *(const bf16x8*)(lds + ((i * 4 + 0) & 63) * 1024 + lane * 16), a1 = *(const bf16x8*)(lds + ((i * 4 + 1) & 63) * 1024 + lane * 16);
    const bf16x8 a2 = *(const bf16x8*)(lds + ((i * 4 + 2) & 63) * 1024 + lane * 16), a3 = *(const bf16x8*)(lds + ((i * 4 + 3) & 63) * 1024 + lane * 16);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b, acc3, 0, 0, 0);
  }
  f32x4 s = acc0 + acc1 + acc2 + acc3;
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane ^ 16) << 2), __builtin_bit_cast(int, s[j])));
  typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float v = s[j]; o[j] = (bf16)(v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * v))); }
  *(bf16x4*)((bf16*)out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = o;
}

// the plain form: nothing but MFMAs (the one that DOES disturb the packed-fp32 kernel inside tools/graph_bisect3.py)
__global__ __launch_bounds__(256) void mfma_plain_kernel(bf16* out, int iters) {
  bf16x8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = (bf16)(0.001f * (float)((threadIdx.x + k) & 15)); b[k] = (bf16)(0.002f * (float)((threadIdx.x * 3 + k) & 15)); }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
  }
  const f32x4 s = acc0 + acc1 + acc2 + acc3;
  typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)s[j];
  *(bf16x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = o;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3000, mfma_wgs = argc > 2 ? atoi(argv[2]) : 256, mfma_iters = argc > 3 ? atoi(argv[3]) : 4000;
  const bool plain = argc > 4 && atoi(argv[4]) != 0;     // graph instances: plain 300-iteration MFMA kernel (256 workgroups) in front of the interpolation
  const int N = 32, H = 20, W = 20, C = 128, Ho = 40, Wo = 40;
  const size_t nin = (size_t)N * H * W * C, nout = (size_t)N * Ho * Wo * C;
  std::vector<unsigned short> hx(nin);
  unsigned s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; const float f = ((float)(s >> 8) / 16777216.f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16 *dx, *dy; float* dm;
  CK(hipMalloc(&dx, nin * 2)); CK(hipMalloc(&dy, nout * 2)); CK(hipMalloc(&dm, (size_t)mfma_wgs * 256 * 8));
  CK(hipMemcpy(dx, hx.data(), nin * 2, hipMemcpyHostToDevice));
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  CK(hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  const unsigned total = (unsigned)(nout / 8);
  std::vector<unsigned short> ref(nout), got(nout);
  lerp_kernel<<<6400, 256, 0, sb>>>(dx, dy, H, W, C, Ho, Wo, total);
  CK(hipStreamSynchronize(sb)); CK(hipMemcpy(ref.data(), dy, nout * 2, hipMemcpyDeviceToHost));
  // alone, repeated: must reproduce itself
  long bad_alone = 0;
  for (int r = 0; r < 50; ++r) {
    CK(hipMemsetAsync(dy, 0, nout * 2, sb));
    lerp_kernel<<<6400, 256, 0, sb>>>(dx, dy, H, W, C, Ho, Wo, total);
    CK(hipStreamSynchronize(sb)); CK(hipMemcpy(got.data(), dy, nout * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nout; ++i) bad_alone += got[i] != ref[i];
  }
  printf("alone: %ld wrong elements in 50 launches\n", bad_alone);
  // beside MFMA waves of another stream
  long bad = 0, bad_launches = 0, odd = 0, hi16 = 0;
  for (int r = 0; r < rounds; ++r) {
    if (r % 4 == 0) mfma_kernel<<<mfma_wgs, 256, 64 * 1024, sa>>>(dm, mfma_iters, dx, (unsigned)(nin * 2));
    CK(hipMemsetAsync(dy, 0, nout * 2, sb));
    lerp_kernel<<<6400, 256, 0, sb>>>(dx, dy, H, W, C, Ho, Wo, total);
    CK(hipStreamSynchronize(sb)); CK(hipMemcpy(got.data(), dy, nout * 2, hipMemcpyDeviceToHost));
    long b = 0;
    for (size_t i = 0; i < nout; ++i)
      if (got[i] != ref[i]) { ++b; odd += i & 1; hi16 += ((i / 8) % 64) >= 48; }     // i / 8 = thread index (one 8-channel vector per thread): lane = (i / 8) % 64
    bad += b; bad_launches += b != 0;
  }
  CK(hipDeviceSynchronize());
  // the same two kernels as hipGraph instances: S graphs (mfma -> lerp into the instance's own buffer), replayed concurrently on S streams
  {
    const int S = 5;
    std::vector<hipStream_t> st(S); std::vector<hipGraphExec_t> ge(S); std::vector<bf16*> outb(S); std::vector<float*> mb(S);
    for (int j = 0; j < S; ++j) {
      CK(hipStreamCreate(&st[j])); CK(hipMalloc(&outb[j], nout * 2)); CK(hipMalloc(&mb[j], (size_t)(mfma_wgs > 256 ? mfma_wgs : 256) * 256 * 8));
      hipGraph_t g;
      CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeThreadLocal));
      if (plain) mfma_plain_kernel<<<256, 256, 0, st[j]>>>((bf16*)mb[j], 300);
      else mfma_kernel<<<mfma_wgs, 256, 64 * 1024, st[j]>>>(mb[j], mfma_iters / 4, dx, (unsigned)(nin * 2));
      lerp_kernel<<<6400, 256, 0, st[j]>>>(dx, outb[j], H, W, C, Ho, Wo, total);
      CK(hipStreamEndCapture(st[j], &g));
      CK(hipGraphInstantiate(&ge[j], g, nullptr, nullptr, 0));
    }
    long gbad = 0, gbad_launch = 0, godd = 0, ghi = 0;
    const int grounds = rounds / 10;
    for (int r = 0; r < grounds; ++r) {
      for (int k = 0; k < 3; ++k)
        for (int j = 0; j < S; ++j) CK(hipGraphLaunch(ge[j], st[j]));
      CK(hipDeviceSynchronize());
      for (int j = 0; j < S; ++j) {
        CK(hipMemcpy(got.data(), outb[j], nout * 2, hipMemcpyDeviceToHost));
        long b = 0;
        for (size_t i = 0; i < nout; ++i)
          if (got[i] != ref[i]) { ++b; godd += i & 1; ghi += ((i / 8) % 64) >= 48; }
        gbad += b; gbad_launch += b != 0;
      }
    }
    printf("as %d hipGraph instances replayed concurrently: %ld wrong elements in %ld of %d graph launches; %ld odd elements, %ld in lanes 48..63\n", S, gbad, gbad_launch,
           grounds * S, godd, ghi);
    bad += gbad;
  }
  printf("beside %d MFMA workgroups x %d iterations: %ld wrong elements in %ld of %d launches; of the wrong ones %ld are odd elements, %ld in lanes 48..63\n",
         mfma_wgs, mfma_iters, bad, bad_launches, rounds, odd, hi16);
  return bad ? 1 : 0;
}
