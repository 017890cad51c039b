// Which packed-fp32 encoding goes wrong beside MFMA waves of other hardware queues?  Self-checking: every lane runs one packed instruction form in a loop
// (inline asm, so the compiler cannot change it), computes the two expected halves with scalar v_fma_f32 / v_mul_f32 and counts disagreements, split by
// half (.lo / .hi) and by lane quarter.  S streams each run [plain MFMA kernel -> checking kernel] back to back.
//   hipcc --offload-arch=gfx950 -O2 -o pk_form_repro pk_form_repro.hip && ./pk_form_repro [rounds] [streams] [neighbour: 0 MFMA, 1 none]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

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
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)s[j];
  *(bf16x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = o;
}

__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// counters: [form 0..9, 10 = the load form][half 0/1][lane quarter 0..3]
template <int FORM>
__global__ __launch_bounds__(256) void pk_check_kernel(unsigned* cnt, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  f32x2 a = {1.f + 0.001f * (float)(tid & 1023), 2.f + 0.003f * (float)(tid & 511)}, b = {0.5f + 0.002f * (float)(tid & 255), 1.5f - 0.001f * (float)(tid & 127)};
  f32x2 c = {0.25f * (float)(lane + 1), -0.125f * (float)(lane + 3)};
  unsigned blo = 0, bhi = 0;
  for (int i = 0; i < iters; ++i) {
    f32x2 r;
    float elo, ehi;
    if (FORM == 0) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.x, b.x, c.x); ehi = sfma(a.y, b.y, c.y); }
    if (FORM == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.x, b.y, c.x); ehi = sfma(a.y, b.x, c.y); }
    if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); elo = smul(a.y, b.x); ehi = smul(a.x, b.y); }
    if (FORM == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); elo = smul(a.x, b.x); ehi = smul(a.y, b.y); }
    if (FORM == 4) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); elo = sadd(a.x, b.x); ehi = sadd(a.y, b.y); }
    if (FORM == 5) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.x, b.x, c.x); ehi = sfma(a.y, b.y, c.x); }
    if (FORM == 6) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.y, b.x, c.x); ehi = sfma(a.x, b.y, c.y); }
    if (FORM == 7) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); elo = smul(a.x, b.y); ehi = smul(a.y, b.x); }
    if (FORM == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.x, b.x, c.y); ehi = sfma(a.y, b.y, c.x); }
    if (FORM == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); elo = sfma(a.x, b.y, c.x); ehi = sfma(a.y, b.y, c.y); }
    if ((FORM == 1 || FORM == 7 || FORM == 9) && __float_as_uint(r.x) != __float_as_uint(elo)) {
      // what did the hardware compute instead?  the value the .lo lane would have with op_sel[1] = 0 (src1.lo instead of src1.hi), or src1's .hi of another lane?
      const float alt = FORM == 7 ? smul(a.x, b.x) : sfma(a.x, b.x, c.x);
      if (__float_as_uint(r.x) == __float_as_uint(alt)) atomicAdd(&cnt[88], 1u); else atomicAdd(&cnt[89], 1u);
      if (atomicAdd(&cnt[90], 1u) < 8u) { const unsigned k = atomicAdd(&cnt[91], 1u); if (k < 8u) { float* ex = (float*)(cnt + 92) + k * 6; ex[0] = r.x; ex[1] = elo; ex[2] = alt; ex[3] = a.x; ex[4] = b.x; ex[5] = b.y; } }
    }
    blo += __float_as_uint(r.x) != __float_as_uint(elo);
    bhi += __float_as_uint(r.y) != __float_as_uint(ehi);
    a.x = a.x * 0.999f + 0.01f; a.y = a.y * 1.001f - 0.01f; b.x = b.x + 0.001f; b.y = b.y - 0.001f;
  }
  if (blo) atomicAdd(&cnt[(FORM * 2 + 0) * 4 + (lane >> 4)], blo);
  if (bhi) atomicAdd(&cnt[(FORM * 2 + 1) * 4 + (lane >> 4)], bhi);
}

// FORM 6: the packed instruction consumes registers that a global load has just delivered (what the interpolation kernel does): four 16-byte loads of
// bf16 pairs, unpacked with shift / and, multiplied pairwise by a weight pair with packed instructions; the expected values are rebuilt from the index
// with integer arithmetic (never loaded), unpacked the same way and multiplied with scalar instructions.
__host__ __device__ __forceinline__ unsigned word_of(unsigned k) { return (0x3f80u + ((k * 2654435761u) >> 25)) << 16 | (0x3f80u + ((k * 40503u + 7u) & 127u)); }

__global__ __launch_bounds__(256) void pk_load_check_kernel(const uint4* __restrict__ src, unsigned n, unsigned* cnt) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  if (i + 3 >= n) return;
  const float w0 = 0.25f + 0.001f * (float)(i & 255), w1 = 1.f - w0;
  uint4 v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = src[i + t];
  unsigned blo = 0, bhi = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned wd[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x2 pr = {__uint_as_float(wd[k] << 16), __uint_as_float(wd[k] & 0xffff0000u)};     // (even, odd) bf16 element of the word
      f32x2 r;
      const f32x2 wp = {w0, w1};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(wp), "v"(pr));      // lo = w1 * even, hi = w0 * odd
      const unsigned e = word_of((i + t) * 4 + k);
      const float elo = smul(w1, __uint_as_float(e << 16)), ehi = smul(w0, __uint_as_float(e & 0xffff0000u));
      blo += __float_as_uint(r.x) != __float_as_uint(elo);
      bhi += __float_as_uint(r.y) != __float_as_uint(ehi);
    }
  }
  if (blo) atomicAdd(&cnt[(10 * 2 + 0) * 4 + (lane >> 4)], blo);
  if (bhi) atomicAdd(&cnt[(10 * 2 + 1) * 4 + (lane >> 4)], bhi);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200, S = argc > 2 ? atoi(argv[2]) : 6, neigh = argc > 3 ? atoi(argv[3]) : 0;
  const int grid = argc > 4 ? atoi(argv[4]) : 1024, iters = argc > 5 ? atoi(argv[5]) : 200;     // checking kernel: workgroups, loop length (short waves: 6400, 3)
  unsigned* cnt; CK(hipMalloc(&cnt, (92 + 48) * 4)); CK(hipMemset(cnt, 0, (92 + 48) * 4));
  const unsigned nsrc = 6400u * 256u + 4u;
  uint4* src; CK(hipMalloc(&src, (size_t)nsrc * 16));
  { std::vector<unsigned> hw((size_t)nsrc * 4); for (size_t k = 0; k < hw.size(); ++k) hw[k] = word_of((unsigned)k); CK(hipMemcpy(src, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); }
  std::vector<hipStream_t> st(S); std::vector<void*> mb(S);
  for (int j = 0; j < S; ++j) { CK(hipStreamCreate(&st[j])); CK(hipMalloc(&mb[j], (size_t)256 * 256 * 8)); }
  for (int r = 0; r < rounds; ++r) {
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < S; ++j) {
        if (neigh == 0) mfma_plain_kernel<<<256, 256, 0, st[j]>>>((bf16*)mb[j], 300);
        if ((r + j) % 2 == 0) { pk_load_check_kernel<<<6400, 256, 0, st[j]>>>(src, nsrc, cnt); continue; }
        switch ((r * 3 + k + j) % 10) {
          case 0: pk_check_kernel<0><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 1: pk_check_kernel<1><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 2: pk_check_kernel<2><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 3: pk_check_kernel<3><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 4: pk_check_kernel<4><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 5: pk_check_kernel<5><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 6: pk_check_kernel<6><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 7: pk_check_kernel<7><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          case 8: pk_check_kernel<8><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
          default: pk_check_kernel<9><<<grid, 256, 0, st[j]>>>(cnt, iters); break;
        }
      }
    CK(hipDeviceSynchronize());
  }
  unsigned h[140]; CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
  const char* names[11] = {"v_pk_fma_f32", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 op_sel_hi:[1,1,0]",
                           "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]", "v_pk_fma_f32 op_sel:[0,1,0]",
                           "v_pk_mul_f32 on freshly loaded registers"};
  printf("%d streams, neighbour %s, %d rounds; wrong results by half and lane quarter (0-15, 16-31, 32-47, 48-63):\n", S, neigh == 0 ? "MFMA loop" : "none", rounds);
  for (int f = 0; f < 11; ++f)
    printf("  %-48s .lo %u %u %u %u   .hi %u %u %u %u\n", names[f], h[(f * 2) * 4], h[(f * 2) * 4 + 1], h[(f * 2) * 4 + 2], h[(f * 2) * 4 + 3], h[(f * 2 + 1) * 4], h[(f * 2 + 1) * 4 + 1],
           h[(f * 2 + 1) * 4 + 2], h[(f * 2 + 1) * 4 + 3]);
  printf("  of the wrong .lo results of the src1-swapped forms: %u equal the result with op_sel[1] = 0 (src1.lo used instead of src1.hi), %u are something else\n", h[88], h[89]);
  for (unsigned k = 0; k < 8 && k < h[91]; ++k) { const float* ex = (const float*)(h + 92) + k * 6; printf("    got %.9g expected %.9g with-src1.lo %.9g   (src0.lo %.9g src1.lo %.9g src1.hi %.9g)\n", ex[0], ex[1], ex[2], ex[3], ex[4], ex[5]); }
  return 0;
}
