// Torch-free, library-free reproduction of the concurrent-execution defect (profiles/r03_graph_replay_root_cause.txt):
//   victim    = mgdt's bilinear kernel, compiled HERE from its source (pointwise.hip is included as is; with packed-fp32 instructions unless the file is
//               built with -Xclang -target-feature -Xclang -packed-fp32-ops),
//   neighbour = a plain MFMA loop (no memory, no LDS),
//   schedule  = S instances of [neighbour -> victim], each on its own stream, launched three times back to back per round, then compared with the
//               victim's output when it ran alone.  argv[1] = rounds, argv[2] = S, argv[3] = 1: hipGraph instances, 0: eager launches on the S streams; argv[4] = neighbour: 0 MFMA loop, 1 VALU FMA loop of the same length, 2 none.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../mgdt_yolo_amd/csrc -o pk_graph_repro pk_graph_repro.hip
#include "../../mgdt_yolo_amd/csrc/pointwise.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
void mgdt_set_error(const char*, ...) {}      // capi.hip's error slot: not needed here
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

// control neighbour: the same duration of plain fp32 FMAs, no MFMA
__global__ __launch_bounds__(256) void valu_plain_kernel(bf16* out, int iters) {
  float a0 = 0.001f * (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < iters * 16; ++i) { a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c); }
  bf16x4 o;
  o[0] = (bf16)a0; o[1] = (bf16)a1; o[2] = (bf16)a2; o[3] = (bf16)a3;
  *(bf16x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = o;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 100, S = argc > 2 ? atoi(argv[2]) : 5, graphs = argc > 3 ? atoi(argv[3]) : 1, neigh = argc > 4 ? atoi(argv[4]) : 0;   // neigh 0: MFMA loop, 1: VALU FMA loop, 2: none
  const int N = 32, H = 20, W = 20, C = 128, Ho = 40, Wo = 40;
  const size_t nin = (size_t)N * H * W * C, nout = (size_t)N * Ho * Wo * C;
  std::vector<unsigned short> hx(nin), ref(nout), got(nout);
  unsigned sd = 12345u;
  for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; const float f = ((float)(sd >> 8) / 16777216.f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  void* dx; CK(hipMalloc(&dx, nin * 2)); CK(hipMemcpy(dx, hx.data(), nin * 2, hipMemcpyHostToDevice));
  auto view = [&](void* p, int h, int w) { mgdt_view v; v.p = p; v.n = N; v.h = h; v.w = w; v.c = C; v.sn = (int64_t)h * w * C; v.sh = (int64_t)w * C; v.sw = C; v.sc = 1; return v; };
  std::vector<hipStream_t> st(S); std::vector<hipGraphExec_t> ge(S); std::vector<void*> outb(S), mb(S);
  const mgdt_view xv = view(dx, H, W);
  for (int j = 0; j < S; ++j) { CK(hipStreamCreate(&st[j])); CK(hipMalloc(&outb[j], nout * 2)); CK(hipMalloc(&mb[j], (size_t)256 * 256 * 8)); }
  { const mgdt_view yv = view(outb[0], Ho, Wo); if (mgdt_bilinear_fwd(&xv, &yv, MGDT_BF16, st[0])) return 3; CK(hipStreamSynchronize(st[0])); CK(hipMemcpy(ref.data(), outb[0], nout * 2, hipMemcpyDeviceToHost)); }
  auto instance = [&](int j) {
    const mgdt_view yv = view(outb[j], Ho, Wo);
    if (neigh == 0) mfma_plain_kernel<<<256, 256, 0, st[j]>>>((bf16*)mb[j], 300);
    else if (neigh == 1) valu_plain_kernel<<<256, 256, 0, st[j]>>>((bf16*)mb[j], 300);
    if (mgdt_bilinear_fwd(&xv, &yv, MGDT_BF16, st[j])) exit(3);
  };
  if (graphs)
    for (int j = 0; j < S; ++j) {
      hipGraph_t g;
      CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeThreadLocal));
      instance(j);
      CK(hipStreamEndCapture(st[j], &g));
      CK(hipGraphInstantiate(&ge[j], g, nullptr, nullptr, 0));
    }
  long bad = 0, nel = 0, odd = 0, hi = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < S; ++j) { if (graphs) CK(hipGraphLaunch(ge[j], st[j])); else instance(j); }
    CK(hipDeviceSynchronize());
    for (int j = 0; j < S; ++j) {
      CK(hipMemcpy(got.data(), outb[j], nout * 2, hipMemcpyDeviceToHost));
      long b = 0;
      for (size_t i = 0; i < nout; ++i)
        if (got[i] != ref[i]) { ++b; odd += i & 1; hi += ((i / 8) % 64) >= 48; }       // one 8-channel vector per thread: lane = (i / 8) % 64
      bad += b != 0; nel += b;
    }
  }
  printf("%s, %d instances of [%s -> bilinear]: %ld wrong outputs of %d (%ld elements; %ld odd channels, %ld in lanes 48..63)\n",
         graphs ? "hipGraph replay" : "eager launches", S, neigh == 0 ? "plain MFMA kernel" : neigh == 1 ? "plain VALU-FMA kernel" : "nothing", bad, rounds * S, nel, odd, hi);
  return 0;
}
