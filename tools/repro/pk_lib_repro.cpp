// No torch: the LIBRARY's bilinear kernel (mgdt_bilinear_fwd from a build that keeps packed-fp32 instructions in pointwise.o) beside a plain MFMA kernel
// (libsynth.so), as hipGraph instances replayed concurrently.   hipcc -O2 -I../../include -o pk_lib_repro pk_lib_repro.cpp -ldl
//   ./pk_lib_repro <path to libmgdt_hip*.so> <path to libsynth.so> [rounds]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mgdt.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int (*bilinear_t)(const mgdt_view*, const mgdt_view*, int, mgdt_stream);
typedef int (*synth_t)(int, void*, int, int, const void*, unsigned, void*);

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  void* hl = dlopen(argv[1], RTLD_NOW); void* hs = dlopen(argv[2], RTLD_NOW);
  if (!hl || !hs) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  bilinear_t bil = (bilinear_t)dlsym(hl, "mgdt_bilinear_fwd"); synth_t syn = (synth_t)dlsym(hs, "synth_launch");
  const int rounds = argc > 3 ? atoi(argv[3]) : 40, S = 5;
  const int N = 32, H = 20, W = 20, C = 128, Ho = 40, Wo = 40;
  const size_t nin = (size_t)N * H * W * C, nout = (size_t)N * Ho * Wo * C;
  std::vector<unsigned short> hx(nin), ref(nout), got(nout);
  unsigned sd = 12345u;
  for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; const float f = ((float)(sd >> 8) / 16777216.f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  void* dx; CK(hipMalloc(&dx, nin * 2)); CK(hipMemcpy(dx, hx.data(), nin * 2, hipMemcpyHostToDevice));
  auto view = [&](void* p, int h, int w) { mgdt_view v; v.p = p; v.n = N; v.h = h; v.w = w; v.c = C; v.sn = (int64_t)h * w * C; v.sh = (int64_t)w * C; v.sw = C; v.sc = 1; return v; };
  std::vector<hipStream_t> st(S); std::vector<hipGraphExec_t> ge(S); std::vector<void*> outb(S), mb(S);
  const mgdt_view xv = view(dx, H, W);
  for (int j = 0; j < S; ++j) {
    CK(hipStreamCreate(&st[j])); CK(hipMalloc(&outb[j], nout * 2)); CK(hipMalloc(&mb[j], (size_t)1024 * 256 * 8));
  }
  { const mgdt_view yv = view(outb[0], Ho, Wo); if (bil(&xv, &yv, MGDT_BF16, st[0])) return 3; CK(hipStreamSynchronize(st[0])); CK(hipMemcpy(ref.data(), outb[0], nout * 2, hipMemcpyDeviceToHost)); }
  for (int j = 0; j < S; ++j) {
    hipGraph_t g;
    const mgdt_view yv = view(outb[j], Ho, Wo);
    CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeThreadLocal));
    if (syn(0, mb[j], 256, 300, dx, (unsigned)(nin * 2), st[j])) return 4;
    if (bil(&xv, &yv, MGDT_BF16, st[j])) return 3;
    CK(hipStreamEndCapture(st[j], &g));
    CK(hipGraphInstantiate(&ge[j], g, nullptr, nullptr, 0));
  }
  long bad = 0, nel = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < S; ++j) CK(hipGraphLaunch(ge[j], st[j]));
    CK(hipDeviceSynchronize());
    for (int j = 0; j < S; ++j) {
      CK(hipMemcpy(got.data(), outb[j], nout * 2, hipMemcpyDeviceToHost));
      long b = 0;
      for (size_t i = 0; i < nout; ++i) b += got[i] != ref[i];
      bad += b != 0; nel += b;
    }
  }
  printf("library bilinear beside a plain MFMA kernel, %d graph instances: %ld wrong outputs of %d (%ld elements)\n", S, bad, rounds * S, nel);
  return 0;
}
