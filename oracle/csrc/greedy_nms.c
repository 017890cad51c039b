/* TEST INFRASTRUCTURE (see oracle/__init__.py) - the CPU checker / CPU baseline, never the product path.
 *
 * greedy_nms: restates torchvision.ops.nms's published CPU kernel (the one call the reference's non_max_suppression makes into a
 * third-party library, /root/reference/yolo/utils/ops.py:249; torchvision is neither vendored nor installed: PARITY UNPINNED at that call).
 * Same arithmetic as oracle/nms.py:greedy_nms (float32, no FMA contraction, strict `>`), written in C because the reference's CPU path runs
 * this step in compiled code too - the numpy loop would make the timed CPU baseline NMS-bound, which the reference is not.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see __graft_entry__.build / oracle/Makefile). */
#include <stdint.h>
#include <stdlib.h>

/* boxes: n x 4 float32 xyxy, ALREADY in descending-score order.  keep: out, positions kept (ascending).  Stops after `limit` keepers
 * (limit <= 0: no limit) - the caller slices [:max_det] anyway (ops.py:250) and greedy order makes the prefix identical. */
int64_t oracle_greedy_nms(const float* boxes, int64_t n, float thr, int64_t limit, int64_t* keep) {
  if (n <= 0) return 0;
  unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
  float* area = (float*)malloc(sizeof(float) * (size_t)n);
  if (!sup || !area) { free(sup); free(area); return -1; }
  for (int64_t i = 0; i < n; ++i) area[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (sup[i]) continue;
    keep[nk++] = i;
    if (limit > 0 && nk >= limit) break;
    const float x1 = boxes[4 * i], y1 = boxes[4 * i + 1], x2 = boxes[4 * i + 2], y2 = boxes[4 * i + 3], ai = area[i];
    for (int64_t j = i + 1; j < n; ++j) {
      if (sup[j]) continue;
      float xx1 = x1 > boxes[4 * j] ? x1 : boxes[4 * j];
      float yy1 = y1 > boxes[4 * j + 1] ? y1 : boxes[4 * j + 1];
      float xx2 = x2 < boxes[4 * j + 2] ? x2 : boxes[4 * j + 2];
      float yy2 = y2 < boxes[4 * j + 3] ? y2 : boxes[4 * j + 3];
      float w = xx2 - xx1, h = yy2 - yy1;
      w = w > 0.f ? w : 0.f;
      h = h > 0.f ? h : 0.f;
      float inter = w * h;
      float ovr = inter / (ai + area[j] - inter);
      if (ovr > thr) sup[j] = 1;
    }
  }
  free(sup);
  free(area);
  return nk;
}
