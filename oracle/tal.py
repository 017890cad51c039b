"""Label-assignment oracle: TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates HeuristicPositiveSampleAssigner_v1 -> TaskAlignedAssigner as wired by the reference's
v8DetectionLoss (yolo/utils/loss.py:125-126 -> yolo/utils/tal.py:56-142,144-353): topk=10,
alpha schedule 0.5*(100-coff)/100 with coff = call_count // 161, beta = 8.0.

Tie rule (documented divergence): torch.topk's choice among EQUAL metrics is implementation
defined; the restatement takes the lowest anchor index first.  Fixtures are chosen so that ties do
not reach the outputs (checked at generation time against the reference).
Empty-label batches: the reference raises AttributeError (tal.py:102-108, attribute commented out at
:70); the restatement returns all-background targets (upstream behaviour, tal.py:209-213).
"""
import torch

from .boxes import ciou_xyxy

TOPK = 10
BETA = 8.0
EPOCH_NUM = 161   # tal.py:74
MAX_EPOCHS = 100  # tal.py:168
EPS = 1e-9


def assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, call_count, nc):
    """pd_scores (B,A,nc) sigmoid-ed, pd_bboxes (B,A,4) xyxy px, anc_points (A,2) px,
    gt_labels (B,N,1), gt_bboxes (B,N,4) xyxy px, mask_gt (B,N,1) float.

    Returns target_labels (B,A) int64, target_bboxes (B,A,4), target_scores (B,A,nc), fg_mask (B,A) bool,
    target_gt_idx (B,A) int64.
    """
    B, A, _ = pd_scores.shape
    N = gt_bboxes.shape[1]
    if N == 0:
        return (torch.full((B, A), nc, dtype=torch.int64), torch.zeros_like(pd_bboxes),
                torch.zeros_like(pd_scores), torch.zeros((B, A), dtype=torch.bool),
                torch.zeros((B, A), dtype=torch.int64))
    coff = call_count // EPOCH_NUM                                   # tal.py:110
    alpha = 0.5 * (MAX_EPOCHS - coff) / MAX_EPOCHS                   # tal.py:266-267

    # select_candidates_in_gts, tal.py:12-26
    lt, rb = gt_bboxes[..., None, :2], gt_bboxes[..., None, 2:]     # (B,N,1,2)
    deltas = torch.cat((anc_points[None, None] - lt, rb - anc_points[None, None]), -1)   # (B,N,A,4)
    in_gts = (deltas.amin(-1) > EPS).to(pd_scores.dtype)            # (B,N,A)

    # get_box_metrics, tal.py:245-271
    m = (in_gts * mask_gt).bool()
    lab = gt_labels.squeeze(-1).long()                              # (B,N)
    sc = pd_scores.permute(0, 2, 1)                                 # (B,nc,A)
    bbox_scores = torch.where(m, torch.gather(sc, 1, lab.clamp(0, nc - 1)[..., None].expand(-1, -1, A)),
                              torch.zeros((), dtype=pd_scores.dtype))
    iou = ciou_xyxy(gt_bboxes[:, :, None, :].expand(-1, -1, A, -1), pd_bboxes[:, None].expand(-1, N, -1, -1))
    overlaps = torch.where(m, iou.squeeze(-1).clamp(min=0), torch.zeros((), dtype=pd_bboxes.dtype))
    align = bbox_scores.pow(alpha) * overlaps.pow(BETA)

    # select_topk_candidates, tal.py:273-308 (stable: lowest index wins ties)
    order = torch.sort(align, dim=-1, descending=True, stable=True)[1][..., :TOPK]      # (B,N,K)
    valid = mask_gt.bool().expand(-1, -1, TOPK)
    order = torch.where(valid, order, torch.zeros_like(order))
    count = torch.zeros((B, N, A), dtype=torch.int32)
    count.scatter_add_(-1, order, torch.ones_like(order, dtype=torch.int32))
    count[count > 1] = 0
    mask_pos = count.to(pd_scores.dtype) * in_gts * mask_gt         # tal.py:240

    # select_highest_overlaps fed with align_metric (fork change), tal.py:29-54,222
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg[:, None] > 1).expand(-1, N, -1)
        best = align.argmax(1)                                      # first maximal index
        is_max = torch.zeros_like(mask_pos).scatter_(1, best[:, None], 1.0)
        mask_pos = torch.where(multi, is_max, mask_pos)
        fg = mask_pos.sum(-2)
    gt_idx = mask_pos.argmax(-2)                                    # (B,A)

    # get_targets, tal.py:310-353
    flat = gt_idx + torch.arange(B)[:, None] * N
    t_labels = gt_labels.long().flatten()[flat].clamp(min=0)
    t_bboxes = gt_bboxes.reshape(-1, 4)[flat]
    t_scores = torch.zeros((B, A, nc), dtype=torch.int64).scatter_(2, t_labels[..., None], 1)
    t_scores = torch.where(fg[..., None] > 0, t_scores, torch.zeros((), dtype=torch.int64))

    # normalise, tal.py:226-231
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_ov / (pos_align + EPS)).amax(-2)[..., None]
    return t_labels, t_bboxes, t_scores * norm, fg.bool(), gt_idx
