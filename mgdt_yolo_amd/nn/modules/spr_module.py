"""SPRModule (reference nn/modules/spr_module.py:8-31): pooling-attention MLP; HIP via ops.spr_attention."""
import torch.nn as nn

from ... import ops

__all__ = ('SPRModule',)


class SPRModule(nn.Module):
    def __init__(self, channels, reduction=4):
        super().__init__()
        self.avg_pool1 = nn.AdaptiveAvgPool2d(1)   # kept for module-tree parity; compute is in the fused kernel
        self.avg_pool2 = nn.AdaptiveAvgPool2d(2)
        self.fc1 = nn.Conv2d(channels * 5, channels // reduction, kernel_size=1, padding=0)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, padding=0)
        self.sigmoid = nn.Sigmoid()

    def group_attention(self, x, groups):
        """softmax over `groups` of SPR(x_group) for all groups at once -> fp32 [B, C] (block.py:268-278)."""
        return ops.spr_attention(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, groups)

    def forward(self, x):
        """sigmoid weights (B, C, 1, 1) of a single group - the reference's standalone call form."""
        raise RuntimeError('SPRModule is evaluated inside MSPA_C2f by the fused pooling-attention kernels '
                           '(use group_attention); the standalone sigmoid form is not on the hot path')
