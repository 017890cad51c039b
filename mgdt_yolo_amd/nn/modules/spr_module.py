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
        """sigmoid(fc2(relu(fc1(cat(avgpool1(x), flatten(avgpool2x2(x))))))) -> (B, C, 1, 1): the reference's standalone call form
        (spr_module.py:20-31), on the pooling + MLP kernels (inside MSPA_C2f the same kernels run for all four groups at once)."""
        if x.shape[1] != self.fc2.out_channels:
            raise RuntimeError(f'SPRModule({self.fc2.out_channels}) got {x.shape[1]} channels')
        w = ops.spr_attention(x if ops.is_nhwc(x) else x.contiguous(memory_format=__import__('torch').channels_last), self.fc1.weight, self.fc1.bias,
                              self.fc2.weight, self.fc2.bias, 1, softmax=False)
        return w.to(x.dtype).view(x.shape[0], x.shape[1], 1, 1)
