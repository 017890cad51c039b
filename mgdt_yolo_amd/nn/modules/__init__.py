"""Module registry (reference: nn/modules/__init__.py:27-34 `__all__`): the names `parse_model` resolves from a YAML."""
from .block import (C2f, DFL, IFM, MSPA_C2f, SPPF, Bottleneck, InjectionMultiSum_Auto_pool, SimFusion_3in, SimFusion_4in,
                    Upsample, h_sigmoid)
from .conv import Concat, Conv, DWConv
from .convnextv2 import ConvNeXtV2_Block
from .head import Conv_GN, Detect, DyDCNv2, Scale, TaskDecomposition, TOODHead
from .spr_module import SPRModule
from .utils import GRN, LayerNorm

__all__ = ('Conv', 'DWConv', 'Concat', 'DFL', 'SPPF', 'C2f', 'MSPA_C2f', 'Bottleneck', 'SPRModule', 'ConvNeXtV2_Block',
           'LayerNorm', 'GRN', 'SimFusion_4in', 'SimFusion_3in', 'IFM', 'h_sigmoid', 'InjectionMultiSum_Auto_pool',
           'Detect', 'Upsample', 'TOODHead', 'DyDCNv2', 'Conv_GN', 'TaskDecomposition', 'Scale')
