"""vitadapter -- MI355X-native ViT-Adapter backbone (host side).

``vitadapter.backbones.ViTAdapter`` has the constructor signature, forward contract and
state_dict layout of the reference's two ``ViTAdapter`` classes
(/root/reference/segmentation/mmseg_custom/models/backbones/vit_adapter.py:19-137 and
/root/reference/detection/mmdet_custom/models/backbones/vit_adapter.py:19-132); the deformable
attention inside it runs on the hand-written gfx950 kernels of libvitadapter_hip.so.
"""
from .backbones import (BEiTAdapter, ViTAdapter, ViTAdapterDet, ViTAdapterSeg, register_backbones,
                        register_beit_adapter)

from .mmcv_attention import MultiScaleDeformableAttention, register_attention

__all__ = ['ViTAdapter', 'ViTAdapterSeg', 'ViTAdapterDet', 'register_backbones', 'BEiTAdapter',
           'register_beit_adapter',
           'MultiScaleDeformableAttention', 'register_attention']
