from .beit_adapter import BEiTAdapter, register_beit_adapter
from .beit_det import BEiTAdapter as BEiTAdapterDet
from .vit_adapter import ViTAdapter, ViTAdapterDet, ViTAdapterSeg, register_backbones

__all__ = ['ViTAdapter', 'ViTAdapterSeg', 'ViTAdapterDet', 'register_backbones', 'BEiTAdapter',
           'register_beit_adapter', 'BEiTAdapterDet']
