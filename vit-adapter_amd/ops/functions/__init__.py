from .ms_deform_attn_func import MSDeformAttnFunction
from .ms_deform_attn_fused import MSDeformAttnFusedFunction, fused_supported

__all__ = ['MSDeformAttnFunction', 'MSDeformAttnFusedFunction', 'fused_supported']
