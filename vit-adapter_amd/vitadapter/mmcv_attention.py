"""mmcv-signature ``MultiScaleDeformableAttention`` on the gfx950 MSDA kernels.

Second consumer of the gather kernel in the reference: ``MSDeformAttnPixelDecoder`` builds its
6 encoder layers from ``attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256,
num_heads=8, num_levels=3, num_points=4, im2col_step=64, dropout=0.0, batch_first=False,
norm_cfg=None, init_cfg=None)`` (ref: segmentation/mmseg_custom/models/plugins/
msdeformattn_pixel_decoder.py:52-62, configs/_base_/models/mask2former_beit.py:41-51) and calls them
through mmcv's BaseTransformerLayer with ``query`` (Lq, N, E), ``query_pos``, ``key_padding_mask``,
``reference_points`` (N, Lq, L, 2), ``spatial_shapes``, ``level_start_index``
(msdeformattn_pixel_decoder.py:230-242); ``init_weights()`` is called explicitly (:154-158).

The class itself lives in mmcv (``mmcv.ops.multi_scale_deform_attn``), which is not part of the
reference tree: its behaviour is restated here from the call sites above and mmcv 1.4's published
module (value = query when no value is given, ``query + query_pos``, (Lq, N, E) <-> (N, Lq, E)
unless ``batch_first``, zero-filled padded values, identity residual after dropout).  PARITY
UNPINNED against mmcv; the tests hold it to this repo's MSDA oracle composed with the same Linear
layers.  Parameter names (state_dict keys) are mmcv's: ``sampling_offsets``, ``attention_weights``,
``value_proj``, ``output_proj`` - the same four as ``ops.modules.MSDeformAttn``, whose forward
(fused softmax + location arithmetic + gather kernel) this class reuses.
"""
import warnings

from ops.modules import MSDeformAttn
from torch import nn


class MultiScaleDeformableAttention(MSDeformAttn):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        if embed_dims % num_heads != 0:
            raise ValueError('embed_dims must be divisible by num_heads, but got {} and {}'
                             .format(embed_dims, num_heads))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            super().__init__(d_model=embed_dims, n_levels=num_levels, n_heads=num_heads, n_points=num_points,
                             ratio=1.0)
        self.im2col_step = im2col_step
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.norm_cfg, self.init_cfg = norm_cfg, init_cfg
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)

    def init_weights(self):
        """mmcv's initial values = the reference op's (zero offset / weight matrices, ring-direction
        offset bias scaled by the point index, Xavier value / output projections)."""
        self._reset_parameters()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:                 # (Lq, N, E) -> (N, Lq, E)
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        out = super().forward(query, reference_points, value, spatial_shapes, level_start_index,
                              key_padding_mask)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


def register_attention(registry=None, force=True):
    """Register under mmcv's name so ``attn_cfgs=dict(type='MultiScaleDeformableAttention', ...)``
    of the reference configs builds this class.  ``registry`` defaults to mmcv's ATTENTION."""
    if registry is None:
        from mmcv.cnn.bricks.registry import ATTENTION as registry
    registry.register_module(name='MultiScaleDeformableAttention', force=force,
                             module=MultiScaleDeformableAttention)
    return MultiScaleDeformableAttention
