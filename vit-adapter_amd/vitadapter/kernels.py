"""Op dispatch of the backbone's hot operators onto libvitadapter_hip.so.

Every function here takes and returns torch tensors and enqueues work on torch's current
stream.  Importing this module loads the HIP library (``_vah``); if it is missing the import
fails -- there is no CPU fallback behind these entry points.
"""
import torch

import _vah  # noqa: F401  (hard requirement: raises ImportError when the .so is not built)


def attention(qkv, scale, dropout_p=0.):
    """Softmax attention on a packed projection.

    qkv: (B, N, 3, heads, head_dim) view of the fused qkv Linear output (any strides).
    Returns (B, N, heads, head_dim).  Arithmetic of the reference's Attention / WindowedAttention
    (/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:83-88,154-159):
    softmax(q k^T * scale) v, with dropout on the probabilities in training.
    """
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)          # each (B, heads, N, hd)
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    if dropout_p > 0.:
        attn = torch.nn.functional.dropout(attn, dropout_p, True)
    return (attn @ v).transpose(1, 2)
