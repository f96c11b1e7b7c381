"""GPU: bf16 MFMA attention (csrc/attn_*.hip) against an fp32 PyTorch statement of the
reference's Attention arithmetic (detection/mmdet_custom/models/backbones/base/vit.py:83-88):
softmax(q k^T * scale) v.  Tolerance is bf16's: inputs and outputs carry 8 significant bits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, scale):
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4).unbind(0)
    a = ((q @ k.transpose(-2, -1)) * scale).softmax(-1)
    return (a @ v).transpose(1, 2)


@pytest.mark.parametrize('B,N,H', [(1, 64, 1), (2, 196, 3), (1, 128, 2), (2, 333, 2), (1, 1024, 4),
                                   (3, 1, 1), (1, 65, 1), (2, 4096, 2)])
def test_attention_forward_backward(B, N, H):
    from vitadapter import kernels
    torch.manual_seed(N + H)
    qkv = (torch.randn(B, N, 3, H, 64, device='cuda') * 1.5).to(torch.bfloat16).requires_grad_(True)
    scale = 64 ** -0.5
    out = kernels.attention(qkv, scale)
    assert out.dtype == torch.bfloat16 and out.shape == (B, N, H, 64)
    qr = qkv.detach().clone().requires_grad_(True)
    ref = _ref(qr, scale)
    err = (out.float() - ref).abs().max().item()
    assert err <= 3e-2 * max(1.0, ref.abs().max().item()), err
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(g)
    gerr = (qkv.grad.float() - qr.grad.float()).abs().max().item()
    assert gerr <= 6e-2 * max(1.0, qr.grad.float().abs().max().item()), gerr


def test_attention_identity_structure():
    """Exact-data check of the fragment maps: with one-hot probabilities the output must pick
    the right value row, for asymmetric V (catches transposed / permuted k orders)."""
    from vitadapter import kernels
    B, N, H = 1, 200, 2
    q = torch.zeros(B, N, H, 64, device='cuda')
    k = torch.zeros(B, N, H, 64, device='cuda')
    idx = torch.arange(N, device='cuda')
    tgt = (idx * 7 + 3) % N                       # query i attends (almost) only to key tgt[i]
    code = torch.randn(N, 64, device='cuda').sign()
    k[0, :, :, :] = code[:, None, :]
    q[0, :, :, :] = code[tgt][:, None, :] * 4.0    # dot = 256 for the target, ~0 otherwise
    v = torch.arange(N * 64, device='cuda', dtype=torch.float32).view(N, 64) % 251 - 125.0
    v = v[None, :, None, :].expand(B, N, H, 64)
    qkv = torch.stack((q, k, v), 2).to(torch.bfloat16)
    out = kernels.attention(qkv, 1.0)
    want = v[0, tgt][:, 0, :].to(torch.bfloat16).float()
    assert (out[0, :, 0].float() - want).abs().max().item() <= 1.0      # bf16 of values up to 125
    assert (out[0, :, 1].float() - want).abs().max().item() <= 1.0


def test_fp32_and_other_head_dims_use_library_gemms():
    from vitadapter import kernels
    qkv = torch.randn(1, 50, 3, 2, 32, device='cuda')
    out = kernels.attention(qkv, 32 ** -0.5)
    assert out.dtype == torch.float32
    assert (out - _ref(qkv, 32 ** -0.5)).abs().max().item() < 1e-4


@pytest.mark.parametrize('B,H,W,heads,win', [(2, 10, 17, 2, 14), (1, 14, 14, 3, 14), (2, 64, 64, 2, 14), (1, 5, 3, 1, 4),
                                                   (1, 9, 20, 1, 8), (1, 11, 10, 2, 10), (1, 13, 25, 1, 12), (1, 13, 14, 1, 13),
                                                   (1, 7, 13, 2, 6), (1, 20, 33, 1, 16)])
def test_window_attention_fused_partition(B, H, W, heads, win):
    """Windows cut by the kernels' addressing vs the reference's sequence (pad AFTER projection,
    partition, attend, merge, crop; base/vit.py:136-167) evaluated in fp32."""
    import math
    import torch.nn.functional as F
    from vitadapter import kernels
    torch.manual_seed(H * W)
    C = heads * 64
    qkv = (torch.randn(B, H * W, 3, heads, 64, device='cuda') * 1.2).to(torch.bfloat16).requires_grad_(True)
    scale = 64 ** -0.5
    out = kernels.window_attention(qkv, scale, H, W, win)
    assert out is not None and out.shape == (B, H * W, heads, 64)
    g = torch.randn(B, H * W, heads, 64, device='cuda')
    out.backward(g.to(torch.bfloat16))

    qr = qkv.detach().float().requires_grad_(True)
    Hp, Wp = math.ceil(H / win) * win, math.ceil(W / win) * win
    t = F.pad(qr.view(B, H, W, 3 * C), (0, 0, 0, Wp - W, 0, Hp - H))
    t = t.view(B, Hp // win, win, Wp // win, win, 3 * C).permute(0, 1, 3, 2, 4, 5)
    t = t.reshape(-1, win * win, 3, heads, 64)
    o = _ref(t, scale).reshape(B, Hp // win, Wp // win, win, win, C).permute(0, 1, 3, 2, 4, 5)
    ref = o.reshape(B, Hp, Wp, C)[:, :H, :W].reshape(B, H * W, heads, 64)
    ref.backward(g)
    assert (out.float() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    gerr = (qkv.grad.float() - qr.grad.view_as(qkv)).abs().max().item()
    assert gerr <= 6e-2 * max(1.0, qr.grad.abs().max().item()), gerr


@pytest.mark.parametrize('B,N,H', [(1, 64, 1), (2, 197, 3), (1, 130, 2), (2, 1601, 2), (3, 5, 1)])
def test_attention_with_bias_forward_backward(B, N, H):
    """softmax(q k^T * scale + bias) v with a (heads, N, N) bias shared by the batch - BEiT's relative position bias
    and class token (segmentation/mmseg_custom/models/backbones/base/beit.py:120-144) - against the fp32 expression on
    the same bf16 q, k, v: output, d(qkv) and d(bias).  The bias enters the kernels as bf16: tolerances as the unbiased
    test, d(bias) (a sum over the batch of dS, written per image in bf16) within 2e-2 of its largest entry."""
    from vitadapter import kernels
    torch.manual_seed(N + H)
    qkv = (torch.randn(B, N, 3, H, 64, device='cuda') * 1.5).to(torch.bfloat16).requires_grad_(True)
    bias = (torch.randn(H, N, N, device='cuda') * 1.5).requires_grad_(True)
    scale = 64 ** -0.5
    out = kernels.attention_bias(qkv, bias, scale)
    assert out is not None and out.dtype == torch.bfloat16 and out.shape == (B, N, H, 64)
    qr = qkv.detach().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    q, k, v = qr.permute(2, 0, 3, 1, 4).unbind(0)
    ref = ((((q @ k.transpose(-2, -1)) * scale) + br.unsqueeze(0)).softmax(-1) @ v).transpose(1, 2)
    assert (out.float() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(g)
    gerr = (qkv.grad.float() - qr.grad).abs().max().item()
    assert gerr <= 6e-2 * max(1.0, qr.grad.abs().max().item()), gerr
    berr = (bias.grad - br.grad).abs().max().item()
    assert berr <= 2e-2 * max(1.0, br.grad.abs().max().item()), berr


@pytest.mark.parametrize('B,hw,H', [(2, (4, 4), 2), (1, (14, 14), 3), (2, (40, 40), 2), (2, (5, 9), 1)])
def test_attention_relative_position_table(B, hw, H):
    """BEiT's form of the bias (base/beit.py:120-131): table (T, heads) indexed by relative_position_index (N, N) with the
    class token's three extra rows.  Kernels build their operands from the table and reduce d(table) from dS: against the
    fp32 expression, output / d(qkv) at the usual bf16 tolerances, d(table) within 2e-2 of its largest entry."""
    from vitadapter import kernels
    from vitadapter.backbones.beit import relative_position_index
    torch.manual_seed(hw[0] * 7 + H)
    index, T = relative_position_index(hw)
    index = index.cuda()
    N = hw[0] * hw[1] + 1
    table = (torch.randn(T, H, device='cuda') * 1.5).requires_grad_(True)
    qkv = (torch.randn(B, N, 3, H, 64, device='cuda') * 1.5).to(torch.bfloat16).requires_grad_(True)
    scale = 64 ** -0.5
    out = kernels.attention_relpos(qkv, table, index, scale)
    assert out is not None and out.shape == (B, N, H, 64)
    qr = qkv.detach().float().requires_grad_(True)
    tr = table.detach().clone().requires_grad_(True)
    bias = tr[index.view(-1)].view(N, N, H).permute(2, 0, 1)
    q, k, v = qr.permute(2, 0, 3, 1, 4).unbind(0)
    ref = ((((q @ k.transpose(-2, -1)) * scale) + bias.unsqueeze(0)).softmax(-1) @ v).transpose(1, 2)
    assert (out.float() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(g)
    assert (qkv.grad.float() - qr.grad).abs().max().item() <= 6e-2 * max(1.0, qr.grad.abs().max().item())
    terr = (table.grad - tr.grad).abs().max().item()
    assert terr <= 2e-2 * max(1.0, tr.grad.abs().max().item()), terr
