"""mmcv-signature MultiScaleDeformableAttention shim.  mmcv's source is not in the
reference tree, so parity with mmcv itself is UNPINNED; these tests hold the shim to this repo's
MSDA oracle composed with the same four Linear layers, through the call signature the reference's
MSDeformAttnPixelDecoder uses (msdeformattn_pixel_decoder.py:230-242)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import msda as oracle_msda


def _inputs(dev, N=2, E=64, shapes=((8, 12), (4, 6), (2, 3)), seed=0):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    query = torch.randn(S, N, E, generator=g)
    pos = torch.randn(S, N, E, generator=g) * 0.3
    ref = torch.rand(N, S, len(shapes), 2, generator=g)
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, -5:] = True
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    return [t.to(dev) for t in (query, pos, ref, mask, ss, lsi)]


def _expected(m, query, pos, ref, mask, ss, lsi):
    """The module's arithmetic, written out with the oracle's torch core (fp32 / fp64 agnostic)."""
    M, L, P = m.num_heads, m.num_levels, m.num_points
    q = (query + pos).permute(1, 0, 2)
    v = query.permute(1, 0, 2)
    N, Lq, E = q.shape
    value = F.linear(v, m.value_proj.weight, m.value_proj.bias).masked_fill(mask[..., None], 0.0)
    value = value.view(N, -1, M, E // M)
    off = F.linear(q, m.sampling_offsets.weight, m.sampling_offsets.bias).view(N, Lq, M, L, P, 2)
    w = F.linear(q, m.attention_weights.weight, m.attention_weights.bias).view(N, Lq, M, L * P)
    w = w.softmax(-1).view(N, Lq, M, L, P)
    norm = torch.stack([ss[..., 1], ss[..., 0]], -1).to(q.dtype)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = oracle_msda.core_torch(value, [tuple(x) for x in ss.tolist()], loc, w)
    out = F.linear(out, m.output_proj.weight, m.output_proj.bias).permute(1, 0, 2)
    return out + query


def _module(dev, E=64, seed=1):
    from vitadapter.mmcv_attention import MultiScaleDeformableAttention
    torch.manual_seed(seed)
    m = MultiScaleDeformableAttention(embed_dims=E, num_heads=4, num_levels=3, num_points=4, dropout=0.0,
                                      batch_first=False)
    m.init_weights()
    with torch.no_grad():                       # away from the all-zero initial offsets / weights
        m.sampling_offsets.weight.normal_(0, 0.05)
        m.attention_weights.weight.normal_(0, 0.2)
    return m.to(dev)


def test_constructor_and_state_dict_follow_mmcv():
    from vitadapter.mmcv_attention import MultiScaleDeformableAttention
    m = MultiScaleDeformableAttention(embed_dims=256, num_heads=8, num_levels=3, num_points=4, im2col_step=64,
                                      dropout=0.0, batch_first=False, norm_cfg=None, init_cfg=None)
    assert sorted(m.state_dict()) == sorted(
        '%s.%s' % (a, b) for a in ('sampling_offsets', 'attention_weights', 'value_proj', 'output_proj')
        for b in ('weight', 'bias'))
    assert m.sampling_offsets.weight.shape == (8 * 3 * 4 * 2, 256) and m.attention_weights.weight.shape == (96, 256)
    m.init_weights()
    assert float(m.sampling_offsets.weight.detach().abs().max()) == 0
    assert float(m.attention_weights.bias.detach().abs().max()) == 0
    b = m.sampling_offsets.bias.detach().view(8, 3, 4, 2)
    assert torch.allclose(b[:, :, 3], 4 * b[:, :, 0]) and torch.allclose(b[0, 0, 0], torch.tensor([1., 0.]))
    with pytest.raises(ValueError):
        MultiScaleDeformableAttention(embed_dims=250, num_heads=8)


def test_host_logic_cpu(monkeypatch):
    """CPU tier: the gather itself is patched with the oracle (the product has no CPU kernel)."""
    import ops.modules.ms_deform_attn as mod

    class _OracleFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return oracle_msda.core_torch(value, shapes, loc, attn)
    monkeypatch.setattr(mod, 'MSDeformAttnFunction', _OracleFunction)
    m = _module('cpu')
    args = _inputs('cpu')
    query, pos, ref, mask, ss, lsi = args
    out = m(query=query, key=None, value=None, query_pos=pos, key_padding_mask=mask, reference_points=ref,
            spatial_shapes=ss, level_start_index=lsi)
    want = _expected(m, *args)
    assert out.shape == query.shape and (out - want).abs().max().item() <= 1e-5
    mb = _module('cpu')
    mb.batch_first = True
    out_b = mb(query=query.permute(1, 0, 2), query_pos=pos.permute(1, 0, 2), key_padding_mask=mask,
               reference_points=ref, spatial_shapes=ss, level_start_index=lsi)
    assert (out_b.permute(1, 0, 2) - want).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('shared_grid', [False, True])
def test_matches_oracle_on_gpu(shared_grid):
    """HIP kernels through the mmcv signature: outputs and every gradient vs the fp32 oracle
    composition on the CPU.  shared_grid: reference points of shape (1, Lq, L, 2) take the fused
    kernels, the pixel decoder's repeated (N, Lq, L, 2) grid the unfused function."""
    m = _module('cuda')
    args = _inputs('cuda')
    if shared_grid:
        args[2] = args[2][:1]
    query, pos, ref, mask, ss, lsi = args
    query.requires_grad_(True)
    out = m(query=query, query_pos=pos, key_padding_mask=mask, reference_points=ref, spatial_shapes=ss,
            level_start_index=lsi)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()
    out.backward(gout)
    m_cpu = _module('cpu')
    m_cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    cargs = [a.detach().cpu() for a in args]
    cargs[0].requires_grad_(True)
    if shared_grid:
        cargs[2] = cargs[2].expand(cargs[0].shape[1], -1, -1, -1)
    want = _expected(m_cpu, *cargs)
    want.backward(gout.cpu())
    assert (out.detach().cpu() - want.detach()).abs().max().item() <= 1e-4
    assert (query.grad.cpu() - cargs[0].grad).abs().max().item() <= 1e-4 * max(1.0, cargs[0].grad.abs().max().item())
    for (k, p), (_, pc) in zip(m.named_parameters(), m_cpu.named_parameters()):
        assert (p.grad.cpu() - pc.grad).abs().max().item() <= 2e-4 * max(1.0, pc.grad.abs().max().item()), k
