"""The 6-layer deformable encoder of Mask2Former's pixel decoder (vitadapter/pixel_decoder.py, SURVEY section 8 f-3,
BASELINE configs[4]).  mmcv / mmdet are not in the reference tree: PARITY UNPINNED against them; these tests hold the
stack to a plain PyTorch evaluation of the same arithmetic (post-norm layer, mmcv FFN, attention with its own identity)
on this repo's MSDA oracle, through the call MSDeformAttnPixelDecoder.forward makes (msdeformattn_pixel_decoder.py:230-242)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import msda as oracle_msda

SHAPES = ((4, 6), (8, 12), (16, 24))          # low to high resolution, as the pixel decoder orders them


def _stack(dev, layers=2, E=64, heads=2, seed=1):
    from vitadapter.pixel_decoder import MSDeformAttnEncoder
    torch.manual_seed(seed)
    m = MSDeformAttnEncoder(num_layers=layers, embed_dims=E, num_heads=heads, num_levels=3, num_points=4, feedforward_channels=4 * E)
    with torch.no_grad():                       # away from the all-zero initial offsets / weights
        for layer in m.layers:
            layer.attentions[0].sampling_offsets.weight.normal_(0, 0.05)
            layer.attentions[0].attention_weights.weight.normal_(0, 0.2)
            for n in layer.norms:
                n.weight.normal_(1, 0.1)
                n.bias.normal_(0, 0.1)
    return m.to(dev)


def _expected(m, query, pos, ref, ss):
    shapes = [tuple(x) for x in ss.tolist()]
    for layer in m.layers:
        a = layer.attentions[0]
        M, L, P = a.num_heads, a.num_levels, a.num_points
        q = (query + pos).permute(1, 0, 2)
        v = query.permute(1, 0, 2)
        N, Lq, E = q.shape
        value = F.linear(v, a.value_proj.weight, a.value_proj.bias).view(N, -1, M, E // M)
        off = F.linear(q, a.sampling_offsets.weight, a.sampling_offsets.bias).view(N, Lq, M, L, P, 2)
        w = F.linear(q, a.attention_weights.weight, a.attention_weights.bias).view(N, Lq, M, L * P).softmax(-1).view(N, Lq, M, L, P)
        norm = torch.stack([ss[..., 1], ss[..., 0]], -1).to(q.dtype)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        out = oracle_msda.core_torch(value, shapes, loc, w)
        x = F.linear(out, a.output_proj.weight, a.output_proj.bias).permute(1, 0, 2) + query
        x = F.layer_norm(x, (E,), layer.norms[0].weight, layer.norms[0].bias, layer.norms[0].eps)
        f = layer.ffns[0]
        h = F.linear(torch.relu(F.linear(x, f.layers[0][0].weight, f.layers[0][0].bias)), f.layers[1].weight, f.layers[1].bias)
        query = F.layer_norm(x + h, (E,), layer.norms[1].weight, layer.norms[1].bias, layer.norms[1].eps)
    return query


def test_state_dict_follows_mmcv_layout():
    from vitadapter.pixel_decoder import MSDeformAttnEncoder
    m = MSDeformAttnEncoder()
    keys = set(m.state_dict())
    assert len(m.layers) == 6
    for k in ('layers.0.attentions.0.sampling_offsets.weight', 'layers.5.attentions.0.output_proj.bias',
              'layers.0.ffns.0.layers.0.0.weight', 'layers.0.ffns.0.layers.1.bias', 'layers.3.norms.0.weight', 'layers.3.norms.1.bias'):
        assert k in keys, k
    a = m.layers[0].attentions[0]
    assert a.sampling_offsets.weight.shape == (8 * 3 * 4 * 2, 256) and m.layers[0].ffns[0].layers[0][0].weight.shape == (1024, 256)
    assert float(a.sampling_offsets.weight.detach().abs().max()) == 0          # the attention's own init runs last (:154-158)


def test_stack_host_logic_cpu(monkeypatch):
    """CPU tier: the gather itself is patched with the oracle (the product has no CPU kernel)."""
    import ops.modules.ms_deform_attn as mod
    from vitadapter.pixel_decoder import encoder_inputs

    class _OracleFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return oracle_msda.core_torch(value, shapes, loc, attn)
    monkeypatch.setattr(mod, 'MSDeformAttnFunction', _OracleFunction)
    m = _stack('cpu')
    query, pos, ref, ss, lsi = encoder_inputs(SHAPES, 2, 64, 'cpu', seed=3)
    out = m(query=query, key=None, value=None, query_pos=pos, query_key_padding_mask=None, spatial_shapes=ss,
            reference_points=ref, level_start_index=lsi)
    want = _expected(m, query, pos, ref, ss)
    assert out.shape == query.shape and (out - want).abs().max().item() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 2])
def test_stack_matches_oracle_on_gpu(batch):
    """HIP kernels under the stack: outputs and every parameter gradient in fp32 against the oracle composition on the CPU
    (batch 1 = configs[4]'s per-GPU batch: the repeated reference grid is shared, the fused core runs; batch 2: the
    plain MSDeformAttnFunction), then the bf16-autocast tier bench.py times, against the fp32 run."""
    from vitadapter.pixel_decoder import encoder_inputs
    m = _stack('cuda')
    query, pos, ref, ss, lsi = encoder_inputs(SHAPES, batch, 64, 'cuda', seed=3)
    query.requires_grad_(True)
    out = m(query=query, query_pos=pos, spatial_shapes=ss, reference_points=ref, level_start_index=lsi)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()
    out.backward(gout)
    m_cpu = _stack('cpu')
    m_cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    cq = query.detach().cpu().requires_grad_(True)
    want = _expected(m_cpu, cq, pos.cpu(), ref.cpu(), ss.cpu())
    want.backward(gout.cpu())
    assert (out.detach().cpu() - want.detach()).abs().max().item() <= 2e-4
    assert (query.grad.cpu() - cq.grad).abs().max().item() <= 2e-4 * max(1.0, cq.grad.abs().max().item())
    g32 = {}
    for (k, p), (_, pc) in zip(m.named_parameters(), m_cpu.named_parameters()):
        assert (p.grad.cpu() - pc.grad).abs().max().item() <= 3e-4 * max(1.0, pc.grad.abs().max().item()), k
        g32[k] = p.grad.clone()
    m.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out16 = m(query=query.detach(), query_pos=pos, spatial_shapes=ss, reference_points=ref, level_start_index=lsi)
    out16.float().backward(gout)
    rel = float((out16.float() - out.detach()).norm() / out.detach().norm())
    assert rel <= 3e-2, rel
    rels = sorted(float((p.grad - g32[k]).norm() / g32[k].norm()) for k, p in m.named_parameters() if float(g32[k].norm()) > 0)
    assert rels[len(rels) // 2] <= 6e-2 and rels[-1] <= 0.3, (rels[len(rels) // 2], rels[-1])
