"""GPU: BASELINE configs[3] and configs[4] END TO END - the backbone at full size into the consumer the configuration names
(UPerHead + FCNHead; the pixel decoder's 6-layer deformable encoder), loss, backward through both, bf16 autocast.
Shapes, finite gradients everywhere, the backbone's gradients reached.  (Named to run last: these are the only tests that
put MIOpen convolutions of new shapes and the largest models of the suite on the card.)"""
import os

import pytest
import torch

os.environ.setdefault('MIOPEN_FIND_MODE', '2')      # MIOpen convolutions of the heads: no exhaustive search per new shape

pytestmark = pytest.mark.gpu


def test_configs3_backbone_plus_uperhead_640_bf16():
    """BASELINE configs[3] end to end: ViT-Adapter-L at 640 x 640, batch 2, into UPerHead + the auxiliary FCNHead of the
    UperNet configs (4 x 1024 channels in, 512 inside, 150 classes: upernet_augreg_adapter_large_512_160k_ade20k.py:30-31),
    cross-entropy on random labels, backward through heads AND backbone under bf16 autocast: logits at 1/4 resolution,
    every gradient finite and the backbone's reached (the heads are restated from mmseg - parity unpinned, tests/test_heads.py)."""
    from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter
    from vitadapter.heads import FCNHead, UPerHead
    kw = dict(PRESETS['large_seg'])
    kw['drop_path_rate'] = 0.0
    torch.manual_seed(0)
    model = ViTAdapter(**kw).cuda().train()
    head = UPerHead(in_channels=(1024,) * 4, channels=512, num_classes=150).cuda().train()
    aux = FCNHead(in_channels=1024, in_index=2, channels=256, num_classes=150).cuda().train()
    x = torch.randn(2, 3, 640, 640, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    labels = torch.randint(0, 150, (2, 640, 640), device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        feats = model(x)
        logits, aux_logits = head(feats), aux(feats)
    assert tuple(logits.shape) == (2, 150, 160, 160) and tuple(aux_logits.shape) == (2, 150, 40, 40)
    up = lambda t: torch.nn.functional.interpolate(t.float(), size=(640, 640), mode='bilinear', align_corners=False)   # noqa: E731
    loss = torch.nn.functional.cross_entropy(up(logits), labels) + 0.4 * torch.nn.functional.cross_entropy(up(aux_logits), labels)
    loss.backward()
    assert torch.isfinite(loss)
    for mod, least in ((model, 500), (head, 30), (aux, 4)):
        grads = [p.grad for p in mod.parameters() if p.grad is not None]
        assert len(grads) >= least and all(torch.isfinite(g).all() for g in grads)
    assert float(model.spm.fc1.weight.grad.abs().max()) > 0 and float(model.blocks[0].attn.qkv.weight.grad.abs().max()) > 0


def test_configs4_backbone_plus_pixel_decoder_encoder_800x1344_bf16():
    """BASELINE configs[4] end to end on the path this repo owns: ViT-Adapter-L at 800 x 1344, one image, its three coarsest
    maps projected to 256 channels (1 x 1 convolution + GroupNorm as msdeformattn_pixel_decoder.py:93-104) and flattened
    from low to high resolution into the 6-layer deformable encoder (Lq = S = 22 050, 8 heads, 3 levels), backward through
    encoder AND backbone under bf16 autocast: shapes, finite gradients, the backbone's reached."""
    from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter
    from vitadapter.pixel_decoder import MSDeformAttnEncoder, encoder_inputs
    kw = dict(PRESETS['large_seg'])
    kw['drop_path_rate'] = 0.0
    torch.manual_seed(0)
    model = ViTAdapter(**kw).cuda().train()
    enc = MSDeformAttnEncoder().cuda().train()
    with torch.no_grad():
        for layer in enc.layers:
            layer.attentions[0].sampling_offsets.weight.normal_(0, 0.02)
            layer.attentions[0].attention_weights.weight.normal_(0, 0.05)
    proj = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Conv2d(1024, 256, 1), torch.nn.GroupNorm(32, 256)) for _ in range(3)]).cuda()
    x = torch.randn(1, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(6))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        feats = model(x)
        levels = [proj[i](feats[3 - i]) for i in range(3)]                     # stride 32, 16, 8: low to high resolution
        shapes = [tuple(t.shape[2:]) for t in levels]
        assert shapes == [(25, 42), (50, 84), (100, 168)]
        query = torch.cat([t.flatten(2).permute(2, 0, 1) for t in levels], 0)  # (Lq, N, 256)
        _, pos, ref, ss, lsi = encoder_inputs(shapes, 1, 256, 'cuda', seed=7)
        memory = enc(query=query.float(), query_pos=pos, spatial_shapes=ss, reference_points=ref, level_start_index=lsi)
    assert tuple(memory.shape) == (22050, 1, 256)
    memory.float().pow(2).mean().backward()
    for mod, least in ((enc, 90), (proj, 9), (model, 500)):
        grads = [p.grad for p in mod.parameters() if p.grad is not None]
        assert len(grads) >= least and all(torch.isfinite(g).all() for g in grads)
    assert float(model.blocks[-1].mlp.fc2.weight.grad.abs().max()) > 0
