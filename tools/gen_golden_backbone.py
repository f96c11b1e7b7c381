"""Backbone goldens (G4 parts + G5 full ViTAdapter) from the reference's own classes.
Imported by tools/gen_golden.py ("backbone" target); container only, never on the GPU box.

The fixture holds expected OUTPUTS (+ digests); weights and inputs are regenerated from
oracle/seeded.py by whoever checks against it.  Config literals live in oracle/backbone_cases.py
so that the tests build the same models.
"""
import json
import os

import numpy as np
import torch

from gen_golden import GOLD, _np, load_reference_backbone, load_reference_ops
from oracle import backbone_cases as bc
from oracle import seeded


def _load_seeded(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(seeded.seeded_state_dict(shapes, seed))
    return shapes


def _grads(outs, inputs, gouts):
    loss = sum((o * g).sum() for o, g in zip(outs, gouts))
    return torch.autograd.grad(loss, inputs, allow_unused=True)


def gen_backbone():
    load_reference_ops()
    g = {}
    meta = {}
    for flavour in ('seg', 'det'):
        vit, am, va = load_reference_backbone(flavour)

        # ---- G5: full model --------------------------------------------------------------
        for name, case in bc.FULL_CASES.items():
            if case['cfg']['flavour'] != flavour:
                continue
            kw = {k: v for k, v in case['cfg'].items() if k != 'flavour'}
            model = va.ViTAdapter(**kw)
            shapes = _load_seeded(model, seed=5)
            meta[name] = {k: list(s) for k, s in shapes.items()}
            for mode in case['modes']:
                model.train(mode == 'train')
                x = bc.full_input(name).requires_grad_(True)
                outs = model(x)
                gouts = bc.full_gouts(name, [o.shape for o in outs])
                params = [p for p in model.parameters()]
                grads = _grads(outs, [x] + params, gouts)
                tag = '%s_%s' % (name, mode)
                for k, o in enumerate(outs):
                    g['%s_f%d' % (tag, k + 1)] = _np(o)
                g[tag + '_gx'] = _np(grads[0])
                for (pn, _), gr in zip(model.named_parameters(), grads[1:]):
                    if gr is not None:
                        g['%s_gp_%s' % (tag, pn)] = seeded.digest(gr)

        if flavour != 'seg':
            continue
        # ---- G4: parts (the two flavours share this code; seg copy used) ------------------
        E, M, Dm = bc.PART['embed'], bc.PART['deform_heads'], bc.PART['ratio']
        H = W = bc.PART['tokens']
        geo1, geo2 = bc.part_geometry()
        x, c = bc.part_tokens()

        inj = am.Injector(dim=E, n_levels=3, num_heads=M, init_values=0., n_points=4, deform_ratio=Dm)
        _load_seeded(inj, 6)
        xi, ci = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
        o = inj(xi, geo1[0], ci, geo1[1], geo1[2])
        gr = _grads([o], [xi, ci], [bc.part_gout('inj', o.shape)])
        g['part_inj_out'], g['part_inj_gx'], g['part_inj_gc'] = _np(o), _np(gr[0]), _np(gr[1])

        ext = am.Extractor(dim=E, num_heads=M, n_points=4, n_levels=1, deform_ratio=Dm,
                           with_cffn=True, cffn_ratio=0.25)
        _load_seeded(ext, 7)
        xi, ci = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
        o = ext(ci, geo2[0], xi, geo2[1], geo2[2], H, W)
        gr = _grads([o], [xi, ci], [bc.part_gout('ext', o.shape)])
        g['part_ext_out'], g['part_ext_gx'], g['part_ext_gc'] = _np(o), _np(gr[0]), _np(gr[1])

        spm = am.SpatialPriorModule(inplanes=bc.PART['inplanes'], embed_dim=E)
        _load_seeded(spm, 8)
        for mode in ('eval', 'train'):
            spm.train(mode == 'train')
            img = bc.part_image().requires_grad_(True)
            outs = spm(img)
            gr = _grads(outs, [img], [bc.part_gout('spm%d' % k, o.shape) for k, o in enumerate(outs)])
            for k, o in enumerate(outs):
                g['part_spm_%s_c%d' % (mode, k + 1)] = _np(o)
            g['part_spm_%s_gimg' % mode] = _np(gr[0])

        for bname, (windowed, Hb, Wb) in bc.BLOCK_CASES.items():
            blk = vit.Block(dim=E, num_heads=bc.PART['heads'], mlp_ratio=4., qkv_bias=True,
                            windowed=windowed, window_size=14, layer_scale=True,
                            norm_layer=torch.nn.LayerNorm)
            _load_seeded(blk, 9)
            t = bc.block_tokens(bname).requires_grad_(True)
            o = blk(t, Hb, Wb)
            gr = _grads([o], [t], [bc.part_gout(bname, o.shape)])
            g['part_%s_out' % bname], g['part_%s_gx' % bname] = _np(o), _np(gr[0])

    g['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, 'backbone.npz'), **g)
    print('backbone goldens written')
