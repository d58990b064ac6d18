import sys, copy, torch, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import test_detector_gpu as T
from unibev_amd import registry as reg, synthetic as syn
from unibev_amd import functional as UF
T.test_detector_forward_bev_matches_the_oracle_chain('linear','ChannelNormWeights')
T.test_detector_forward_bev_matches_the_oracle_chain('cat',None)
orig = UF.ModulatedDeformConv2dFunction.backward
log=[]
def bwd(ctx, g):
    outs = orig(ctx, g)
    names=['gx','goff','gmask','gw','gb']
    st={n: (o is None or bool(torch.isfinite(o).all())) for n,o in zip(names, outs[:5])}
    st['gin']=bool(torch.isfinite(g).all()); st['shape']=tuple(g.shape)
    log.append(st)
    return outs
UF.ModulatedDeformConv2dFunction.backward = staticmethod(bwd)
cfg=T._model_cfg(); 
nbad=0
for it in range(12):
    torch.manual_seed(1)
    det=reg.DETECTORS.build(copy.deepcopy(cfg)).cuda().train()
    pts=[torch.from_numpy(c).cuda() for c in T._clouds()]
    metas=syn.img_metas(2,T.NCAM,T.IMG_HW)
    det.voxelize(pts)
    imgs=torch.randn(2,T.NCAM,3,*T.IMG_HW,device='cuda')
    np.random.seed(0); log.clear()
    with torch.no_grad():
        img_feats, pts_feats, _ = det.extract_feat(imgs, pts, None, metas)
        fin = lambda t: bool(torch.isfinite(t).all())
        tr = det.pts_bbox_head.transformer
        np.random.seed(0)
        fused, img_bev, pts_bev = None, None, None
        bq, bpos = det.pts_bbox_head.bev_inputs(2, img_feats[0].dtype, img_feats[0].device)
        res = tr.encode(img_feats, pts_feats, bq, T.BEV_H, T.BEV_W, bev_pos=bpos, img_metas=metas, return_parts=True)
        ok = [fin(img_feats[0]), fin(pts_feats[0])] + [fin(r) for r in res]
        if not all(ok):
            nbad += 1
            print(it, 'img_feats, pts_feats, fused, img_bev, pts_bev finite:', ok, 'flags', tr.c_flag, tr.l_flag)
    continue
    bad=[n for n,p in det.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if bad:
        nbad+=1
        print(it, 'nbad params', len(bad), bad[:3])
        for i,st in enumerate(log):
            if not all(v for k,v in st.items() if k!='shape'): print('   dcn bwd call', i, st)
    del det
print('bad iterations', nbad)
