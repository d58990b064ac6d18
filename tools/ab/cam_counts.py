import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import bench_lift as BL
value, offlog, ref, vis0, count, gout, geom, is_grid, center = BL.instance('img', 2, torch.float32, torch.device('cuda'))
print('visible per camera:', vis0.sum(1).tolist(), 'total', int(vis0.sum()))
