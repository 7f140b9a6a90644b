import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench_module_api as M
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
b = 3
vols, slices = M.synthetic_inputs(M.GEOM, b, dev)
def eng(p): 
    (os.environ.get('NOEMPTY') or torch.cuda.empty_cache()); r = M.engine_rate(M.GEOM, b, dev, p, 12, 3, vols, slices)[0]; print('engine', p, round(r, 2), flush=True)
def mod(p):
    (os.environ.get('NOEMPTY') or torch.cuda.empty_cache()); r = M.module_rate(M.GEOM, b, dev, p, 12, 3, vols, slices)[0]; print('module', p, round(r, 2), flush=True)
for seq in sys.argv[1:]:
    print('--', seq)
    for tok in seq.split(','):
        kind, p = tok.split(':')
        (eng if kind == 'e' else mod)(p)
