import numpy as np, sys
sys.path.insert(0,'/root/repo')
from gnina_amd import capi, synth
from oracle import vina as V
from tests import vina_scene
capi.init(0)
T=V.Tables()
sc=vina_scene.build(0); lig=sc['lig']
gd=V.setup_grid_dims(sc['center'],sc['size'])
types=sorted(set(int(t) for t in lig['smt'] if t>1))
grids={t:V.cache_populate(T,gd,sc['rec_xyz'],sc['rec_smt'],t) for t in types}
S=V.Scene(T,gd,grids,V.LigandHandle(lig))
vina=capi.Vina(); vina.set_receptor(sc['rec_xyz'],sc['rec_smt']); vina.build_cache(list(gd.begin),list(gd.end),list(gd.n),types,1e3); vina.set_ligand(lig)
rng=np.random.RandomState(12)
confs=np.stack([synth.random_conf(rng,lig,sc['center'],spread=1.0) for _ in range(6)])
v=(10.,10.,10.)
for it in (1,2,3,5,10,19):
    e,cf,g,ev=vina.bfgs_batch(confs,v,max_iters=it)
    row=[]
    for b in range(len(confs)):
        e0,c0,g0,ev0=S.bfgs(confs[b],v,max_iters=it)
        row.append(f"{e[b]:.4f}/{e0:.4f}({ev[b]}/{ev0})")
    print(it,' '.join(row))
