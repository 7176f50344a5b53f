import sys; import os; R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, stereo_vo_amd as S, oracle_lib as O
from test_pipeline import _seq, _ora_pipe
ctx = S.Context(1280,720,max_batch=4,max_corners=4096,max_candidates=1<<17,max_features=4096)
for md,maxc,batch in [(30.0,300,1),(12.0,300,4),(8.0,1500,3)]:
    n=12; p,L,R=_seq(n)
    pp=S.pipeline_default_params(); pp.cam.focal,pp.cam.cx,pp.cam.cy,pp.cam.baseline=p.focal,p.cx,p.cy,p.baseline
    pp.width,pp.height=p.width,p.height; pp.max_corners,pp.min_feature_distance,pp.max_features=maxc,md,max(400,maxc); pp.ba_max_time_s=0.0
    g=S.Pipeline(ctx,pp); o=_ora_pipe(p,min_feature_distance=md,max_corners=maxc,max_features=max(400,maxc))
    for b0 in range(0,n,batch):
        rg=g.process_batch(L[b0:b0+batch],R[b0:b0+batch])
        for k,r in enumerate(rg):
            ro=o.process(L[b0+k],R[b0+k])
            d=np.abs(np.array(list(r.pose7))-np.array(list(ro.pose7)))
            print(md,b0+k,r.is_keyframe,r.n_tracked,r.n_inliers,r.ba_iterations,"maxdiff %.2e"%d.max(), "t=",np.array(list(r.pose7))[4:])
