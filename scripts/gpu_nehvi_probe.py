import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_nehvi_gpu import _setup
from baybe_amd.nehvi import HipNEHVI, compute_ref_point
from oracle import nehvi_oracle as no
for m, signs in ((2,None),(3,None)):
    X, Xt, Y, signs, engines, models = _setup(m, signs=signs)
    ref = compute_ref_point(Y * signs[None, :])
    S, seed = 32, 11
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=False)
    hv.prepare(seed)
    print("ext jitter per target", [o.ext.jitter for o in hv.outputs], "engine jitter", [o.engine.jitter for o in hv.outputs])
    sg = hv.score(torch.from_numpy(X).cuda()).cpu().numpy()
    z = no.sobol_normal_base_samples_nd(S, len(Xt) + 1, m, seed)
    orc = no.NEHVIOracle(models, signs, Xt, ref, z)
    so = orc.values(X[:60])
    dup = np.array([(np.abs(Xt - x).sum(1) < 1e-12).any() for x in X[:60]])
    print("m", m, "max |diff| regular", np.abs(sg[:60] - so)[~dup].max(), "duplicates", int(dup.sum()))
    # per-sample candidate values: device conditional mean/var vs oracle joint samples
    for o in range(m):
        tm = hv.outputs[o].ext.posterior_columns(torch.from_numpy(X[:60]).cuda()).cpu().numpy()
        _, var = hv.outputs[o].ext.posterior(torch.from_numpy(X[:60]).cuda())
        var = var.cpu().numpy()
        fd = tm + np.sqrt(np.maximum(var,0))[:,None]*hv.zx[None,:,o]
        fo = np.stack([orc.candidate_samples(x)[:,o] for x in X[:60]])
        print("   target",o,"max |f_dev - f_orc| regular", np.abs(fd-fo)[~dup].max(), " var min", var[~dup].min())
