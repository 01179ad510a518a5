import sys
sys.path.insert(0, "rp-tree_amd/python"); sys.path.insert(0, ".")
import numpy as np, torch
import rptree_amd as rp
ctx = rp.Context(0)
N, d, T = 1000000, 128, 32
import bench
X = bench.synth(N, d, 1234, torch.device("cuda", 0))
Q = bench.synth(64, d, 4321, torch.device("cuda", 0))
cfg = rp.rpTreeCfg(128, N, d)
ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_F64, keep=X)
_, R = rp.gen.forest_hyperplanes(1235137, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, 128, rp.RPT_PROJ_MFMA)
off, ids = rp.candidatesBatch(f, Q.cpu().numpy())
tot = uniq = 0
for i in range(64):
    c = ids[off[i * T]:off[(i + 1) * T]]
    tot += len(c); uniq += len(np.unique(c))
print("candidates per query %.1f unique %.1f ratio %.3f" % (tot / 64, uniq / 64, uniq / tot))
