"""How far apart are two VALID fp32 evaluations of the segmentation step?  The CPU oracle (fp64 BatchNorm statistics), the same with
torch's own fp32 BatchNorm kernel (what nn.BatchNorm1d executes), and one thread instead of eight, each against the float64
evaluation, at configs[3] (16 x 4096 x 6): relative L2 error per gradient tensor + logits.  CPU only (~1 min).
    python tools/fp32_spread.py > profiles/r03/seg_fp32_spread.txt"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import seg_ref, torch_ref
from tests.util import seg_state
B=16
r = np.random.RandomState(3); n = B * 4096
coord = (r.rand(n, 3) * 2 - 1).astype(np.float32); rgb = r.rand(n, 3).astype(np.float32)
offset = (np.arange(1, B+1) * 4096).astype(np.int32); label = r.randint(0, 13, n).astype(np.int64)
np.random.seed(17); flips = np.where(np.random.rand(B) < 0.5, 1.0, -1.0).astype(np.float32)
torch.set_num_threads(8)
truth = seg_ref.step(seg_state(), coord, rgb, offset, label, flips, dtype=torch.float64)
a = seg_ref.step(seg_state(), coord, rgb, offset, label, flips)
orig = seg_ref._bn_train
seg_ref._bn_train = torch_ref._bn_train_fp32
b = seg_ref.step(seg_state(), coord, rgb, offset, label, flips)
seg_ref._bn_train = orig
torch.set_num_threads(1)
c = seg_ref.step(seg_state(), coord, rgb, offset, label, flips)
def err(x):
    out={}
    for k in truth['grads']:
        t=truth['grads'][k].numpy().ravel(); nrm=np.linalg.norm(t)
        if nrm<1e-5: continue
        out[k]=np.linalg.norm(x['grads'][k].double().numpy().ravel()-t)/nrm
    return out
ea,eb,ec=err(a),err(b),err(c)
for k in ('surface_constructor.mlps.1.weight','sa4.mlp_f0.weight','sa3.bn_f0.weight','sa1.bn_f0.weight','fp1.mlp_bns.1.weight','classifier.0.weight','classifier.1.bias'):
    print(k, 'A(fp64 BN stats, 8 thr) %.3g  B(torch fp32 BN) %.3g  C(1 thread) %.3g'%(ea[k],eb[k],ec[k]))
print('median', np.median(list(ea.values())), np.median(list(eb.values())), np.median(list(ec.values())))
print('max', max(ea.values()), max(eb.values()), max(ec.values()))
for nm,x in (('A',a),('B',b),('C',c)):
    print(nm,'logits', np.abs(x['logits'].detach().double().numpy()-truth['logits'].detach().numpy()).max())
