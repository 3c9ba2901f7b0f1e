"""bf16 mode of the shared-MLP stacks against the fp32 path, next to the torch executor under autocast(bfloat16) (GPU box):
the numbers tests/test_mlp_gpu.py::test_bf16_sa_stack_within_restated_tolerance asserts against."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_gpu import make_cd, run_cd, CASES
from repsurf_amd import mlp
cosf = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
for groups, ns, pos, feat, widths in CASES[:4]:
    mod = make_cd(pos, feat, widths, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    out_f, g_f = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    mlp.set_precision("bf16")
    out_b, g_b = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    mlp.set_precision("fp32")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_a, g_a = run_cd(copy.deepcopy(mod), x, ns, pos, "torch", w)
    print(groups, ns, widths, "out err hip-bf16 %.3e  autocast %.3e" % ((out_b - out_f).abs().max().item(), (out_a.float() - out_f).abs().max().item()))
    for name in g_f:
        if g_f[name].abs().max() == 0:
            continue
        print("   %-16s cos hip-bf16 %.5f   torch-autocast %.5f" % (name, cosf(g_b[name], g_f[name]), cosf(g_a[name].float(), g_f[name])))
