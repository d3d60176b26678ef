import sys, torch
sys.path.insert(0, "/root/repo")
import mpa_amd
from mpa_amd import ops
torch.manual_seed(0)
for M, K, C in ((2, 2048, 512), (2, 512, 256), (3, 64, 64), (64, 2048, 512)):
    x = torch.randn(M, K).cuda()
    lin = torch.nn.Linear(K, C).cuda(); bn = torch.nn.BatchNorm1d(C).cuda().train()
    bn2 = torch.nn.BatchNorm1d(C).cuda().train()
    ref = torch.relu(bn2(lin(x).double().float()))
    got = ops.linear_bn_act(x, lin.weight, lin.bias, bn, 0.0)
    ref64 = torch.relu(torch.nn.functional.batch_norm(lin.double()(x.double()), None, None, bn2.weight.double(), bn2.bias.double(), True, 0.1, 1e-5))
    print(M, K, C, "max |got-ref32| %.3e  |got-ref64| %.3e  |ref32-ref64| %.3e" % (float((got - ref).abs().max()), float((got.double() - ref64).abs().max()), float((ref.double() - ref64).abs().max())))
