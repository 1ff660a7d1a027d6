import sys, torch
sys.path[:0]=['/root/repo','/root/repo/mode-2022_amd','/root/repo/tests']
from mode_hip import functional as HF
from oracle import mode_ref
import plain_ops
torch.manual_seed(63)
for scale in (1.0, 4.0):
    lg = (torch.randn(1,1,48,32,32)*scale)
    ref = mode_ref.disparity_head(lg.double(), 192, 128, 128)
    got = HF.head_fwd(lg.cuda(), (192,128,128)).cpu().double()
    pl = plain_ops.head(lg.cuda(), (192,128,128)).cpu().double()
    print('scale',scale,'hip vs fp64 max %.3e mean %.3e | vendor-composition vs fp64 max %.3e mean %.3e | hip vs vendor %.3e' % (
        (got-ref).abs().max(), (got-ref).abs().mean(), (pl-ref).abs().max(), (pl-ref).abs().mean(), (got-pl).abs().max()))
    la = lg.double().requires_grad_(True)
    pr = mode_ref.disparity_head(la, 192, 128, 128)
    g = torch.randn(1,1,128,128)
    pr.backward(g.double())
    gl = HF.head_bwd(lg.cuda(), g.cuda(), (192,128,128)).cpu().double()
    print('   bwd: max err %.3e of max %.3e' % ((gl-la.grad).abs().max(), la.grad.abs().max()))
