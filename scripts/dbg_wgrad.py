import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops
n,h,w,cin,cout = 2,7,5,64,128
x = torch.ones(n,h,w,cin, device="cuda"); dy = torch.ones(n,h,w,cout, device="cuda")
dw = torch.zeros(cout,3,3,cin, device="cuda")
ops.conv3x3_wgrad(x, None, None, dy, dw, n,h,w,cin,cout)
print("got  ", dw[0,:,:,0].cpu())
exp = torch.tensor([[n*(h-abs(kh-1))*(w-abs(kw-1)) for kw in range(3)] for kh in range(3)])
print("want ", exp)
print("uniform over co,ci:", bool((dw == dw[0:1,:,:,0:1]).all()))
# ramp test: x = pixel index, dy = 1 -> dw[tap] = sum of shifted pixel indices
m = torch.arange(n*h*w, device="cuda", dtype=torch.float32).view(n,h,w,1).expand(n,h,w,cin).contiguous()
ops.conv3x3_wgrad(m, None, None, dy, dw, n,h,w,cin,cout)
import torch.nn.functional as F
ref = torch.nn.grad.conv2d_weight(m.permute(0,3,1,2).double().cpu(), (cout,cin,3,3), dy.permute(0,3,1,2).double().cpu(), padding=1)
print("ramp got ", dw[0,:,:,0].cpu()); print("ramp want", ref[0,0])
