import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import copy, sys, torch
pass
from slr_sfs_amd import nets
torch.manual_seed(3)
dec = nets.DecoderPconv2(64, 3).eval()
with torch.no_grad():
    for m in dec.modules():
        if hasattr(m, "stored_mean"):
            m.stored_mean.normal_(0, 0.3); m.stored_var.uniform_(0.5, 1.5)
    for hw in ((72, 136), (192, 320)):
        x = torch.randn(1, 64, *hw); x[:, :, 20:50, 30:80] = 0
        with nets.cpu_reference():
            y32 = dec(x); y64 = copy.deepcopy(dec).double()(x.double())
        y = dec.cuda()(x.cuda()).cpu(); dec.cpu()
        print(hw, "out max", y64.abs().max().item(), "err hip", (y.double() - y64).abs().max().item(), "err torch-fp32 cpu", (y32.double() - y64).abs().max().item(),
              "tanh-domain err hip", (torch.tanh(y.double()) - torch.tanh(y64)).abs().max().item())
