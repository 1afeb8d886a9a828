"""Drop-in check against the real reference tree (build container only): with
install_into_reference() the reference's model files import OUR softsplat / euler modules
(no cupy anywhere), the models construct, and their forward_flow reaches our operators."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def test_reference_models_import_and_call_our_ops():
    import slr_sfs_amd as S
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.") or k.startswith("options")
             or k in ("cupy", "cv2", "av", "lz4framed", "torchvision")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        for n in ("cv2", "av", "lz4framed"):
            sys.modules[n] = types.ModuleType(n)
        tv = types.ModuleType("torchvision")
        tv.transforms = types.ModuleType("torchvision.transforms")
        tv.models = types.SimpleNamespace(vgg19=None)
        tv.utils = types.ModuleType("torchvision.utils")
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tv.transforms, "torchvision.utils": tv.utils})
        assert "cupy" not in sys.modules
        ss, eim = S.install_into_reference()
        import models.animating_softmax_splating as A               # does `from models import softsplat`
        assert A.softsplat is ss and A.euler_integration is eim.euler_integration
        assert "cupy" not in sys.modules                             # nothing pulled cupy in
        from options.train_options import ArgumentParser
        opt, _ = ArgumentParser().parse(
            "--model_type softmax_splating --refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm "
            "--pconv pconv_pbn_woresbias --norm_G sync:spectral_batch --train_Z --losses 1.0_l1 --W 16")
        model = A.AnimatingSoftmaxSplating(opt).eval()
        assert isinstance(model.softsplater, ss.ModuleSoftsplat) and model.softsplater.strType == "summation"
        assert isinstance(model.euler_integration, eim.EulerIntegration)
        # on this GPU-less box the call chain must end in OUR operator's refusal of CPU tensors
        # (same exception type as the reference's own CPU branch, softsplat.py:418-419)
        torch.Tensor.cuda_backup = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            batch = {"features": [(torch.zeros(1, 64, 16, 16), torch.zeros(1, 1, 16, 16))],
                     "images": [torch.zeros(1, 3, 16, 16)], "motions": [torch.zeros(1, 2, 16, 16)],
                     "index": torch.tensor([[0, 1, 4]])}
            with pytest.raises(NotImplementedError, match="ROCm device"):
                model.forward_flow(batch)
        finally:
            torch.Tensor.cuda = torch.Tensor.cuda_backup
            del torch.Tensor.cuda_backup
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k.startswith("options")
                  or k in ("cv2", "av", "lz4framed", "torchvision", "torchvision.transforms", "torchvision.utils")]:
            del sys.modules[k]
        sys.modules.update(saved)
