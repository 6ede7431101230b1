import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "whisper-finetune_amd"))
import torch, numpy as np, bench
from wfe_amd.frontend import HipFrontEnd
dev = torch.device("cuda", 0)
for n_mel in (80, 128):
    fe = HipFrontEnd(n_mel=n_mel, device=0)
    wav = bench.device_clips(64, dev, 0)
    lens = torch.full((64,), 480000, dtype=torch.int32, device=dev)
    out = fe.logmel(wav, lens)          # final values g = (max(v, max-8)+4)/4
    floor = out.amin(dim=(1, 2), keepdim=True)      # clamp floor per utterance (if any value was clamped, it equals g(max-8))
    gmax = out.amax(dim=(1, 2), keepdim=True)
    print(n_mel, "mel: floor-gmax", float((gmax - floor).mean()), "(2.0 means the clamp is active)")
    at = (out <= floor + 1e-7)
    B, M, T = out.shape
    Tp = T - T % 16
    a = at[:, :, :Tp].reshape(B, M, Tp // 16, 16)
    print("  values at floor:", float(at.float().mean()))
    print("  16-frame groups (all mel) with any:", float(a.any(dim=3).any(dim=1).float().mean()))
    print("  32-frame x mel-half blocks with any:", float(at[:, :, :T - T % 32].reshape(B, 2, M // 2, -1, 32).any(dim=4).any(dim=2).float().mean()))
    print("  16-frame x 16-mel blocks with any:", float(a.any(dim=3).reshape(B, M // 16, 16, -1).any(dim=2).float().mean()))
    print("  16-frame x 4-mel blocks with any:", float(a.any(dim=3).reshape(B, M // 4, 4, -1).any(dim=2).float().mean()))
    print("  16-frame x 1-mel rows with any:", float(a.any(dim=3).float().mean()))
