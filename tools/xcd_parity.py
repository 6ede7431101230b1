import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "whisper-finetune_amd"))
import torch, numpy as np, bench
from wfe_amd.frontend import HipFrontEnd
dev = torch.device("cuda", 0)
fe = HipFrontEnd(n_mel=80, device=0)
B = 256
wav = bench.device_clips(B, dev, 0)
wav = torch.cat([wav, wav[:, :40960]], dim=1).contiguous()
rs = np.random.default_rng(99)
lens = torch.from_numpy(rs.integers(16000, 15 * 16000, size=B).astype(np.int32)).to(dev)
for n in (480000, 481280, 491520, 501760, 409600):
    out = torch.empty((B, 80, n // 160), device=dev)
    for _ in range(30): fe.logmel(wav, lens, n_samples=n, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): fe.logmel(wav, lens, n_samples=n, out=out)
    b.record(); torch.cuda.synchronize()
    print(f"n_samples {n}: tiles/utterance {(n // 160 + 63) // 64}  {a.elapsed_time(b) / 200:.4f} ms/step")
