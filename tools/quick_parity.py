import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/whisper-finetune_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import cases
from oracle import logmel as ol
from wfe_amd.frontend import HipFrontEnd, pack_ragged
fe = HipFrontEnd(n_mel=80, device=0)
w = [cases.load_test_wav(), cases.synth_clip(0)]
flat, off, lens = pack_ragged(w, fe.device)
got = fe.logmel(flat, lens, offsets=off).cpu().numpy()
want = ol.logmel_f32(w, n_mel=80)
print(os.environ.get("WFE_LIB_PATH", "default"), "max err", np.abs(got - want).max())
