"""wfe_amd.launch: the reference's scripts run unchanged -- `utils.reader` / `utils.data_utils`
come from this package, every other `utils.*` module from the script's own tree, and
`--num_workers 0` is injected with the reason.  The stand-in tree below mimics the reference's
layout (finetune.py:10-14 imports; a utils/data_utils.py that cannot be imported, as the
reference's own: SURVEY 0.1); none of the reference's files is copied."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-finetune_amd")

SCRIPT = '''
import argparse, json, sys
from utils.callback import SavePeftModelCallback
from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding
from utils.model_utils import load_from_checkpoint
from utils.reader import CustomDataset
from utils.utils import print_arguments, make_inputs_require_grad, add_arguments
import utils, utils.callback, utils.data_utils, utils.model_utils, utils.reader, utils.utils
p = argparse.ArgumentParser()
p.add_argument("--num_workers", type=int, default=8)
p.add_argument("--base_model", type=str, default="x")
a = p.parse_args()
print(json.dumps({"files": {m.__name__: m.__file__ for m in (utils, utils.callback, utils.data_utils, utils.model_utils,
                                                             utils.reader, utils.utils)},
                  "num_workers": a.num_workers, "base_model": a.base_model, "argv0": sys.argv[0], "name": __name__,
                  "path0": sys.path[0], "collator": DataCollatorSpeechSeq2SeqWithPadding.__module__,
                  "dataset": CustomDataset.__module__}))
'''


def _tree(tmp_path):
    u = tmp_path / "ref" / "utils"
    u.mkdir(parents=True)
    (u / "__init__.py").write_text("")
    (u / "callback.py").write_text("class SavePeftModelCallback: pass\n")
    (u / "model_utils.py").write_text("def load_from_checkpoint(*a, **k): pass\n")
    (u / "utils.py").write_text("def print_arguments(*a): pass\ndef make_inputs_require_grad(*a): pass\n"
                                "def add_arguments(*a, **k): pass\n")
    (u / "reader.py").write_text("raise ImportError('the tree-local reader must not be imported')\n")
    (u / "data_utils.py").write_text("from utils.data_utils import DataCollatorSpeechSeq2SeqWithPaddingforEval\n"
                                     "raise ImportError('broken, like the reference file of this name')\n")
    script = tmp_path / "ref" / "finetune.py"
    script.write_text(SCRIPT)
    return script


def _launch(script, args, cwd, extra_env=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = PKG + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-m", "wfe_amd.launch", str(script)] + args, cwd=str(cwd), env=env,
                          capture_output=True, text=True, timeout=300)


def test_reference_style_script_runs_unchanged(tmp_path):
    script = _tree(tmp_path)
    for cwd in (script.parent, tmp_path):                    # from the checkout and from elsewhere
        r = _launch(script, ["--base_model", "openai/whisper-tiny"], cwd)
        assert r.returncode == 0, r.stderr[-3000:]
        got = json.loads(r.stdout.strip().splitlines()[-1])
        ours = os.path.join(PKG, "wfe_amd", "data.py")
        assert got["files"]["utils.reader"] == ours and got["files"]["utils.data_utils"] == ours
        for name in ("utils", "utils.callback", "utils.model_utils", "utils.utils"):
            assert got["files"][name].startswith(str(script.parent / "utils")), (name, got["files"][name])
        assert got["collator"] == got["dataset"] == "wfe_amd.data"
        assert got["num_workers"] == 0 and got["base_model"] == "openai/whisper-tiny"
        assert got["argv0"] == str(script) and got["name"] == "__main__" and got["path0"] == str(script.parent)
        assert "--num_workers (default 8) -> 0" in r.stderr and "forked" in r.stderr


def test_explicit_num_workers_is_replaced_loudly(tmp_path):
    script = _tree(tmp_path)
    for args in (["--num_workers", "8"], ["--num_workers=4"]):
        r = _launch(script, args, script.parent)
        assert r.returncode == 0, r.stderr[-3000:]
        assert json.loads(r.stdout.strip().splitlines()[-1])["num_workers"] == 0
        assert "-> 0: the GPU collator runs in the rank's main process" in r.stderr
    r = _launch(script, ["--num_workers", "0"], script.parent)
    assert r.returncode == 0 and "[wfe_amd.launch]" not in r.stderr      # nothing to say


def test_scripts_that_do_not_use_the_front_end_keep_their_arguments(tmp_path):
    s = tmp_path / "other.py"
    s.write_text("import sys, json\nprint(json.dumps(sys.argv[1:]))\n")
    r = _launch(s, ["--num_workers", "3", "x"], tmp_path)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1]) == ["--num_workers", "3", "x"]
    r = _launch(tmp_path / "missing.py", [], tmp_path)
    assert r.returncode != 0 and "no such script" in r.stderr


def test_under_torchrun_one_rank_stays_in_process(tmp_path):
    """`torchrun ... -m wfe_amd.launch finetune.py`: the rank torchrun started runs the script itself."""
    script = _tree(tmp_path)
    env = dict(os.environ)
    env["PYTHONPATH"] = PKG + os.pathsep + env.get("PYTHONPATH", "")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "wfe_amd.launch", str(script)],
                       cwd=str(script.parent), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and all(ln["num_workers"] == 0 and ln["collator"] == "wfe_amd.data" for ln in lines)


TRAIN_SCRIPT = '''
import argparse, json, os, sys, tempfile
from utils.reader import CustomDataset
from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding
p = argparse.ArgumentParser()
p.add_argument("--num_workers", type=int, default=8)
a = p.parse_args()
import torch
from transformers import Seq2SeqTrainingArguments
# as finetune.py:111-133 writes them: the old keyword, no dataloader_pin_memory
targs = Seq2SeqTrainingArguments(output_dir=tempfile.mkdtemp(), evaluation_strategy="no", report_to=[],
                                 dataloader_num_workers=a.num_workers, remove_unused_columns=False)
explicit = Seq2SeqTrainingArguments(output_dir=tempfile.mkdtemp(), report_to=[], dataloader_pin_memory=True)
print(json.dumps({"device_output": os.environ.get("WFE_DEVICE_OUTPUT"), "pin": targs.dataloader_pin_memory,
                  "pin_explicit": explicit.dataloader_pin_memory, "eval": str(getattr(targs, "eval_strategy", None)),
                  "prefetch": bool(getattr(torch.utils.data.DataLoader.__iter__, "__wfe_prefetch__", False)),
                  "workers": targs.dataloader_num_workers}))
'''


def test_launcher_keeps_features_on_the_gpu_and_prefetches(tmp_path):
    """under the launcher the unchanged script gets WFE_DEVICE_OUTPUT=1, training arguments whose
    dataloader_pin_memory defaults to False (an explicit True stays), the renamed evaluation_strategy keyword
    of finetune.py:120 accepted, --num_workers 0 and the one-batch-ahead DataLoader iterator."""
    script = _tree(tmp_path)
    script.write_text(TRAIN_SCRIPT)
    r = _launch(script, [], script.parent)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["device_output"] == "1" and got["pin"] is False and got["pin_explicit"] is True
    assert got["eval"].endswith("no") or got["eval"] == "IntervalStrategy.NO"
    assert got["prefetch"] is True and got["workers"] == 0
    r = _launch(script, [], script.parent, {"WFE_DEVICE_OUTPUT": "0"})      # the user's setting wins
    assert json.loads(r.stdout.strip().splitlines()[-1])["device_output"] == "0"


def test_prefetch_iterator_keeps_order_and_raises_in_place():
    sys.path.insert(0, PKG)
    import torch
    from wfe_amd import data as wd

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            if i == 9:
                raise RuntimeError("decode failed")
            return i
    it = wd.PrefetchIterator(iter(torch.utils.data.DataLoader(DS(), batch_size=2)))
    got = []
    try:
        for b in it:
            got.append(b.tolist())
        raised = None
    except RuntimeError as e:
        raised = str(e)
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7]] and raised == "decode failed"
    full = list(wd.PrefetchIterator(iter(torch.utils.data.DataLoader(list(range(5)), batch_size=2))))
    assert [b.tolist() for b in full] == [[0, 1], [2, 3], [4]]
