#!/usr/bin/env python3
"""Golden key maps for the checkpoint handling of the reference drivers.

TEST INFRASTRUCTURE (build container only: /root/reference does not exist on the GPU box).  The drivers load checkpoints
with a few lines of plain dictionary code INSIDE their `__main__` blocks:

    /root/reference/train.py:280-293       DDP branch: `module.` prefixed keys, official GMFlow file under `module.GMFlow.`
    /root/reference/train.py:312-337       single-GPU branch: filters, the `backbone.pvtv2_en` rename, GMFlow under `GMFlow.`
    /root/reference/train.py:340-342       the name-substring freeze rule
    /root/reference/test.py:81-89          inference: optional `module.` strip (test_long.py:92-100 is the same text)
    /root/reference/train_long.py:391-402  short-term weights into Model_long (+ copies for injector1 / dr1 / decoder)
    /root/reference/train_long.py:404-406  freeze everything under `short_term`

This script READS those line ranges as text at run time, dedents and `exec`s them -- the reference's own statements, not
a restatement -- on stand-in objects: `model.state_dict()` returns the key manifests of tests/golden (every value a tag
"init:<key>"), `torch.load` returns scenario dictionaries whose values are tags ("ckpt:<key>", "flow:<key>"), and
`model.load_state_dict` is strict like nn.Module's (missing or unexpected keys raise) and records what it was given.
The result per scenario -- destination key -> source tag, or the exception class -- goes to tests/golden/ckpt_maps.json.gz;
tests/test_checkpoint_cpu.py runs emip_amd/checkpoint.py on the same scenarios and compares.  Nothing of the reference's
text is stored: the JSON holds scenario inputs (keys) and outputs (key maps)."""
import json
import os
import textwrap

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def ref_lines(fname, first, last):
    with open(os.path.join(REF, fname)) as f:
        lines = f.readlines()
    return textwrap.dedent("".join(lines[first - 1:last]))


class Param:
    def __init__(self):
        self.requires_grad = True

    def requires_grad_(self, flag=True):
        self.requires_grad = flag
        return self


class Model:
    """state_dict / load_state_dict / named_parameters of an nn.Module, on tags"""

    def __init__(self, manifest, prefix=""):
        self.keys = [prefix + k for k in manifest]
        # parameters = floating-point entries that are not BatchNorm running statistics
        self.params = {prefix + k: Param() for k, (shape, dt) in manifest.items()
                       if dt.startswith("float") and not k.endswith(("running_mean", "running_var"))}
        self.loaded = None

    def state_dict(self):
        return {k: "init:" + k for k in self.keys}

    def load_state_dict(self, sd):
        missing = [k for k in self.keys if k not in sd]
        unexpected = [k for k in sd if k not in set(self.keys)]
        if missing or unexpected:
            raise RuntimeError("strict load: missing %s unexpected %s" % (missing[:3], unexpected[:3]))
        self.loaded = dict(sd)

    def named_parameters(self):
        return list(self.params.items())


class _Torch:
    def __init__(self, files):
        self.files = files

    def load(self, path, *a, **k):
        return self.files[path]


def run(code, model, files, config=None, opt=None):
    ns = {"model": model, "torch": _Torch(files), "config": config, "opt": opt, "print": lambda *a, **k: None}
    try:
        exec(compile(code, "<reference lines>", "exec"), ns)            # noqa: S102 -- the reference's own statements
    except Exception as e:                                              # noqa: BLE001
        return {"error": type(e).__name__}
    out = {"map": {k: v for k, v in model.loaded.items() if not v.startswith("init:")}} if model.loaded is not None else {}
    out["frozen"] = sorted(n for n, p in model.named_parameters() if not p.requires_grad)
    return out


class Opt:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def scenarios(short, long_):
    """checkpoint key lists (inputs) per scenario"""
    sk = list(short)
    old_prefix = [k.replace("backbone.feat_net.pvtv2_en", "backbone.pvtv2_en") for k in sk if k.startswith("backbone.feat_net.pvtv2_en.")]
    pre = old_prefix[:40] + [k for k in sk if k.startswith(("decoder.", "dr1.", "injector1."))][:60]
    flow = [k[len("GMFlow."):] for k in sk if k.startswith("GMFlow.")]
    return {
        # a segmentation pre-train file in the old naming + entries the filter must drop
        "train_single_pretrain": dict(ckpt=pre + ["mask_downscaling.0.weight", "something.else", "PromptInteract.PatchEmbed.proj.weight"],
                                      flow=flow[:50] + ["not.in.model"]),
        # the same file, no GMFlow checkpoint configured
        "train_single_no_flow": dict(ckpt=pre, flow=None),
        # a full EMIP-short checkpoint (resume from Net_epoch_best.pth)
        "train_single_full": dict(ckpt=sk, flow=flow),
        # an entry the rename sends to a key the model does not have -> the strict load fails in the reference
        "train_single_unknown_after_rename": dict(ckpt=pre + ["backbone.pvtv2_en.not_a_layer.weight"], flow=None),
        "train_ddp_full": dict(ckpt=sk, flow=flow[:50] + ["not.in.model"]),
        "test_plain": dict(ckpt=sk[:300] + ["extra.key"]),
        "test_plain_given_ddp_file": dict(ckpt=["module." + k for k in sk[:300]]),
        "test_multi_gpu": dict(ckpt=["module." + k for k in sk] + ["module.extra.key"]),
        "test_multi_gpu_given_plain_file": dict(ckpt=sk[:300]),
        "train_long_from_short": dict(ckpt=sk + ["extra.key"]),
        "test_long_multi_gpu": dict(ckpt=["module." + k for k in long_]),
    }


def main():
    short = json.load(open(os.path.join(GOLDEN, "short_state_manifest.json")))
    long_ = json.load(open(os.path.join(GOLDEN, "long_state_manifest.json")))
    sc = scenarios(short, long_)
    single = ref_lines("train.py", 312, 337) + ref_lines("train.py", 340, 342)
    ddp = ref_lines("train.py", 280, 293) + ref_lines("train.py", 340, 342)
    test = ref_lines("test.py", 81, 89)
    test_long = ref_lines("test_long.py", 92, 100)
    long_train = ref_lines("train_long.py", 391, 402) + ref_lines("train_long.py", 404, 406)
    res = {}
    for name, s in sc.items():
        files = {"ckpt.pth": {k: "ckpt:" + k for k in s["ckpt"]}}
        cfg = {"load": {"path": "ckpt.pth", "flow_path": None}}
        if s.get("flow") is not None:
            files["flow.pth"] = {"model": {k: "flow:" + k for k in s["flow"]}}
            cfg["load"]["flow_path"] = "flow.pth"
        if name.startswith("train_single"):
            r = run(single, Model(short), files, cfg)
        elif name.startswith("train_ddp"):
            r = run(ddp, Model(short, "module."), files, cfg)
        elif name.startswith("test_long"):
            r = run(test_long, Model(long_), files, opt=Opt(snap_path="ckpt.pth", multi_gpu="multi_gpu" in name))
        elif name.startswith("test_"):
            r = run(test, Model(short), files, opt=Opt(snap_path="ckpt.pth", multi_gpu="multi_gpu" in name))
        else:
            r = run(long_train, Model(long_), files, cfg)
        res[name] = r          # the inputs are scenarios() of this file: the test rebuilds them from the same manifests
        print("%-38s %s" % (name, r.get("error") or "%d entries taken, %d frozen" % (len(r["map"]), len(r["frozen"]))))
    import gzip
    with gzip.GzipFile(os.path.join(GOLDEN, "ckpt_maps.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(res, sort_keys=True).encode())


if __name__ == "__main__":
    main()
