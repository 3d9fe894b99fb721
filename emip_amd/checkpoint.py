"""Checkpoint loading / saving with the reference drivers' key handling.

Because `state_dict` keys and shapes of `emip_amd`'s modules are the reference's (tests/golden/*_state_manifest.json),
files written by either side load on the other.  The functions below restate the filtering / renaming the reference's
drivers do inline before `load_state_dict` (pure host-side dictionary logic, no kernels):

  load_for_inference   /root/reference/test.py:81-89, test_long.py:92-100
  load_short_pretrain  /root/reference/train.py:312-337   (segmentation pre-train + official GMFlow weights)
  load_short_pretrain_ddp  /root/reference/train.py:280-293 (the --multi_gpu branch: the wrapped model's `module.` keys)
  load_long_pretrain   /root/reference/train_long.py:391-406 (short-term weights into Model_long, then freeze them)
  save                 /root/reference/train.py:90,162 (`torch.save(model.state_dict(), path)`; DDP adds `module.`)
"""
import torch


def load_for_inference(model, checkpoint, multi_gpu=False):
    """test.py:81-89: keep the entries the model knows (after stripping DDP's `module.` when multi_gpu), merge into
    the model's own state_dict and load strictly.  `checkpoint`: a state dict (or a path)."""
    checkpoint = _read(checkpoint)
    model_dict = model.state_dict()
    if multi_gpu:
        pretrained = {k.split('module.')[-1]: v for k, v in checkpoint.items() if k.split('module.')[-1] in model_dict}
    else:
        pretrained = {k: v for k, v in checkpoint.items() if k in model_dict}
    model_dict.update(pretrained)
    model.load_state_dict(model_dict)
    return sorted(pretrained)


def load_short_pretrain(model, checkpoint, flow_checkpoint=None):
    """train.py:312-337.  Entries of the pre-train file that the model knows are taken (except the two excluded
    substrings), `backbone.pvtv2_en.*` is renamed to `backbone.feat_net.pvtv2_en.*`, `PromptInteract` entries are kept
    and duplicated under `cod_adaptor_prompt`; the official GMFlow checkpoint (`{'model': ...}`) is mapped under `GMFlow.`.
    Keys the model does not have are dropped, because the merged dictionary is then loaded strictly."""
    checkpoint = _read(checkpoint)
    model_dict = model.state_dict()
    ori = {k: v for k, v in checkpoint.items()
           if ((k in model_dict and 'PromptInteract.PatchEmbed.proj.weight' not in k and "mask_downscaling" not in k)
               or ('backbone.pvtv2_en' in k))}
    pretrained = {}
    for k, v in ori.items():
        if 'backbone.pvtv2_en' in k:
            pretrained[k.replace('backbone.pvtv2_en', 'backbone.feat_net.pvtv2_en')] = v
        elif 'PromptInteract' in k:
            pretrained[k] = v
            pretrained[k.replace('PromptInteract', 'cod_adaptor_prompt')] = v
        else:
            pretrained[k] = v
    # the reference updates model_dict with everything and would fail in load_state_dict on unknown keys; the shipped
    # configuration never produces any (both renames land on existing keys), so unknown keys are an error here too
    unknown = [k for k in pretrained if k not in model_dict]
    if unknown:
        raise KeyError("checkpoint entries without a destination in the model: %s" % unknown[:5])
    model_dict.update(pretrained)
    loaded_flow = []
    if flow_checkpoint is not None:
        flow = _read(flow_checkpoint)
        flow_dict = {'GMFlow.' + k: v for k, v in flow['model'].items() if 'GMFlow.' + k in model_dict}
        model_dict.update(flow_dict)
        loaded_flow = sorted(flow_dict)
    model.load_state_dict(model_dict)
    return sorted(pretrained), loaded_flow


def load_short_pretrain_ddp(wrapped, checkpoint, flow_checkpoint=None):
    """train.py:280-293, the --multi_gpu branch: `wrapped` is the DistributedDataParallel-wrapped model, whose state_dict
    keys carry `module.`.  Entries of the file that exist under that prefix are taken AS THEY ARE -- this branch has neither
    the `backbone.pvtv2_en` rename nor the exclusions of the single-GPU branch -- and the official GMFlow checkpoint goes
    under `module.GMFlow.`."""
    checkpoint = _read(checkpoint)
    model_dict = wrapped.state_dict()
    pretrained = {'module.' + k: v for k, v in checkpoint.items() if 'module.' + k in model_dict}
    model_dict.update(pretrained)
    loaded_flow = []
    if flow_checkpoint is not None:
        flow = _read(flow_checkpoint)
        flow_dict = {'module.GMFlow.' + k: v for k, v in flow['model'].items() if 'module.GMFlow.' + k in model_dict}
        model_dict.update(flow_dict)
        loaded_flow = sorted(flow_dict)
    wrapped.load_state_dict(model_dict)
    return sorted(pretrained), loaded_flow


def load_long_pretrain(model_long, short_checkpoint, freeze=True):
    """train_long.py:391-406: every short-term entry goes under `short_term.`; `injector1.*`, `dr1.*`, `decoder.*` ALSO
    initialise the long branch's own copies; then everything under `short_term` is frozen."""
    checkpoint = _read(short_checkpoint)
    model_dict = model_long.state_dict()
    pretrained = {}
    for k, v in checkpoint.items():
        if 'short_term.' + k in model_dict:
            pretrained['short_term.' + k] = v
        if k.split('.')[0] in ['injector1', 'dr1', 'decoder']:
            pretrained[k] = v
    model_dict.update(pretrained)
    model_long.load_state_dict(model_dict)
    if freeze:
        for name, para in model_long.named_parameters():
            if "short_term" in name:
                para.requires_grad_(False)
    return sorted(pretrained)


def save(model, path, ddp_prefix=False):
    """train.py:90: `torch.save(model.state_dict(), path)`; ddp_prefix=True writes the keys the way a DDP-wrapped reference
    model does (`module.` prefix), which `load_for_inference(..., multi_gpu=True)` strips again."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if ddp_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    torch.save(sd, path)


def _read(obj):
    return torch.load(obj, map_location="cpu") if isinstance(obj, str) else obj
