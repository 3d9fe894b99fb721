"""Checkpoint key handling of the reference drivers (train.py:280-293,312-342, test.py:81-89, test_long.py:92-100,
train_long.py:391-406), host only.  The pin: tests/golden/ckpt_maps.json.gz holds, per scenario, the key map the
REFERENCE'S OWN STATEMENTS produced (oracle/make_golden_ckpt.py executes those line ranges on key manifests);
emip_amd/checkpoint.py must produce the same map, the same frozen set, and fail where the reference fails."""
import gzip
import json
import os

import pytest
import torch


def _golden_maps():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_maps.json.gz")
    return json.loads(gzip.open(p).read().decode())


def _run_ours(name, s, short_manifest, long_manifest):
    """emip_amd.checkpoint on the stand-in model of the golden generator (values are source tags, not tensors)"""
    from emip_amd import checkpoint as C
    from emip_amd.train import freeze_like_reference, freeze_short_term
    from oracle.make_golden_ckpt import Model
    ck = {k: "ckpt:" + k for k in s["ckpt"]}
    flow = {"model": {k: "flow:" + k for k in s["flow"]}} if s.get("flow") is not None else None
    if name.startswith("train_single"):
        m = Model(short_manifest)
        C.load_short_pretrain(m, ck, flow)
        freeze_like_reference(m)
    elif name.startswith("train_ddp"):
        m = Model(short_manifest, "module.")
        C.load_short_pretrain_ddp(m, ck, flow)
        freeze_like_reference(m)
    elif name.startswith("test_long"):
        m = Model(long_manifest)
        C.load_for_inference(m, ck, multi_gpu="multi_gpu" in name)
    elif name.startswith("test_"):
        m = Model(short_manifest)
        C.load_for_inference(m, ck, multi_gpu="multi_gpu" in name)
    else:
        m = Model(long_manifest)
        C.load_long_pretrain(m, ck, freeze=False)
        freeze_short_term(m)
    return ({k: v for k, v in m.loaded.items() if not v.startswith("init:")},
            sorted(n for n, p in m.named_parameters() if not p.requires_grad))


def test_key_maps_equal_what_the_reference_statements_produce(short_manifest, long_manifest):
    from oracle.make_golden_ckpt import scenarios
    gold = _golden_maps()
    sc = scenarios(short_manifest, long_manifest)
    assert set(gold) == set(sc) and len(gold) >= 11
    for name, s in sc.items():
        exp = gold[name]
        if "error" in exp:                       # the reference's strict load_state_dict raises: so must this side
            with pytest.raises((KeyError, RuntimeError)):
                _run_ours(name, s, short_manifest, long_manifest)
            continue
        got_map, got_frozen = _run_ours(name, s, short_manifest, long_manifest)
        assert got_map == exp["map"], (name, sorted(set(got_map.items()) ^ set(exp["map"].items()))[:6])
        assert got_frozen == exp["frozen"], (name, sorted(set(got_frozen) ^ set(exp["frozen"]))[:6])
    # the scenarios are not vacuous: renames, the GMFlow prefix, the long branch's copies and both freeze rules occur
    m = gold["train_single_pretrain"]["map"]
    assert any(v.startswith("ckpt:backbone.pvtv2_en.") and k.startswith("backbone.feat_net.pvtv2_en.") for k, v in m.items())
    assert any(v.startswith("flow:") and k.startswith("GMFlow.") for k, v in m.items())
    assert not any("mask_downscaling" in k or "something.else" in k for k in m)
    lm = gold["train_long_from_short"]["map"]
    assert lm["decoder.conv5.weight"] == lm["short_term.decoder.conv5.weight"] == "ckpt:decoder.conv5.weight"
    assert gold["test_plain_given_ddp_file"]["map"] == {} and len(gold["test_multi_gpu"]["map"]) == 1438
    assert len(gold["train_single_full"]["frozen"]) == 123 and all("GMFlow" in n for n in gold["train_single_full"]["frozen"])


def _short(model_args):
    from emip_amd.model.EMIP_short.model import CoUpdater
    return CoUpdater(model_args)


def test_inference_load_roundtrip_and_ddp_prefix(tmp_path, model_args, short_sd):
    from emip_amd import checkpoint as C
    a = _short(model_args)
    a.load_state_dict(short_sd)
    p = str(tmp_path / "Net_epoch_best.pth")
    C.save(a, p, ddp_prefix=True)
    b = _short(model_args)
    taken = C.load_for_inference(b, p, multi_gpu=True)
    assert len(taken) == len(short_sd)
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    # without the multi_gpu flag the prefixed keys are ignored (test.py:87): the model keeps its own initial values
    c = _short(model_args)
    before = {k: v.clone() for k, v in c.state_dict().items()}
    assert C.load_for_inference(c, p, multi_gpu=False) == []
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in before.items())


def test_short_pretrain_renames_and_flow_prefix(model_args, short_sd):
    from emip_amd import checkpoint as C
    m = _short(model_args)
    pre = {}
    for k, v in short_sd.items():
        if k.startswith("backbone.feat_net.pvtv2_en."):                     # the pre-train file uses the old prefix
            pre[k.replace("backbone.feat_net.pvtv2_en", "backbone.pvtv2_en")] = v + 1.0 if v.is_floating_point() else v
        elif k.startswith("decoder."):
            pre[k] = v + 2.0 if v.is_floating_point() else v
    pre["mask_downscaling.0.weight"] = torch.zeros(3)                        # excluded / unknown entries are dropped
    pre["something.else"] = torch.zeros(1)
    flow = {"model": {k[len("GMFlow."):]: (v + 3.0 if v.is_floating_point() else v) for k, v in short_sd.items()
                      if k.startswith("GMFlow.transformer.")}}
    flow["model"]["not.in.model"] = torch.zeros(2)
    base = {k: v.clone() for k, v in m.state_dict().items()}
    taken, taken_flow = C.load_short_pretrain(m, pre, flow)
    sd = m.state_dict()
    k1 = "backbone.feat_net.pvtv2_en.block1.0.attn.q.weight"
    assert torch.equal(sd[k1], short_sd[k1] + 1.0) and k1 in taken
    assert torch.equal(sd["decoder.conv5.weight"], short_sd["decoder.conv5.weight"] + 2.0)
    k2 = "GMFlow.transformer.layers.0.self_attn.q_proj.weight"
    assert torch.equal(sd[k2], short_sd[k2] + 3.0) and k2 in taken_flow
    assert torch.equal(sd["dr1.reduce.0.conv.weight"], base["dr1.reduce.0.conv.weight"])     # untouched
    assert "something.else" not in taken and all("mask_downscaling" not in k for k in taken)


def test_long_pretrain_copies_and_freezes(model_args, short_sd, long_sd):
    from emip_amd import checkpoint as C
    from emip_amd.model.EMIP_long.model_long import Model_long
    m = Model_long(model_args)
    m.load_state_dict(long_sd)
    bumped = {k: (v + 0.5 if v.is_floating_point() else v) for k, v in short_sd.items()}
    taken = C.load_long_pretrain(m, bumped)
    sd = m.state_dict()
    assert torch.equal(sd["short_term.decoder.conv5.weight"], short_sd["decoder.conv5.weight"] + 0.5)
    assert torch.equal(sd["decoder.conv5.weight"], short_sd["decoder.conv5.weight"] + 0.5)      # the long branch's copy
    assert torch.equal(sd["injector1.transformer.attn.q.weight"], short_sd["injector1.transformer.attn.q.weight"] + 0.5)
    assert torch.equal(sd["LTM.KV_M_r4.Key.weight"], long_sd["LTM.KV_M_r4.Key.weight"])        # not in the short file
    assert any(k.startswith("short_term.") for k in taken) and "dr1.reduce.0.conv.weight" in taken
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    assert frozen and all("short_term" in n for n in frozen)
    assert all(p.requires_grad for n, p in m.named_parameters() if "short_term" not in n)
