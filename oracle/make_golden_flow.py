"""Golden-vector generator for the flow outputs under the well-conditioned filler.  TEST INFRASTRUCTURE ONLY; runs ONLY in
the build container (imports the reference from /root/reference with the placeholder modules of oracle/make_golden.py).

CoUpdater of the reference, eval mode, weights = emip_amd.filler.flow_conditioned(standard filler), input =
emip_amd.filler.textured_pair().  Stored: flow_fw / flow_bw (model/EMIP_short/motion/gmflow/gmflow.py:130-155) sampled every
4th pixel plus full-tensor statistics, the mask sampled every 4th pixel, and how far the reference's own flow moves when only
its thread count changes (the conditioning of the problem).

usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_flow.py [--out tests/golden]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd.filler import filled_state_dict, flow_conditioned, textured_pair  # noqa: E402
from oracle.make_golden import REF, f32, install_placeholders, stats  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    install_placeholders()
    import yaml
    from model.EMIP_short.model import CoUpdater
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "configs.yaml")))
    net = CoUpdater(args=cfg["model"]["args"])
    net.load_state_dict(flow_conditioned(filled_state_dict(net.state_dict(), seed=0)))
    net.eval()
    im1, im2 = textured_pair()
    outs = []
    for th in (8, 3):
        torch.set_num_threads(th)
        with torch.no_grad():
            m, fw, bw = net(im1, im2)
        outs.append((m, fw[0], bw[0]))
    m, fw, bw = outs[0]
    sens = max((outs[0][1] - outs[1][1]).abs().max().item(), (outs[0][2] - outs[1][2]).abs().max().item())
    print("flow repeatability across thread counts: %.2e px; |flow| up to %.1f px" % (sens, max(fw.abs().max().item(), bw.abs().max().item())))
    np.savez_compressed(os.path.join(args.out, "short_eval_flow.npz"), fw=f32(fw[:, :, ::4, ::4]), bw=f32(bw[:, :, ::4, ::4]),
                        fw_stats=stats(fw), bw_stats=stats(bw), mask=f32(m[:, :, ::4, ::4]), mask_stats=stats(m),
                        thread_sensitivity_px=np.float64(sens))


if __name__ == "__main__":
    main()
