"""emip_amd: the EMIP two-stream hot path (PVTv2 + frozen GMFlow + prompting + decoder, its losses and training step)
on MI355X, behind the reference's own nn.Module surfaces.

Drop-in use from the reference's drivers: the drivers import

    from model.EMIP_short.model import CoUpdater as Network        (train.py:25, test.py:12, test_of.py:9)
    from model.EMIP_long.model_long import Model_long as Network   (train_long.py:26, test_long.py:12)
    from loss.loss_pred import hybrid_e_loss                       (train.py:28, train_long.py:29)
    from loss.loss_flow import unFlowLoss                          (train.py:29)
    from utils.utils import clip_gradient                          (train.py:26)
    import eval.metrics as Measure                                 (train.py:24)

`install_aliases()` registers those module names in `sys.modules` as aliases of the emip_amd packages, so the import
lines resolve to this implementation without editing the driver; `python -m emip_amd.run train.py --config ...` does it
and then runs the script.  Packages the drivers import that are not part of the path (dataset.*, tensorboardX, ...)
are left alone and resolve to the reference's own files.
"""
import importlib
import sys

# reference module name -> emip_amd module that carries the same public names
ALIASES = {
    "model": "emip_amd.model",
    "model.EMIP_short": "emip_amd.model.EMIP_short",
    "model.EMIP_short.model": "emip_amd.model.EMIP_short.model",
    "model.EMIP_short.create_backbone": "emip_amd.model.EMIP_short.create_backbone",
    "model.EMIP_short.motion": "emip_amd.model.EMIP_short.motion",
    "model.EMIP_short.motion.PromptInteract": "emip_amd.model.EMIP_short.motion.PromptInteract",
    "model.EMIP_short.motion.common": "emip_amd.model.EMIP_short.motion.common",
    "model.EMIP_short.motion.gmflow": "emip_amd.model.EMIP_short.motion.gmflow",
    "model.EMIP_short.motion.gmflow.gmflow": "emip_amd.model.EMIP_short.motion.gmflow.gmflow",
    "model.EMIP_short.motion.gmflow.backbone": "emip_amd.model.EMIP_short.motion.gmflow.backbone",
    "model.EMIP_short.motion.gmflow.transformer": "emip_amd.model.EMIP_short.motion.gmflow.transformer",
    "model.EMIP_long": "emip_amd.model.EMIP_long",
    "model.EMIP_long.model_long": "emip_amd.model.EMIP_long.model_long",
    "model.EMIP_long.LTM": "emip_amd.model.EMIP_long.LTM",
    "lib": "emip_amd.lib",
    "lib.pvt_v2": "emip_amd.lib.pvt_v2",
    "loss": "emip_amd.loss",
    "loss.loss_pred": "emip_amd.loss.loss_pred",
    "loss.loss_flow": "emip_amd.loss.loss_flow",
    "loss.warp_utils": "emip_amd.loss.warp_utils",
    "utils": "emip_amd.utils",
    "utils.utils": "emip_amd.utils.utils",
    "eval": "emip_amd.eval_pkg",
    "eval.metrics": "emip_amd.eval_metrics",
}


def install_aliases(force=False):
    """Make the reference's import lines resolve to emip_amd.  Idempotent; with force=False a module of that name that
    is ALREADY imported from somewhere else is a conflict and raises (mixing two implementations silently is worse)."""
    done = []
    for name, target in ALIASES.items():
        mod = importlib.import_module(target)
        cur = sys.modules.get(name)
        if cur is not None and cur is not mod and not force:
            raise ImportError(f"'{name}' is already imported from {getattr(cur, '__file__', '?')}; call "
                              "emip_amd.install_aliases() before the driver's own imports (or pass force=True)")
        sys.modules[name] = mod
        done.append(name)
    # attribute access on the parents (import model.EMIP_short.model; model.EMIP_short.model.CoUpdater)
    for name in done:
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    return done
