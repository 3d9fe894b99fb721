"""The reference's own import lines resolve to this implementation (SURVEY.md section 8b "import paths to preserve"):
train.py:24-29, test.py:12, train_long.py:26-29, test_long.py:12 -- through emip_amd.install_aliases() and through the
`python -m emip_amd.run <script>` launcher, which runs a driver script unedited."""
import os
import subprocess
import sys
import textwrap

import torch

from tests.conftest import ROOT

# the import block of the reference's drivers, as a maintainer's script has it (names only: the interface under test)
DRIVER_IMPORTS = textwrap.dedent("""
    import eval.metrics as Measure
    from model.EMIP_short.model import CoUpdater as Network
    from utils.utils import clip_gradient
    from loss.loss_pred import hybrid_e_loss
    from loss.loss_flow import unFlowLoss
    from model.EMIP_long.model_long import Model_long
""")


def _run(code, *argv):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, *argv], input=code, capture_output=True, text=True, env=env, timeout=300)


def test_reference_import_lines_resolve_to_emip_amd():
    code = "import emip_amd; emip_amd.install_aliases()\n" + DRIVER_IMPORTS + textwrap.dedent("""
        import emip_amd.model.EMIP_short.model as a, emip_amd.model.EMIP_long.model_long as b
        import emip_amd.loss.loss_pred as c, emip_amd.loss.loss_flow as d, emip_amd.utils.utils as e, emip_amd.eval_metrics as f
        assert Network is a.CoUpdater and Model_long is b.Model_long
        assert hybrid_e_loss is c.hybrid_e_loss and unFlowLoss is d.unFlowLoss and clip_gradient is e.clip_gradient
        assert Measure.MAE is f.MAE and Measure.Smeasure is f.Smeasure and Measure.WeightedFmeasure is f.WeightedFmeasure
        import model.EMIP_short.model, lib.pvt_v2
        assert model.EMIP_short.model.CoUpdater is Network and lib.pvt_v2.pvt_v2_b5 is not None
        print("ok")
    """)
    r = _run(code, "-")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_launcher_runs_an_unedited_driver_script(tmp_path):
    """python -m emip_amd.run script.py args: the script's own directory leads sys.path (as for `python script.py`), the
    aliases win over a same-named package lying next to the script, argv is the script's"""
    (tmp_path / "model").mkdir()
    (tmp_path / "model" / "__init__.py").write_text("raise ImportError('the local model package must not be imported')\n")
    (tmp_path / "dataset").mkdir()
    (tmp_path / "dataset" / "__init__.py").write_text("")
    (tmp_path / "dataset" / "dataset.py").write_text("def get_loader():\n    return 'the reference loader'\n")
    script = tmp_path / "train.py"
    script.write_text(DRIVER_IMPORTS + textwrap.dedent("""
        import sys
        from dataset.dataset import get_loader            # not part of the path: resolves next to the script
        if __name__ == '__main__':
            print(Network.__module__, Model_long.__module__, get_loader(), sys.argv[1:])
    """))
    r = _run(None, "-m", "emip_amd.run", str(script), "--config", "x.yaml")
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["emip_amd.model.EMIP_short.model", "emip_amd.model.EMIP_long.model_long", "the", "reference",
                                "loader", "['--config',", "'x.yaml']"], r.stdout


def test_install_aliases_refuses_to_shadow_an_imported_package(tmp_path):
    (tmp_path / "loss").mkdir()
    (tmp_path / "loss" / "__init__.py").write_text("")
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {str(tmp_path)!r})
        import loss
        import emip_amd
        try:
            emip_amd.install_aliases()
        except ImportError as e:
            print("refused")
        emip_amd.install_aliases(force=True)
        from loss.loss_pred import hybrid_e_loss
        print(hybrid_e_loss.__module__)
    """)
    r = _run(code, "-")
    assert r.returncode == 0 and r.stdout.split() == ["refused", "emip_amd.loss.loss_pred"], r.stdout + r.stderr[-1500:]


def test_clip_gradient_is_the_elementwise_clamp():
    """utils/utils.py:1-11: clamp, not a norm clip; parameters without a gradient are skipped"""
    from emip_amd.utils.utils import clip_gradient
    a, b, c = (torch.nn.Parameter(torch.zeros(5)) for _ in range(3))
    a.grad = torch.tensor([-2.0, -0.5, 0.0, 0.25, 3.0])
    b.grad = torch.full((5,), 0.4)
    opt = torch.optim.AdamW([{"params": [a, c]}, {"params": [b]}], lr=1e-5)
    clip_gradient(opt, 0.5)
    assert torch.equal(a.grad, torch.tensor([-0.5, -0.5, 0.0, 0.25, 0.5])) and torch.equal(b.grad, torch.full((5,), 0.4))
    assert c.grad is None
