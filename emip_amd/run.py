"""Run one of the reference's driver scripts UNEDITED on the MI355X implementation:

    python -m emip_amd.run train.py --config configs/configs.yaml
    python -m emip_amd.run test.py --snap_path ...

Installs the module aliases (emip_amd.install_aliases) and then executes the script as __main__ with its own argv, from
its own directory on sys.path -- exactly what `python train.py ...` does, except that `model.*`, `loss.*`, `lib.*`,
`utils.utils` and `eval.metrics` are this implementation."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    import emip_amd
    emip_amd.install_aliases()
    script = argv[0]
    sys.argv = argv
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
