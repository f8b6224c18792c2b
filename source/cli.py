"""source/cli.py by name: the callbacks the YAML files list under trainer.callbacks (configs/poco.yaml:13-25, configs/profiler.yaml:3).
ppsurf_amd.runner drives the hooks itself and ignores trainer.callbacks; under a real pytorch_lightning these subclass its bars."""
try:                                                    # pragma: no cover  (Lightning is not in the build image)
    from pytorch_lightning.callbacks.progress.tqdm_progress import TQDMProgressBar as _Bar
    from pytorch_lightning.profilers import SimpleProfiler as _Profiler
except Exception:
    class _Bar:
        def __init__(self, *args, **kwargs):
            self.predict_progress_bar = None
            self.test_progress_bar = None

    class _Profiler:
        def __init__(self, *args, **kwargs):
            pass


class PPSProgressBar(_Bar):
    """source/cli.py:14-33: the stock tqdm bar (the reference only widens the postfix)."""


class PPSProfiler(_Profiler):
    """source/cli.py:36-40."""
