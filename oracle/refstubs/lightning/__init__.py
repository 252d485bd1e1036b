"""Import-only stand-in for lightning (absent here), so that the reference's flowmap/model/model_wrapper_overfit.py can be imported by the
tests of flowmap_amd.install(): a LightningModule that is a torch module with what training_step touches — `log`, `global_step` — and nothing
of a trainer.  Test infrastructure (tests/test_install_reference.py); see oracle/make_golden.py for the other stand-ins."""
from torch import nn


class LightningModule(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.global_step = 0
        self.logged = {}

    def log(self, name, value, **kwargs) -> None:
        self.logged[name] = value.detach() if hasattr(value, "detach") else value
