"""What every scoring model of this package exposes to the datasets / training / evaluation loops.

The contract is the upstream project's (forward, get_model_parameters, save_model, is_dataset_compatible,
important_hypeparams; checkpoints are the two-element list ``[state_dict, kwargs]``) so that models written against it
and checkpoints saved by it interchange.
"""
import torch
from torch import nn


class NCF(nn.Module):
    """Base of BasicNCF / MF / AttentionNCF.

    ``compatible_datasets`` lists dataset class NAMES; compatibility is decided by name along the candidate's MRO, so
    the upstream project's own dataset classes are accepted without importing that project.
    """

    compatible_datasets: tuple = ()

    def forward(self, *inputs):  # pragma: no cover - abstract
        raise NotImplementedError(f"{type(self).__name__}.forward")

    def get_model_parameters(self) -> dict:  # pragma: no cover - abstract
        raise NotImplementedError(f"{type(self).__name__}.get_model_parameters")

    def is_dataset_compatible(self, dataset_class) -> bool:
        names = {klass.__name__ for klass in getattr(dataset_class, "__mro__", ())}
        return not names.isdisjoint(self.compatible_datasets)

    def important_hypeparams(self) -> str:
        return ""

    def save_model(self, file) -> None:
        """Checkpoint = [state_dict, constructor kwargs]; ``util.load_model`` reads it back."""
        torch.save([self.state_dict(), self.get_model_parameters()], file)


class GNN_NCF(NCF):
    """Base of graph models: ``forward(graph, userIds (B,), itemIds (B,), device, mask_targets=True) -> (B, 1)``."""
