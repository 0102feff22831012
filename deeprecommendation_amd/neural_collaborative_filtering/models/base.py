"""Model contract of the reference (models/base.py:6-42): forward / get_model_parameters / save_model /
is_dataset_compatible / important_hypeparams, and the ``[state_dict, kwargs]`` checkpoint format."""
from abc import abstractmethod

import torch
from torch import nn


def _dataset_names(dataset_class):
    return {c.__name__ for c in getattr(dataset_class, "__mro__", ())}


class NCF(nn.Module):
    # names of the dataset classes (reference's or this package's — matched by class name through the MRO so that
    # the reference's own FixedPointwiseDataset etc. are accepted without importing the reference)
    compatible_datasets = ()

    def __init__(self):
        super().__init__()

    @abstractmethod
    def forward(self, *args):
        raise NotImplementedError

    @abstractmethod
    def get_model_parameters(self) -> dict:
        raise NotImplementedError

    def save_model(self, file):
        torch.save([self.state_dict(), self.get_model_parameters()], file)  # reference models/base.py:18-19

    def is_dataset_compatible(self, dataset_class):
        return bool(_dataset_names(dataset_class) & set(self.compatible_datasets))

    def important_hypeparams(self) -> str:
        return ''


class GNN_NCF(NCF):
    @abstractmethod
    def forward(self, *args, **kwargs):
        """forward(graph, userIds (B,), itemIds (B,), device, mask_targets=True) -> (B, 1)"""
        raise NotImplementedError
