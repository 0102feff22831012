"""GraphPointwiseDataset (reference datasets/gnn_datasets.py:6-29): samples are (user node id, item node id, target)."""
import numpy as np
import torch

from .base import PointwiseDataset, ResidentInputs


class GraphPointwiseDataset(PointwiseDataset):
    def __init__(self, file_or_frame, graph_content_provider):
        super().__init__(file_or_frame)
        self.gcp = graph_content_provider
        # node ids for the whole file at once (the reference resolves them per sample through two dicts)
        self._unode = np.asarray(self.gcp.get_user_nodeID(self._u))
        self._inode = np.asarray(self.gcp.get_item_nodeID(self._i))

    def __getitem__(self, item):
        return self._unode[item], self._inode[item], self._r[item]

    def resident_inputs(self, device=None, batch_size=None):
        return ResidentInputs((torch.as_tensor(self._unode, dtype=torch.int64), torch.as_tensor(self._inode, dtype=torch.int64)), self._targets())

    def get_graph(self, device):
        return self.gcp.get_graph().to(device)

    def use_collate(self):
        def collate(batch):
            u, i, t = zip(*batch)
            return torch.as_tensor(np.asarray(u), dtype=torch.int64), torch.as_tensor(np.asarray(i), dtype=torch.int64), torch.as_tensor(np.asarray(t), dtype=torch.float32)
        return collate

    @staticmethod
    def do_forward(model, batch, device, graph, *args):
        userIds, itemIds, y_batch = batch
        return model(graph.to(device), userIds.long().to(device), itemIds.long().to(device), device, *args), y_batch
