"""``fdiff.utils.dataclasses`` mirror (reference: src/fdiff/utils/dataclasses.py:7-18; the training-side collate_batch is out of scope)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class DiffusableBatch:
    X: torch.Tensor
    y: Optional[torch.Tensor] = None
    timesteps: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return len(self.X)

    @property
    def device(self) -> torch.device:
        return self.X.device
