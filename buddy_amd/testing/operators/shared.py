"""Operator base class -- reference ``testing/operators/shared.py:5-28``."""
import abc

import torch.nn as nn


class Operator(nn.Module):
    @abc.abstractmethod
    def degradation(self, *args, **kwargs):
        """Forward pass of the degradation with the current parameters."""

    @abc.abstractmethod
    def update_params(self, *args, **kwargs):
        """Update parameters (blind scenario / new settings)."""

    def prepare_optimization(self, x_den, y):
        return x_den, y

    def constrain_params(self):
        pass
