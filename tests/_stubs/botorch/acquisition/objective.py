"""Callable wrappers of ``botorch.acquisition.objective`` (published interface: ``forward(samples, X=None)``)."""

from __future__ import annotations

import torch


class MCAcquisitionObjective(torch.nn.Module):
    def forward(self, samples, X=None):  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self, samples, X=None):
        return self.forward(samples, X)


class IdentityMCObjective(MCAcquisitionObjective):
    def forward(self, samples, X=None):
        return samples.squeeze(-1)


class GenericMCObjective(MCAcquisitionObjective):
    def __init__(self, objective):
        super().__init__()
        self.objective = objective

    def forward(self, samples, X=None):
        try:
            return self.objective(samples, X)
        except TypeError:
            return self.objective(samples)


class LinearMCObjective(MCAcquisitionObjective):
    def __init__(self, weights):
        super().__init__()
        self.weights = weights

    def forward(self, samples, X=None):
        return samples @ self.weights.to(samples)


class PosteriorTransform(torch.nn.Module):
    pass


class ScalarizedPosteriorTransform(PosteriorTransform):
    def __init__(self, weights, offset: float = 0.0):
        super().__init__()
        self.weights, self.offset = weights, offset
