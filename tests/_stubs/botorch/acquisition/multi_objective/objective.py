"""``GenericMCMultiOutputObjective`` of the test stand-in: applies the given callable."""

from botorch.acquisition.objective import GenericMCObjective, MCAcquisitionObjective  # noqa: F401


class MCMultiOutputObjective(MCAcquisitionObjective):
    pass


class GenericMCMultiOutputObjective(GenericMCObjective, MCMultiOutputObjective):
    pass
