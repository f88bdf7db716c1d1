"""Exception types at the plug-in boundary.  When BayBE is importable its own classes are
re-exported (``baybe/exceptions.py:50-178``) so that ``Campaign.recommend`` catches exactly what
it expects (``campaign.py:602-630``); otherwise same-named stand-ins are defined."""

try:  # pragma: no cover - baybe is not importable in the build container
    from baybe.exceptions import (  # type: ignore
        IncompatibilityError,
        IncompatibleAcquisitionFunctionError,
        IncompatibleSurrogateError,
        ModelNotTrainedError,
        NotEnoughPointsLeftError,
    )
except Exception:  # noqa: BLE001

    class IncompatibilityError(Exception):
        """Incompatible components are used together."""

    class IncompatibleSurrogateError(IncompatibilityError):
        """An incompatible surrogate was selected."""

    class IncompatibleAcquisitionFunctionError(IncompatibilityError):
        """An incompatible acquisition function was selected."""

    class NotEnoughPointsLeftError(Exception):
        """More recommendations are requested than there are viable candidates left."""

    class ModelNotTrainedError(Exception):
        """A model was used before being trained."""


from baybe_amd.engine import ModelFittingError  # noqa: E402,F401
