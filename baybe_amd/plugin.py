"""BayBE's own types for the HIP path: genuine subclasses of ``baybe.surrogates.base.Surrogate`` and
``baybe.recommenders.pure.bayesian.base.BayesianRecommender``.

Why subclasses: ``Campaign.get_surrogate`` / ``acquisition_values`` demand
``isinstance(recommender, BayesianRecommender)`` (``campaign.py:750-807``), the serialisation registry finds
subclasses only (``serialization/core.py:109-146``), and BayBE's test loops enumerate subclasses and construct them as
``cls()`` (``tests/test_iterations.py:75-105``).

How: ``Surrogate`` is a *slotted* attrs class (``surrogates/base.py:81-82``); a second slotted attrs base cannot be
mixed in ("multiple bases have instance lay-out conflict").  The behaviour therefore lives in field-less mixins
with empty ``__slots__`` (``baybe_amd.surrogates.HipGPSurrogateImpl``, ``baybe_amd.recommenders.HipRecommenderImpl``)
and ``attrs.make_class`` attaches the fields on top of BayBE's base, which keeps contributing its own runtime
fields (``_searchspace``, ``_objective``, ``_measurements_hash``; ``acquisition_function``, ``_objective``).  The
mixin comes first in the MRO, so its ``fit`` / ``_setup_botorch_acqf`` / ``_recommend_discrete`` ... override the
base's, while ``BayesianRecommender.recommend`` itself (validation, dataframe preprocessing, ``Settings`` context,
``pure/bayesian/base.py:130-197``) is BayBE's own code driving those overrides.

``tests/test_plugin_layout_cpu.py`` builds the classes on replicas of the two bases that copy their attrs / slots
layout and call flow (BayBE itself cannot be imported in the build container: no ``cattrs``).
"""

from __future__ import annotations

import attrs

from baybe_amd.recommenders import HipRecommenderImpl, recommender_fields
from baybe_amd.surrogates import HipCompositeImpl, HipGPSurrogateImpl, composite_fields, gp_surrogate_fields


def make_baybe_classes(surrogate_base=None, recommender_base=None, discrete_compatibility=None):
    """(HipGaussianProcessSurrogate, HipCompositeSurrogate, HipBotorchRecommender) as subclasses of BayBE's bases.

    Without arguments the bases are imported from ``baybe``; the parameters exist so that the layout can be tested
    against replicas of those bases where ``baybe`` is not importable."""
    if surrogate_base is None:
        from baybe.surrogates.base import Surrogate as surrogate_base
    if recommender_base is None:
        from baybe.recommenders.pure.bayesian.base import BayesianRecommender as recommender_base
    if discrete_compatibility is None:
        from baybe.searchspace.core import SearchSpaceType

        discrete_compatibility = SearchSpaceType.HYBRID  # discrete, hybrid and (through the hybrid fallback) continuous spaces

    surrogate = attrs.make_class("HipGaussianProcessSurrogate", gp_surrogate_fields(with_runtime_state=False),
                                 bases=(HipGPSurrogateImpl, surrogate_base), slots=True)
    surrogate.__doc__ = "A Gaussian process surrogate evaluated on an MI355X (``baybe.surrogates.base.Surrogate``)."
    # the per-target replication of the HIP path keeps the engines resident; it is not BayBE's CompositeSurrogate
    # (deep copies of a template wrapped into a BoTorch ModelListGP, surrogates/composite.py:101-134)
    composite = attrs.make_class("HipCompositeSurrogate", composite_fields(surrogate), bases=(HipCompositeImpl,), slots=True)
    surrogate._composite_class = composite

    recommender = attrs.make_class(
        "HipBotorchRecommender", recommender_fields(with_base_fields=False, surrogate_factory=surrogate),
        bases=(HipRecommenderImpl, recommender_base), kw_only=True, slots=False)
    recommender.__doc__ = ("Bayesian recommender scoring the full discrete candidate set on an MI355X "
                           "(``baybe.recommenders.pure.bayesian.base.BayesianRecommender``).")
    recommender.compatibility = discrete_compatibility
    # BayBE's recommend() (argument validation, dataframe preprocessing, Settings context) stays in charge; it calls
    # _setup_botorch_acqf and, through PureRecommender.recommend, _recommend_with_discrete_parts - both overridden
    recommender.recommend = recommender_base.recommend
    for cls in (surrogate, composite, recommender):
        cls.__module__ = __name__
    return surrogate, composite, recommender
