"""On the MI355X: the ``recommend()`` calls the REFERENCE's own ``Campaign`` / ``TwoPhaseMetaRecommender`` / ``simulate_experiment``
made to the plug-in (recorded in the build container by ``tests/golden/make_reference_traces.py`` - the reference tree does not exist
on this box) replayed through ``libbaybe_hip.so``: same search-space arrays, keep-masks, measurements, pending rows, batch sizes and
torch RNG states, one recommender object per scenario (so fit caching, the resident candidate matrix and the shrinking masks behave as
in the campaign).  Expected: the recorded index labels, which are the oracle's (``tests/test_reference_campaign_cpu.py``)."""

import numpy as np
import pytest

from _replay import check_repins, load_traces, replay

pytestmark = pytest.mark.gpu

META, DATA = load_traces()


@pytest.mark.parametrize("name", sorted(META))
def test_reference_recorded_calls_on_the_device(name):
    from baybe_amd.recommenders import HipBotorchRecommender

    rec = HipBotorchRecommender()
    results = replay(rec, META[name], DATA)
    assert len(results) == len(META[name]) >= 1
    for i, (want, got) in enumerate(results):
        assert want == got, f"{name} call {i}: device picked {got}, the reference run (oracle double) {want}"
    # calls compared on the RECORDED hyper-parameters (tests/_replay.py::recommend_on_recorded_fit): counted, recorded, allow-listed
    check_repins("traces", name, replay.repinned, len(results))
    model = rec._surrogate_model
    engines = [m.engine for m in model.models] if hasattr(model, "models") else [model.engine]
    assert all(e._handle is not None for e in engines)  # the picks came through the C-ABI


def test_replay_keeps_the_candidate_matrix_resident():
    """One upload per search space over the calls of a scenario (the campaign's masks shrink, the matrix stays)."""
    from baybe_amd.recommenders import HipBotorchRecommender

    rec = HipBotorchRecommender()
    seen = []
    replay(rec, META["cfg1_max"], DATA, on_call=lambda c, got: seen.append(rec._cand_cache[1].data_ptr()))
    assert len(set(seen)) == 1 and len(seen) == 3
