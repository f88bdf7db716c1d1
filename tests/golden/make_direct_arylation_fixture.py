"""Writes ``tests/golden/direct_arylation_comp.npz``: the computational representation of the reference's direct-arylation benchmark
domain with CATEGORICAL (one-hot) encodings - ``/root/reference/benchmarks/domains/direct_arylation/convergence.py:33-72``, scenario
"Categorical": ``SearchSpace.from_product`` of three ``CategoricalParameter``s (Solvent, Base, Ligand) and two
``NumericalDiscreteParameter``s (Concentration, Temp_C) - together with the measured yield of every row from the lookup table
``/root/reference/benchmarks/data/direct_arylation/data.csv`` (1 728 reactions).  Built with the reference's own ``SearchSpace`` in
the build container (``tests/_reference.py``); the GPU box, where the reference tree does not exist, closes the optimisation loop on
these arrays (``tests/test_benchmark_domain_gpu.py``).  The RDKit / Mordred scenarios of the same benchmark need chemistry packages
that are not installed."""

import sys
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))


def build():
    from _reference import reference_baybe

    reference_baybe()
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter
    from baybe.searchspace import SearchSpace

    data = pd.read_table("/root/reference/benchmarks/data/direct_arylation/data.csv", sep=",", index_col=0)
    params = [CategoricalParameter(name=s, values=data[s].unique()) for s in ["Solvent", "Base", "Ligand"]] + [
        NumericalDiscreteParameter(name="Concentration", values=sorted(data["Concentration"].unique())),
        NumericalDiscreteParameter(name="Temp_C", values=sorted(data["Temp_C"].unique()))]
    space = SearchSpace.from_product(parameters=params)
    exp, comp = space.discrete.exp_rep, space.discrete.comp_rep
    cols = [p.name for p in params]
    merged = exp.merge(data[cols + ["yield"]], on=cols, how="left")
    assert len(merged) == len(exp) == 1728 and not merged["yield"].isna().any()
    bounds = space.scaling_bounds[list(comp.columns)].to_numpy(dtype=np.float64)
    return data, space, dict(comp=comp.to_numpy(dtype=np.float64), columns=np.array(list(comp.columns)), bounds=bounds,
                             y=merged["yield"].to_numpy(dtype=np.float64))


if __name__ == "__main__":
    _, _, arrays = build()
    out = HERE / "direct_arylation_comp.npz"
    np.savez_compressed(out, **arrays)
    print(f"wrote {out}: comp {arrays['comp'].shape}, optimum {arrays['y'].max()}, {out.stat().st_size} bytes")
