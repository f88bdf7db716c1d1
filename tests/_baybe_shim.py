"""Minimal stand-ins for the BayBE data model (TEST INFRASTRUCTURE) for boxes WITHOUT the reference tree.

The reference's own ``Campaign`` / ``SearchSpace`` run in the build container (``tests/_reference.py``;
``tests/test_reference_campaign_cpu.py`` drives the plug-in with them and pins THIS file to them:
``test_shim_campaign_makes_the_calls_the_real_campaign_makes``).  ``/root/reference`` does not exist on the GPU box, so
the ``-m gpu`` drop-in tests drive the HIP recommender through objects that expose exactly the attributes the real
classes expose on this path:
  SearchSpace      .discrete .continuous .parameters .transform() .scaling_bounds .task_idx .n_tasks
                   (baybe/searchspace/core.py:246-295, 469-515)
  SubspaceDiscrete .exp_rep .comp_rep .get_candidates() .n_subsets (baybe/searchspace/discrete.py:695-702)
  NumericalTarget / SingleTargetObjective  .name .minimize .targets
  Campaign         recommend(): keep-mask from recommended/measured/pending rows, recommender call,
                   metadata update (baybe/campaign.py:495-642, default flags 267-285)
"""

from __future__ import annotations

import itertools

import numpy as np
import pandas as pd


class NumericalDiscreteParameter:
    is_numerical, is_discrete, is_continuous = True, True, False  # baybe/parameters/base.py flags

    def __init__(self, name, values):
        self.name, self.values = name, tuple(float(v) for v in values)

    @property
    def comp_bounds(self):
        return min(self.values), max(self.values)


class TaskParameter:
    """INT-encoded task column (baybe/parameters/categorical.py:86-91)."""

    is_numerical, is_discrete, is_continuous = False, True, False

    def __init__(self, name, values, active_values=None):
        self.name, self.values = name, tuple(values)
        self.active_values = tuple(active_values) if active_values is not None else tuple(values)

    def encode(self, col):
        return col.map({v: float(i) for i, v in enumerate(self.values)})


class _EmptyContinuous:
    is_empty = True


class SubspaceDiscrete:
    n_subsets = 0

    def __init__(self, parameters, exp_rep, mask_keep=None):
        self.parameters = parameters
        self.exp_rep = exp_rep
        comp = pd.DataFrame(index=exp_rep.index)
        for p in parameters:
            comp[p.name] = p.encode(exp_rep[p.name]) if isinstance(p, TaskParameter) else exp_rep[p.name].astype(float)
        self.comp_rep = comp
        self._mask = mask_keep

    @property
    def mask_keep(self):
        """``FilteredSubspaceDiscrete.mask_keep`` (searchspace/_filtered.py:17-22); None when unfiltered."""
        return self._mask

    def filtered(self, mask_keep):
        sub = SubspaceDiscrete.__new__(SubspaceDiscrete)
        sub.parameters, sub.exp_rep, sub.comp_rep, sub._mask = self.parameters, self.exp_rep, self.comp_rep, mask_keep
        return sub

    def get_candidates(self):
        if self._mask is None:
            return self.exp_rep, self.comp_rep
        return self.exp_rep.loc[self._mask], self.comp_rep.loc[self._mask]


class SearchSpace:
    def __init__(self, discrete):
        self.discrete = discrete
        self.continuous = _EmptyContinuous()

    @classmethod
    def from_product(cls, parameters):
        cols = [p.name for p in parameters]
        vals = [p.active_values if isinstance(p, TaskParameter) else p.values for p in parameters]
        exp = pd.DataFrame(list(itertools.product(*vals)), columns=cols)
        return cls(SubspaceDiscrete(parameters, exp))

    @classmethod
    def from_dataframe(cls, df):
        params = [NumericalDiscreteParameter(c, sorted(set(df[c]))) for c in df.columns]
        return cls(SubspaceDiscrete(params, df.reset_index(drop=True)))

    @property
    def parameters(self):
        return self.discrete.parameters

    @property
    def comp_rep_columns(self):
        return tuple(p.name for p in self.parameters)

    @property
    def task_idx(self):
        idx = [i for i, p in enumerate(self.parameters) if isinstance(p, TaskParameter)]
        return idx[0] if idx else None

    @property
    def n_tasks(self):
        t = [p for p in self.parameters if isinstance(p, TaskParameter)]
        return len(t[0].values) if t else 1

    @property
    def scaling_bounds(self):
        lo, hi = [], []
        for p in self.parameters:
            if isinstance(p, TaskParameter):
                lo.append(0.0), hi.append(float(len(p.values) - 1))
            else:
                a, b = p.comp_bounds
                lo.append(a), hi.append(b)
        return pd.DataFrame([lo, hi], index=["min", "max"], columns=self.comp_rep_columns)

    def transform(self, df, allow_extra=False):
        out = pd.DataFrame(index=df.index)
        for p in self.parameters:
            out[p.name] = p.encode(df[p.name]) if isinstance(p, TaskParameter) else df[p.name].astype(float)
        return out

    def filtered(self, mask_keep):
        return SearchSpace(self.discrete.filtered(mask_keep))


class NumericalTarget:
    def __init__(self, name, minimize=False):
        self.name, self.minimize = name, minimize


class SingleTargetObjective:
    is_multi_output = False

    def __init__(self, target):
        self._target = target

    @property
    def targets(self):
        return (self._target,)


class ParetoObjective:
    is_multi_output = True

    def __init__(self, targets):
        self._targets = tuple(targets)

    @property
    def targets(self):
        return self._targets


class Campaign:
    """recommend() plumbing of baybe/campaign.py:495-642 for discrete spaces.  Default flags as in the reference
    (campaign.py:254-285): already RECOMMENDED and PENDING rows are excluded from the candidates of a discrete space,
    already MEASURED rows are not (``allow_recommending_already_measured`` resolves to True)."""

    def __init__(self, searchspace, objective, recommender, allow_recommending_already_measured=True):
        self.searchspace, self.objective, self.recommender = searchspace, objective, recommender
        self.allow_recommending_already_measured = allow_recommending_already_measured
        n = len(searchspace.discrete.exp_rep)
        self.measurements = pd.DataFrame()
        self._meta = pd.DataFrame({"recommended": np.zeros(n, bool), "measured": np.zeros(n, bool)},
                                  index=searchspace.discrete.exp_rep.index)

    def _match(self, df):
        """Index labels of the search-space rows matching the rows of ``df`` on the parameter columns it has (all of them
        for measurements; a subset - e.g. only the task column - for candidate toggles, campaign.py:404-470)."""
        exp = self.searchspace.discrete.exp_rep
        cols = [c for c in exp.columns if c in df.columns]
        merged = df[cols].drop_duplicates().merge(exp.reset_index(), on=cols, how="left")
        return merged["index"].dropna().astype(int).to_numpy()

    @property
    def parameters(self):
        return self.searchspace.parameters

    def add_measurements(self, df):
        """campaign.py:330-372: append, and mark the search-space rows the measurements belong to - exact rows where there
        are any, otherwise the nearest rows (``fuzzy_row_match``: numerical values may deviate from the grid)."""
        self.measurements = pd.concat([self.measurements, df], ignore_index=True)
        hits = self._match(df)
        if len(hits) < len(df):
            from baybe_amd.dataframe import fuzzy_row_match

            hits = fuzzy_row_match(self.searchspace.discrete.exp_rep, df, self.searchspace.parameters)
        self._meta.loc[hits, "measured"] = True

    def toggle_discrete_candidates(self, constraints, exclude, complement=False):
        """``Campaign.toggle_discrete_candidates`` (campaign.py:404-470) for a dataframe of rows."""
        hit = np.zeros(len(self._meta), bool)
        hit[self._meta.index.get_indexer(self._match(constraints))] = True
        if complement:
            hit = ~hit
        self._meta["excluded"] = self._meta.get("excluded", False) | hit if exclude else self._meta.get("excluded", False) & ~hit

    def recommend(self, batch_size, pending_experiments=None):
        drop = self._meta["recommended"].copy()
        if not self.allow_recommending_already_measured:
            drop = drop | self._meta["measured"]
        if "excluded" in self._meta:
            drop = drop | self._meta["excluded"]
        if pending_experiments is not None:
            drop = drop.copy()
            drop.loc[self._match(pending_experiments)] = True
        space = self.searchspace.filtered((~drop).to_numpy())
        rec = self.recommender.recommend(batch_size, space, self.objective, self.measurements, pending_experiments)
        self._meta.loc[rec.index, "recommended"] = True
        return rec
